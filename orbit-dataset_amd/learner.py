"""Counterpart of the reference's single-step-learner.py for the native path (train / test on synthetic tasks).

Keeps the reference CLI's flag names for everything that touches the hot path (reference utils/args.py:12-192:
--feature_extractor --learn_extractor --adapt_features --classifier --logit_scale --clip_length --frame_size
--batch_size --tasks_per_batch --with_lite --num_lite_samples --gpu --seed --mode --model_path) and the test
loop's call order (reference single-step-learner.py:298-375): per task personalise() -> per target video
{attach_frame_history -> predict -> frame accuracy} -> _reset(), with 'personalise' and per-frame 'inference' timers
(here device-synchronised; the reference's time.time() pairs do not sync). The ORBIT dataset is not available
offline, so tasks come from `synthetic.make_task` in the task_dict layout of reference data/datasets.py:584-597;
`--feature_extractor` additionally accepts resnet18 and `--frame_size` any size (BASELINE.json configs).

Training (reference single-step-learner.py:128-243): per task `train_task` or `train_task_with_lite` (per query
batch {personalise_with_lite -> predict_a_batch -> N/(H*tasks_per_batch) * CE + 0.001 * l2 -> backward -> _reset}),
optimizer.step()/zero_grad() every `tasks_per_batch` tasks, optimizer groups as utils/optim.py:11-33 (extractor
parameters in their own group with `extractor_lr_scale`). All gradients come from the native backward kernels
(model/autograd.py); the optimizer is torch.optim on the device-resident parameters. LR schedulers, validation and
checkpoint rotation of the reference loop are not part of the hot path and are left out.

Multi-GPU: launched under torchrun, tasks are dealt round-robin to ranks (task i -> rank i % world). Test mode has no
data-path collective (per-task statistics are all-gathered at the end). Training all-reduces (SUM, RCCL) one flat
gradient bucket per optimizer step: every task's loss already carries 1/tasks_per_batch, so the sum over ranks
reproduces single-GPU accumulation; BatchNorm running statistics are combined at the same point as the reference's
sequential updates would have left them (dist.RunningStatSync).
"""
import argparse
import json
import math
import sys
import time

import numpy as np
import torch

from . import dist as odist
from . import synthetic
from .data.utils import attach_frame_history, unpack_task
from .model.few_shot_recognisers import SingleStepFewShotRecogniser
from .optim import apply_lr_scale, cross_entropy, init_optimizer  # noqa: F401  (re-exported: bench.py, tools)


def build_parser():
    p = argparse.ArgumentParser(description="single-step learner on the MI355X-native recogniser (synthetic tasks)")
    # flags shared with the reference (utils/args.py)
    p.add_argument("--feature_extractor", default="efficientnet_b0", choices=["efficientnet_b0", "resnet18"])
    p.add_argument("--learn_extractor", action="store_true")
    p.add_argument("--adapt_features", action="store_true")
    p.add_argument("--classifier", default="proto", choices=["proto", "proto_cosine", "versa", "mahalanobis", "linear"])
    p.add_argument("--logit_scale", type=float, default=1.0)
    p.add_argument("--clip_length", type=int, default=1)
    p.add_argument("--frame_size", type=int, default=224)
    p.add_argument("--batch_size", type=int, default=256)
    p.add_argument("--tasks_per_batch", type=int, default=16)
    p.add_argument("--with_lite", action="store_true")
    p.add_argument("--num_lite_samples", type=int, default=16)
    p.add_argument("--gpu", type=int, default=0)
    p.add_argument("--seed", type=int, default=synthetic.DEFAULT_SEED)
    p.add_argument("--mode", default="test", choices=["train", "test", "train_test"])
    p.add_argument("--model_path", default=None)
    p.add_argument("--save_model_path", default=None,
                   help="where --mode train / train_test writes model.state_dict() after training (the reference saves "
                        "checkpoint_dir/final.pt, single-step-learner.py:185)")
    # synthetic-task shape (the dataset-side flags of the reference have no meaning here)
    p.add_argument("--way", type=int, default=5)
    p.add_argument("--shots", type=int, default=5)
    p.add_argument("--frames_per_shot", type=int, default=8)
    p.add_argument("--num_query_videos", type=int, default=4)
    p.add_argument("--frames_per_video", type=int, default=50)
    p.add_argument("--num_test_tasks", type=int, default=8)
    p.add_argument("--num_train_tasks", type=int, default=16)
    p.add_argument("--epochs", "-e", type=int, default=1)
    p.add_argument("--num_val_tasks", type=int, default=0,
                   help="validation tasks run after every epoch from --validation_on_epoch on (reference validate(), "
                        "single-step-learner.py:245-296; 0 = no validation)")
    p.add_argument("--validation_on_epoch", type=int, default=1, help="epoch to turn on validation (reference utils/args.py:117)")
    p.add_argument("--save_best_model_path", default=None,
                   help="where the model with the best validation frame accuracy so far is written (reference "
                        "checkpoint_dir/best.pt, single-step-learner.py:290-293)")
    p.add_argument("--learning_rate", "-lr", type=float, default=5e-6)
    p.add_argument("--extractor_lr_scale", type=float, default=1.0)
    p.add_argument("--optimizer", default="adam", choices=["adam", "sgd"])
    p.add_argument("--weight_decay", type=float, default=0.2)
    p.add_argument("--epsilon", type=float, default=1e-6)
    p.add_argument("--p2p_gradients", action="store_true",
                   help="multi-GPU: all-reduce the gradient bucket with the direct peer-to-peer reduce-scatter + all-gather "
                        "(csrc/comm.hip) instead of the backend's ring")
    p.add_argument("--fused_optimizer", action="store_true",
                   help="torch's fused multi-tensor Adam (same update, ~7 ms less host time per efficientnet_b0 step; the "
                        "native plans are re-synchronised from an optimizer step hook, optim.mark_parameters_changed)")
    p.add_argument("--betas", type=float, nargs=2, default=(0.9, 0.98))
    p.add_argument("--momentum", type=float, default=0.0)
    p.add_argument("--print_by_step", action="store_true")
    p.add_argument("--results_path", default=None)
    p.add_argument("--data_root", default=None,
                   help="--mode test: a frame directory laid out like ORBIT (root/<user>/<object>/<clean|clutter>/<video>/*.jpg, "
                        "reference data/datasets.py:139-200). Tasks are the users' (context = clean videos, target = clutter "
                        "videos); frames are decoded with PIL by --num_workers threads, uploaded as 8-bit on a copy stream and "
                        "normalised on the GPU (data/pipeline.TaskPrefetcher) while the previous task runs")
    # dataset-side flags of the reference (utils/args.py:41-73), used with --data_root
    p.add_argument("--test_way_method", default="max", choices=["random", "max"])
    p.add_argument("--test_object_cap", type=int, default=15)
    p.add_argument("--test_context_shot_method", default="max", choices=["specific", "fixed", "random", "max"])
    p.add_argument("--test_target_shot_method", default="max", choices=["specific", "fixed", "random", "max"])
    p.add_argument("--context_shot", type=int, default=5)
    p.add_argument("--target_shot", type=int, default=2)
    p.add_argument("--context_video_type", default="clean", choices=["clean"])
    p.add_argument("--target_video_type", default="clutter", choices=["clutter", "clean"])
    p.add_argument("--subsample_factor", type=int, default=30)
    p.add_argument("--test_context_clip_method", default="uniform", choices=["random", "random_200", "max", "uniform"])
    p.add_argument("--test_target_clip_method", default="random_200", choices=["random", "random_200", "max"])
    p.add_argument("--max_test_tasks", type=int, default=None,
                   help="with --data_root: evaluate only the first N tasks of the user-major order (default: all users)")
    p.add_argument("--num_test_tasks_per_user", type=int, default=1,
                   help="with --data_root: tasks sampled per test user (reference --num_test_tasks, utils/args.py)")
    p.add_argument("--num_workers", type=int, default=4, help="decode threads (reference data/queues.py:34: 4 in test mode)")
    p.add_argument("--frame_norm_method", default="imagenet", choices=["imagenet", "imagenet_inception", "openai_clip"])
    return p


def verify_args(args):
    """reference utils/args.py:203-217"""
    if "train" in args.mode and not args.learn_extractor and not args.adapt_features:
        sys.exit("error: at least one of --learn_extractor and --adapt_features must be used when training")
    if args.frame_size % 1 or args.frame_size < 32:
        sys.exit("error: --frame_size must be >= 32")


def frame_accuracy(logits, label):
    """reference utils/eval_metrics.py:27-36"""
    return (logits.argmax(dim=-1) == label).float().mean().item()


def mean_ci(values):
    """mean and 95 % confidence half-width, as the reference's evaluators report (utils/eval_metrics.py:24-25)."""
    v = np.asarray(values, dtype=np.float64)
    if len(v) == 0:  # a rank (or a run) that saw no task: no statistic (None: json.dump would write a bare NaN, invalid JSON)
        return None, 0.0
    return float(v.mean()), float(1.96 * v.std() / math.sqrt(len(v))) if len(v) > 1 else 0.0


def _shown(stat):
    """(mean, ci) for a print statement: NaN where mean_ci had no data"""
    return (float("nan") if stat[0] is None else stat[0]), stat[1]


class Learner:
    def __init__(self, args):
        self.args = args
        verify_args(args)
        self.rank, self.world, self.local_rank = odist.init_from_env()
        np.random.seed(args.seed)  # the reference leaves numpy unseeded; LITE's permutation needs it
        torch.manual_seed(args.seed)
        index = args.gpu if self.world == 1 else self.local_rank
        torch.cuda.set_device(index)
        self.device = torch.device("cuda", index)
        self.init_model()
        import gc
        gc.collect()
        gc.freeze()  # keep torch + the model out of the cyclic collector's working set (full collections cost ms)

    def init_model(self):
        a = self.args
        self.model = SingleStepFewShotRecogniser(a.feature_extractor, a.adapt_features, a.classifier, a.clip_length,
                                                 a.batch_size, a.learn_extractor, a.num_lite_samples, a.logit_scale)
        if a.model_path:
            # reference single-step-learner.py:300-302. The FiLM generator's gamma0/beta0 snapshot is not part of the file
            # (feature_adapters.py:55-58); the recogniser re-takes it from the loaded extractor (load_state_dict hook)
            self.model.load_state_dict(torch.load(a.model_path, map_location="cpu"))
        else:
            synthetic.init_parameters_(self.model, seed=a.seed,
                                       film_strength=0.02 if a.feature_extractor == "efficientnet_b0" else 0.1)
        self.model._set_device(self.device)
        self.model._send_to_device()

    def make_task(self, index):
        a = self.args
        T = a.clip_length
        per_class = a.shots * a.frames_per_shot
        per_class -= per_class % T
        task = synthetic.make_task(index, a.way, 1, per_class, a.num_query_videos * a.frames_per_video, a.frame_size,
                                   clip_length=T, seed=a.seed)
        # query frames grouped into videos of one object each (reference: target_frames_by_video)
        frames = task["target_clips"][:, 0] if T == 1 else task["target_clips"][:, -1]
        labels = task["target_labels"]
        videos = []
        for v in range(a.num_query_videos):
            sl = slice(v * a.frames_per_video, (v + 1) * a.frames_per_video)
            videos.append((frames[sl], labels[sl]))
        return task["context_clips"], task["context_labels"], videos

    def run(self):
        """train / test / train_test as the reference's run() (single-step-learner.py:136-193): after training, train_test
        tests the FINAL model and then the best-validation checkpoint (`test(self.checkpoint_path_validation)`, :186-188), so
        the model `best_validation` selected is the one a second test block reports (`test_best_validation`)."""
        stats = {}
        if "train" in self.args.mode:
            stats["train"] = self.train()
        if "test" in self.args.mode:
            stats["test"] = self.test()
            best_path = getattr(self.args, "save_best_model_path", None)
            if "train" in self.args.mode and getattr(self, "best_validation", None) is not None and best_path:
                if self.world > 1:  # rank 0 wrote the file (validate()): every rank loads it after the write is complete
                    import torch.distributed as dist
                    dist.barrier()
                self.model.load_state_dict(torch.load(best_path, map_location="cpu"))
                self.model._send_to_device()
                if self.rank == 0:
                    print("testing the best-validation checkpoint %s" % best_path)
                stats["test_best_validation"] = self.test()
        return stats

    # ---- meta-training (reference single-step-learner.py:128-243) ----------------------------------------------------
    def make_train_task(self, index):
        a = self.args
        per_class = a.shots * a.frames_per_shot
        per_class -= per_class % a.clip_length
        return synthetic.make_task(10_000 + index, a.way, 1, per_class, a.num_query_videos * a.frames_per_video,
                                   a.frame_size, clip_length=a.clip_length, seed=a.seed)

    def train_task(self, task):
        a = self.args
        self.model.personalise(task["context_clips"], self._labels_to_device(task["context_labels"]))
        target_logits = self.model.predict(task["target_clips"])
        task_loss = cross_entropy(target_logits, task["target_labels"].to(self.device)) / a.tasks_per_batch
        task_loss = task_loss + 0.001 * self.model.film_generator.regularization_term()
        task_loss.backward()
        self.model._reset()
        return task_loss.detach(), target_logits.detach()

    def train_task_with_lite(self, task):
        a = self.args
        context_clips, context_labels = task["context_clips"], self._labels_to_device(task["context_labels"])
        target_clips, target_labels = task["target_clips"], task["target_labels"]
        self.model._clear_caches()
        task_loss, target_logits = 0, []
        num_clips = len(target_clips)
        num_batches = int(np.ceil(float(num_clips) / float(a.batch_size)))
        for batch in range(num_batches):
            self.model.personalise_with_lite(context_clips, context_labels)
            lo, hi = batch * a.batch_size, min((batch + 1) * a.batch_size, num_clips)
            batch_target_logits = self.model.predict_a_batch(target_clips[lo:hi])
            target_logits.append(batch_target_logits.detach())
            loss_scaling = len(context_labels) / (a.num_lite_samples * a.tasks_per_batch)
            batch_loss = loss_scaling * cross_entropy(batch_target_logits, target_labels[lo:hi].to(self.device))
            batch_loss = batch_loss + 0.001 * self.model.film_generator.regularization_term()
            batch_loss.backward()
            task_loss += batch_loss.detach()
            self.model._reset()
        return task_loss, torch.cat(target_logits)

    def _sync_gradients(self):
        """X3 of SURVEY.md §8e: one all-reduce(SUM) of the persistent flat gradient bucket per optimizer step (+ the
        running statistics, averaged). Parameters that receive no gradient on any rank keep grad None, so the optimizer
        skips them as the single-GPU run and the reference do (dist.GradientBucket)."""
        if self.world == 1:
            return
        self.grad_bucket.sync()
        if self.model.learn_extractor:
            # train-mode BatchNorm: the window's running-statistic updates of all ranks, combined as the reference's
            # sequential updates would have been (dist.RunningStatSync; round 2 averaged the ranks' values)
            self.stat_sync.sync()  # (in-place copies: the plans see the new buffer versions and re-upload them)

    def train(self):
        a = self.args
        self.optimizer = init_optimizer(self.model, a.learning_rate, a.optimizer, a, a.extractor_lr_scale)
        apply_lr_scale(self.optimizer, a.learning_rate)  # constant schedule (the reference's timm scheduler applies it)
        self.grad_bucket = None
        if self.world > 1:
            p2p = None
            if getattr(a, "p2p_gradients", False):
                cap = sum(-(-q.numel() // 64) * 64 for q in self.model.parameters() if q.requires_grad)
                p2p = odist.P2PAllReduce(self.rank, self.world, odist.P2PAllReduce.floats_for_bucket(cap, self.world))
            self.grad_bucket = odist.GradientBucket(self.model.parameters(), p2p=p2p)
            if self.model.learn_extractor:
                self.stat_sync = odist.RunningStatSync(self.model.feature_extractor,
                                                       getattr(self.model.feature_extractor, "bn_momentum", 0.1))
        train_task_fn = self.train_task_with_lite if a.with_lite else self.train_task
        losses, accs, times = [], [], []
        prev = torch.is_grad_enabled()
        torch.set_grad_enabled(True)
        try:
            self.model.set_test_mode(False)
            for epoch in range(a.epochs):
                total_steps = a.num_train_tasks
                # this rank's tasks of the epoch are generated up front (the reference prefetches with DataLoader
                # workers): generating them between steps leaves the intra-op CPU threads spinning against the thread
                # that enqueues kernels, which showed up as random 50-80 ms stalls per task
                mine = [s_ for s_ in range(total_steps) if s_ % self.world == self.rank]
                task_bytes = 4.0 * 3 * a.frame_size ** 2 * a.way * (a.shots * a.frames_per_shot
                                                                   + a.num_query_videos * a.frames_per_video)
                pregen = {s_: self.make_train_task(epoch * total_steps + s_) for s_ in mine} \
                    if task_bytes * len(mine) < 16e9 else {}
                for step in range(total_steps):
                    if step % self.world == self.rank:
                        task = pregen.pop(step, None) or self.make_train_task(epoch * total_steps + step)
                        # LITE draws its subsets from np.random (few_shot_recognisers.py:330); seeding per TASK (the
                        # reference leaves numpy unseeded) makes a run independent of how tasks are dealt to ranks
                        np.random.seed((a.seed + 7919 * (epoch * total_steps + step + 1)) % (2 ** 32))
                        torch.cuda.synchronize()
                        t0 = time.perf_counter()
                        task_loss, logits = train_task_fn(task)
                        torch.cuda.synchronize()
                        times.append(1e3 * (time.perf_counter() - t0))
                        losses.append(float(task_loss))
                        accs.append(frame_accuracy(logits.cpu(), task["target_labels"]))
                        if a.print_by_step:
                            print("epoch [%d/%d][%d/%d] rank %d train loss %.7f frame_acc %.3f  %.1f ms/task"
                                  % (epoch + 1, a.epochs, step + 1, total_steps, self.rank, losses[-1], accs[-1], times[-1]))
                    if (step + 1) % a.tasks_per_batch == 0 or step == total_steps - 1:
                        self._sync_gradients()
                        if not hasattr(self, "gradless_parameters"):  # diagnostic, first optimizer step only
                            self.gradless_parameters = sorted(n for n, p in self.model.named_parameters()
                                                              if p.requires_grad and p.grad is None)
                        self.optimizer.step()
                        if self.grad_bucket is not None:
                            if self.model.learn_extractor:
                                self.stat_sync.begin()  # the next window's running statistics start from here
                            self.grad_bucket.zero_()  # one memset; gradients keep living inside the flat bucket
                        else:
                            self.optimizer.zero_grad()
                if a.num_val_tasks > 0 and (epoch + 1) >= a.validation_on_epoch:
                    self.validate(epoch + 1)
                    torch.set_grad_enabled(True)
                    self.model.set_test_mode(False)
        finally:
            torch.set_grad_enabled(prev)
        if self.grad_bucket is not None and self.grad_bucket.p2p is not None:
            # every optimizer step checked the exchange before it (GradientBucket.sync); this covers the last one: a peer
            # that stalled makes this rank raise instead of saving parameters updated from NaN-poisoned gradients
            torch.cuda.synchronize()
            self.grad_bucket.p2p.raise_on_error()
        if a.save_model_path and self.rank == 0:  # every rank holds the same parameters after the last step
            torch.save({k: v.detach().cpu() for k, v in self.model.state_dict().items()}, a.save_model_path)
        stats = {"loss": mean_ci(losses) if losses else (0.0, 0.0), "frame_acc": mean_ci(accs) if accs else (0.0, 0.0),
                 "ms_per_task": mean_ci(times) if times else (0.0, 0.0), "num_tasks": len(losses),
                 "world_size": self.world, "validation": getattr(self, "validation_history", None),
                 "best_validation": getattr(self, "best_validation", None)}
        if self.rank == 0:
            print("train: loss %.5f | frame_acc %.2f %% | %.1f ms/task | %d tasks on this rank, %d GPU(s)"
                  % (stats["loss"][0], 100 * stats["frame_acc"][0], stats["ms_per_task"][0], stats["num_tasks"],
                     self.world))
        return stats

    def validate(self, epoch=0):
        """The reference's validate() (single-step-learner.py:245-296) on held-out synthetic tasks: test mode, no grad,
        personalise on the context clips, predict every target video on its frame history, per-video frame accuracy; the
        model whose mean per-video accuracy beats the best so far (ValidationEvaluator.is_better, utils/eval_metrics.py:351-357:
        strictly greater, starting from 0) is written to --save_best_model_path. Tasks are dealt to the ranks and the
        per-video accuracies gathered, so every rank takes the same decision."""
        a = self.args
        self.model.set_test_mode(True)
        accs = []
        with torch.no_grad():
            for t in odist.tasks_for_rank(a.num_val_tasks, self.rank, self.world):
                context_clips, context_labels, videos = self.make_task(20_000 + t)
                self.model.personalise(context_clips, self._labels_to_device(context_labels))
                for frames, labels in videos:
                    logits = self.model.predict_video(frames)  # = predict(attach_frame_history(frames, clip_length))
                    accs.append(frame_accuracy(logits.cpu(), labels))
                self.model._reset()
        if self.world > 1:
            import torch.distributed as dist
            gathered = [None] * self.world
            dist.all_gather_object(gathered, accs)
            accs = [x for g in gathered for x in g]
        stat = mean_ci(accs)
        best = getattr(self, "best_validation", (0.0, 0.0))
        better = stat[0] is not None and stat[0] > best[0]
        if better:
            self.best_validation = stat
            if a.save_best_model_path and self.rank == 0:
                torch.save({k: v.detach().cpu() for k, v in self.model.state_dict().items()}, a.save_best_model_path)
        self.validation_history = getattr(self, "validation_history", []) + [(epoch, stat[0], better)]
        if self.rank == 0:
            print("validation (epoch %d): per-video frame_acc %.2f (%.2f) %% over %d videos%s"
                  % (epoch, 100 * _shown(stat)[0], 100 * stat[1], len(accs), " - best so far, model saved" if better else ""))
        return stat

    def _labels_to_device(self, labels):
        """the reference's unpack_task (data/utils.py:42-43) for the context labels; the label set rides along from the host copy"""
        return unpack_task({"context_labels": labels, "target_labels": None, "context_clips": None, "target_clips": None},
                           self.device)[2]

    def test_directory(self):
        """The reference's test loop (single-step-learner.py:298-375) over a JPEG directory: one task per user, personalise
        on the context clips, then per target VIDEO predict on its frame history, frame accuracy per video averaged per
        task. Decode, 8-bit upload and normalisation of task i+1 overlap the extractor work of task i."""
        import random
        from concurrent.futures import ThreadPoolExecutor
        from .data.datasets import UserEpisodicORBITDataset
        from .data.pipeline import DatasetTaskSource, TaskPrefetcher
        a = self.args
        self.model.set_test_mode(True)
        self.model.frame_norm_method = a.frame_norm_method
        # the reference's test queue (data/queues.py:44-46 -> data/datasets.py; pinned by fixture G14): every user's objects,
        # all their videos, context clips sampled uniformly, 200 random clips per target video
        pool = ThreadPoolExecutor(max_workers=max(1, int(a.num_workers)))
        dataset = UserEpisodicORBITDataset(
            a.data_root, a.test_way_method, a.test_object_cap, (a.test_context_shot_method, a.test_target_shot_method),
            (a.context_shot, a.target_shot), (a.context_video_type, a.target_video_type), a.subsample_factor,
            (a.test_context_clip_method, a.test_target_clip_method), a.clip_length, a.frame_size, a.frame_norm_method, [],
            ([], []), True, False, False, None, frames="uint8", rng=random.Random(a.seed + self.rank), decode_pool=pool)
        # TaskSampler order (data/samplers.py:24-30, no shuffle): user 0 x num_tasks, user 1 x num_tasks, ...; dealt to ranks
        # - EVERY user x num_test_tasks_per_user, as the reference's loop visits them (single-step-learner.py:313-314). An
        # optional total cap (--max_test_tasks; default none) is applied BEFORE the tasks are dealt, so the evaluated set does
        # not depend on the world size, and is announced when it truncates the user list
        order = [u for u in range(len(dataset)) for _ in range(a.num_test_tasks_per_user)]
        cap = getattr(a, "max_test_tasks", None)
        if cap is not None and cap < len(order):
            if self.rank == 0:
                print("test (%s): --max_test_tasks %d truncates %d tasks (%d users x %d) - NOT the full test set"
                      % (a.data_root, cap, len(order), len(dataset), a.num_test_tasks_per_user))
            order = order[:cap]
        source = DatasetTaskSource(dataset, order[self.rank::self.world])
        task_acc, personalise_ms, inference_ms, frames = [], [], [], 0
        t_all = time.perf_counter()
        prefetch = TaskPrefetcher(source, self.device, depth=3, frame_norm_method=a.frame_norm_method)
        with torch.no_grad():
            for task in prefetch:
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                self.model.personalise(task["context_clips"], task["context_labels"])
                torch.cuda.synchronize()
                personalise_ms.append(1e3 * (time.perf_counter() - t0))
                accs = []
                target = task["target_clips"][:, 0]  # [M,3,H,W]: the videos' frames, in video order
                for lo, hi in task["target_videos"]:
                    torch.cuda.synchronize()
                    t0 = time.perf_counter()
                    logits = self.model.predict_video(target[lo:hi])  # = predict(attach_frame_history(frames, clip_length))
                    torch.cuda.synchronize()
                    inference_ms.append(1e3 * (time.perf_counter() - t0) / float((hi - lo) * self.model.clip_length))
                    accs.append(frame_accuracy(logits, task["target_labels"][lo:hi]))
                    frames += hi - lo
                task_acc.append(float(np.mean(accs)))
                self.model._reset()
        prefetch.close()
        pool.shutdown()
        wall = time.perf_counter() - t_all
        if self.world > 1:  # per-task results of every rank, as test() gathers them
            import torch.distributed as dist
            gathered = [None] * self.world
            dist.all_gather_object(gathered, (task_acc, personalise_ms, inference_ms, frames))
            task_acc = [x for g in gathered for x in g[0]]
            personalise_ms = [x for g in gathered for x in g[1]]
            inference_ms = [x for g in gathered for x in g[2]]
            frames = sum(g[3] for g in gathered)
        stats = {"frame_acc": mean_ci(task_acc), "personalise_ms": mean_ci(personalise_ms),
                 "inference_ms_per_frame": mean_ci(inference_ms), "num_tasks": len(task_acc), "world_size": self.world,
                 "target_frames": frames, "wall_s": wall, "data_root": a.data_root}
        if self.rank == 0:
            print("test (%s): frame_acc %.2f (%.2f) %% | time to personalise %.2f (%.2f) ms | inference %.4f (%.4f) ms/frame "
                  "| %d tasks, %d target frames in %.1f s incl. JPEG decode (%d threads)"
                  % (a.data_root, 100 * _shown(stats["frame_acc"])[0], 100 * stats["frame_acc"][1], *_shown(stats["personalise_ms"]),
                     *_shown(stats["inference_ms_per_frame"]), stats["num_tasks"], frames, wall, a.num_workers))
            if a.results_path:
                with open(a.results_path, "w") as f:
                    json.dump(stats, f)
        return stats

    def test(self):
        a = self.args
        if getattr(a, "data_root", None):
            return self.test_directory()
        self.model.set_test_mode(True)
        # (synthetic tasks live on the host and are uploaded per mini-batch: in its default mode, overlap_query = "auto", the
        # recogniser runs the query pass of predict() on its own stream for such clips)
        task_acc, personalise_ms, inference_ms = [], [], []
        with torch.no_grad():
            for t in odist.tasks_for_rank(a.num_test_tasks, self.rank, self.world):
                context_clips, context_labels, videos = self.make_task(t)
                context_labels = self._labels_to_device(context_labels)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                self.model.personalise(context_clips, context_labels)
                torch.cuda.synchronize()
                personalise_ms.append(1e3 * (time.perf_counter() - t0))
                accs = []
                for frames, labels in videos:
                    torch.cuda.synchronize()
                    t0 = time.perf_counter()
                    # = predict(attach_frame_history(frames, clip_length)) of the reference loop, frames not duplicated
                    logits = self.model.predict_video(frames)
                    torch.cuda.synchronize()
                    inference_ms.append(1e3 * (time.perf_counter() - t0) / float(len(frames) * self.model.clip_length))
                    accs.append(frame_accuracy(logits.cpu(), labels))
                task_acc.append(float(np.mean(accs)))
                self.model._reset()
        if self.world > 1:
            import torch.distributed as dist
            gathered = [None] * self.world
            dist.all_gather_object(gathered, (task_acc, personalise_ms, inference_ms))
            task_acc = [x for g in gathered for x in g[0]]
            personalise_ms = [x for g in gathered for x in g[1]]
            inference_ms = [x for g in gathered for x in g[2]]
        stats = {"frame_acc": mean_ci(task_acc), "personalise_ms": mean_ci(personalise_ms),
                 "inference_ms_per_frame": mean_ci(inference_ms), "num_tasks": len(task_acc), "world_size": self.world}
        if self.rank == 0:
            print("test: frame_acc %.2f (%.2f) %% | time to personalise %.2f (%.2f) ms | inference %.4f (%.4f) ms/frame "
                  "| %d tasks on %d GPU(s)" % (100 * _shown(stats["frame_acc"])[0], 100 * stats["frame_acc"][1],
                                               *_shown(stats["personalise_ms"]), *_shown(stats["inference_ms_per_frame"]),
                                               stats["num_tasks"], self.world))
            if a.results_path:
                with open(a.results_path, "w") as f:
                    json.dump(stats, f)
        return stats


class MultiStepLearner(Learner):
    """Counterpart of the reference's multi-step-learner.py (FineTuner, test only): per task a fresh copy of the model is
    personalised by `personalize_num_grad_steps` optimizer steps on the context set, then evaluated per target video
    (multi-step-learner.py:132-196)."""

    def init_model(self):
        a = self.args
        from .model.few_shot_recognisers import MultiStepFewShotRecogniser
        self.model = MultiStepFewShotRecogniser(a.feature_extractor, a.adapt_features, a.classifier, a.clip_length,
                                                a.batch_size, a.learn_extractor, a.logit_scale)
        if a.model_path:
            self.model.load_state_dict(torch.load(a.model_path, map_location="cpu"), strict=False)
        else:
            synthetic.init_parameters_(self.model, seed=a.seed,
                                       film_strength=0.02 if a.feature_extractor == "efficientnet_b0" else 0.1)
        self.model._set_device(self.device)
        self.model._send_to_device()
        self.base_state = {k: v.clone() for k, v in self.model.state_dict().items()}

    def run(self):
        return {"test": self.test()}

    def test(self):
        a = self.args
        learning_args = {"num_grad_steps": a.personalize_num_grad_steps, "learning_rate": a.personalize_learning_rate,
                         "extractor_lr_scale": a.personalize_extractor_lr_scale, "loss_fn": cross_entropy,
                         "optimizer": a.personalize_optimizer, "momentum": a.personalize_momentum,
                         "weight_decay": a.personalize_weight_decay, "betas": tuple(a.personalize_betas),
                         "epsilon": a.personalize_epsilon}
        self.model.set_test_mode(True)
        task_acc, personalise_ms, inference_ms = [], [], []
        for t in odist.tasks_for_rank(a.num_test_tasks, self.rank, self.world):
            context_clips, context_labels, videos = self.make_task(t)
            # the finetuner starts every task from the initial parameters (multi-step-learner.py:153-154)
            self.model.load_state_dict(self.base_state, strict=False)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            self.model.personalise(context_clips, context_labels.to(self.device), dict(learning_args))
            torch.cuda.synchronize()
            personalise_ms.append(1e3 * (time.perf_counter() - t0))
            accs = []
            with torch.no_grad():
                for frames, labels in videos:
                    clips = attach_frame_history(frames, a.clip_length)
                    torch.cuda.synchronize()
                    t0 = time.perf_counter()
                    logits = self.model.predict(clips)
                    torch.cuda.synchronize()
                    inference_ms.append(1e3 * (time.perf_counter() - t0) / float(len(clips) * self.model.clip_length))
                    accs.append(frame_accuracy(logits.cpu(), labels))
            task_acc.append(float(np.mean(accs)))
            self.model._reset()
        stats = {"frame_acc": mean_ci(task_acc), "personalise_ms": mean_ci(personalise_ms),
                 "inference_ms_per_frame": mean_ci(inference_ms), "num_tasks": len(task_acc), "world_size": self.world}
        if self.rank == 0:
            print("finetuner test: frame_acc %.2f (%.2f) %% | time to personalise %.2f (%.2f) ms (%d steps) | inference "
                  "%.4f (%.4f) ms/frame | %d tasks" % (100 * _shown(stats["frame_acc"])[0], 100 * stats["frame_acc"][1],
                                                       *_shown(stats["personalise_ms"]), a.personalize_num_grad_steps,
                                                       *_shown(stats["inference_ms_per_frame"]), stats["num_tasks"]))
        return stats


def build_multistep_parser():
    p = build_parser()
    p.set_defaults(classifier="linear", mode="test")
    p.add_argument("--personalize_num_grad_steps", type=int, default=50)
    p.add_argument("--personalize_learning_rate", type=float, default=0.001)
    p.add_argument("--personalize_optimizer", default="adam", choices=["sgd", "adam"])
    p.add_argument("--personalize_weight_decay", type=float, default=0.0)
    p.add_argument("--personalize_extractor_lr_scale", type=float, default=1.0)
    p.add_argument("--personalize_epsilon", type=float, default=1e-8)
    p.add_argument("--personalize_betas", type=float, nargs=2, default=(0.9, 0.999))
    p.add_argument("--personalize_momentum", type=float, default=0.0)
    return p


def main(argv=None):
    args = build_parser().parse_args(argv)
    return Learner(args).run()


def main_multistep(argv=None):
    args = build_multistep_parser().parse_args(argv)
    return MultiStepLearner(args).run()


if __name__ == "__main__":
    main()
