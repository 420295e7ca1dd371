"""Counterpart of the reference's single-step-learner.py for the native path (test mode on synthetic tasks).

Keeps the reference CLI's flag names for everything that touches the hot path (reference utils/args.py:12-192:
--feature_extractor --learn_extractor --adapt_features --classifier --logit_scale --clip_length --frame_size
--batch_size --tasks_per_batch --with_lite --num_lite_samples --gpu --seed --mode --model_path) and the test
loop's call order (reference single-step-learner.py:298-375): per task personalise() -> per target video
{attach_frame_history -> predict -> frame accuracy} -> _reset(), with 'personalise' and per-frame 'inference' timers
(here device-synchronised; the reference's time.time() pairs do not sync). The ORBIT dataset is not available
offline, so tasks come from `synthetic.make_task` in the task_dict layout of reference data/datasets.py:584-597;
`--feature_extractor` additionally accepts resnet18 and `--frame_size` any size (BASELINE.json configs).

Multi-GPU: launched under torchrun, tasks are dealt round-robin to ranks (task i -> rank i % world) and the
per-task statistics are all-gathered at the end; no data-path collective.
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
from .data.utils import attach_frame_history
from .model.few_shot_recognisers import SingleStepFewShotRecogniser


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
    # synthetic-task shape (the dataset-side flags of the reference have no meaning here)
    p.add_argument("--way", type=int, default=5)
    p.add_argument("--shots", type=int, default=5)
    p.add_argument("--frames_per_shot", type=int, default=8)
    p.add_argument("--num_query_videos", type=int, default=4)
    p.add_argument("--frames_per_video", type=int, default=50)
    p.add_argument("--num_test_tasks", type=int, default=8)
    p.add_argument("--results_path", default=None)
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
    return float(v.mean()), float(1.96 * v.std() / math.sqrt(len(v))) if len(v) > 1 else 0.0


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

    def init_model(self):
        a = self.args
        self.model = SingleStepFewShotRecogniser(a.feature_extractor, a.adapt_features, a.classifier, a.clip_length,
                                                 a.batch_size, a.learn_extractor, a.num_lite_samples, a.logit_scale)
        if a.model_path:
            self.model.load_state_dict(torch.load(a.model_path, map_location="cpu"))
        else:
            synthetic.init_parameters_(self.model, seed=a.seed,
                                       film_strength=0.02 if a.feature_extractor == "efficientnet_b0" else 0.1)
            if a.adapt_features:
                from .model.film import get_film_parameters
                self.model.film_generator.initial_film_parameters = get_film_parameters(
                    self.model.film_parameter_names, self.model.feature_extractor)
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
        if "train" in self.args.mode:
            raise NotImplementedError("meta-training (backward through the native extractor) is the next scope row; "
                                      "run with --mode test")
        return self.test()

    def test(self):
        a = self.args
        self.model.set_test_mode(True)
        task_acc, personalise_ms, inference_ms = [], [], []
        with torch.no_grad():
            for t in odist.tasks_for_rank(a.num_test_tasks, self.rank, self.world):
                context_clips, context_labels, videos = self.make_task(t)
                context_labels = context_labels.to(self.device)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                self.model.personalise(context_clips, context_labels)
                torch.cuda.synchronize()
                personalise_ms.append(1e3 * (time.perf_counter() - t0))
                accs = []
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
                  "| %d tasks on %d GPU(s)" % (100 * stats["frame_acc"][0], 100 * stats["frame_acc"][1],
                                               *stats["personalise_ms"], *stats["inference_ms_per_frame"],
                                               stats["num_tasks"], self.world))
            if a.results_path:
                with open(a.results_path, "w") as f:
                    json.dump(stats, f)
        return stats


def main(argv=None):
    args = build_parser().parse_args(argv)
    return Learner(args).run()


if __name__ == "__main__":
    main()
