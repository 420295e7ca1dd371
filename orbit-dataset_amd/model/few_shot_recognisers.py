"""Episodic few-shot recognisers on the MI355X-native runtime.

Drop-in mirror of the reference API (reference model/few_shot_recognisers.py):
  FewShotRecogniser            :46-183   wiring, batched feature extraction, BN-state policy
  SingleStepFewShotRecogniser  :271-473  personalise / personalise_with_lite / predict / predict_a_batch /
                                         _reset / _clear_caches
  MultiStepFewShotRecogniser   :185-269  FineTuner: personalise(context, labels, learning_args) by gradient steps
Same constructor arguments, method names, argument meaning, side effects (`film_dict`, classifier state,
LITE caches) and errors. What differs is where the arithmetic runs: every frame batch goes through
`orbit_extractor_forward` (hand-written gfx950 kernels), FiLM parameters come from one grouped generator
launch, the head is the wavefront-reduction prototype kernel. Frames may arrive on the host (they are moved
per mini-batch, as in the reference) or already resident in HBM.

Meta-training: with autograd enabled the same calls record native tapes and
`loss.backward()` runs the native backward kernels (model/autograd.py) — train-mode BatchNorm, conv dgrad/wgrad,
depthwise / squeeze-excite / SiLU backward, FiLM-generator and set-encoder gradients — for both extractors.
"""
import os

import numpy as np
import torch
import torch.nn as nn

from .. import _lib
from ..data.utils import get_batch_indices, ready_event
from .classifier_heads import create_classifier
from .feature_adapters import FilmParameterGenerator, NullGenerator
from .feature_extractors import create_feature_extractor
from .film import get_film_parameter_sizes, get_film_parameters
from .poolers import MeanPooler
from .set_encoders import NullSetEncoder, SetEncoder


class FewShotRecogniser(nn.Module):
    """Generic few-shot classification model (reference :46-183)."""

    def __init__(self, feature_extractor_name: str, adapt_features: bool, classifier: str, clip_length: int,
                 batch_size: int, learn_extractor: bool, logit_scale: float = 1.0):
        super().__init__()
        self.adapt_features = adapt_features
        self.learn_extractor = learn_extractor
        self.clip_length = clip_length
        self.batch_size = batch_size
        self.logit_scale = logit_scale
        self.test_mode = False

        self.feature_extractor, self.film_parameter_names = create_feature_extractor(
            feature_extractor_name=feature_extractor_name,
            pretrained=True,
            with_film=self.adapt_features,
            learn_extractor=self.learn_extractor,
        )
        self.classifier_name = classifier
        self.classifier = create_classifier(classifier, self.feature_extractor.output_size, self.logit_scale)
        self.frame_pooler = MeanPooler(T=self.clip_length)
        self.device = torch.device("cuda", torch.cuda.current_device()) if torch.cuda.is_available() else None

    def _set_device(self, device):
        self.device = torch.device(device)

    def _send_to_device(self):
        if self.device is None or self.device.type != "cuda":
            raise _lib.OrbitHipError("the native recogniser runs on a HIP device only (got %s)" % (self.device,))
        self.to(self.device)

    # ---- feature extraction ----------------------------------------------------------------------
    def _film_vectors(self, film_dict):
        """(gamma, beta) concatenated in FiLM-slot order, or None."""
        if not film_dict:
            return None
        gen = getattr(self, "film_generator", None)
        if gen is not None and film_dict is getattr(self, "_generated_film_dict", None) and gen.last_film is not None:
            return gen.last_film
        gammas, betas = [], []
        for name, _ in self.feature_extractor.film_slot_modules():
            gammas.append(film_dict[name + ".weight"].reshape(-1))
            betas.append(film_dict[name + ".bias"].reshape(-1))
        return torch.cat(gammas).float().contiguous(), torch.cat(betas).float().contiguous()

    def _get_features(self, clips, film_dict={}, ops_counter=None):
        """clips [n,T,3,H,W] or frames [B,3,H,W] -> frame features [n*T, D] (reference :99-122)."""
        if clips.dim() == 5:
            clips = clips.flatten(end_dim=1)
        clips = self._upload(clips)
        return self.feature_extractor(clips, film=self._film_vectors(film_dict))

    def _get_features_in_batches(self, clips, film_dict={}, ops_counter=None):
        """Same, `batch_size` clips per extractor launch (reference :124-153). Each batch writes its slice of
        one preallocated [N*T, D] tensor instead of a final torch.cat."""
        num_clips = len(clips)
        frames_per_clip = clips.shape[1] if clips.dim() == 5 else 1
        D = self.feature_extractor.output_size
        film = self._film_vectors(film_dict)
        num_batches = int(np.ceil(float(num_clips) / float(self.batch_size)))
        if self.feature_extractor.wants_grad(film):  # meta-training: autograd nodes per batch, concatenated
            parts = []
            for batch in range(num_batches):
                lo, hi = get_batch_indices(batch, num_clips, self.batch_size)
                batch_clips = clips[lo:hi]
                if batch_clips.dim() == 5:
                    batch_clips = batch_clips.flatten(end_dim=1)
                parts.append(self.feature_extractor(self._upload(batch_clips), film=film,
                                                    check_sync=(batch == 0)))
            return torch.cat(parts, dim=0)
        features = torch.empty(num_clips * frames_per_clip, D, device=self.device, dtype=torch.float32)
        for batch in range(num_batches):
            lo, hi = get_batch_indices(batch, num_clips, self.batch_size)
            batch_clips = clips[lo:hi]
            if batch_clips.dim() == 5:
                batch_clips = batch_clips.flatten(end_dim=1)
            batch_clips = self._upload(batch_clips)
            self.feature_extractor(batch_clips, film=film,
                                   out=features[lo * frames_per_clip: hi * frames_per_clip],
                                   check_sync=(batch == 0))
        return features

    def _pool_features(self, features, ops_counter=None):
        return self.frame_pooler(features)

    frame_norm_method = "imagenet"  # statistics applied to 8-bit clips (reference data/datasets.py:82-87)

    def _upload(self, clips):
        """Mini-batch of clips -> the model device. fp32 clips are moved as the reference does (:112,142). 8-bit clips
        ([..., 3, H, W] uint8, decoded but not yet normalised) cross PCIe as bytes and get the reference's
        to_tensor + normalize on the GPU (data/utils.frames_from_uint8): a quarter of the upload, same values."""
        if clips.dtype == torch.uint8:
            from ..data.utils import frames_from_uint8
            return frames_from_uint8(clips, self.device, self.frame_norm_method, channels_last=False)
        return clips.to(self.device, non_blocking=True)

    def _index_to_device(self, idx):
        """Host index array (LITE's numpy permutation) -> int64 tensor on the model device WITHOUT stalling the host: a
        pageable host-to-device copy issued on the compute stream waits for every kernel queued before it (that made
        the LITE step host-bound); an idle side stream performs it at once and the compute stream waits on its event."""
        if isinstance(idx, torch.Tensor) and idx.is_cuda:
            return idx
        t = torch.as_tensor(idx, dtype=torch.int64)
        side = self.__dict__.get("_copy_stream")
        if side is None:
            side = self.__dict__["_copy_stream"] = torch.cuda.Stream(device=self.device)
        main = torch.cuda.current_stream(self.device)
        with torch.cuda.stream(side):
            d = t.to(self.device)
        main.wait_stream(side)
        d.record_stream(main)
        return d

    @staticmethod
    def _take(x, idx_host, idx_dev):
        """Rows of x selected by the same indices given twice: host array for host tensors, device tensor otherwise."""
        return x.index_select(0, idx_dev) if x.is_cuda else x[idx_host]

    def set_test_mode(self, test_mode):
        self.test_mode = test_mode

    def _set_batch_norm_state(self):
        """eval() everywhere, but the extractor in train() (batch-statistics BatchNorm, running-stat updates) when
        meta-training an unfrozen extractor (reference :176-183)."""
        want_train = bool(self.learn_extractor and not self.test_mode)
        if self.training or (self.feature_extractor.training and not want_train):
            self.eval()
        if want_train and not self.feature_extractor.training:
            self.feature_extractor.train()


class MultiStepFewShotRecogniser(FewShotRecogniser):
    """Few-shot model personalised by gradient steps on the context set — the FineTuner (reference :185-269).

    `personalise` adds a fresh linear head and takes `num_grad_steps` optimizer steps on the whole context set
    (gradient accumulated over mini-batches of `batch_size` clips, each scaled by its share of the set); with
    `adapt_features` the FiLM-tagged BatchNorm weights / biases of the extractor are unfrozen, with `learn_extractor`
    everything is. All gradients come from the native backward kernels; test-time BatchNorm stays in eval mode
    (the learner calls set_test_mode(True), multi-step-learner.py:119-123)."""

    def __init__(self, feature_extractor_name: str, adapt_features: bool, classifier: str, clip_length: int,
                 batch_size: int, learn_extractor: bool, logit_scale: float = 1.0):
        FewShotRecogniser.__init__(self, feature_extractor_name, adapt_features, classifier, clip_length, batch_size,
                                   learn_extractor, logit_scale)
        if self.adapt_features:
            from .film import unfreeze_film
            self.film_parameter_sizes = get_film_parameter_sizes(self.film_parameter_names, self.feature_extractor)
            unfreeze_film(self.film_parameter_names, self.feature_extractor)

    def _reset(self):
        self.classifier.reset()

    def personalise(self, context_clips, context_labels, learning_args, ops_counter=None):
        from argparse import Namespace

        from ..optim import init_optimizer
        self._set_batch_norm_state()
        learning_args = dict(learning_args)
        num_grad_steps = learning_args.pop("num_grad_steps")
        learning_rate = learning_args.pop("learning_rate")
        optimizer = learning_args.pop("optimizer")
        loss_fn = learning_args.pop("loss_fn")
        extractor_lr_scale = learning_args.pop("extractor_lr_scale")
        optimizer_kwargs = Namespace(**learning_args)

        num_classes = len(torch.unique(context_labels))
        self.init_classifier(num_classes)
        personalize_optimizer = init_optimizer(self, learning_rate, optimizer, optimizer_kwargs, extractor_lr_scale)

        set_size = len(context_labels)
        num_batches = int(np.ceil(float(set_size) / float(self.batch_size)))
        with torch.enable_grad():
            for _ in range(num_grad_steps):
                for batch in range(num_batches):
                    lo, hi = get_batch_indices(batch, set_size, self.batch_size)
                    batch_features = self._get_features(context_clips[lo:hi])
                    batch_features = self._pool_features(batch_features)
                    batch_logits = self.classifier.predict(batch_features)
                    loss = loss_fn(batch_logits, context_labels[lo:hi].to(self.device))
                    loss = loss * ((hi - lo) / set_size)
                    loss.backward()
                personalize_optimizer.step()
                personalize_optimizer.zero_grad()

    def predict(self, clips, ops_counter=None):
        self._set_batch_norm_state()
        features = self._get_features_in_batches(clips)
        features = self._pool_features(features)
        return self.classifier.predict(features)

    def personalise_with_lite(self, context_clips, context_labels):
        raise NotImplementedError  # the reference leaves this unimplemented as well (:258-259)

    def init_classifier(self, num_classes: int):
        self.classifier.init(num_classes)
        self.classifier.to(self.device)


class SingleStepFewShotRecogniser(FewShotRecogniser):
    """Few-shot model personalised in a single forward step — ProtoNets / CNAPs-style (reference :271-473)."""

    def __init__(self, feature_extractor_name: str, adapt_features: bool, classifier: str, clip_length: int,
                 batch_size: int, learn_extractor: bool, num_lite_samples: int, logit_scale: float = 1.0):
        FewShotRecogniser.__init__(self, feature_extractor_name, adapt_features, classifier, clip_length,
                                   batch_size, learn_extractor, logit_scale)
        self.num_lite_samples = num_lite_samples
        if self.adapt_features:
            self.set_encoder = SetEncoder()
            self.film_parameter_sizes = get_film_parameter_sizes(self.film_parameter_names, self.feature_extractor)
            initial_film_parameters = get_film_parameters(self.film_parameter_names, self.feature_extractor)
            self.film_generator = FilmParameterGenerator(
                self.film_parameter_sizes,
                initial_film_parameters,
                pooled_size=self.set_encoder.output_size,
                hidden_size=self.set_encoder.output_size,
                slot_names=[n for n, _ in self.feature_extractor.film_slot_modules()],
            )
            # The reference snapshots the FiLM layers' BatchNorm weights / biases of the PRETRAINED timm network at
            # construction (`get_film_parameters` clones them, model/film.py:81-87) and keeps the snapshot outside the
            # state_dict (model/feature_adapters.py:55-58); FiLM-replaced BatchNorm parameters never receive a gradient,
            # so in every checkpoint the reference writes they still equal that snapshot. Here nothing is downloaded at
            # construction - the "pretrained" values ARE whatever load_state_dict brings - so the snapshot is re-taken
            # from the extractor after every load (it stays out of the state_dict, as in the reference).
            self.register_load_state_dict_post_hook(SingleStepFewShotRecogniser._refresh_film_snapshot_hook)
        else:
            self.set_encoder = NullSetEncoder()
            self.film_generator = NullGenerator()
        self.film_dict = None
        self._generated_film_dict = None
        self.reps_cache = None
        self.features_cache = None
        # overlap_query: in test mode predict() runs the query clips through the extractor on a second HIP stream, so it
        # overlaps with the support pass personalise() queued on the caller's stream (the two passes are independent
        # given the FiLM parameters; small late layers do not fill the chip on their own: -13 % per efficientnet task).
        # The query clips must then be READY when the second stream starts on them - clips that an earlier, still pending
        # operation on the caller's stream produces after personalise() would be read too early. Modes:
        #   "auto" (default, round 6): overlap exactly when that is known to hold - host-resident clips (their upload is issued
        #          on the second stream) and device tensors carrying a readiness event (`data.utils.mark_ready`: recorded by
        #          whoever produced the clips - `TaskPrefetcher` does it for the tasks it yields - the second stream waits for
        #          it); any other device tensor takes the serial order on the caller's stream;
        #   True   the caller vouches for every query tensor (round 2-5's opt-in);   False  never;
        #   2      ("pipelined") additionally runs the HEAD on the second stream and does NOT join it back: the
        # caller's stream is free to start the next task's personalise() while this task's query pass is still running, so
        # consecutive tasks overlap out of phase (one stream in its HBM-bound early layers while the other is in its small
        # late layers) instead of both passes of a task marching through the same layers together. The returned logits are
        # then produced on the second stream: `logits_ready` (an event) must be waited for - or the device synchronised -
        # before another stream reads them.
        self.overlap_query = "auto"
        # LITE: the H-subset pass of a task's first query batch runs beside the cache pass (_get_features_with_split_batch)
        self.lite_overlap = os.environ.get("ORBIT_LITE_OVERLAP", "1") != "0"
        self._lite_stream = None
        # ... and the query batch's taped pass runs on a third stream from the same fork point, beside the cache pass still in
        # flight (clips on the host: always, their upload is issued on that stream; clips on the device: opt-in like
        # overlap_query, they must be READY - not produced by pending work on the caller's stream - when predict_a_batch()
        # is called): two 200-frame passes overlap in their small late layers
        # as the support / query passes of the test path do. Its running-statistics update is deferred as well (order kept:
        # cache pass, subset, query batch).
        self.lite_query_overlap = False
        self._lite_query_stream = None
        self._lite_fork = None
        self._film_ready = None
        self._configured = None
        self.logits_ready = None

    def refresh_initial_film_parameters(self):
        """Re-take the generator's snapshot of the FiLM layers' BatchNorm weights / biases from the extractor's current
        values (what the reference's constructor does once, on the pretrained network). Called automatically after
        load_state_dict; call it by hand after writing extractor parameters in any other way."""
        if self.adapt_features:
            self.film_generator.initial_film_parameters = get_film_parameters(self.film_parameter_names,
                                                                              self.feature_extractor)

    @staticmethod
    def _refresh_film_snapshot_hook(module, incompatible_keys):
        module.refresh_initial_film_parameters()

    def _reset(self):
        self.film_dict = None
        self._generated_film_dict = None
        self._film_ready = None
        self._configured = None
        self.classifier.reset()

    def _clear_caches(self):
        self.reps_cache = None
        self.features_cache = None
        # Once per task (single-step-learner.py:220): a `lite_query` tape whose graph was never backpropagated - an exception,
        # or a forward run with autograd on only to read the logits - would otherwise keep the persistent buffers busy for the
        # rest of the run and silently disable the query-pass overlap; a fork event of an abandoned task must not start the next
        # task's query pass either. The features predict_a_batch() returned on that path are views of the persistent
        # buffer: they are valid until the next task's predict_a_batch().
        self._lite_fork = None
        release = getattr(self.feature_extractor, "persistent_release", None)
        if release is not None:
            release("lite_query")

    # ---- personalise ---------------------------------------------------------------------------------
    def personalise(self, context_clips, context_labels, ops_counter=None):
        self._set_batch_norm_state()
        # the label set does not depend on the features: its resolution is started here, on a side stream, and the host only
        # waits for the class COUNT when the head kernel of predict() is launched - after both extractor passes are queued
        # (classifier_heads.PendingLabelSet; the reference's torch.unique + .item() loop, classifier_heads.py:96-100, would
        # drain the launch queue once per task). Heads that size workspaces by the class count take the blocking form.
        class_ids = self._label_set(context_labels)
        task_embedding = self._get_task_embedding_in_batches(context_clips, ops_counter)
        self.film_dict = self._generate_film_params(task_embedding, ops_counter)
        if self.overlap_query and not torch.is_grad_enabled() and not self.feature_extractor.training:
            # everything the query pass needs from this stream: the FiLM vectors AND the extractor's parameters inside
            # the native plan. The upload / repack / BatchNorm fold is queued HERE, explicitly, before the event — the
            # first forward below would otherwise queue it after the event, and a query pass on the side stream (whose
            # own sync() sees an up-to-date stamp) could run against half-packed filters on the first task after any
            # parameter change.
            fe = self.feature_extractor
            fe.sync(fe._plan(int(context_clips.shape[-2]), int(context_clips.shape[-1])))
            self._film_ready = torch.cuda.Event()
            self._film_ready.record(torch.cuda.current_stream(self.device))
        context_features = self._get_features_in_batches(context_clips, self.film_dict, ops_counter)
        context_features = self._pool_features(context_features, ops_counter)
        self.classifier.configure(context_features, context_labels, ops_counter, class_ids=class_ids)
        if self.overlap_query == 2:
            self._configured = torch.cuda.Event()
            self._configured.record(torch.cuda.current_stream(self.device))

    def _label_set(self, context_labels):
        get = getattr(self.classifier, "label_set", None)
        if get is not None and context_labels.is_cuda:
            return get(context_labels, self.device)
        return self.classifier.unique_labels(context_labels, self.device)

    def personalise_with_lite(self, context_clips, context_labels):
        """LITE forward (reference :328-343): a random subset of `num_lite_samples` clips is re-encoded on
        every call, the rest comes from the per-task caches; features/labels are reordered by the permutation.
        The permutation comes from np.random, exactly as in the reference (seed numpy to reproduce)."""
        self._set_batch_norm_state()
        # the label SET is permutation-invariant: resolve it from the task's own label tensor (memoised per task; the first
        # query batch of a task resolves it on a side stream, see personalise)
        class_ids = self._label_set(context_labels)
        shuffled_idxs = np.random.permutation(len(context_clips))
        grad_idxs = shuffled_idxs[0:self.num_lite_samples]
        no_grad_idxs = shuffled_idxs[self.num_lite_samples:]
        # one stall-free upload of the permutation; the split-batch helpers slice it on the device
        perm_dev = self._index_to_device(shuffled_idxs)
        self._lite_idx = (grad_idxs, perm_dev[:self.num_lite_samples], no_grad_idxs, perm_dev[self.num_lite_samples:])
        task_embedding = self._get_task_embedding_with_split_batch(context_clips, grad_idxs, no_grad_idxs)
        self.film_dict = self._generate_film_params(task_embedding)
        context_features = self._get_features_with_split_batch(context_clips, self.film_dict, grad_idxs, no_grad_idxs)
        context_features = self._pool_features(context_features)
        labels = self._take(context_labels, shuffled_idxs, perm_dev)
        self._lite_idx = None
        self.classifier.configure(context_features, labels, class_ids=class_ids)

    # ---- task embedding ------------------------------------------------------------------------------
    def _get_task_embedding(self, context_clips, ops_counter=None, aggregation="mean"):
        context_clips = self._upload(context_clips)
        reps = self.set_encoder(context_clips)
        return self.set_encoder.aggregate(reps, aggregation=aggregation)

    def _get_task_embedding_in_batches(self, context_clips, ops_counter=None, aggregation="mean"):
        if isinstance(self.set_encoder, NullSetEncoder):
            return None
        num_clips = len(context_clips)
        frames_per_clip = context_clips.shape[1] if context_clips.dim() == 5 else 1
        num_batches = int(np.ceil(float(num_clips) / float(self.batch_size)))
        if self.set_encoder.wants_grad():  # meta-training without LITE: autograd nodes per batch
            parts = []
            for batch in range(num_batches):
                lo, hi = get_batch_indices(batch, num_clips, self.batch_size)
                parts.append(self.set_encoder(self._upload(context_clips[lo:hi]),
                                              check_sync=(batch == 0)))
            return self.set_encoder.aggregate(parts, aggregation=aggregation)
        reps = torch.empty(num_clips * frames_per_clip, self.set_encoder.output_size, device=self.device,
                           dtype=torch.float32)
        for batch in range(num_batches):
            lo, hi = get_batch_indices(batch, num_clips, self.batch_size)
            batch_clips = self._upload(context_clips[lo:hi])
            if batch_clips.dim() == 5:
                batch_clips = batch_clips.flatten(end_dim=1)
            self.set_encoder(batch_clips, out=reps[lo * frames_per_clip: hi * frames_per_clip],
                             check_sync=(batch == 0))
        return self.set_encoder.aggregate(reps, aggregation=aggregation)

    def _get_task_embedding_with_split_batch(self, context_clips, grad_idxs, no_grad_idxs):
        if isinstance(self.set_encoder, NullSetEncoder):
            return None
        self._set_batch_norm_state()
        if self.reps_cache is None:
            with torch.no_grad():
                self.reps_cache = self._get_task_embedding_in_batches(context_clips, aggregation="none")
        grad_dev, no_grad_dev = self._split_indices(grad_idxs, no_grad_idxs)
        reps_with_grads = self._get_task_embedding(self._take(context_clips, grad_idxs, grad_dev), aggregation="none")
        reps_without_grads = self.reps_cache.index_select(0, no_grad_dev)
        # mean over the concatenation (reference :413 returns a [64] vector here, not [1,64])
        return self.set_encoder.aggregate(torch.cat((reps_with_grads, reps_without_grads)), "mean").reshape(-1)

    def _get_features_with_split_batch(self, context_clips, film_dict, grad_idxs, no_grad_idxs):
        self._set_batch_norm_state()
        grad_dev, no_grad_dev = self._split_indices(grad_idxs, no_grad_idxs)
        fe = self.feature_extractor
        if (self.features_cache is None and self.lite_overlap and fe.training and not film_dict and len(grad_idxs) > 0
                and hasattr(fe, "deferred_stats")):
            # First query batch of a task: the cache pass over the WHOLE context set (no autograd) and the re-encoding of the
            # H-clip subset are independent, and the subset's kernels (16 frames) are launch-bound: it runs on a second stream
            # beside the cache pass (1.6 ms of a 29 ms step when serial). Both passes are train-mode BatchNorm forwards of one
            # plan; the reference updates the running statistics with the cache pass first (:404-408) - the subset's update
            # is deferred and applied after the join, so the statistics end up as in the serial order. The subset is issued
            # FIRST on the host: its tape must not be an allocator block the cache pass frees while its kernels still run.
            h_clips = self._upload(self._take(context_clips, grad_idxs, grad_dev))
            if h_clips.dim() == 5:
                h_clips = h_clips.flatten(end_dim=1)
            h_clips = h_clips.contiguous().float()
            fe.prepare(h_clips.shape[-2], h_clips.shape[-1])  # parameter upload (if stale) on THIS stream, before the fork
            side = self._lite_side_stream()
            fork, done = torch.cuda.Event(), torch.cuda.Event()
            fork.record()
            self._lite_fork = fork  # (predict_a_batch: the query pass may start from here too, lite_query_overlap)
            side.wait_event(fork)
            with fe.deferred_stats(fe) as deferred, _lib.use_stream(side):
                features_with_grads = fe(h_clips, film=None)
            done.record(side)
            with torch.no_grad():
                self.features_cache = self._get_features_in_batches(context_clips, film_dict)
            torch.cuda.current_stream().wait_event(done)
            deferred.apply()
        else:
            if self.features_cache is None:
                with torch.no_grad():
                    self.features_cache = self._get_features_in_batches(context_clips, film_dict)
            features_with_grads = self._get_features(self._take(context_clips, grad_idxs, grad_dev), film_dict)
        if context_clips.dim() == 5 and context_clips.shape[1] > 1:
            # the cache holds frame features [N*T, D]; the reference indexes it with clip indices (:434),
            # which is only meaningful for T == 1 — keep clip granularity here
            T = context_clips.shape[1]
            frame_idx = (no_grad_dev[:, None] * T + torch.arange(T, device=no_grad_dev.device)[None, :]).reshape(-1)
            features_without_grads = self.features_cache.index_select(0, frame_idx)
        else:
            features_without_grads = self.features_cache.index_select(0, no_grad_dev)
        return torch.cat((features_with_grads, features_without_grads))

    def _lite_side_stream(self):
        if self._lite_stream is None:
            self._lite_stream = torch.cuda.Stream(device=self.device)
        return self._lite_stream

    def _split_indices(self, grad_idxs, no_grad_idxs):
        """Device copies of the LITE index arrays: the slices personalise_with_lite prepared, or (when the helpers are
        called on their own, as the reference allows) fresh stall-free uploads."""
        cached = getattr(self, "_lite_idx", None)
        if cached is not None and cached[0] is grad_idxs and cached[2] is no_grad_idxs:
            return cached[1], cached[3]
        return self._index_to_device(grad_idxs), self._index_to_device(no_grad_idxs)

    def _generate_film_params(self, task_embedding, ops_counter=None):
        film_dict = self.film_generator(task_embedding)
        self._generated_film_dict = film_dict
        return film_dict

    # ---- predict ----------------------------------------------------------------------------------------
    def predict(self, target_clips):
        self._set_batch_norm_state()
        ready = ready_event(target_clips) if target_clips.is_cuda else None
        mode = self.overlap_query
        if mode == "auto":  # overlap exactly when the clips are known to be ready: on the host, or marked (data.utils.mark_ready)
            mode = (not target_clips.is_cuda) or ready is not None
        if mode and not torch.is_grad_enabled() and not self.feature_extractor.training:
            side = self.__dict__.get("_query_stream")
            if side is None:
                side = self.__dict__["_query_stream"] = torch.cuda.Stream(device=self.device)
            main = torch.cuda.current_stream(self.device)
            if self._film_ready is not None:
                side.wait_event(self._film_ready)  # FiLM vectors + uploaded parameters: all the query pass needs
            else:  # personalised through another route (sharded / LITE): order after everything queued so far
                side.wait_stream(main)
            if ready is not None:
                side.wait_event(ready)  # whatever produced the clips (an upload on a copy stream, a decode kernel ...)
            if mode == 2 and self._configured is not None:
                # pipelined: extractor, pooling AND head on the second stream; no join (see __init__)
                with torch.cuda.stream(side):
                    target_features = self._get_features_in_batches(target_clips, self.film_dict)
                    side.wait_event(self._configured)  # the head needs this task's class weights
                    logits = self.classifier.predict(self._pool_features(target_features))
                    self.logits_ready = torch.cuda.Event()
                    self.logits_ready.record(side)
                # Everything the second stream's kernels still read was allocated on the caller's stream and is dropped by
                # _reset() / the next personalise() (or, for clips a TaskPrefetcher slot owns, handed back when the next task
                # is requested) while those kernels may still be queued: the class weights, the FiLM vectors of this task
                # (film_dict views and the generator's gamma / beta they are cut from) and the query clips. record_stream
                # makes the caching allocator wait for the second stream before it reuses their memory.
                held = [getattr(self.classifier, "weight", None), getattr(self.classifier, "bias", None), target_clips]
                if self.film_dict:
                    held += list(self.film_dict.values())
                held += list(getattr(self.film_generator, "last_film", None) or ())
                for t in held:
                    if isinstance(t, torch.Tensor) and t.is_cuda:
                        t.record_stream(side)
                logits.record_stream(main)
                return logits
            with torch.cuda.stream(side):
                target_features = self._get_features_in_batches(target_clips, self.film_dict)
            main.wait_stream(side)
            target_features.record_stream(main)
        else:
            target_features = self._get_features_in_batches(target_clips, self.film_dict)
        target_features = self._pool_features(target_features)
        return self.classifier.predict(target_features)

    def predict_video(self, video_frames):
        """Logits for every frame of ONE video, equal to `predict(attach_frame_history(video_frames, clip_length))` (the
        reference's test loop, single-step-learner.py:327-334) but with each frame through the extractor once: the clips
        of a video are sliding windows over its frames, so for clip_length T the reference's tensor repeats every frame
        T times. Frame features are pooled over the windows by orbit_history_mean_pool (same summation order as
        MeanPooler on the expanded clips: bit-identical logits in test mode)."""
        if video_frames.dim() != 4:
            raise ValueError("expected the frames of one video [F,3,H,W], got %s" % (tuple(video_frames.shape),))
        T = int(self.clip_length)
        if T == 1 or torch.is_grad_enabled():
            from ..data.utils import attach_frame_history
            return self.predict(attach_frame_history(video_frames, T))
        self._set_batch_norm_state()
        feats = self._get_features_in_batches(video_frames, self.film_dict)  # [F, D], mini-batches of batch_size frames
        F_, D = feats.shape
        pooled = torch.empty_like(feats)
        if F_ > 0:
            _lib.check(_lib.load().orbit_history_mean_pool(_lib.dptr(feats, torch.float32), F_, T, D, _lib.dptr(pooled),
                                                           _lib.stream_handle()), "orbit_history_mean_pool")
        return self.classifier.predict(pooled)

    def predict_a_batch(self, target_clips):
        self._set_batch_norm_state()
        fe, fork = self.feature_extractor, self._lite_fork
        self._lite_fork = None
        # (clips that live on the HOST are always safe - their upload is issued on the query stream; device-resident clips
        # need the opt-in: they must not be the product of work still pending on the caller's stream)
        resident = (target_clips.is_cuda and self.lite_query_overlap and target_clips.dtype == torch.float32
                    and target_clips.is_contiguous())
        if (fork is not None and self.lite_overlap and (resident or not target_clips.is_cuda) and fe.training
                and not self.film_dict and torch.is_grad_enabled() and len(target_clips) > 0
                and fe.persistent_available("lite_query") and fe.wants_grad(None)
                and fe.in_sync(target_clips.shape[-2], target_clips.shape[-1])):
            if self._lite_query_stream is None:
                self._lite_query_stream = torch.cuda.Stream(device=self.device)
            side = self._lite_query_stream
            side.wait_event(fork)  # after the previous step's backward + this step's parameter upload, NOT after the cache pass
            clips = target_clips
            if not resident:
                with torch.cuda.stream(side), _lib.use_stream(side):
                    clips = self._upload(target_clips).contiguous().float()
                clips.record_stream(torch.cuda.current_stream())  # (the tape's backward reads the frames on the caller's stream)
            if clips.dim() == 5:
                clips = clips.flatten(end_dim=1)
            with fe.deferred_stats(fe) as deferred, fe.persistent_buffers(fe, "lite_query"), _lib.use_stream(side):
                target_features = fe(clips, film=None)
            done = torch.cuda.Event()
            done.record(side)
            torch.cuda.current_stream().wait_event(done)
            deferred.apply()
            return self.classifier.predict(self._pool_features(target_features))
        target_features = self._get_features(target_clips, self.film_dict)
        target_features = self._pool_features(target_features)
        return self.classifier.predict(target_features)
