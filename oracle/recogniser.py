"""ORACLE (test infrastructure) — PyTorch-CPU restatement of the recogniser flows.

Restates reference model/few_shot_recognisers.py:99-166 (feature batching, pooling), :313-343 (personalise,
personalise_with_lite), :345-451 (task embedding, LITE split batches, FiLM generation) and :453-473 (predict).
Pinned by the golden fixtures G5/G6 (the imported reference run with this oracle's extractor injected).
This is also the timed CPU baseline of bench.py (`cpu_baseline.kind = "port"`).
"""
import numpy as np
import torch
from torch.func import functional_call

from . import blocks, extractors


class OracleRecogniser:
    def __init__(self, feature_extractor_name, adapt_features, classifier, clip_length, batch_size,
                 num_lite_samples=16, logit_scale=1.0):
        if classifier not in ("proto", "proto_cosine", "versa", "mahalanobis"):
            raise ValueError(f"Classifier {classifier} not valid.")
        self.classifier = classifier
        self.fe = extractors.create(feature_extractor_name).eval()
        if classifier == "versa":  # hyper-networks of the Versa head (classifier_heads.py:134-135)
            D = self.fe.output_size
            self.weight_processor = blocks.DenseResidualBlock(D, D).eval()
            self.bias_processor = blocks.DenseResidualBlock(D, 1).eval()
        self.adapt_features = adapt_features
        self.distance_fn = "cosine" if classifier == "proto_cosine" else "euclidean"
        self.clip_length, self.batch_size = clip_length, batch_size
        self.num_lite_samples, self.logit_scale = num_lite_samples, logit_scale
        self.set_encoder = blocks.SetEncoder().eval() if adapt_features else None
        self.film_generator = None
        self.film_dict, self.W, self.b, self.class_ids = {}, None, None, None
        self.reps_cache = self.features_cache = None

    def build_film_generator(self):
        """call after the extractor's parameters are loaded (the reference snapshots gamma0/beta0 at
        construction, few_shot_recognisers.py:283-292 / film.py:81-87)."""
        names = []
        for n in self.fe.film_slot_names():
            names += [n + ".weight", n + ".bias"]
        params = dict(self.fe.named_parameters())
        sizes = {n: len(params[n]) for n in names}
        initial = {n: params[n].detach().clone() for n in names}
        self.film_generator = blocks.FilmParameterGenerator(sizes, initial).eval()
        return self.film_generator

    # ---- features ----
    def _features(self, frames, film_dict):
        if frames.dim() == 5:
            frames = frames.flatten(end_dim=1)
        if film_dict:
            return functional_call(self.fe, film_dict, (frames,))       # :114-115
        return self.fe(frames)

    def _features_in_batches(self, clips, film_dict):
        out, n = [], len(clips)
        for i in range(int(np.ceil(n / float(self.batch_size)))):
            lo, hi = blocks.get_batch_indices(i, n, self.batch_size)
            out.append(self._features(clips[lo:hi], film_dict))
        return torch.cat(out, dim=0)

    def _task_embedding_in_batches(self, clips, aggregation="mean"):
        if self.set_encoder is None:
            return None
        reps, n = [], len(clips)
        for i in range(int(np.ceil(n / float(self.batch_size)))):
            lo, hi = blocks.get_batch_indices(i, n, self.batch_size)
            reps.append(self.set_encoder(clips[lo:hi]))
        return self.set_encoder.aggregate(reps, aggregation)

    def _film(self, z):
        return self.film_generator(z) if self.film_generator is not None else {}

    def _configure(self, f, labels):
        if self.classifier == "versa":
            self.class_ids, self.W, self.b = blocks.versa_configure(f, labels, self.weight_processor, self.bias_processor)
        elif self.classifier == "mahalanobis":
            self.class_ids, self.means, self.precisions, _, _ = blocks.mahalanobis_configure(f, labels)
            self.W = self.b = True  # "configured" markers
        else:
            self.class_ids, self.W, self.b = blocks.proto_configure(f, labels, self.distance_fn)

    def _logits(self, f):
        if self.classifier == "versa":
            return self.logit_scale * (f @ self.W.t() + self.b)
        if self.classifier == "mahalanobis":
            return blocks.mahalanobis_predict(f, self.means, self.precisions, self.logit_scale)
        return blocks.proto_predict(f, self.W, self.b, self.logit_scale, self.distance_fn)

    # ---- API ----
    @torch.no_grad()
    def personalise(self, context_clips, context_labels):
        z = self._task_embedding_in_batches(context_clips)
        self.film_dict = self._film(z)
        f = blocks.mean_pool(self._features_in_batches(context_clips, self.film_dict), self.clip_length)
        self._configure(f, context_labels)

    @torch.no_grad()
    def personalise_with_lite(self, context_clips, context_labels):
        """forward semantics of :328-343 (permutation from np.random, caches, concat order)."""
        perm = np.random.permutation(len(context_clips))
        g_idx, ng_idx = perm[: self.num_lite_samples], perm[self.num_lite_samples:]
        z = None
        if self.set_encoder is not None:
            if self.reps_cache is None:
                self.reps_cache = self._task_embedding_in_batches(context_clips, "none")
            z = torch.cat((self.set_encoder(context_clips[g_idx]), self.reps_cache[ng_idx])).mean(dim=0)
        self.film_dict = self._film(z)
        if self.features_cache is None:
            self.features_cache = self._features_in_batches(context_clips, self.film_dict)
        f = torch.cat((self._features(context_clips[g_idx], self.film_dict), self.features_cache[ng_idx]))
        f = blocks.mean_pool(f, self.clip_length)
        self.class_ids, self.W, self.b = blocks.proto_configure(f, context_labels[perm], self.distance_fn)

    @torch.no_grad()
    def predict(self, target_clips):
        f = blocks.mean_pool(self._features_in_batches(target_clips, self.film_dict), self.clip_length)
        return self._logits(f)

    def reset(self):
        self.film_dict, self.W, self.b, self.class_ids = {}, None, None, None

    def clear_caches(self):
        self.reps_cache = self.features_cache = None
