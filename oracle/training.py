"""ORACLE (test infrastructure) — PyTorch-CPU restatement of the LITE meta-training step with autograd.

Restates reference model/few_shot_recognisers.py:176-183 (BatchNorm policy), :328-343 (personalise_with_lite),
:388-437 (split batches: no-grad caches + with-grad LITE subset), :464-473 (predict_a_batch) and
single-step-learner.py:212-243 (Learner.train_task_with_lite: loss scaling N/(H*tasks_per_batch), + 0.001*l2,
backward per query batch). Gradient flow facts it keeps: the head's weight/bias are detached from the support
features (classifier_heads.py:261-263 re-wraps them in nn.Parameter), so gradients reach the extractor only through the
query batch, and the set encoder / FiLM generator only through the FiLM parameters used by that query batch.
Pinned by the golden fixtures G6 (frozen extractor + FiLM) and G8 (unfrozen extractor), which hold gradients recorded
from the reference itself (tests/golden/make_golden.py); tests/test_oracle_golden.py checks it against them.
"""
import numpy as np
import torch
import torch.nn.functional as F

from . import blocks


class LiteTrainer:
    """Wraps an OracleRecogniser (oracle/recogniser.py) whose modules carry requires_grad as the run needs."""

    def __init__(self, recogniser, learn_extractor, tasks_per_batch):
        self.r = recogniser
        self.learn_extractor = learn_extractor
        self.tasks_per_batch = tasks_per_batch
        if not learn_extractor:
            for p in self.r.fe.parameters():
                p.requires_grad = False

    def parameters(self):
        mods = [self.r.fe] + ([self.r.set_encoder, self.r.film_generator] if self.r.set_encoder is not None else [])
        for m in mods:
            yield from m.named_parameters()

    def _set_batch_norm_state(self):
        self.r.fe.train(self.learn_extractor)          # :176-183
        if self.r.set_encoder is not None:
            self.r.set_encoder.eval(), self.r.film_generator.eval()

    def personalise_with_lite(self, context_clips, context_labels):
        r = self.r
        self._set_batch_norm_state()
        perm = np.random.permutation(len(context_clips))
        g_idx, ng_idx = perm[: r.num_lite_samples], perm[r.num_lite_samples:]
        z = None
        if r.set_encoder is not None:
            if r.reps_cache is None:
                with torch.no_grad():
                    r.reps_cache = r._task_embedding_in_batches(context_clips, "none")
            z = torch.cat((r.set_encoder(context_clips[g_idx]), r.reps_cache[ng_idx])).mean(dim=0)
        r.film_dict = r._film(z)
        if r.features_cache is None:
            with torch.no_grad():
                r.features_cache = r._features_in_batches(context_clips, r.film_dict)
        f = torch.cat((r._features(context_clips[g_idx], r.film_dict), r.features_cache[ng_idx]))
        f = blocks.mean_pool(f, r.clip_length)
        r.class_ids, r.W, r.b = blocks.proto_configure(f.detach(), context_labels[perm], r.distance_fn)

    def predict_a_batch(self, target_clips):
        r = self.r
        self._set_batch_norm_state()
        f = blocks.mean_pool(r._features(target_clips, r.film_dict), r.clip_length)
        return blocks.proto_predict(f, r.W, r.b, r.logit_scale, r.distance_fn)

    def train_task_with_lite(self, context_clips, context_labels, target_clips, target_labels, seeds=None):
        """One task of Learner.train_task_with_lite; returns per-batch (logits, loss). `seeds[b]` reseeds numpy before
        batch b's permutation (how the goldens were recorded)."""
        r = self.r
        r.clear_caches()
        out = []
        n, bs = len(target_clips), r.batch_size
        for b in range(int(np.ceil(n / float(bs)))):
            if seeds is not None:
                np.random.seed(seeds[b])
            self.personalise_with_lite(context_clips, context_labels)
            lo, hi = blocks.get_batch_indices(b, n, bs)
            logits = self.predict_a_batch(target_clips[lo:hi])
            scaling = len(context_labels) / (r.num_lite_samples * self.tasks_per_batch)
            loss = scaling * F.cross_entropy(logits, target_labels[lo:hi])
            if r.film_generator is not None:
                loss = loss + 0.001 * r.film_generator.regularization_term()
            loss.backward()
            out.append((logits.detach(), loss.detach()))
            r.reset()
        return out


class FineTuner:
    """Restates reference model/few_shot_recognisers.py:185-269 (MultiStepFewShotRecogniser with the linear head of
    classifier_heads.py:38-79) for the oracle extractor; pinned by golden G11."""

    def __init__(self, fe, adapt_features, learn_extractor, batch_size, logit_scale=1.0):
        self.fe, self.batch_size, self.logit_scale = fe.eval(), batch_size, logit_scale  # test-time: eval BatchNorm
        film = set()
        if adapt_features:
            for n in fe.film_slot_names():
                film.add(n + ".weight"), film.add(n + ".bias")
        for name, p in fe.named_parameters():
            p.requires_grad = bool(learn_extractor or name in film)                     # :196-199
        self.W = self.b = None

    def personalise(self, context_clips, context_labels, num_grad_steps, learning_rate, extractor_lr_scale,
                    betas=(0.9, 0.999), eps=1e-8):
        C = len(torch.unique(context_labels))
        D = self.fe.output_size
        self.W = torch.zeros(C, D, requires_grad=True)                                     # classifier.init, :56-60
        self.b = torch.zeros(C, requires_grad=True)
        # utils/optim.py:28-31 only TAGS the extractor group with lr_scale (timm's scheduler would apply it; no scheduler
        # runs inside personalise), so both groups step at `learning_rate`
        del extractor_lr_scale
        opt = torch.optim.Adam([{"params": [self.W, self.b]}, {"params": list(self.fe.parameters())}],
                               lr=learning_rate, betas=betas, eps=eps)
        n = len(context_labels)
        for _ in range(num_grad_steps):
            for i in range(int(np.ceil(n / float(self.batch_size)))):
                lo, hi = blocks.get_batch_indices(i, n, self.batch_size)
                clips = context_clips[lo:hi]
                f = self.fe(clips.flatten(end_dim=1) if clips.dim() == 5 else clips)
                logits = self.logit_scale * F.linear(f, self.W, self.b)
                loss = F.cross_entropy(logits, context_labels[lo:hi]) * ((hi - lo) / n)
                loss.backward()
            opt.step()
            opt.zero_grad()

    @torch.no_grad()
    def predict(self, clips):
        f = self.fe(clips.flatten(end_dim=1) if clips.dim() == 5 else clips)
        return self.logit_scale * F.linear(f, self.W, self.b)
