"""Prototypical-network head on the native wavefront-reduction kernels (csrc/head.hip).

Mirror of the head protocol of reference model/classifier_heads.py:
  configure(features[N,D], labels[N], ops_counter=None)   :232-263 (+ _build_class_reps :94-119)
  predict(features[M,D]) -> logits[M,C]                    :202-230
  reset()                                                  :197-200
Semantics kept from the reference:
  * logit columns follow the ascending unique label values (labels need not be contiguous);
  * euclidean logits are  s * (2 q.mu - mu.mu)  — W = 2 mu, b = -mu.mu (:253-255), the -q.q term is dropped;
  * cosine logits are s * cos(q, W_c) with eps 1e-8;
  * the configured weight/bias are detached from the support features (the reference re-wraps them in
    nn.Parameter, :261-263, which cuts the autograd graph);
  * predict before configure raises AttributeError (:210-211).
The 'proto' / 'proto_cosine' heads are the hot path named by BASELINE.json. 'versa' (CNAPs) and 'mahalanobis' (Simple
CNAPs), the other two single-step recipes of the reference README, are built on csrc/heads_extra.hip;
'linear' is the head of the multi-step finetuner (MultiStepFewShotRecogniser).
"""
import ctypes
import time

import torch
import torch.nn as nn

from .. import _lib


class PendingLabelSet:
    """The label set of a task (ascending unique values = logit column order) being resolved on the device WITHOUT a host
    sync: `orbit_label_set` runs on a side stream as soon as the labels are ready and its count travels to a pinned host
    slot; the host only waits (on that side stream's event, not on the compute queue) when it first needs the NUMBER of
    classes - to shape the logits, i.e. at the head kernel of predict(), after both extractor passes are queued.
    The reference pays a `torch.unique` + `.item()` loop inside configure (model/classifier_heads.py:96-100,246-248), which
    on a GPU drains the launch queue once per task."""

    CAP = 32  # class slots the head kernels are launched over until the count is known (ORBIT users own <= ~20 objects)
    # Pinned host ints the counts travel to: every PendingLabelSet OWNS one slot from a free list until it is resolved or
    # dropped (a shared ring of 64 was overwritten when more than 64 label sets were pending, and resolve() then cut
    # weight / class_ids to another task's class count). A released slot may be handed out while the previous owner's 4-byte
    # copy is still queued: every copy of a device goes through that device's one label stream, so the new owner's copy lands
    # after it, and the new owner only reads the slot after its own `done` event.
    _chunks = {}  # device key -> list of pinned int32[64] tensors
    _free = {}    # device key -> free (chunk, index) pairs
    _lock = __import__("threading").Lock()
    wait_seconds = 0.0  # host time spent blocked in resolve() since the process started (the kernel itself takes microseconds:
    #                     a long wait means the host ran ahead of the stream the labels were produced on)

    @classmethod
    def _take_slot(cls, key):
        with cls._lock:
            free = cls._free.setdefault(key, [])
            if not free:
                chunk = torch.zeros(64, dtype=torch.int32).pin_memory()
                cls._chunks.setdefault(key, []).append(chunk)
                free.extend((chunk, i) for i in range(63, -1, -1))
            return free.pop()

    def _release_slot(self):
        slot, self.slot = self.slot, None
        if slot is not None:
            with PendingLabelSet._lock:
                PendingLabelSet._free.setdefault(self._key, []).append(slot)

    def __del__(self):
        try:
            self._release_slot()
        except Exception:  # (interpreter shutdown)
            pass

    def __init__(self, labels, device):
        cls = PendingLabelSet
        self.slot = None
        self._key = str(device)
        self.labels = labels
        self.slot = cls._take_slot(self._key)
        lab = labels.to(torch.int64).contiguous()
        self.ids = torch.empty(self.CAP, dtype=torch.int64, device=device)
        self.count_dev = torch.empty(1, dtype=torch.int32, device=device)
        main = torch.cuda.current_stream(device)
        side = _label_stream(device)
        ready = torch.cuda.Event()
        ready.record(main)          # the labels may still be in flight on the caller's stream
        side.wait_event(ready)
        _lib.check(_lib.load().orbit_label_set(_lib.dptr(lab, torch.int64), lab.numel(), _lib.dptr(self.ids, torch.int64),
                                               self.CAP, _lib.dptr(self.count_dev, torch.int32),
                                               ctypes.c_void_p(side.cuda_stream)), "orbit_label_set")
        chunk, idx = self.slot
        with torch.cuda.stream(side):
            chunk[idx:idx + 1].copy_(self.count_dev, non_blocking=True)
        self.done = torch.cuda.Event()
        self.done.record(side)
        for t in (lab, self.ids, self.count_dev):
            t.record_stream(side)
        self._resolved = None

    def wait_on(self, stream):
        """device-side: `stream` may read `ids` after this"""
        stream.wait_event(self.done)

    def resolve(self):
        """host-side: the exact class ids [C] (or None when the task has more than CAP classes: the caller then takes the
        exact path). Blocks only until the side stream's kernel + 4-byte copy are done."""
        if self._resolved is None:
            t0 = time.perf_counter()
            self.done.synchronize()
            PendingLabelSet.wait_seconds += time.perf_counter() - t0  # (bench.py reports it beside the host's enqueue time)
            chunk, idx = self.slot
            c = int(chunk[idx])
            self._release_slot()
            self._resolved = (self.ids[:c] if c <= self.CAP else None,)
        return self._resolved[0]


_label_streams = {}


def _label_stream(device):
    key = str(device)
    st = _label_streams.get(key)
    if st is None:
        st = _label_streams[key] = torch.cuda.Stream(device=device)
    return st


class PrototypicalClassifier(nn.Module):
    def __init__(self, logit_scale: float = 1.0, distance_fn: str = "euclidean"):
        super().__init__()
        if distance_fn not in ("euclidean", "cosine"):
            raise ValueError(f"Distance function {distance_fn} not valid.")
        self.logit_scale = logit_scale
        self.distance_fn = distance_fn
        # hook for the support-sharded multi-GPU variant: called with the contiguous [C*D + C] buffer of
        # per-class sums and counts between configure and finalize (an in-place all-reduce(SUM))
        self.partial_reduce = None
        self.reset()

    def reset(self):
        self._pending = None
        self.weight = None
        self.class_ids = None
        if self.distance_fn == "euclidean":
            self.bias = None

    # weight [C, D], bias [C], class_ids [C]: the head state the reference exposes (classifier_heads.py:261-263). When
    # configure() ran on a PendingLabelSet they are views of the CAP-slot buffers, cut to the true class count on first access
    def _resolve(self):
        pend = self.__dict__.get("_pending")
        if pend is None:
            return
        self.__dict__["_pending"] = None
        pending, w_full, b_full, redo = pend
        ids = pending.resolve()
        if ids is None:  # more classes than slots: the exact path, now that a sync has happened anyway
            redo()
            return
        C = int(ids.numel())
        self.__dict__["_weight"], self.__dict__["_class_ids"] = w_full[:C], ids
        if b_full is not None:
            self.__dict__["_bias"] = b_full[:C]
        self._memoise(pending.labels, ids)

    def _get(self, name):
        self._resolve()
        return self.__dict__.get(name)

    weight = property(lambda self: self._get("_weight"), lambda self, v: self.__dict__.__setitem__("_weight", v))
    bias = property(lambda self: self._get("_bias"), lambda self, v: self.__dict__.__setitem__("_bias", v))
    class_ids = property(lambda self: self._get("_class_ids"), lambda self, v: self.__dict__.__setitem__("_class_ids", v))

    @property
    def _cosine(self):
        return 1 if self.distance_fn == "cosine" else 0

    # (data_ptr, version, numel, device) -> (weak reference to the label tensor, class_ids). Shared by the heads of a
    # process, guarded by a lock; entries hold the label tensor only WEAKLY (round 2 pinned up to 256 of them alive) and
    # die with it, so a recycled storage address cannot hit a stale entry
    _unique_cache = {}
    _unique_lock = __import__("threading").RLock()  # re-entrant: `drop` below is a weakref finaliser and may fire on the thread
                                                    # that already holds the lock (a GC pass inside the locked section)

    @classmethod
    def unique_labels(cls, context_labels, device):
        """Ascending unique label values on `device` (column order of the logits, classifier_heads.py:246-248).
        Host-resident labels are reduced on the host (no device sync); device labels need one sync for the class
        count, exactly like the reference's torch.unique(...).item() loop (:96-100). The result is memoised per
        label tensor (storage address + version counter): LITE re-personalises the same task once per query batch
        (single-step-learner.py:220-222), and a resident task set is re-used across epochs."""
        key = (context_labels.data_ptr(), context_labels._version, context_labels.numel(), str(context_labels.device))
        with cls._unique_lock:
            hit = cls._unique_cache.get(key)
            if hit is not None and hit[0]() is context_labels:
                return hit[1]
        if context_labels.is_cuda:
            ids = torch.unique(context_labels.to(torch.int64))
        else:
            ids = torch.unique(context_labels.to(torch.int64)).to(device, non_blocking=True)
        cls._memoise(context_labels, ids)
        return ids

    @classmethod
    def label_set(cls, context_labels, device):
        """`unique_labels` that never blocks the host: the memoised / host-side result when there is one, otherwise a
        PendingLabelSet (device labels the head has not seen: resolved on a side stream, waited for at the head kernel)."""
        if not context_labels.is_cuda:
            return cls.unique_labels(context_labels, device)
        key = (context_labels.data_ptr(), context_labels._version, context_labels.numel(), str(context_labels.device))
        with cls._unique_lock:
            hit = cls._unique_cache.get(key)
            if hit is not None and hit[0]() is context_labels:
                return hit[1]
        return PendingLabelSet(context_labels, device)

    @classmethod
    def register_label_set(cls, device_labels, host_labels):
        """`device_labels` is an upload of `host_labels`: memoise its label set from the host copy (no device work, no sync)."""
        ids = torch.unique(host_labels.to(torch.int64)).to(device_labels.device, non_blocking=True)
        cls._memoise(device_labels, ids)
        return ids

    @classmethod
    def _memoise(cls, context_labels, ids):
        import weakref
        key = (context_labels.data_ptr(), context_labels._version, context_labels.numel(), str(context_labels.device))
        cache, lock = cls._unique_cache, cls._unique_lock

        def drop(ref, key=key):  # the label tensor died: its address may be handed out again
            with lock:
                if cache.get(key, (None,))[0] is ref:
                    del cache[key]
        with lock:
            if len(cache) > 256:
                cache.clear()
            cache[key] = (weakref.ref(context_labels, drop), ids)

    def configure(self, context_features, context_labels, ops_counter=None, frames_per_clip: int = 1,
                  class_ids=None):
        """context_features: [N*frames_per_clip, D] (clip-major); context_labels: [N].
        class_ids: optional precomputed `unique_labels` (lets the caller take the count before it queues the
        extractor work, and gives the global label set when the support set is sharded over ranks)."""
        _lib.require_gpu()
        T = int(frames_per_clip)
        assert context_features.size(0) == context_labels.size(0) * T, \
            "context features and labels are different sizes!"
        feats = context_features.detach().contiguous().float()
        dev = feats.device
        labels = context_labels.to(device=dev, dtype=torch.int64).contiguous()
        if class_ids is None:
            class_ids = self.unique_labels(labels, dev)
        pending = None
        if isinstance(class_ids, PendingLabelSet):
            if self.partial_reduce is not None or feats.shape[0] == 0:  # the sharded exchange needs the exact payload
                ids = class_ids.resolve()
                class_ids = ids if ids is not None else self.unique_labels(labels, dev)
            else:
                pending = class_ids
                pending.wait_on(torch.cuda.current_stream(dev))
                class_ids = pending.ids  # CAP slots; the true count is read when weight / bias are first used
        C, (NT, D) = int(class_ids.numel()), feats.shape
        N = NT // T
        payload = torch.empty(C * D + C, device=dev, dtype=torch.float32)
        sums, counts = payload[: C * D], payload[C * D:]
        lib, st = _lib.load(), _lib.stream_handle()
        if N > 0:
            _lib.check(lib.orbit_proto_configure(_lib.dptr(feats, torch.float32), _lib.dptr(labels, torch.int64),
                                                 _lib.dptr(class_ids, torch.int64), 1, N, T, D, C,
                                                 _lib.dptr(sums), _lib.dptr(counts), st), "orbit_proto_configure")
        else:  # an empty local shard contributes zeros to the all-reduce
            payload.zero_()
        if self.partial_reduce is not None:
            self.partial_reduce(payload)
        weight = torch.empty(C, D, device=dev, dtype=torch.float32)
        bias = None if self._cosine else torch.empty(C, device=dev, dtype=torch.float32)
        _lib.check(lib.orbit_proto_finalize(_lib.dptr(sums), _lib.dptr(counts), 1, C, D, self._cosine,
                                            _lib.dptr(weight), _lib.dptr(bias), st), "orbit_proto_finalize")
        if pending is not None:
            def redo(feats=feats, labels=labels, T=T):
                self.configure(feats, labels, frames_per_clip=T, class_ids=self.unique_labels(labels, dev))
            self.__dict__["_pending"] = (pending, weight, bias, redo)
            return
        self.weight = weight
        self.class_ids = class_ids
        if not self._cosine:
            self.bias = bias

    def predict(self, features, ops_counter=None, frames_per_clip: int = 1, return_argmax: bool = False):
        if self.weight is None or (self.distance_fn == "euclidean" and self.bias is None):
            raise AttributeError("Weight and/or bias not set - is model personalised?")
        _lib.require_gpu()
        T = int(frames_per_clip)
        if features.requires_grad and torch.is_grad_enabled() and not return_argmax:
            # meta-training: the gradient flows into the query features only (weight/bias are constants, :261-263)
            from .autograd import ProtoPredictFunction
            return ProtoPredictFunction.apply(features, self.weight, None if self._cosine else self.bias, T,
                                              float(self.logit_scale), self._cosine)
        q = features.detach().contiguous().float()
        MT, D = q.shape
        M = MT // T
        C = self.weight.size(0)
        logits = torch.empty(M, C, device=q.device, dtype=torch.float32)
        argmax = torch.empty(M, device=q.device, dtype=torch.int32) if return_argmax else None
        if M > 0:
            _lib.check(_lib.load().orbit_proto_predict(
                _lib.dptr(q, torch.float32), _lib.dptr(self.weight), _lib.dptr(None if self._cosine else self.bias),
                1, M, T, D, C, float(self.logit_scale), self._cosine, _lib.dptr(logits), _lib.dptr(argmax),
                _lib.stream_handle()), "orbit_proto_predict")
        return (logits, argmax) if return_argmax else logits


class LinearClassifier(nn.Module):
    """Linear head of the multi-step finetuner (reference classifier_heads.py:38-79): `init(num_classes)` creates zero
    weight / bias Parameters per task, `predict` is s * F.linear on the prototype-distance kernel (euclidean form)."""

    def __init__(self, feat_dim, logit_scale: float = 1.0):
        super().__init__()
        self.feat_dim = feat_dim
        self.logit_scale = logit_scale

    def init(self, num_classes: int):
        self.weight = nn.Parameter(torch.zeros(num_classes, self.feat_dim), requires_grad=True)
        self.bias = nn.Parameter(torch.zeros(num_classes), requires_grad=True)

    def predict(self, features, ops_counter=None):
        _lib.require_gpu()
        from .autograd import LinearPredictFunction
        if torch.is_grad_enabled() and (features.requires_grad or self.weight.requires_grad):
            return LinearPredictFunction.apply(features, self.weight, self.bias, float(self.logit_scale))
        with torch.no_grad():
            return LinearPredictFunction.apply(features, self.weight, self.bias, float(self.logit_scale))

    def reset(self):
        self.weight = None
        self.bias = None


def _class_means(features, labels, class_ids, T=1):
    """Per-class means [C, D] through the prototype kernels (ascending row order, fused clip pooling)."""
    lib, st = _lib.load(), _lib.stream_handle()
    feats = features.detach().contiguous().float()
    C, D = int(class_ids.numel()), feats.shape[1]
    N = feats.shape[0] // T
    sums = torch.empty(C, D, device=feats.device)
    counts = torch.empty(C, device=feats.device)
    _lib.check(lib.orbit_proto_configure(_lib.dptr(feats), _lib.dptr(labels, torch.int64), _lib.dptr(class_ids, torch.int64),
                                         1, N, T, D, C, _lib.dptr(sums), _lib.dptr(counts), st), "orbit_proto_configure")
    means = torch.empty(C, D, device=feats.device)
    _lib.check(lib.orbit_proto_finalize(_lib.dptr(sums), _lib.dptr(counts), 1, C, D, 1, _lib.dptr(means), _lib.dptr(None),
                                        st), "orbit_proto_finalize")
    return means.mul_(0.5)  # the kernel writes the prototype-head weight 2*mu (exact halving)


class VersaClassifier(nn.Module):
    """CNAPs head (reference classifier_heads.py:121-180): two hyper-networks map each class mean to the weight row and
    the bias of a linear layer; predict is s * (q . W^T + b). As in the reference the generated weight / bias are
    re-wrapped (detached) before use (:179-180)."""

    unique_labels = PrototypicalClassifier.unique_labels

    def __init__(self, in_size, logit_scale: float = 1.0):
        super().__init__()
        from .mlps import DenseResidualBlock
        self.logit_scale = logit_scale
        self.weight_processor = DenseResidualBlock(in_size, in_size)
        self.bias_processor = DenseResidualBlock(in_size, 1)
        self.reset()

    def reset(self):
        self.weight = None
        self.bias = None
        self.class_ids = None

    def configure(self, context_features, context_labels, ops_counter=None, class_ids=None):
        _lib.require_gpu()
        assert context_features.size(0) == context_labels.size(0), "context features and labels are different sizes!"
        dev = context_features.device
        labels = context_labels.to(device=dev, dtype=torch.int64).contiguous()
        if class_ids is None:
            class_ids = PrototypicalClassifier.unique_labels(labels, dev)
        means = _class_means(context_features, labels, class_ids)
        C = means.shape[0]
        weight, bias = [], []
        for lo in range(0, C, 16):  # the dense-rows kernel takes up to 16 classes per launch
            weight.append(self.weight_processor(means[lo:lo + 16]))
            bias.append(self.bias_processor(means[lo:lo + 16]))
        self.weight = torch.cat(weight, dim=0)
        self.bias = torch.cat(bias, dim=0).reshape(C)
        self.class_ids = class_ids

    def predict(self, target_features, ops_counter=None):
        if self.weight is None or self.bias is None:
            raise AttributeError("Weight and/or bias not set - is model personalised?")
        _lib.require_gpu()
        if target_features.requires_grad and torch.is_grad_enabled():
            from .autograd import ProtoPredictFunction
            return ProtoPredictFunction.apply(target_features, self.weight, self.bias, 1, float(self.logit_scale), 0)
        q = target_features.detach().contiguous().float()
        M, D = q.shape
        C = self.weight.size(0)
        logits = torch.empty(M, C, device=q.device, dtype=torch.float32)
        if M > 0:
            _lib.check(_lib.load().orbit_proto_predict(_lib.dptr(q), _lib.dptr(self.weight), _lib.dptr(self.bias), 1, M, 1,
                                                       D, C, float(self.logit_scale), 0, _lib.dptr(logits), _lib.dptr(None),
                                                       _lib.stream_handle()), "orbit_proto_predict")
        return logits


class MahalanobisClassifier(nn.Module):
    """Simple CNAPs head (reference classifier_heads.py:265-368): class means + regularised class covariances inverted
    to precisions; logits are negative squared Mahalanobis distances."""

    unique_labels = PrototypicalClassifier.unique_labels

    def __init__(self, logit_scale: float = 1.0):
        super().__init__()
        self.logit_scale = logit_scale
        self.reset()

    def reset(self):
        self.means = None
        self.precisions = None
        self.task_mean = None
        self.task_precision = None
        self.class_ids = None

    def configure(self, context_features, context_labels, ops_counter=None, class_ids=None):
        _lib.require_gpu()
        assert context_features.size(0) == context_labels.size(0), "context features and labels are different sizes!"
        feats = context_features.detach().contiguous().float()
        dev = feats.device
        labels = context_labels.to(device=dev, dtype=torch.int64).contiguous()
        if class_ids is None:
            class_ids = PrototypicalClassifier.unique_labels(labels, dev)
        N, D = feats.shape
        C = int(class_ids.numel())
        lib = _lib.load()
        ws = torch.empty(lib.orbit_mahalanobis_workspace_bytes(N, 1, D, C), dtype=torch.uint8, device=dev)
        means = torch.empty(C, D, device=dev)
        task_mean = torch.empty(D, device=dev)
        precisions = torch.empty(C, D, D, device=dev)
        task_precision = torch.empty(D, D, device=dev)
        _lib.check(lib.orbit_mahalanobis_configure(_lib.dptr(feats), _lib.dptr(labels, torch.int64),
                                                   _lib.dptr(class_ids, torch.int64), N, D, C, _lib.dptr(means),
                                                   _lib.dptr(task_mean), _lib.dptr(precisions), _lib.dptr(task_precision),
                                                   _lib.dptr(ws, torch.uint8), ws.numel(), _lib.stream_handle()),
                   "orbit_mahalanobis_configure")
        self.means, self.task_mean, self.precisions, self.task_precision = means, task_mean, precisions, task_precision
        self.class_ids = class_ids

    def predict(self, target_features, ops_counter=None):
        if self.means is None or self.precisions is None:
            raise AttributeError("Means and/or precisions not set - is model personalised?")
        _lib.require_gpu()
        if target_features.requires_grad and torch.is_grad_enabled() and target_features.shape[0] > 0:
            from .autograd import MahalanobisPredictFunction
            return MahalanobisPredictFunction.apply(target_features, self.means, self.precisions, float(self.logit_scale))
        q = target_features.detach().contiguous().float()
        M, D = q.shape
        C = self.means.size(0)
        lib = _lib.load()
        logits = torch.empty(M, C, device=q.device, dtype=torch.float32)
        if M > 0:
            ws = torch.empty(lib.orbit_mahalanobis_workspace_bytes(2, M, D, C), dtype=torch.uint8, device=q.device)
            _lib.check(lib.orbit_mahalanobis_predict(_lib.dptr(q), _lib.dptr(self.means), _lib.dptr(self.precisions), M, D,
                                                     C, float(self.logit_scale), _lib.dptr(logits),
                                                     _lib.dptr(ws, torch.uint8), ws.numel(), _lib.stream_handle()),
                       "orbit_mahalanobis_predict")
        return logits


def create_classifier(classifier: str, feat_dim: int, logit_scale: float):
    """Head factory used by FewShotRecogniser.__init__ (reference few_shot_recognisers.py:70-84)."""
    if classifier == "proto":
        return PrototypicalClassifier(logit_scale)
    if classifier == "proto_cosine":
        return PrototypicalClassifier(logit_scale, distance_fn="cosine")
    if classifier == "versa":
        return VersaClassifier(feat_dim, logit_scale)
    if classifier == "mahalanobis":
        return MahalanobisClassifier(logit_scale)
    if classifier == "linear":
        return LinearClassifier(feat_dim, logit_scale)
    raise ValueError(f"Classifier {classifier} not valid.")
