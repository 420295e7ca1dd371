"""Multi-GPU forms of the episodic path: one process per GPU, torch.distributed (backend "nccl" = RCCL over xGMI
on ROCm; "gloo" on CPU for the tests).

The reference is single-process (no collectives anywhere). The path shards three ways:
  1. task-parallel   tasks are independent units -> task i runs on rank i % world; NO data-path collective
                     (frame-accuracy counts are reduced after the fact). This is bench.py's N>1 form (weak scaling).
  2. support-sharded ONE task's support frames are split over ranks; each rank extracts features for its slice and
                     produces per-class partial sums [C][D] + counts [C] (orbit_proto_configure's outputs, 25.6 KB at
                     C=5, D=1280); one all-reduce(SUM) of that payload gives every rank identical prototypes.
                     Exact for eval-mode BatchNorm only (test mode or frozen extractor). With FiLM adaptation the
                     set-encoder embedding sums (64 floats + count) are reduced the same way.
  3. query-sharded   query frames are independent given (W, b, FiLM) -> split, no collective (optional gather).
"""
import os

import torch
import torch.distributed as dist


def init_from_env(backend=None):
    """Initialise the default process group from RANK / WORLD_SIZE / LOCAL_RANK / MASTER_* (torchrun). Returns
    (rank, world, local_rank). Single-process when WORLD_SIZE is unset or 1."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:  # ORBIT_DIST_BACKEND=gloo lets several ranks share one GPU (RCCL refuses that)
            backend = os.environ.get("ORBIT_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        kwargs = {}
        if backend == "nccl":
            torch.cuda.set_device(local_rank)
            kwargs["device_id"] = torch.device("cuda", local_rank)
        dist.init_process_group(backend, rank=rank, world_size=world, **kwargs)
    return rank, world, local_rank


def tasks_for_rank(num_tasks, rank, world):
    """Task-parallel assignment: task i -> rank i % world."""
    return range(rank, num_tasks, world)


def shard_bounds(n, rank, world):
    """Contiguous balanced split of n items: the first n % world ranks get one extra item."""
    q, r = divmod(n, world)
    lo = rank * q + min(rank, r)
    return lo, lo + q + (1 if rank < r else 0)


def allreduce_sum_(tensor, group=None):
    """In-place all-reduce(SUM); a no-op without a process group. Used as PrototypicalClassifier.partial_reduce."""
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(tensor, op=dist.ReduceOp.SUM, group=group)
    return tensor


class P2PAllReduce:
    """Peer-to-peer all-reduce(SUM) of device tensors over IPC-mapped inboxes (csrc/comm.hip).

    Small tensors (<= ONE_SHOT_FLOATS; X1/X2: prototype sums, embedding sums): every rank pushes its payload into a slot of
    every peer's inbox over xGMI, raises a flag, and sums the world slots of its own inbox in rank order - one hop of
    latency instead of a ring's 2 (N-1). Large tensors (X3: the flat gradient bucket; pass max_floats >= 2 * numel / world):
    direct reduce-scatter + all-gather - rank p sums shard p of all ranks and pushes the sum back, two hops with all links
    busy. Either way every element is summed once in rank order: bit-identical results on all ranks.
    The 64-byte IPC handles are exchanged once through the torch.distributed group (any backend). Tensors that do not fit
    (or are not contiguous fp32 on the device) fall back to torch.distributed.all_reduce."""

    ONE_SHOT_FLOATS = 32768

    def __init__(self, rank, world, max_floats=16384, group=None):
        import ctypes

        from . import _lib
        self._lib, self._ct = _lib, ctypes
        self.rank, self.world, self.group, self.max_floats = rank, world, group, int(max_floats)
        lib = _lib.load()
        h = ctypes.c_void_p()
        _lib.check(lib.orbit_p2p_create(rank, world, self.max_floats, ctypes.byref(h)), "orbit_p2p_create")
        self.handle = h
        # 1 uncached / 2 fine-grained device memory (coherent while kernels run); 3 = coarse-grained, only with
        # ORBIT_P2P_ALLOW_COARSE=1 and only safe when all ranks share one GPU's L2
        self.memory_kind = int(lib.orbit_p2p_memory_kind(h))
        mine = ctypes.create_string_buffer(64)
        _lib.check(lib.orbit_p2p_export(h, mine), "orbit_p2p_export")
        gathered = [None] * world
        dist.all_gather_object(gathered, bytes(mine.raw), group=group)
        table = ctypes.create_string_buffer(b"".join(gathered), 64 * world)
        _lib.check(lib.orbit_p2p_connect(h, table), "orbit_p2p_connect")
        dist.barrier(group=group)  # every inbox is mapped everywhere before the first push

    def __call__(self, tensor):
        t = tensor
        n = t.numel()
        ok = t.is_cuda and t.dtype == torch.float32 and t.is_contiguous() and n > 0
        lib = self._lib.load()
        if ok and n <= min(self.max_floats, self.ONE_SHOT_FLOATS):
            self._lib.check(lib.orbit_p2p_allreduce_sum(self.handle, self._ct.c_void_p(t.data_ptr()), n,
                                                        self._lib.stream_handle()), "orbit_p2p_allreduce_sum")
        elif ok and 2 * (-(-n // self.world) + 1) <= self.max_floats:
            self._lib.check(lib.orbit_p2p_allreduce_sum_sharded(self.handle, self._ct.c_void_p(t.data_ptr()), n,
                                                                self._lib.stream_handle()),
                            "orbit_p2p_allreduce_sum_sharded")
        else:
            return allreduce_sum_(tensor, self.group)
        return tensor

    @staticmethod
    def floats_for_bucket(numel, world):
        """max_floats that lets a bucket of `numel` floats take the sharded form"""
        return 2 * (-(-int(numel) // world) + 2)

    def error(self):
        """0 = no flag wait has timed out among the all-reduces that have completed; k > 0 = waiting for rank (k-1) % 100
        timed out, and that call's tensor was filled with NaN instead of a partial sum. Does not synchronise the device."""
        return int(self._lib.load().orbit_p2p_error(self.handle))

    def raise_on_error(self):
        """Called once per optimizer step (GradientBucket.sync) / per sharded personalise: a stalled peer becomes an
        exception on the host instead of NaN gradients feeding optimizer.step()."""
        e = self.error()
        if e:
            raise RuntimeError("P2P all-reduce: rank %d timed out waiting for rank %d's flag (%s phase); the affected buffer "
                               "was poisoned with NaN" % (self.rank, (e - 1) % 100, "all-gather" if e > 100 else "reduce"))

    def close(self):
        if getattr(self, "handle", None):
            self._lib.load().orbit_p2p_destroy(self.handle)
            self.handle = None


class SupportSharding:
    """Attach to a SingleStepFewShotRecogniser to personalise ONE task with its support clips split over ranks.
    `p2p` (a P2PAllReduce) serves the small exchange steps - prototype sums + counts, set-encoder embedding sums - with
    the one-shot peer-to-peer kernel; without it they go through torch.distributed (RCCL on a multi-GPU node)."""

    def __init__(self, rank, world, group=None, p2p=None):
        self.rank, self.world, self.group, self.p2p = rank, world, group, p2p

    def bounds(self, n):
        return shard_bounds(n, self.rank, self.world)

    def reduce_(self, tensor):
        if self.p2p is not None:
            self.p2p.raise_on_error()  # exchanges completed so far (host-mapped word, no device sync)
            return self.p2p(tensor)
        return allreduce_sum_(tensor, self.group)


@torch.no_grad()
def personalise_support_sharded(model, context_clips, context_labels, sharding):
    """personalise() with the support set sharded over ranks (form 2). Every rank passes the FULL label vector
    (tiny) and either the full clip tensor or at least its own slice [lo:hi) of it; features are extracted only
    for the local slice. After the call every rank holds identical classifier weights."""
    model._set_batch_norm_state()
    N = len(context_labels)
    lo, hi = sharding.bounds(N)
    class_ids = model.classifier.unique_labels(context_labels, model.device)  # global label set
    local_clips = context_clips[lo:hi] if len(context_clips) == N else context_clips
    local_labels = context_labels[lo:hi]
    z = None
    if model.adapt_features:
        reps = model._get_task_embedding_in_batches(local_clips, aggregation="none")
        # sum of the local embeddings, reduced over ranks, divided by the global frame count
        total = torch.zeros(reps.shape[1] + 1, device=reps.device, dtype=torch.float32)
        if reps.shape[0] > 0:
            total[:-1] = model.set_encoder.aggregate(reps, "mean").reshape(-1) * reps.shape[0]
            total[-1] = reps.shape[0]
        sharding.reduce_(total)
        z = (total[:-1] / total[-1]).reshape(1, -1)
    model.film_dict = model._generate_film_params(z)
    feats = model._get_features_in_batches(local_clips, model.film_dict)
    T = local_clips.shape[1] if local_clips.dim() == 5 else 1
    model.classifier.partial_reduce = sharding.reduce_
    try:
        model.classifier.configure(feats, local_labels, frames_per_clip=T, class_ids=class_ids)
    finally:
        model.classifier.partial_reduce = None


@torch.no_grad()
def predict_query_sharded(model, target_clips, sharding, gather=True):
    """predict() with the query clips split over ranks (form 3). Returns the full [M, C] logits on every rank when
    `gather`, else (local logits, (lo, hi))."""
    M = len(target_clips)
    lo, hi = sharding.bounds(M)
    local = model.predict(target_clips[lo:hi])
    if not gather or sharding.world == 1:
        return local if gather else (local, (lo, hi))
    C = local.shape[1]
    full = torch.zeros(M, C, device=local.device, dtype=local.dtype)
    full[lo:hi] = local
    sharding.reduce_(full)  # disjoint slices: a SUM all-reduce is an all-gather for ragged shards
    return full


class GradientBucket:
    """Persistent flat gradient bucket of the task-parallel training step (X3 of SURVEY §8e): ONE all-reduce(SUM) per
    optimizer step on ONE contiguous buffer the gradients already live in - no per-step torch.cat and no copy-back.

    Which parameters receive a gradient is a property of the configuration, not of the data (FiLM-replaced BatchNorm
    weights, the detached Versa hyper-networks, ... never do). The first `sync()` therefore all-reduces (MAX) a presence
    mask: parameters that have a gradient on NO rank keep `grad = None` - the optimizer skips them exactly as the
    single-process run and the reference do (zero-filling them would let Adam's L2 weight decay move them); parameters
    with a gradient on at least one rank get a view into the flat bucket as their `.grad` (zero where this rank had
    none), and autograd accumulates into those views in place from then on. Call `zero_()` instead of
    `optimizer.zero_grad()`: one memset, views stay attached.
    """

    def __init__(self, params, group=None, p2p=None, collective_layout_check=True):
        self.params = [p for p in params if p.requires_grad]
        self.group = group
        self.p2p = p2p         # a P2PAllReduce sized with floats_for_bucket(): direct RS + AG instead of the backend's ring
        # The decision to (re)build the layout must be the SAME on every rank: _build() issues a collective of its own, so a
        # rank that rebuilds alone would pair its mask all-reduce with its peers' bucket all-reduce. Every
        # sync() therefore first all-reduces (MAX) a one-element "my layout is stale" flag - one tiny collective and one
        # host read per OPTIMIZER step (the learner already synchronises once per task).
        self.collective_layout_check = collective_layout_check
        self.flat = None
        self.views = None      # per parameter: view into `flat`, or None (no gradient on any rank)
        self.nbytes = 0

    def _active(self):
        return dist.is_available() and dist.is_initialized() and dist.get_world_size(self.group) > 1

    def _build(self):
        dev = self.params[0].device
        mask = torch.tensor([0.0 if p.grad is None else 1.0 for p in self.params], device=dev)
        if self._active():
            dist.all_reduce(mask, op=dist.ReduceOp.MAX, group=self.group)
        present = [m > 0.5 for m in mask.tolist()]  # one host sync, first optimizer step only
        total = sum(-(-p.numel() // 64) * 64 for p, keep in zip(self.params, present) if keep)  # 256-byte aligned slots
        self.flat = torch.zeros(max(total, 1), device=dev, dtype=torch.float32)
        self.nbytes = 4 * total
        self.views, off = [], 0
        for p, keep in zip(self.params, present):
            if not keep:
                self.views.append(None)
                continue
            v = self.flat[off:off + p.numel()].view_as(p)
            off += -(-p.numel() // 64) * 64
            if p.grad is not None:
                v.copy_(p.grad)
            p.grad = v
            self.views.append(v)

    def _attached(self):
        """Make every present parameter's .grad the bucket view again (after an optimizer.zero_grad() dropped them):
        a detached view takes the freshly allocated gradient's value, or zero if this rank produced none."""
        for p, v in zip(self.params, self.views):
            if v is None:
                if p.grad is not None:
                    return False  # a parameter started to receive gradients: the layout is stale, rebuild
            elif p.grad is not v:
                if p.grad is not None:
                    v.copy_(p.grad)
                else:
                    v.zero_()
                p.grad = v
        return True

    def sync(self):
        """All-reduce(SUM) the bucket. Afterwards every rank holds identical gradients; parameters without a gradient on
        any rank still have grad None."""
        if not self.params:
            return
        stale = self.flat is None or not self._attached()
        if self._active() and self.collective_layout_check:
            flag = torch.tensor([1.0 if stale else 0.0], device=self.params[0].device)
            dist.all_reduce(flag, op=dist.ReduceOp.MAX, group=self.group)
            stale = bool(flag.item() > 0.5)  # every rank rebuilds, or none does
        if stale:
            self._build()
        if self._active():
            if self.p2p is not None:
                self.p2p(self.flat)
                # A peer that stalls poisons this bucket with NaN and raises the error word when the kernel gives up. The
                # word is checked AFTER the exchange has run (one event wait per optimizer step; the learner synchronises
                # per task anyway), so a timeout is an exception here - before optimizer.step() can write NaN into the
                # parameters and the optimizer state (round 3 only looked at the previous step's exchange at this point).
                done = torch.cuda.Event()
                done.record(torch.cuda.current_stream(self.flat.device))
                done.synchronize()
                self.p2p.raise_on_error()
            else:
                dist.all_reduce(self.flat, op=dist.ReduceOp.SUM, group=self.group)

    def zero_(self):
        """Start the next accumulation window (replaces optimizer.zero_grad())."""
        if self.flat is None:
            for p in self.params:
                p.grad = None
            return
        self.flat.zero_()
        for p, v in zip(self.params, self.views):
            p.grad = v  # None stays None


def allreduce_tensors(tensors, average=False):
    """All-reduce a list of tensors as ONE flat bucket (a ring all-reduce over xGMI is per-link bound: one large
    message beats many small ones). In place; no-op without an initialised process group."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1 or not tensors:
        return
    flat = torch.cat([t.reshape(-1) for t in tensors])
    dist.all_reduce(flat, op=dist.ReduceOp.SUM)
    if average:
        flat /= dist.get_world_size()
    off = 0
    for t in tensors:
        n = t.numel()
        t.copy_(flat[off:off + n].view_as(t))
        off += n


class RunningStatSync:
    """BatchNorm running statistics of a task-parallel training window, combined as SEQUENTIAL updates would have been.

    The reference trains on one device: every train-mode forward does r <- (1 - m) r + m s, so after the N forwards of a
    window r_end = a^N r_0 + (1 - a^N) s_avg (a = 1 - m, s_avg a recency-weighted mean of the batch statistics). Under task
    parallelism rank k only sees its n_k forwards: r_k = a^n_k r_0 + (1 - a^n_k) s_k. Plainly averaging the r_k (round 2)
    keeps a^(N / world) of the window's starting value instead of a^N - with 8 ranks the statistics would adapt 8 times
    slower than the reference's. Here every rank recovers its s_k, the ranks all-reduce sum_k n_k s_k and sum_k n_k in ONE
    flat message, and every rank sets r <- a^N r_0 + (1 - a^N) sum_k n_k s_k / N, num_batches_tracked += N - n_k: exact when
    the ranks' batch statistics agree, otherwise off only by the order in which the recency weights fall on the tasks
    (bounded by tests/test_gpu_dist.py). Call `begin()` after every optimizer step (window start), `sync()` before it."""

    def __init__(self, module, momentum=0.1, group=None):
        self.momentum, self.group = float(momentum), group
        self.stats, self.counters = [], []
        for name, buf in module.named_buffers():
            if name.endswith("running_mean") or name.endswith("running_var"):
                self.stats.append(buf)
            elif name.endswith("num_batches_tracked"):
                self.counters.append(buf)
        self.begin()

    def begin(self):
        self.start = [b.detach().clone() for b in self.stats]
        # the window's first forward count stays ON the device (all layers of a network advance together): begin() and sync()
        # run once per optimizer step inside a host-bound loop, and a .item() here was a device synchronisation each time
        self.start_count = self.counters[0].detach().clone() if self.counters else None

    def sync(self):
        if not self.stats or not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(self.group) == 1:
            return
        import math
        dev = self.stats[0].device
        # a^n in float64 on the device (a^n enters (cur - a^n r0) / (1 - a^n): an fp32 exp costs digits there); momentum >= 1
        # means a = 0: a^n = 0 for every n > 0 (and log(0) must not be taken - ADVICE r4)
        a = max(1.0 - self.momentum, 0.0)
        n = ((self.counters[0] - self.start_count).to(torch.float32) if self.counters
             else torch.zeros((), device=dev, dtype=torch.float32)).reshape(1)            # forwards this rank ran, on the device

        def a_pow(k):
            if a <= 0.0:
                return torch.where(k > 0, torch.zeros_like(k), torch.ones_like(k))
            return torch.exp(k.to(torch.float64) * math.log(a)).to(torch.float32)
        an = a_pow(n)
        live = n > 0
        denom = torch.where(live, 1.0 - an, torch.ones_like(an))
        cur = torch.cat([b.reshape(-1).to(torch.float32) for b in self.stats])
        r0 = torch.cat([b.reshape(-1).to(torch.float32) for b in self.start])
        s_k = torch.where(live, (cur - an * r0) / denom * n, torch.zeros_like(cur))         # n_k * s_k (0 for an idle rank)
        flat = torch.cat([s_k, n])
        dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.group)                       # ONE flat message, counters included
        N = flat[-1:]
        aN = a_pow(N)
        new = aN * r0 + (1.0 - aN) * flat[:-1] / N.clamp(min=1.0)
        new = torch.where(N > 0, new, cur)                                                  # nobody ran a forward: unchanged
        off = 0
        for b in self.stats:
            k = b.numel()
            b.copy_(new[off:off + k].view_as(b))
            off += k
        extra = torch.round(N - n).to(torch.int64)
        for c in self.counters:
            c += extra.reshape(c.shape) if c.dim() == 0 else extra
