"""Multi-rank logic on CPU (gloo, world_size 2): sharding helpers, and the support-sharded prototype exchange —
per-rank partial sums + counts, ONE all-reduce(SUM) of the [C*D + C] payload, identical prototypes on every rank,
equal to the single-process result. The per-rank partials are produced by a numpy restatement of
orbit_proto_configure's contract (the HIP kernel itself needs a GPU; its N>1 use is covered on-device)."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import orbit_dataset_amd  # noqa: E402,F401
from oracle import blocks  # noqa: E402
from orbit_dataset_amd import dist as odist  # noqa: E402


def test_sharding_helpers():
    for n in (0, 1, 5, 200, 201, 207):
        for world in (1, 2, 3, 8):
            spans = [odist.shard_bounds(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1
    assert list(odist.tasks_for_rank(10, 1, 4)) == [1, 5, 9]
    assert sorted(sum((list(odist.tasks_for_rank(64, r, 8)) for r in range(8)), [])) == list(range(64))


def partial_payload(feats, labels, class_ids, T=1):
    """numpy restatement of orbit_proto_configure: per-class sums of per-clip means + counts, ascending clip order."""
    C, D = len(class_ids), feats.shape[1]
    pooled = feats.reshape(-1, T, D).mean(1)
    out = np.zeros(C * D + C, dtype=np.float32)
    for c, cid in enumerate(class_ids):
        rows = pooled[labels == cid]
        out[c * D:(c + 1) * D] = rows.sum(0, dtype=np.float32) if len(rows) else 0
        out[C * D + c] = len(rows)
    return out


def _worker(rank, world, port, feats, labels, q, result):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    r, w, _ = odist.init_from_env("gloo")
    sh = odist.SupportSharding(r, w)
    lo, hi = sh.bounds(len(labels))
    class_ids = np.unique(labels.numpy())
    payload = torch.from_numpy(partial_payload(feats[lo:hi].numpy(), labels[lo:hi].numpy(), class_ids))
    sh.reduce_(payload)  # the one exchange step
    C, D = len(class_ids), feats.shape[1]
    sums, counts = payload[:C * D].reshape(C, D), payload[C * D:]
    W = 2 * sums / counts[:, None]
    b = -((sums / counts[:, None]) ** 2).sum(1)
    logits = q @ W.t() + b
    # query-sharded gather through the same helper the GPU path uses
    ql, qh = sh.bounds(len(q))
    full = torch.zeros_like(logits)
    full[ql:qh] = logits[ql:qh]
    sh.reduce_(full)
    result[rank] = (W.numpy(), b.numpy(), full.numpy())
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("N,way", [(200, 5), (7, 3)])
def test_support_sharded_prototypes_world2(N, way):
    g = torch.Generator().manual_seed(N)
    labels = torch.tensor([3, 7, 9, 11, 20][:way])[torch.randint(0, way, (N,), generator=g)]
    labels[:way] = torch.tensor([3, 7, 9, 11, 20][:way])  # every class present at least once (rank 0 only, for N=7)
    feats = torch.rand(N, 96, generator=g)
    q = torch.rand(16, 96, generator=g)
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mgr = mp.Manager()
    result = mgr.dict()
    mp.spawn(_worker, args=(2, port, feats, labels, q, result), nprocs=2, join=True)
    ids, W, b = blocks.proto_configure(feats, labels)
    want = blocks.proto_predict(q, W, b)
    for r in range(2):
        Wr, br, lr = result[r]
        assert np.allclose(Wr, W.numpy(), atol=1e-5) and np.allclose(br, b.numpy(), atol=1e-4)
        assert np.allclose(lr, want.numpy(), atol=1e-3)
    assert np.array_equal(result[0][0], result[1][0])  # bit-identical prototypes on both ranks


# ---- X3: gradient all-reduce of the task-parallel LITE training step ------------------------------------------------
def _lite_grads(task_ids, tasks_per_batch):
    """Accumulated gradients of the oracle's LITE step over `task_ids` (oracle/training.py, pinned by G6/G8)."""
    from oracle.recogniser import OracleRecogniser
    from oracle.training import LiteTrainer
    from orbit_dataset_amd import synthetic
    ref = OracleRecogniser("resnet18", False, "proto", 1, 4, num_lite_samples=2)
    synthetic.init_parameters_(ref.fe)
    tr = LiteTrainer(ref, True, tasks_per_batch)
    for t in task_ids:
        task = synthetic.make_task(300 + t, way=3, shots=1, frames_per_shot=2, num_query=4, frame_size=32)
        tr.train_task_with_lite(task["context_clips"], task["context_labels"], task["target_clips"],
                                task["target_labels"], seeds=(900 + t,))
    return [p.grad for _, p in sorted(ref.fe.named_parameters()) if p.grad is not None]


def _train_worker(rank, world, port, result):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    torch.set_num_threads(2)
    r, w, _ = odist.init_from_env("gloo")
    grads = _lite_grads(list(odist.tasks_for_rank(2, r, w)), tasks_per_batch=2)
    odist.allreduce_tensors(grads, average=False)  # the one exchange step per optimizer step
    result[rank] = [g.numpy() for g in grads[:4]] + [grads[-1].numpy()]
    dist.barrier()
    dist.destroy_process_group()


def test_gradient_allreduce_matches_single_process_accumulation():
    """Two ranks, one task each, SUM all-reduce of one flat bucket == one process accumulating both tasks (the loss
    already carries 1/tasks_per_batch, single-step-learner.py:231)."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mgr = mp.Manager()
    result = mgr.dict()
    mp.spawn(_train_worker, args=(2, port, result), nprocs=2, join=True)
    torch.set_num_threads(2)
    want = _lite_grads([0, 1], tasks_per_batch=2)
    want = [g.numpy() for g in want[:4]] + [want[-1].numpy()]
    for r in range(2):
        for got, ref in zip(result[r], want):
            assert np.abs(got - ref).max() <= 2e-5 * max(np.abs(ref).max(), 1e-12)
    for a, b in zip(result[0], result[1]):
        assert np.array_equal(a, b)  # identical gradients on both ranks -> identical optimizer steps


# ---- persistent flat gradient bucket (dist.GradientBucket): presence mask, None grads stay None ------------------------
def _toy():
    torch.manual_seed(5)
    m = torch.nn.Module()
    m.a = torch.nn.Parameter(torch.randn(7, 3))    # gradient on every rank
    m.b = torch.nn.Parameter(torch.randn(5))       # gradient on rank 0's tasks only
    m.c = torch.nn.Parameter(torch.randn(4, 2))    # never receives a gradient (like a FiLM-replaced BatchNorm weight)
    return m


def _toy_loss(m, task, tasks_per_batch):
    x = torch.full((3,), float(task + 1))
    loss = (m.a @ x).pow(2).sum()
    if task % 2 == 0:
        loss = loss + (m.b * (task + 1)).sum()
    return loss / tasks_per_batch


def _toy_train(rank, world, steps=3, tasks_per_batch=4):
    m = _toy()
    opt = torch.optim.Adam(m.parameters(), lr=0.05, weight_decay=0.2)
    bucket = odist.GradientBucket(m.parameters()) if world > 1 else None
    for step in range(steps):
        for t in range(tasks_per_batch):
            if t % world == rank:
                _toy_loss(m, step * tasks_per_batch + t, tasks_per_batch).backward()
        if bucket is not None:
            bucket.sync()
            assert m.c.grad is None  # no gradient anywhere -> stays None -> Adam's weight decay must not touch it
            assert m.a.grad.data_ptr() == bucket.flat.data_ptr()  # gradients live inside the flat bucket
        opt.step()
        if bucket is not None:
            if step == 1:
                opt.zero_grad()  # a caller that drops the views: the bucket must re-attach them
            else:
                bucket.zero_()
        else:
            opt.zero_grad()
    return [p.detach().clone() for p in (m.a, m.b, m.c)]


def _bucket_worker(rank, world, port, result):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    r, w, _ = odist.init_from_env("gloo")
    result[rank] = [t.numpy() for t in _toy_train(r, w)]
    dist.barrier()
    dist.destroy_process_group()


def test_gradient_bucket_matches_single_process_and_skips_gradless_parameters():
    """zero-filling None gradients before the all-reduce let Adam's L2 weight decay move parameters
    that never receive a gradient (Versa hyper-networks, FiLM-replaced BatchNorm). With the presence mask they keep
    grad None on every rank; the task-parallel run equals single-process accumulation over the same tasks."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mgr = mp.Manager()
    result = mgr.dict()
    mp.spawn(_bucket_worker, args=(2, port, result), nprocs=2, join=True)
    want = _toy_train(0, 1)
    init = _toy()
    for r in range(2):
        a, b, c = result[r]
        assert np.allclose(a, want[0].numpy(), atol=1e-6) and np.allclose(b, want[1].numpy(), atol=1e-6)
        assert np.array_equal(c, init.c.detach().numpy()) and np.array_equal(c, want[2].numpy())
    assert all(np.array_equal(x, y) for x, y in zip(result[0], result[1]))  # bit-identical replicas


# ---- the layout decision of GradientBucket is collective -------------------------------------------------
def _late_param_worker(rank, world, port, result):
    """Window 1: only `a` has a gradient (on both ranks). Window 2: `c` starts to receive a gradient on rank 0 while rank 1
    had NO task in that window (total_steps % world != 0): rank 0 alone sees a stale layout. Both ranks must rebuild
    together - a rank rebuilding alone pairs its mask all-reduce with the peer's bucket all-reduce (hang / corruption)."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    r, w, _ = odist.init_from_env("gloo")
    m = _toy()
    bucket = odist.GradientBucket(m.parameters())
    (m.a.sum() * (r + 1)).backward()
    bucket.sync()
    assert m.c.grad is None and torch.allclose(m.a.grad, torch.full_like(m.a, 3.0))
    bucket.zero_()
    if r == 0:
        (m.a.sum() + 2.0 * m.c.sum()).backward()
    bucket.sync()
    assert m.c.grad is not None and torch.allclose(m.c.grad, torch.full_like(m.c, 2.0))  # on BOTH ranks
    assert torch.allclose(m.a.grad, torch.ones_like(m.a))
    assert m.c.grad.untyped_storage().data_ptr() == bucket.flat.untyped_storage().data_ptr()
    result[rank] = True
    dist.barrier()
    dist.destroy_process_group()


def test_gradient_bucket_rebuilds_collectively_when_one_rank_sees_a_new_gradient():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mgr = mp.Manager()
    result = mgr.dict()
    mp.spawn(_late_param_worker, args=(2, port, result), nprocs=2, join=True)
    assert result[0] and result[1]


# ---- BatchNorm running statistics of a task-parallel window = the reference's sequential updates ---------------------
def _stat_worker(rank, world, port, result):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    odist.init_from_env("gloo")
    bn = torch.nn.BatchNorm2d(6)
    g = torch.Generator().manual_seed(1)
    bn.running_mean.copy_(torch.randn(6, generator=g)), bn.running_var.copy_(torch.rand(6, generator=g) + 0.5)
    sync = odist.RunningStatSync(bn, momentum=0.1)
    # a window of 7 train-mode forwards dealt round-robin (rank 0: 4, rank 1: 3); batch statistics differ per task
    batches = [torch.randn(5, 6, 4, 4, generator=g) * (1.0 + 0.05 * t) + 0.1 * t for t in range(7)]
    bn.train()
    for t, x in enumerate(batches):
        if t % world == rank:
            bn(x)
    sync.sync()
    result[rank] = (bn.running_mean.clone().numpy(), bn.running_var.clone().numpy(), int(bn.num_batches_tracked))
    dist.barrier()
    dist.destroy_process_group()


def test_running_statistics_combine_like_sequential_updates():
    """dist.RunningStatSync (rank-AVERAGED statistics were an unbounded deviation): after a 7-forward window on
    2 ranks the statistics carry a^7 of the window's starting value, exactly as 7 sequential updates do, and differ from
    the single-process result only by the recency weighting of the tasks (bounded here); plain averaging is far off."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mgr = mp.Manager()
    result = mgr.dict()
    mp.spawn(_stat_worker, args=(2, port, result), nprocs=2, join=True)
    bn = torch.nn.BatchNorm2d(6)
    g = torch.Generator().manual_seed(1)
    bn.running_mean.copy_(torch.randn(6, generator=g)), bn.running_var.copy_(torch.rand(6, generator=g) + 0.5)
    start = bn.running_mean.clone()
    batches = [torch.randn(5, 6, 4, 4, generator=g) * (1.0 + 0.05 * t) + 0.1 * t for t in range(7)]
    bn.train()
    for x in batches:
        bn(x)
    want_mean, want_var = bn.running_mean.numpy(), bn.running_var.numpy()
    assert np.array_equal(result[0][0], result[1][0]) and np.array_equal(result[0][1], result[1][1])  # identical replicas
    assert result[0][2] == result[1][2] == 7
    gap = np.abs(want_mean - start.numpy()).max()  # how far the window moved the statistics
    assert np.abs(result[0][0] - want_mean).max() < 0.08 * gap
    assert np.abs(result[0][1] - want_var).max() < 0.08 * np.abs(want_var).max()
    # identical batch statistics on every forward: the combination is exact
    a = 0.9
    r0, s = 2.0, 5.0
    seq = a ** 7 * r0 + (1 - a ** 7) * s
    parts = [(a ** n * r0 + (1 - a ** n) * s, n) for n in (4, 3)]
    comb = a ** 7 * r0 + (1 - a ** 7) * sum(n * (r - a ** n * r0) / (1 - a ** n) for r, n in parts) / 7
    assert abs(comb - seq) < 1e-12 and abs(sum(r for r, _ in parts) / 2 - seq) > 0.3
