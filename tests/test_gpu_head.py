"""GPU parity of the prototype head (C-ABI) against the oracle restatement of classifier_heads.py, and the
size-independent properties that hold at BASELINE.json's full sizes."""
import pytest
import torch

pytestmark = pytest.mark.gpu

import orbit_dataset_amd  # noqa: E402,F401
from oracle import blocks  # noqa: E402
from orbit_dataset_amd.model.classifier_heads import PrototypicalClassifier  # noqa: E402


def _task(N, M, D, way, seed, label_values=None, T=1):
    g = torch.Generator().manual_seed(seed)
    centers = torch.randn(way, D, generator=g) * 0.3
    cls = torch.arange(way).repeat_interleave(N // way)[torch.randperm(N // way * way, generator=g)]
    feats = torch.relu(centers[cls].repeat_interleave(T, 0) + torch.randn(len(cls) * T, D, generator=g) * 0.5)
    qc = torch.randint(0, way, (M,), generator=g)
    q = torch.relu(centers[qc].repeat_interleave(T, 0) + torch.randn(M * T, D, generator=g) * 0.5)
    vals = torch.arange(way) if label_values is None else torch.tensor(label_values)
    return feats, vals[cls], q, vals[qc]


@pytest.mark.parametrize("D", [512, 1280, 100])
@pytest.mark.parametrize("way,label_values", [(5, None), (10, None), (3, (3, 7, 9))])
@pytest.mark.parametrize("dist,scale", [("euclidean", 1.0), ("cosine", 32.0)])
def test_head_matches_oracle(device, D, way, label_values, dist, scale):
    feats, labels, q, qlab = _task(200 // way * way, 200, D, way, seed=D + way, label_values=label_values)
    head = PrototypicalClassifier(scale, dist)
    head.configure(feats.to(device), labels.to(device))
    logits, amax = head.predict(q.to(device), return_argmax=True)
    ids, W, b = blocks.proto_configure(feats, labels, dist)
    want = blocks.proto_predict(q, W, b, scale, dist)
    assert head.class_ids.cpu().tolist() == ids
    assert (head.weight.cpu() - W).abs().max().item() < 1e-5
    if dist == "euclidean":
        assert (head.bias.cpu() - b).abs().max().item() < 1e-4 * max(1.0, b.abs().max().item())
    err = (logits.cpu() - want).abs().max().item()
    assert err < 1e-3, err
    assert torch.equal(logits.cpu().argmax(1), want.argmax(1))          # identical frame accuracy
    assert torch.equal(amax.cpu().long(), logits.cpu().argmax(1))       # fused argmax == argmax of its logits


@pytest.mark.parametrize("T", [1, 8])
def test_head_fused_pooling(device, T):
    """T frames per clip averaged inside the kernels == MeanPooler followed by the head (poolers.py:13-16)."""
    feats, labels, q, _ = _task(25, 40, 512, 5, seed=T, T=T)
    head = PrototypicalClassifier(1.0, "euclidean")
    head.configure(feats.to(device), labels.to(device), frames_per_clip=T)
    logits = head.predict(q.to(device), frames_per_clip=T).cpu()
    ids, W, b = blocks.proto_configure(blocks.mean_pool(feats, T), labels)
    want = blocks.proto_predict(blocks.mean_pool(q, T), W, b)
    assert (logits - want).abs().max().item() < 1e-3


def test_head_edge_cases(device):
    # zero query row under cosine -> exactly 0 (SURVEY §8c G1 edge), 1-shot (N == C), predict before configure
    head = PrototypicalClassifier(1.0, "cosine")
    with pytest.raises(AttributeError):
        head.predict(torch.zeros(2, 8, device=device))
    feats = torch.eye(5, 64)[:, :64] + 0.1
    labels = torch.arange(5)
    head.configure(feats.to(device), labels.to(device))
    q = torch.zeros(3, 64)
    q[1] = feats[2]
    logits = head.predict(q.to(device)).cpu()
    assert torch.all(logits[0] == 0) and torch.all(logits[2] == 0)
    assert logits[1].argmax().item() == 2 and abs(logits[1, 2].item() - 1.0) < 1e-6
    head.reset()
    with pytest.raises(AttributeError):
        head.predict(q.to(device))
    with pytest.raises(AssertionError):
        PrototypicalClassifier().configure(feats.to(device), labels[:4].to(device))


def test_head_full_size_properties(device):
    """At full size (M = 64 tasks x 200 queries, D = 1280) check properties instead of an oracle run:
    (i) a query equal to prototype c scores highest on column c with logit mu.mu (euclidean: 2 mu.mu - mu.mu);
    (ii) logits are linear in logit_scale; (iii) permuting the support set leaves W, b unchanged up to fp32
    summation order."""
    g = torch.Generator().manual_seed(0)
    D, way, N = 1280, 10, 200
    feats = torch.rand(N, D, generator=g).to(device)
    labels = torch.arange(way).repeat_interleave(N // way).to(device)
    h1, h2 = PrototypicalClassifier(1.0), PrototypicalClassifier(4.0)
    h1.configure(feats, labels)
    h2.configure(feats, labels)
    mu = h1.weight / 2
    logits = h1.predict(mu)
    assert torch.equal(logits.argmax(1).cpu(), torch.arange(way))
    assert (logits.diagonal() + h1.bias).abs().max().item() < 1e-2      # mu.mu == -b
    q = torch.rand(64 * 200, D, generator=g).to(device)
    a, b = h1.predict(q), h2.predict(q)
    assert (4.0 * a - b).abs().max().item() < 1e-3
    perm = torch.randperm(N, generator=g).to(device)
    h3 = PrototypicalClassifier(1.0)
    h3.configure(feats[perm], labels[perm])
    assert (h3.weight - h1.weight).abs().max().item() < 1e-5


# ---- Versa / Mahalanobis heads -----------------------------------------------------------------
import os  # noqa: E402

import numpy as np  # noqa: E402

_GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _gold(name):
    return {k: (torch.from_numpy(v) if v.dtype.kind in "fiu" and v.ndim > 0 else v)
            for k, v in np.load(os.path.join(_GOLD, name + ".npz"), allow_pickle=False).items()}


@pytest.mark.parametrize("tag,D", [("w5", 64), ("single", 96)])
def test_G10_versa_head(device, tag, D):
    from orbit_dataset_amd import synthetic
    from orbit_dataset_amd.model.classifier_heads import VersaClassifier
    g = _gold("G10_heads_versa_mahalanobis")
    head = VersaClassifier(D, logit_scale=2.0)
    synthetic.init_parameters_(head, prefix="classifier.")
    head = head.to(device)
    head.configure(g[tag + "_features"].to(device), g[tag + "_labels"].to(device))
    assert (head.weight.cpu() - g[tag + "_versa_weight"]).abs().max().item() < 2e-5
    assert (head.bias.cpu() - g[tag + "_versa_bias"]).abs().max().item() < 2e-5
    logits = head.predict(g[tag + "_query"].to(device)).cpu()
    want = g[tag + "_versa_logits"]
    assert (logits - want).abs().max().item() < 1e-3
    assert torch.equal(logits.argmax(1), want.argmax(1))


@pytest.mark.parametrize("tag", ["w5", "single"])
def test_G10_mahalanobis_head(device, tag):
    from orbit_dataset_amd.model.classifier_heads import MahalanobisClassifier
    g = _gold("G10_heads_versa_mahalanobis")
    head = MahalanobisClassifier(1.0)
    head.configure(g[tag + "_features"].to(device), g[tag + "_labels"].to(device))
    assert (head.means.cpu() - g[tag + "_maha_means"]).abs().max().item() < 1e-5
    P = g[tag + "_maha_precisions"]
    assert (head.precisions.cpu() - P).abs().max().item() < 1e-4 * P.abs().max().item()
    TP = g[tag + "_maha_task_precision"]
    assert (head.task_precision.cpu() - TP).abs().max().item() < 1e-4 * TP.abs().max().item()
    logits = head.predict(g[tag + "_query"].to(device)).cpu()
    want = g[tag + "_maha_logits"]
    assert (logits - want).abs().max().item() < 1e-3 * want.abs().max().item()
    assert torch.equal(logits.argmax(1), want.argmax(1))


@pytest.mark.parametrize("D,way,shots,M", [(512, 5, 40, 200), (1280, 5, 40, 200), (512, 10, 7, 33)])
def test_versa_mahalanobis_at_extractor_width(device, D, way, shots, M):
    """Both heads at the extractors' feature widths (resnet18 512, efficientnet_b0 1280) against the oracle in fp64."""
    from oracle import blocks
    from orbit_dataset_amd import synthetic
    from orbit_dataset_amd.model.classifier_heads import MahalanobisClassifier, VersaClassifier
    g = torch.Generator().manual_seed(D + way)
    lab = torch.arange(way).repeat_interleave(shots)[torch.randperm(way * shots, generator=g)]
    centres = torch.randn(way, D, generator=g)
    feats = centres[lab] + 0.7 * torch.randn(way * shots, D, generator=g)
    q = centres[torch.randint(0, way, (M,), generator=g)] + 0.7 * torch.randn(M, D, generator=g)
    # Versa
    head = VersaClassifier(D, 1.0)
    synthetic.init_parameters_(head, prefix="classifier.")
    ref_w, ref_b = blocks.DenseResidualBlock(D, D), blocks.DenseResidualBlock(D, 1)
    ref_w.load_state_dict(head.weight_processor.state_dict()), ref_b.load_state_dict(head.bias_processor.state_dict())
    head = head.to(device)
    head.configure(feats.to(device), lab.to(device))
    with torch.no_grad():
        ids, W, b = blocks.versa_configure(feats.double(), lab, ref_w.double(), ref_b.double())
        want = q.double() @ W.t() + b
    assert (head.weight.cpu().double() - W).abs().max().item() < 1e-4 * W.abs().max().item()
    got = head.predict(q.to(device)).cpu().double()
    assert (got - want).abs().max().item() < 1e-4 * want.abs().max().item()
    assert torch.equal(got.argmax(1), want.argmax(1))
    # Mahalanobis
    maha = MahalanobisClassifier(1.0)
    maha.configure(feats.to(device), lab.to(device))
    ids, means, precisions, task_mean, task_precision = blocks.mahalanobis_configure(feats.double(), lab)
    assert (maha.precisions.cpu().double() - precisions).abs().max().item() < 1e-4 * precisions.abs().max().item()
    assert (maha.task_mean.cpu().double() - task_mean).abs().max().item() < 1e-5
    want = blocks.mahalanobis_predict(q.double(), means, precisions)
    got = maha.predict(q.to(device)).cpu().double()
    assert (got - want).abs().max().item() < 1e-3 * want.abs().max().item()
    assert torch.equal(got.argmax(1), want.argmax(1))


def test_spd_inverse(device):
    from orbit_dataset_amd import _lib
    lib = _lib.load()
    for n, batch in ((64, 3), (100, 2), (512, 2)):
        g = torch.Generator().manual_seed(n)
        a = torch.randn(batch, n, 2 * n, generator=g, dtype=torch.float64)
        A = a @ a.transpose(1, 2) / (2 * n) + torch.eye(n, dtype=torch.float64)
        t = A.float().to(device).contiguous()
        out = torch.empty_like(t)
        _lib.check(lib.orbit_spd_inverse(_lib.dptr(t), _lib.dptr(out), n, batch, _lib.stream_handle()), "orbit_spd_inverse")
        torch.cuda.synchronize()
        # the fp64 reference inverse on ONE thread (with the suite's 16 intra-op threads the host LAPACK of this image returned
        # a wrong batched inverse: diagonal entries below 1 / lambda_max) and, independently of any host inverse, the residual
        threads = torch.get_num_threads()
        torch.set_num_threads(1)
        try:
            want = torch.linalg.inv(A)
        finally:
            torch.set_num_threads(threads)
        assert (want @ A - torch.eye(n, dtype=torch.float64)).abs().max().item() < 1e-9
        assert (out.cpu().double() - want).abs().max().item() < 1e-5 * want.abs().max().item()
        assert (out.cpu().double() @ A - torch.eye(n, dtype=torch.float64)).abs().max().item() < 2e-5


def test_mahalanobis_predict_backward(device):
    from oracle import blocks
    from orbit_dataset_amd.model.classifier_heads import MahalanobisClassifier
    g = torch.Generator().manual_seed(5)
    D, way, shots, M = 512, 5, 12, 37
    lab = torch.arange(way).repeat_interleave(shots)
    feats = torch.randn(way, D, generator=g)[lab] + 0.7 * torch.randn(way * shots, D, generator=g)
    q = torch.randn(M, D, generator=g)
    dl = torch.randn(M, way, generator=g)
    head = MahalanobisClassifier(2.0)
    head.configure(feats.to(device), lab.to(device))
    qd = q.to(device).requires_grad_(True)
    head.predict(qd).backward(dl.to(device))
    qr = q.double().requires_grad_(True)
    blocks.mahalanobis_predict(qr, head.means.cpu().double(), head.precisions.cpu().double(), 2.0).backward(dl.double())
    assert (qd.grad.cpu().double() - qr.grad).abs().max().item() < 2e-5 * qr.grad.abs().max().item()


def test_many_pending_label_sets_keep_their_own_count(device):
    """label sets resolved on the side stream carried their class count through a shared ring of 64 pinned slots;
    with more than 64 of them pending, an earlier one read a later one's count. Each PendingLabelSet now owns a slot from a
    free list: 150 pending sets with class counts 1..9 all resolve to their own ids, dropped (never resolved) sets give their
    slot back, and a head configured on a pending set still exposes the exact weight / class_ids."""
    from orbit_dataset_amd.model.classifier_heads import PendingLabelSet
    dev = torch.device(device)
    g = torch.Generator().manual_seed(3)
    labels = []
    for i in range(150):
        c = 1 + i % 9
        vals = torch.randperm(40, generator=g)[:c].sort().values
        labels.append((vals, vals[torch.randint(0, c, (30,), generator=g)].scatter_(0, torch.arange(c), vals)))
    pend = [PendingLabelSet(lab.to(dev), dev) for _, lab in labels]
    slots = {(id(p.slot[0]), p.slot[1]) for p in pend}
    assert len(slots) == 150  # distinct slots while all are pending
    for (vals, _), p in zip(reversed(labels), reversed(pend)):  # resolve in the opposite order they were issued
        assert p.resolve().cpu().tolist() == vals.tolist()
        assert p.slot is None  # given back
    free_before = len(PendingLabelSet._free[str(dev)])
    dropped = [PendingLabelSet(lab.to(dev), dev) for _, lab in labels[:10]]
    assert len(PendingLabelSet._free[str(dev)]) == free_before - 10
    del dropped
    import gc
    gc.collect()
    assert len(PendingLabelSet._free[str(dev)]) == free_before
    # a head configured on a pending set, with 70 other sets issued before it is read
    feats, lab, q, _ = _task(60, 20, 64, 6, seed=11, label_values=(1, 4, 5, 8, 13, 21))
    head = PrototypicalClassifier(1.0, "euclidean")
    lab_dev = lab.to(dev)
    head.configure(feats.to(dev), lab_dev, class_ids=PrototypicalClassifier.label_set(lab_dev, dev))
    later = [PendingLabelSet(l.to(dev), dev) for _, l in labels[:70]]
    assert head.class_ids.cpu().tolist() == [1, 4, 5, 8, 13, 21] and head.weight.shape == (6, 64)
    ids, W, b = blocks.proto_configure(feats, lab, "euclidean")
    assert (head.weight.cpu() - W).abs().max().item() < 1e-5
    del later
