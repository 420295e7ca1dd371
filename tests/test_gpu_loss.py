"""Native cross-entropy (csrc/loss.hip) against the reference's loss, which IS `F.cross_entropy` (utils/optim.py:8-9):
torch on the CPU, in fp64, is the oracle here. Forward at 1e-6, gradients at 1e-6 of the largest entry."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _case(N, C, seed, spread=4.0):
    g = torch.Generator().manual_seed(seed)
    z = spread * torch.randn(N, C, generator=g)
    lab = torch.randint(0, C, (N,), generator=g)
    return z, lab


@pytest.mark.parametrize("N,C", [(1, 1), (1, 5), (16, 5), (200, 10), (256, 5), (257, 64), (1000, 100), (33, 1000)])
@pytest.mark.parametrize("reduction", ["mean", "sum", "none"])
def test_cross_entropy_matches_torch(N, C, reduction):
    from orbit_dataset_amd.optim import cross_entropy
    z, lab = _case(N, C, N * 131 + C)
    zr = z.double().requires_grad_(True)
    ref = F.cross_entropy(zr, lab, reduction=reduction)
    w = torch.linspace(0.5, 1.5, N).double() if reduction == "none" else torch.tensor(0.37, dtype=torch.float64)
    (ref * w).sum().backward()
    zg = z.cuda().requires_grad_(True)
    out = cross_entropy(zg, lab.cuda(), reduction=reduction)
    assert out.shape == ref.shape and out.dtype == torch.float32
    (out * w.float().cuda()).sum().backward()
    assert torch.allclose(out.detach().cpu().double(), ref.detach(), rtol=2e-6, atol=2e-6)
    scale = zr.grad.abs().max().item()
    assert (zg.grad.cpu().double() - zr.grad).abs().max().item() <= 1e-6 * max(scale, 1e-30) + 1e-9


def test_cross_entropy_extreme_logits_and_determinism():
    from orbit_dataset_amd.optim import cross_entropy
    z = torch.tensor([[1e4, -1e4, 0.0], [-80.0, -90.0, -100.0], [3.0, 3.0, 3.0]])
    lab = torch.tensor([1, 0, 2])
    ref = F.cross_entropy(z.double(), lab, reduction="none")
    out = cross_entropy(z.cuda(), lab.cuda(), reduction="none").cpu().double()
    assert torch.isfinite(out).all() and torch.allclose(out, ref, rtol=1e-6, atol=1e-6)
    z, lab = _case(3000, 10, 5)
    a = cross_entropy(z.cuda(), lab.cuda())
    b = cross_entropy(z.cuda(), lab.cuda())
    assert a.item() == b.item()  # one summation order


def test_cross_entropy_edge_cases_are_loud():
    from orbit_dataset_amd.optim import cross_entropy
    z, lab = _case(8, 5, 3)
    bad = lab.clone()
    bad[3] = 7  # outside [0, C)
    assert torch.isnan(cross_entropy(z.cuda(), bad.cuda()))
    assert torch.isnan(cross_entropy(torch.empty(0, 5).cuda(), torch.empty(0, dtype=torch.long).cuda()))  # torch: nan
    assert cross_entropy(torch.empty(0, 5).cuda(), torch.empty(0, dtype=torch.long).cuda(), reduction="sum").item() == 0.0
    with pytest.raises(ValueError):
        cross_entropy(z.cuda(), lab.cuda(), reduction="median")
    with pytest.raises(ValueError):
        cross_entropy(z.cuda()[0], lab.cuda())
    with pytest.raises(RuntimeError):
        cross_entropy(z, lab)  # no CPU form


def test_learner_loss_scaling_chain():
    """The learner's LITE loss (single-step-learner.py:225-232): scaling * CE + 0-dim extra term, through autograd."""
    from orbit_dataset_amd.optim import cross_entropy
    z, lab = _case(200, 5, 11)
    scaling = 200 / (16 * 8)
    zr = z.double().requires_grad_(True)
    (scaling * F.cross_entropy(zr, lab) + 0.001 * zr.pow(2).sum()).backward()
    zg = z.cuda().requires_grad_(True)
    (scaling * cross_entropy(zg, lab.cuda()) + 0.001 * zg.pow(2).sum()).backward()
    assert (zg.grad.cpu().double() - zr.grad).abs().max().item() <= 2e-6 * zr.grad.abs().max().item()
