"""CPU self-test of tests/relu_align.py (the ReLU-mask alignment the GPU gradient tests rely on): a planted fragile unit
that the "other implementation" resolved the other way is found, flipping it reproduces that implementation's gradients
exactly, and a genuine gradient bug is NOT explained away."""
import os
import sys

import pytest
import torch
import torch.nn as nn

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from relu_align import ReluTap, aligned_error  # noqa: E402


def _setup():
    torch.manual_seed(0)
    net = nn.Sequential(nn.Linear(8, 16), nn.ReLU(), nn.Linear(16, 16), nn.ReLU(), nn.Linear(16, 4)).double()
    x = torch.randn(5, 8, dtype=torch.float64)
    with torch.no_grad():  # plant a fragile unit: pre-activation 1e-7
        z = net[0](x)
        net[0].bias[3] -= z[2, 3] - 1e-7
    tap = ReluTap(net)

    def run():
        net.zero_grad()
        net(x).sum().backward()
        return [p.grad.clone() for p in net.parameters()]
    return net, tap, run


def _err(a, b):
    return max(float((u - v).abs().max() / v.abs().max().clamp_min(1e-30)) for u, v in zip(a, b))


def test_flipped_unit_is_found_and_explains_the_difference():
    _, tap, run = _setup()
    tap.begin()
    run()  # records the pre-activation shapes
    tap.set_flips([(0, (2, 3))])
    tap.begin()
    other = run()  # the "other implementation": same network, unit (call 0, [2, 3]) on the other side of zero
    err, flipped, plain = aligned_error(run, lambda g: _err(other, g), tap, exact=1e-9)
    assert plain > 1e-4 and err < 1e-9 and flipped == [(0, (2, 3))]


def test_a_real_bug_is_not_explained_away():
    _, tap, run = _setup()
    tap.begin()
    wrong = [g * 1.01 for g in run()]  # 1 % off everywhere: no mask flip produces that
    err, flipped, plain = aligned_error(run, lambda g: _err(wrong, g), tap, exact=1e-6)
    assert err > 1e-3
    tap.remove()
    with pytest.raises(AssertionError):  # and with no fragile unit at all the helper refuses outright
        net = nn.Sequential(nn.Linear(4, 4), nn.ReLU(), nn.Linear(4, 2)).double()
        x = torch.randn(3, 4, dtype=torch.float64)
        t2 = ReluTap(net)

        def run2():
            net.zero_grad()
            net(x).sum().backward()
            return [p.grad.clone() for p in net.parameters()]
        t2.begin()
        bad = [g + 1.0 for g in run2()]
        aligned_error(run2, lambda g: _err(bad, g), t2, exact=1e-9, fragile=1e-12)
