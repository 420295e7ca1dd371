"""ReLU-mask alignment for gradient parity tests ("tighten the gradient tests").

Fact being demonstrated: the native backward kernels are fp32-exact (2e-5 of each gradient's largest entry against a
float64 oracle). The only way a ReLU network's gradient can differ more is a pre-activation so close to zero (~1e-6)
that CPU and GPU rounding put it on different sides: the forward value is unaffected, but the backward MASK of that one
unit flips, which changes every upstream gradient by 1e-3..3e-2. Instead of tolerating that with loose bounds, the tests
align the masks: the oracle's ReLUs are instrumented (pre-activations recorded per call), and when the plain comparison
fails, the handful of units with |z| below `fragile` are flipped in the ORACLE - one at a time, then greedily a second
and third - until the oracle's gradients equal the GPU's to the fp32-exact bound. If no such flip set exists the test
fails: a kernel bug cannot hide behind the tolerance any more.
"""
import torch
import torch.nn as nn


class _FlipReLU(torch.autograd.Function):
    @staticmethod
    def forward(ctx, z, flip):
        mask = z > 0
        if flip is not None:
            mask = mask ^ flip
        ctx.save_for_backward(mask)
        return torch.where(mask, z, torch.zeros_like(z))  # a flipped unit passes its ~1e-6 pre-activation

    @staticmethod
    def backward(ctx, g):
        (mask,) = ctx.saved_tensors
        return g * mask.to(g.dtype), None


class ReluTap:
    """Instruments every nn.ReLU of `modules`: records the pre-activation of each call (in call order) and applies
    per-call boolean flip masks to the ReLU's on/off decision."""

    def __init__(self, *modules):
        self.calls, self.flips = [], {}
        self._patched = []
        for root in modules:
            if root is None:
                continue
            for m in root.modules():
                if isinstance(m, nn.ReLU) and "forward" not in m.__dict__:
                    m.forward = self._forward  # instance attribute shadows the class method
                    self._patched.append(m)

    def _forward(self, z):
        k = len(self.calls)
        self.calls.append(z.detach())
        return _FlipReLU.apply(z, self.flips.get(k))

    def begin(self):
        self.calls = []

    def remove(self):
        for m in self._patched:
            del m.__dict__["forward"]
        self._patched = []

    def fragile_units(self, fragile, limit=None):
        """Units with |pre-activation| < fragile, nearest to zero first (at most `limit`)."""
        out = []
        for k, z in enumerate(self.calls):
            idx = (z.abs() < fragile).nonzero()
            out += [(float(z[tuple(row)].abs()), k, tuple(int(v) for v in row)) for row in idx]
        out.sort()
        return [(k, idx) for _, k, idx in (out[:limit] if limit else out)]

    def set_flips(self, units):
        self.flips = {}
        for k, idx in units:
            if k not in self.flips:
                self.flips[k] = torch.zeros(self.calls[k].shape, dtype=torch.bool)
            self.flips[k][idx] = True


def aligned_error(run_oracle, error_of, tap, exact, fragile=2e-5, max_flips=3, max_candidates=32):
    """run_oracle() -> oracle gradients with the tap's current flips; error_of(grads) -> worst relative error against the
    GPU gradients. Returns (error, flipped units, plain error)."""
    tap.flips = {}
    tap.begin()
    plain = error_of(run_oracle())
    if plain < exact:
        return plain, [], plain
    candidates = tap.fragile_units(fragile, limit=max_candidates)  # the ones nearest to zero are the ones that flip
    assert candidates, "gradient error %g with no pre-activation within %g of zero: not a mask flip" % (plain, fragile)
    chosen, best = [], plain
    for _ in range(max_flips):
        trial_best, trial_unit = best, None
        for u in candidates:
            if u in chosen:
                continue
            tap.set_flips(chosen + [u])
            tap.begin()
            e = error_of(run_oracle())
            if e < trial_best:
                trial_best, trial_unit = e, u
        if trial_unit is None:
            break
        chosen.append(trial_unit)
        best = trial_best
        if best < exact:
            break
    tap.flips = {}
    return best, chosen, plain
