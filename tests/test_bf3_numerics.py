"""The arithmetic of the opt-in `conv_bf3` path (orbit-dataset_amd/csrc/bf3.h, conv_bf3.hip), restated in numpy on the CPU:
the three-way bf16 split of an fp32 value is exact (x = x0 + x1 + x2, both residual subtractions exact in fp32), every bf16 x bf16
product is exact in fp32, and the six products the kernels sum differ from the exact product by at most 2^-23 of it (the three
dropped terms: |x1 w2|, |x2 w1| <= 2^-24 |x w| each; measured max 2^-24.4, median 2^-29). No GPU: this pins the algorithm, the -m gpu tests pin the kernels."""
import numpy as np


def bf16_rne(x):
    """float32 -> nearest bfloat16 (ties to even), returned as float32 (what v_cvt_pk_bf16_f32 does for finite values)."""
    u = np.asarray(x, dtype=np.float32).view(np.uint32).astype(np.uint64)
    r = (u + 0x7FFF + ((u >> 16) & 1)) & 0xFFFF0000
    return r.astype(np.uint32).view(np.float32)


def split3(x):
    x = np.asarray(x, dtype=np.float32)
    x0 = bf16_rne(x)
    r1 = (x - x0).astype(np.float32)
    x1 = bf16_rne(r1)
    r2 = (r1 - x1).astype(np.float32)
    x2 = bf16_rne(r2)
    return x0, x1, x2, r1, r2


def _samples(n=200000, seed=0):
    g = np.random.default_rng(seed)
    mant = g.standard_normal(n).astype(np.float32)
    expo = g.integers(-20, 20, n)
    x = (mant * np.exp2(expo)).astype(np.float32)
    edge = np.array([0.0, 1.0, -1.0, 1.0 + 2.0 ** -8, 1.0 + 2.0 ** -9, 1.0 - 2.0 ** -9, 255.5, 3.0e38, -3.0e38, 1e-30, 2.0 ** -100,
                     np.float32(1.0) + np.float32(2.0 ** -23), np.float32(0.1), np.float32(1 / 3)], dtype=np.float32)
    return np.concatenate([x, edge])


def test_three_way_split_is_exact():
    x = _samples()
    x0, x1, x2, r1, r2 = split3(x)
    # the residuals are exactly representable: computing them in float64 gives the same numbers
    assert np.array_equal(r1.astype(np.float64), x.astype(np.float64) - x0.astype(np.float64))
    assert np.array_equal(r2.astype(np.float64), r1.astype(np.float64) - x1.astype(np.float64))
    # 24 significand bits in three pieces of 8: the third piece takes what is left, exactly
    assert np.array_equal(x2.astype(np.float64), r2.astype(np.float64))
    assert np.array_equal(x0.astype(np.float64) + x1.astype(np.float64) + x2.astype(np.float64), x.astype(np.float64))
    # each piece is a bfloat16 (low 16 bits clear) and the pieces shrink by 2^-8 per level (round to nearest: <= half an ulp)
    for p in (x0, x1, x2):
        assert not np.any(p.view(np.uint32) & 0xFFFF)
    nz = x != 0
    assert np.all(np.abs(r1[nz]) <= np.abs(x[nz]) * 2.0 ** -8) and np.all(np.abs(r2[nz]) <= np.abs(x[nz]) * 2.0 ** -16)


def test_six_products_against_the_fp32_product():
    g = np.random.default_rng(1)
    x, w = _samples(seed=2)[:100000], _samples(seed=3)[:100000]
    w = w[g.permutation(len(w))]
    x = np.clip(x, -1e15, 1e15).astype(np.float32)
    w = np.clip(w, -1e15, 1e15).astype(np.float32)
    xs, ws = split3(x)[:3], split3(w)[:3]
    d = lambda a: a.astype(np.float64)
    # every bf16 x bf16 product has <= 16 significand bits: exact in fp32 (and so in the MFMA's fp32 accumulator input)
    for a in xs:
        for b in ws:
            p = d(a) * d(b)
            assert np.array_equal(p, (a * b).astype(np.float64))
    six = d(xs[0]) * d(ws[0]) + (d(xs[0]) * d(ws[1]) + d(xs[1]) * d(ws[0])) + (d(xs[1]) * d(ws[1]) + d(xs[0]) * d(ws[2]) + d(xs[2]) * d(ws[0]))
    exact = d(x) * d(w)
    nz = exact != 0
    rel = np.abs(six[nz] - exact[nz]) / np.abs(exact[nz])
    assert rel.max() <= 2.0 ** -23, rel.max()            # the three dropped terms: x1 w2 + x2 w1 + x2 w2
    # ... typically far below rounding the exact product to fp32 once (half an ulp = 2^-24 relative)
    assert np.median(rel[rel > 0]) < 2.0 ** -27
