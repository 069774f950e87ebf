"""The arithmetic behind csrc/conv1d_x3.hip and csrc/attn_full_x3.hip, restated in numpy (no GPU): an fp32 number is exactly the sum
of three bf16 numbers, and six of the nine pairwise products of two such triples reproduce an fp32 dot product to fp32 quality.
(The device code does the same bit operations -- common.h:split8_bf16x3, bigvgan.hip:put -- and the GPU tests hold its results to the
oracle; this file pins the scheme itself.)"""
import numpy as np


def split3(x):
    """x (float32) -> h, m, l (float32, each representable in bf16): the remainder before each piece, rounded to nearest (ties away)."""
    x = np.asarray(x, dtype=np.float32)
    mask = np.uint32(0xFFFF0000)

    def top(v):
        return ((v.view(np.uint32) + np.uint32(0x8000)) & mask).view(np.float32)

    h = top(x)
    r = (x - h).astype(np.float32)
    m = top(r)
    return h, m, (r - m).astype(np.float32)


def test_three_bf16_pieces_sum_to_the_fp32_value_exactly():
    rng = np.random.default_rng(0)
    x = np.concatenate([
        (rng.standard_normal(200000) * np.exp(rng.uniform(-30, 30, 200000))).astype(np.float32),
        np.array([0.0, -0.0, 1.0, -1.0, 3.0, 1 + 2.0 ** -23, 1 - 2.0 ** -24, 255.99998, 65504.0, 1e-30, -1e-30, 3.3e38], dtype=np.float32),  # (|x| within half a bf16 ulp of FLT_MAX would round up to inf)
        np.float32(1.0) + np.arange(4096, dtype=np.float32) * np.float32(2.0 ** -23),  # every low-bit pattern of one binade
    ])
    h, m, l = split3(x)
    for p in (h, m, l):
        assert not np.any(p.view(np.uint32) & np.uint32(0xFFFF)), "a piece is not representable in bf16"
    assert np.array_equal((h.astype(np.float64) + m.astype(np.float64) + l.astype(np.float64)).astype(np.float32), x)
    assert np.array_equal(h.astype(np.float64) + m.astype(np.float64) + l.astype(np.float64), x.astype(np.float64))  # no rounding in the sum at all
    # the pieces shrink by 2^-8 each (what makes the three dropped products negligible)
    nz = x != 0
    assert np.all(np.abs(m[nz]) <= np.abs(x[nz]) * 2.0 ** -8) and np.all(np.abs(l[nz]) <= np.abs(x[nz]) * 2.0 ** -16)


def test_six_partial_products_give_an_fp32_quality_dot_product():
    rng = np.random.default_rng(1)
    for K in (512, 2560, 8448):
        x = rng.standard_normal((64, K)).astype(np.float32)
        w = (rng.standard_normal((K, 48)) / np.sqrt(K)).astype(np.float32)
        ref = x.astype(np.float64) @ w.astype(np.float64)
        xh, xm, xl = (p.astype(np.float64) for p in split3(x))
        wh, wm, wl = (p.astype(np.float64) for p in split3(w))
        # every partial product of two bf16 numbers is exact in fp32 (8 x 8 significand bits); summed here in float64 to isolate the
        # error of DROPPING l*m', m*l', l*l' from the error of the accumulation order
        six = xl @ wh + xh @ wl + xm @ wm + xm @ wh + xh @ wm + xh @ wh
        dropped = np.abs(six - ref).max() / np.abs(ref).max()
        f32 = np.abs((x @ w).astype(np.float64) - ref).max() / np.abs(ref).max()  # an fp32 GEMM's own rounding, for scale
        assert dropped < 2.0 ** -21, (K, dropped)          # a few units of 2^-24 per term, not accumulating coherently
        assert dropped < 0.5 * max(f32, 2.0 ** -24) + 2.0 ** -22, (K, dropped, f32)


def test_truncated_pieces_would_be_biased():
    """Why the pieces are rounded, not truncated: with truncation every piece has the sign of its value and the dropped products of a
    same-sign dot product all err one way."""
    rng = np.random.default_rng(2)
    K = 4096
    x = np.abs(rng.standard_normal((32, K))).astype(np.float32)
    w = np.abs(rng.standard_normal((K, 16))).astype(np.float32)
    ref = x.astype(np.float64) @ w.astype(np.float64)

    def trunc3(v):
        mask = np.uint32(0xFFFF0000)
        h = (v.view(np.uint32) & mask).view(np.float32)
        r = (v - h).astype(np.float32)
        m = (r.view(np.uint32) & mask).view(np.float32)
        return h, m, (r - m).astype(np.float32)

    def six(split):
        xh, xm, xl = (p.astype(np.float64) for p in split(x))
        wh, wm, wl = (p.astype(np.float64) for p in split(w))
        return xl @ wh + xh @ wl + xm @ wm + xm @ wh + xh @ wm + xh @ wh

    e_round, e_trunc = (six(split3) - ref) / ref, (six(trunc3) - ref) / ref
    assert np.all(e_trunc <= 0) and abs(e_trunc.mean()) > 20 * abs(e_round.mean())
