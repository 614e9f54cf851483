"""CPU tier: the error bound behind the hi-slab filter passes (search: ``hvs_search_rescore``; k-means:
``kmeans_pick_kernel``), checked on an emulation of the split in numpy.

A row a is scaled by a power of two s so that max|a s| lies in [2^13, 2^14) and stored as hi = fp16(a s),
lo = fp16(a s - hi).  The filter contracts hi slabs only; the library claims
    |q.x - (qh.xh) / (s_q s_x)| <= 2^-10 (1 + 2^-10) |q| |x|
and adds dim * 2^-24 for the fp32 accumulation of the exact fp16 x fp16 products.  Everything here is float64
arithmetic on exactly representable inputs, so the first claim is tested as stated."""

import numpy as np
import pytest

BOUND = 2.0 ** -10 * (1.0 + 2.0 ** -10)


def split_hi(a):
    """(hi slab as float64, scale) of each fp32 row of a, the way split_rows_kernel forms them"""
    a = np.asarray(a, dtype=np.float32)
    m = np.abs(a).max(axis=1)
    ex = np.zeros(len(a), dtype=np.int64)
    nz = m > 0
    ex[nz] = 14 - np.frexp(m[nz])[1]  # m = f 2^e with f in [0.5, 1)  ->  m 2^(14 - e) in [2^13, 2^14)
    s = np.ldexp(1.0, ex)
    with np.errstate(over="raise"):
        hi = (a.astype(np.float64) * s[:, None]).astype(np.float32).astype(np.float16)
    return hi.astype(np.float64), s


def check(q, x):
    qh, sq = split_hi(q)
    xh, sx = split_hi(x)
    exact = q.astype(np.float64) @ x.astype(np.float64).T
    approx = (qh @ xh.T) / (sq[:, None] * sx[None, :])
    qn = np.linalg.norm(q.astype(np.float64), axis=1)
    xn = np.linalg.norm(x.astype(np.float64), axis=1)
    err = np.abs(exact - approx)
    lim = BOUND * qn[:, None] * xn[None, :]
    assert np.all(err <= lim * (1 + 1e-12)), float((err / np.maximum(lim, 1e-300)).max())
    return float((err / np.maximum(lim, 1e-300)).max())


@pytest.mark.parametrize("dim", [8, 96, 768, 1536])
def test_bound_holds_on_random_rows_of_any_scale(dim):
    rng = np.random.default_rng(dim)
    q = rng.standard_normal((16, dim)).astype(np.float32)
    x = rng.standard_normal((64, dim)).astype(np.float32)
    x[:16] *= np.float32(3e4)
    x[16:32] *= np.float32(1e-6)
    x[32:40] *= (10.0 ** rng.uniform(-3, 3, (8, dim))).astype(np.float32)  # six decades inside a row
    worst = check(q, x)
    assert worst < (0.5 if dim >= 96 else 1.0)  # random rounding errors do not line up: in long rows far inside the worst case


def test_bound_holds_when_every_rounding_error_lines_up():
    """Elements just above a rounding midpoint of the fp16 grid (relative error close to 2^-11 each, all of one sign)
    in both operands, all products positive: the case the bound is written for."""
    dim = 512
    # values in [2^13, 2^14) after scaling: fp16 spacing there is 8, so v = 8 j + 4 - tiny rounds down by almost 4
    j = np.arange(1024 + 1, 1024 + 1 + dim)  # 8 j in [2^13, 2^14)
    v = (8.0 * j + 3.99).astype(np.float32)  # max element keeps the scale at 1
    v[0] = np.float32(2.0 ** 14 - 8)  # the row's largest element pins the exponent
    q = v[None, :].copy()
    x = v[None, ::-1].copy()
    x[0, -1] = np.float32(2.0 ** 14 - 8)
    worst = check(q, x)
    assert 0.2 < worst <= 1.0  # lined-up errors reach a good part of the bound, and stay inside it
    # the smallest magnitudes of the binade, where the relative spacing is widest
    lo_v = np.full((1, dim), 2.0 ** 13 + 3.99, dtype=np.float32)
    lo_v[0, 0] = np.float32(2.0 ** 14 - 8)
    assert check(lo_v, lo_v.copy()) <= 1.0


def test_kmeans_value_bound_follows():
    """|c|^2 - 2 x.c from the hi slabs errs by at most 2 * BOUND * |x| |c| (the quantity kmeans_pick_kernel widens
    every value by, before the 2^-15 of the index byte)."""
    rng = np.random.default_rng(5)
    x = rng.standard_normal((200, 256)).astype(np.float32)
    c = x[:32] * np.float32(1.001) + np.float32(1e-3) * rng.standard_normal((32, 256)).astype(np.float32)
    xh, sx = split_hi(x)
    ch, sc = split_hi(c)
    cn = np.sum(c.astype(np.float64) ** 2, axis=1)
    exact = cn[None, :] - 2.0 * (x.astype(np.float64) @ c.astype(np.float64).T)
    approx = cn[None, :] - 2.0 * (xh @ ch.T) / (sx[:, None] * sc[None, :])
    lim = 2.0 * BOUND * np.linalg.norm(x.astype(np.float64), axis=1)[:, None] * np.sqrt(cn)[None, :]
    assert np.all(np.abs(exact - approx) <= lim)
    # and the selection rule: a centroid whose lower bound lies above the smallest upper bound is never the argmin
    lo, hi = approx - lim, approx + lim
    possible = lo <= hi.min(axis=1, keepdims=True)
    best = exact.argmin(axis=1)
    assert np.all(possible[np.arange(len(x)), best])


def measured_rel(a):
    """|a s - hi| / |a s| per row, the quantity split_rows_kernel hands to the filter passes"""
    hi, s = split_hi(a)
    scaled = np.asarray(a, dtype=np.float32).astype(np.float64) * s[:, None]
    n = np.linalg.norm(scaled, axis=1)
    return np.where(n > 0, np.linalg.norm(scaled - hi, axis=1) / np.maximum(n, 1e-300), 0.0)


def check_measured(q, x):
    """|q.x - qh.xh / scales| <= (rel_x + rel_q (1 + rel_x)) |q| |x|: Cauchy-Schwarz on q.(x - xh) + (q - qh).xh"""
    qh, sq = split_hi(q)
    xh, sx = split_hi(x)
    exact = q.astype(np.float64) @ x.astype(np.float64).T
    approx = (qh @ xh.T) / (sq[:, None] * sx[None, :])
    qn = np.linalg.norm(q.astype(np.float64), axis=1)
    xn = np.linalg.norm(x.astype(np.float64), axis=1)
    rq, rx = measured_rel(q), measured_rel(x)
    lim = (rx[None, :] + rq[:, None] * (1.0 + rx[None, :])) * qn[:, None] * xn[None, :]
    err = np.abs(exact - approx)
    assert np.all(err <= lim * (1 + 1e-9) + 1e-300), float((err / np.maximum(lim, 1e-300)).max())
    assert np.all(rx <= BOUND / 2 * (1 + 1e-6)) and np.all(rq <= BOUND / 2 * (1 + 1e-6))  # never looser than the rounding unit's
    return float((err / np.maximum(lim, 1e-300)).max()), float(rx.max())


@pytest.mark.parametrize("dim", [8, 96, 768, 1536])
def test_measured_bound_holds_and_is_tighter_on_random_rows(dim):
    rng = np.random.default_rng(100 + dim)
    q = rng.standard_normal((16, dim)).astype(np.float32)
    x = rng.standard_normal((64, dim)).astype(np.float32)
    x[:16] *= np.float32(3e4)
    x[16:32] *= np.float32(1e-6)
    x[32:40] *= (10.0 ** rng.uniform(-3, 3, (8, dim))).astype(np.float32)
    worst, rx = check_measured(q, x)
    assert worst <= 1.0
    if dim >= 768:
        assert rx < 0.6 * BOUND / 2  # rows of a normal distribution sit at about half of the rounding unit's bound


def test_measured_bound_when_every_rounding_error_lines_up():
    dim = 512
    j = np.arange(1024 + 1, 1024 + 1 + dim)
    v = (8.0 * j + 3.99).astype(np.float32)
    v[0] = np.float32(2.0 ** 14 - 8)
    ones = np.full((1, dim), 8.0 * 1500, dtype=np.float32)  # on the grid: no error of its own, and parallel to the row's error
    worst, _ = check_measured(ones, v[None, :])
    assert 0.9 < worst <= 1.0  # the bound is attained, not exceeded
    worst, _ = check_measured(v[None, :], v[None, ::-1].copy())
    assert 0.5 < worst <= 1.0
