"""CPU tier: the oracle against the golden fixtures, and the two oracles against each other."""

import glob
import os

import numpy as np
import pytest

from oracle import kmeans_oracle, pgv_scan, pgvector_oracle as po


def _load(golden_dir, name):
    return np.load(os.path.join(golden_dir, name), allow_pickle=False)


@pytest.mark.parametrize("name", ["cosine_d8_f32.npz", "cosine_d768_f32.npz", "cosine_d1536_f16.npz"])
def test_cosine_matches_sklearn_golden(golden_dir, name):
    g = _load(golden_dir, name)
    rows, queries, sim = g["rows"], g["queries"], g["sim"]
    for j in range(queries.shape[0]):
        d = po.cosine_distance(rows, queries[j], accumulate="f64")
        np.testing.assert_allclose(d, 1.0 - np.clip(sim[:, j], -1, 1), rtol=0, atol=1e-12)
        # pgvector-style float accumulators stay within the stated fp32 tolerance of the exact value
        d32 = po.cosine_distance(rows, queries[j], accumulate="f32")
        np.testing.assert_allclose(d32, d, rtol=0, atol=1e-5)
    np.testing.assert_allclose(po.cosine_matrix(rows, queries), sim, rtol=0, atol=1e-12)


def test_golden_orderings(golden_dir):
    g = _load(golden_dir, "cosine_d8_f32.npz")
    rows, queries, sim = g["rows"], g["queries"], g["sim"]
    # m = 3 LEAST fold, top-10, default threshold
    idx, dist = po.query(rows, queries, k=10, threshold=0.8)
    expect = np.min(1.0 - sim, axis=1)
    ok = np.nonzero(expect < 0.8)[0]
    order = ok[np.lexsort((ok, expect[ok]))][:10]
    assert idx.tolist() == order.tolist()
    np.testing.assert_allclose(dist, expect[order], atol=1e-12)
    # rows 3 and 7 are identical: the tie is broken by row index
    assert list(idx[:2]) == [3, 7]
    assert dist[0] == dist[1]


def test_c_oracle_agrees_with_numpy_oracle():
    rng = np.random.default_rng(5)
    n, d = 3000, 96
    rows = rng.standard_normal((n, d)).astype(np.float32)
    rows[17] = 0.0  # zero-norm row: NaN distance, never returned
    q = rng.standard_normal((3, d)).astype(np.float32)
    groups = (np.arange(n) % 7).astype(np.int32)
    keep = rng.random(n) < 0.6
    for m in (1, 3):
        for thr in (0.8, np.inf, 0.95):
            for gf in (-1, 3):
                k = 25
                kp = keep if gf == 3 else None
                km = None if kp is None else kp & (groups == gf)
                if gf >= 0 and kp is None:
                    km = groups == gf
                ei, ed = po.query(rows, q[:m], k=k, threshold=None if np.isinf(thr) else thr, keep=km)
                ci, cd = pgv_scan.scan_topk(rows, q[:m], k, threshold=thr, group_codes=groups, group_filter=gf, keep=kp)
                assert 17 not in ci
                np.testing.assert_allclose(cd, ed, atol=1e-5)
                # ids identical except where two distances are closer than the fp32 noise
                for a, b, da in zip(ci, ei, ed):
                    if a != b:
                        db = po.cosine_distance(rows[[a]], q[0])[0] if m == 1 else None
                        assert db is None or abs(db - da) < 1e-6


def test_c_oracle_16bit_storage():
    rng = np.random.default_rng(6)
    n, d = 500, 64
    x = rng.standard_normal((n, d)).astype(np.float32)
    q = rng.standard_normal((1, d)).astype(np.float32)
    xh = x.astype(np.float16)
    qh = q.astype(np.float16).astype(np.float32)
    ei, ed = po.query(x, q, k=10, threshold=None, storage="f16")
    ci, cd = pgv_scan.scan_topk(xh.view(np.uint16), qh, 10, storage="f16")
    assert ci.tolist() == ei.tolist()
    np.testing.assert_allclose(cd, ed, atol=1e-5)
    xb = po.bf16_round(x)
    bits = (xb.view(np.uint32) >> 16).astype(np.uint16)
    qb = po.bf16_round(q)
    ei, ed = po.query(x, q, k=10, threshold=None, storage="bf16")
    ci, cd = pgv_scan.scan_topk(bits, qb, 10, storage="bf16")
    assert ci.tolist() == ei.tolist()
    np.testing.assert_allclose(cd, ed, atol=1e-5)


def test_threshold_is_strict_and_nan_excluded():
    rows = np.array([[1, 0], [0, 1], [0, 0], [-1, 0], [1, 1]], dtype=np.float32)
    q = np.array([[1, 0]], dtype=np.float32)
    # distances: 0, 1, NaN, 2, 1-1/sqrt(2)
    idx, dist = po.query(rows, q, k=10, threshold=1.0)
    assert idx.tolist() == [0, 4]
    idx, dist = po.query(rows, q, k=10, threshold=None)
    assert idx.tolist() == [0, 4, 1, 3]
    assert dist.tolist()[0] == 0.0 and dist.tolist()[-1] == 2.0
    # zero query: every distance NaN -> empty
    idx, _ = po.query(rows, np.zeros((1, 2), np.float32), k=10, threshold=None)
    assert idx.size == 0
    # LEAST ignores a NaN member unless all are NaN
    idx, dist = po.query(rows, np.array([[0, 0], [0, 1]], np.float32), k=1, threshold=None)
    assert idx.tolist() == [1] and dist[0] == 0.0


def test_dpr_restatement_orders_by_cosine():
    rng = np.random.default_rng(7)
    c = rng.standard_normal((200, 32)).astype(np.float32)
    c[5] = 0
    q = rng.standard_normal(32).astype(np.float32)
    order, ns = po.dpr_scores(c, q, 10)
    sim = po.cosine_matrix(c, q[None])[:, 0]
    assert order.tolist() == np.argsort(-sim)[:10].tolist()
    assert ns.max() == 1.0 and ns.min() == 0.0


def test_synth_rows_shape_and_moments():
    x = po.synth_rows(1234, 100, 64, 1536)
    assert x.dtype == np.float32 and x.shape == (64, 1536)
    assert abs(float(x.mean())) < 0.02 and abs(float(x.std()) - 1.0) < 0.02
    # counter-based: any sub-block regenerates identically
    np.testing.assert_array_equal(po.synth_rows(1234, 110, 5, 1536), x[10:15])


@pytest.mark.parametrize("path", sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "kmeans_*.npz"))))
def test_kmeans_oracle_matches_sklearn_golden(path):
    g = np.load(path, allow_pickle=False)
    a, margin = kmeans_oracle.assign(g["x"], g["c0"])
    diff = a != g["assign0"]
    assert not np.any(diff & (margin > 1e-4))
    assert diff.sum() == 0
    new_c, _, counts = kmeans_oracle.update(g["x"], a, g["c0"])
    np.testing.assert_allclose(new_c, g["centers1"], atol=1e-5)
    np.testing.assert_array_equal(pgv_scan.kmeans_assign(g["x"], g["c0"]), a)


def test_c_oracle_under_address_and_undefined_behaviour_sanitizers():
    """`make -C oracle asan`: pgv_scan.c built with -fsanitize=address,undefined, walked by its self-checking driver
    (scan over every element kind and odd sizes with exact-size allocations, the bounded heap at k = 1 .. > n, every
    16-bit pattern through the conversions).  CPU only."""
    import shutil
    import subprocess

    if shutil.which("gcc") is None and shutil.which("cc") is None:
        pytest.skip("no C compiler")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run(["make", "-C", os.path.join(root, "oracle"), "asan"], capture_output=True, text=True, timeout=600)
    out = r.stdout + r.stderr
    assert r.returncode == 0 and "pgv_selftest ok" in out, out[-3000:]
    assert "AddressSanitizer" not in out and "runtime error" not in out
