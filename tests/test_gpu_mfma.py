"""GPU tier: the batched MFMA search path and the k-means E/M step against the oracle."""

import math
import os

import numpy as np
import pytest

from oracle import kmeans_oracle, pgvector_oracle as po
from tests import parity

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def hvs():
    import hirag_prod_amd

    assert hirag_prod_amd.device_count() >= 1
    return hirag_prod_amd


def _check_all(rows, q, fold, k, dist, ids, cnt, thr=None, keep=None, label="", storage="f32"):
    nq = q.shape[0] // fold
    dm = parity.distance_matrix(rows, q, storage)
    for i in range(nq):
        full = po.least_fold(dm[i * fold : (i + 1) * fold])
        parity.check_topk(dist[i], ids[i], cnt[i], full, k, threshold=thr, keep=keep, label=f"{label} q={i}")


@pytest.mark.parametrize("dtype", ["f32", "f16", "bf16"])
@pytest.mark.parametrize("n,d", [(1000, 96), (20000, 256), (5000, 1536), (130, 40)])
def test_mfma_search_matches_oracle(hvs, n, d, dtype):
    rng = np.random.default_rng(n * 7 + d)
    rows = rng.standard_normal((n, d)).astype(np.float32)
    rows[3] = 0.0
    rows[8] = rows[2]
    idx = hvs.HvsIndex(d, dtype)
    idx.add(rows)
    for nq, k in ((1, 10), (37, 1), (160, 10), (64, 100), (300, 128)):
        q = rng.standard_normal((nq, d)).astype(np.float32)
        q[0] = rows[2]  # an exact hit with an exact tie (rows 2 and 8)
        dist, ids, cnt = idx.search(q, k=k, algo="mfma")
        _check_all(rows, q, 1, k, dist, ids, cnt, label=f"mfma {dtype} n={n} d={d} nq={nq} k={k}", storage=dtype)
        # the two independent kernels agree with each other as well
        d2, i2, c2 = idx.search(q[:8], k=k, algo="scan")
        np.testing.assert_array_equal(c2, cnt[:8])
        np.testing.assert_allclose(d2, dist[:8], atol=2e-6)
        if dtype == "f32":
            # "mfma" ran on the split copy (16-bit pipe): as a filter with exact fp32 scores for the candidates up to
            # k = 32, all three products beyond; those three products for every k and the fp32 MFMA kernel are two more routes
            for other in ("mfma_f32", "mfma_split"):
                d3, i3, c3 = idx.search(q, k=k, algo=other)
                _check_all(rows, q, 1, k, d3, i3, c3, label=f"{other} n={n} d={d} nq={nq} k={k}", storage=dtype)
                np.testing.assert_array_equal(c3, cnt)
                np.testing.assert_allclose(d3, dist, atol=2e-6)
    idx.close()


def test_split_copy_keeps_fp32_grade_scores(hvs):
    """fp32 rows searched on the 16-bit matrix pipe through the [hi | lo] fp16 split: rows of every
    magnitude, wide dynamic range inside a row, one-hot rows (the worst case for a dropped lo.lo term),
    appended and overwritten rows; scores stay within 2e-6 of float64 and the ranking is the oracle's."""
    rng = np.random.default_rng(2024)
    n, d = 6000, 384
    rows = rng.standard_normal((n, d)).astype(np.float32)
    rows[:500] *= np.exp(rng.uniform(-12, 0, (500, d))).astype(np.float32)      # six decades inside a row
    rows[500:1000] *= np.float32(1e15)   # |row|^2 must stay inside fp32 for every route (and for the reference's float sums)
    rows[1000:1500] *= np.float32(1e-15)
    rows[1500:1600] = 0.0
    rows[1500:1600][np.arange(100), rng.integers(0, d, 100)] = rng.uniform(0.3, 3.0, 100)  # one-hot rows
    rows[1600] = 0.0
    q = rng.standard_normal((96, d)).astype(np.float32)
    q[1] = 0.0
    q[1, 5] = 1.7   # a one-hot query against one-hot rows
    q[2] = rows[700]  # exact hits at both ends of the magnitude range
    q[3] = rows[1200]
    q[4] *= np.float32(1e-15)
    idx = hvs.HvsIndex(d, "f32")
    idx.add(rows[:4000])
    dist, ids, cnt = idx.search(q, k=10, algo="mfma")          # builds the split copy of 4000 rows
    _check_all(rows[:4000], q, 1, 10, dist, ids, cnt, label="split 4000")
    idx.add(rows[4000:])                                        # the copy is extended, not rebuilt
    new = rng.standard_normal((50, d)).astype(np.float32)
    where = rng.permutation(n)[:50]
    idx.update(where, new)                                      # overwritten rows invalidate it
    rows[where] = new
    dm = parity.distance_matrix(rows, q, "f32")
    for k, algo in ((1, "mfma_split"), (10, "mfma_split"), (100, "mfma_split"), (1, "mfma"), (10, "mfma"), (32, "mfma"), (64, "mfma")):
        dist, ids, cnt = idx.search(q, k=k, algo=algo)  # "mfma" with k <= 64: hi slabs as a filter, exact fp32 rescoring
        _check_all(rows, q, 1, k, dist, ids, cnt, label=f"{algo} k={k}")
        for i in range(q.shape[0]):
            c = int(cnt[i])
            assert np.abs(dist[i, :c] - dm[i, ids[i, :c]]).max() < 2e-6, f"q={i}"
        d3, i3, c3 = idx.search(q, k=k, algo="mfma_f32")
        np.testing.assert_array_equal(c3, cnt)
        np.testing.assert_allclose(d3, dist, atol=2e-6)
    dist, ids, cnt = idx.search(q, k=10, algo="mfma")
    assert ids[2, 0] == 700 and dist[2, 0] < 2e-6 and ids[3, 0] == 1200 and dist[3, 0] < 2e-6
    # fold over the split path, and the auto dispatch (from 16 embeddings on fp32 rows take it)
    dist, ids, cnt = idx.search(q, k=10, fold=3, algo="mfma")
    _check_all(rows, q, 3, 10, dist, ids, cnt, label="split fold 3")
    dist, ids, cnt = idx.search(q, k=10)
    _check_all(rows, q, 1, 10, dist, ids, cnt, label="split auto")
    idx.close()


@pytest.mark.parametrize("fold,dtype", [(2, "f32"), (3, "f32"), (4, "f32"), (3, "f16"), (2, "bf16")])
def test_mfma_fold_threshold_filters(hvs, fold, dtype):
    rng = np.random.default_rng(fold)
    n, d = 9000, 128
    rows = rng.standard_normal((n, d)).astype(np.float32)
    groups = (np.arange(n) % 3).astype(np.int32)
    idx = hvs.HvsIndex(d, dtype)
    idx.add(rows, group_codes=groups)
    nq = 40
    q = rng.standard_normal((nq * fold, d)).astype(np.float32)
    keep_bits = rng.random(n) < 0.5
    mask = np.zeros((n + 31) // 32, dtype=np.uint32)
    for r in np.nonzero(keep_bits)[0]:
        mask[r >> 5] |= np.uint32(1 << (r & 31))
    for gf, use_mask, thr in ((-1, False, math.inf), (1, False, 0.9), (-1, True, 0.85), (2, True, math.inf)):
        dist, ids, cnt = idx.search(q, k=10, fold=fold, group_filter=gf, row_mask=mask if use_mask else None,
                                    max_distance=thr, algo="mfma")
        keep = np.ones(n, bool)
        if gf >= 0:
            keep &= groups == gf
        if use_mask:
            keep &= keep_bits
        _check_all(rows, q, fold, 10, dist, ids, cnt, thr=None if math.isinf(thr) else thr, keep=keep,
                   label=f"fold={fold} gf={gf} mask={use_mask} thr={thr}", storage=dtype)
    # deleted rows and row ranges
    idx.set_deleted(np.arange(0, n, 2), True)
    dist, ids, cnt = idx.search(q, k=5, fold=fold, row_begin=1001, row_end=7777, algo="mfma")
    keep = np.zeros(n, bool)
    keep[1001:7777] = True
    keep[::2] = False
    _check_all(rows, q, fold, 5, dist, ids, cnt, keep=keep, label=f"fold={fold} deleted+range", storage=dtype)
    idx.close()


@pytest.mark.parametrize("dtype,algo", [("f32", "mfma_f32"), ("f32", "mfma"), ("f32", "mfma_split"), ("bf16", "mfma")])
def test_filtered_batch_runs_over_compacted_rows(hvs, dtype, algo):
    """A selective filter makes the batched path contract over a dense copy of the live rows; the answer
    must be the one the in-epilogue filter gives (same keys, ties included) and the oracle's."""
    rng = np.random.default_rng(77)
    n, d = 33_333, 200
    rows = rng.standard_normal((n, d)).astype(np.float32)
    rows[1000:1040] = rows[17]  # many exact ties inside one group and across groups
    groups = (np.arange(n) % 64).astype(np.int32)
    ext_ids = (np.arange(n, dtype=np.int64) * 3 + 5)
    idx = hvs.HvsIndex(d, dtype)
    idx.add(rows, ids=ext_ids, group_codes=groups)
    idx.set_deleted(np.arange(3, n, 640), True)  # some of group 3's rows are tombstoned
    q = rng.standard_normal((48, d)).astype(np.float32)
    q[0] = rows[17]
    keep_bits = rng.random(n) < 0.1
    mask = np.zeros((n + 31) // 32, dtype=np.uint32)
    np.bitwise_or.at(mask, np.nonzero(keep_bits)[0] >> 5, (np.uint32(1) << (np.nonzero(keep_bits)[0] & 31).astype(np.uint32)))
    deleted = np.zeros(n, bool)
    deleted[np.arange(3, n, 640)] = True
    for gf, use_mask, rb, re_ in ((3, False, 0, n), (17, False, 5001, 29_000), (-1, True, 0, n), (3, True, 100, n), (99, False, 0, n)):
        kw = dict(k=20, group_filter=gf, row_mask=mask if use_mask else None, row_begin=rb, row_end=re_, max_distance=1.2,
                  algo=algo)  # fp32 rows: the fp32 MFMA kernel, or the split copy (the dense copy is then gathered from it)
        dist, ids, cnt = idx.search(q, **kw)
        os.environ["HVS_NO_COMPACT"] = "1"
        try:
            d0, i0, c0 = idx.search(q, **kw)
        finally:
            del os.environ["HVS_NO_COMPACT"]
        np.testing.assert_array_equal(cnt, c0)
        np.testing.assert_array_equal(ids, i0)
        np.testing.assert_array_equal(dist, d0)
        keep = ~deleted
        keep[:rb] = False
        keep[re_:] = False
        if gf >= 0:
            keep &= groups == gf
        if use_mask:
            keep &= keep_bits
        dm = parity.distance_matrix(rows, q, dtype)
        for i in range(q.shape[0]):
            parity.check_topk(dist[i], ids[i], cnt[i], dm[i], 20, threshold=1.2, keep=keep, id_of_row=ext_ids,
                              label=f"compacted {dtype} gf={gf} mask={use_mask} q={i}")
        if gf == 99:
            assert int(cnt.max()) == 0
    idx.close()


def test_mfma_auto_dispatch_and_synthetic(hvs):
    n, d = 40_000, 1536
    idx = hvs.HvsIndex(d, "f32")
    idx.fill_synthetic(n, seed=1234)
    rows = po.synth_rows(1234, 0, n, d)
    q = po.synth_rows(4321, 0, 64, d)
    # planted neighbours: rows 100..109 = q[5] + 0.05 * noise
    noise = po.synth_rows(99, 0, 10, d)
    planted = q[5][None, :] + 0.05 * noise
    idx.update(np.arange(100, 110), planted)
    rows[100:110] = planted
    dist, ids, cnt = idx.search(q, k=10)  # auto -> MFMA for 64 queries
    _check_all(rows, q, 1, 10, dist, ids, cnt, label="synthetic")
    assert sorted(ids[5].tolist()) == list(range(100, 110))
    idx.close()


@pytest.mark.parametrize("route", ["default", "split", "fp32"])
@pytest.mark.parametrize("n,d,kc", [(5000, 64, 5), (3500, 72, 100), (3000, 100, 200), (4100, 1536, 300), (2500, 48, 700)])
def test_kmeans_assign_matches_oracle(hvs, n, d, kc, route, monkeypatch):
    """From 64 centroids on the default route contracts the hi slabs of the [hi | lo] fp16 split copies on the 16-bit
    pipe (128-tile up to 128 centroids, 256-tile beyond), keeps the best two values per centroid tile and row and
    settles what their bound leaves open with exact fp32 values; HVS_NO_FILTER contracts all three slab products for
    every row; HVS_NO_SPLIT keeps the fp32 MFMA kernel."""
    if route == "fp32":
        monkeypatch.setenv("HVS_NO_SPLIT", "1")
    if route == "split":
        monkeypatch.setenv("HVS_NO_FILTER", "1")
    rng = np.random.default_rng(n + kc)
    x = rng.standard_normal((n, d)).astype(np.float32)
    x[: n // 4] *= np.float32(300.0)   # rows of very different norms: every row has its own split scale
    x[n // 4 : n // 2] *= np.float32(0.004)
    c = x[rng.choice(n, kc, replace=False)].copy()
    c[1] = c[0]  # duplicate centroid: ties go to the lower index
    idx = hvs.HvsIndex(d, "f32")
    idx.add(x)
    assign, sums, counts = idx.kmeans_assign(c)
    exp, margin = kmeans_oracle.assign(x, c)
    # centroids 0 and 1 are identical: an exact tie, which must go to index 0 on the GPU as in numpy
    tie01 = exp == 0
    assert tie01.any() and not np.any(exp == 1)
    np.testing.assert_array_equal(assign[tie01], 0)
    assert not np.any(assign == 1)
    # below this an fp32 dot product may legitimately flip the argmin (the contending centroids are of the row's own size)
    xn = np.sum(x.astype(np.float64) ** 2, axis=1)
    safe = (margin > 1e-5 * xn) | tie01
    assert safe.mean() > 0.85
    np.testing.assert_array_equal(assign[safe], exp[safe])
    # rows that are themselves centroids are assigned to them (distance 0)
    _, esums, ecounts = kmeans_oracle.update(x, assign, c)
    np.testing.assert_array_equal(counts, ecounts)
    np.testing.assert_allclose(sums, esums, rtol=1e-4, atol=1e-3 * float(np.abs(esums).max()))
    # assignment only
    a2, s2, c2 = idx.kmeans_assign(c, want_sums=False)
    np.testing.assert_array_equal(a2, assign)
    assert s2 is None and c2 is None
    idx.close()


def test_kmeans_filter_flags_rows_between_near_centroids(hvs):
    """Centroids in tight bunches (several within the hi-slab bound of each other in one centroid tile): the rows near a
    bunch cannot be settled by the best two values of the tile and go through the three products; a table where most
    rows are like that takes the three products as a whole.  Exact either way."""
    from hirag_prod_amd import _lib

    rng = np.random.default_rng(77)
    n, d, kc = 6000, 96, 512
    c = rng.standard_normal((kc, d)).astype(np.float32)
    # eight bunches of four centroids about 1e-3 apart (relative), each bunch inside one 256-centroid tile
    for b in range(8):
        at = 16 + 60 * b
        c[at : at + 4] = c[at] * (1.0 + 1e-3 * rng.standard_normal((4, 1)).astype(np.float32)) + 2e-3 * rng.standard_normal((4, d)).astype(np.float32)
    bunched = np.concatenate([np.arange(16 + 60 * b, 20 + 60 * b) for b in range(8)])
    x = rng.standard_normal((n, d)).astype(np.float32)
    near = rng.choice(n, 600, replace=False)  # a tenth of the rows sit right at a bunch
    x[near] = c[rng.choice(bunched, near.size)] + 1e-2 * rng.standard_normal((near.size, d)).astype(np.float32)
    exp, margin = kmeans_oracle.assign(x, c)
    xn = np.sum(x.astype(np.float64) ** 2, axis=1)
    safe = margin > 1e-5 * xn
    idx = hvs.HvsIndex(d, "f32")
    idx.add(x)
    idx.profile(True)
    idx.profile_reset()
    assign, _, counts = idx.kmeans_assign(c)
    _, flagged = idx.profile_get(_lib.HVS_PROF_REDO)
    assert near.size // 2 < flagged <= n // 4, flagged  # the rows at a bunch, not the table
    np.testing.assert_array_equal(assign[safe], exp[safe])
    assert int(counts.sum()) == n
    # every row at a bunch: more than a quarter flagged -> the three products for the whole table, same answers
    x2 = (c[rng.choice(bunched, n)] + 1e-2 * rng.standard_normal((n, d))).astype(np.float32)
    idx2 = hvs.HvsIndex(d, "f32")
    idx2.add(x2)
    exp2, margin2 = kmeans_oracle.assign(x2, c)
    safe2 = margin2 > 1e-5 * np.sum(x2.astype(np.float64) ** 2, axis=1)
    for _ in range(2):  # the second call skips the filter pass (back-off) and must agree
        a2, _, c2 = idx2.kmeans_assign(c)
        np.testing.assert_array_equal(a2[safe2], exp2[safe2])
        assert int(c2.sum()) == n
    idx.close()
    idx2.close()


def test_kmeans_golden(hvs, golden_dir):
    for name in ("kmeans_n512_d16_k8.npz", "kmeans_n300_d64_k5.npz"):
        g = np.load(os.path.join(golden_dir, name), allow_pickle=False)
        idx = hvs.HvsIndex(g["x"].shape[1], "f32")
        idx.add(g["x"])
        assign, sums, counts = idx.kmeans_assign(g["c0"])
        np.testing.assert_array_equal(assign, g["assign0"])  # sklearn's own E-step
        new_c = sums / np.maximum(counts, 1)[:, None]
        np.testing.assert_allclose(new_c, g["centers1"], atol=1e-4)
        idx.close()


def test_kmeans_sums_are_reproducible_and_cover_small_k(hvs):
    """The M-step sums add a centroid's rows in ascending row order: two runs give the same bits; a
    handful of big clusters (sliced over many blocks) and deleted rows are handled too."""
    rng = np.random.default_rng(12)
    n, d = 20_000, 70
    rows = rng.standard_normal((n, d)).astype(np.float32)
    idx = hvs.HvsIndex(d, "f32")
    idx.add(rows)
    gone = np.arange(5, n, 97)
    idx.set_deleted(gone, True)
    for kc in (3, 257):
        cent = rows[rng.choice(n, kc, replace=False)].copy()
        a1, s1, c1 = idx.kmeans_assign(cent)
        a2, s2, c2 = idx.kmeans_assign(cent)
        np.testing.assert_array_equal(a1, a2)
        np.testing.assert_array_equal(s1, s2)  # bit for bit
        np.testing.assert_array_equal(c1, c2)
        assert np.all(a1[gone] == -1) and int(c1.sum()) == n - len(gone)
        live = a1 >= 0
        _, exp_sums, exp_counts = kmeans_oracle.update(rows[live], a1[live], cent)
        np.testing.assert_array_equal(c1, exp_counts)
        np.testing.assert_allclose(s1, exp_sums, rtol=1e-4, atol=1e-3)
    idx.close()


def test_filter_pass_falls_back_when_its_bound_proves_nothing(hvs):
    """fp32 table, k <= 64: the hi-slab filter keeps 4 k candidates per query and the answer stands only if the k-th
    exact distance clears the bound of the rows left out.  Clouds of near-duplicates around a query defeat that
    bound (hundreds of rows within 1e-4 of each other, the filter errs by up to 1e-3): such queries go through the
    filter again with 256 candidates, and what that cannot prove either (clouds of 400 rows here) must come back
    exact all the same -- searched again by a scan when it is one query, as a batch with all three products
    otherwise."""
    from hirag_prod_amd import _lib

    rng = np.random.default_rng(31337)
    n, d = 30_000, 256
    rows = rng.standard_normal((n, d)).astype(np.float32)
    centre = rng.standard_normal((3, d)).astype(np.float32)
    for c in range(3):  # 400 rows within ~3e-5 (cosine distance) of each centre, distinct from one another
        rows[1000 * (c + 1) : 1000 * (c + 1) + 400] = centre[c] + 8e-3 * rng.standard_normal((400, d)).astype(np.float32)
    idx = hvs.HvsIndex(d, "f32")
    idx.add(rows)
    q = rng.standard_normal((200, d)).astype(np.float32)
    q[5], q[77] = centre[0], centre[2]
    dm = parity.distance_matrix(rows, q, "f32")

    def run(queries, k, **kw):
        idx.search(queries, k=k, algo="mfma", **kw)
        idx.profile(True)
        idx.profile_reset()
        out = idx.search(queries, k=k, algo="mfma", **kw)
        scans, full = idx.profile_get(_lib.HVS_PROF_SCAN)[1], idx.profile_get(_lib.HVS_PROF_MFMA_SEARCH)[1]
        idx.profile(False)
        return out, scans, full

    # two unproven queries out of 200: the filter again with the long lists (their clouds hold more rows than a list),
    # then as a batch of their own, all three products
    (dist, ids, cnt), scans, full = run(q, 10)
    assert scans == 0 and full == 3
    for i in range(200):
        parity.check_topk(dist[i], ids[i], cnt[i], dm[i], 10, label=f"few q={i}")
    # the same with a threshold that cuts inside the cloud and a row range
    thr = float(np.sort(dm[5])[25])
    (dist, ids, cnt), scans, full = run(q, 32, max_distance=thr, row_begin=500, row_end=25_000)
    keep = np.zeros(n, bool)
    keep[500:25_000] = True
    for i in range(200):
        parity.check_topk(dist[i], ids[i], cnt[i], dm[i], 32, threshold=thr, keep=keep, label=f"thr q={i}")
    # a single unproven query: one exact scan
    (dist1, ids1, cnt1), scans, full = run(q[:40], 10)
    assert scans == 1 and full == 1
    for i in range(40):
        parity.check_topk(dist1[i], ids1[i], cnt1[i], dm[i], 10, label=f"one q={i}")
    # every query inside a cloud: both filter passes prove nothing, the whole batch is searched again with all three products
    qc = (centre[rng.integers(0, 3, 64)] + 1e-3 * rng.standard_normal((64, d))).astype(np.float32)
    os.environ["HVS_NO_ADAPT"] = "1"  # pin the route: a failed filter pass otherwise keeps the next calls off it
    try:
        (dist, ids, cnt), scans, full = run(qc, 10)
    finally:
        del os.environ["HVS_NO_ADAPT"]
    assert scans == 0 and full == 3
    dmc = parity.distance_matrix(rows, qc, "f32")
    for i in range(64):
        parity.check_topk(dist[i], ids[i], cnt[i], dmc[i], 10, label=f"many q={i}")
    # left to itself the table remembers: the call after a batch whose lists proved nothing skips the filter pass (one
    # kernel, all three products), four calls later the filter is tried again; the answers do not change
    (dist2, ids2, cnt2), scans, full = run(qc, 10)
    assert scans == 0 and full == 1
    assert np.array_equal(ids2, ids) and np.array_equal(cnt2, cnt)
    for _ in range(3):
        idx.search(qc, k=10, algo="mfma")
    (dist3, ids3, cnt3), scans, full = run(q, 10)  # run() searches twice: the filter is back (with the long lists from the start: the table remembers that too), and proves 198 of 200 lists
    assert scans == 0 and full == 2
    for i in range(200):
        parity.check_topk(dist3[i], ids3[i], cnt3[i], dm[i], 10, label=f"after the back-off q={i}")
    # nothing to prove when the candidate list was not cut: fewer live rows than candidates
    (dist, ids, cnt), scans, full = run(q[:20], 10, row_begin=1000, row_end=1030)
    assert scans == 0 and full == 1
    keep = np.zeros(n, bool)
    keep[1000:1030] = True
    for i in range(20):
        parity.check_topk(dist[i], ids[i], cnt[i], dm[i], 10, keep=keep, label=f"short q={i}")
    idx.close()


def test_filter_route_settles_clustered_tables_with_the_long_lists(hvs):
    """A table of tight clusters (about 120 rows within 1e-4 .. 1e-3 of each other around every query, the next cluster far
    away): 40 candidates prove nothing, 256 hold the whole cluster and prove everything -- no query is searched again in
    full, and after one such batch the table starts with the long lists."""
    from hirag_prod_amd import _lib

    rng = np.random.default_rng(4242)
    d, n_clusters, per = 192, 60, 120
    centres = rng.standard_normal((n_clusters, d)).astype(np.float32)
    rows = (np.repeat(centres, per, axis=0) + 2e-2 * rng.standard_normal((n_clusters * per, d))).astype(np.float32)
    rows = rows[rng.permutation(len(rows))]
    q = (centres[rng.integers(0, n_clusters, 96)] + 2e-2 * rng.standard_normal((96, d))).astype(np.float32)
    dm = parity.distance_matrix(rows, q, "f32")
    idx = hvs.HvsIndex(d, "f32")
    idx.add(rows)
    for call, launches in ((0, 2), (1, 1)):  # first call: short lists, then long ones for the flagged queries; second: long ones at once
        idx.profile(True)
        idx.profile_reset()
        dist, ids, cnt = idx.search(q, k=10, algo="mfma")
        full, again = idx.profile_get(_lib.HVS_PROF_MFMA_SEARCH)[1], idx.profile_get(_lib.HVS_PROF_REDO)[1]
        scans = idx.profile_get(_lib.HVS_PROF_SCAN)[1]
        idx.profile(False)
        assert (full, again, scans) == (launches, 0, 0), (call, full, again, scans)
        for i in range(96):
            parity.check_topk(dist[i], ids[i], cnt[i], dm[i], 10, label=f"clustered call {call} q={i}")
    idx.close()


def test_filter_route_on_ladders_of_close_rows(hvs):
    """Rows at cosine distances delta, 2 delta, 3 delta, ... from a query, for spacings from far below to about the
    filter's error bound (1e-3): the hi-slab scores cannot order such rows, so the answer is right only if the
    proof refuses every list it cannot vouch for.  Exact order expected at every spacing (the spacings stay above
    the tie tolerance), with rows of mixed magnitude."""
    rng = np.random.default_rng(99)
    n_fill, d, steps = 6000, 384, 300
    spacings = (1e-5, 3e-5, 1e-4, 3e-4, 1e-3)
    rows = [rng.standard_normal((n_fill, d))]
    q = rng.standard_normal((len(spacings), d))
    for s, delta in enumerate(spacings):
        qh = q[s] / np.linalg.norm(q[s])
        u = rng.standard_normal((steps, d))
        u -= (u @ qh)[:, None] * qh[None, :]
        u /= np.linalg.norm(u, axis=1, keepdims=True)
        cos = 1.0 - delta * np.arange(1, steps + 1)
        ladder = cos[:, None] * qh[None, :] + np.sqrt(1.0 - cos * cos)[:, None] * u
        rows.append(ladder * np.exp(rng.uniform(-3, 3, (steps, 1))))
    rows = np.concatenate(rows).astype(np.float32)
    perm = rng.permutation(rows.shape[0])
    rows = rows[perm]
    q = q.astype(np.float32)
    idx = hvs.HvsIndex(d, "f32")
    idx.add(rows)
    dm = parity.distance_matrix(rows, q, "f32")
    for k in (1, 10, 32, 64):
        for algo in ("mfma", "auto"):
            qq = np.tile(q, (4, 1)) if algo == "auto" else q  # 20 embeddings: AUTO takes the matrix pipe
            dist, ids, cnt = idx.search(qq, k=k, algo=algo)
            for i in range(qq.shape[0]):
                full = dm[i % len(spacings)]
                parity.check_topk(dist[i], ids[i], cnt[i], full, k, label=f"ladder {spacings[i % len(spacings)]} k={k} {algo}")
                exp = np.argsort(full, kind="stable")[:k]
                assert ids[i][:k].tolist() == exp.tolist(), f"ladder {spacings[i % len(spacings)]} k={k} {algo}: order"
    # thresholds that cut a ladder between two rungs
    for s, delta in enumerate(spacings):
        thr = float(delta * 7.5)
        dist, ids, cnt = idx.search(q[s : s + 1], k=32, max_distance=thr, algo="mfma")
        assert cnt[0] == 7 and ids[0][:7].tolist() == np.argsort(dm[s], kind="stable")[:7].tolist(), f"thr on ladder {delta}"
    idx.close()


def _rows_at_rounding_midpoints(rng, n, d, down):
    """Positive rows whose scaled elements lie 3.99 above (down[i]: rounded down by fp16) or 3.99 below (rounded up) a
    point of the fp16 grid of [2^13, 2^14) (spacing 8); one exact element pins every row's scale at 1."""
    j = rng.integers(1024, 2047, (n, d))
    x = 8.0 * j + np.where(down[:, None], 3.99, -3.99)
    x[:, 0] = 2.0 ** 14 - 8
    return x.astype(np.float32)


def test_filter_bound_when_rounding_errors_line_up(hvs):
    """The filter passes bound the error of a hi-slab product by how far THESE rows and queries lie from their hi slabs
    (measured when they are split), which for ordinary data is well under the 2^-10 of the rounding unit.  Rows built to
    use that bound up: every element of a row rounds the same way by almost half a grid step, half of the rows up and
    half down, against all-positive queries -- the hi-slab scores of the two halves are off by 3e-4 in opposite
    directions while the rows lie about 1e-3 apart, so the approximate order is wrong everywhere.  Exact answers
    expected from the searches and from the k-means assignment (whose centroids are such rows too)."""
    rng = np.random.default_rng(2718)
    n, d = 6000, 256
    down = rng.random(n) < 0.5
    rows = _rows_at_rounding_midpoints(rng, n, d, down)
    q_exact = (8.0 * rng.integers(1024, 2047, (24, d))).astype(np.float32)  # on the grid: the queries' own error is zero
    q_mid = _rows_at_rounding_midpoints(rng, 24, d, rng.random(24) < 0.5)    # ... and queries that round like the rows
    q = np.concatenate([q_exact, q_mid])
    dm = parity.distance_matrix(rows, q, "f32")
    # the construction does what it says: hi slabs alone misplace the two halves by about 3e-4 against each other
    hi = rows.astype(np.float16).astype(np.float64)
    approx = 1.0 - (q_exact.astype(np.float64) @ hi.T) / (np.linalg.norm(q_exact.astype(np.float64), axis=1)[:, None] * np.linalg.norm(rows.astype(np.float64), axis=1)[None, :])
    shift = approx - dm[:24]
    assert shift[:, down].min() > 2.5e-4 and shift[:, ~down].max() < -2.5e-4
    idx = hvs.HvsIndex(d, "f32")
    idx.add(rows)
    for k in (1, 10, 32):
        for algo in ("mfma", "auto"):
            dist, ids, cnt = idx.search(q, k=k, algo=algo)
            for i in range(q.shape[0]):
                parity.check_topk(dist[i], ids[i], cnt[i], dm[i], k, label=f"lined-up k={k} {algo} q={i}")
    thr = float(np.sort(dm[3])[5] + 1e-6)
    dist, ids, cnt = idx.search(q, k=10, max_distance=thr, algo="mfma")
    for i in range(q.shape[0]):
        parity.check_topk(dist[i], ids[i], cnt[i], dm[i], 10, threshold=thr, label=f"lined-up thr q={i}")
    # k-means: 128 of the rows as centroids
    c = rows[rng.choice(n, 128, replace=False)].copy()
    assign, _, counts = idx.kmeans_assign(c)
    exp, margin = kmeans_oracle.assign(rows, c)
    safe = margin > 1e-5 * np.sum(rows.astype(np.float64) ** 2, axis=1)
    assert safe.mean() > 0.9
    np.testing.assert_array_equal(assign[safe], exp[safe])
    assert int(counts.sum()) == n
    idx.close()


def test_kmeans_on_an_empty_index_returns_zero_sums(hvs):
    """A rank whose shard is empty must feed zeros into the centroid all-reduce."""
    import torch

    idx = hvs.HvsIndex(32, "f32")
    c = np.random.default_rng(1).standard_normal((5, 32)).astype(np.float32)
    assign, sums, counts = idx.kmeans_assign(c)
    assert assign.shape == (0,) and not sums.any() and not counts.any()
    ct = torch.from_numpy(c).cuda()
    assign, sums, counts = idx.kmeans_assign(ct)
    assert int(sums.abs().sum().item()) == 0 and int(counts.sum().item()) == 0
    idx.close()


def test_timing_hooks_are_refused_by_a_library_built_without_them(hvs, monkeypatch):
    """HVS_DBG switches parts of the batched kernel off for timing experiments (wrong results); the library the tests
    run against is built without those hooks and must say so instead of ignoring the variable."""
    rng = np.random.default_rng(1)
    idx = hvs.HvsIndex(64, "f16")
    idx.add(rng.standard_normal((4096, 64)).astype(np.float32))
    q = rng.standard_normal((200, 64)).astype(np.float32)
    monkeypatch.setenv("HVS_DBG", "8")
    monkeypatch.setenv("HVS_STREAM", "0")
    with pytest.raises(hvs.HvsError, match="HOOKS=1"):
        idx.search(q, k=5, algo="mfma")
    monkeypatch.delenv("HVS_DBG")
    dist, ids, cnt = idx.search(q, k=5, algo="mfma")
    assert np.all(cnt == 5)
    idx.close()
