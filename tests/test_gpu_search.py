"""GPU tier: the HIP scan path (through the C ABI) against the oracle."""

import math
import os

import numpy as np
import pytest

from oracle import pgv_scan, pgvector_oracle as po
from tests import parity

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def hvs():
    import hirag_prod_amd

    assert hirag_prod_amd.device_count() >= 1, "no GPU visible to libhvs"
    return hirag_prod_amd


def _mk(hvs, rows, dtype, groups=None):
    idx = hvs.HvsIndex(rows.shape[1], dtype)
    idx.add(rows, group_codes=groups)
    assert len(idx) == rows.shape[0]
    return idx


@pytest.mark.parametrize("dtype", ["f32", "f16", "bf16"])
@pytest.mark.parametrize("n,d", [(3000, 8), (2500, 100), (4096, 768), (3001, 1536)])
def test_scan_topk_matches_oracle(hvs, dtype, n, d):
    rng = np.random.default_rng(n + d)
    rows = rng.standard_normal((n, d)).astype(np.float32)
    rows[5] = 0.0  # zero-norm row -> NaN distance, never returned
    rows[9] = rows[4]  # exact tie
    q = rng.standard_normal((6, d)).astype(np.float32)
    idx = _mk(hvs, rows, dtype)
    for k in (1, 10, 100, 300) + ((1000,) if d == 100 else ()):
        dist, ids, cnt = idx.search(q, k=k, algo="scan")
        for i in range(q.shape[0]):
            full = parity.full_distances(rows, q[i], dtype)
            parity.check_topk(dist[i], ids[i], cnt[i], full, k, label=f"{dtype} n={n} d={d} k={k} q={i}")
            assert 5 not in ids[i][: cnt[i]]
    # fold = 3 (LEAST over three embeddings), threshold 0.8 as the reference's default
    dist, ids, cnt = idx.search(q, k=10, fold=3, max_distance=0.95, algo="scan")
    for i in range(2):
        full = parity.full_distances(rows, q[3 * i : 3 * i + 3], dtype)
        parity.check_topk(dist[i], ids[i], cnt[i], full, 10, threshold=0.95, label=f"fold3 {dtype} q={i}")
    idx.close()


def test_filters_threshold_deleted(hvs):
    rng = np.random.default_rng(11)
    n, d = 6000, 64
    rows = rng.standard_normal((n, d)).astype(np.float32)
    groups = (np.arange(n) % 5).astype(np.int32)
    q = rng.standard_normal((2, d)).astype(np.float32)
    idx = _mk(hvs, rows, "f32", groups)
    keep_bits = rng.random(n) < 0.3
    mask = np.zeros((n + 31) // 32, dtype=np.uint32)
    for r in np.nonzero(keep_bits)[0]:
        mask[r >> 5] |= np.uint32(1 << (r & 31))
    for gf in (-1, 2):
        for use_mask in (False, True):
            for thr in (math.inf, 0.9, 0.7):
                dist, ids, cnt = idx.search(q, k=20, group_filter=gf, row_mask=mask if use_mask else None,
                                            max_distance=thr, algo="scan")
                keep = np.ones(n, bool)
                if gf >= 0:
                    keep &= groups == gf
                if use_mask:
                    keep &= keep_bits
                for i in range(2):
                    full = parity.full_distances(rows, q[i], "f32")
                    parity.check_topk(dist[i], ids[i], cnt[i], full, 20, threshold=None if math.isinf(thr) else thr,
                                      keep=keep, label=f"gf={gf} mask={use_mask} thr={thr}")
    # delete the current best rows: they must disappear, and come back when undeleted
    dist, ids, cnt = idx.search(q[:1], k=5, algo="scan")
    gone = ids[0][:3].copy()
    idx.set_deleted(gone, True)
    d2, i2, c2 = idx.search(q[:1], k=5, algo="scan")
    keep = np.ones(n, bool)
    keep[gone] = False
    parity.check_topk(d2[0], i2[0], c2[0], parity.full_distances(rows, q[0], "f32"), 5, keep=keep, label="deleted")
    idx.set_deleted(gone, False)
    d3, i3, c3 = idx.search(q[:1], k=5, algo="scan")
    assert i3[0].tolist() == ids[0].tolist()
    # row range
    d4, i4, c4 = idx.search(q[:1], k=5, row_begin=1000, row_end=2000, algo="scan")
    keep = np.zeros(n, bool)
    keep[1000:2000] = True
    parity.check_topk(d4[0], i4[0], c4[0], parity.full_distances(rows, q[0], "f32"), 5, keep=keep, label="range")
    idx.close()


def test_edge_cases(hvs):
    d = 16
    idx = hvs.HvsIndex(d, "f32")
    q = np.ones((1, d), np.float32)
    # empty index
    dist, ids, cnt = idx.search(q, k=10, algo="scan")
    assert cnt[0] == 0 and np.all(np.isinf(dist)) and np.all(ids == -1)
    rows = np.zeros((5, d), np.float32)
    rows[0, 0] = 1.0  # distance 1 - 1/4 = 0.75
    rows[1] = 1.0  # distance 0
    rows[2] = -1.0  # distance 2
    rows[3, 0] = 1.0
    rows[3, 1] = -1.0  # orthogonal: distance exactly 1
    idx.add(rows)
    dist, ids, cnt = idx.search(q, k=10, algo="scan")
    assert cnt[0] == 4 and ids[0][:4].tolist() == [1, 0, 3, 2]  # row 4 is all-zero: NaN, excluded
    np.testing.assert_allclose(dist[0][:4], [0.0, 0.75, 1.0, 2.0], atol=1e-6)
    # strict threshold: distance == 1.0 is excluded by "< 1.0"
    dist, ids, cnt = idx.search(q, k=10, max_distance=1.0, algo="scan")
    assert ids[0][: cnt[0]].tolist() == [1, 0]
    # zero query: every distance NaN
    dist, ids, cnt = idx.search(np.zeros((1, d), np.float32), k=3, algo="scan")
    assert cnt[0] == 0
    # LEAST ignores the NaN member
    dist, ids, cnt = idx.search(np.stack([np.zeros(d, np.float32), np.ones(d, np.float32)]), k=1, fold=2, algo="scan")
    assert cnt[0] == 1 and ids[0][0] == 1
    # external ids
    idx2 = hvs.HvsIndex(d, "f16")
    idx2.add(rows, ids=np.array([100, 200, 300, 400, 500]))
    dist, ids, cnt = idx2.search(q, k=2, algo="scan")
    assert ids[0].tolist() == [200, 100]
    # bad arguments surface as errors, not crashes
    with pytest.raises(hvs.HvsError):
        idx.search(q, k=0)
    with pytest.raises(hvs.HvsError):
        idx.search(q, k=5000)
    with pytest.raises(ValueError):
        idx.search(np.ones((1, d + 1), np.float32))
    idx.close()
    idx2.close()


def test_golden_fixtures_on_gpu(hvs, golden_dir):
    for name, dtype in (("cosine_d8_f32.npz", "f32"), ("cosine_d768_f32.npz", "f32"), ("cosine_d1536_f16.npz", "f16")):
        g = np.load(os.path.join(golden_dir, name), allow_pickle=False)
        rows, queries, sim = g["rows"], g["queries"], g["sim"]
        idx = _mk(hvs, rows, dtype)
        m = queries.shape[0]
        dist, ids, cnt = idx.search(queries.astype(np.float32), k=10, fold=m, algo="scan")
        full = np.min(1.0 - np.clip(sim, -1, 1), axis=1)
        parity.check_topk(dist[0], ids[0], cnt[0], full, 10, label=name)
        # the full distance column agrees too
        col = idx.distances(queries.astype(np.float32))
        np.testing.assert_allclose(col, full, atol=1e-5)
        idx.close()


def test_gather_and_update(hvs):
    rng = np.random.default_rng(3)
    rows = rng.standard_normal((300, 40)).astype(np.float32)
    for dtype in ("f32", "f16", "bf16"):
        idx = _mk(hvs, rows, dtype)
        pos = np.array([7, 0, 299, 7])
        got = idx.gather(pos)
        np.testing.assert_array_equal(got, po.round_storage(rows[pos], dtype).astype(np.float32))
        new = rng.standard_normal((2, 40)).astype(np.float32)
        idx.update(np.array([3, 4]), new)
        np.testing.assert_array_equal(idx.gather([3, 4]), po.round_storage(new, dtype).astype(np.float32))
        rows2 = rows.copy()
        rows2[[3, 4]] = new
        dist, ids, cnt = idx.search(new[:1], k=3, algo="scan")
        parity.check_topk(dist[0], ids[0], cnt[0], parity.full_distances(rows2, new[0], dtype), 3, label="update")
        idx.close()


def test_synthetic_fill_regenerates_on_host(hvs):
    n, d = 5000, 1536
    idx = hvs.HvsIndex(d, "f32")
    idx.fill_synthetic(n, seed=1234, first_global_row=1_000_000, n_groups=64)
    pos = np.array([0, 1, 77, 4999])
    np.testing.assert_array_equal(idx.gather(pos), po.synth_rows(1234, 1_000_000, n, d)[pos])
    rows = po.synth_rows(1234, 1_000_000, n, d)
    q = po.synth_rows(4321, 0, 2, d)
    dist, ids, cnt = idx.search(q, k=10, algo="scan")
    for i in range(2):
        full = parity.full_distances(rows, q[i], "f32")
        parity.check_topk(dist[i], ids[i] - 1_000_000, cnt[i], full, 10, label="synthetic")
    # group code = global row number mod 64
    dist, ids, cnt = idx.search(q[:1], k=10, group_filter=3, algo="scan")
    assert np.all(ids[0][: cnt[0]] % 64 == 3)
    idx.close()


def test_c_oracle_and_gpu_agree_at_medium_size(hvs):
    rng = np.random.default_rng(8)
    n, d = 200_000, 256
    rows = rng.standard_normal((n, d)).astype(np.float32)
    q = rng.standard_normal((1, d)).astype(np.float32)
    idx = _mk(hvs, rows, "f32")
    dist, ids, cnt = idx.search(q, k=10, max_distance=0.8, algo="scan")
    ci, cd = pgv_scan.scan_topk(rows, q, 10, threshold=0.8)
    assert cnt[0] == len(ci)
    np.testing.assert_allclose(dist[0][: cnt[0]], cd, atol=1e-5)
    assert ids[0][: cnt[0]].tolist() == ci.tolist()
    idx.close()


@pytest.mark.parametrize("dtype", ["f16", "bf16", "f32"])
def test_tiny_magnitudes_are_not_flushed(hvs, dtype):
    """fp16 subnormals (|x| < 6.1e-5) and very small bf16 / fp32 values must take part in the dot
    product exactly as in the oracle: cosine distance is scale-free, so a table of tiny vectors has
    the same neighbours as its scaled-up copy."""
    rng = np.random.default_rng(21)
    n, d = 2000, 192
    scale = {"f16": 2e-5, "bf16": 1e-15, "f32": 1e-15}[dtype]  # squares must stay inside the fp32 accumulator range
    rows = (rng.standard_normal((n, d)) * scale).astype(np.float32)
    q = rng.standard_normal((3, d)).astype(np.float32)
    idx = _mk(hvs, rows, dtype)
    for algo in ("scan", "mfma"):
        dist, ids, cnt = idx.search(q, k=10, fold=1, algo=algo)
        for i in range(3):
            full = parity.full_distances(rows, q[i], dtype)
            assert np.isfinite(full).sum() > n // 2
            parity.check_topk(dist[i], ids[i], cnt[i], full, 10, label=f"tiny {dtype} {algo} q={i}")
    dist, ids, cnt = idx.search(q, k=5, fold=3, algo="scan")
    parity.check_topk(dist[0], ids[0], cnt[0], parity.full_distances(rows, q, dtype), 5, label=f"tiny fold {dtype}")
    idx.close()


@pytest.mark.parametrize("dtype", ["f32", "f16", "bf16"])
def test_tombstones_compaction_and_snapshot(hvs, dtype, tmp_path):
    """Deleted rows stay out of results, compaction removes them without changing any answer, and a
    snapshot file brings back the same bits (rows, ids, group codes, tombstones)."""
    rng = np.random.default_rng(31)
    n, d = 5000, 72  # 72 elements: a padded row pitch for every storage type
    rows = rng.standard_normal((n, d)).astype(np.float32)
    ext_ids = rng.permutation(n).astype(np.int64) + 10_000
    groups = (np.arange(n) % 9).astype(np.int32)
    idx = hvs.HvsIndex(d, dtype)
    idx.add(rows, ids=ext_ids, group_codes=groups)
    gone = rng.choice(n, 700, replace=False)
    idx.set_deleted(gone, True)
    assert idx.deleted_count() == 700
    idx.set_deleted(gone[:50], True)  # deleting twice does not count twice
    assert idx.deleted_count() == 700
    idx.set_deleted(np.array([n - 1 if (n - 1) not in gone else n - 2]), False)  # restoring a live row changes nothing
    assert idx.deleted_count() == 700
    idx.set_deleted(gone[:100], False)
    assert idx.deleted_count() == 600
    dead = np.zeros(n, bool)
    dead[gone[100:]] = True
    q = rng.standard_normal((3, d)).astype(np.float32)

    def answers(ix):
        return [ix.search(q, k=20, algo="scan"), ix.search(q, k=20, group_filter=4, algo="scan"), ix.search(np.tile(q, (16, 1)), k=7, algo="mfma")]

    before = answers(idx)
    for i in range(3):
        parity.check_topk(before[0][0][i], before[0][1][i], before[0][2][i], parity.full_distances(rows, q[i], dtype), 20,
                          keep=~dead, id_of_row=ext_ids, label=f"tombstones {dtype}")
    # export: the stored bits, unpadded
    raw, ids, grp, dl = idx.export()
    stored = po.round_storage(rows, dtype)
    exp_raw = parity.bf16_bits(rows) if dtype == "bf16" else stored.astype(raw.dtype)
    np.testing.assert_array_equal(raw, exp_raw)
    np.testing.assert_array_equal(ids, ext_ids)
    np.testing.assert_array_equal(grp, groups)
    np.testing.assert_array_equal(dl, dead)
    part = idx.export(1234, 100)
    np.testing.assert_array_equal(part[0], exp_raw[1234:1334])
    np.testing.assert_array_equal(part[3], dead[1234:1334])

    path = str(tmp_path / f"t_{dtype}.hvs")
    idx.save(path)
    twin = hvs.HvsIndex.load(path)
    assert len(twin) == n and twin.deleted_count() == 600 and twin.dtype == dtype
    for a, b in zip(before, answers(twin)):
        for x, y in zip(a, b):
            np.testing.assert_array_equal(x, y)
    twin.close()

    assert idx.compact() == n - 600 and idx.deleted_count() == 0 and len(idx) == n - 600
    for a, b in zip(before, answers(idx)):
        for x, y in zip(a, b):
            np.testing.assert_array_equal(x, y)  # external ids moved with their rows
    raw2, ids2, grp2, dl2 = idx.export()
    np.testing.assert_array_equal(raw2, exp_raw[~dead])
    np.testing.assert_array_equal(ids2, ext_ids[~dead])
    np.testing.assert_array_equal(grp2, groups[~dead])
    assert not dl2.any()
    assert idx.compact() == n - 600  # nothing to do
    # rows can be appended after a compaction, and positions restart from the compacted size
    lo, hi = idx.add(rows[:10], ids=np.arange(10, dtype=np.int64), group_codes=groups[:10])
    assert (lo, hi) == (n - 600, n - 590)
    assert idx.compact(renumber_ids=True) == n - 590
    np.testing.assert_array_equal(idx.export()[1], np.arange(n - 590))
    idx.close()
    empty = hvs.HvsIndex(d, dtype)
    empty.save(path)
    e2 = hvs.HvsIndex.load(path)
    assert len(e2) == 0
    e2.close()
    empty.close()


@pytest.mark.parametrize("dtype", ["f16", "f32"])
def test_dimension_4096(hvs, dtype):
    """EMBEDDING_DIMENSION = 4096 (the reference's Qwen3 setting, .env.example:14): every path at that width,
    including four embeddings folded by LEAST (the query staging no longer fits eight at once)."""
    rng = np.random.default_rng(4096)
    n, d = 3000, 4096
    rows = rng.standard_normal((n, d)).astype(np.float32)
    q = rng.standard_normal((8, d)).astype(np.float32)
    idx = _mk(hvs, rows, dtype)
    dm = parity.distance_matrix(rows, q, dtype)
    for algo in ("scan", "mfma"):
        dist, ids, cnt = idx.search(q, k=10, algo=algo)
        for i in range(8):
            parity.check_topk(dist[i], ids[i], cnt[i], dm[i], 10, label=f"d4096 {dtype} {algo} q={i}")
        dist, ids, cnt = idx.search(q, k=10, fold=4, max_distance=1.1, algo=algo)
        for i in range(2):
            parity.check_topk(dist[i], ids[i], cnt[i], po.least_fold(dm[4 * i : 4 * i + 4]), 10, threshold=1.1,
                              label=f"d4096 fold4 {dtype} {algo} q={i}")
    col = idx.distances(q[:3])
    np.testing.assert_allclose(col, po.least_fold(dm[:3]), atol=1e-5)
    idx.close()


@pytest.mark.parametrize("dtype", ["f32", "f16", "bf16"])
def test_range_search_matches_the_distance_column(hvs, dtype):
    """Every row below a distance bound: the selection the consumer of the projected column makes
    (``search_sentence_cosine_distance < 0.4`` in the reference), against the oracle's column."""
    rng = np.random.default_rng(77)
    n, d = 20011, 96
    centre = rng.standard_normal(d).astype(np.float32)
    rows = rng.standard_normal((n, d)).astype(np.float32)
    close = rng.random(n) < 0.2
    rows[close] = centre + 0.6 * rows[close]  # a fifth of the table lies near the first query
    rows[17] = 0.0  # NaN distance: never selected
    groups = (np.arange(n) % 3).astype(np.int32)
    idx = _mk(hvs, rows, dtype, groups)
    idx.set_deleted(np.array([3, 4, 5, 19999]), True)
    live = np.ones(n, bool)
    live[[3, 4, 5, 19999]] = False
    q = np.stack([centre, rng.standard_normal(d).astype(np.float32), np.zeros(d, np.float32)])
    keep_bits = rng.random(n) < 0.5
    mask = np.packbits(keep_bits, bitorder="little")
    mask = np.concatenate([mask, np.zeros((-len(mask)) % 4, np.uint8)]).view(np.uint32)
    for fold in (1, 3):
        full = parity.full_distances(rows, q[:fold], dtype).astype(np.float64)
        for thr in (0.4, 0.25, 1.0, 1e-9, 3.0):
            for gf, use_mask, lo, hi in ((-1, False, 0, -1), (1, False, 0, -1), (-1, True, 0, -1), (2, True, 1234, 15000)):
                keep = live.copy()
                if gf >= 0:
                    keep &= groups == gf
                if use_mask:
                    keep &= keep_bits
                keep[:lo] = False
                if hi >= 0:
                    keep[hi:] = False
                dist, ids, total = idx.range_search(q[:fold], thr, group_filter=gf, row_mask=mask if use_mask else None,
                                                    row_begin=lo, row_end=hi)
                with np.errstate(invalid="ignore"):
                    sure = keep & (full < thr - 2e-6)
                    maybe = keep & (full < thr + 2e-6)
                got = np.zeros(n, bool)
                got[ids] = True
                label = f"{dtype} fold={fold} thr={thr} gf={gf} mask={use_mask}"
                assert total == len(ids) == got.sum(), label
                assert np.all(np.diff(ids) > 0), label  # ascending row position, no duplicates
                assert not np.any(sure & ~got) and not np.any(got & ~maybe), label
                np.testing.assert_allclose(dist, full[ids], atol=1e-5, err_msg=label)
                assert np.all(dist < thr) and 17 not in ids
    # capacity smaller than the result: a prefix, and the full count
    dist, ids, total = idx.range_search(q[:1], 0.4)
    assert total > 100
    d2, i2, t2 = idx.range_search(q[:1], 0.4, capacity=100)
    assert t2 == total and i2.tolist() == ids[:100].tolist() and d2.tolist() == dist[:100].tolist()
    # agreement with the top-k route when k covers every hit
    k = min(total, 512)
    dk, ik, ck = idx.search(q[:1], k=k, max_distance=0.4, algo="scan")
    if total <= 512:
        assert sorted(ik[0][: ck[0]].tolist()) == ids.tolist()
    else:
        assert set(ik[0].tolist()) <= set(ids.tolist())
    # nothing in range / empty range / empty index
    assert idx.range_search(q[2:3], 0.4)[2] == 0  # zero query: every distance NaN
    assert idx.range_search(q[:1], 0.4, row_begin=500, row_end=500)[2] == 0
    idx.close()
    empty = hvs.HvsIndex(d, dtype)
    assert empty.range_search(q[:1], 0.4)[2] == 0
    empty.close()


@pytest.mark.parametrize("dtype", ["f32", "f16", "bf16"])
def test_distance_column_of_many_embeddings(hvs, dtype):
    """The projected column of cross-language search takes one embedding per sentence of the query
    (``LEAST`` over all of them, pgvector.py:561-585): up to 128 in one call.  From five embeddings on the
    column comes from one pass on the matrix pipe (fp32 tables: through the split copy, built from 16 on);
    the other route, one streaming pass per eight embeddings, must give the same column."""
    rng = np.random.default_rng(909)
    for n, d in ((3001, 96), (70001, 200), (5000, 1536)):
        rows = rng.standard_normal((n, d)).astype(np.float32)
        rows[11] = 0.0  # NaN for every embedding: stays NaN
        rows[n - 1] *= 1e-3
        q = rng.standard_normal((128, d)).astype(np.float32)
        q[2] = 0.0  # a zero embedding takes no part in LEAST
        q[40] = rows[n - 2]  # distance ~0 to one row
        idx = _mk(hvs, rows, dtype)
        dm = parity.distance_matrix(rows, q, dtype)
        for m in (5, 8, 9, 16, 31, 33, 64, 65, 100, 128):
            full = po.least_fold(dm[:m])
            for lo, hi in ((0, -1), (257, n - 300)):
                want = full[lo : (n if hi < 0 else hi)]
                cols = {}
                for route in ("matrix", "stream"):
                    if route == "stream":
                        os.environ["HVS_NO_MFMA_DIST"] = "1"
                    try:
                        cols[route] = idx.distances(q[:m], row_begin=lo, row_end=hi)
                    finally:
                        os.environ.pop("HVS_NO_MFMA_DIST", None)
                    label = f"{dtype} n={n} d={d} m={m} [{lo},{hi}) {route}"
                    assert cols[route].shape == want.shape, label
                    assert np.array_equal(np.isnan(cols[route]), np.isnan(want)), label
                    np.testing.assert_allclose(cols[route], want, atol=1e-5, err_msg=label)
                np.testing.assert_allclose(cols["matrix"], cols["stream"], atol=3e-6)
        # the selection built on the column
        full = po.least_fold(dm[:50]).astype(np.float64)
        dist, ids, total = idx.range_search(q[:50], 0.9)
        with np.errstate(invalid="ignore"):
            sure, maybe = full < 0.9 - 2e-6, full < 0.9 + 2e-6
        got = np.zeros(n, bool)
        got[ids] = True
        assert total == len(ids) and not np.any(sure & ~got) and not np.any(got & ~maybe)
        np.testing.assert_allclose(dist, full[ids], atol=1e-5)
        with pytest.raises(hvs.HvsError):
            idx.distances(np.ones((129, d), np.float32))
        idx.close()


def test_prepare_builds_the_search_side_copy(hvs, monkeypatch):
    """``hvs_index_prepare``: an fp32 table gets its split copy ahead of the first search; from then on a single
    query on a large table takes the filter route (matrix-pipe launch, no scan launch) and returns the oracle's rows."""
    from hirag_prod_amd import _lib

    n, d = (1 << 20) + 77, 64
    monkeypatch.setenv("HVS_FILTER_MIN_ROWS", "1000000")  # (by default the route starts at 2 GB of fp32 rows)
    idx = hvs.HvsIndex(d, "f32")
    idx.fill_synthetic(n, seed=99)
    q = po.synth_rows(98, 0, 2, d)
    rows = po.synth_rows(99, 0, n, d)

    def launches():
        idx.profile(True)
        idx.profile_reset()
        out = idx.search(q[:1], k=10)
        c = idx.profile_get(_lib.HVS_PROF_SCAN)[1], idx.profile_get(_lib.HVS_PROF_MFMA_SEARCH)[1]
        idx.profile(False)
        return out, c

    (d0, i0, c0), (scans, mf) = launches()
    assert scans == 1 and mf == 0
    assert idx.prepare() is True
    (d1, i1, c1), (scans, mf) = launches()
    assert scans == 0 and mf == 1
    full = parity.full_distances(rows, q[0], "f32")
    parity.check_topk(d0[0], i0[0], c0[0], full, 10, label="scan")
    parity.check_topk(d1[0], i1[0], c1[0], full, 10, label="filter route")
    idx.close()
    half = hvs.HvsIndex(d, "f16")
    half.add(rows[:100])
    assert half.prepare() is False
    half.close()


@pytest.mark.parametrize("dtype", ["f16", "f32"])
def test_repeated_small_searches_follow_the_table(hvs, dtype):
    """The serving pattern: the same shape of search over and over on a small table, through reused workspaces and
    the pinned result buffer.  Every call must see its own query, and every change to the table (append, overwrite,
    delete, compaction) must reach the next answer."""
    rng = np.random.default_rng(12)
    n, d = 5000, 192
    rows = rng.standard_normal((n, d)).astype(np.float32)
    idx = _mk(hvs, rows, dtype)
    groups_live = np.ones(n, bool)

    def check(rows_now, keep, label, reps=5, **kw):
        for r in range(reps):
            q = rng.standard_normal((3, d)).astype(np.float32)
            dist, ids, cnt = idx.search(q, k=10, fold=3, max_distance=1.1, algo="scan", **kw)
            lo, hi = kw.get("row_begin", 0), kw.get("row_end", -1)
            kp = keep.copy()
            kp[:lo] = False
            if hi >= 0:
                kp[hi:] = False
            parity.check_topk(dist[0], ids[0], cnt[0], parity.full_distances(rows_now, q, dtype), 10, threshold=1.1, keep=kp,
                              label=f"{label} rep {r}")

    check(rows, groups_live, "fresh")
    check(rows, groups_live, "range", row_begin=100, row_end=4000)
    more = rng.standard_normal((700, d)).astype(np.float32)
    idx.add(more)
    rows2 = np.concatenate([rows, more])
    keep = np.ones(len(rows2), bool)
    check(rows2, keep, "appended")
    new = rng.standard_normal((5, d)).astype(np.float32)
    idx.update(np.arange(5), new)
    rows2[:5] = new
    q = np.concatenate([new[:1], new[1:3]])
    dist, ids, cnt = idx.search(q, k=10, fold=3, max_distance=1.1, algo="scan")
    assert set(ids[0][:3].tolist()) == {0, 1, 2} and dist[0][2] < 1e-3
    check(rows2, keep, "overwritten")
    idx.set_deleted(np.arange(0, len(rows2), 3), True)
    keep[::3] = False
    check(rows2, keep, "deleted")
    idx.compact(renumber_ids=True)
    check(rows2[keep], np.ones(int(keep.sum()), bool), "compacted")
    # single queries of other shapes in between, and timing switched on (the ordinary path) and off again
    for k in (1, 10, 50):
        q1 = rng.standard_normal((1, d)).astype(np.float32)
        for _ in range(3):
            dist, ids, cnt = idx.search(q1, k=k, algo="scan")
            parity.check_topk(dist[0], ids[0], cnt[0], parity.full_distances(rows2[keep], q1[0], dtype), k, label=f"k={k}")
    idx.profile(True)
    check(rows2[keep], np.ones(int(keep.sum()), bool), "timed", reps=2)
    idx.profile(False)
    check(rows2[keep], np.ones(int(keep.sum()), bool), "again", reps=3)
    idx.close()


@pytest.mark.parametrize("dtype", ["f32", "f16"])
def test_attribute_filters_are_evaluated_on_the_device(hvs, dtype):
    """hvs_search_filtered / hvs_range_search_filtered: workspace-only, knowledge-base-only, uri IN (...) and private =
    clauses over the attribute columns, alone and combined with the group code, a row range and tombstones, on the scan
    and the batched paths; the attributes move with their rows through a compaction."""
    rng = np.random.default_rng(2024)
    n, d = 50_003, 96
    rows = rng.standard_normal((n, d)).astype(np.float32)
    ws = rng.integers(0, 5, n).astype(np.int32)
    kb = rng.integers(0, 3, n).astype(np.int32)
    uri = rng.integers(0, 20_000, n).astype(np.int32)
    priv = rng.random(n) < 0.3
    groups = (ws * 3 + kb).astype(np.int32)
    idx = hvs.HvsIndex(d, dtype)
    idx.add(rows[:30_000], group_codes=groups[:30_000])
    idx.set_attributes(0, workspace=ws[:30_000], knowledge_base=kb[:30_000], uri=uri[:30_000], private=priv[:30_000])
    idx.add(rows[30_000:], group_codes=groups[30_000:])  # a second batch: the attribute column grows with the table
    idx.set_attributes(30_000, workspace=ws[30_000:], knowledge_base=kb[30_000:], uri=uri[30_000:], private=priv[30_000:])
    dead = rng.choice(n, 500, replace=False)
    idx.set_deleted(dead)
    alive = np.ones(n, bool)
    alive[dead] = False
    q = rng.standard_normal((40, d)).astype(np.float32)
    big_uri_list = rng.choice(20_000, 6_000, replace=False)
    cases = [
        ({"workspace": 2}, ws == 2),
        ({"knowledge_base": 1}, kb == 1),
        ({"access": True}, priv),
        ({"access": False, "workspace": 0}, ~priv & (ws == 0)),
        ({"uris": [7, 19_999, 123, 123]}, np.isin(uri, [7, 19_999, 123])),
        ({"uris": big_uri_list, "knowledge_base": 2}, np.isin(uri, big_uri_list) & (kb == 2)),
        ({"group": 7, "access": True}, (groups == 7) & priv),
        ({"group": 4}, groups == 4),
        ({"uris": []}, np.zeros(n, bool)),
    ]
    for filt, keep in cases:
        keep = keep & alive
        for algo, nq, k in (("scan", 3, 10), ("mfma", 40, 10), ("auto", 1, 100)):
            dist, ids, cnt = idx.search(q[:nq], k=k, filter=filt, max_distance=1.2, algo=algo)
            for i in range(nq if algo != "mfma" else 6):
                parity.check_topk(dist[i], ids[i], cnt[i], parity.full_distances(rows, q[i], dtype), k, threshold=1.2, keep=keep,
                                  label=f"{dtype} {sorted(filt)} {algo} q={i}")
        rkeep = keep.copy()
        rkeep[:1000] = False
        rkeep[45_000:] = False
        dist, ids, cnt = idx.search(q[:2], k=10, filter=filt, row_begin=1000, row_end=45_000, algo="scan")
        parity.check_topk(dist[0], ids[0], cnt[0], parity.full_distances(rows, q[0], dtype), 10, keep=rkeep, label=f"{dtype} {sorted(filt)} range")
        rd, ri, total = idx.range_search(q[:3], 0.9, filter=filt)
        full = parity.full_distances(rows, q[:3], dtype)
        with np.errstate(invalid="ignore"):
            exp = np.flatnonzero(keep & (full < 0.9))
        near = int((keep & (np.abs(full - 0.9) <= 1e-6)).sum())
        assert abs(total - len(exp)) <= near and (near or ri.tolist() == exp.tolist())
    # torch queries on the device take the same route
    import torch

    dt, it, ct = idx.search(torch.from_numpy(q[:3]).cuda(), k=10, filter={"workspace": 2}, max_distance=1.2, algo="scan")
    dn, in_, cn = idx.search(q[:3], k=10, filter={"workspace": 2}, max_distance=1.2, algo="scan")
    assert np.array_equal(it.cpu().numpy(), in_) and np.array_equal(dt.cpu().numpy(), dn)
    # compaction: attributes follow their rows
    assert idx.compact(renumber_ids=True) == n - 500
    live_rows = rows[alive]
    for filt, keep in cases[:6]:
        dist, ids, cnt = idx.search(q[:2], k=10, filter=filt, algo="scan")
        for i in range(2):
            parity.check_topk(dist[i], ids[i], cnt[i], parity.full_distances(live_rows, q[i], dtype), 10, keep=keep[alive],
                              label=f"{dtype} {sorted(filt)} after compaction")
    with pytest.raises(hvs.HvsError):
        idx.search(q[:1], k=5, filter={"uris": np.arange(9000)})  # more codes than the device's list holds
    with pytest.raises(ValueError):
        idx.search(q[:1], k=5, filter={"workspace": 1}, group_filter=2)
    idx.close()


def test_device_positions_are_range_checked(hvs):
    """hvs_gather / hvs_index_update with positions in device memory: a position outside the table is refused before any
    kernel uses it."""
    import ctypes

    import torch

    from hirag_prod_amd import _lib

    lib = hvs.load_library()
    idx = hvs.HvsIndex(16, "f32")
    rows = np.random.default_rng(3).standard_normal((100, 16)).astype(np.float32)
    idx.add(rows)
    out = torch.zeros((3, 16), device="cuda")
    good = torch.tensor([5, 99, 0], dtype=torch.int64, device="cuda")
    torch.cuda.synchronize()
    _lib.check(lib.hvs_gather(idx._h, ctypes.c_void_p(good.data_ptr()), 3, ctypes.c_void_p(out.data_ptr()), 1))
    idx.sync()
    assert np.array_equal(out.cpu().numpy(), rows[[5, 99, 0]])
    for bad_value in (100, -1, 1 << 40):
        bad = torch.tensor([5, bad_value, 0], dtype=torch.int64, device="cuda")
        torch.cuda.synchronize()
        assert lib.hvs_gather(idx._h, ctypes.c_void_p(bad.data_ptr()), 3, ctypes.c_void_p(out.data_ptr()), 1) != 0
        assert b"outside" in lib.hvs_last_error()
        new = torch.ones((3, 16), device="cuda")
        torch.cuda.synchronize()
        assert lib.hvs_index_update(idx._h, ctypes.c_void_p(bad.data_ptr()), ctypes.c_void_p(new.data_ptr()), _lib.HVS_F32, 3, 1) != 0
    assert np.array_equal(idx.gather(np.arange(100)), rows)  # nothing was written
    idx.close()
