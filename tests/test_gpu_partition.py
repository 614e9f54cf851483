"""GPU tier: the group-partitioned table order (hvs_index_partition) -- the (workspaceId, knowledgeBaseId) filter of
every PGVector.query (pgvector.py:577-580) as a row range -- against the oracle and against the unpartitioned table."""

import math
import os

import numpy as np
import pytest

from oracle import kmeans_oracle
from tests import parity

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def hvs():
    import hirag_prod_amd

    assert hirag_prod_amd.device_count() >= 1, "no GPU visible to libhvs"
    return hirag_prod_amd


def _check(idx, rows, groups, dead, q, k, g, dtype, algo, id_of_row, label, fold=1, thr=None, mask=None):
    """one filtered search against the float64 oracle; rows/groups/dead are in INSERTION order, ids name insertion numbers"""
    keep = (groups == g) & ~dead
    kw = {}
    if mask is not None:
        keep &= mask[0]
        kw["row_mask"] = mask[1]
    dist, ids, cnt = idx.search(q, k=k, fold=fold, group_filter=g, algo=algo, max_distance=math.inf if thr is None else thr, **kw)
    for i in range(q.shape[0] // fold):
        full = parity.full_distances(rows, q[i * fold : (i + 1) * fold], dtype)
        parity.check_topk(dist[i], ids[i], cnt[i], full, k, threshold=thr, keep=keep, id_of_row=id_of_row, label=f"{label} q={i}")
    return dist, ids, cnt


@pytest.mark.parametrize("dtype", ["f32", "f16", "bf16"])
def test_partitioned_search_matches_oracle_and_the_unpartitioned_table(hvs, dtype):
    rng = np.random.default_rng(5)
    n, d = 9000, 96
    rows = rng.standard_normal((n, d)).astype(np.float32)
    groups = rng.integers(0, 7, n).astype(np.int32)
    groups[groups == 4] = 5000  # a code beyond one digit of the sort, and a gap in the codes
    rows[17] = rows[3]
    groups[17] = groups[3]  # an exact tie inside one group
    q = rng.standard_normal((40, d)).astype(np.float32)
    idx = hvs.HvsIndex(d, dtype)
    idx.add(rows, group_codes=groups)
    dead = np.zeros(n, bool)
    dead[rng.choice(n, 300, replace=False)] = True
    idx.set_deleted(np.flatnonzero(dead))
    before = {}
    for algo, nq in (("scan", 3), ("mfma", 40), ("auto", 1)):
        for g in (0, 3, 5000):
            before[(algo, g)] = idx.search(q[:nq], k=10, group_filter=g, algo=algo)
    assert idx.partition_info() == (0, 0)

    old = idx.partition()
    live_old = np.flatnonzero(~dead)
    assert len(idx) == len(live_old) and len(old) == len(idx) and idx.deleted_count() == 0
    # stable: by group code, insertion order inside a group; tombstoned rows are gone
    order = live_old[np.argsort(groups[live_old], kind="stable")]
    assert np.array_equal(old, order)
    assert np.array_equal(idx.ids(), order)  # ids travel with their rows
    rows_now, ids_now, groups_now, deleted_now = idx.export()
    assert np.array_equal(groups_now, groups[order]) and not deleted_now.any()
    ordered, n_groups = idx.partition_info()
    assert ordered == len(idx) and n_groups == len(np.unique(groups[live_old]))
    stored = np.asarray(idx.gather(np.arange(len(idx))))
    ref = hvs.HvsIndex(d, dtype)
    ref.add(rows[order])
    assert np.array_equal(stored, np.asarray(ref.gather(np.arange(len(ref)))))
    ref.close()

    id_of_row = np.arange(n)  # ids are insertion numbers
    for algo, nq in (("scan", 3), ("mfma", 40), ("auto", 1)):
        for g in (0, 3, 5000, 4, 77):  # 4 and 77: codes no row carries
            dist, ids, cnt = _check(idx, rows, groups, dead, q[:nq], 10, g, dtype, algo, id_of_row, f"{dtype} {algo} g={g}")
            if (algo, g) in before:
                b_dist, b_ids, b_cnt = before[(algo, g)]
                assert np.array_equal(cnt, b_cnt) and np.array_equal(ids, b_ids), f"{algo} g={g}: ids moved with the partition"
                assert np.array_equal(dist, b_dist), f"{algo} g={g}: a row's distance depends on where the row lies"
            else:
                assert int(cnt.max()) == 0
    # folded embeddings, a threshold, a row bitmask over the new positions, k above the group's size
    _check(idx, rows, groups, dead, q[:6], 10, 2, dtype, "scan", id_of_row, "fold3", fold=3, thr=0.95)
    _check(idx, rows, groups, dead, q[:8], 10, 2, dtype, "mfma", id_of_row, "fold2 mfma", fold=2, thr=0.97)
    bits_new = rng.random(len(idx)) < 0.4
    bits_by_insertion = np.zeros(n, bool)
    bits_by_insertion[order] = bits_new
    mask = np.packbits(np.concatenate([bits_new, np.zeros((-len(idx)) % 32, bool)]), bitorder="little").view(np.uint32)
    for algo, nq in (("scan", 2), ("mfma", 24)):
        _check(idx, rows, groups, dead, q[:nq], 10, 1, dtype, algo, id_of_row, f"mask {algo}", mask=(bits_by_insertion, mask))
    size_g6 = int(((groups == 6) & ~dead).sum())
    dist, ids, cnt = _check(idx, rows, groups, dead, q[:2], 1024, 6, dtype, "scan", id_of_row, "k > group")
    assert int(cnt[0]) == min(1024, size_g6)
    # a second call finds the table in order: nothing moves
    again = idx.partition()
    assert np.array_equal(again, np.arange(len(idx))) and np.array_equal(idx.ids(), order)
    idx.close()


def test_append_query_partition_delete_query(hvs):
    """appends land in a tail that is searched with the per-row test and merged with the group's range"""
    rng = np.random.default_rng(9)
    d, dtype = 64, "f16"
    rows = rng.standard_normal((6000, d)).astype(np.float32)
    groups = (np.arange(6000) % 4).astype(np.int32)
    q = rng.standard_normal((20, d)).astype(np.float32)
    idx = hvs.HvsIndex(d, dtype)
    idx.add(rows[:4000], group_codes=groups[:4000])
    dead = np.zeros(6000, bool)
    known = 4000

    def check_all(label):
        for algo, nq, k in (("scan", 2, 10), ("mfma", 20, 10), ("auto", 6, 100)):
            for g in (0, 3, 9):
                _check(idx, rows[:known], groups[:known], dead[:known], q[:nq], k, g, dtype, algo, np.arange(known), f"{label} {algo} g={g}")

    check_all("unpartitioned")
    idx.partition()
    check_all("partitioned")
    # append: rows 4000 .. 5000 arrive behind the ordered part, in any group; one of them duplicates an ordered row
    rows[4100] = rows[8]
    groups[4100] = groups[8]
    idx.add(rows[4000:5000], ids=np.arange(4000, 5000), group_codes=groups[4000:5000])
    known = 5000
    assert idx.partition_info()[0] == 4000
    check_all("range + tail")
    dist, ids, cnt = idx.search(rows[8:9], k=2, group_filter=int(groups[8]), algo="scan")
    assert ids[0].tolist() == [8, 4100], "equal distances: the ordered part lies below the tail"
    # a group that exists only in the tail
    groups[5000:5100] = 11
    idx.add(rows[5000:5100], ids=np.arange(5000, 5100), group_codes=groups[5000:5100])
    known = 5100
    _check(idx, rows[:known], groups[:known], dead[:known], q[:3], 10, 11, dtype, "scan", np.arange(known), "tail-only group")
    # deletes in both parts stay tombstones until the next partition / compaction
    pos_of_id = {int(v): p for p, v in enumerate(idx.ids())}
    gone = rng.choice(known, 400, replace=False)
    dead[gone] = True
    idx.set_deleted(np.asarray([pos_of_id[int(v)] for v in gone]))
    check_all("tombstones")
    # range search: ascending position = the group's range, then the tail
    for g in (1, 11):
        rd, ri, total = idx.range_search(q[:3], 0.9, group_filter=g)
        full = parity.full_distances(rows[:known], q[:3], dtype)
        expect = {int(i) for i in np.flatnonzero((groups[:known] == g) & ~dead[:known] & (full < 0.9 - 1e-6))}
        maybe = {int(i) for i in np.flatnonzero((groups[:known] == g) & ~dead[:known] & (full < 0.9 + 1e-6))}
        assert expect <= set(ri.tolist()) <= maybe and total == len(ri)
        pos = [pos_of_id[int(v)] for v in ri]
        assert pos == sorted(pos)
    # compaction keeps the order: the ordered part shrinks to its live rows, the ranges follow
    n_live_part = int((~dead[idx.ids()[:4000]]).sum())
    idx.compact()
    assert idx.partition_info()[0] == n_live_part
    check_all("compacted")
    # the allocation-free search takes the range too once no tail is left
    idx.partition()
    assert idx.partition_info()[0] == len(idx)
    check_all("partitioned again")
    idx.close()


def test_partition_needs_nonnegative_codes_and_survives_an_empty_table(hvs):
    idx = hvs.HvsIndex(16, "f32")
    assert len(idx.partition()) == 0
    rows = np.random.default_rng(0).standard_normal((100, 16)).astype(np.float32)
    idx.add(rows, group_codes=np.full(100, -3, np.int32))
    with pytest.raises(hvs.HvsError, match="codes >= 0"):
        idx.partition()
    assert len(idx) == 100  # left as it was
    idx.set_deleted(np.arange(100))
    assert len(idx.partition()) == 0 and len(idx) == 0  # every row deleted: nothing left, nothing to order
    idx.close()


def test_counting_sort_behind_the_kmeans_sums_takes_more_than_one_digit(hvs):
    """4096 keys fit one 12-bit digit of the sort; 5001 (5000 centroids + the key of deleted rows) need two passes"""
    rng = np.random.default_rng(21)
    n, d, K = 30000, 32, 5000
    rows = rng.standard_normal((n, d)).astype(np.float32)
    idx = hvs.HvsIndex(d, "f32")
    idx.add(rows)
    idx.set_deleted(np.asarray([5, 77, 20000]))
    c = rows[rng.choice(n, K, replace=False)] + 0.01 * rng.standard_normal((K, d)).astype(np.float32)
    assign, sums, counts = idx.kmeans_assign(c)
    assign2, sums2, counts2 = idx.kmeans_assign(c)
    assert np.array_equal(sums, sums2) and np.array_equal(counts, counts2)  # same bits on every run
    exp, margin = kmeans_oracle.assign(rows, c)
    live = np.ones(n, bool)
    live[[5, 77, 20000]] = False
    ok = live & (margin > 1e-3)
    assert np.array_equal(assign[ok], exp[ok]) and np.all(assign[~live] == -1)
    exp_counts = np.bincount(assign[live], minlength=K)
    assert np.array_equal(counts, exp_counts)
    exp_sums = np.zeros((K, d))
    np.add.at(exp_sums, assign[live], rows[live].astype(np.float64))
    assert np.abs(sums - exp_sums).max() <= 1e-4
    idx.close()


def test_a_table_loaded_in_group_order_is_only_indexed(hvs):
    """rows that arrive ordered by code (a bulk load by workspace): hvs_index_partition moves nothing and builds the ranges"""
    n, d, runs = 10_007, 32, 7
    idx = hvs.HvsIndex(d, "f16")
    idx.fill_synthetic(n, seed=3, n_groups=-runs)
    rows, ids, groups, deleted = idx.export()
    assert np.array_equal(groups, (np.arange(n, dtype=np.int64) * runs // n).astype(np.int32))
    old = idx.partition()
    assert np.array_equal(old, np.arange(n)) and idx.partition_info() == (n, runs)
    assert np.array_equal(idx.ids(), ids)
    q = np.random.default_rng(2).standard_normal((5, d)).astype(np.float32)
    for algo, nq in (("scan", 1), ("mfma", 5)):
        _check(idx, rows, groups, np.zeros(n, bool), q[:nq], 10, 4, "f16", algo, ids, f"in-order {algo}")
    idx.close()


def test_range_and_tail_in_one_scan_launch_and_a_dense_filtered_range(hvs):
    """a single query on a table with an ordered part and a tail is one scan launch (range without the code test, tail
    with it); a filtered scan over a short range whose rows all pass shares the range among the waves"""
    from hirag_prod_amd import _lib

    rng = np.random.default_rng(31)
    n, d, dtype = 60_000, 64, "f16"
    rows = rng.standard_normal((n, d)).astype(np.float32)
    groups = (np.arange(n) % 5).astype(np.int32)
    idx = hvs.HvsIndex(d, dtype)
    idx.add(rows[:45_000], group_codes=groups[:45_000])
    idx.partition()
    idx.add(rows[45_000:], ids=np.arange(45_000, n), group_codes=groups[45_000:])
    dead = np.zeros(n, bool)
    q = rng.standard_normal((6, d)).astype(np.float32)
    for fold, nq in ((1, 1), (3, 1), (1, 4)):
        idx.profile(True)
        idx.profile_reset()
        _check(idx, rows, groups, dead, q[: nq * fold], 10, 2, dtype, "auto", np.arange(n), f"one launch fold={fold}", fold=fold)
        launches = idx.profile_get(_lib.HVS_PROF_SCAN)[1]
        idx.profile(False)
        assert launches == 1, f"range + tail took {launches} scan launches"
    # tombstones in both parts: the range is walked in spans too
    pos_of_id = {int(v): p for p, v in enumerate(idx.ids())}
    gone = rng.choice(n, 900, replace=False)
    dead[gone] = True
    idx.set_deleted(np.asarray([pos_of_id[int(v)] for v in gone]))
    for g in (2, 4):
        _check(idx, rows, groups, dead, q[:3], 10, g, dtype, "scan", np.arange(n), f"tombstones g={g}")
    _check(idx, rows, groups, dead, q[:3], 10, 3, dtype, "scan", np.arange(n), "tombstones fold", fold=3)
    # no filter at all on a tombstoned table: every row of a short range is live but for a few
    dist, ids, cnt = idx.search(q[:2], k=10, algo="scan")
    for i in range(2):
        parity.check_topk(dist[i], ids[i], cnt[i], parity.full_distances(rows, q[i], dtype), 10, keep=~dead, label=f"dense q={i}")
    idx.close()
