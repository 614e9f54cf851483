"""GPU tier: batched searches whose blocks see only a few corpus tiles each -- the launches in which the blocks pool
the minima of their first tile for a common starting bound and append their survivors to one list per query -- against
the oracle and against the same launches with that switched off (HVS_NO_XCHG=1)."""

import math
import os

import numpy as np
import pytest

from tests import parity

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def hvs():
    import hirag_prod_amd

    assert hirag_prod_amd.device_count() >= 1, "no GPU visible to libhvs"
    return hirag_prod_amd


def _both(idx, q, **kw):
    got = idx.search(q, algo="mfma", **kw)
    os.environ["HVS_NO_XCHG"] = "1"
    try:
        ref = idx.search(q, algo="mfma", **kw)
    finally:
        del os.environ["HVS_NO_XCHG"]
    return got, ref


@pytest.mark.parametrize("dtype", ["f16", "bf16", "f32"])
@pytest.mark.parametrize("nq,k", [(8, 10), (100, 100), (300, 40), (700, 10)])
def test_pooled_first_tiles_give_the_same_answers(hvs, dtype, nq, k):
    rng = np.random.default_rng(nq * 1000 + k)
    n, d = 40_000 + 37, 80  # a ragged last tile
    rows = rng.standard_normal((n, d)).astype(np.float32)
    rows[11] = 0.0  # NaN distance: never returned, and it must not poison a minimum
    rows[300:600] = rows[5]  # 300 exact ties
    rows[rng.choice(n, 50, replace=False)] *= 1e-3  # norms across three decades
    q = rng.standard_normal((nq, d)).astype(np.float32)
    q[1] = rows[5]  # its top-k is made of ties
    q[2] = 0.0  # a zero query: every distance NaN, count 0
    idx = hvs.HvsIndex(d, dtype)
    idx.add(rows, group_codes=(np.arange(n) % 3).astype(np.int32))
    idx.set_deleted(np.arange(100, 140))
    dead = np.zeros(n, bool)
    dead[100:140] = True
    # every row of the first tiles filtered away, so that some blocks have nothing to publish but +inf
    bits = np.ones(n, bool)
    bits[:5000] = False
    bits[rng.choice(n, 8000, replace=False)] = False
    mask = np.packbits(np.concatenate([bits, np.zeros((-n) % 32, bool)]), bitorder="little").view(np.uint32)
    cases = [({}, ~dead, None), ({"max_distance": 0.9}, ~dead, 0.9), ({"row_mask": mask}, bits & ~dead, None),
             ({"group_filter": 1}, (np.arange(n) % 3 == 1) & ~dead, None)]
    for kw, keep, thr in cases:
        (dist, ids, cnt), (r_dist, r_ids, r_cnt) = _both(idx, q, k=k, **kw)
        assert np.array_equal(cnt, r_cnt) and np.array_equal(ids, r_ids) and np.array_equal(dist, r_dist), f"{dtype} {kw}: differs from HVS_NO_XCHG=1"
        assert int(cnt[2]) == 0
        for i in list(range(min(nq, 6))) + [nq - 1]:
            full = parity.full_distances(rows, q[i], dtype)
            parity.check_topk(dist[i], ids[i], cnt[i], full, k, threshold=thr, keep=keep, label=f"{dtype} nq={nq} k={k} {kw} q={i}")
    # folded embeddings (LEAST over 2 and over 3, padded to 4)
    for fold in (2, 3):
        m = nq // fold * fold
        if m == 0:
            continue
        (dist, ids, cnt), (r_dist, r_ids, r_cnt) = _both(idx, q[:m], k=k, fold=fold, max_distance=0.95)
        assert np.array_equal(ids, r_ids) and np.array_equal(dist, r_dist) and np.array_equal(cnt, r_cnt)
        for i in range(min(m // fold, 4)):
            full = parity.full_distances(rows, q[i * fold : (i + 1) * fold], dtype)
            parity.check_topk(dist[i], ids[i], cnt[i], full, k, threshold=0.95, keep=~dead, label=f"{dtype} fold={fold} q={i}")
    idx.close()


def test_long_pooled_lists_are_ranked_in_rounds(hvs):
    """k = 256 with many splits: a query's pooled list passes the 1024 keys one round ranks"""
    rng = np.random.default_rng(3)
    n, d = 60_000, 64
    rows = rng.standard_normal((n, d)).astype(np.float32)
    q = rng.standard_normal((20, d)).astype(np.float32)
    idx = hvs.HvsIndex(d, "f16")
    idx.add(rows)
    for k in (200, 256):
        (dist, ids, cnt), (r_dist, r_ids, r_cnt) = _both(idx, q, k=k)
        assert np.array_equal(ids, r_ids) and np.array_equal(dist, r_dist) and np.array_equal(cnt, r_cnt)
        for i in range(4):
            parity.check_topk(dist[i], ids[i], cnt[i], parity.full_distances(rows, q[i], "f16"), k, label=f"k={k} q={i}")
    idx.close()


@pytest.mark.parametrize("dtype", ["f32", "f16"])
def test_blocks_in_lockstep_or_free_running_give_the_same_answers(hvs, dtype):
    """the arrival counters that keep the blocks of a corpus stream on the same tile only change timing"""
    rng = np.random.default_rng(12)
    n, d, nq, k = 70_000, 96, 600, 10  # three query tiles of 256 (two of 128 for fp32 rows): several blocks per stream
    rows = rng.standard_normal((n, d)).astype(np.float32)
    q = rng.standard_normal((nq, d)).astype(np.float32)
    idx = hvs.HvsIndex(d, dtype)
    idx.add(rows)
    algo = "mfma_f32" if dtype == "f32" else "mfma"
    locked = idx.search(q, k=k, algo=algo)
    os.environ["HVS_NO_LOCKSTEP"] = "1"
    try:
        free = idx.search(q, k=k, algo=algo)
    finally:
        del os.environ["HVS_NO_LOCKSTEP"]
    for a, b in zip(locked, free):
        assert np.array_equal(a, b)
    for i in (0, 255, 256, nq - 1):
        parity.check_topk(locked[0][i], locked[1][i], locked[2][i], parity.full_distances(rows, q[i], dtype), k, label=f"{dtype} lockstep q={i}")
    idx.close()


@pytest.mark.parametrize("dtype", ["f16", "bf16"])
def test_pooling_with_between_one_and_a_half_and_two_k_values(hvs, dtype):
    """16 query tiles of 256 leave 16 splits of the rows: 64 published values per query, with k = 40 more than 1.5 k and fewer
    than 2 k.  The launch pools its first tiles (one search launch, no pass over a sample) and gives the answers of the launch
    that does not; a sample of the queries is held to the oracle."""
    rng = np.random.default_rng(4040)
    n, d, nq, k = 60_000 + 11, 64, 3900, 40
    rows = rng.standard_normal((n, d)).astype(np.float32)
    rows[7] = 0.0
    rows[1000:1100] = rows[3]
    q = rng.standard_normal((nq, d)).astype(np.float32)
    q[1] = rows[3]
    idx = hvs.HvsIndex(d, dtype)
    idx.add(rows)
    from hirag_prod_amd import _lib

    idx.profile(True)
    idx.profile_reset()
    dist, ids, cnt = idx.search(q, k=k, algo="mfma")
    _, launches = idx.profile_get(_lib.HVS_PROF_MFMA_SEARCH)
    idx.profile(False)
    assert launches == 1, f"{launches} search launches: the blocks did not pool their first tiles"
    os.environ["HVS_NO_XCHG"] = "1"
    try:
        r_dist, r_ids, r_cnt = idx.search(q, k=k, algo="mfma")
    finally:
        del os.environ["HVS_NO_XCHG"]
    assert np.array_equal(cnt, r_cnt) and np.array_equal(ids, r_ids) and np.array_equal(dist, r_dist)
    sample = np.concatenate([[0, 1, 2], rng.choice(nq, 21, replace=False)])
    dm = parity.distance_matrix(rows, q[sample], dtype)
    for j, i in enumerate(sample):
        parity.check_topk(dist[i], ids[i], cnt[i], dm[j], k, label=f"{dtype} query {i}")
