"""GPU tier, collected last: ``hvs_search_noalloc`` (the search that allocates nothing) against ``hvs_search``.

The entry point has no counterpart in the reference; it exists for callers that cannot call ``hipMalloc``.  It must
return what ``hvs_search(..., on_device=1, algo=SCAN)`` returns, bit for bit, on consecutive calls that share the one
scratch array of the GPU -- the second call here is smaller than the first, so stale contents of the first would show.
"""

import ctypes
import math

import numpy as np
import pytest
import torch

from tests import parity

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def hvs():
    import hirag_prod_amd

    assert hirag_prod_amd.device_count() >= 1, "no GPU visible to libhvs"
    return hirag_prod_amd


def _ptr(t):
    return None if t is None else ctypes.c_void_p(t.data_ptr())


def _both(hvs, idx, q, nq, fold, k, group=-1, mask=None, thr=math.inf, row_begin=0, row_end=-1):
    """-> ((dist, ids, cnt) of hvs_search on the scan path, the same of hvs_search_noalloc), as numpy arrays"""
    from hirag_prod_amd import _lib

    lib = hvs.load_library()
    dev = q.device
    outs = []
    for which in ("search", "noalloc"):
        dist = torch.full((nq, k), -7.0, dtype=torch.float32, device=dev)
        ids = torch.full((nq, k), -7, dtype=torch.int64, device=dev)
        cnt = torch.full((nq,), -7, dtype=torch.int32, device=dev)
        torch.cuda.synchronize()
        if which == "search":
            rc = lib.hvs_search(idx._h, _ptr(q), nq, fold, k, group, _ptr(mask), float(thr), row_begin, row_end, _ptr(dist),
                                _ptr(ids), _ptr(cnt), 1, _lib.HVS_ALGO_SCAN)
        else:
            free0, _ = torch.cuda.mem_get_info(dev)
            rc = lib.hvs_search_noalloc(idx._h, _ptr(q), nq, fold, k, group, _ptr(mask), float(thr), row_begin, row_end,
                                        _ptr(dist), _ptr(ids), _ptr(cnt))
            free1, _ = torch.cuda.mem_get_info(dev)
            assert free1 == free0, "device memory was taken or released during hvs_search_noalloc"
        _lib.check(rc)
        idx.sync()
        outs.append((dist.cpu().numpy(), ids.cpu().numpy(), cnt.cpu().numpy()))
    return outs


def _same(a, b, label):
    for x, y, name in zip(a, b, ("dist", "ids", "count")):
        assert x.dtype == y.dtype and x.shape == y.shape
        assert np.array_equal(x.view(np.uint8), y.view(np.uint8)), f"{label}: {name} differ"


@pytest.mark.parametrize("dtype", ["f32", "f16", "bf16"])
def test_two_calls_in_a_row_equal_hvs_search(hvs, dtype):
    rng = np.random.default_rng(77)
    n, d = 300_000, 128  # enough rows for a full grid: 4096 wave lists
    rows = rng.standard_normal((n, d)).astype(np.float32)
    rows[17] = 0.0
    rows[23] = rows[19]
    groups = (np.arange(n) % 7).astype(np.int32)
    idx = hvs.HvsIndex(d, dtype)
    idx.add(rows, group_codes=groups)
    q_host = rng.standard_normal((12, d)).astype(np.float32)
    q = torch.from_numpy(q_host).cuda()
    keep = rng.random(n) < 0.4
    mask = torch.from_numpy(np.packbits(keep, bitorder="little").view(np.uint32).copy() if n % 32 == 0 else
                            np.packbits(np.concatenate([keep, np.zeros(32 - n % 32, bool)]), bitorder="little").view(np.uint32).copy()).cuda()

    # first call: 12 queries, k = 100; second: 2 queries of 3 folded embeddings, k = 7, a filter and a threshold
    ref, got = _both(hvs, idx, q, 12, 1, 100)
    _same(ref, got, f"{dtype} first call")
    for i in (0, 11):
        parity.check_topk(got[0][i], got[1][i], got[2][i], parity.full_distances(rows, q_host[i], dtype), 100, label=f"{dtype} q={i}")
    ref, got = _both(hvs, idx, q[:6].contiguous(), 2, 3, 7, group=3, mask=mask, thr=0.93, row_begin=1000, row_end=250_000)
    _same(ref, got, f"{dtype} second call")
    ok = keep & (groups == 3)
    ok[:1000] = False
    ok[250_000:] = False
    for i in range(2):
        parity.check_topk(got[0][i], got[1][i], got[2][i], parity.full_distances(rows, q_host[3 * i:3 * i + 3], dtype), 7,
                          threshold=0.93, keep=ok, label=f"{dtype} filtered q={i}")
    # third: smaller again, one query, k = 1
    ref, got = _both(hvs, idx, q[5:6].contiguous(), 1, 1, 1)
    _same(ref, got, f"{dtype} third call")
    idx.close()


def test_limits_are_refused_with_a_message(hvs):
    from hirag_prod_amd import _lib

    lib = hvs.load_library()
    rng = np.random.default_rng(5)
    n, d = 300_000, 64
    idx = hvs.HvsIndex(d, "f32")
    idx.add(rng.standard_normal((n, d)).astype(np.float32))
    q = torch.from_numpy(rng.standard_normal((4, d)).astype(np.float32)).cuda()
    k = 900  # one query: 1024 blocks are resident, and 1024 block lists of 900 keys do not fit 7 MiB
    dist = torch.empty((1, k), dtype=torch.float32, device="cuda")
    ids = torch.empty((1, k), dtype=torch.int64, device="cuda")
    cnt = torch.empty((1,), dtype=torch.int32, device="cuda")
    rc = lib.hvs_search_noalloc(idx._h, _ptr(q), 1, 1, k, -1, None, math.inf, 0, -1, _ptr(dist), _ptr(ids), _ptr(cnt))
    assert rc != 0 and b"scratch" in lib.hvs_last_error()
    # a small table runs fewer blocks, so the same k fits
    small = hvs.HvsIndex(d, "f32")
    small.add(rng.standard_normal((3000, d)).astype(np.float32))
    ref, got = _both(hvs, small, q, 4, 1, 1024)
    _same(ref, got, "k = 1024 on a small table")
    # too many query embeddings for the array's query region
    many = torch.zeros((8192, d), dtype=torch.float32, device="cuda")
    dist = torch.empty((8192, 1), dtype=torch.float32, device="cuda")
    ids = torch.empty((8192, 1), dtype=torch.int64, device="cuda")
    cnt = torch.empty((8192,), dtype=torch.int32, device="cuda")
    rc = lib.hvs_search_noalloc(small._h, _ptr(many), 8192, 1, 1, -1, None, math.inf, 0, -1, _ptr(dist), _ptr(ids), _ptr(cnt))
    assert rc != 0 and b"query embeddings" in lib.hvs_last_error()
    # and the call after a refusal works
    ref, got = _both(hvs, small, q, 4, 1, 5)
    _same(ref, got, "after a refusal")
    idx.close()
    small.close()


def test_limit_on_a_full_grid_is_the_documented_one(hvs):
    """1024 block lists (the resident blocks) and the 32 lists of their first merge level: k * 8 bytes * 1056 must fit 7 MiB, i.e. k <= 868"""
    lib = hvs.load_library()
    rng = np.random.default_rng(6)
    n, d = 300_000, 64
    rows = rng.standard_normal((n, d)).astype(np.float32)
    idx = hvs.HvsIndex(d, "f32")
    idx.add(rows)
    q_host = rng.standard_normal((2, d)).astype(np.float32)
    q = torch.from_numpy(q_host).cuda()
    one = q[1:2].contiguous()  # one query per pass keeps four blocks per CU resident: the full grid
    ref, got = _both(hvs, idx, one, 1, 1, 868)
    _same(ref, got, "k = 868")
    parity.check_topk(got[0][0], got[1][0], got[2][0], parity.full_distances(rows, q_host[1], "f32"), 868, label="k = 868")
    dist = torch.empty((1, 869), dtype=torch.float32, device="cuda")
    ids = torch.empty((1, 869), dtype=torch.int64, device="cuda")
    cnt = torch.empty((1,), dtype=torch.int32, device="cuda")
    rc = lib.hvs_search_noalloc(idx._h, _ptr(one), 1, 1, 869, -1, None, math.inf, 0, -1, _ptr(dist), _ptr(ids), _ptr(cnt))
    assert rc != 0 and b"scratch" in lib.hvs_last_error()
    idx.close()


def test_a_table_in_group_order_answers_from_the_row_range(hvs):
    rng = np.random.default_rng(8)
    n, d = 120_000, 64
    rows = rng.standard_normal((n, d)).astype(np.float32)
    groups = rng.integers(0, 9, n).astype(np.int32)
    idx = hvs.HvsIndex(d, "f16")
    idx.add(rows, group_codes=groups)
    q_host = rng.standard_normal((3, d)).astype(np.float32)
    q = torch.from_numpy(q_host).cuda()
    before = _both(hvs, idx, q, 3, 1, 20, group=4)[1]
    idx.partition()
    ref, got = _both(hvs, idx, q, 3, 1, 20, group=4)
    _same(ref, got, "group range")
    _same(before, got, "before and after the partition")
    for i in range(3):
        parity.check_topk(got[0][i], got[1][i], got[2][i], parity.full_distances(rows, q_host[i], "f16"), 20, keep=groups == 4,
                          label=f"group range q={i}")
    # rows behind the ordered part: one launch walks the group's range and then the tail, which keeps the per-row test
    extra = rng.standard_normal((500, d)).astype(np.float32)
    extra[17] = rows[np.flatnonzero(groups == 4)[0]]  # a tie between the ordered part and the tail
    extra_groups = np.where(np.arange(500) % 3 == 0, 2, 4).astype(np.int32)
    idx.add(extra, ids=np.arange(n, n + 500), group_codes=extra_groups)
    ref, got = _both(hvs, idx, q, 3, 1, 20, group=4)
    _same(ref, got, "group range + tail")
    all_rows = np.concatenate([rows, extra])
    keep = np.concatenate([groups == 4, extra_groups == 4])
    for i in range(3):
        parity.check_topk(got[0][i], got[1][i], got[2][i], parity.full_distances(all_rows, q_host[i], "f16"), 20, keep=keep,
                          label=f"tail q={i}")
    idx.close()


def test_the_call_can_be_captured_into_a_graph(hvs):
    """what the entry point is for: a caller that records its launches (no allocation may happen while it does)"""
    lib = hvs.load_library()
    rng = np.random.default_rng(9)
    n, d, nq, k = 50_000, 96, 4, 10
    rows = rng.standard_normal((n, d)).astype(np.float32)
    stream = torch.cuda.Stream()
    idx = hvs.HvsIndex(d, "f16", stream=stream.cuda_stream)
    idx.add(rows)
    q_host = rng.standard_normal((nq, d)).astype(np.float32)
    q = torch.from_numpy(q_host).cuda()
    dist = torch.full((nq, k), -7.0, dtype=torch.float32, device="cuda")
    ids = torch.full((nq, k), -7, dtype=torch.int64, device="cuda")
    cnt = torch.full((nq,), -7, dtype=torch.int32, device="cuda")

    def call():
        rc = lib.hvs_search_noalloc(idx._h, _ptr(q), nq, 1, k, -1, None, math.inf, 0, -1, _ptr(dist), _ptr(ids), _ptr(cnt))
        assert rc == 0, lib.hvs_last_error()

    call()  # once outside: kernel attributes are set on first use
    idx.sync()
    expect = (dist.cpu().numpy().copy(), ids.cpu().numpy().copy(), cnt.cpu().numpy().copy())
    dist.fill_(-7.0)
    ids.fill_(-7)
    cnt.fill_(-7)
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph, stream=stream):
        call()
    assert float(dist[0, 0].item()) == -7.0, "capturing must record the launches, not run them"
    q.copy_(torch.from_numpy(q_host[::-1].copy()))  # new inputs in the captured buffers
    graph.replay()
    torch.cuda.synchronize()
    for i in range(nq):
        parity.check_topk(dist[i].cpu().numpy(), ids[i].cpu().numpy(), cnt[i].cpu().numpy(), parity.full_distances(rows, q_host[nq - 1 - i], "f16"), k,
                          label=f"replayed q={i}")
    q.copy_(torch.from_numpy(q_host))
    graph.replay()
    torch.cuda.synchronize()
    _same(expect, (dist.cpu().numpy(), ids.cpu().numpy(), cnt.cpu().numpy()), "replay with the first inputs")
    del graph
    idx.close()
