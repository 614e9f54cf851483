"""CPU tier: the N > 1 exchange logic (row shards, all-gather, merge, centroid all-reduce) on two
``gloo`` ranks.  Local search / assignment and the merge are oracle-backed test doubles here; the
product wires the HIP kernels (``ShardedSearch.for_index``)."""

import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import kmeans_oracle, pgvector_oracle as po


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _merge_cpu(d, i, c, k):
    """[nq, P, k] lists -> [nq, k] by (distance, id): same contract as hvs_merge_topk."""
    nq, P, _ = d.shape
    od = torch.full((nq, k), float("inf"))
    oi = torch.full((nq, k), -1, dtype=torch.int64)
    oc = torch.zeros(nq, dtype=torch.int32)
    for q in range(nq):
        pairs = []
        for p in range(P):
            for j in range(int(c[q, p])):
                pairs.append((float(d[q, p, j]), int(i[q, p, j])))
        pairs.sort()
        pairs = pairs[:k]
        oc[q] = len(pairs)
        for j, (dd, ii) in enumerate(pairs):
            od[q, j] = dd
            oi[q, j] = ii
    return od, oi, oc


def _worker(rank, world, port, n, dim, k, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from hirag_prod_amd.sharded import ShardedSearch, shard_range, sharded_kmeans_step

    rows = po.synth_rows(1234, 0, n, dim)
    rows[10] = rows[n - 5]  # a cross-shard tie: the lower global id must come first
    lo, hi = shard_range(n, rank, world)
    queries = po.synth_rows(4321, 0, 6, dim)
    queries[0] = rows[10]

    def local_search(q, kk, **kw):
        q = q.numpy()
        d = torch.full((q.shape[0], kk), float("inf"))
        i = torch.full((q.shape[0], kk), -1, dtype=torch.int64)
        c = torch.zeros(q.shape[0], dtype=torch.int32)
        for j in range(q.shape[0]):
            idx, dd = po.query(rows[lo:hi], q[j], k=kk, threshold=kw.get("max_distance"))
            c[j] = len(idx)
            i[j, : len(idx)] = torch.from_numpy(idx + lo)  # global ids
            d[j, : len(idx)] = torch.from_numpy(dd.astype(np.float32))
        return d, i, c

    s = ShardedSearch(local_search, _merge_cpu)
    d, i, c = s.search(torch.from_numpy(queries), k)
    d2, i2, c2 = s.search(torch.from_numpy(queries), k, max_distance=0.97)

    cents = torch.from_numpy(rows[:7].copy())

    def assign_fn(cc):
        a, _ = kmeans_oracle.assign(rows[lo:hi], cc.numpy())
        _, sums, counts = kmeans_oracle.update(rows[lo:hi], a, cc.numpy())
        return torch.from_numpy(a), torch.from_numpy(sums.astype(np.float32)), torch.from_numpy(counts)

    new_c, a, counts = sharded_kmeans_step(assign_fn, cents)
    torch.save({"d": d, "i": i, "c": c, "d2": d2, "i2": i2, "c2": c2, "new_c": new_c, "a": a, "counts": counts,
                "lo": lo, "hi": hi}, os.path.join(out_dir, f"rank{rank}.pt"))
    dist.destroy_process_group()


def test_two_rank_search_and_kmeans(tmp_path):
    n, dim, k, world = 3001, 48, 10, 2
    port = _free_port()
    mp.spawn(_worker, args=(world, port, n, dim, k, str(tmp_path)), nprocs=world, join=True)
    r0 = torch.load(tmp_path / "rank0.pt")
    r1 = torch.load(tmp_path / "rank1.pt")
    assert (r0["lo"], r0["hi"], r1["lo"], r1["hi"]) == (0, 1501, 1501, 3001)
    rows = po.synth_rows(1234, 0, n, dim)
    rows[10] = rows[n - 5]
    queries = po.synth_rows(4321, 0, 6, dim)
    queries[0] = rows[10]
    for r in (r0, r1):  # every rank holds the merged result
        for j in range(6):
            idx, dd = po.query(rows, queries[j], k=k, threshold=None)
            assert r["i"][j].tolist() == idx.tolist()
            np.testing.assert_allclose(r["d"][j].numpy(), dd, atol=1e-6)
            idx, dd = po.query(rows, queries[j], k=k, threshold=0.97)
            assert int(r["c2"][j]) == len(idx) and r["i2"][j][: len(idx)].tolist() == idx.tolist()
        assert r["i"][0][:2].tolist() == [10, n - 5]  # tie across shards: lower global id first
    a, _ = kmeans_oracle.assign(rows, rows[:7])
    new_c, _, counts = kmeans_oracle.update(rows, a, rows[:7])
    np.testing.assert_array_equal(torch.cat([r0["a"], r1["a"]]).numpy(), a)
    for r in (r0, r1):
        np.testing.assert_array_equal(r["counts"].numpy(), counts)
        np.testing.assert_allclose(r["new_c"].numpy(), new_c, atol=1e-5)


def test_shard_range_covers_rows():
    from hirag_prod_amd.sharded import shard_range

    for n in (0, 1, 7, 10_000_000):
        for w in (1, 2, 3, 8):
            spans = [shard_range(n, r, w) for r in range(w)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        shard_range(10, 2, 2)
