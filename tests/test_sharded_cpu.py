"""CPU tier: the N > 1 exchange plumbing on two ``gloo`` ranks.  What crosses between ranks on the host-staged
transport of ``hvs_comm`` are the two callbacks of ``hirag_prod_amd.sharded.host_exchange``; here they are invoked as
the C side invokes them (through their ctypes function pointers, on raw buffers), with the per-rank records built from
the oracle's answers, and the merged result is checked against the oracle over the whole corpus.  The GPU tier runs the
same exchange under the real kernels (tests/test_gpu_sharded.py)."""

import ctypes
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


def _worker(rank, world, port, n, dim, k, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from hirag_prod_amd import _lib
    from hirag_prod_amd.sharded import host_exchange, shard_range

    allgather, allreduce = host_exchange(None)
    rows = po.synth_rows(1234, 0, n, dim)
    rows[10] = rows[n - 5]  # a cross-shard tie: the lower global id must come first
    lo, hi = shard_range(n, rank, world)
    queries = po.synth_rows(4321, 0, 6, dim)
    queries[0] = rows[10]
    # this rank's record, laid out as hvs_search_sharded packs it: [nq, k] f32 | [nq, k] i64 | [nq] i32
    nq = queries.shape[0]
    rec = np.zeros(nq * k * 12 + nq * 4, dtype=np.uint8)
    d = rec[: nq * k * 4].view(np.float32).reshape(nq, k)
    i = rec[nq * k * 4 : nq * k * 12].view(np.int64).reshape(nq, k)
    c = rec[nq * k * 12 :].view(np.int32)
    d[:] = np.inf
    i[:] = -1
    for j in range(nq):
        idx, dd = po.query(rows[lo:hi], queries[j], k=k, threshold=0.97)
        c[j] = len(idx)
        i[j, : len(idx)] = idx + lo
        d[j, : len(idx)] = dd.astype(np.float32)
    recv = np.zeros(world * rec.size, dtype=np.uint8)
    rc = allgather(None, rec.ctypes.data, recv.ctypes.data, rec.size)
    assert rc == 0
    # k-means partial sums through the all-reduce callback
    cents = rows[:7].copy()
    a, _ = kmeans_oracle.assign(rows[lo:hi], cents)
    _, sums, counts = kmeans_oracle.update(rows[lo:hi], a, cents)
    sums = np.ascontiguousarray(sums, dtype=np.float32)
    counts = np.ascontiguousarray(counts, dtype=np.int64)
    assert allreduce(None, sums.ctypes.data, sums.size, _lib.HVS_F32) == 0
    assert allreduce(None, counts.ctypes.data, counts.size, _lib.HVS_REDUCE_I64) == 0
    np.savez(os.path.join(out_dir, f"rank{rank}.npz"), recv=recv, rec_bytes=rec.size, sums=sums, counts=counts, a=a, lo=lo, hi=hi)
    dist.destroy_process_group()


def _merge(recv, rec_bytes, world, nq, k):
    out = []
    for j in range(nq):
        pairs = []
        for r in range(world):
            rec = recv[r * rec_bytes : (r + 1) * rec_bytes]
            d = rec[: nq * k * 4].view(np.float32).reshape(nq, k)
            i = rec[nq * k * 4 : nq * k * 12].view(np.int64).reshape(nq, k)
            c = rec[nq * k * 12 :].view(np.int32)
            pairs += [(float(d[j, t]), int(i[j, t])) for t in range(int(c[j]))]
        out.append(sorted(pairs)[:k])
    return out


def test_two_rank_exchange_callbacks(tmp_path):
    n, dim, k, world = 3001, 48, 10, 2
    port = _free_port()
    mp.spawn(_worker, args=(world, port, n, dim, k, str(tmp_path)), nprocs=world, join=True)
    r0 = np.load(tmp_path / "rank0.npz")
    r1 = np.load(tmp_path / "rank1.npz")
    assert (int(r0["lo"]), int(r0["hi"]), int(r1["lo"]), int(r1["hi"])) == (0, 1501, 1501, 3001)
    assert np.array_equal(r0["recv"], r1["recv"])  # every rank holds every record, in rank order
    rows = po.synth_rows(1234, 0, n, dim)
    rows[10] = rows[n - 5]
    queries = po.synth_rows(4321, 0, 6, dim)
    queries[0] = rows[10]
    merged = _merge(r0["recv"], int(r0["rec_bytes"]), world, 6, k)
    for j in range(6):
        idx, dd = po.query(rows, queries[j], k=k, threshold=0.97)
        assert [p[1] for p in merged[j]] == idx.tolist()
        np.testing.assert_allclose([p[0] for p in merged[j]], dd, atol=1e-6)
    assert [p[1] for p in merged[0]][:2] == [10, n - 5]  # tie across shards: lower global id first
    a, _ = kmeans_oracle.assign(rows, rows[:7])
    _, sums, counts = kmeans_oracle.update(rows, a, rows[:7])
    np.testing.assert_array_equal(np.concatenate([r0["a"], r1["a"]]), a)
    for r in (r0, r1):
        np.testing.assert_array_equal(r["counts"], counts)
        np.testing.assert_allclose(r["sums"], sums, rtol=1e-5, atol=1e-4)


def test_kmeans_step_division_keeps_empty_clusters():
    from hirag_prod_amd.sharded import sharded_kmeans_step

    class OneShard:  # stands in for an HvsIndex: the E-step of the oracle
        def __init__(self, rows):
            self.rows = rows

        def kmeans_assign(self, c):
            a, _ = kmeans_oracle.assign(self.rows, c)
            _, sums, counts = kmeans_oracle.update(self.rows, a, c)
            return a, sums.astype(np.float32), counts

    rows = po.synth_rows(5, 0, 400, 16)
    cents = np.concatenate([rows[:3], 1e3 * np.ones((1, 16), np.float32)])  # the last centroid attracts nothing
    new_c, a, counts = sharded_kmeans_step(OneShard(rows), cents)
    assert counts[3] == 0 and np.array_equal(new_c[3], cents[3])
    exp, _, _ = kmeans_oracle.update(rows, a, cents)
    np.testing.assert_allclose(new_c[:3], exp[:3], atol=1e-5)


def test_bench_argument_path_for_two_gpus_stops_at_the_device(tmp_path):
    """`bench.py --gpus 2` without a launcher names the launcher; under one it reaches the GPU set-up (which a CPU box lacks)."""
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "1"], capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and "torch.distributed.run" in (r.stderr + r.stdout)
    if torch.cuda.is_available():
        return  # on a GPU box the launcher would start the real benchmark
    port = _free_port()
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", str(port), os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       capture_output=True, text=True, timeout=600)
    # both ranks parsed their arguments, read RANK / WORLD_SIZE, met in a process group and stopped at the missing device
    assert r.returncode != 0 and (r.stderr + r.stdout).count("no HIP device visible") >= 2, (r.stderr + r.stdout)[-2000:]
    # the process group is gloo (token, barriers, a max over ranks): libhvs's RCCL communicator is the only one a rank ever forms
    assert (r.stderr + r.stdout).count("process group backend: gloo") == 2, (r.stderr + r.stdout)[-2000:]


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


def test_bench_child_leg_reports_a_failed_leg_in_its_place():
    """The per-configuration legs of bench.py run in child processes; one that dies (here: no GPU in this container)
    must come back as an error object, not as an exception in the parent."""
    import importlib.util

    if torch.cuda.is_available():
        pytest.skip("on a GPU box the leg would run for real")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("bench_module", os.path.join(root, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    out = bench.child_leg("cfg2", 64, timeout_s=300)
    assert set(out) >= {"error"} and "exit status" in out["error"], out
    assert "HIP" in out.get("stderr_tail", "") or "hip" in out.get("stderr_tail", "").lower(), out
