"""GPU tier: the cross-shard merge kernel, with two shards of one corpus held as two indexes on the
same GPU (what two ranks would hold), merged exactly as ShardedSearch does after its all-gather."""

import os

import numpy as np
import pytest
import torch

from oracle import pgvector_oracle as po
from tests import parity

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("k", [1, 10, 100])
def test_merge_of_two_shards_equals_one_corpus(k):
    import hirag_prod_amd as hvs
    from hirag_prod_amd.sharded import ShardedSearch, shard_range

    n, d, world = 30_001, 128, 3
    rows = po.synth_rows(7, 0, n, d)
    rows[5] = rows[n - 2]  # tie across shards
    q = po.synth_rows(8, 0, 40, d)
    q[0] = rows[5]
    qt = torch.from_numpy(q).cuda()
    shards = []
    for r in range(world):
        lo, hi = shard_range(n, r, world)
        idx = hvs.HvsIndex(d, "f32")
        idx.add(rows[lo:hi], ids=np.arange(lo, hi))
        shards.append(idx)
    parts = [s.search(qt, k=k, max_distance=1.05, algo="mfma" if i % 2 else "scan") for i, s in enumerate(shards)]
    gd = torch.stack([p[0] for p in parts], dim=1).contiguous()  # [nq, P, k]
    gi = torch.stack([p[1] for p in parts], dim=1).contiguous()
    gc = torch.stack([p[2] for p in parts], dim=1).contiguous()
    md, mi, mc = hvs.merge_topk(gd, gi, gc, k)
    torch.cuda.synchronize()
    md, mi, mc = md.cpu().numpy(), mi.cpu().numpy(), mc.cpu().numpy()
    for j in range(q.shape[0]):
        full = parity.full_distances(rows, q[j], "f32")
        parity.check_topk(md[j], mi[j], mc[j], full, k, threshold=1.05, label=f"merged k={k} q={j}")
    if k >= 2:
        assert mi[0][:2].tolist() == [5, n - 2]
    # without a communicator ShardedSearch is index.search
    one = ShardedSearch.for_index(shards[0])
    d1, i1, c1 = one.search(qt, k)
    assert torch.equal(i1, shards[0].search(qt, k=k)[1])
    for s in shards:
        s.close()


def _one_index_answers():
    """What the sharded runs must reproduce: the same calls on ONE index holding the whole corpus."""
    import hirag_prod_amd as hvs
    from hirag_prod_amd.sharded import ShardedSearch, sharded_kmeans_step
    from tests import sharded_worker as w

    rows, q, groups = w.corpus()
    idx = hvs.HvsIndex(w.DIM, "f32")
    idx.add(rows, group_codes=groups)
    out = w.searches(ShardedSearch.for_index(idx), q, torch)
    cents = torch.from_numpy(rows[:11].copy()).cuda()
    new_c, assign, counts = sharded_kmeans_step(idx, cents)
    out.update(new_c=new_c.cpu().numpy(), assign=assign.cpu().numpy(), counts=counts.cpu().numpy())
    idx.close()
    return rows, q, groups, out


def _check_against_oracle(rows, q, groups, got):
    for j in range(6):
        parity.check_topk(got["scan_k10_d"][j], got["scan_k10_i"][j], got["scan_k10_c"][j], parity.full_distances(rows, q[j], "f32"), 10,
                          label=f"sharded scan q={j}")
    for j in range(q.shape[0]):
        full = parity.full_distances(rows, q[j], "f32")
        parity.check_topk(got["mfma_k10_d"][j], got["mfma_k10_i"][j], got["mfma_k10_c"][j], full, 10, label=f"sharded mfma q={j}")
        parity.check_topk(got["auto_k100_thr_d"][j], got["auto_k100_thr_i"][j], got["auto_k100_thr_c"][j], full, 100, threshold=0.93,
                          label=f"sharded auto q={j}")
    for j in range(q.shape[0] // 3):
        parity.check_topk(got["fold3_group_host_d"][j], got["fold3_group_host_i"][j], got["fold3_group_host_c"][j],
                          parity.full_distances(rows, q[3 * j : 3 * j + 3], "f32"), 7, threshold=0.99, keep=groups == 1,
                          label=f"sharded fold q={j}")
    assert got["scan_k10_i"][0][:2].tolist() == [5, len(rows) - 2]  # the tie across shards: lower global id first


def _run(args, timeout):
    import subprocess
    import sys

    return subprocess.Popen([sys.executable, os.path.join(os.path.dirname(__file__), "sharded_worker.py")] + [str(a) for a in args],
                            stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True), timeout


def test_two_processes_on_one_gpu_exchange_through_the_host_transport(tmp_path):
    """Two fresh processes, each building its shard with the real kernels on GPU 0 and meeting in hvs_search_sharded /
    hvs_allreduce_centroids through the host-staged transport (gloo): every rank's merged answer equals the oracle's and
    the one-index answer bit for bit."""
    import socket

    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    world = 2
    procs = [_run(["two_ranks", r, world, port, tmp_path / f"rank{r}.npz"], 240) for r in range(world)]
    logs = []
    for p, timeout in procs:
        try:
            logs.append(p.communicate(timeout=timeout)[0])
        except Exception:
            for q_, _ in procs:
                q_.kill()
            raise
    for (p, _), log in zip(procs, logs):
        assert p.returncode == 0, log[-3000:]
    rows, q, groups, one = _one_index_answers()
    ranks = [np.load(tmp_path / f"rank{r}.npz") for r in range(world)]
    for r in ranks:
        _check_against_oracle(rows, q, groups, r)
        for name in ("scan_k10", "mfma_k10", "auto_k100_thr", "fold3_group_host"):
            for part in ("_d", "_i", "_c"):
                assert np.array_equal(r[name + part].view(np.uint8), one[name + part].view(np.uint8)), f"{name}{part} differs from the one-index answer"
        np.testing.assert_array_equal(r["counts"], one["counts"])
        np.testing.assert_allclose(r["new_c"], one["new_c"], rtol=1e-5, atol=1e-6)  # sums added in another order
    np.testing.assert_array_equal(np.concatenate([r["assign"] for r in ranks]), one["assign"])
    assert [(int(r["lo"]), int(r["hi"])) for r in ranks] == [(0, 20_002), (20_002, 40_003)]


def test_eight_processes_on_one_gpu_at_k_100(tmp_path):
    """The record strides and the 8-way merge of hvs_search_sharded as the 8-GPU run uses them (k = 100, one rank without
    a row, one with fewer rows than k, a filter that leaves a third without a result), through the host transport; and a
    rank whose local search fails: every rank gets an error instead of waiting in the exchange, and the next call works."""
    import socket

    import hirag_prod_amd as hvs
    from hirag_prod_amd.sharded import ShardedSearch
    from tests import sharded_worker as w

    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    world = 8
    procs = [_run(["eight_ranks", r, world, port, tmp_path / f"rank{r}.npz"], 400) for r in range(world)]
    logs = []
    for p, timeout in procs:
        try:
            logs.append(p.communicate(timeout=timeout)[0])
        except Exception:
            for q_, _ in procs:
                q_.kill()
            raise
    for (p, _), log in zip(procs, logs):
        assert p.returncode == 0, log[-3000:]
    rows, q, groups = w.eight_corpus()
    idx = hvs.HvsIndex(w.EIGHT_DIM, "f16")
    idx.add(rows, group_codes=groups)
    one = w.eight_searches(ShardedSearch.for_index(idx), q, torch)
    after = idx.search(q[:2], k=10, algo="scan")[1]
    idx.close()
    ranks = [np.load(tmp_path / f"rank{r}.npz") for r in range(world)]
    for r, got in enumerate(ranks):
        for name in ("scan", "auto", "group0_host", "k10"):
            for part in ("_d", "_i", "_c"):
                assert np.array_equal(got[name + part].view(np.uint8), one[name + part].view(np.uint8)), f"rank {r}: {name}{part} differs from the one-index answer"
        assert int(got["failure_seen"]) == 1 and int(got["failure_names_rank"]) == 1, f"rank {r} did not see rank 5's failure"
        assert np.array_equal(got["after_failure_i"], after)
    got = ranks[0]
    for j in range(5):
        parity.check_topk(got["scan_d"][j], got["scan_i"][j], got["scan_c"][j], parity.full_distances(rows, q[j], "f16"), w.EIGHT_K, label=f"8 ranks scan q={j}")
    for j in range(q.shape[0]):
        parity.check_topk(got["auto_d"][j], got["auto_i"][j], got["auto_c"][j], parity.full_distances(rows, q[j], "f16"), w.EIGHT_K, threshold=0.97,
                          label=f"8 ranks auto q={j}")
    for j in range(q.shape[0] // 2):
        parity.check_topk(got["group0_host_d"][j], got["group0_host_i"][j], got["group0_host_c"][j],
                          parity.full_distances(rows, q[2 * j : 2 * j + 2], "f16"), w.EIGHT_K, keep=groups == 0, label=f"8 ranks filter q={j}")
    assert got["scan_i"][0][:2].tolist() == [17, 12010]  # the tie across shards: lower global id first


def test_rccl_communicator_of_one_rank(tmp_path):
    """ncclCommInitRank / ncclAllGather / ncclAllReduce inside libhvs on real hardware (one rank: all a one-GPU box
    allows), in a process of its own so that trouble in the collective library cannot take the test run down."""
    p, timeout = _run(["rccl_one_rank", tmp_path / "rccl.npz"], 240)
    try:
        log = p.communicate(timeout=timeout)[0]
    except Exception:
        p.kill()
        raise
    assert p.returncode == 0, log[-3000:]
    rows, q, groups, one = _one_index_answers()
    got = np.load(tmp_path / "rccl.npz")
    _check_against_oracle(rows, q, groups, got)
    for name in ("scan_k10", "mfma_k10", "auto_k100_thr", "fold3_group_host"):
        for part in ("_d", "_i", "_c"):
            assert np.array_equal(got[name + part].view(np.uint8), one[name + part].view(np.uint8)), f"{name}{part}"
    np.testing.assert_array_equal(got["assign"], one["assign"])
    np.testing.assert_array_equal(got["new_c"], one["new_c"])
