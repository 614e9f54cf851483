"""Child process of tests/test_gpu_sharded.py: one rank of a row-sharded table.

    python tests/sharded_worker.py two_ranks <rank> <world> <port> <out.npz>
        Both ranks run on GPU 0 with the real kernels; the exchange is hvs_comm's host-staged transport over a gloo group
        (two processes cannot form an RCCL communicator on one GPU).
    python tests/sharded_worker.py rccl_one_rank <out.npz>
        A one-rank RCCL communicator: ncclCommInitRank, ncclAllGather and ncclAllReduce inside libhvs on real hardware.
"""

import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

N, DIM, NQ = 40_003, 256, 24


def corpus():
    from oracle import pgvector_oracle as po

    rows = po.synth_rows(77, 0, N, DIM)
    rows[5] = rows[N - 2]  # tie across shards
    rows[9] = 0.0
    q = po.synth_rows(78, 0, NQ, DIM)
    q[0] = rows[5]
    groups = (np.arange(N) % 3).astype(np.int32)
    return rows, q, groups


def searches(searcher, q, torch):
    """the calls both sides of the comparison make; -> dict of numpy arrays"""
    out = {}
    qt = torch.from_numpy(q).cuda()

    def put(name, res):
        d, i, c = res
        if hasattr(d, "cpu"):
            d, i, c = d.cpu().numpy(), i.cpu().numpy(), c.cpu().numpy()
        out[name + "_d"], out[name + "_i"], out[name + "_c"] = d, i, c

    put("scan_k10", searcher.search(qt[:6].contiguous(), 10, algo="scan"))
    put("mfma_k10", searcher.search(qt, 10, algo="mfma_f32"))
    put("auto_k100_thr", searcher.search(qt, 100, max_distance=0.93))
    put("fold3_group_host", searcher.search(q, 7, fold=3, group_filter=1, max_distance=0.99, algo="scan"))  # host buffers
    return out


def two_ranks(rank, world, port, out_path):
    import torch
    import torch.distributed as dist

    import hirag_prod_amd as hvs
    from hirag_prod_amd.sharded import HvsComm, ShardedSearch, shard_range, sharded_kmeans_step

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    rows, q, groups = corpus()
    lo, hi = shard_range(N, rank, world)
    idx = hvs.HvsIndex(DIM, "f32", device=0)
    idx.add(rows[lo:hi], ids=np.arange(lo, hi), group_codes=groups[lo:hi])
    comm = HvsComm.from_process_group(device=0, transport="host")
    assert (comm.rank, comm.world, comm.transport) == (rank, world, "host")
    out = searches(ShardedSearch.for_index(idx, comm), q, torch)
    cents = torch.from_numpy(rows[:11].copy()).cuda()
    new_c, assign, counts = sharded_kmeans_step(idx, cents, comm)
    out.update(new_c=new_c.cpu().numpy(), assign=assign.cpu().numpy(), counts=counts.cpu().numpy(), lo=lo, hi=hi)
    np.savez(out_path, **out)
    comm.close()
    idx.close()
    dist.destroy_process_group()


# eight shards of one corpus: one holds no row at all, one fewer rows than k, the others a few thousand each
EIGHT_BOUNDS = [0, 3000, 6000, 9000, 9000, 12000, 12030, 16000, 20_011]
EIGHT_DIM, EIGHT_K = 96, 100


def eight_corpus():
    from oracle import pgvector_oracle as po

    n = EIGHT_BOUNDS[-1]
    rows = po.synth_rows(91, 0, n, EIGHT_DIM)
    rows[12010] = rows[17]  # a tie between the short shard and the first
    q = po.synth_rows(92, 0, 24, EIGHT_DIM)
    q[0] = rows[17]
    groups = np.zeros(n, np.int32)
    groups[6000:9000] = 1  # a filter on group 0 leaves rank 2 without a result
    return rows, q, groups


def eight_searches(searcher, q, torch):
    out = {}
    qt = torch.from_numpy(q).cuda()

    def put(name, res):
        d, i, c = res
        if hasattr(d, "cpu"):
            d, i, c = d.cpu().numpy(), i.cpu().numpy(), c.cpu().numpy()
        out[name + "_d"], out[name + "_i"], out[name + "_c"] = d, i, c

    put("scan", searcher.search(qt[:5].contiguous(), EIGHT_K, algo="scan"))
    put("auto", searcher.search(qt, EIGHT_K, max_distance=0.97))
    put("group0_host", searcher.search(q, EIGHT_K, fold=2, group_filter=0, algo="scan"))
    put("k10", searcher.search(qt, 10, algo="mfma"))
    return out


def eight_ranks(rank, world, port, out_path):
    """one of eight processes on GPU 0: the record layout and the 8-way merge of hvs_search_sharded at k = 100, and what
    happens when one rank's local search fails"""
    import torch
    import torch.distributed as dist

    import hirag_prod_amd as hvs
    from hirag_prod_amd.sharded import HvsComm, ShardedSearch

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    rows, q, groups = eight_corpus()
    lo, hi = EIGHT_BOUNDS[rank], EIGHT_BOUNDS[rank + 1]
    idx = hvs.HvsIndex(EIGHT_DIM, "f16", device=0)
    if hi > lo:
        idx.add(rows[lo:hi], ids=np.arange(lo, hi), group_codes=groups[lo:hi])
    comm = HvsComm.from_process_group(device=0, transport="host")
    searcher = ShardedSearch.for_index(idx, comm)
    out = eight_searches(searcher, q, torch)
    # rank 5's local search "fails": every rank must come back with an error instead of waiting in the exchange,
    # and the communicator must still work afterwards
    os.environ["HVS_TEST_FAIL_RANK"] = "5"
    try:
        searcher.search(q[:2], 10, algo="scan")
        out["failure_seen"] = np.array(0)
    except hvs.HvsError as e:
        out["failure_seen"] = np.array(1)
        out["failure_names_rank"] = np.array(1 if ("rank 5" in str(e) or "on purpose" in str(e)) else 0)
    del os.environ["HVS_TEST_FAIL_RANK"]
    d, i, c = searcher.search(q[:2], 10, algo="scan")
    out["after_failure_i"] = i
    np.savez(out_path, **out)
    comm.close()
    idx.close()
    dist.destroy_process_group()


def rccl_one_rank(out_path):
    import torch

    import hirag_prod_amd as hvs
    from hirag_prod_amd.sharded import HvsComm, ShardedSearch, sharded_kmeans_step

    rows, q, groups = corpus()
    idx = hvs.HvsIndex(DIM, "f32", device=0)
    idx.add(rows, group_codes=groups)
    comm = HvsComm.rccl(0, 1, HvsComm.unique_id(), 0)
    out = searches(ShardedSearch.for_index(idx, comm), q, torch)
    cents = torch.from_numpy(rows[:11].copy()).cuda()
    new_c, assign, counts = sharded_kmeans_step(idx, cents, comm)
    out.update(new_c=new_c.cpu().numpy(), assign=assign.cpu().numpy(), counts=counts.cpu().numpy())
    np.savez(out_path, **out)
    comm.close()
    idx.close()


if __name__ == "__main__":
    if sys.argv[1] == "two_ranks":
        two_ranks(int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), sys.argv[5])
    elif sys.argv[1] == "eight_ranks":
        eight_ranks(int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), sys.argv[5])
    else:
        rccl_one_rank(sys.argv[2])
