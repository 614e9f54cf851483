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
    else:
        rccl_one_rank(sys.argv[2])
