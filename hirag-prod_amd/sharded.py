"""Row-sharded search and k-means over the GPUs of one node: one process per GPU, RCCL over xGMI.

The corpus is split into contiguous row ranges, one per rank (ids carry the global row number), the
queries are replicated, every rank computes its local top-k with the HIP kernels, and a single
all-gather of ``k`` (distance, id) pairs per query and rank is followed by a k-way merge on every
rank.  The merge orders by (distance, id), so the result does not depend on the number of shards.
k-means all-reduces the ``[K, D]`` partial sums and the ``[K]`` counts.

The reference is single-node (one PostgreSQL server); this module has no counterpart there.

``ShardedSearch`` takes the local-search and merge callables as arguments so that the exchange
logic can be exercised on CPU with the ``gloo`` backend; ``ShardedSearch.for_index`` wires it to
an ``HvsIndex`` and the HIP merge kernel, which is the only product configuration.
"""

from __future__ import annotations

from typing import Callable, Optional, Tuple

import torch
import torch.distributed as dist


def shard_range(n_rows: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous shard ``[lo, hi)`` of rank ``rank``; sizes differ by at most one row."""
    if world <= 0 or not (0 <= rank < world):
        raise ValueError(f"bad rank/world {rank}/{world}")
    base, rem = divmod(int(n_rows), world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


class ShardedSearch:
    """Exact top-k over a row-sharded corpus.

    local_search(queries, k, **kw) -> (dist [nq, k] f32, ids [nq, k] i64 global ids, count [nq] i32)
    merge(dist [nq, P, k], ids [nq, P, k], count [nq, P], k) -> (dist [nq, k], ids [nq, k], count [nq])
    Tensors live on the device of the process group's backend (GPU for nccl, CPU for gloo).
    """

    def __init__(self, local_search: Callable, merge: Callable, group: Optional[dist.ProcessGroup] = None):
        self.local_search = local_search
        self.merge = merge
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0

    @classmethod
    def for_index(cls, index, group: Optional[dist.ProcessGroup] = None) -> "ShardedSearch":
        from .index import merge_topk

        def local(queries, k, **kw):
            return index.search(queries, k=k, **kw)

        return cls(local, merge_topk, group)

    def search(self, queries: torch.Tensor, k: int, **kw):
        d, i, c = self.local_search(queries, k, **kw)
        if self.world == 1:
            return d, i, c
        nq = d.shape[0]
        # outputs are the ranks' tensors concatenated along dim 0 (the layout every backend accepts)
        gd = torch.empty((self.world * nq, k), dtype=d.dtype, device=d.device)
        gi = torch.empty((self.world * nq, k), dtype=i.dtype, device=i.device)
        gc = torch.empty((self.world * nq,), dtype=c.dtype, device=c.device)
        # three small collectives; each moves world * nq * k * {4, 8} (or nq * 4) bytes
        dist.all_gather_into_tensor(gd, d.contiguous(), group=self.group)
        dist.all_gather_into_tensor(gi, i.contiguous(), group=self.group)
        dist.all_gather_into_tensor(gc, c.contiguous(), group=self.group)
        w = self.world
        return self.merge(gd.view(w, nq, k).permute(1, 0, 2).contiguous(), gi.view(w, nq, k).permute(1, 0, 2).contiguous(),
                          gc.view(w, nq).permute(1, 0).contiguous(), k)


def sharded_kmeans_step(assign_fn: Callable, centroids: torch.Tensor, group: Optional[dist.ProcessGroup] = None):
    """One Lloyd iteration over row shards.

    assign_fn(centroids) -> (assign [n_local] i32, sums [K, D] f32, counts [K] i64) on this rank's rows.
    Returns (new_centroids [K, D] f32, assign, counts [K] i64 global).  Empty clusters keep their
    previous centroid.  The sums are added in the collective's order, so centroids can differ in the
    last bits between GPU counts; assignments for given centroids do not.
    """
    assign, sums, counts = assign_fn(centroids)
    if dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(sums, op=dist.ReduceOp.SUM, group=group)
        dist.all_reduce(counts, op=dist.ReduceOp.SUM, group=group)
    new_c = centroids.clone().to(torch.float32)
    nz = counts > 0
    new_c[nz] = sums[nz] / counts[nz].unsqueeze(1).to(torch.float32)
    return new_c, assign, counts
