"""Row-sharded search and k-means over the GPUs of one node: one process per GPU, RCCL over xGMI.

The corpus is split into contiguous row ranges, one per rank (ids carry the global row number), the
queries are replicated, every rank computes its local top-k with the HIP kernels, ONE all-gather of a
packed ``(distances, ids, counts)`` record per rank is followed by a k-way merge on every rank.  The
merge orders by (distance, id), so the result does not depend on the number of shards.  k-means
all-reduces the ``[K, D]`` partial sums and the ``[K]`` counts.

All of that happens below Python, in ``libhvs.so`` (``hvs_comm_*``, ``hvs_search_sharded``,
``hvs_allreduce_centroids``): this module only makes the communicator and passes pointers.  Two
transports: RCCL (``HvsComm.rccl``; the rendezvous token travels over whatever the caller has, here
``torch.distributed``) and host-staged callbacks (``HvsComm.host``; the exchange itself runs on a
``gloo`` group) for ranks that cannot form an RCCL communicator.

The reference is single-node (one PostgreSQL server); this module has no counterpart there.
"""

from __future__ import annotations

import ctypes
import math
from typing import Optional, Tuple

import numpy as np

from . import _lib
from ._lib import ALGO_CODE, HvsError, check, load_library


def shard_range(n_rows: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous shard ``[lo, hi)`` of rank ``rank``; sizes differ by at most one row."""
    if world <= 0 or not (0 <= rank < world):
        raise ValueError(f"bad rank/world {rank}/{world}")
    base, rem = divmod(int(n_rows), world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def host_exchange(group=None):
    """The two functions ``hvs_comm_init_host`` takes, over a ``torch.distributed`` group whose backend
    moves host tensors (gloo).  They see raw addresses: ``allgather(ctx, send, recv, bytes)``,
    ``allreduce(ctx, buf, count, dtype)``; 0 = success, as the C side expects."""
    import torch
    import torch.distributed as dist

    def _view(addr, nbytes, dtype=np.uint8):
        buf = (ctypes.c_char * nbytes).from_address(addr)
        return torch.from_numpy(np.frombuffer(buf, dtype=dtype))

    def allgather(ctx, send, recv, nbytes):
        try:
            world = dist.get_world_size(group)
            out = _view(recv, nbytes * world)
            dist.all_gather([out[r * nbytes:(r + 1) * nbytes] for r in range(world)], _view(send, nbytes).clone(), group=group)
            return 0
        except Exception:  # an exception must not unwind through the C frames
            import traceback

            traceback.print_exc()
            return 1

    def allreduce(ctx, buf, count, dtype):
        try:
            np_dtype, size = (np.float32, 4) if dtype == _lib.HVS_F32 else (np.int64, 8)
            dist.all_reduce(_view(buf, count * size, np_dtype), op=dist.ReduceOp.SUM, group=group)
            return 0
        except Exception:
            import traceback

            traceback.print_exc()
            return 1

    return _lib.ALLGATHER_FN(allgather), _lib.ALLREDUCE_FN(allreduce)


class HvsComm:
    """``hvs_comm``: this rank's end of the exchange between the row shards."""

    def __init__(self, handle, rank: int, world: int, transport: str, keep=()):
        self._lib = load_library()
        self._h = handle
        self.rank, self.world, self.transport = rank, world, transport
        self._keep = keep  # the ctypes callbacks must outlive the communicator

    @staticmethod
    def unique_id() -> bytes:
        """The RCCL rendezvous token; rank 0 makes it and hands it to the others."""
        buf = ctypes.create_string_buffer(_lib.HVS_UNIQUE_ID_BYTES)
        check(load_library().hvs_comm_unique_id(buf, _lib.HVS_UNIQUE_ID_BYTES))
        return buf.raw

    @classmethod
    def rccl(cls, rank: int, world: int, unique_id: bytes, device: int) -> "HvsComm":
        if len(unique_id) != _lib.HVS_UNIQUE_ID_BYTES:
            raise ValueError(f"unique_id must be {_lib.HVS_UNIQUE_ID_BYTES} bytes")
        h = ctypes.c_void_p()
        check(load_library().hvs_comm_init(ctypes.byref(h), rank, world, unique_id, device))
        return cls(h, rank, world, "rccl")

    @classmethod
    def host(cls, rank: int, world: int, allgather, allreduce) -> "HvsComm":
        """``allgather`` / ``allreduce``: ``_lib.ALLGATHER_FN`` / ``_lib.ALLREDUCE_FN`` objects (see ``host_exchange``)."""
        h = ctypes.c_void_p()
        check(load_library().hvs_comm_init_host(ctypes.byref(h), rank, world, allgather, allreduce, None))
        return cls(h, rank, world, "host", keep=(allgather, allreduce))

    @classmethod
    def from_process_group(cls, device: int, group=None, transport: str = "rccl") -> "HvsComm":
        """One communicator per rank of an initialised ``torch.distributed`` group.  ``rccl``: rank 0's token is
        broadcast through the group, then every rank joins ``ncclCommInitRank`` inside libhvs.  ``host``: the exchanges
        run as host collectives on the group (which must be able to move CPU tensors: gloo)."""
        import torch.distributed as dist

        rank, world = dist.get_rank(group), dist.get_world_size(group)
        if transport == "host":
            return cls.host(rank, world, *host_exchange(group))
        if transport != "rccl":
            raise ValueError("transport must be 'rccl' or 'host'")
        token = [cls.unique_id() if rank == 0 else None]
        dist.broadcast_object_list(token, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
        return cls.rccl(rank, world, token[0], device)

    def close(self) -> None:
        if getattr(self, "_h", None) is not None and self._h.value:
            self._lib.hvs_comm_destroy(self._h)
            self._h = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def _ptr(x):
    if x is None:
        return None
    if hasattr(x, "data_ptr"):
        return ctypes.c_void_p(x.data_ptr())
    return x.ctypes.data_as(ctypes.c_void_p)


class ShardedSearch:
    """Exact top-k over a row-sharded corpus: ``index`` holds this rank's rows (global row numbers as ids).
    Without a communicator (one GPU) it is ``index.search``."""

    def __init__(self, index, comm: Optional[HvsComm] = None):
        self.index = index
        self.comm = comm
        self.world = comm.world if comm else 1
        self.rank = comm.rank if comm else 0

    @classmethod
    def for_index(cls, index, comm: Optional[HvsComm] = None) -> "ShardedSearch":
        return cls(index, comm)

    def search(self, queries, k: int, fold: int = 1, group_filter: int = -1, row_mask=None, max_distance: float = math.inf,
               algo: str = "auto"):
        """``queries``: ``[nq * fold, dim]`` float32, a torch tensor on the index's GPU (-> torch results) or a numpy
        array (-> numpy results).  Every rank passes the same queries and receives the same merged answer."""
        if self.comm is None:
            return self.index.search(queries, k=k, fold=fold, group_filter=group_filter, row_mask=row_mask, max_distance=max_distance,
                                     algo=algo)
        idx = self.index
        on_device = hasattr(queries, "data_ptr")
        if on_device:
            import torch

            q = queries.contiguous().to(torch.float32)
            nq = q.shape[0] // fold
            torch.cuda.current_stream(q.device).synchronize()  # the index works on its own stream
            dist_ = torch.empty((nq, k), dtype=torch.float32, device=q.device)
            ids = torch.empty((nq, k), dtype=torch.int64, device=q.device)
            cnt = torch.empty((nq,), dtype=torch.int32, device=q.device)
            mask = row_mask.contiguous() if row_mask is not None else None
        else:
            q = np.ascontiguousarray(np.atleast_2d(np.asarray(queries)), dtype=np.float32)
            nq = q.shape[0] // fold
            dist_ = np.empty((nq, k), dtype=np.float32)
            ids = np.empty((nq, k), dtype=np.int64)
            cnt = np.empty((nq,), dtype=np.int32)
            mask = np.ascontiguousarray(row_mask, dtype=np.uint32) if row_mask is not None else None
        if q.shape[1] != idx.dim or q.shape[0] % fold:
            raise ValueError(f"queries must be [nq*fold, {idx.dim}]")
        with idx._lock:
            check(idx._lib.hvs_search_sharded(idx._h, self.comm._h, _ptr(q), nq, fold, k, group_filter, _ptr(mask), float(max_distance),
                                              _ptr(dist_), _ptr(ids), _ptr(cnt), 1 if on_device else 0, ALGO_CODE[algo]))
            if on_device:
                check(idx._lib.hvs_sync(idx._h))
        return dist_, ids, cnt


def sharded_kmeans_step(index, centroids, comm: Optional[HvsComm] = None):
    """One Lloyd iteration over row shards: the local E-step and partial sums (``hvs_kmeans_assign``), the
    all-reduce of sums and counts (``hvs_allreduce_centroids``), the division.

    ``centroids``: ``[K, D]`` float32, torch tensor on the index's GPU or numpy array.  Returns
    ``(new_centroids [K, D], assign [n_local] i32, counts [K] i64 global)``.  Empty clusters keep their previous
    centroid.  The sums are added in the collective's order, so centroids can differ in the last bits between GPU
    counts; assignments for given centroids do not.
    """
    assign, sums, counts = index.kmeans_assign(centroids)
    on_device = hasattr(sums, "data_ptr")
    if comm is not None and comm.world > 1:
        with index._lock:
            check(index._lib.hvs_allreduce_centroids(index._h, comm._h, _ptr(sums), _ptr(counts), int(sums.shape[0]), 1 if on_device else 0))
            check(index._lib.hvs_sync(index._h))
    if on_device:
        import torch

        new_c = centroids.clone().to(torch.float32)
        nz = counts > 0
        new_c[nz] = sums[nz] / counts[nz].unsqueeze(1).to(torch.float32)
    else:
        new_c = np.array(centroids, dtype=np.float32, copy=True)
        nz = counts > 0
        new_c[nz] = sums[nz] / counts[nz, None].astype(np.float32)
    return new_c, assign, counts


__all__ = ["HvsComm", "HvsError", "ShardedSearch", "host_exchange", "shard_range", "sharded_kmeans_step"]
