"""``HvsIndex``: one embedded table (Chunks / Items / Triplets) resident in HBM.

A thin object wrapper over the C ABI (``include/hvs.h``).  Host data travels as
numpy arrays; data already on the GPU travels as torch tensors, of which only
``data_ptr()`` is used.  Every compute call goes to the HIP kernels in
``csrc/libhvs.so``; nothing here computes a distance on the CPU.
"""

from __future__ import annotations

import ctypes
import math
import threading
from typing import Optional, Tuple

import numpy as np

from . import _lib
from ._lib import ALGO_CODE, DTYPE_CODE, HvsError, check, load_library

_NP_OF = {"f32": np.float32, "f16": np.float16}


def _is_torch(x) -> bool:
    return type(x).__module__.startswith("torch") and hasattr(x, "data_ptr")


def _np_ptr(a: Optional[np.ndarray]):
    return None if a is None else a.ctypes.data_as(ctypes.c_void_p)


class HvsIndex:
    """Rows of one table on one GPU.

    dtype: "f32" (pgvector ``vector``), "f16" (``halfvec``, the reference's default:
    ``src/hirag_prod/configs/envs.py:12-13``) or "bf16".
    """

    def __init__(self, dim: int, dtype: str = "f16", device: int = 0, stream: Optional[int] = None):
        if dtype not in DTYPE_CODE:
            raise ValueError(f"dtype must be one of {sorted(DTYPE_CODE)}, got {dtype!r}")
        self._lib = load_library()
        self.dim = int(dim)
        self.dtype = dtype
        self.device = int(device)
        self._h = ctypes.c_void_p()
        self._lock = threading.Lock()  # calls on one index must not overlap
        check(self._lib.hvs_index_create(ctypes.byref(self._h), self.device, self.dim, DTYPE_CODE[dtype],
                                         ctypes.c_void_p(stream) if stream else None))

    # -- life cycle ------------------------------------------------------------------------------
    def close(self) -> None:
        if getattr(self, "_h", None) is not None and self._h.value:
            self._lib.hvs_index_destroy(self._h)
            self._h = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __len__(self) -> int:
        n = ctypes.c_int64(0)
        check(self._lib.hvs_index_size(self._h, ctypes.byref(n)))
        return n.value

    def reserve(self, capacity_rows: int) -> None:
        with self._lock:
            check(self._lib.hvs_index_reserve(self._h, int(capacity_rows)))

    def sync(self) -> None:
        check(self._lib.hvs_sync(self._h))

    # -- ingest ----------------------------------------------------------------------------------
    def _rows_arg(self, rows) -> Tuple[ctypes.c_void_p, int, int, int, object]:
        """-> (pointer, src dtype code, n, on_device, keepalive)"""
        if _is_torch(rows):
            import torch

            if rows.dim() != 2 or rows.shape[1] != self.dim:
                raise ValueError(f"rows must be [n, {self.dim}], got {tuple(rows.shape)}")
            code = {torch.float32: _lib.HVS_F32, torch.float16: _lib.HVS_F16, torch.bfloat16: _lib.HVS_BF16}.get(rows.dtype)
            if code is None:
                raise ValueError(f"unsupported tensor dtype {rows.dtype}")
            rows = rows.contiguous()
            if rows.device.type != "cuda" or (rows.device.index or 0) != self.device:
                raise ValueError("device tensors must live on the index's GPU")
            torch.cuda.current_stream(rows.device).synchronize()  # the index works on its own stream
            return ctypes.c_void_p(rows.data_ptr()), code, rows.shape[0], 1, rows
        a = np.asarray(rows)
        if a.ndim != 2 or a.shape[1] != self.dim:
            raise ValueError(f"rows must be [n, {self.dim}], got {a.shape}")
        if a.dtype == np.float16 and self.dtype == "f16":
            code = _lib.HVS_F16
        elif a.dtype == np.uint16 and self.dtype == "bf16":
            code = _lib.HVS_BF16  # raw bf16 bit patterns
        else:
            a = a.astype(np.float32, copy=False)
            code = _lib.HVS_F32
        a = np.ascontiguousarray(a)
        return _np_ptr(a), code, a.shape[0], 0, a

    def add(self, rows, ids=None, group_codes=None) -> Tuple[int, int]:
        """Append rows; returns the half-open range of row positions they got."""
        ptr, code, n, on_dev, keep = self._rows_arg(rows)
        ids_keep = groups_keep = None
        ids_ptr = groups_ptr = None
        if ids is not None:
            if on_dev:
                ids_keep = ids.contiguous()
                ids_ptr = ctypes.c_void_p(ids_keep.data_ptr())
            else:
                ids_keep = np.ascontiguousarray(ids, dtype=np.int64)
                ids_ptr = _np_ptr(ids_keep)
            if len(ids_keep) != n:
                raise ValueError("ids and rows differ in length")
        if group_codes is not None:
            if on_dev:
                groups_keep = group_codes.contiguous()
                groups_ptr = ctypes.c_void_p(groups_keep.data_ptr())
            else:
                groups_keep = np.ascontiguousarray(group_codes, dtype=np.int32)
                groups_ptr = _np_ptr(groups_keep)
            if len(groups_keep) != n:
                raise ValueError("group_codes and rows differ in length")
        with self._lock:
            first = len(self)
            check(self._lib.hvs_index_add(self._h, ptr, code, n, ids_ptr, groups_ptr, on_dev))
        del keep
        return first, first + n

    def update(self, positions, rows) -> None:
        ptr, code, n, on_dev, keep = self._rows_arg(rows)
        if on_dev:
            raise ValueError("update takes host arrays")
        pos = np.ascontiguousarray(positions, dtype=np.int64)
        if len(pos) != n:
            raise ValueError("positions and rows differ in length")
        with self._lock:
            check(self._lib.hvs_index_update(self._h, _np_ptr(pos), ptr, code, n, 0))

    def set_deleted(self, positions, deleted: bool = True) -> None:
        pos = np.ascontiguousarray(positions, dtype=np.int64)
        with self._lock:
            check(self._lib.hvs_index_set_deleted(self._h, _np_ptr(pos), len(pos), 1 if deleted else 0))

    def fill_synthetic(self, n: int, seed: int = 1234, first_global_row: int = 0, n_groups: int = 0) -> None:
        with self._lock:
            check(self._lib.hvs_index_fill_synthetic(self._h, int(n), int(seed), int(first_global_row), int(n_groups)))

    def fill_clustered(self, n: int, seed: int = 1234, first_global_row: int = 0, n_groups: int = 0, n_clusters: int = 10_000) -> None:
        """Append ``n`` rows of the clustered synthetic corpus (``hvs_index_fill_clustered``)."""
        with self._lock:
            check(self._lib.hvs_index_fill_clustered(self._h, int(n), int(seed), int(first_global_row), int(n_groups), int(n_clusters)))

    def set_attributes(self, first: int, workspace=None, knowledge_base=None, uri=None, private=None) -> None:
        """Attribute columns of rows ``[first, first + n)``: int32 codes of workspaceId / knowledgeBaseId / uri and the
        ``private`` flag, the columns ``hvs_filter`` tests on the GPU (``hvs_index_set_attributes``)."""
        cols = [None if c is None else np.ascontiguousarray(c, dtype=np.int32) for c in (workspace, knowledge_base, uri)]
        priv = None if private is None else np.ascontiguousarray(private, dtype=np.uint8)
        lengths = {len(c) for c in cols + [priv] if c is not None}
        if not lengths:
            return
        if len(lengths) != 1:
            raise ValueError("attribute arrays differ in length")
        with self._lock:
            check(self._lib.hvs_index_set_attributes(self._h, int(first), lengths.pop(), _np_ptr(cols[0]), _np_ptr(cols[1]), _np_ptr(cols[2]),
                                                     _np_ptr(priv)))

    @staticmethod
    def _filter_struct(filter: dict):
        """``filter``: keys group / workspace / knowledge_base (codes), access (True = private rows, False = public rows),
        uris (codes) -- each optional -> (hvs_filter, keepalive)"""
        unknown = set(filter) - {"group", "workspace", "knowledge_base", "access", "uris"}
        if unknown:
            raise ValueError(f"unknown filter keys {sorted(unknown)}")
        uris = filter.get("uris")
        uris = None if uris is None else list(uris)
        # uri IN (): a non-NULL list with zero codes matches nothing, so an empty list still needs an address
        keep = None if uris is None else np.ascontiguousarray(uris or [0], dtype=np.int32)
        access = filter.get("access")
        f = _lib.HvsFilterStruct(int(filter.get("group", -1)), int(filter.get("workspace", -1)), int(filter.get("knowledge_base", -1)),
                                 -1 if access is None else int(bool(access)), None if keep is None else keep.ctypes.data,
                                 0 if uris is None else len(uris))
        return f, keep

    # -- search ----------------------------------------------------------------------------------
    def search(self, queries, k: int = 10, fold: int = 1, group_filter: int = -1, row_mask=None,
               max_distance: float = math.inf, row_begin: int = 0, row_end: int = -1, algo: str = "auto", filter: Optional[dict] = None):
        """Exact cosine top-k.  ``queries``: ``[nq*fold, dim]`` float32 (numpy -> numpy results,
        torch on this GPU -> torch results).  Returns ``(dist [nq,k] f32, ids [nq,k] i64, count [nq] i32)``.
        ``filter`` (see ``_filter_struct``) replaces ``group_filter`` / ``row_mask`` by attribute clauses evaluated
        on the GPU.
        """
        if algo not in ALGO_CODE:
            raise ValueError(f"algo must be one of {sorted(ALGO_CODE)}")
        if filter is not None:
            if row_mask is not None or group_filter >= 0:
                raise ValueError("give either filter or group_filter / row_mask")
            return self._search_filtered(queries, k, fold, filter, max_distance, row_begin, row_end, algo)
        if _is_torch(queries):
            import torch

            q = queries.contiguous().to(torch.float32)
            if q.dim() != 2 or q.shape[1] != self.dim or q.shape[0] % fold:
                raise ValueError(f"queries must be [nq*fold, {self.dim}]")
            nq = q.shape[0] // fold
            dev = q.device
            torch.cuda.current_stream(dev).synchronize()  # the index works on its own stream
            dist = torch.empty((nq, k), dtype=torch.float32, device=dev)
            ids = torch.empty((nq, k), dtype=torch.int64, device=dev)
            cnt = torch.empty((nq,), dtype=torch.int32, device=dev)
            mask_ptr = None
            if row_mask is not None:
                row_mask = row_mask.contiguous()
                mask_ptr = ctypes.c_void_p(row_mask.data_ptr())
            with self._lock:
                check(self._lib.hvs_search(self._h, ctypes.c_void_p(q.data_ptr()), nq, fold, k, group_filter, mask_ptr,
                                           float(max_distance), row_begin, row_end, ctypes.c_void_p(dist.data_ptr()),
                                           ctypes.c_void_p(ids.data_ptr()), ctypes.c_void_p(cnt.data_ptr()), 1,
                                           ALGO_CODE[algo]))
                check(self._lib.hvs_sync(self._h))
            return dist, ids, cnt
        q = np.ascontiguousarray(np.atleast_2d(np.asarray(queries)), dtype=np.float32)
        if q.shape[1] != self.dim or q.shape[0] % fold:
            raise ValueError(f"queries must be [nq*fold, {self.dim}], got {q.shape}")
        nq = q.shape[0] // fold
        dist = np.empty((nq, k), dtype=np.float32)
        ids = np.empty((nq, k), dtype=np.int64)
        cnt = np.empty((nq,), dtype=np.int32)
        mask = None
        if row_mask is not None:
            mask = np.ascontiguousarray(row_mask, dtype=np.uint32)
            if mask.size < (len(self) + 31) // 32:
                raise ValueError("row_mask has fewer bits than the index has rows")
        with self._lock:
            check(self._lib.hvs_search(self._h, _np_ptr(q), nq, fold, k, group_filter, _np_ptr(mask), float(max_distance),
                                       row_begin, row_end, _np_ptr(dist), _np_ptr(ids), _np_ptr(cnt), 0, ALGO_CODE[algo]))
        return dist, ids, cnt

    def _search_filtered(self, queries, k, fold, filter, max_distance, row_begin, row_end, algo):
        f, keep = self._filter_struct(filter)
        if _is_torch(queries):
            import torch

            q = queries.contiguous().to(torch.float32)
            if q.dim() != 2 or q.shape[1] != self.dim or q.shape[0] % fold:
                raise ValueError(f"queries must be [nq*fold, {self.dim}]")
            nq = q.shape[0] // fold
            torch.cuda.current_stream(q.device).synchronize()
            dist = torch.empty((nq, k), dtype=torch.float32, device=q.device)
            ids = torch.empty((nq, k), dtype=torch.int64, device=q.device)
            cnt = torch.empty((nq,), dtype=torch.int32, device=q.device)
            ptrs = [ctypes.c_void_p(t.data_ptr()) for t in (q, dist, ids, cnt)]
            on_dev = 1
        else:
            q = np.ascontiguousarray(np.atleast_2d(np.asarray(queries)), dtype=np.float32)
            if q.shape[1] != self.dim or q.shape[0] % fold:
                raise ValueError(f"queries must be [nq*fold, {self.dim}], got {q.shape}")
            nq = q.shape[0] // fold
            dist = np.empty((nq, k), dtype=np.float32)
            ids = np.empty((nq, k), dtype=np.int64)
            cnt = np.empty((nq,), dtype=np.int32)
            ptrs = [_np_ptr(a) for a in (q, dist, ids, cnt)]
            on_dev = 0
        with self._lock:
            check(self._lib.hvs_search_filtered(self._h, ptrs[0], nq, fold, k, ctypes.byref(f), float(max_distance), row_begin, row_end,
                                                ptrs[1], ptrs[2], ptrs[3], on_dev, ALGO_CODE[algo]))
            if on_dev:
                check(self._lib.hvs_sync(self._h))
        del keep
        return dist, ids, cnt

    def distances(self, queries, row_begin: int = 0, row_end: int = -1) -> np.ndarray:
        """``LEAST_j(row <=> q_j)`` for every row of the range (float32, NaN where undefined)."""
        q = np.ascontiguousarray(np.atleast_2d(np.asarray(queries)), dtype=np.float32)
        if q.shape[1] != self.dim:
            raise ValueError(f"queries must be [m, {self.dim}]")
        n = len(self)
        row_end = n if row_end < 0 or row_end > n else row_end
        out = np.empty((max(0, row_end - row_begin),), dtype=np.float32)
        with self._lock:
            check(self._lib.hvs_distances(self._h, _np_ptr(q), q.shape[0], row_begin, row_end, _np_ptr(out), 0))
        return out

    def range_search(self, queries, max_distance: float, group_filter: int = -1, row_mask=None, row_begin: int = 0,
                     row_end: int = -1, capacity: int | None = None, filter: Optional[dict] = None):
        """Every live row with ``LEAST_j(row <=> q_j) < max_distance`` in ascending row position: the rows
        the reference keeps from its projected distance column
        (cross_language_search/functions.py:330-340).  Returns ``(dist f32, ids i64, total)``; with a
        ``capacity`` smaller than ``total`` only the first ``capacity`` rows come back."""
        q = np.ascontiguousarray(np.atleast_2d(np.asarray(queries)), dtype=np.float32)
        if q.shape[1] != self.dim:
            raise ValueError(f"queries must be [m, {self.dim}]")
        mask = None
        if row_mask is not None:
            mask = np.ascontiguousarray(row_mask, dtype=np.uint32)
            if mask.size < (len(self) + 31) // 32:
                raise ValueError("row_mask has fewer bits than the index has rows")
        total = ctypes.c_int64(0)

        f = keep = None
        if filter is not None:
            if row_mask is not None or group_filter >= 0:
                raise ValueError("give either filter or group_filter / row_mask")
            f, keep = self._filter_struct(filter)

        def call(cap):
            dist = np.empty((cap,), dtype=np.float32)
            ids = np.empty((cap,), dtype=np.int64)
            if f is not None:
                check(self._lib.hvs_range_search_filtered(self._h, _np_ptr(q), q.shape[0], ctypes.byref(f), float(max_distance), row_begin,
                                                          row_end, cap, _np_ptr(dist), _np_ptr(ids), ctypes.byref(total), 0))
                return dist, ids
            check(self._lib.hvs_range_search(self._h, _np_ptr(q), q.shape[0], group_filter, _np_ptr(mask), float(max_distance),
                                             row_begin, row_end, cap, _np_ptr(dist), _np_ptr(ids), ctypes.byref(total), 0))
            return dist, ids

        with self._lock:
            if capacity is None:  # most selections are small: one pass with room for 64k rows, a second only if they did not fit
                capacity = 65536
                dist, ids = call(capacity)
                if total.value > capacity:
                    capacity = total.value
                    dist, ids = call(capacity)
            else:
                capacity = max(int(capacity), 0)  # 0: count only
                dist, ids = call(capacity)
        m = min(capacity, total.value)
        return dist[:m], ids[:m], total.value

    def gather(self, positions) -> np.ndarray:
        """Stored vectors at the given row positions, widened to float32 ``[n, dim]``."""
        pos = np.ascontiguousarray(positions, dtype=np.int64)
        out = np.empty((len(pos), self.dim), dtype=np.float32)
        with self._lock:
            check(self._lib.hvs_gather(self._h, _np_ptr(pos), len(pos), _np_ptr(out), 0))
        return out

    def kmeans_assign(self, centroids, want_sums: bool = True):
        """One E-step (+ M-step partial sums).  numpy in -> numpy out; torch in -> torch out."""
        if _is_torch(centroids):
            import torch

            c = centroids.contiguous().to(torch.float32)
            kc = c.shape[0]
            n = len(self)
            torch.cuda.current_stream(c.device).synchronize()  # the index works on its own stream
            assign = torch.empty((n,), dtype=torch.int32, device=c.device)
            sums = torch.empty((kc, self.dim), dtype=torch.float32, device=c.device) if want_sums else None
            counts = torch.empty((kc,), dtype=torch.int64, device=c.device) if want_sums else None
            with self._lock:
                check(self._lib.hvs_kmeans_assign(self._h, ctypes.c_void_p(c.data_ptr()), kc, ctypes.c_void_p(assign.data_ptr()),
                                                  ctypes.c_void_p(sums.data_ptr()) if want_sums else None,
                                                  ctypes.c_void_p(counts.data_ptr()) if want_sums else None, 1))
                check(self._lib.hvs_sync(self._h))
            return assign, sums, counts
        c = np.ascontiguousarray(centroids, dtype=np.float32)
        if c.ndim != 2 or c.shape[1] != self.dim:
            raise ValueError(f"centroids must be [K, {self.dim}]")
        kc = c.shape[0]
        assign = np.empty((len(self),), dtype=np.int32)
        sums = np.empty((kc, self.dim), dtype=np.float32) if want_sums else None
        counts = np.empty((kc,), dtype=np.int64) if want_sums else None
        with self._lock:
            check(self._lib.hvs_kmeans_assign(self._h, _np_ptr(c), kc, _np_ptr(assign), _np_ptr(sums), _np_ptr(counts), 0))
        return assign, sums, counts

    # -- measurement -----------------------------------------------------------------------------
    def profile(self, on: bool) -> None:
        check(self._lib.hvs_profile_enable(self._h, 1 if on else 0))

    def profile_reset(self) -> None:
        check(self._lib.hvs_profile_reset(self._h))

    def profile_get(self, which: int) -> Tuple[float, int]:
        ms = ctypes.c_double(0.0)
        n = ctypes.c_int64(0)
        check(self._lib.hvs_profile_get(self._h, which, ctypes.byref(ms), ctypes.byref(n)))
        return ms.value, n.value

    def timer_start(self) -> None:
        check(self._lib.hvs_timer_start(self._h))

    def timer_stop(self) -> float:
        ms = ctypes.c_double(0.0)
        check(self._lib.hvs_timer_stop(self._h, ctypes.byref(ms)))
        return ms.value

    # -- tombstones, compaction, snapshots -------------------------------------------------------------
    def deleted_count(self) -> int:
        n = ctypes.c_int64(0)
        check(self._lib.hvs_index_deleted_count(self._h, ctypes.byref(n)))
        return n.value

    def prepare(self) -> bool:
        """Build (or extend) the search-side copy of an fp32 table now instead of on the first batched search; True
        when it is in place.  With it, searches with k <= 64 -- one query at a time on a large table included --
        stream half the bytes and score their candidates exactly afterwards.  No-op (False) for 16-bit tables."""
        ready = ctypes.c_int(0)
        with self._lock:
            check(self._lib.hvs_index_prepare(self._h, ctypes.byref(ready)))
        return bool(ready.value)

    def search_copy(self, allow: Optional[bool] = None) -> int:
        """Device bytes the search-side copy of an fp32 table holds (0: none).  ``allow=False`` frees it and keeps the
        table from ever building one (it costs as much HBM again as the rows); ``allow=True`` lets searches build it."""
        held = ctypes.c_int64(0)
        with self._lock:
            check(self._lib.hvs_index_search_copy(self._h, -1 if allow is None else int(bool(allow)), ctypes.byref(held)))
        return held.value

    def compact(self, renumber_ids: bool = False) -> int:
        """Drop deleted rows and close the gaps (order kept); returns the new row count.  Row
        positions change: position i afterwards is the i-th live row before.  Ids move with their
        rows, or become the new positions when ``renumber_ids`` is set."""
        n = ctypes.c_int64(0)
        with self._lock:
            check(self._lib.hvs_index_compact(self._h, 1 if renumber_ids else 0, ctypes.byref(n)))
        return n.value

    def partition(self, renumber_ids: bool = False) -> np.ndarray:
        """Order the rows by group code (deleted rows are dropped, rows of one code keep their order), so that a search
        with a group filter scans that group's row range and nothing else (``hvs_index_partition``).  Returns, for
        every new row position, the position the row had.  Ids travel with their rows, or become the new positions
        when ``renumber_ids`` is set.  Rows added afterwards are searched with the per-row test until the next call."""
        with self._lock:
            old = np.empty(len(self), dtype=np.int64)
            n = ctypes.c_int64(0)
            check(self._lib.hvs_index_partition(self._h, 1 if renumber_ids else 0, ctypes.byref(n), _np_ptr(old)))
        return old[: n.value]

    def partition_info(self) -> Tuple[int, int]:
        """``(rows at the front of the table that are in group order, distinct group codes among them)``; (0, 0) for a
        table that was never partitioned."""
        rows, groups = ctypes.c_int64(0), ctypes.c_int64(0)
        check(self._lib.hvs_index_partition_info(self._h, ctypes.byref(rows), ctypes.byref(groups)))
        return rows.value, groups.value

    def set_ids(self, first: int, ids) -> None:
        """Overwrite the external ids of rows ``[first, first + len(ids))``."""
        a = np.ascontiguousarray(ids, dtype=np.int64)
        with self._lock:
            check(self._lib.hvs_index_set_ids(self._h, int(first), len(a), _np_ptr(a)))

    def ids(self, first: int = 0, n: Optional[int] = None) -> np.ndarray:
        """External ids of rows ``[first, first + n)``."""
        n = len(self) - first if n is None else int(n)
        out = np.empty(n, dtype=np.int64)
        with self._lock:
            check(self._lib.hvs_index_export(self._h, int(first), n, None, _np_ptr(out), None, None))
        return out

    def export(self, first: int = 0, n: Optional[int] = None):
        """Rows ``[first, first + n)`` as stored -> ``(rows, ids, group_codes, deleted)``; ``rows`` is
        float32 / float16, or uint16 bit patterns for a bf16 index."""
        total = len(self)
        n = total - first if n is None else int(n)
        np_dtype = {"f32": np.float32, "f16": np.float16, "bf16": np.uint16}[self.dtype]
        rows = np.empty((n, self.dim), dtype=np_dtype)
        ids = np.empty(n, dtype=np.int64)
        groups = np.empty(n, dtype=np.int32)
        deleted = np.empty(n, dtype=np.uint8)
        with self._lock:
            check(self._lib.hvs_index_export(self._h, int(first), n, _np_ptr(rows), _np_ptr(ids), _np_ptr(groups), _np_ptr(deleted)))
        return rows, ids, groups, deleted.astype(bool)

    def save(self, path: str) -> None:
        """Write the table to one snapshot file (``snapshot.write_index``)."""
        from .snapshot import write_index

        write_index(self, path)

    @classmethod
    def load(cls, path: str, device: int = 0, stream: Optional[int] = None) -> "HvsIndex":
        """Rebuild a table from a snapshot file: stored values go back bit for bit (no re-rounding),
        inverse norms are recomputed on the device, tombstones are restored."""
        from .snapshot import read_index

        return read_index(path, lambda dim, dtype: cls(dim, dtype, device=device, stream=stream))


def merge_topk(dist, ids, count, k: int, stream: int = 0):
    """Merge ``[nq, P, k]`` per-shard candidate lists (torch tensors on one GPU) into ``[nq, k]``."""
    import torch

    nq, P = int(dist.shape[0]), int(dist.shape[1])
    dev = dist.device
    out_d = torch.empty((nq, k), dtype=torch.float32, device=dev)
    out_i = torch.empty((nq, k), dtype=torch.int64, device=dev)
    out_c = torch.empty((nq,), dtype=torch.int32, device=dev)
    dist, ids, count = dist.contiguous(), ids.contiguous(), count.contiguous()
    stream = stream or torch.cuda.current_stream(dev).cuda_stream
    check(load_library().hvs_merge_topk(dev.index or 0, ctypes.c_void_p(stream), ctypes.c_void_p(dist.data_ptr()),
                                        ctypes.c_void_p(ids.data_ptr()), ctypes.c_void_p(count.data_ptr()), nq, P, k,
                                        ctypes.c_void_p(out_d.data_ptr()), ctypes.c_void_p(out_i.data_ptr()),
                                        ctypes.c_void_p(out_c.data_ptr())))
    return out_d, out_i, out_c


__all__ = ["HvsIndex", "HvsError", "merge_topk"]
