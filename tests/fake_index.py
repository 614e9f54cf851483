"""Test double for ``HvsIndex`` built on the oracle, so the store's HOST logic can run on CPU.

Only tests construct it (through ``HipVectorStore(index_factory=...)``); the product's default
factory is ``HvsIndex`` and needs the GPU.
"""

import functools
import math
import threading

import numpy as np

from oracle import pgvector_oracle as po


def _serialised(fn):
    """calls on one index do not overlap (HvsIndex holds a lock of its own for the same reason)"""

    @functools.wraps(fn)
    def wrapper(self, *a, **kw):
        with self._lock:
            return fn(self, *a, **kw)

    return wrapper


class OracleIndex:
    def __init__(self, dim, dtype, device=0):
        self._lock = threading.RLock()
        self.dim, self.dtype = dim, dtype
        self.rows = np.zeros((0, dim), dtype=np.float32)
        self.groups = np.zeros(0, dtype=np.int32)
        self.deleted = np.zeros(0, dtype=bool)
        self.attrs = np.zeros((0, 4), dtype=np.int32)  # workspace, knowledge base, uri, private (-1: not set yet, matches no clause)
        self.ids_arr = np.zeros(0, dtype=np.int64)  # external ids (position at insertion unless given)
        self.part_n = 0  # rows at the front that are in group order
        self.fail_next_add = None  # an exception the next add() raises (what a failed hipMalloc looks like to the store)

    def __len__(self):
        return self.rows.shape[0]

    @_serialised
    def add(self, rows, ids=None, group_codes=None):
        if self.fail_next_add is not None:
            exc, self.fail_next_add = self.fail_next_add, None
            raise exc
        first = len(self)
        rows = np.asarray(rows)
        if rows.dtype == np.uint16:  # bf16 bit patterns, as export() hands them out
            rows = (rows.astype(np.uint32) << 16).view(np.float32)
        rows = rows.astype(np.float32)
        self.rows = np.concatenate([self.rows, rows])
        g = np.zeros(rows.shape[0], np.int32) if group_codes is None else np.asarray(group_codes, np.int32)
        self.groups = np.concatenate([self.groups, g])
        self.deleted = np.concatenate([self.deleted, np.zeros(rows.shape[0], bool)])
        self.attrs = np.concatenate([self.attrs, np.full((rows.shape[0], 4), -1, np.int32)])
        new_ids = np.arange(first, first + rows.shape[0], dtype=np.int64) if ids is None else np.asarray(ids, np.int64)
        self.ids_arr = np.concatenate([self.ids_arr, new_ids])
        return first, len(self)

    @_serialised
    def set_attributes(self, first, workspace=None, knowledge_base=None, uri=None, private=None):
        for col, values in enumerate((workspace, knowledge_base, uri, private)):
            if values is not None:
                v = np.asarray(values).astype(np.int32)
                self.attrs[first : first + len(v), col] = v

    @_serialised
    def set_deleted(self, positions, deleted=True):
        self.deleted[np.asarray(positions, dtype=np.int64)] = deleted

    def _keep(self, group_filter, row_mask, filter=None):
        keep = ~self.deleted
        if filter is not None:
            group_filter = filter.get("group", -1)
            for col, key in enumerate(("workspace", "knowledge_base")):
                if filter.get(key, -1) >= 0:
                    keep &= self.attrs[:, col] == filter[key]
            if filter.get("uris") is not None:
                keep &= np.isin(self.attrs[:, 2], np.asarray(list(filter["uris"]), dtype=np.int32))
            if filter.get("access") is not None:
                keep &= (self.attrs[:, 3] >= 0) & ((self.attrs[:, 3] != 0) == bool(filter["access"]))
        if group_filter >= 0:
            keep &= self.groups == group_filter
        if row_mask is not None:
            bits = np.unpackbits(np.asarray(row_mask, dtype=np.uint32).view(np.uint8), bitorder="little")
            keep &= bits[: len(self)].astype(bool)
        return keep

    @_serialised
    def search(self, queries, k=10, fold=1, group_filter=-1, row_mask=None, max_distance=math.inf, filter=None, **kw):
        q = np.atleast_2d(np.asarray(queries, dtype=np.float32))
        nq = q.shape[0] // fold
        dist = np.full((nq, k), np.inf, dtype=np.float32)
        ids = np.full((nq, k), -1, dtype=np.int64)
        cnt = np.zeros(nq, dtype=np.int32)
        keep = self._keep(group_filter, row_mask, filter)
        for i in range(nq):
            idx, d = po.query(self.rows, q[i * fold : (i + 1) * fold], k=k,
                              threshold=None if math.isinf(max_distance) else max_distance, keep=keep,
                              storage=self.dtype)
            cnt[i] = len(idx)
            ids[i, : len(idx)] = self.ids_arr[idx]
            dist[i, : len(idx)] = d
        return dist, ids, cnt

    @_serialised
    def distances(self, queries, row_begin=0, row_end=-1):
        q = np.atleast_2d(np.asarray(queries, dtype=np.float32))
        rows_s = po.round_storage(self.rows, self.dtype)
        q_s = po.round_storage(q, self.dtype)
        d = np.stack([po.cosine_distance(rows_s, q_s[i]) for i in range(q.shape[0])])
        return po.least_fold(d).astype(np.float32)

    @_serialised
    def range_search(self, queries, max_distance, group_filter=-1, row_mask=None, row_begin=0, row_end=-1, capacity=None, filter=None):
        d = self.distances(queries)
        with np.errstate(invalid="ignore"):
            hit = self._keep(group_filter, row_mask, filter) & (d.astype(np.float64) < max_distance)
        n = len(self)
        hit[:row_begin] = False
        hit[n if row_end < 0 else row_end :] = False
        pos = np.flatnonzero(hit).astype(np.int64)
        total = len(pos)
        if capacity is not None:
            pos = pos[:capacity]
        return d[pos], self.ids_arr[pos], total

    @_serialised
    def gather(self, positions):
        return po.round_storage(self.rows[np.asarray(positions, dtype=np.int64)], self.dtype).astype(np.float32)

    def deleted_count(self):
        return int(self.deleted.sum())

    @_serialised
    def compact(self, renumber_ids=False):
        keep = ~self.deleted
        self.part_n = int(keep[: self.part_n].sum())
        self.rows, self.groups, self.deleted = self.rows[keep], self.groups[keep], np.zeros(int(keep.sum()), bool)
        self.attrs = self.attrs[keep]
        self.ids_arr = np.arange(len(self), dtype=np.int64) if renumber_ids else self.ids_arr[keep]
        return len(self)

    @_serialised
    def partition(self, renumber_ids=False):
        live = np.flatnonzero(~self.deleted)
        order = live[np.argsort(self.groups[live], kind="stable")]
        self.rows, self.groups, self.attrs = self.rows[order], self.groups[order], self.attrs[order]
        self.deleted = np.zeros(len(order), bool)
        self.ids_arr = np.arange(len(order), dtype=np.int64) if renumber_ids else self.ids_arr[order]
        self.part_n = len(order)
        return order.astype(np.int64)

    def partition_info(self):
        return self.part_n, len(np.unique(self.groups[: self.part_n]))

    @_serialised
    def set_ids(self, first, ids):
        ids = np.asarray(ids, np.int64)
        self.ids_arr[first : first + len(ids)] = ids

    def ids(self, first=0, n=None):
        n = len(self) - first if n is None else n
        return self.ids_arr[first : first + n].copy()

    @_serialised
    def export(self, first=0, n=None):
        n = len(self) - first if n is None else n
        sl = slice(first, first + n)
        stored = po.round_storage(self.rows[sl], self.dtype)
        if self.dtype == "bf16":
            stored = (stored.astype(np.float32).view(np.uint32) >> 16).astype(np.uint16)
        elif self.dtype == "f16":
            stored = stored.astype(np.float16)
        else:
            stored = stored.astype(np.float32)
        return stored, self.ids_arr[sl].copy(), self.groups[sl].copy(), self.deleted[sl].copy()

    def close(self):
        pass
