"""``HvsGraph``: the knowledge graph of one knowledge base as a CSR adjacency in HBM, and
personalised PageRank on it.

A thin object wrapper over the graph part of the C ABI (``include/hvs.h``).  The arithmetic the
reference gets from ``networkx.pagerank`` (``src/hirag_prod/storage/pgvector.py:432-434``) runs in
the HIP kernels of ``csrc/hvs_graph.hip``; nothing here computes a score on the CPU.
"""

from __future__ import annotations

import ctypes
import threading
from typing import Optional, Tuple

import numpy as np

from ._lib import HvsError, check, load_library

# networkx.pagerank's defaults (the reference passes only alpha and the personalisation)
PAGERANK_MAX_ITER = 100
PAGERANK_TOL = 1.0e-6


class PowerIterationFailedConvergence(HvsError):
    """The power iteration did not reach ``N * tol`` within ``max_iter`` iterations (networkx raises
    an exception of this name in the reference's call)."""


def _ptr(a: Optional[np.ndarray]):
    return None if a is None else a.ctypes.data_as(ctypes.c_void_p)


def pack_mask(flags) -> np.ndarray:
    """bool [n] -> uint32 words, bit ``j & 31`` of word ``j >> 5``."""
    bits = np.packbits(np.asarray(flags, dtype=bool), bitorder="little")
    pad = (-len(bits)) % 4
    if pad:
        bits = np.concatenate([bits, np.zeros(pad, dtype=np.uint8)])
    return bits.view(np.uint32) if len(bits) else np.zeros(1, dtype=np.uint32)


class HvsGraph:
    """Undirected simple graph over nodes ``0 .. n_nodes-1`` built from directed rows ``(src, dst)``."""

    def __init__(self, n_nodes: int, src, dst, is_chunk, device: int = 0, stream: Optional[int] = None):
        src = np.ascontiguousarray(src, dtype=np.int32)
        dst = np.ascontiguousarray(dst, dtype=np.int32)
        if src.shape != dst.shape or src.ndim != 1:
            raise ValueError("src and dst must be 1-D arrays of the same length")
        flags = np.asarray(is_chunk, dtype=bool)
        if flags.shape != (n_nodes,):
            raise ValueError(f"is_chunk must hold one flag per node ({n_nodes}), got {flags.shape}")
        self._lib = load_library()
        self._lock = threading.Lock()  # calls on one graph must not overlap
        self._h = ctypes.c_void_p()
        self.device = int(device)
        mask = pack_mask(flags)
        check(self._lib.hvs_graph_create(self.device, ctypes.c_void_p(stream) if stream else None, int(n_nodes), _ptr(src), _ptr(dst),
                                         len(src), _ptr(mask), 0, ctypes.byref(self._h)))
        n, nnz, dang = ctypes.c_int64(0), ctypes.c_int64(0), ctypes.c_int64(0)
        check(self._lib.hvs_graph_info(self._h, ctypes.byref(n), ctypes.byref(nnz), ctypes.byref(dang)))
        self.n_nodes, self.nnz, self.n_dangling = n.value, nnz.value, dang.value

    def close(self) -> None:
        if getattr(self, "_h", None) is not None and self._h.value:
            self._lib.hvs_graph_destroy(self._h)
            self._h = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def csr(self) -> Tuple[np.ndarray, np.ndarray]:
        """``(row_ptr int64 [n+1], col int32 [nnz])`` of the stored adjacency."""
        row_ptr = np.empty(self.n_nodes + 1, dtype=np.int64)
        col = np.empty(max(self.nnz, 1), dtype=np.int32)
        with self._lock:
            check(self._lib.hvs_graph_export(self._h, _ptr(row_ptr), _ptr(col)))
        return row_ptr, col[: self.nnz]

    def pagerank(self, seed_nodes, seed_weights, alpha: float = 0.85, max_iter: int = PAGERANK_MAX_ITER,
                 tol: float = PAGERANK_TOL) -> Tuple[int, float]:
        """Runs the power iteration; the scores stay on the device.  -> ``(iterations, last L1 change)``."""
        nodes = np.ascontiguousarray(seed_nodes, dtype=np.int32)
        weights = np.ascontiguousarray(seed_weights, dtype=np.float64)
        if nodes.shape != weights.shape or nodes.ndim != 1 or len(nodes) == 0:
            raise ValueError("seed_nodes and seed_weights must be non-empty 1-D arrays of the same length")
        iters, err = ctypes.c_int(0), ctypes.c_double(0.0)
        with self._lock:
            rc = self._lib.hvs_graph_pagerank(self._h, _ptr(nodes), _ptr(weights), len(nodes), float(alpha), int(max_iter), float(tol),
                                              ctypes.byref(iters), ctypes.byref(err))
        if rc == 3:
            raise PowerIterationFailedConvergence(self._lib.hvs_last_error().decode())
        check(rc)
        return iters.value, err.value

    def scores(self, first: int = 0, n: Optional[int] = None) -> np.ndarray:
        n = self.n_nodes - first if n is None else n
        out = np.empty(n, dtype=np.float64)
        with self._lock:
            check(self._lib.hvs_graph_scores(self._h, int(first), int(n), _ptr(out), 0))
        return out

    def top_chunks(self, k: int) -> Tuple[np.ndarray, np.ndarray]:
        """``(nodes int32 [m], scores float64 [m])``, m = min(k, chunk nodes): descending score, equal
        scores in ascending node number."""
        nodes = np.empty(k, dtype=np.int32)
        scores = np.empty(k, dtype=np.float64)
        count = ctypes.c_int(0)
        with self._lock:
            check(self._lib.hvs_graph_top_chunks(self._h, int(k), _ptr(nodes), _ptr(scores), ctypes.byref(count)))
        return nodes[: count.value], scores[: count.value]
