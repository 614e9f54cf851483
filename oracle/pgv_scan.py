"""ctypes binding of ``oracle/pgv_scan.c`` (TEST INFRASTRUCTURE ONLY).

``build()`` compiles the C file with the Makefile beside it; ``load()`` opens
the result.  Used by the tests as a second, independently written checker and
by ``bench.py`` as the timed CPU baseline (kind "port").
"""

from __future__ import annotations

import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "libpgv_oracle.so")
_lib = None

ELEM_KIND = {"f32": 0, "f16": 1, "bf16": 2}


def build(force: bool = False) -> str:
    src = os.path.join(_HERE, "pgv_scan.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s"] + (["-B"] if force else []))
    return _SO


def load():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        lib = ctypes.CDLL(_SO)
        lib.pgv_cosine_distance_f32.restype = ctypes.c_double
        lib.pgv_cosine_distance_f32.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
        lib.pgv_scan_topk.restype = ctypes.c_int
        lib.pgv_scan_topk.argtypes = [
            ctypes.c_void_p, ctypes.c_int, ctypes.c_int64, ctypes.c_int64, ctypes.c_int,
            ctypes.c_void_p, ctypes.c_int,
            ctypes.c_void_p, ctypes.c_int32, ctypes.c_void_p,
            ctypes.c_double, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p,
        ]
        lib.pgv_kmeans_assign.restype = None
        lib.pgv_kmeans_assign.argtypes = [
            ctypes.c_void_p, ctypes.c_int64, ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p,
        ]
        _lib = lib
    return _lib


def _ptr(a):
    return None if a is None else a.ctypes.data_as(ctypes.c_void_p)


def scan_topk(rows, queries, k, threshold=np.inf, group_codes=None, group_filter=-1, keep=None,
              storage="f32", row_begin=0, row_end=None):
    """``rows``: float32 ``[n, d]`` (f32) or uint16 bit patterns (f16 / bf16)."""
    lib = load()
    rows = np.ascontiguousarray(rows)
    if storage == "f32":
        assert rows.dtype == np.float32
    else:
        assert rows.dtype == np.uint16
    n, d = rows.shape
    queries = np.ascontiguousarray(np.atleast_2d(queries), dtype=np.float32)
    assert queries.shape[1] == d
    if group_codes is not None:
        group_codes = np.ascontiguousarray(group_codes, dtype=np.int32)
    if keep is not None:
        keep = np.ascontiguousarray(keep, dtype=np.uint8)
    row_end = n if row_end is None else row_end
    out_idx = np.empty(max(k, 1), dtype=np.int64)
    out_dist = np.empty(max(k, 1), dtype=np.float64)
    cnt = lib.pgv_scan_topk(_ptr(rows), ELEM_KIND[storage], row_begin, row_end, d, _ptr(queries), queries.shape[0],
                            _ptr(group_codes), int(group_filter), _ptr(keep), float(threshold), int(k),
                            _ptr(out_idx), _ptr(out_dist))
    return out_idx[:cnt].copy(), out_dist[:cnt].copy()


def kmeans_assign(rows, centroids):
    lib = load()
    rows = np.ascontiguousarray(rows, dtype=np.float32)
    centroids = np.ascontiguousarray(centroids, dtype=np.float32)
    out = np.empty(rows.shape[0], dtype=np.int32)
    lib.pgv_kmeans_assign(_ptr(rows), rows.shape[0], rows.shape[1], _ptr(centroids), centroids.shape[0], _ptr(out))
    return out
