"""ctypes binding of ``csrc/libhvs.so`` (the C ABI declared in ``include/hvs.h``).

There is no CPU implementation behind this module: if the shared library is
missing it raises, and if no GPU is visible every compute entry point fails
with the library's own error message.
"""

from __future__ import annotations

import ctypes
import os
import sys
from ctypes import POINTER, c_char_p, c_double, c_float, c_int, c_int32, c_int64, c_uint32, c_uint64, c_void_p

HERE = os.path.dirname(os.path.abspath(__file__))
# HVS_LIB: another build of the same library (experiments: a build with the kernels' timing hooks, `make HOOKS=1`)
LIB_PATH = os.environ.get("HVS_LIB") or os.path.join(HERE, "csrc", "libhvs.so")

HVS_F32, HVS_F16, HVS_BF16 = 0, 1, 2
HVS_ALGO_AUTO, HVS_ALGO_SCAN, HVS_ALGO_MFMA, HVS_ALGO_MFMA_F32, HVS_ALGO_MFMA_SPLIT = 0, 1, 2, 3, 4
HVS_PROF_SCAN, HVS_PROF_MFMA_SEARCH, HVS_PROF_KMEANS_ASSIGN, HVS_PROF_KMEANS_UPDATE, HVS_PROF_MERGE, HVS_PROF_REDO = range(6)
HVS_MAX_K = 1024
HVS_MAX_FOLD = 8
HVS_MAX_DIST_FOLD = 128

DTYPE_CODE = {"f32": HVS_F32, "f16": HVS_F16, "bf16": HVS_BF16}
ALGO_CODE = {"auto": HVS_ALGO_AUTO, "scan": HVS_ALGO_SCAN, "mfma": HVS_ALGO_MFMA, "mfma_f32": HVS_ALGO_MFMA_F32,
             "mfma_split": HVS_ALGO_MFMA_SPLIT}

HVS_UNIQUE_ID_BYTES = 128
HVS_REDUCE_I64 = 100
# the host-staged transport's exchange functions (hvs_allgather_fn / hvs_allreduce_fn)
ALLGATHER_FN = ctypes.CFUNCTYPE(c_int, c_void_p, c_void_p, c_void_p, c_uint64)
ALLREDUCE_FN = ctypes.CFUNCTYPE(c_int, c_void_p, c_void_p, c_uint64, c_int)

HVS_MAX_FILTER_URIS = 8192


class HvsFilterStruct(ctypes.Structure):
    """``hvs_filter``: the attribute clauses of a search, evaluated on the GPU."""

    _fields_ = [("group", c_int32), ("workspace", c_int32), ("knowledge_base", c_int32), ("access", c_int32),
                ("uris", c_void_p), ("n_uris", c_int32)]


# name -> (restype, argtypes): every symbol include/hvs.h declares
SIGNATURES = {
    "hvs_last_error": (c_char_p, []),
    "hvs_version": (c_int, []),
    "hvs_device_count": (c_int, [POINTER(c_int)]),
    "hvs_index_create": (c_int, [POINTER(c_void_p), c_int, c_int, c_int, c_void_p]),
    "hvs_index_destroy": (c_int, [c_void_p]),
    "hvs_index_reserve": (c_int, [c_void_p, c_int64]),
    "hvs_index_size": (c_int, [c_void_p, POINTER(c_int64)]),
    "hvs_index_info": (c_int, [c_void_p, POINTER(c_int), POINTER(c_int), POINTER(c_int), POINTER(c_int64)]),
    "hvs_sync": (c_int, [c_void_p]),
    "hvs_index_add": (c_int, [c_void_p, c_void_p, c_int, c_int64, c_void_p, c_void_p, c_int]),
    "hvs_index_update": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int64, c_int]),
    "hvs_index_set_deleted": (c_int, [c_void_p, c_void_p, c_int64, c_int]),
    "hvs_index_fill_synthetic": (c_int, [c_void_p, c_int64, c_uint64, c_int64, c_int32]),
    "hvs_index_fill_clustered": (c_int, [c_void_p, c_int64, c_uint64, c_int64, c_int32, c_int64]),
    "hvs_search": (
        c_int,
        [c_void_p, c_void_p, c_int, c_int, c_int, c_int32, c_void_p, c_double, c_int64, c_int64,
         c_void_p, c_void_p, c_void_p, c_int, c_int],
    ),
    "hvs_search_noalloc": (
        c_int,
        [c_void_p, c_void_p, c_int, c_int, c_int, c_int32, c_void_p, c_double, c_int64, c_int64, c_void_p, c_void_p, c_void_p],
    ),
    "hvs_index_set_attributes": (c_int, [c_void_p, c_int64, c_int64, c_void_p, c_void_p, c_void_p, c_void_p]),
    "hvs_search_filtered": (
        c_int,
        [c_void_p, c_void_p, c_int, c_int, c_int, POINTER(HvsFilterStruct), c_double, c_int64, c_int64, c_void_p, c_void_p, c_void_p, c_int, c_int],
    ),
    "hvs_range_search_filtered": (c_int, [c_void_p, c_void_p, c_int, POINTER(HvsFilterStruct), c_double, c_int64, c_int64, c_int64, c_void_p,
                                          c_void_p, POINTER(c_int64), c_int]),
    "hvs_comm_unique_id": (c_int, [c_void_p, c_int]),
    "hvs_comm_init": (c_int, [POINTER(c_void_p), c_int, c_int, c_char_p, c_int]),
    "hvs_comm_init_host": (c_int, [POINTER(c_void_p), c_int, c_int, ALLGATHER_FN, ALLREDUCE_FN, c_void_p]),
    "hvs_comm_destroy": (c_int, [c_void_p]),
    "hvs_comm_info": (c_int, [c_void_p, POINTER(c_int), POINTER(c_int), POINTER(c_int)]),
    "hvs_search_sharded": (
        c_int,
        [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int32, c_void_p, c_double, c_void_p, c_void_p, c_void_p, c_int, c_int],
    ),
    "hvs_allreduce_centroids": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int]),
    "hvs_distances": (c_int, [c_void_p, c_void_p, c_int, c_int64, c_int64, c_void_p, c_int]),
    "hvs_gather": (c_int, [c_void_p, c_void_p, c_int64, c_void_p, c_int]),
    "hvs_merge_topk": (
        c_int,
        [c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p],
    ),
    "hvs_index_deleted_count": (c_int, [c_void_p, POINTER(c_int64)]),
    "hvs_index_compact": (c_int, [c_void_p, c_int, POINTER(c_int64)]),
    "hvs_index_partition": (c_int, [c_void_p, c_int, POINTER(c_int64), c_void_p]),
    "hvs_index_partition_info": (c_int, [c_void_p, POINTER(c_int64), POINTER(c_int64)]),
    "hvs_index_set_ids": (c_int, [c_void_p, c_int64, c_int64, c_void_p]),
    "hvs_index_prepare": (c_int, [c_void_p, POINTER(c_int)]),
    "hvs_index_search_copy": (c_int, [c_void_p, c_int, POINTER(c_int64)]),
    "hvs_index_export": (c_int, [c_void_p, c_int64, c_int64, c_void_p, c_void_p, c_void_p, c_void_p]),
    "hvs_range_search": (c_int, [c_void_p, c_void_p, c_int, ctypes.c_int32, c_void_p, c_double, c_int64, c_int64, c_int64, c_void_p,
                                 c_void_p, POINTER(c_int64), c_int]),
    "hvs_graph_create": (c_int, [c_int, c_void_p, c_int64, c_void_p, c_void_p, c_int64, c_void_p, c_int, POINTER(c_void_p)]),
    "hvs_graph_destroy": (c_int, [c_void_p]),
    "hvs_graph_info": (c_int, [c_void_p, POINTER(c_int64), POINTER(c_int64), POINTER(c_int64)]),
    "hvs_graph_export": (c_int, [c_void_p, c_void_p, c_void_p]),
    "hvs_graph_pagerank": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_double, c_int, c_double, POINTER(c_int), POINTER(c_double)]),
    "hvs_graph_scores": (c_int, [c_void_p, c_int64, c_int64, c_void_p, c_int]),
    "hvs_graph_top_chunks": (c_int, [c_void_p, c_int, c_void_p, c_void_p, POINTER(c_int)]),
    "hvs_cosine_matrix": (c_int, [c_int, c_void_p, c_void_p, c_int, c_void_p, c_int, c_int, c_void_p, c_int]),
    "hvs_kmeans_assign": (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_int]),
    "hvs_profile_enable": (c_int, [c_void_p, c_int]),
    "hvs_profile_get": (c_int, [c_void_p, c_int, POINTER(c_double), POINTER(c_int64)]),
    "hvs_profile_reset": (c_int, [c_void_p]),
    "hvs_timer_start": (c_int, [c_void_p]),
    "hvs_timer_stop": (c_int, [c_void_p, POINTER(c_double)]),
}

_lib = None


class HvsError(RuntimeError):
    """A non-zero status from libhvs.so; the message is ``hvs_last_error()``."""


def load_library() -> ctypes.CDLL:
    """Open ``csrc/libhvs.so`` and declare every entry point; raises if it was not built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise HvsError(
            f"{LIB_PATH} not found: build it with `make -C {os.path.dirname(LIB_PATH)}` "
            "(or __graft_entry__.build()); there is no CPU fallback"
        )
    # PyTorch's wheel carries its own HIP and HSA runtimes.  A process that uses both must load torch's first: libhvs
    # then binds to the HSA runtime already in the process, whereas torch imported after libhvs has touched the GPU
    # brings up a second HSA runtime, which finds no device ("No HIP GPUs are available").
    if "torch" not in sys.modules:
        try:
            import torch  # noqa: F401
        except Exception:  # noqa: BLE001 -- no torch, or a torch that does not load here: the library then brings its own runtime
            pass
    lib = ctypes.CDLL(LIB_PATH)
    for name, (restype, argtypes) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the library lacks a declared symbol
        fn.restype = restype
        fn.argtypes = argtypes
    _lib = lib
    return lib


def check(rc: int) -> None:
    if rc != 0:
        msg = load_library().hvs_last_error()
        raise HvsError(msg.decode("utf-8", "replace") if msg else f"libhvs status {rc}")


def device_count() -> int:
    n = c_int(0)
    check(load_library().hvs_device_count(ctypes.byref(n)))
    return n.value
