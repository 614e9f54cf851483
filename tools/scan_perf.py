"""Quick device-time probe of the scan (GEMV) path: achieved GB/s per configuration.

    python tools/scan_perf.py [--rows 1000000] [--iters 20]
"""

import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import hirag_prod_amd as hvs  # noqa: E402
from hirag_prod_amd import _lib  # noqa: E402


def run(dtype, dim, n, nq, fold, k, iters, group_filter=-1, n_groups=0):
    idx = hvs.HvsIndex(dim, dtype)
    idx.fill_synthetic(n, seed=1234, n_groups=n_groups)
    rng = np.random.default_rng(4321)
    q = rng.standard_normal((nq * fold, dim)).astype(np.float32)
    for _ in range(3):
        idx.search(q, k=k, fold=fold, group_filter=group_filter, algo="scan")
    idx.profile(True)
    idx.profile_reset()
    for _ in range(iters):
        idx.search(q, k=k, fold=fold, group_filter=group_filter, algo="scan")
    ms, launches = idx.profile_get(_lib.HVS_PROF_SCAN)
    mms, ml = idx.profile_get(_lib.HVS_PROF_MERGE)
    idx.profile(False)
    esz = 4 if dtype == "f32" else 2
    per = ms / launches
    frac_rows = 1.0 if group_filter < 0 else 1.0 / max(1, n_groups)
    gbs = n * dim * esz * frac_rows / (per * 1e-3) / 1e9
    print(f"{dtype:5s} n={n:>9d} d={dim:5d} nq={nq} fold={fold} k={k:4d} gf={group_filter:3d}: scan {per:8.3f} ms  "
          f"{gbs:8.1f} GB/s ({gbs / 8000 * 100:5.1f}% of 8 TB/s)  merge {mms / max(ml, 1):7.3f} ms", flush=True)
    idx.close()


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=1_000_000)
    ap.add_argument("--iters", type=int, default=20)
    a = ap.parse_args()
    run("f32", 1536, a.rows, 1, 1, 10, a.iters)
    run("f32", 1536, a.rows, 1, 3, 10, a.iters)
    run("f32", 1536, a.rows, 4, 1, 10, a.iters)
    run("f32", 1536, a.rows, 1, 1, 100, a.iters)
    run("f16", 1536, a.rows, 1, 1, 10, a.iters)
    run("f16", 1536, a.rows, 1, 4, 10, a.iters)
    run("bf16", 768, a.rows * 4, 1, 1, 100, a.iters)
    run("bf16", 768, a.rows * 4, 1, 1, 100, a.iters, group_filter=3, n_groups=64)
    run("f32", 1536, a.rows * 8, 1, 1, 10, max(3, a.iters // 4))
