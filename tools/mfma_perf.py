"""Quick device-time probe of the batched MFMA search and the k-means step.

    python tools/mfma_perf.py [--rows 1000000] [--nq 4096] [--iters 3]
"""

import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import hirag_prod_amd as hvs  # noqa: E402
from hirag_prod_amd import _lib  # noqa: E402

if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=1_000_000)
    ap.add_argument("--dim", type=int, default=1536)
    ap.add_argument("--nq", type=int, default=4096)
    ap.add_argument("--k", type=int, default=10)
    ap.add_argument("--iters", type=int, default=3)
    ap.add_argument("--kmeans", type=int, default=0, help="number of centroids (0 = skip)")
    ap.add_argument("--dtype", default="f32")
    a = ap.parse_args()
    import torch

    idx = hvs.HvsIndex(a.dim, a.dtype)
    peak = 157.3 if a.dtype == "f32" else 2500.0
    idx.fill_synthetic(a.rows, seed=1234)
    q = torch.randn((a.nq, a.dim), device="cuda", generator=torch.Generator("cuda").manual_seed(4321))
    idx.search(q, k=a.k, algo="mfma")
    idx.profile(True)
    idx.profile_reset()
    for _ in range(a.iters):
        idx.search(q, k=a.k, algo="mfma")
    ms, n = idx.profile_get(_lib.HVS_PROF_MFMA_SEARCH)
    mm, nm = idx.profile_get(_lib.HVS_PROF_MERGE)
    per = ms / a.iters  # every launch of a search counts (a sample pass ahead of the main one, a second list length, ...)
    flops = 2.0 * a.nq * a.rows * a.dim
    print(f"mfma search {a.dtype} rows={a.rows} dim={a.dim} nq={a.nq} k={a.k}: {per:.2f} ms/batch  {flops / per / 1e9:.1f} TFLOP/s "
          f"({flops / per / 1e9 / peak * 100:.1f}% of {peak}; {n / a.iters:g} launches per search)  merge {mm / a.iters:.3f} ms  -> {a.nq / (per + mm / a.iters) * 1e3:.0f} QPS", flush=True)
    if a.kmeans:
        c = torch.randn((a.kmeans, a.dim), device="cuda")
        idx.profile_reset()
        idx.kmeans_assign(c)
        idx.profile_reset()
        for _ in range(a.iters):
            idx.kmeans_assign(c)
        ms, n = idx.profile_get(_lib.HVS_PROF_KMEANS_ASSIGN)
        mu, nu = idx.profile_get(_lib.HVS_PROF_KMEANS_UPDATE)
        flops = 2.0 * a.kmeans * a.rows * a.dim
        print(f"kmeans K={a.kmeans}: assign {ms / n:.2f} ms ({flops / (ms / n) / 1e9:.1f} TFLOP/s)  update {mu / nu:.2f} ms", flush=True)
    idx.close()
