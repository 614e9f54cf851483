"""Batch-size sweep of the search paths against the HBM floor (one pass over the rows).

    python tools/batch_sweep.py [--rows 4000000] [--dim 1536]
"""

import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=4_000_000)
    ap.add_argument("--dim", type=int, default=1536)
    ap.add_argument("--dtypes", default="f16,f32")
    a = ap.parse_args()
    import torch

    import hirag_prod_amd as hvs
    from hirag_prod_amd import _lib

    n, d = a.rows, a.dim
    for dtype in a.dtypes.split(","):
        idx = hvs.HvsIndex(d, dtype)
        idx.fill_synthetic(n, seed=1)
        floor = n * d * (4 if dtype == "f32" else 2) / 8e12 * 1e3
        for nq in (1, 4, 8, 16, 32, 64, 128, 256, 512, 1024):
            q = torch.randn((nq, d), device="cuda")
            for algo in (("scan", "mfma") if 4 <= nq <= 32 else (("scan",) if nq < 4 else ("mfma",))):
                idx.search(q, k=10, algo=algo)
                idx.profile(True)
                idx.profile_reset()
                for _ in range(3):
                    idx.search(q, k=10, algo=algo)
                ms, _ = idx.profile_get(_lib.HVS_PROF_SCAN if algo == "scan" else _lib.HVS_PROF_MFMA_SEARCH)
                idx.profile(False)
                per = ms / 3
                print(f"{dtype} nq={nq:5d} {algo:5s}: {per:7.2f} ms/batch  ({per / floor:5.2f} x the {floor:.2f} ms HBM floor; "
                      f"{2.0 * nq * n * d / per / 1e9:7.1f} TFLOP/s; {nq / per * 1e3:9.0f} queries/s)", flush=True)
        idx.close()
