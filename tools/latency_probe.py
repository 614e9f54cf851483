"""End-to-end latency of one query on small and large tables (host buffers in, host buffers out), next to
the device time of its kernels: what a caller of `HvsIndex.search` waits for.

    python tools/latency_probe.py
"""

import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

if __name__ == "__main__":
    import hirag_prod_amd as hvs
    from hirag_prod_amd import _lib

    d = 1536
    rng = np.random.default_rng(0)
    for n in (8192, 65536, 1_000_000):
        idx = hvs.HvsIndex(d, "f16")
        idx.fill_synthetic(n, seed=1)
        for fold in (1, 3):
            q = rng.standard_normal((fold, d)).astype(np.float32)
            for _ in range(20):
                idx.search(q, k=10, fold=fold, max_distance=0.8)
            idx.profile(True)
            idx.profile_reset()
            reps = 200
            t0 = time.perf_counter()
            for _ in range(reps):
                idx.search(q, k=10, fold=fold, max_distance=0.8)
            wall = (time.perf_counter() - t0) / reps * 1e6
            sms, sn = idx.profile_get(_lib.HVS_PROF_SCAN)
            mms, mn = idx.profile_get(_lib.HVS_PROF_MERGE)
            idx.profile(False)
            t0 = time.perf_counter()
            for _ in range(reps):
                idx.search(q, k=10, fold=fold, max_distance=0.8)
            wall_np = (time.perf_counter() - t0) / reps * 1e6
            print(f"n={n:8d} fold={fold}: {wall_np:7.1f} us per call ({wall:7.1f} with event timing on); scan kernel {sms / sn * 1e3:6.1f} us, "
                  f"merge {mms / max(mn, 1) * 1e3:6.1f} us", flush=True)
        idx.close()
