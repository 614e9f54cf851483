"""Durations of the first 600 host-buffer calls of a process, one by one (8 192 x 1536 f16, one embedding).
    python tools/first_calls_probe.py [nogc]      nogc: with Python's cyclic garbage collector switched off"""
import gc
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import hirag_prod_amd as hvs  # noqa: E402

if "nogc" in sys.argv[1:]:
    gc.disable()
d, n = 1536, 8192
idx = hvs.HvsIndex(d, "f16")
idx.fill_synthetic(n, seed=1)
qh = np.random.default_rng(0).standard_normal((1, d)).astype(np.float32)
t = []
for _ in range(600):
    t0 = time.perf_counter()
    idx.search(qh, k=10, max_distance=0.8)
    t.append((time.perf_counter() - t0) * 1e6)
t = np.array(t)
print("garbage collector", "off" if not gc.isenabled() else "on", "- collections so far per generation:", [g["collections"] for g in gc.get_stats()])
for lo in range(0, 600, 50):
    w = t[lo:lo + 50]
    print(f"calls {lo:3d}-{lo + 49:3d}: median {np.median(w):7.1f} us, max {w.max():9.1f} us, mean {w.mean():7.1f} us", flush=True)
