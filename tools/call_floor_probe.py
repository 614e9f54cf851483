"""What one search call costs when the table is tiny (the per-call floor: wrapper, launches, synchronisation), device and host buffers,
and where the Python side of it goes (cProfile).
    python tools/call_floor_probe.py"""
import cProfile
import os
import pstats
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import hirag_prod_amd as hvs  # noqa: E402

d = 768
rng = np.random.default_rng(3)
for n in (1000, 100_000):
    idx = hvs.HvsIndex(d, "f16")
    idx.add(rng.standard_normal((n, d)).astype(np.float32), group_codes=(np.arange(n) % 8).astype(np.int32))
    idx.partition()
    for fold in (1, 3):
        qd = torch.randn((fold, d), device="cuda")
        qh = qd.cpu().numpy()
        for name, q in (("device", qd), ("host", qh)):
            f = lambda: idx.search(q, k=10, fold=fold, group_filter=3)  # noqa: E731
            for _ in range(20):
                f()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(1000):
                f()
            torch.cuda.synchronize()
            print(f"n={n} fold={fold} {name} buffers: {(time.perf_counter() - t0) * 1e3:.1f} us per call", flush=True)
    if n == 1000:
        pr = cProfile.Profile()
        pr.enable()
        for _ in range(2000):
            idx.search(qd, k=10, fold=3, group_filter=3)
        pr.disable()
        pstats.Stats(pr).sort_stats("tottime").print_stats(12)
    idx.close()
