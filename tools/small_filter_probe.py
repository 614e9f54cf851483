"""Batched filtered search on small row ranges: dense copy of the live rows (default) against the in-epilogue test
(HVS_NO_COMPACT=1).   python tools/small_filter_probe.py"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import hirag_prod_amd as hvs  # noqa: E402

d = 768
for n in (20_000, 100_000, 400_000, 1_600_000):
    idx = hvs.HvsIndex(d, "f16")
    idx.fill_synthetic(n, seed=3, n_groups=8)
    q = torch.randn((75, d), device="cuda", generator=torch.Generator(device="cuda").manual_seed(1))
    out = []
    for env in (None, "1"):
        if env:
            os.environ["HVS_NO_COMPACT"] = env
        else:
            os.environ.pop("HVS_NO_COMPACT", None)
        f = lambda: idx.search(q, k=10, fold=3, group_filter=3, algo="mfma")  # noqa: E731
        for _ in range(5):
            f()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(50):
            f()
        torch.cuda.synchronize()
        out.append((time.perf_counter() - t0) / 50 * 1e3)
    os.environ.pop("HVS_NO_COMPACT", None)
    print(f"n={n}: dense copy {out[0]:.3f} ms, in-epilogue test {out[1]:.3f} ms", flush=True)
    idx.close()
