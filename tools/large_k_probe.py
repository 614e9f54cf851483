"""k beyond the matrix pipe's lists (k > 256): what the scan fallback costs.  python tools/large_k_probe.py [rows] [dim] [dtype]"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import hirag_prod_amd as hvs  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 2_000_000
d = int(sys.argv[2]) if len(sys.argv) > 2 else 1536
dtype = sys.argv[3] if len(sys.argv) > 3 else "f16"
dev = torch.device("cuda", 0)
idx = hvs.HvsIndex(d, dtype)
idx.fill_synthetic(n, seed=1234)
gen = torch.Generator(device=dev).manual_seed(4321)
q = torch.randn((256, d), device=dev, dtype=torch.float32, generator=gen)
for k in (256, 500, 1024):
    for nq in (1, 16, 64, 256):
        qq = q[:nq].contiguous()
        f = lambda: idx.search(qq, k=k)  # noqa: E731
        for _ in range(2):
            f()
        ts = []
        for _ in range(5):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            f()
            torch.cuda.synchronize()
            ts.append((time.perf_counter() - t0) * 1e3)
        print(f"{n} x {d} {dtype} k={k} nq={nq}: {np.median(ts):.3f} ms per call, {np.median(ts) / nq:.3f} ms per query", flush=True)
