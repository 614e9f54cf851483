"""Batch-size sweep over a whole table (no filter): python tools/whole_sweep.py rows dim dtype k  (PROBE_NQ, libhvs switches apply)"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import hirag_prod_amd as hvs  # noqa: E402
from hirag_prod_amd import _lib  # noqa: E402

n, d, dtype, k = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], int(sys.argv[4])
dev = torch.device("cuda", 0)
idx = hvs.HvsIndex(d, dtype)
idx.fill_synthetic(n, seed=1234)
if dtype == "f32":
    idx.prepare()
gen = torch.Generator(device=dev).manual_seed(4321)
q = torch.randn((4096, d), device=dev, dtype=torch.float32, generator=gen)
for nq in [int(x) for x in os.environ.get("PROBE_NQ", "8,32,64,128").split(",")]:
    qq = q[:nq].contiguous()
    f = lambda: idx.search(qq, k=k, algo="mfma")  # noqa: E731
    for _ in range(3):
        f()
    ts = []
    for _ in range(10):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        f()
        torch.cuda.synchronize()
        ts.append((time.perf_counter() - t0) * 1e3)
    print(f"{n} x {d} {dtype} nq={nq} k={k}: call {np.median(ts):.3f} ms", flush=True)
