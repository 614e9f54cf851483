"""Does the operand data change the 16-bit MFMA search kernel's time?  Random, all-zero and all-one rows and
queries (2M x 1536 f16, 4096 queries).  Zeros toggle no multiplier inputs: the difference to random data is
the share of the time the chip's power management takes.

    python tools/mfma_data_probe.py
"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import hirag_prod_amd as hvs
from hirag_prod_amd import _lib
n, d, nq = 2_000_000, 1536, 4096
for kind in ("random", "zeros", "ones"):
    idx = hvs.HvsIndex(d, "f16")
    if kind == "random":
        idx.fill_synthetic(n, seed=1)
        q = torch.randn((nq, d), device="cuda")
    else:
        v = 0.0 if kind == "zeros" else 1.0
        block = torch.full((250_000, d), v, dtype=torch.float16, device="cuda")
        for _ in range(n // 250_000):
            idx.add(block)
        q = torch.full((nq, d), v, device="cuda")
    idx.search(q, k=10, algo="mfma")
    idx.profile(True); idx.profile_reset()
    for _ in range(3):
        idx.search(q, k=10, algo="mfma")
    ms, c = idx.profile_get(_lib.HVS_PROF_MFMA_SEARCH)
    print(kind, f"{ms / c:.2f} ms", f"{2.0 * nq * n * d / (ms / c) / 1e9:.0f} TFLOP/s", flush=True)
    idx.close()
