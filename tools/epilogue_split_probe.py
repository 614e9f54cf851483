"""Where the batched kernel's selection time goes: the same launch with the usual bound, and with a bound no score can
pass (max_distance = 1e-9: the epilogue's fixed part only -- level-1 tests, accumulator reset, tile-end barrier).
    python tools/epilogue_split_probe.py [rows] [k]"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import hirag_prod_amd as hvs  # noqa: E402
from hirag_prod_amd import _lib  # noqa: E402

rows = int(sys.argv[1]) if len(sys.argv) > 1 else 4_000_000
k = int(sys.argv[2]) if len(sys.argv) > 2 else 10
d, nq = 1536, 4096
idx = hvs.HvsIndex(d, "f16")
idx.fill_synthetic(rows, seed=1234)
q = torch.randn((nq, d), device="cuda", dtype=torch.float32, generator=torch.Generator(device="cuda").manual_seed(1))
for label, thr in (("usual bound", float("inf")), ("no score passes", 1e-9), ("usual bound", float("inf")), ("no score passes", 1e-9)):
    f = lambda: idx.search(q, k=k, algo="mfma", max_distance=thr)  # noqa: E731
    for _ in range(2):
        f()
    idx.profile(True)
    ts = []
    for _ in range(5):
        idx.profile_reset()
        f()
        ts.append(idx.profile_get(_lib.HVS_PROF_MFMA_SEARCH)[0])
    idx.profile(False)
    ms = float(np.median(ts))
    print(f"{label:16s}: kernel {ms:.2f} ms  {2.0 * nq * rows * d / ms / 1e9:.1f} TFLOP/s", flush=True)
