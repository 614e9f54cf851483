"""Batch-size sweep on a group range of the cfg5 shard in group order, and on the whole shard.
    python tools/range_sweep.py [k]   (PROBE_NQ=8,32,... picks the batch sizes; libhvs environment switches apply)"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import hirag_prod_amd as hvs  # noqa: E402
from hirag_prod_amd import _lib  # noqa: E402

k = int(sys.argv[1]) if len(sys.argv) > 1 else 100
n, d = int(os.environ.get("PROBE_ROWS", 12_500_000)), int(os.environ.get("PROBE_DIM", 768))
dtype = os.environ.get("PROBE_DTYPE", "bf16")
dev = torch.device("cuda", 0)
idx = hvs.HvsIndex(d, dtype)
idx.fill_synthetic(n, seed=1234, first_global_row=25_000_000, n_groups=int(os.environ.get("PROBE_GROUPS", 64)))
idx.partition()
gen = torch.Generator(device=dev).manual_seed(4321)
q = torch.randn((1024, d), device=dev, dtype=torch.float32, generator=gen)


def timed(fn, iters=20, warm=3):
    for _ in range(warm):
        fn()
    out = []
    for _ in range(iters):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        fn()
        torch.cuda.synchronize()
        out.append((time.perf_counter() - t0) * 1e3)
    return float(np.median(out))


def kern(fn, iters=8):
    idx.profile(True)
    out = []
    for _ in range(iters):
        idx.profile_reset()
        fn()
        out.append(idx.profile_get(_lib.HVS_PROF_MFMA_SEARCH)[0] + idx.profile_get(_lib.HVS_PROF_SCAN)[0])
    idx.profile(False)
    return float(np.median(out))


for nq in [int(x) for x in os.environ.get("PROBE_NQ", "8,32,64,128,256,512,1024").split(",")]:
    qq = q[:nq].contiguous()
    for label, kw in (("range", {"group_filter": 3}), ("whole", {})):
        if label == "whole" and os.environ.get("PROBE_NO_WHOLE"):
            continue
        f = lambda: idx.search(qq, k=k, algo="mfma", **kw)  # noqa: E731
        ref = idx.search(qq[:4], k=k, algo="scan", **kw)
        got = f()
        ok = bool((got[1][:4] == ref[1]).all().item())
        print(f"{label} nq={nq} k={k}: call {timed(f):.3f} ms  kernels {kern(f):.3f} ms  ids equal to the scan's: {ok}", flush=True)
