"""cfg5 shard (12.5M x 768 bf16, 64 workspaces) in group order: time of the filtered searches, single and batch 256.

    python tools/partition_probe.py [rows] [dim] [dtype] [k]
Environment switches of libhvs (HVS_TILE, HVS_PREPASS, ...) apply; PROBE_NQ=1,256 picks the batch sizes."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import hirag_prod_amd as hvs  # noqa: E402
from hirag_prod_amd import _lib  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 12_500_000
d = int(sys.argv[2]) if len(sys.argv) > 2 else 768
dtype = sys.argv[3] if len(sys.argv) > 3 else "bf16"
k = int(sys.argv[4]) if len(sys.argv) > 4 else 100
dev = torch.device("cuda", 0)
idx = hvs.HvsIndex(d, dtype)
idx.fill_synthetic(n, seed=1234, first_global_row=25_000_000, n_groups=64)
gen = torch.Generator(device=dev).manual_seed(4321)
q = torch.randn((256, d), device=dev, dtype=torch.float32, generator=gen)


def timed(fn, iters=30, warm=3):
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


def kern(fn, which, iters=10):
    idx.profile(True)
    out = []
    for _ in range(iters):
        idx.profile_reset()
        fn()
        out.append(idx.profile_get(which)[0])
    idx.profile(False)
    return float(np.median(out))


sizes = [int(x) for x in os.environ.get("PROBE_NQ", "1,256").split(",")]
ref = {}
for nq in sizes:
    qq = q[:nq].contiguous()
    algo = "scan" if nq == 1 else "mfma"
    ref[nq] = idx.search(qq, k=k, algo=algo, group_filter=3)
    print(f"unpartitioned nq={nq}: call {timed(lambda: idx.search(qq, k=k, algo=algo, group_filter=3)):.3f} ms", flush=True)
t0 = time.perf_counter()
idx.partition()
print(f"partition: {time.perf_counter() - t0:.3f} s, info {idx.partition_info()}", flush=True)
for nq in sizes:
    qq = q[:nq].contiguous()
    for algo in (("scan", "auto") if nq == 1 else ("mfma", "auto", "scan")):
        f = lambda: idx.search(qq, k=k, algo=algo, group_filter=3)  # noqa: E731
        dist, ids, cnt = f()
        same = bool((ids == ref[nq][1]).all().item()) and bool((dist == ref[nq][0]).all().item())
        which = _lib.HVS_PROF_SCAN if algo == "scan" else _lib.HVS_PROF_MFMA_SEARCH
        print(f"partitioned nq={nq} algo={algo}: call {timed(f):.3f} ms  kernel {kern(f, which):.3f} ms  merge {kern(f, _lib.HVS_PROF_MERGE):.3f} ms  "
              f"same as before: {same}", flush=True)
