"""The production calls on the cfg5 shard in group order, many times over (for `rocprofv3 --hip-trace --stats`: which runtime calls the host side spends its time in).
    python tools/host_api_probe.py [single|batch] [calls]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import hirag_prod_amd as hvs  # noqa: E402

what = sys.argv[1] if len(sys.argv) > 1 else "batch"
calls = int(sys.argv[2]) if len(sys.argv) > 2 else 200
n, d = 12_500_000, 768
idx = hvs.HvsIndex(d, "bf16")
idx.fill_synthetic(n, seed=1234, first_global_row=25_000_000, n_groups=64)
idx.partition()
nq = 1 if what == "single" else 256
q = torch.randn((nq, d), device="cuda", generator=torch.Generator(device="cuda").manual_seed(1))
for _ in range(10):
    idx.search(q, k=100, group_filter=3)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(calls):
    idx.search(q, k=100, group_filter=3)
    torch.cuda.synchronize()
print(f"{what}: {(time.perf_counter() - t0) / calls * 1e3:.4f} ms per call", flush=True)
