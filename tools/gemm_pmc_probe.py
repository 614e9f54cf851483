"""The vendor GEMM on the shape of the 16-bit batched search (4096 queries x 2M rows x 1536, f16, product written out),
for counter passes beside tools/mfma_perf.py:  python tools/gemm_pmc_probe.py [rows]"""
import sys
import time

import torch

rows = int(sys.argv[1]) if len(sys.argv) > 1 else 2_000_000
a = torch.randn((4096, 1536), device="cuda", dtype=torch.float16)
b = torch.randn((rows, 1536), device="cuda", dtype=torch.float16)
c = torch.empty((4096, rows), device="cuda", dtype=torch.float16)
for _ in range(2):
    torch.matmul(a, b.T, out=c)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(5):
    torch.matmul(a, b.T, out=c)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / 5
print(f"vendor GEMM f16 4096 x {rows} x 1536: {dt * 1e3:.2f} ms  {2.0 * 4096 * rows * 1536 / dt / 1e12:.1f} TFLOP/s", flush=True)
