"""What the vendor GEMM (torch.matmul -> hipBLASLt / rocBLAS) sustains on this GPU, as a yardstick for the search
kernels' fraction of the nominal MFMA peak.  Plain GEMMs only: no selection, the product is written out.

    python tools/gemm_ref_probe.py
"""

import time

import torch


def bench(m, n, k, dtype, reps=10):
    a = torch.randn((m, k), device="cuda", dtype=dtype)
    b = torch.randn((n, k), device="cuda", dtype=dtype)
    for _ in range(3):
        c = a @ b.T
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        c = a @ b.T
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    del c
    return 2.0 * m * n * k / dt / 1e12, dt * 1e3


if __name__ == "__main__":
    for dtype, peak in ((torch.float16, 2500.0), (torch.bfloat16, 2500.0), (torch.float32, 157.3)):
        for m, n, k in ((8192, 8192, 8192), (4096, 262144, 1536), (4096, 65536, 1536)):
            if dtype == torch.float32 and n > 65536 and m * n * 4 > 3e9:
                continue
            tf, ms = bench(m, n, k, dtype)
            print(f"{str(dtype):16s} M={m:6d} N={n:7d} K={k:5d}: {ms:8.2f} ms  {tf:7.1f} TFLOP/s  ({tf / peak * 100:5.1f} % of {peak})", flush=True)
