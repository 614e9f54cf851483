"""Is the 129-256-query regime bound by power?  The same launch on random rows and on all-zero rows (zero operands
switch far fewer gates in the matrix pipe: whatever time they save is clock the chip was holding back)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import hirag_prod_amd as hvs  # noqa: E402


def main():
    import torch

    rows, dim = 4_000_000, 1536
    for label in ("random", "zeros"):
        idx = hvs.HvsIndex(dim, "f16")
        if label == "random":
            idx.fill_synthetic(rows, seed=1234)
        else:
            z = torch.zeros((500_000, dim), dtype=torch.float32, device="cuda")
            for _ in range(rows // 500_000):
                idx.add(z)
            del z
        for nq in (64, 128, 256, 512, 4096):
            q = torch.randn((nq, dim), device="cuda", generator=torch.Generator("cuda").manual_seed(4321))
            if label == "zeros":
                q.zero_()
            idx.search(q, k=10, algo="mfma")
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(5):
                idx.search(q, k=10, algo="mfma")
            torch.cuda.synchronize()
            print(f"{label}: {rows} x {dim} f16, {nq} queries: {(time.perf_counter() - t0) / 5 * 1e3:.3f} ms per search", flush=True)
        idx.close()


if __name__ == "__main__":
    main()
