"""A few batched searches of one shape (default: the cfg5 shard, 12.5M x 768 bf16, top-100, 256 queries) for kernel traces
and counter passes; ROWS, DIM, NQ, K, DTYPE, ALGO, ITERS come from the environment."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import hirag_prod_amd as hvs  # noqa: E402


def main():
    import torch

    rows = int(os.environ.get("ROWS", 12_500_000))
    dim = int(os.environ.get("DIM", 768))
    nq = int(os.environ.get("NQ", 256))
    k = int(os.environ.get("K", 100))
    dtype = os.environ.get("DTYPE", "bf16")
    algo = os.environ.get("ALGO", "mfma")
    iters = int(os.environ.get("ITERS", 5))
    idx = hvs.HvsIndex(dim, dtype)
    idx.fill_synthetic(rows, seed=1234, n_groups=64)
    q = torch.randn((nq, dim), device="cuda", generator=torch.Generator("cuda").manual_seed(4321))
    idx.search(q, k=k, algo=algo)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        idx.search(q, k=k, algo=algo)
    torch.cuda.synchronize()
    print(f"{rows} x {dim} {dtype}, {nq} queries, k={k}: {(time.perf_counter() - t0) / iters * 1e3:.3f} ms per search ({algo})", flush=True)
    idx.close()


if __name__ == "__main__":
    main()
