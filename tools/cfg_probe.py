"""Device-time probes of the BASELINE.json configurations that bench.py does not headline.

    python tools/cfg_probe.py cfg4      # k-means K=4096 over 10M x 1536 fp32, one Lloyd iteration
    python tools/cfg_probe.py cfg5      # 12.5M x 768 bf16 shard (1/8 of 100M), top-100, workspace filter
"""

import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import hirag_prod_amd as hvs  # noqa: E402
from hirag_prod_amd import _lib  # noqa: E402


def cfg4(rows=10_000_000, dim=1536, K=4096, iters=2):
    import torch

    idx = hvs.HvsIndex(dim, "f32")
    idx.fill_synthetic(rows, seed=1234)
    c = torch.from_numpy(idx.gather(np.arange(K))).cuda()  # centroids = first K rows
    idx.kmeans_assign(c)
    idx.profile(True)
    idx.profile_reset()
    t0 = time.perf_counter()
    for _ in range(iters):
        assign, sums, counts = idx.kmeans_assign(c)
        c2 = torch.where((counts > 0).unsqueeze(1), sums / counts.clamp(min=1).unsqueeze(1).float(), c)  # empty clusters keep theirs
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) / iters
    a_ms, a_n = idx.profile_get(_lib.HVS_PROF_KMEANS_ASSIGN)
    _, flagged = idx.profile_get(_lib.HVS_PROF_REDO)
    u_ms, u_n = idx.profile_get(_lib.HVS_PROF_KMEANS_UPDATE)
    flops = 2.0 * K * rows * dim
    print(f"cfg4 kmeans rows={rows} dim={dim} K={K}: assign {a_ms / a_n:.1f} ms = {flops / (a_ms / a_n) / 1e9:.1f} TFLOP/s "
          f"({flops / (a_ms / a_n) / 1e9 / 157.3 * 100:.1f}% of the 157.3 fp32 MFMA peak; the split operands run on the 16-bit pipe), update {u_ms / u_n:.1f} ms "
          f"({rows * dim * 4 / (u_ms / u_n) / 1e6:.0f} GB/s of added bytes), iteration wall {wall * 1e3:.1f} ms; rows sent through all three products per iteration: {flagged / max(a_n, 1):.0f} "
          f"({flagged / max(a_n, 1) / rows * 100:.2f} %)", flush=True)
    assert int(counts.sum().item()) == rows
    idx.close()


def cfg5(rows=12_500_000, dim=768, k=100, iters=10):
    import torch

    idx = hvs.HvsIndex(dim, "bf16")
    idx.fill_synthetic(rows, seed=1234, n_groups=64)
    q = torch.randn((256, dim), device="cuda", generator=torch.Generator("cuda").manual_seed(4321))
    for label, kw, nq, algo in (
        ("single query, no filter", {}, 1, "scan"),
        ("single query, workspace filter 1/64", {"group_filter": 3}, 1, "scan"),
        ("batch 256, no filter", {}, 256, "mfma"),
        ("batch 256, workspace filter 1/64", {"group_filter": 3}, 256, "mfma"),
    ):
        qq = q[:nq].contiguous()
        idx.search(qq, k=k, algo=algo, **kw)
        idx.profile(True)
        idx.profile_reset()
        t0 = time.perf_counter()
        for _ in range(iters):
            idx.search(qq, k=k, algo=algo, **kw)
        wall = (time.perf_counter() - t0) / iters
        which = _lib.HVS_PROF_SCAN if algo == "scan" else _lib.HVS_PROF_MFMA_SEARCH
        ms, n = idx.profile_get(which)
        mm, mn = idx.profile_get(_lib.HVS_PROF_MERGE)
        idx.profile(False)
        frac = 1.0 / 64 if kw else 1.0
        gb = rows * dim * 2 * frac / 1e9
        print(f"cfg5 {label}: kernel {ms / n:.3f} ms ({gb / (ms / n) * 1e3:.0f} GB/s of algorithmic bytes), merge {mm / mn:.3f} ms, "
              f"wall {wall * 1e3:.3f} ms -> {nq / wall:.0f} QPS", flush=True)
    idx.close()


if __name__ == "__main__":
    which = sys.argv[1] if len(sys.argv) > 1 else "cfg5"
    {"cfg4": cfg4, "cfg5": cfg5}[which]()
