"""A batch of 256 filtered queries on the cfg5 shard in group order with a tail of 5 % behind it (for kernel traces)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import hirag_prod_amd as hvs  # noqa: E402

n, d = 12_500_000, 768
idx = hvs.HvsIndex(d, "bf16")
idx.reserve(n + n // 20)
idx.fill_synthetic(n, seed=1234, first_global_row=25_000_000, n_groups=64)
idx.partition()
idx.fill_synthetic(n // 20, seed=1234, first_global_row=25_000_000 + n, n_groups=64)
q = torch.randn((256, d), device="cuda", generator=torch.Generator(device="cuda").manual_seed(1))
for _ in range(int(sys.argv[1]) if len(sys.argv) > 1 else 20):
    idx.search(q, k=100, algo="mfma", group_filter=3)
torch.cuda.synchronize()
print("done")
