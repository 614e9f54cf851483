"""One filtered batched search on the group-ordered cfg5 shard, repeated (for kernel traces and the timing hooks).
    python tools/range_probe.py [nq] [k] [reps]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import hirag_prod_amd as hvs  # noqa: E402

nq = int(sys.argv[1]) if len(sys.argv) > 1 else 256
k = int(sys.argv[2]) if len(sys.argv) > 2 else 100
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 20
n, d = int(os.environ.get("PROBE_ROWS", 12_500_000)), int(os.environ.get("PROBE_DIM", 768))
dev = torch.device("cuda", 0)
idx = hvs.HvsIndex(d, os.environ.get("PROBE_DTYPE", "bf16"))
idx.fill_synthetic(n, seed=1234, first_global_row=25_000_000, n_groups=64)
idx.partition()
gen = torch.Generator(device=dev).manual_seed(4321)
q = torch.randn((nq, d), device=dev, dtype=torch.float32, generator=gen)
algo = os.environ.get("PROBE_ALGO", "scan" if nq == 1 else "mfma")
for _ in range(reps):
    idx.search(q, k=k, algo=algo, group_filter=3)
torch.cuda.synchronize()
print("done", flush=True)
