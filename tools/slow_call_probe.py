"""Host-buffer calls with very little GPU work (8 192 x 1536 f16, one embedding: 25 us of kernels per call) run at one of two speeds.
This probe shows which: the same call timed cold, after a burst of denser GPU work, and after a pause.
    python tools/slow_call_probe.py"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import hirag_prod_amd as hvs  # noqa: E402

d, n = 1536, 8192
idx = hvs.HvsIndex(d, "f16")
idx.fill_synthetic(n, seed=1)
rng = np.random.default_rng(0)
qh = rng.standard_normal((1, d)).astype(np.float32)
qd = torch.from_numpy(qh).cuda()


def timed(name, f, reps=300):
    for _ in range(20):
        f()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        f()
    torch.cuda.synchronize()
    print(f"{name}: {(time.perf_counter() - t0) / reps * 1e6:.1f} us per call", flush=True)


host = lambda: idx.search(qh, k=10, max_distance=0.8)  # noqa: E731
dev = lambda: idx.search(qd, k=10, max_distance=0.8)  # noqa: E731
timed("host buffers, first thing in the process", host)
timed("host buffers, again", host)
timed("device buffers", dev)
timed("host buffers, right after", host)
timed("host buffers, again", host)
time.sleep(2.0)
timed("host buffers, after a pause of 2 s", host)
timed("host buffers, again", host)
big = torch.randn((8192, 8192), device="cuda")
for _ in range(50):
    big @ big
torch.cuda.synchronize()
timed("host buffers, after 50 large matrix products", host)
time.sleep(2.0)
timed("device buffers, after a pause of 2 s", dev)
