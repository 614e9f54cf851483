"""A group filter on a table with an ordered part and a tail: time of one call with host buffers, kernel times.
    python tools/tail_probe.py"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import hirag_prod_amd as hvs  # noqa: E402
from hirag_prod_amd import _lib  # noqa: E402

rng = np.random.default_rng(1)
n, d = 100_000, 768
rows = rng.standard_normal((n, d)).astype(np.float32)
groups = (np.arange(n) % 8).astype(np.int32)
idx = hvs.HvsIndex(d, "f16")
idx.add(rows[:80_000], group_codes=groups[:80_000])
idx.partition()
idx.add(rows[80_000:], ids=np.arange(80_000, n), group_codes=groups[80_000:])
print("partition", idx.partition_info())
for fold, k in ((3, 10), (1, 10)):
    q = rng.standard_normal((fold, d)).astype(np.float32)
    f = lambda: idx.search(q, k=k, fold=fold, group_filter=3, max_distance=1.5)  # noqa: E731
    for _ in range(5):
        f()
    t0 = time.perf_counter()
    for _ in range(200):
        f()
    dt = (time.perf_counter() - t0) / 200
    idx.profile(True)
    idx.profile_reset()
    for _ in range(20):
        f()
    s, ns = idx.profile_get(_lib.HVS_PROF_SCAN)
    m, nm = idx.profile_get(_lib.HVS_PROF_MERGE)
    idx.profile(False)
    print(f"fold={fold}: {dt * 1e6:.1f} us per call; scan {s / 20 * 1e3:.1f} us in {ns / 20:.0f} launches, merges {m / 20 * 1e3:.1f} us in {nm / 20:.0f} groups", flush=True)
