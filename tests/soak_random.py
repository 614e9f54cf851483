"""Not collected by pytest: a timed soak of randomly drawn searches against the float64 oracle on one GPU.

    python tests/soak_random.py [seconds] [seed]

Every round draws a table (row type, dimension, rows, group codes, tombstones), optionally puts it in group order, appends
a tail, and sends random batches (embeddings, fold, k, threshold, group filter) through every route the shape allows;
each answer is held to the oracle by tests/parity.check_topk.  Prints one line per round; the first mismatch raises.
"""

import math
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import hirag_prod_amd as hvs  # noqa: E402
from oracle import pgvector_oracle as po  # noqa: E402
from tests import parity  # noqa: E402


def one_round(rng, label):
    dtype = str(rng.choice(["f32", "f16", "bf16"]))
    dim = int(rng.choice([1, 7, 31, 64, 65, 96, 128, 200, 384, 768, 1000]))
    n = int(rng.choice([40, 300, 2049, 9000, 40_000, 150_000, 600_000, 1_200_000], p=[0.1, 0.15, 0.15, 0.15, 0.15, 0.1, 0.1, 0.1]))
    if n > 150_000:
        dim = min(dim, 128)  # the oracle's float64 distance matrix is the slow part
    n_groups = int(rng.choice([1, 3, 17]))
    rows = rng.standard_normal((n, dim)).astype(np.float32)
    if n > 10:
        rows[3] = 0.0
        rows[7] = rows[2]
    groups = rng.integers(0, n_groups, n).astype(np.int32)
    idx = hvs.HvsIndex(dim, dtype)
    first = n if rng.random() < 0.5 else max(1, int(n * rng.uniform(0.3, 0.95)))
    idx.add(rows[:first], ids=np.arange(first), group_codes=groups[:first])
    dead = np.zeros(n, bool)
    if rng.random() < 0.6:
        gone = rng.permutation(first)[: first // 12]
        dead[gone] = True
        idx.set_deleted(gone, True)
    ordered = rng.random() < 0.6
    if ordered:
        idx.partition()
    if first < n:
        idx.add(rows[first:], ids=np.arange(first, n), group_codes=groups[first:])
    checks = 0
    for _ in range(3):
        fold = int(rng.choice([1, 1, 2, 3, 4]))
        nq = int(rng.choice([1, 2, 5, 24, 70, 130, 300]))
        if n * nq * fold > 40_000_000:
            nq = max(1, 40_000_000 // (n * fold))
        k = int(rng.choice([1, 10, 40, 100, 200, 500]))
        thr = math.inf if rng.random() < 0.5 else float(rng.uniform(0.7, 1.1))
        gf = -1 if rng.random() < 0.4 else int(rng.integers(0, n_groups))
        q = rng.standard_normal((nq * fold, dim)).astype(np.float32)
        keep = ~dead & ((groups == gf) if gf >= 0 else True)
        dm = parity.distance_matrix(rows, q, dtype)
        routes = ["scan", "auto"]
        if fold <= 4 and k <= 768:
            routes += ["mfma"] + (["mfma_f32", "mfma_split"] if dtype == "f32" and nq * fold <= 140 else [])
        for algo in routes:
            if algo == "scan" and nq > 40:
                continue
            dist, ids, cnt = idx.search(q, k=k, fold=fold, group_filter=gf, max_distance=thr, algo=algo)
            for i in range(nq):
                full = po.least_fold(dm[i * fold : (i + 1) * fold])
                parity.check_topk(dist[i], ids[i], cnt[i], full, k, threshold=None if math.isinf(thr) else thr, keep=keep,
                                  label=f"{label} {dtype} {algo} dim={dim} n={n} first={first} ordered={ordered} nq={nq} fold={fold} k={k} gf={gf} thr={thr} q={i}")
                checks += 1
    # range search: every live row of the group below the bound, in ascending position
    if n <= 150_000:
        fold = int(rng.choice([1, 2]))
        q = rng.standard_normal((fold, dim)).astype(np.float32)
        gf = -1 if rng.random() < 0.5 else int(rng.integers(0, n_groups))
        bound = float(rng.uniform(0.6, 1.0))
        full = po.least_fold(parity.distance_matrix(rows, q, dtype))
        keep = ~dead & ((groups == gf) if gf >= 0 else True)
        rd, ri, total = idx.range_search(q, bound, group_filter=gf)
        sure = {int(i) for i in np.flatnonzero(keep & (full < bound - 2e-6))}
        maybe = {int(i) for i in np.flatnonzero(keep & (full < bound + 2e-6))}
        got = set(ri.tolist())
        assert sure <= got <= maybe and total == len(ri), f"{label} range search {dtype} dim={dim} n={n} gf={gf} bound={bound}: {len(got)} rows, {len(sure)}..{len(maybe)} expected"
        checks += 1
    idx.close()
    return f"{dtype} dim={dim} n={n} groups={n_groups} ordered={ordered} tail={n - first} deleted={int(dead.sum())}: {checks} answers"


def main():
    seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else int(time.time())
    rng = np.random.default_rng(seed)
    t0, rounds = time.time(), 0
    print(f"soak: seed {seed}, {seconds:.0f} s", flush=True)
    while time.time() - t0 < seconds:
        rounds += 1
        print(f"round {rounds}: " + one_round(rng, f"seed {seed} round {rounds}"), flush=True)
    print(f"soak ok: {rounds} rounds in {time.time() - t0:.0f} s", flush=True)


if __name__ == "__main__":
    main()
