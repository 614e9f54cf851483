"""Personalised PageRank on a synthetic knowledge graph of production shape: timing of the build, of one
query (power iteration + top chunks) and the bandwidth of the iteration kernels against their
algorithmic bytes (12 B per stored adjacency entry: 4 B column + 8 B gathered score; 60 B per node).

    python tools/pagerank_perf.py [--chunks 10000000] [--entities 2000000] [--per-chunk 6] [--relations 10000000] [--cpu]
"""

import argparse
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--chunks", type=int, default=10_000_000)
    ap.add_argument("--entities", type=int, default=2_000_000)
    ap.add_argument("--per-chunk", type=int, default=6)
    ap.add_argument("--relations", type=int, default=10_000_000)
    ap.add_argument("--alpha", type=float, default=0.5)
    ap.add_argument("--repeat", type=int, default=5)
    a = ap.parse_args()
    import hirag_prod_amd as hvs

    rng = np.random.default_rng(1)
    n = a.chunks + a.entities
    # entity popularity is heavy-tailed: a few entities occur in very many chunks
    pop = (rng.zipf(1.3, a.chunks * a.per_chunk) - 1) % a.entities
    src = np.concatenate([np.repeat(np.arange(a.chunks, dtype=np.int32), a.per_chunk),
                          a.chunks + rng.integers(0, a.entities, a.relations, dtype=np.int32)])
    dst = np.concatenate([(a.chunks + pop).astype(np.int32), a.chunks + rng.integers(0, a.entities, a.relations, dtype=np.int32)])
    is_chunk = np.arange(n) < a.chunks
    t0 = time.perf_counter()
    g = hvs.HvsGraph(n, src, dst, is_chunk)
    build_s = time.perf_counter() - t0
    seeds = np.concatenate([rng.permutation(a.chunks)[:9], a.chunks + rng.permutation(a.entities)[:30]])
    weights = rng.random(len(seeds)) + 0.05
    iters, err = g.pagerank(seeds, weights, alpha=a.alpha)  # warm-up
    g.top_chunks(10)
    t_pr, t_top = [], []
    for _ in range(a.repeat):
        t0 = time.perf_counter()
        iters, err = g.pagerank(seeds, weights, alpha=a.alpha)
        t1 = time.perf_counter()
        top, scores = g.top_chunks(10)
        t2 = time.perf_counter()
        t_pr.append(t1 - t0)
        t_top.append(t2 - t1)
    pr_ms, top_ms = 1e3 * min(t_pr), 1e3 * min(t_top)
    bytes_per_iter = 12.0 * g.nnz + 60.0 * n
    out = {"nodes": n, "edges_in": int(len(src)), "nnz": g.nnz, "build_s": round(build_s, 3), "alpha": a.alpha, "iterations": iters,
           "pagerank_ms": round(pr_ms, 3), "ms_per_iteration": round(pr_ms / iters, 4), "top_chunks_ms": round(top_ms, 3),
           "algorithmic_GB_per_iteration": round(bytes_per_iter / 1e9, 3),
           "GBps_incl_launch_and_sync": round(bytes_per_iter * iters / (pr_ms * 1e-3) / 1e9, 1),
           "top": [int(t) for t in top[:3]]}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
