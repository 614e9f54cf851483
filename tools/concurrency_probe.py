"""Concurrent `HipVectorStore.query` calls with and without shared launches.

    python tools/concurrency_probe.py [--rows 100000] [--dim 768] [--inflight 25] [--rounds 20]

The reference's evaluation keeps 25 queries in flight (benchmark/2wiki/eval.py:170-180); each is three
query strings folded by LEAST.  The embedding function is a local stand-in (seeded random vectors).
"""

import argparse
import asyncio
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=100_000)
    ap.add_argument("--dim", type=int, default=768)
    ap.add_argument("--inflight", type=int, default=25)
    ap.add_argument("--rounds", type=int, default=20)
    a = ap.parse_args()
    from hirag_prod_amd.store import HipVectorStore

    async def embed(texts):
        return [np.random.default_rng(abs(hash(t)) % (2**32)).standard_normal(a.dim).astype(np.float32) for t in texts]

    async def run(coalesce):
        store = HipVectorStore.create(embed, vector_type="halfvec", coalesce_queries=coalesce)
        await store._init_vdb(embedding_dimension=a.dim)
        rng = np.random.default_rng(1)
        for first in range(0, a.rows, 10_000):
            n = min(10_000, a.rows - first)
            vecs = rng.standard_normal((n, a.dim)).astype(np.float32)
            await store.load_rows("Chunks", ({"documentKey": f"c{first + i}", "workspaceId": "ws", "knowledgeBaseId": "kb",
                                              "text": "t", "vector": vecs[i]} for i in range(n)))

        async def one(j):
            return await store.query([f"q{j}", f"q{j} again", f"q{j} more"], "ws", "kb", "Chunks", distance_threshold=1.5,
                                     columns_to_select=["documentKey"])

        await asyncio.gather(*[one(-1 - j) for j in range(a.inflight)])
        t0 = time.perf_counter()
        out = []
        for r in range(a.rounds):
            out += await asyncio.gather(*[one(r * a.inflight + j) for j in range(a.inflight)])
        dt = time.perf_counter() - t0
        nq = a.rounds * a.inflight
        co = store._coalescer
        print(f"coalesce={coalesce}: {nq} queries, {a.inflight} in flight: {dt / nq * 1e3:.3f} ms per query, {nq / dt:.0f} queries/s"
              + (f"  ({co.requests} requests in {co.launches} launches)" if co else ""), flush=True)
        store.close()
        return [[r["documentKey"] for r in res] for res in out]

    x = asyncio.run(run(False))
    y = asyncio.run(run(True))
    print("same answers:", x == y)


if __name__ == "__main__":
    main()
