"""cProfile of sequential `HipVectorStore.query` calls (one in flight): where the host side of one query goes.
    python tools/store_profile.py [rows] [dim] [calls]"""
import asyncio
import cProfile
import os
import pstats
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
rows = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000
dim = int(sys.argv[2]) if len(sys.argv) > 2 else 768
calls = int(sys.argv[3]) if len(sys.argv) > 3 else 300
from hirag_prod_amd.store import HipVectorStore  # noqa: E402

table = np.random.default_rng(5).standard_normal((4096, dim)).astype(np.float32)


async def embed(texts):  # a stand-in that costs next to nothing
    return [table[hash(t) % 4096] for t in texts]


async def main():
    store = HipVectorStore.create(embed, vector_type="halfvec")
    await store._init_vdb(embedding_dimension=dim)
    rng = np.random.default_rng(1)
    for first in range(0, rows, 10_000):
        n = min(10_000, rows - first)
        vecs = rng.standard_normal((n, dim)).astype(np.float32)
        await store.load_rows("Chunks", ({"documentKey": f"c{first + i}", "workspaceId": f"ws{(first + i) % 8}", "knowledgeBaseId": "kb",
                                          "text": "t", "vector": vecs[i]} for i in range(n)))

    async def one(j):
        return await store.query([f"q{j}", f"q{j} again", f"q{j} more"], "ws3", "kb", "Chunks", distance_threshold=1.5,
                                 columns_to_select=["documentKey"])

    for j in range(20):
        await one(-j)
    t0 = time.perf_counter()
    for j in range(calls):
        await one(j)
    dt = time.perf_counter() - t0
    co = store._coalescer
    print(f"{calls} sequential queries: {dt / calls * 1e3:.3f} ms per query; last launch {co._last_launch}, {co.inline_launches} of {co.launches} launches on the loop's thread; "
          f"partition {store.tables['Chunks'].index.partition_info()}", flush=True)
    pr = cProfile.Profile()
    pr.enable()
    for j in range(calls):
        await one(calls + j)
    pr.disable()
    pstats.Stats(pr).sort_stats("cumulative").print_stats(28)
    store.close()


asyncio.run(main())
