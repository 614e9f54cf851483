"""One query at a time on an fp32 table: the streaming scan against the filter route (hi slabs of the split copy on the
ring kernel, exact fp32 scores for the candidates), end to end through the C ABI with host buffers.

    python tools/single_query_probe.py [--dim 1536]
"""

import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--dim", type=int, default=1536)
    a = ap.parse_args()
    import hirag_prod_amd as hvs

    rng = np.random.default_rng(0)
    for n in (65_536, 131_072, 262_144, 524_288, 1_000_000, 2_000_000, 4_000_000):
        idx = hvs.HvsIndex(a.dim, "f32")
        idx.fill_synthetic(n, seed=5)
        idx.prepare()
        q = rng.standard_normal((64, a.dim)).astype(np.float32)
        line = f"n={n:8d}:"
        ids = {}
        for name, env in (("scan", None), ("filter", "1")):
            if env:
                os.environ["HVS_FILTER_MIN_ROWS"] = env
                os.environ["HVS_STREAM"] = "1"  # the ring kernel below 65 536 rows too
            algo = "scan" if name == "scan" else "auto"
            for i in range(4):
                idx.search(q[i : i + 1], k=10, algo=algo)
            t0 = time.perf_counter()
            out = [idx.search(q[i : i + 1], k=10, algo=algo)[1] for i in range(64)]
            dt = (time.perf_counter() - t0) / 64
            ids[name] = np.concatenate(out)
            os.environ.pop("HVS_FILTER_MIN_ROWS", None)
            os.environ.pop("HVS_STREAM", None)
            line += f"  {name} {dt * 1e3:7.3f} ms"
        print(line + f"  same ids: {bool((ids['scan'] == ids['filter']).all())}", flush=True)
        idx.close()
