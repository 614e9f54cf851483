"""Distance column with many embeddings (LEAST over m sentence embeddings of every row): one pass on the
matrix pipe against one streaming pass per eight embeddings, both against the HBM floor of one pass.

    python tools/distcol_perf.py [--rows 4000000] [--dim 1536]
"""

import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=4_000_000)
    ap.add_argument("--dim", type=int, default=1536)
    ap.add_argument("--dtypes", default="f16,f32")
    a = ap.parse_args()
    import numpy as np

    import hirag_prod_amd as hvs
    from hirag_prod_amd import _lib

    n, d = a.rows, a.dim
    rng = np.random.default_rng(0)
    for dtype in a.dtypes.split(","):
        idx = hvs.HvsIndex(d, dtype)
        idx.fill_synthetic(n, seed=1)
        floor = n * d * (4 if dtype == "f32" else 2) / 8e12 * 1e3
        for m in (8, 16, 32, 64, 128):
            q = rng.standard_normal((m, d)).astype(np.float32)
            for route in ("matrix", "stream"):
                if route == "stream":
                    os.environ["HVS_NO_MFMA_DIST"] = "1"
                else:
                    os.environ.pop("HVS_NO_MFMA_DIST", None)
                idx.distances(q)
                idx.profile(True)
                idx.profile_reset()
                for _ in range(3):
                    idx.distances(q)
                ms = idx.profile_get(_lib.HVS_PROF_SCAN)[0] + idx.profile_get(_lib.HVS_PROF_MFMA_SEARCH)[0]
                idx.profile(False)
                per = ms / 3
                print(f"{dtype} m={m:4d} {route:6s}: {per:7.2f} ms  ({per / floor:5.2f} x the {floor:.2f} ms HBM floor)", flush=True)
            os.environ.pop("HVS_NO_MFMA_DIST", None)
        idx.close()
