"""Wall time of hvs_kmeans_assign alone (no torch work around it), filter route against the three products."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import hirag_prod_amd as hvs  # noqa: E402
from hirag_prod_amd import _lib  # noqa: E402


def main(rows=10_000_000, dim=1536, K=4096):
    import torch

    idx = hvs.HvsIndex(dim, "f32")
    idx.fill_synthetic(rows, seed=1234)
    c = torch.from_numpy(idx.gather(np.arange(K))).cuda()
    for label, env in (("filter", {}), ("three products", {"HVS_NO_FILTER": "1"})):
        os.environ.update(env)
        for sums in (True, False):
            idx.kmeans_assign(c, want_sums=sums)
            torch.cuda.synchronize()
            idx.profile(True)
            idx.profile_reset()
            t0 = time.perf_counter()
            for _ in range(3):
                idx.kmeans_assign(c, want_sums=sums)
            torch.cuda.synchronize()
            wall = (time.perf_counter() - t0) / 3
            a_ms, a_n = idx.profile_get(_lib.HVS_PROF_KMEANS_ASSIGN)
            u_ms, u_n = idx.profile_get(_lib.HVS_PROF_KMEANS_UPDATE)
            idx.profile(False)
            print(f"{label}, sums={sums}: wall {wall * 1e3:.1f} ms, assign {a_ms / max(a_n, 1):.1f} ms, update {u_ms / max(u_n, 1):.1f} ms", flush=True)
        for k in env:
            del os.environ[k]
    idx.close()


if __name__ == "__main__":
    main()
