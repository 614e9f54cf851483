import sys, numpy as np
sys.path.insert(0, '/root/repo')
import hirag_prod_amd as hvs
from tests import parity
rng = np.random.default_rng(0)
for dtype in ("f16", "bf16"):
    for d in (96, 64, 1536):
        for nq in (1, 2, 3, 8):
            rows = rng.standard_normal((500, d)).astype(np.float32)
            q = rng.standard_normal((nq, d)).astype(np.float32)
            idx = hvs.HvsIndex(d, dtype); idx.add(rows)
            dist, ids, cnt = idx.search(q, k=3, algo="scan")
            dm = parity.distance_matrix(rows, q, dtype)
            exp = np.sort(dm, axis=1)[:, :3]
            print(dtype, d, nq, "sim ratio per query:", np.round((1 - dist[:, 0]) / (1 - exp[:, 0]), 4))
            idx.close()
