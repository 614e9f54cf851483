"""GPU tier: randomly drawn shapes (dimension, rows, embeddings, fold, k, filters) through every search
route of every row type, each against the float64 oracle -- the odd sizes the fixed cases do not hit
(dimension 1, one short of / one past a 64-element k-step, fewer rows than a tile, 33 or 65 embeddings)."""

import math

import numpy as np
import pytest

from oracle import pgvector_oracle as po
from tests import parity

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def hvs():
    import hirag_prod_amd

    assert hirag_prod_amd.device_count() >= 1
    return hirag_prod_amd


FIXED = [  # (dim, rows, query groups, fold, k)
    (1, 300, 5, 1, 3), (3, 70, 2, 2, 10), (63, 1000, 33, 1, 10), (64, 129, 65, 1, 7), (65, 2049, 9, 3, 20),
    (127, 31, 40, 1, 40), (129, 5000, 17, 4, 128), (200, 4097, 129, 1, 1), (777, 2500, 6, 2, 50),
    (96, 3000, 256, 1, 10), (130, 700, 100, 2, 100), (50, 260, 60, 3, 128),  # 129 - 256 A rows
    (40, 1500, 300, 1, 10), (72, 900, 128, 4, 16),  # two query tiles of 256 A rows
    (100, 3000, 40, 1, 256), (48, 2000, 140, 2, 200),  # k beyond 128 on the matrix pipe (candidate arrays of 1024)
    # k beyond 256: the blocks' best k go unsorted to one list per query (dpr_recall_chunks asks for topk = 500, hirag.py:1257)
    (64, 6000, 30, 1, 300), (80, 9000, 130, 1, 500), (33, 5000, 20, 2, 768), (120, 400, 5, 1, 500),
]


def _cases():
    rng = np.random.default_rng(20240611)
    out = list(FIXED)
    for _ in range(12):
        out.append((int(rng.integers(2, 900)), int(rng.integers(20, 6000)), int(rng.integers(1, 140)), int(rng.integers(1, 5)),
                    int(rng.choice([1, 5, 10, 64, 100]))))
    return out


@pytest.mark.parametrize("stream", ["auto", "ring", "sample"])
@pytest.mark.parametrize("dtype", ["f32", "f16", "bf16"])
def test_random_shapes_all_routes(hvs, dtype, stream, monkeypatch):
    if stream == "ring":  # up to 128 A rows: the single-query-tile ring kernel, whatever the row count
        monkeypatch.setenv("HVS_STREAM", "1")
    if stream == "sample":  # every batched search starts from the k-th distances of a pass over a sample of the rows
        monkeypatch.setenv("HVS_PREPASS", "1")
    rng = np.random.default_rng(7)
    for dim, n, nq, fold, k in _cases():
        rows = rng.standard_normal((n, dim)).astype(np.float32)
        if n > 10:
            rows[3] = 0.0
            rows[7] = rows[2]
        groups = rng.integers(0, 3, n).astype(np.int32)
        q = rng.standard_normal((nq * fold, dim)).astype(np.float32)
        idx = hvs.HvsIndex(dim, dtype)
        idx.add(rows, group_codes=groups)
        gone = rng.permutation(n)[: n // 10]
        idx.set_deleted(gone, True)
        keep_all = np.ones(n, bool)
        keep_all[gone] = False
        dm = parity.distance_matrix(rows, q, dtype)
        for gf, thr in ((-1, math.inf), (1, 1.05)):
            keep = keep_all & ((groups == gf) if gf >= 0 else True)
            routes = ["scan", "mfma"] + (["mfma_f32", "mfma_split"] if dtype == "f32" else [])
            if fold > 4:
                routes = ["scan"]
            for algo in routes:
                if algo != "scan" and k > 768:
                    continue
                dist, ids, cnt = idx.search(q, k=k, fold=fold, group_filter=gf, max_distance=thr, algo=algo)
                for i in range(nq):
                    full = po.least_fold(dm[i * fold : (i + 1) * fold])
                    parity.check_topk(dist[i], ids[i], cnt[i], full, k, threshold=None if math.isinf(thr) else thr, keep=keep,
                                      label=f"{dtype} {algo} dim={dim} n={n} nq={nq} fold={fold} k={k} gf={gf} q={i}")
        idx.close()
