"""Generates the golden fixtures in this directory.  Run ONCE in the build container:

    PYTHONPATH=/root/reference/src MPLBACKEND=Agg python tests/golden/make_golden.py

Sources of truth (none of them is this repo's own code):

* ``cosine_*.npz``  -- scikit-learn ``cosine_similarity`` on float64 copies of the
  inputs: the routine the reference calls for its in-process cosine sites
  (``src/hirag_prod/hirag.py:571-588``,
  ``src/hirag_prod/cross_language_search/functions.py:93-100``).  The pgvector
  extension that serves ``PGVector.query`` is not available here, so these are
  the only outside anchor of the cosine formula.
* ``ward_*.npz``    -- the reference's own ``hirag_prod.cluster.HierarchicalClustering``
  (``src/hirag_prod/cluster/hierachical_cluster.py:51-103``), imported from
  ``/root/reference/src`` (importable here; nothing else on the path is).
* ``kmeans_*.npz``  -- ``sklearn.cluster.KMeans`` run for exactly one Lloyd
  iteration from given centroids (the reference has no k-means; this pins the
  build-defined step to the library the reference's clustering uses).

Only inputs and expected outputs are stored; no reference source text.
"""

import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))


def cosine_cases():
    from sklearn.metrics.pairwise import cosine_similarity

    cases = [
        # name, n, d, m, storage dtype of the fixture
        ("cosine_d8_f32", 512, 8, 3, np.float32),
        ("cosine_d768_f32", 32, 768, 1, np.float32),
        ("cosine_d1536_f16", 48, 1536, 3, np.float16),
    ]
    for i, (name, n, d, m, dt) in enumerate(cases):
        rng = np.random.default_rng(1000 + i)
        rows = rng.standard_normal((n, d)).astype(np.float32).astype(dt)
        queries = rng.standard_normal((m, d)).astype(np.float32).astype(dt)
        # planted near-duplicates and an exact duplicate pair so orderings are not all random
        rows[3] = (queries[0].astype(np.float32) + 0.05 * rng.standard_normal(d).astype(np.float32)).astype(dt)
        rows[7] = rows[3]
        rows[11] = (-queries[0].astype(np.float32)).astype(dt)  # anti-parallel: distance 2
        sim = cosine_similarity(rows.astype(np.float64), queries.astype(np.float64))  # [n, m]
        np.savez_compressed(os.path.join(HERE, name + ".npz"), rows=rows, queries=queries, sim=sim)
        print(name, rows.shape, queries.shape, sim.shape)


def ward_cases():
    sys.path.insert(0, "/root/reference/src")
    os.environ.setdefault("MPLBACKEND", "Agg")
    from hirag_prod.cluster import HierarchicalClustering

    for i, (n, d) in enumerate([(10, 16), (8, 64), (10, 1536), (5, 32)]):
        rng = np.random.default_rng(2000 + i)
        x = rng.standard_normal((n, d)).astype(np.float32)
        labels, _, sil, link = HierarchicalClustering(n_clusters=3, linkage_method="ward").fit(x)
        np.savez_compressed(
            os.path.join(HERE, f"ward_n{n}_d{d}.npz"),
            x=x,
            labels=np.asarray(labels, dtype=np.int64),
            silhouette=np.float64(sil if sil is not None else np.nan),
            linkage=np.asarray(link, dtype=np.float64),
        )
        print("ward", n, d, labels, sil)


def kmeans_cases():
    from sklearn.cluster import KMeans

    for i, (n, d, k) in enumerate([(512, 16, 8), (300, 64, 5)]):
        rng = np.random.default_rng(3000 + i)
        x = rng.standard_normal((n, d)).astype(np.float32)
        c0 = x[:k].copy()
        km = KMeans(n_clusters=k, init=c0, n_init=1, max_iter=1, algorithm="lloyd", tol=0.0).fit(x)
        # labels_ after max_iter=1 are the assignment to c0's successors; the assignment to c0
        # itself is what one E-step produces: recover it with a zero-iteration predict on c0.
        km0 = KMeans(n_clusters=k, init=c0, n_init=1, max_iter=1, algorithm="lloyd", tol=0.0)
        km0.cluster_centers_ = c0.astype(np.float32)
        km0._n_threads = 1
        km0._n_features_out = k
        km0.n_features_in_ = d
        assign0 = km0.predict(x)
        np.savez_compressed(
            os.path.join(HERE, f"kmeans_n{n}_d{d}_k{k}.npz"),
            x=x,
            c0=c0,
            assign0=assign0.astype(np.int64),
            centers1=km.cluster_centers_.astype(np.float64),
        )
        print("kmeans", n, d, k, np.bincount(assign0, minlength=k))


if __name__ == "__main__":
    cosine_cases()
    ward_cases()
    kmeans_cases()
