"""float64 statement of one Lloyd iteration (TEST INFRASTRUCTURE ONLY).

The reference has no k-means (its only clustering is scikit-learn ward
agglomeration of <= 10 retrieved chunks, ``src/hirag_prod/cluster/
hierachical_cluster.py:42-46``), so this step is defined by the build and its
parity is unpinned by the reference.  It is cross-checked in the tests against
``sklearn.cluster.KMeans(init=<centroids>, n_init=1, max_iter=1,
algorithm="lloyd")``, the library the reference's clustering code is built on.

Definition: ``assign[i] = argmin_j |x_i - c_j|^2`` with the lowest ``j`` on
exact ties (numpy ``argmin``); new centroid = mean of its members; an empty
cluster keeps its previous centroid.
"""

from __future__ import annotations

import numpy as np


def assign(rows: np.ndarray, centroids: np.ndarray, block: int = 4096):
    """Returns ``(assign int32 [n], margin float64 [n])``; margin = gap between the best and
    second-best squared distance, which tests use to skip rows an fp32 path may flip."""
    x = np.asarray(rows, dtype=np.float64)
    c = np.asarray(centroids, dtype=np.float64)
    cn = np.einsum("ij,ij->i", c, c)
    out = np.empty(x.shape[0], dtype=np.int32)
    margin = np.empty(x.shape[0], dtype=np.float64)
    for s in range(0, x.shape[0], block):
        xb = x[s : s + block]
        v = cn[None, :] - 2.0 * (xb @ c.T)
        a = np.argmin(v, axis=1)
        out[s : s + block] = a
        if c.shape[0] > 1:
            part = np.partition(v, 1, axis=1)
            margin[s : s + block] = part[:, 1] - part[:, 0]
        else:
            margin[s : s + block] = np.inf
    return out, margin


def update(rows: np.ndarray, assignment: np.ndarray, centroids: np.ndarray):
    """Returns ``(new_centroids float64 [K, D], sums float64 [K, D], counts int64 [K])``."""
    x = np.asarray(rows, dtype=np.float64)
    k = centroids.shape[0]
    sums = np.zeros((k, x.shape[1]), dtype=np.float64)
    np.add.at(sums, assignment, x)
    counts = np.bincount(assignment, minlength=k).astype(np.int64)
    new_c = np.asarray(centroids, dtype=np.float64).copy()
    nz = counts > 0
    new_c[nz] = sums[nz] / counts[nz, None]
    return new_c, sums, counts
