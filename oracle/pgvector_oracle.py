"""numpy restatement of ``PGVector.query`` + pgvector's ``<=>`` operator.

TEST INFRASTRUCTURE ONLY (see ``oracle/__init__.py``); parity unpinned: no
golden vector for this path exists in the reference, the anchor is
scikit-learn's ``cosine_similarity`` (``tests/golden/``).

What is restated, and from where:

* ``cosine_distance``  -- pgvector extension, ``vector.c`` / ``halfvec.c``
  ``cosine_distance``: three float accumulators ``dot, |a|^2, |b|^2``,
  ``sim = dot / sqrt(|a|^2 * |b|^2)`` in double, clamp to [-1, 1], return
  ``1 - sim`` as float8.  A zero-norm operand gives 0/0 = NaN.  Called by the
  reference as ``model.vector.cosine_distance(q_emb)``
  (``src/hirag_prod/storage/pgvector.py:561-563``).
* ``least_fold``       -- ``func.least(*distance_expressions)``
  (``pgvector.py:565-569``): PostgreSQL ``LEAST`` with float8 ordering, where
  NaN sorts above every number, so NaN only survives if every input is NaN.
* ``query``            -- filters, strict ``distance < threshold``, ``ORDER BY
  distance ASC LIMIT topk`` (``pgvector.py:573-585``).  PostgreSQL leaves the
  order of equal distances unspecified; the oracle (and the GPU path) break
  ties by ascending row index.
* halfvec storage      -- ``HALFVEC(dim)`` column (``schema/vector_config.py:11``)
  rounds every stored component to IEEE fp16; the query vector is bound with
  the column's type and is rounded the same way (``pgvector.py:558-563``).
* ``dpr_scores``       -- the in-repo numpy statement of cosine top-k,
  ``src/hirag_prod/hirag.py:1305-1327``.
"""

from __future__ import annotations

import numpy as np

DEFAULT_TOP_K = 10  # configs/hi_rag_config.py:42
DEFAULT_TOP_N = 5  # configs/hi_rag_config.py:43
DEFAULT_DISTANCE_THRESHOLD = 0.8  # configs/hi_rag_config.py:56


def round_storage(x: np.ndarray, storage: str) -> np.ndarray:
    """Value held by the column for ``storage`` in {"f32", "f16", "bf16"}, as float64."""
    x = np.asarray(x)
    if storage == "f32":
        return x.astype(np.float32).astype(np.float64)
    if storage == "f16":
        return x.astype(np.float32).astype(np.float16).astype(np.float64)
    if storage == "bf16":
        return bf16_round(x.astype(np.float32)).astype(np.float64)
    raise ValueError(f"unknown storage {storage!r}")


def bf16_round(x32: np.ndarray) -> np.ndarray:
    """Round-to-nearest-even fp32 -> bf16, returned widened back to fp32."""
    x32 = np.ascontiguousarray(x32, dtype=np.float32)
    u = x32.view(np.uint32).astype(np.uint64)
    nan = np.isnan(x32)
    r = ((u + 0x7FFF + ((u >> 16) & 1)) >> 16) << 16
    out = (r & 0xFFFFFFFF).astype(np.uint32).view(np.float32).copy()
    out[nan] = np.nan
    return out


def cosine_distance(rows: np.ndarray, q: np.ndarray, accumulate: str = "f64") -> np.ndarray:
    """``rows <=> q`` for every row; float64 result, NaN where a norm is zero.

    ``accumulate="f64"`` is the exact-arithmetic statement the GPU path is held
    to; ``"f32"`` mimics pgvector's three float accumulators (sequential order).
    """
    rows = np.asarray(rows)
    q = np.asarray(q)
    if accumulate == "f64":
        a = rows.astype(np.float64)
        b = q.astype(np.float64)
        dot = a @ b
        na = np.einsum("ij,ij->i", a, a)
        nb = float(b @ b)
    elif accumulate == "f32":
        a = rows.astype(np.float32)
        b = q.astype(np.float32)
        dot = np.zeros(a.shape[0], dtype=np.float32)
        na = np.zeros(a.shape[0], dtype=np.float32)
        nb = np.float32(0.0)
        for j in range(a.shape[1]):  # the C loop's order, one rounding per step
            dot = dot + a[:, j] * b[j]
            na = na + a[:, j] * a[:, j]
            nb = np.float32(nb + b[j] * b[j])
        dot = dot.astype(np.float64)
        na = na.astype(np.float64)
        nb = float(nb)
    else:
        raise ValueError(accumulate)
    with np.errstate(invalid="ignore", divide="ignore"):
        sim = dot / np.sqrt(na * nb)
    sim = np.where(sim > 1.0, 1.0, np.where(sim < -1.0, -1.0, sim))  # keeps NaN
    return 1.0 - sim


def least_fold(dists: np.ndarray) -> np.ndarray:
    """PostgreSQL ``LEAST`` down axis 0 of an ``[m, N]`` array (NaN = largest)."""
    dists = np.asarray(dists, dtype=np.float64)
    all_nan = np.isnan(dists).all(axis=0)
    with np.errstate(invalid="ignore"):
        out = np.nanmin(np.where(np.isnan(dists), np.inf, dists), axis=0)
    out = np.where(all_nan, np.nan, out)
    return out


def order_by_limit(dist: np.ndarray, keep: np.ndarray | None, threshold: float | None, k: int):
    """``WHERE keep AND dist < threshold ORDER BY dist ASC LIMIT k`` -> (indices, dists)."""
    dist = np.asarray(dist, dtype=np.float64)
    ok = ~np.isnan(dist)
    if threshold is not None:
        with np.errstate(invalid="ignore"):
            ok &= dist < float(threshold)
    if keep is not None:
        ok &= np.asarray(keep, dtype=bool)
    idx = np.nonzero(ok)[0]
    order = np.lexsort((idx, dist[idx]))[: max(0, int(k))]  # primary dist, then row index
    sel = idx[order]
    return sel.astype(np.int64), dist[sel]


def query(
    rows: np.ndarray,
    q_embs: np.ndarray,
    k: int = DEFAULT_TOP_K,
    threshold: float | None = DEFAULT_DISTANCE_THRESHOLD,
    keep: np.ndarray | None = None,
    storage: str = "f32",
    accumulate: str = "f64",
):
    """One ``PGVector.query`` call: ``q_embs`` is ``[m, D]`` (m = 1..4 query variants)."""
    q_embs = np.atleast_2d(np.asarray(q_embs))
    rows_s = round_storage(rows, storage)
    # halfvec columns bind the parameter as halfvec; vector columns as float4
    q_s = round_storage(q_embs, storage)
    d = np.stack([cosine_distance(rows_s, q_s[i], accumulate) for i in range(q_s.shape[0])])
    return order_by_limit(least_fold(d), keep, threshold, k)


def dpr_scores(chunk_matrix: np.ndarray, query_vec: np.ndarray, k: int):
    """``HiRAG.dpr_recall_chunks`` step 3-4 (``hirag.py:1305-1327``): returns (order, norm_scores)."""
    chunk_matrix = np.array(chunk_matrix, dtype=np.float32)
    query_vec = np.array(query_vec, dtype=np.float32)
    denom = np.linalg.norm(chunk_matrix, ord=2, axis=1, keepdims=True)
    denom[denom == 0] = 1.0
    cm = chunk_matrix / denom
    qn = query_vec / (np.linalg.norm(query_vec) + 1e-12)
    scores = cm @ qn
    s_min = float(scores.min())
    s_max = float(scores.max())
    if s_max > s_min:
        norm_scores = (scores - s_min) / (s_max - s_min)
    else:
        norm_scores = np.zeros_like(scores)
    order = np.argsort(-norm_scores)[: max(0, k)]
    return order, norm_scores


def cosine_matrix(a: np.ndarray, b: np.ndarray) -> np.ndarray:
    """float64 ``cosine_similarity(A, B)`` as scikit-learn defines it (zero rows -> 0)."""
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    na = np.linalg.norm(a, axis=1, keepdims=True)
    nb = np.linalg.norm(b, axis=1, keepdims=True)
    na[na == 0] = 1.0
    nb[nb == 0] = 1.0
    return (a / na) @ (b / nb).T


# ---------------------------------------------------------------------------
# Synthetic corpus: the same counter-based generator the device fills HBM with
# (hvs_index_fill_synthetic).  Pure integer hashing + ONE float multiply, so
# the host regenerates any row bit-for-bit.
# ---------------------------------------------------------------------------

_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)
SYNTH_SCALE = np.float32(1.0 / 37837.0)  # sum of four U{0..65535}: sigma = 37837.2


def _splitmix64(x: np.ndarray) -> np.ndarray:
    with np.errstate(over="ignore"):
        z = (x + np.uint64(0x9E3779B97F4A7C15)) & _M64
        z = ((z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)) & _M64
        z = ((z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)) & _M64
        return z ^ (z >> np.uint64(31))


def synth_rows(seed: int, row_begin: int, n: int, dim: int) -> np.ndarray:
    """Rows ``[row_begin, row_begin+n)`` of the synthetic corpus, float32 ``[n, dim]``."""
    rows = (np.arange(row_begin, row_begin + n, dtype=np.uint64))[:, None]
    cols = np.arange(dim, dtype=np.uint64)[None, :]
    with np.errstate(over="ignore"):
        ctr = (rows * np.uint64(dim) + cols) & _M64
        h = _splitmix64(ctr ^ _splitmix64(np.full((1, 1), seed, dtype=np.uint64)))
    s = (
        (h & np.uint64(0xFFFF)).astype(np.int64)
        + ((h >> np.uint64(16)) & np.uint64(0xFFFF)).astype(np.int64)
        + ((h >> np.uint64(32)) & np.uint64(0xFFFF)).astype(np.int64)
        + ((h >> np.uint64(48)) & np.uint64(0xFFFF)).astype(np.int64)
        - 131070
    )
    return s.astype(np.float32) * SYNTH_SCALE


CLUSTER_SPREAD = (1.0e-3, 1.4e-3, 2.0e-3, 2.8e-3, 4.0e-3, 5.6e-3, 8.0e-3, 1.0e-2)


def synth_cluster_of(seed: int, row_begin: int, n: int, n_clusters: int) -> np.ndarray:
    """Cluster number of global rows ``[row_begin, row_begin + n)`` of the clustered synthetic corpus."""
    rows = np.arange(row_begin, row_begin + n, dtype=np.uint64)
    with np.errstate(over="ignore"):
        mix = _splitmix64(np.array([np.uint64(seed) ^ np.uint64(0xC1)], dtype=np.uint64))
        return (_splitmix64(rows ^ mix) % np.uint64(n_clusters)).astype(np.int64)


def synth_clustered_rows(seed: int, row_begin: int, n: int, dim: int, n_clusters: int) -> np.ndarray:
    """Host twin of the device's clustered generator (``hvs_index_fill_clustered``), float32 ``[n, dim]``, bit for bit:
    centre of the row's cluster (row c of ``synth_rows(seed + 1, ...)``) + sigma(c) * ``synth_rows(seed, ...)``, the
    product and the sum each rounded to float32."""
    cl = synth_cluster_of(seed, row_begin, n, n_clusters)
    with np.errstate(over="ignore"):
        mix = _splitmix64(np.array([np.uint64(seed) ^ np.uint64(0xC2)], dtype=np.uint64))
        level = (_splitmix64(cl.astype(np.uint64) ^ mix) & np.uint64(7)).astype(np.int64)
    sigma = np.sqrt(np.array(CLUSTER_SPREAD, dtype=np.float64)).astype(np.float32)[level]
    uniq, inv = np.unique(cl, return_inverse=True)
    centres = np.concatenate([synth_rows(seed + 1, int(c), 1, dim) for c in uniq]) if len(uniq) else np.zeros((0, dim), np.float32)
    noise = synth_rows(seed, row_begin, n, dim)
    prod = (sigma[:, None] * noise).astype(np.float32)
    return (centres[inv] + prod).astype(np.float32)
