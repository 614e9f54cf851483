"""CPU statement of the step after retrieval: reset weights + personalised PageRank over the
knowledge graph (TEST INFRASTRUCTURE ONLY: imported by tests, the smoke check and the CPU leg of
benchmarks, never by the product path).

What it restates
* ``PGVector.pagerank_top_chunks_with_reset`` (``src/hirag_prod/storage/pgvector.py:381-450``):
  a ``networkx.DiGraph`` from the ``(source, target)`` rows, ``to_undirected()``, the reset weights
  filtered to finite positive values on nodes of the graph and normalised, ``networkx.pagerank`` with
  that personalisation, the ``chunk-`` nodes sorted by descending score (stable), first ``topk``.
* ``QueryService.pagerank_chunks`` (``src/hirag_prod/storage/query_service.py:290-379``): how the
  reset weights are formed from the ranked chunks and the ranked triplets.

The PageRank itself lives in a third-party dependency of the reference, ``networkx==3.3``
(``pyproject.toml:21``); its algorithm (``networkx/algorithms/link_analysis/pagerank_alg.py``,
``_pagerank_scipy``: power iteration on the row-normalised adjacency, L1 stopping rule
``err < N * tol``, ``tol = 1e-6``, at most 100 iterations, dangling mass spread by the
personalisation vector) is restated below with numpy only.  Pinned by golden vectors generated with
the networkx of this image (3.4.2, same routine) through ``tests/golden/make_pagerank_golden.py``.
"""

from __future__ import annotations

import math
from typing import Dict, Iterable, List, Sequence, Tuple

import numpy as np


class PowerIterationFailedConvergence(Exception):
    pass


def node_index(edges: Iterable[Tuple[str, str]]) -> Dict[str, int]:
    """Node order of ``G.add_edge(source, target)`` over the rows (pgvector.py:406-409): first
    appearance, source before target; rows with an empty endpoint are skipped."""
    index: Dict[str, int] = {}
    for s, t in edges:
        if s and t:
            if s not in index:
                index[s] = len(index)
            if t not in index:
                index[t] = len(index)
    return index


def undirected_csr(n: int, src: np.ndarray, dst: np.ndarray):
    """Symmetric 0/1 adjacency of the undirected simple graph (parallel and antiparallel rows
    collapse; a self loop is one diagonal entry, as ``to_scipy_sparse_array`` stores it).
    Returns ``(row_ptr int64 [n+1], col int32 [nnz])`` with ascending columns in each row."""
    src = np.asarray(src, dtype=np.int64)
    dst = np.asarray(dst, dtype=np.int64)
    keys = np.unique(np.concatenate([src * n + dst, dst * n + src])) if len(src) else np.zeros(0, np.int64)
    rows = keys // n if n else keys
    col = (keys - rows * n).astype(np.int32)
    row_ptr = np.zeros(n + 1, dtype=np.int64)
    np.add.at(row_ptr, rows + 1, 1)
    return np.cumsum(row_ptr), col


def pagerank(n: int, src, dst, personalization: np.ndarray, alpha: float = 0.85, max_iter: int = 100, tol: float = 1.0e-6):
    """``_pagerank_scipy`` on the undirected graph.  ``personalization``: dense float64 [n] (any
    positive scale).  Returns ``(scores float64 [n], iterations)``."""
    if n == 0:
        return np.zeros(0), 0
    row_ptr, col = undirected_csr(n, src, dst)
    deg = np.diff(row_ptr).astype(np.float64)
    inv = np.zeros(n)
    inv[deg != 0] = 1.0 / deg[deg != 0]
    p = np.asarray(personalization, dtype=np.float64)
    if p.sum() == 0:
        raise ZeroDivisionError
    p = p / p.sum()
    dangling = np.where(deg == 0)[0]
    x = np.repeat(1.0 / n, n)
    rows_of = np.repeat(np.arange(n), np.diff(row_ptr))
    for it in range(1, max_iter + 1):
        xlast = x
        y = x * inv
        # (x @ A)[j] = sum over neighbours i of x[i] / deg[i], ascending i (the matrix is symmetric)
        xa = np.zeros(n)
        np.add.at(xa, rows_of, y[col])
        x = alpha * (xa + x[dangling].sum() * p) + (1 - alpha) * p
        if np.absolute(x - xlast).sum() < n * tol:
            return x, it
    raise PowerIterationFailedConvergence(max_iter)


def pagerank_top_chunks_with_reset(edges: Sequence[Tuple[str, str]], reset_weights: Dict[str, float], topk: int,
                                   alpha: float = 0.85) -> List[Tuple[str, float]]:
    """pgvector.py:381-450 for the rows of one (workspace, knowledge base)."""
    if topk <= 0 or not edges:
        return []
    index = node_index(edges)
    if not index:
        return []
    personalization: Dict[str, float] = {}
    total = 0.0
    for node, w in (reset_weights or {}).items():
        if node in index:
            try:
                val = float(w)
            except Exception:
                continue
            if not math.isfinite(val) or val <= 0:
                continue
            personalization[node] = val
            total += val
    if total <= 0:
        return []
    p = np.zeros(len(index))
    for node, val in personalization.items():
        p[index[node]] = val / total
    pairs = [(index[s], index[t]) for s, t in edges if s and t]
    src = np.array([a for a, _ in pairs], dtype=np.int64)
    dst = np.array([b for _, b in pairs], dtype=np.int64)
    scores, _ = pagerank(len(index), src, dst, p, alpha=alpha)
    chunk_scores = [(node, float(scores[i])) for node, i in index.items() if str(node).startswith("chunk-")]
    chunk_scores.sort(key=lambda t: t[1], reverse=True)
    return chunk_scores[:topk]


def fusion_reset_weights(query_chunk_ids: Sequence[str], query_triplets: Sequence[dict], query_entity_ids: Sequence[str],
                         entity_chunk_count: Dict[str, int], link_top_k: int = 30, passage_node_weight: float = 0.6):
    """query_service.py:290-379 -> ``(reset_weights, phrase_weights)``; an empty ``phrase_weights``
    means "no fact signal": the caller returns the retrieval order unchanged."""
    passage: Dict[str, float] = {}
    raw = []
    for rank, cid in enumerate(query_chunk_ids):
        if cid:
            w = 1.0 / (rank + 1)
            passage[cid] = w
            raw.append(w)
    if raw:
        lo, hi = min(raw), max(raw)
        scale = (hi - lo) if (hi - lo) > 0 else 1.0
        for cid in list(passage):
            passage[cid] = ((passage[cid] - lo) / scale) * passage_node_weight
    phrase: Dict[str, float] = {}
    if query_triplets:
        occurrences: Dict[str, int] = {}
        for rank, rel in enumerate(query_triplets):
            base = 1.0 / (rank + 1)
            for ent in (rel.get("source"), rel.get("target")):
                if not ent:
                    continue
                phrase[ent] = phrase.get(ent, 0.0) + base
                occurrences[ent] = occurrences.get(ent, 0) + 1
        for ent in query_entity_ids:
            freq = entity_chunk_count.get(ent, 0)
            denom = float(freq) if freq and freq > 0 else 1.0
            phrase[ent] = (phrase[ent] / denom) / float(occurrences.get(ent, 1))
        phrase = dict(sorted(phrase.items(), key=lambda kv: kv[1], reverse=True)[: max(1, link_top_k)])
    reset: Dict[str, float] = {}
    for k, v in passage.items():
        if v > 0:
            reset[k] = reset.get(k, 0.0) + v
    for k, v in phrase.items():
        if v > 0:
            reset[k] = reset.get(k, 0.0) + v
    return reset, phrase
