"""The reference's small dense cosine sites, served by one device routine (``hvs_cosine_matrix``).

Each function keeps the name, arguments and result shape of the reference code it stands in for,
so a caller swaps the import and nothing else:

* ``cosine_similarity(A, B)``        -- scikit-learn's function as the reference imports it
  (``hirag.py:575-577``, ``cross_language_search/functions.py``: zero rows give similarity 0)
* ``calculate_similarity``           -- ``HiRAG.calculate_similarity`` (``hirag.py:571-588``)
* ``validate_similarity``            -- ``cross_language_search/functions.py:85-113``
* ``filter_synonyms``                -- the synonym filter inside
  ``get_synonyms_and_validate_and_translate`` (``functions.py:152-171``)
* ``dpr_rank``                       -- steps 3-4 of ``HiRAG.dpr_recall_chunks`` (``hirag.py:1305-1329``)

There is no CPU path: without the HIP library the call raises.
"""

from __future__ import annotations

import ctypes
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np

from ._lib import check, load_library
from .vdb_interface import DEFAULT_QUERY_TOP_K


def cosine_similarity(a, b=None, device: int = 0) -> np.ndarray:
    """``[m, dim] x [n, dim] -> [m, n]`` float32 similarities computed on the GPU.

    Accepts what scikit-learn's function accepts at the reference's call sites: nested lists or
    arrays, one- or two-dimensional; ``b=None`` means ``b = a``.
    """
    a = np.ascontiguousarray(np.atleast_2d(np.asarray(a, dtype=np.float32)))
    b = a if b is None else np.ascontiguousarray(np.atleast_2d(np.asarray(b, dtype=np.float32)))
    if a.ndim != 2 or b.ndim != 2:
        raise ValueError("cosine_similarity expects 2-D inputs")
    if a.shape[1] != b.shape[1]:
        raise ValueError(f"Incompatible dimension for X and Y matrices: X.shape[1] == {a.shape[1]} while Y.shape[1] == {b.shape[1]}")
    m, n, dim = a.shape[0], b.shape[0], a.shape[1]
    out = np.empty((m, n), dtype=np.float32)
    if m == 0 or n == 0:
        return out
    check(load_library().hvs_cosine_matrix(device, None, a.ctypes.data_as(ctypes.c_void_p), m,
                                           b.ctypes.data_as(ctypes.c_void_p), n, dim,
                                           out.ctypes.data_as(ctypes.c_void_p), 0))
    return out


async def calculate_similarity(sentence_embedding: List[float], references: Dict[str, Optional[List[float]]]) -> List[Dict[str, float]]:
    """One launch for all references instead of one scikit-learn call per reference; entries whose
    embedding is ``None`` are skipped and the input order is kept, as in the reference's loop."""
    keys = [k for k, e in references.items() if e is not None]
    if not keys:
        return []
    sims = cosine_similarity([sentence_embedding], [references[k] for k in keys])[0]
    return [{"documentKey": k, "similarity": s} for k, s in zip(keys, sims)]


async def validate_similarity(str_embedding_np_array, search_embedding_np_array, matched_list_dict_batch, threshold: float = 0.8) -> None:
    """Blank (set to ``None``) every matched span whose best similarity to any search embedding is
    ``<= threshold``; spans are consumed in the reference's order: per dict, ``"original"`` then
    ``"translation"``."""
    best = cosine_similarity(str_embedding_np_array, search_embedding_np_array).max(axis=1).tolist()
    cur = 0
    for matched in matched_list_dict_batch:
        for field in ("original", "translation"):
            spans = matched[field]
            if spans is not None:
                for i in range(len(spans)):
                    if best[cur] <= threshold:
                        spans[i] = None
                    cur += 1


def filter_synonyms(synonym_list: Sequence[str], embedding_np_array: np.ndarray, min_similarity: float = 0.75) -> Tuple[List[str], np.ndarray]:
    """``embedding_np_array`` holds the synonyms' embeddings first and the search string's last.
    Keeps the synonyms whose similarity to the search exceeds ``min_similarity``, most similar first
    (stable for equal similarities, as ``list.sort`` is), and returns them with their embeddings."""
    n = len(synonym_list)
    sims = cosine_similarity(embedding_np_array[:n], embedding_np_array[-1:]).flatten().tolist()
    kept = [(synonym_list[i], embedding_np_array[i : i + 1], s) for i, s in enumerate(sims) if s > min_similarity]
    kept.sort(key=lambda t: t[2], reverse=True)
    if not kept:
        return [], np.empty((0, np.asarray(embedding_np_array).shape[1]), dtype=np.asarray(embedding_np_array).dtype)
    return [t[0] for t in kept], np.concatenate([t[1] for t in kept], axis=0)


def dpr_rank(chunk_matrix, query_vec, topk: Optional[int] = None) -> Tuple[np.ndarray, np.ndarray]:
    """Cosine scores of the candidate pool against the query, min-max normalised, and the indices of
    the ``topk`` largest in the order ``np.argsort(-scores)`` gives.  Returns ``(order, norm_scores)``.

    A zero candidate row scores 0 (the reference divides it by 1), and so does a zero query (divided
    by 1e-12); both are what the device routine returns for a zero norm.
    """
    chunk_matrix = np.asarray(chunk_matrix, dtype=np.float32)
    query_vec = np.asarray(query_vec, dtype=np.float32).reshape(1, -1)
    scores = cosine_similarity(chunk_matrix, query_vec)[:, 0]
    s_min = float(scores.min())
    s_max = float(scores.max())
    if s_max > s_min:
        norm_scores = (scores - s_min) / (s_max - s_min)
    else:
        norm_scores = np.zeros_like(scores)
    order = np.argsort(-norm_scores)[: max(0, topk or DEFAULT_QUERY_TOP_K)]  # falsy top-k -> configured default (hirag.py:1325-1327)
    return order, norm_scores


__all__ = ["cosine_similarity", "calculate_similarity", "validate_similarity", "filter_synonyms", "dpr_rank"]
