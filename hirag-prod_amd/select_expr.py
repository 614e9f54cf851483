"""Native forms of the SQL-expression arguments of ``PGVector.query_by_keys``
(``src/hirag_prod/storage/pgvector.py:606-660``): ``additional_data_to_select``, ``additional_where_clause_list`` and
``order_by``.  The reference passes SQLAlchemy expressions that PostgreSQL evaluates; ``HipVectorStore.query_by_keys``
takes these plain objects under the same argument names and evaluates them itself -- the distance column on the GPU,
the rest over the metadata rows on the host.

    additional_data_to_select = {"search_sentence_cosine_distance": LeastCosineDistance(sentence_embeddings)}
    additional_where_clause_list = after_cursor(order_by, last_row)          # keyset pagination, or any row -> bool
    order_by = ["type", "fileName", Key(lambda r: r["pageNumber"] if r["pageNumber"] is not None else -1), Desc("chunkIdx")]

is the call ``cross_language_search/search_service.py:100-131`` makes, restated.
"""

from __future__ import annotations

import functools
from typing import Any, Callable, List, Sequence, Union

import numpy as np


class LeastCosineDistance:
    """``LEAST(vector <=> e_1, ..., vector <=> e_m)`` as a selected column (``func.least(*[Item.vector.cosine_distance(e)
    ...])``, ``search_service.py:102-109``).  NULL (``None``) where every term is undefined (zero-norm row)."""

    def __init__(self, embeddings):
        e = np.asarray(embeddings, dtype=np.float32)
        if e.ndim == 1:
            e = e[None, :]
        if e.ndim != 2 or e.shape[0] == 0:
            raise ValueError("LeastCosineDistance needs at least one embedding")
        self.embeddings = e


class Key:
    """An ORDER BY term computed from the row: ``Key(lambda row: ...)`` or ``Key("column")``; ascending, NULLs last
    (PostgreSQL's default)."""

    descending = False

    def __init__(self, source: Union[str, Callable[[dict], Any]]):
        self.source = source

    def value(self, row: dict):
        return row.get(self.source) if isinstance(self.source, str) else self.source(row)


class Desc(Key):
    """The same, descending, NULLs first (PostgreSQL's default for DESC)."""

    descending = True


def _as_key(term) -> Key:
    if isinstance(term, Key):
        return term
    if isinstance(term, str) or callable(term):
        return Key(term)
    if isinstance(term, tuple) and len(term) == 2 and str(term[1]).lower() in ("asc", "desc"):
        return Desc(term[0]) if str(term[1]).lower() == "desc" else Key(term[0])
    raise TypeError(f"order_by term {term!r}: pass a column name, a callable, Key(...), Desc(...) or (column, 'asc' | 'desc')")


def _cmp_values(a, b) -> int:
    """ascending comparison with NULLs last"""
    if a is None or b is None:
        return 0 if a is b else (1 if a is None else -1)
    return (a > b) - (a < b)


def order_values(order_by: Sequence[Any], row: dict) -> tuple:
    return tuple(_as_key(t).value(row) for t in order_by)


def compare_rows(order_by: Sequence[Any], va: tuple, vb: tuple) -> int:
    """< 0 when a row with ORDER BY values ``va`` sorts before one with ``vb``"""
    for term, a, b in zip(order_by, va, vb):
        c = _cmp_values(a, b)
        if c:
            return -c if _as_key(term).descending else c
    return 0


def sort_rows(order_by: Sequence[Any], rows: List[dict], positions: List[int]) -> List[int]:
    """``positions`` reordered by ``order_by`` over ``rows[position]``; equal rows keep their table order."""
    keys = list(order_by)
    vals = {pos: order_values(keys, rows[pos]) for pos in positions}
    return sorted(positions, key=functools.cmp_to_key(lambda p, q: compare_rows(keys, vals[p], vals[q]) or (p > q) - (p < q)))


def after_cursor(order_by: Sequence[Any], last_row: dict) -> Callable[[dict], bool]:
    """Keyset pagination: the rows that sort strictly after ``last_row`` under ``order_by`` (what the tuple comparison
    the reference builds as its ``last_cursor`` clause selects, ``search_service.py:114``)."""
    keys = list(order_by)
    last = order_values(keys, last_row)
    return lambda row: compare_rows(keys, order_values(keys, row), last) > 0


def is_native(additional_data_to_select, additional_where_clause_list, order_by) -> bool:
    """True when every extra argument is one of this module's forms (None included)."""
    if additional_data_to_select is not None:
        if not isinstance(additional_data_to_select, dict):
            return False
        if not all(isinstance(v, LeastCosineDistance) or callable(v) for v in additional_data_to_select.values()):
            return False
    if additional_where_clause_list is not None:
        clauses = additional_where_clause_list if isinstance(additional_where_clause_list, (list, tuple)) else [additional_where_clause_list]
        if not all(callable(c) for c in clauses):
            return False
    if order_by is not None:
        try:
            for t in order_by:
                _as_key(t)
        except TypeError:
            return False
    return True
