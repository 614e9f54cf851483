"""CPU tier: ordering, NULL placement and keyset cursors of the plain forms ``HipVectorStore.query_by_keys`` takes in place
of SQL expressions (PostgreSQL semantics: ascending puts NULLs last, descending first; a cursor selects the rows that
sort strictly after the last row of the previous page)."""

import numpy as np
import pytest

from hirag_prod_amd import select_expr as se


ROWS = [
    {"type": "pdf", "page": 2, "y": 0.5, "i": 0},
    {"type": "pdf", "page": None, "y": 0.1, "i": 1},
    {"type": "md", "page": 1, "y": None, "i": 2},
    {"type": "pdf", "page": 2, "y": 0.9, "i": 3},
    {"type": "md", "page": 1, "y": 0.3, "i": 4},
    {"type": "pdf", "page": 1, "y": 0.5, "i": 5},
]


def _order(order_by):
    return [ROWS[p]["i"] for p in se.sort_rows(order_by, ROWS, list(range(len(ROWS))))]


def test_ascending_puts_nulls_last_and_descending_first():
    assert _order(["type", "page", "y"]) == [4, 2, 5, 0, 3, 1]
    assert _order(["type", se.Desc("page"), ("y", "desc")]) == [2, 4, 1, 3, 0, 5]
    # computed keys: coalesce(page, -1) as the reference orders pages (search_service.py:118)
    assert _order([se.Key(lambda r: r["page"] if r["page"] is not None else -1), "i"]) == [1, 2, 4, 5, 0, 3]


def test_equal_rows_keep_table_order():
    assert _order(["type"]) == [2, 4, 0, 1, 3, 5]


def test_cursor_pages_cover_every_row_once():
    order = ["type", se.Desc("page"), "y", "i"]
    full = se.sort_rows(order, ROWS, list(range(len(ROWS))))
    seen, cursor = [], None
    while True:
        live = [p for p in range(len(ROWS)) if cursor is None or cursor(ROWS[p])]
        page = se.sort_rows(order, ROWS, live)[:2]
        if not page:
            break
        seen.extend(page)
        cursor = se.after_cursor(order, ROWS[page[-1]])
    assert seen == full


def test_native_forms_are_recognised_and_others_are_not():
    e = np.ones((2, 4), dtype=np.float32)
    assert se.is_native({"d": se.LeastCosineDistance(e), "f": len}, [lambda r: True], ["a", ("b", "desc"), se.Desc("c")])
    assert se.is_native(None, None, None)
    assert not se.is_native({"d": object()}, None, None)
    assert not se.is_native(None, object(), None)
    assert not se.is_native(None, None, [object()])
    with pytest.raises(ValueError):
        se.LeastCosineDistance(np.zeros((0, 4)))
    assert se.LeastCosineDistance(np.ones(4)).embeddings.shape == (1, 4)
