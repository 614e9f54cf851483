"""``HipVectorStore``: drop-in for the reference's pgvector store on the dense-retrieval path.

It implements ``BaseVDB`` (``src/hirag_prod/storage/base_vdb.py:5-62``) with the argument meaning,
defaults, return shapes and errors of ``PGVector`` (``src/hirag_prod/storage/pgvector.py``), and is
injected with ``HiRAG.create(vdb=HipVectorStore.create(embedding_func))``.  Vectors live in HBM
(one ``HvsIndex`` per embedded table); the distance arithmetic, the LEAST fold, the threshold and
the ORDER BY ... LIMIT all run in the HIP kernels.  Row metadata (the non-vector columns) stays on
the host in plain Python structures: it is what PostgreSQL would return next to the distance and
takes no part in the arithmetic.

The graph tables and the personalised PageRank over them (``upsert_graph``, ``query_node``,
``pagerank_top_chunks_with_reset``, ``clean_table`` on Graph / Nodes) are served by a ``HipGraphStore``
(``graph_store.py``) this store owns, unless it was created with ``delegate=<PGVector instance>``, which
then receives those calls.  The extra arguments of ``query_by_keys`` (``additional_data_to_select``,
``additional_where_clause_list``, ``order_by``) are served in the plain forms of ``select_expr.py`` (the LEAST distance
column on the GPU, predicates and sort keys over the host rows); SQLAlchemy expression objects, which only a SQL
database can evaluate, go to the delegate or raise ``NotImplementedError``.
"""

from __future__ import annotations

import asyncio
import logging
import math
import os
import time
from collections.abc import Mapping
from datetime import datetime
from typing import Any, Callable, Dict, List, Literal, Optional, Union

import numpy as np

from . import select_expr
from .vdb_interface import (
    DEFAULT_DISTANCE_THRESHOLD,
    DEFAULT_QUERY_TOP_K,
    DEFAULT_QUERY_TOP_N,
    TABLES,
    BaseVDB,
)

logger = logging.getLogger(__name__)

_HALF_MAX = 65504.0
_MAX_FOLD = 8  # HVS_MAX_FOLD
_MAX_DIST_FOLD = 128  # HVS_MAX_DIST_FOLD: embeddings per call of the distance column / range search
_MAX_FILTER_URIS = 8192  # HVS_MAX_FILTER_URIS: uri codes a device-evaluated filter takes

_TRUE_WORDS = ("1", "true", "t", "yes", "y", "on")
_FALSE_WORDS = ("0", "false", "f", "no", "n", "off")


def env_use_half_vec(environ=None) -> bool:
    """``USE_HALF_VEC`` as the reference's settings model reads it (``configs/envs.py:12-13``: a pydantic bool that
    defaults to True), looked up when called, not at import time."""
    raw = (os.environ if environ is None else environ).get("USE_HALF_VEC")
    if raw is None or raw.strip() == "":
        return True
    word = raw.strip().lower()
    if word in _TRUE_WORDS:
        return True
    if word in _FALSE_WORDS:
        return False
    raise ValueError(f"USE_HALF_VEC={raw!r}: expected one of {_TRUE_WORDS + _FALSE_WORDS}")


def env_embedding_dimension(environ=None) -> Optional[int]:
    """``EMBEDDING_DIMENSION`` (``configs/envs.py:12``; the reference's column type takes its width from it,
    ``schema/vector_config.py:8-11``), or None when the variable is absent."""
    raw = (os.environ if environ is None else environ).get("EMBEDDING_DIMENSION")
    if raw is None or raw.strip() == "":
        return None
    dim = int(raw)
    if dim <= 0:
        raise ValueError(f"EMBEDDING_DIMENSION={raw!r}: expected a positive integer")
    return dim


class StoredVector(list):
    """A stored embedding as the reference's callers expect it: a list that also answers
    ``to_list()`` (``pgvector.HalfVector``) and ``tolist()`` (``numpy``), see
    ``src/hirag_prod/storage/query_service.py:216-220``."""

    def to_list(self):
        return list(self)

    def tolist(self):
        return list(self)


class _RowView(Mapping):
    """One metadata row of a ``_RowStore``, read in place: behaves like the row's dict (every column of the table is a
    key; a column the row was inserted without reads ``None``, which is what ``dict.get`` gave before)."""

    __slots__ = ("_cols", "_names", "_pos")

    def __init__(self, cols, names, pos):
        self._cols, self._names, self._pos = cols, names, pos

    def __getitem__(self, key):
        return self._cols[key][self._pos]

    def __iter__(self):
        return iter(self._names)

    def __len__(self):
        return len(self._names)


class _RowStore:
    """The metadata rows of a table, column by column: one Python list per column instead of one dict per row.  A dict
    of a dozen entries costs about 400 bytes beyond the values it points to, a column slot costs 8 -- at the 12.5M rows
    of one GPU's share of a 100M-row table that is 5 GB of host memory against 1.2 GB.  List-like over positions:
    ``rows[pos]`` is a read-only mapping (or ``None`` for a deleted row), ``rows[pos] = None`` deletes, ``append`` takes a
    dict; deleting drops the references the row held."""

    def __init__(self, columns):
        self.names = tuple(c for c in columns if c != "vector")
        self._cols = {c: [] for c in self.names}
        self._alive = bytearray()

    def __len__(self):
        return len(self._alive)

    def append(self, row) -> None:
        get = row.get if row is not None else (lambda c: None)
        for c, col in self._cols.items():
            col.append(get(c))
        self._alive.append(0 if row is None else 1)

    def extend(self, rows) -> None:
        for row in rows:
            self.append(row)

    def __getitem__(self, pos):
        if not self._alive[pos]:  # IndexError past the end, like a list
            return None
        return _RowView(self._cols, self.names, pos if pos >= 0 else pos + len(self._alive))

    def __setitem__(self, pos, row) -> None:
        self._alive[pos] = 0 if row is None else 1
        for c, col in self._cols.items():
            col[pos] = None if row is None else row.get(c)

    def __iter__(self):
        cols, names, alive = self._cols, self.names, self._alive
        for pos in range(len(alive)):
            yield _RowView(cols, names, pos) if alive[pos] else None

    def live_positions(self) -> np.ndarray:
        """Positions of the rows that are not deleted, ascending (int64) -- without building a view per row."""
        return np.flatnonzero(np.frombuffer(bytes(self._alive), dtype=np.uint8)).astype(np.int64)

    def column(self, name):
        """The column's list, position-aligned (entries of deleted rows are None)."""
        return self._cols[name]

    def take(self, positions) -> "_RowStore":
        """A new store holding the rows at ``positions``, in that order (compaction)."""
        out = _RowStore(self.names)
        for c, col in self._cols.items():
            out._cols[c] = [col[p] for p in positions]
        out._alive = bytearray(self._alive[p] for p in positions)
        return out


class _Table:
    """Host-side state of one table: metadata rows, primary-key map, filter columns."""

    def __init__(self, name: str):
        self.name = name
        self.columns, self.pk, self.has_vector = TABLES[name]
        self.colset = set(self.columns)
        self.rows = _RowStore(self.columns)  # position -> row mapping (None = deleted), stored by column
        self.pk_map: Dict[tuple, int] = {}
        self.key_index: Dict[str, Dict[Any, List[int]]] = {}
        self.index = None  # HvsIndex for embedded tables
        # filter columns as arrays, grown geometrically
        self.group = np.zeros(0, dtype=np.int32)
        self.ws_code = np.zeros(0, dtype=np.int32)
        self.kb_code = np.zeros(0, dtype=np.int32)
        self.uri_code = np.zeros(0, dtype=np.int32)
        self.private = np.zeros(0, dtype=bool)
        self.live = np.zeros(0, dtype=bool)
        # row positions change only in compact(): the epoch is odd while one runs and even otherwise; a search whose
        # epoch moved between its launch and its metadata look-up is run again (PostgreSQL gives the reference snapshot
        # isolation for the same situation)
        self.epoch = 0
        # Row positions on the device differ from the positions in `rows` once the table has been put in group order
        # (HvsIndex.partition): d2h[device position] = position here, h2d the inverse, both None while they are equal.
        # The index's ids are positions HERE, so what a search returns needs no translation; what goes to the index
        # by position (gather, tombstones, a host-built row mask, a distance column) goes through these.  Rows appended
        # after a partition sit at the same position on both sides (the maps cover the rows of the last partition).
        self.d2h: Optional[np.ndarray] = None
        self.h2d: Optional[np.ndarray] = None
        self.part_rows = 0       # rows the last partition put in group order
        self.part_failed_n = 0   # row count at which a partition last failed (device memory): not tried again below twice that

    def to_device(self, positions) -> np.ndarray:
        """positions in ``rows`` -> row positions of the index"""
        pos = np.asarray(positions, dtype=np.int64)
        if self.h2d is None or pos.size == 0:
            return pos
        out = pos.copy()
        inside = pos < len(self.h2d)
        out[inside] = self.h2d[pos[inside]]
        return out

    def host_order(self, column: np.ndarray) -> np.ndarray:
        """a per-row array in the index's row order -> the same in the order of ``rows``"""
        if self.h2d is None:
            return column
        out = column.copy()
        out[: len(self.h2d)] = column[self.h2d]
        return out

    def device_order(self, column: np.ndarray) -> np.ndarray:
        """a per-row array in the order of ``rows`` -> the same in the index's row order"""
        if self.d2h is None:
            return column
        out = column.copy()
        out[: len(self.d2h)] = column[self.d2h]
        return out

    # -- secondary indexes on scalar columns (documentKey, uri, ...): value -> ascending row positions.
    # Built on first use, kept current by note_insert / note_delete, dropped when positions change.
    def positions_where(self, column: str, values) -> Optional[List[int]]:
        """Live positions whose ``column`` is one of ``values``, ascending; None if the column holds
        unhashable values (the caller then scans)."""
        index = self.key_index.get(column)
        if index is None:
            index = {}
            try:
                for pos, row in enumerate(self.rows):
                    if row is not None:
                        index.setdefault(row.get(column), []).append(pos)
            except TypeError:
                return None
            self.key_index[column] = index
        hits: List[int] = []
        try:
            for v in values:
                hits.extend(index.get(v, ()))
        except TypeError:
            return None
        hits.sort()
        return hits

    def note_insert(self, pos: int, row: dict) -> None:
        for column, index in list(self.key_index.items()):
            try:
                index.setdefault(row.get(column), []).append(pos)
            except TypeError:
                del self.key_index[column]

    def note_delete(self, pos: int, row: dict) -> None:
        for column, index in list(self.key_index.items()):
            try:
                index[row.get(column)].remove(pos)
            except (KeyError, ValueError, TypeError):
                del self.key_index[column]

    def _grow(self, n: int) -> None:
        if n <= len(self.group):
            return
        cap = max(n, 2 * len(self.group), 1024)
        for name in ("group", "ws_code", "kb_code", "uri_code", "private", "live"):
            old = getattr(self, name)
            new = np.zeros(cap, dtype=old.dtype)
            new[: len(old)] = old
            setattr(self, name, new)


class _SearchCoalescer:
    """Concurrent ``query`` coroutines of one event loop share launches.

    The reference serves every query with its own SQL statement and runs many at once (25 concurrent
    queries in ``benchmark/2wiki/eval.py:170-180``).  On the GPU one pass over a table answers a batch
    at the cost of one query, so requests that agree on (table, workspace/knowledge-base code, top-k,
    number of folded embeddings, threshold) and carry no row bitmask are stacked into one
    ``hvs_search`` call.  There is no timer: the first request starts at once (after one turn of the
    loop), and whatever arrives while a launch is in flight forms the next batch.
    """

    MAX_BATCH = 1024  # requests per launch
    INLINE_BELOW = 0.4e-3  # seconds: a table whose last launch was shorter than this is served on the loop's own thread

    def __init__(self, run_batch: Callable):
        self._run_batch = run_batch  # (key, [q, ...]) -> [result, ...]; blocking
        self._pending: Dict[tuple, list] = {}
        self._worker: Optional[asyncio.Task] = None
        self._loop = None
        self._last_launch: Dict[Any, float] = {}  # table -> seconds its last launch took
        self.launches = 0
        self.requests = 0
        self.inline_launches = 0

    async def submit(self, key: tuple, q: np.ndarray):
        loop = asyncio.get_running_loop()
        if loop is not self._loop:  # a new event loop (tests run one per case): nothing of the old one is left to serve
            self._loop, self._pending, self._worker = loop, {}, None
        fut = loop.create_future()
        self._pending.setdefault(key, []).append((q, fut))
        if self._worker is None or self._worker.done():
            self._worker = loop.create_task(self._drain())
        return await fut

    async def _drain(self):
        await asyncio.sleep(0)  # the coroutines that are ready right now join the first batch
        while self._pending:
            # one batch per key; the batches of one round run side by side (tables have their own HIP streams, and
            # launches on one table queue behind that table's lock)
            round_ = []
            for key in list(self._pending):
                waiting = self._pending[key]
                batch, rest = waiting[: self.MAX_BATCH], waiting[self.MAX_BATCH :]
                if rest:
                    self._pending[key] = rest
                else:
                    del self._pending[key]
                round_.append((key, batch))
            await asyncio.gather(*[self._serve(key, batch) for key, batch in round_])

    async def _serve(self, key: tuple, batch: list) -> None:
        def timed(qs):
            t0 = time.perf_counter()
            out = self._run_batch(key, qs)
            self._last_launch[key[0]] = time.perf_counter() - t0
            return out

        try:
            # A launch normally runs in a thread, so that the loop keeps taking requests (they form the next batch).  Handing
            # a launch to a thread and hearing back costs about 0.2 ms, though: a search over a small table or a group's row
            # range is shorter than that, and blocks the loop for less than the hand-over would take.
            if self._last_launch.get(key[0], 1.0) < self.INLINE_BELOW:
                results = timed([q for q, _ in batch])
                self.inline_launches += 1
            else:
                results = await asyncio.to_thread(timed, [q for q, _ in batch])
            self.launches += 1
            self.requests += len(batch)
            for (_, fut), res in zip(batch, results):
                if not fut.done():
                    fut.set_result(res)
        except BaseException as exc:  # every waiter of the batch sees the failure; the other batches are not affected
            for _, fut in batch:
                if not fut.done():
                    fut.set_exception(exc if isinstance(exc, Exception) else RuntimeError("search abandoned: the event loop is shutting down"))
            if not isinstance(exc, Exception):
                raise


class HipVectorStore(BaseVDB):
    """GPU-resident vector store with the reference's ``PGVector`` behaviour on the retrieval path."""

    def __init__(
        self,
        embedding_func: Optional[Callable],
        vector_type: Optional[Literal["vector", "halfvec", "bf16"]] = None,
        embedding_dimension: Optional[int] = None,
        device: int = 0,
        delegate: Any = None,
        tokenizer: Optional[Callable[[str], tuple]] = None,
        translator: Optional[Callable[[str], Any]] = None,
        index_factory: Optional[Callable[[int, str, int], Any]] = None,
        graph_factory: Optional[Callable[..., Any]] = None,
        coalesce_queries: bool = True,
        search_copy: bool = True,
        auto_partition: bool = True,
        partition_min_rows: int = 32768,
        partition_tail_fraction: float = 0.25,
    ):
        """``search_copy``: a ``vector`` (fp32) table serves batched searches, single queries on large tables and
        k-means from a second, fp16-slab copy of its rows on the 16-bit matrix pipe -- several times faster, and as
        much device memory again as the rows.  False keeps every table at its own size (fp32 kernels only).  The
        store logs one line per table when such a copy appears.

        ``auto_partition``: every ``query`` of the reference filters by (workspaceId, knowledgeBaseId)
        (``pgvector.py:577-580``); a table kept in the order of that pair answers the filter with a row range
        (``partition``).  With this set, a table is put in that order when the rows added since the last time are at
        least ``partition_min_rows`` and at least ``partition_tail_fraction`` of the ordered rows: the copying stays
        proportional to the table's growth (a quarter: every row moves about five times while a table grows, 27 ms per
        move for 12.5M x 768 bf16 rows), and a filtered search never walks a tail longer than that fraction of the
        table with the per-row test.  ``prepare`` orders a table whatever the size.

        ``vector_type`` None: the environment decides, as it does for the reference's column type
        (``USE_HALF_VEC``, true unless set otherwise -> ``halfvec``; ``schema/vector_config.py:8-11``).
        ``embedding_dimension`` None: ``EMBEDDING_DIMENSION`` when the variable is set (read when the first index is
        made), else the width of the first vector stored."""
        if vector_type is None:
            vector_type = "halfvec" if env_use_half_vec() else "vector"
        if vector_type not in ("vector", "halfvec", "bf16"):
            raise ValueError(f"vector_type must be 'vector', 'halfvec' or 'bf16', got {vector_type!r}")
        self.embedding_func = embedding_func
        self.vector_type = vector_type
        self.dtype = {"vector": "f32", "halfvec": "f16", "bf16": "bf16"}[vector_type]
        self.embedding_dimension = embedding_dimension
        self.device = device
        self.delegate = delegate
        self.tokenizer = tokenizer
        self.translator = translator
        self._index_factory = index_factory
        self._graph_factory = graph_factory
        self._graph_store = None
        self.tables: Dict[str, _Table] = {name: _Table(name) for name in TABLES}
        self._ws_codes: Dict[str, int] = {}
        self._kb_codes: Dict[str, int] = {}
        self._uri_codes: Dict[str, int] = {}
        self._group_codes: Dict[tuple, int] = {}
        self._lock = asyncio.Lock()
        self._coalescer = _SearchCoalescer(self._search_batch) if coalesce_queries else None
        self.search_copy = bool(search_copy)
        self._copy_logged: Dict[str, int] = {}
        self.auto_partition = bool(auto_partition)
        self.partition_min_rows = int(partition_min_rows)
        self.partition_tail_fraction = float(partition_tail_fraction)

    # the reference's factory convention: PGVector.create(embedding_func, vector_type) (pgvector.py:70-77)
    @classmethod
    def create(cls, embedding_func: Optional[Callable], vector_type: Optional[str] = None, **kwargs) -> "HipVectorStore":
        return cls(embedding_func, vector_type=vector_type, **kwargs)

    # -- helpers -----------------------------------------------------------------------------------
    def get_model(self, table_name: str) -> _Table:
        table = self.tables.get(table_name)
        if not table:
            if self.delegate is not None and table_name in getattr(self.delegate, "tables", {}):
                raise NotImplementedError(f"table {table_name} is served by the delegate store")
            raise ValueError(f"No table found for table {table_name}")  # pgvector.py:63-67
        return table

    def _make_index(self, dim: int):
        if self._index_factory is not None:
            return self._index_factory(dim, self.dtype, self.device)
        from .index import HvsIndex  # fails loudly when libhvs.so or the GPU is missing

        return HvsIndex(dim, self.dtype, device=self.device)

    def _ensure_index(self, table: _Table, dim: int):
        if table.index is None:
            if self.embedding_dimension is None:
                self.embedding_dimension = env_embedding_dimension() or dim
            table.index = self._make_index(self.embedding_dimension)
            if not self.search_copy and hasattr(table.index, "search_copy"):
                table.index.search_copy(False)
        return table.index

    def _note_search_copy(self, table: _Table) -> None:
        """One log line when a table's search-side copy appears or grows to another size class (nothing here is silent
        about doubling a table's device memory)."""
        if self.dtype != "f32" or table.index is None or not hasattr(table.index, "search_copy"):
            return
        held = table.index.search_copy()
        if held and held != self._copy_logged.get(table.name):
            self._copy_logged[table.name] = held
            logger.info("[search copy] table '%s': %.2f GB of device memory hold the fp16-slab copy of its fp32 rows "
                        "(HipVectorStore(search_copy=False) keeps tables at their own size)", table.name, held / 1e9)

    @staticmethod
    def _code(mapping: Dict[Any, int], key: Any) -> int:
        code = mapping.get(key)
        if code is None:
            code = len(mapping)
            mapping[key] = code
        return code

    @staticmethod
    def _to_list(embedding):
        return embedding.tolist() if hasattr(embedding, "tolist") else list(embedding)

    async def _init_vdb(self, embedding_dimension: Optional[int] = None, *args, **kwargs):
        """Create the embedded tables' indexes (``PGVector._init_vdb``, pgvector.py:704-726)."""
        if embedding_dimension is not None:
            self.embedding_dimension = int(embedding_dimension)
        if self.embedding_dimension is None:
            self.embedding_dimension = env_embedding_dimension()
        if self.embedding_dimension is None:
            return
        for table in self.tables.values():
            if table.has_vector:
                await asyncio.to_thread(self._ensure_index, table, self.embedding_dimension)

    # -- ingest ------------------------------------------------------------------------------------
    async def upsert_texts(
        self,
        texts_to_upsert: List[str],
        properties_list: List[dict],
        table_name: str,
        with_tokenization: bool = False,
        with_translation: bool = False,
        mode: Literal["append", "overwrite"] = "append",
    ):
        """Embed and insert; rows whose primary key already exists are left untouched
        (``INSERT ... ON CONFLICT DO NOTHING``, pgvector.py:80-187)."""
        if len(texts_to_upsert) != len(properties_list):
            raise ValueError("texts_to_upsert and properties_list must have the same length")
        table = self.get_model(table_name)
        if not table.has_vector:
            raise ValueError(f"table {table_name} has no vector column")
        start = time.perf_counter()
        embs = await self.embedding_func(texts_to_upsert)
        now = datetime.now()
        rows: List[dict] = []
        for i in range(len(properties_list)):
            row = dict(properties_list[i] or {})
            if with_tokenization and self.tokenizer is not None:
                (row["token_list"], row["token_start_index_list"], row["token_end_index_list"]) = self.tokenizer(
                    texts_to_upsert[i]
                )
            if with_translation and self.translator is not None:
                translated = self.translator(texts_to_upsert[i])
                if asyncio.iscoroutine(translated):
                    translated = await translated
                row["translation"] = getattr(translated, "text", translated)
                if with_tokenization and self.tokenizer is not None:
                    (
                        row["translation_token_list"],
                        row["translation_token_start_index_list"],
                        row["translation_token_end_index_list"],
                    ) = self.tokenizer(row["translation"])
            row["vector"] = self._to_list(embs[i])
            row["updatedAt"] = now
            rows.append({k: v for k, v in row.items() if k in table.colset})
        if not rows:
            logger.info(f"[upsert_texts] No rows to upsert into '{table_name}', mode={mode}, elapsed={time.perf_counter() - start:.3f}s")
            return rows

        await self._insert_rows(table, table_name, rows)
        logger.info("[upsert_texts] Upserted %d into '%s', mode=%s, elapsed=%.3fs", len(rows), table_name, mode,
                    time.perf_counter() - start)
        return rows

    async def load_rows(self, table_name: str, rows, vector_key: str = "vector_float_array") -> int:
        """Bulk ingest of rows that already carry their embedding: the sync-from-PostgreSQL path.  ``rows``
        is an iterable of mappings as ``dict(orm_row)`` of the reference's models gives them; the vector is
        taken from ``vector_float_array`` (the ``cast(vector, ARRAY(Float(4)))`` column property,
        ``src/hirag_prod/schema/chunk.py:46-48``, ``item.py:59``, ``triplets.py:26``), else from ``vector``
        (a list, a numpy array or a pgvector ``HalfVector`` / ``Vector`` with ``to_list()``).  Same rules
        as ``upsert_texts`` otherwise (first write wins on the primary key, type rounding, range checks);
        no embedding call.  Returns the number of rows offered."""
        table = self.get_model(table_name)
        if not table.has_vector:
            raise ValueError(f"table {table_name} has no vector column")
        prepared: List[dict] = []
        for src in rows:
            row = dict(src)
            vec = row.pop(vector_key, None)
            if vec is None:
                vec = row.get("vector")
            if vec is None:
                raise ValueError(f"row without a vector ({vector_key!r} or 'vector') for table {table_name}")
            row["vector"] = vec.to_list() if hasattr(vec, "to_list") else self._to_list(vec)
            prepared.append({k: v for k, v in row.items() if k in table.colset})
        if prepared:
            await self._insert_rows(table, table_name, prepared)
        return len(prepared)

    async def _insert_rows(self, table: _Table, table_name: str, rows: List[dict]) -> None:
        """Append the rows whose primary key is new (``ON CONFLICT DO NOTHING``) to the host table and the index."""
        async with self._lock:
            new_vecs, new_rows = [], []
            seen_in_batch = set()
            for row in rows:
                pk = tuple(row.get(c) for c in table.pk)
                if any(v is None for v in pk):
                    raise ValueError(f"null value in primary key {table.pk} of table {table_name}")
                if pk in table.pk_map or pk in seen_in_batch:
                    continue  # first write wins
                vec = np.asarray(row["vector"], dtype=np.float64)
                dim = self.embedding_dimension or env_embedding_dimension() or vec.shape[0]
                if vec.ndim != 1 or vec.shape[0] != dim:
                    raise ValueError(f"expected {dim} dimensions, not {vec.shape[0] if vec.ndim == 1 else vec.shape}")
                if not np.all(np.isfinite(vec)):
                    raise ValueError("NaN or infinite value not allowed in vector")
                if self.dtype == "f16" and np.any(np.abs(vec) > _HALF_MAX):
                    raise ValueError("value out of range for type halfvec")
                seen_in_batch.add(pk)
                new_vecs.append(vec.astype(np.float32))
                new_rows.append((pk, row))
            if new_rows:
                dim = len(new_vecs[0])
                index = self._ensure_index(table, dim)
                first = len(table.rows)
                n_new = len(new_rows)
                groups = np.empty(n_new, dtype=np.int32)
                attrs = np.empty((4, n_new), dtype=np.int32)
                for j, (pk, row) in enumerate(new_rows):
                    ws, kb = row.get("workspaceId"), row.get("knowledgeBaseId")
                    groups[j] = self._code(self._group_codes, (ws, kb))
                    attrs[0, j] = self._code(self._ws_codes, ws)
                    attrs[1, j] = self._code(self._kb_codes, kb)
                    attrs[2, j] = self._code(self._uri_codes, row.get("uri"))
                    attrs[3, j] = bool(row.get("private", False))
                # the device first: if it refuses the rows (out of memory, a dimension mismatch) nothing on the host has
                # changed yet, the keys stay absent and the caller's retry inserts them -- as a rolled-back INSERT would
                lo, hi = await asyncio.to_thread(self._add_rows, index, np.stack(new_vecs), groups, attrs)
                if lo != first or hi != first + n_new:
                    raise RuntimeError(f"table {table_name}: the index put the rows at [{lo}, {hi}), the host table expected "
                                       f"[{first}, {first + n_new})")
                table._grow(first + n_new)
                table.group[first:hi] = groups
                table.ws_code[first:hi], table.kb_code[first:hi], table.uri_code[first:hi] = attrs[0], attrs[1], attrs[2]
                table.private[first:hi] = attrs[3] != 0
                table.live[first:hi] = True
                for j, (pk, row) in enumerate(new_rows):
                    stored = dict(row)
                    stored.pop("vector", None)  # the vector lives in HBM
                    table.rows.append(stored)
                    table.pk_map[pk] = first + j
                    table.note_insert(first + j, stored)
                tail = len(table.rows) - table.part_rows
                if (self.auto_partition and len(self._group_codes) > 1 and tail >= max(self.partition_min_rows, int(table.part_rows * self.partition_tail_fraction))
                        and len(table.rows) >= 2 * table.part_failed_n and hasattr(index, "partition")):
                    try:
                        await self._partition_table(table)
                    except Exception as exc:  # noqa: BLE001 -- the rows are in; the table merely stays in arrival order
                        table.part_failed_n = len(table.rows)
                        logger.warning("[partition] table '%s' stays in arrival order: %s", table_name, exc)

    @staticmethod
    def _add_rows(index, mat, groups, attrs):
        lo, hi = index.add(mat, None, groups)
        index.set_attributes(lo, workspace=attrs[0], knowledge_base=attrs[1], uri=attrs[2], private=attrs[3])
        return lo, hi

    async def upsert_file(self, file, table_name: str = "Files", mode: str = "append"):
        """Insert one Files row unless its primary key exists (pgvector.py:475-503)."""
        table = self.get_model(table_name)
        row = dict(file)
        row["updatedAt"] = datetime.now()
        row = {k: v for k, v in row.items() if k in table.colset}
        pk = tuple(row.get(c) for c in table.pk)
        async with self._lock:
            if pk not in table.pk_map:
                pos = len(table.rows)
                table._grow(pos + 1)
                table.ws_code[pos] = self._code(self._ws_codes, row.get("workspaceId"))
                table.kb_code[pos] = self._code(self._kb_codes, row.get("knowledgeBaseId"))
                table.uri_code[pos] = self._code(self._uri_codes, row.get("uri"))
                table.private[pos] = bool(row.get("private", False))
                table.live[pos] = True
                table.rows.append(row)
                table.pk_map[pk] = pos
                table.note_insert(pos, row)
        return row

    async def clean_table(self, table_name: str, where: Dict[str, Any]) -> bool:
        """Delete every row matching all ``where`` items; True if any row went (pgvector.py:452-473)."""
        try:
            table = self.get_model(table_name)
        except NotImplementedError:
            return await self.delegate.clean_table(table_name=table_name, where=where)
        except ValueError:
            if table_name in ("Graph", "Nodes"):
                if self.delegate is not None:
                    return await self.delegate.clean_table(table_name=table_name, where=where)
                return await self.graph_store.clean_table(table_name, where)
            raise
        unknown = [k for k in where if k not in table.colset]
        if unknown:
            raise AttributeError(f"table {table_name} has no column {unknown[0]!r}")
        async with self._lock:
            candidates = None
            for k, v in where.items():  # narrow by the first condition that has (or can get) an index
                candidates = table.positions_where(k, [v])
                if candidates is not None:
                    break
            if candidates is None:
                candidates = range(len(table.rows))
            hit = [
                pos
                for pos in candidates
                if table.rows[pos] is not None and all(table.rows[pos].get(k) == v for k, v in where.items())
            ]
            if hit:
                # tombstones on the device first: from then on no search returns these positions, and a search already
                # under way that does finds the host rows gone and skips them
                if table.index is not None:
                    await asyncio.to_thread(table.index.set_deleted, table.to_device(hit), True)
                for pos in hit:
                    row = table.rows[pos]
                    table.pk_map.pop(tuple(row.get(c) for c in table.pk), None)
                    table.note_delete(pos, row)
                    table.rows[pos] = None
                    table.live[pos] = False
        return len(hit) != 0

    # -- query -------------------------------------------------------------------------------------
    def _row_filter(self, table: _Table, workspace_id, knowledge_base_id, uri_list, require_access):
        """-> (filter, empty?) mirroring the WHERE clauses of pgvector.py:573-580.  ``filter`` is a hashable tuple
        (group, workspace, knowledge_base, access, uris) of codes that ``_filter_dict`` turns into what
        ``HvsIndex.search(filter=...)`` takes: every clause is evaluated on the GPU over the attribute columns."""
        group = ws = kb = -1
        access = None
        uris = None
        has = table.colset.__contains__
        use_ws = bool(workspace_id) and has("workspaceId")
        use_kb = bool(knowledge_base_id) and has("knowledgeBaseId")
        if use_ws and use_kb:
            group = self._group_codes.get((workspace_id, knowledge_base_id))
            if group is None:
                return None, True
        elif use_ws or use_kb:
            code = (self._ws_codes.get(workspace_id) if use_ws else self._kb_codes.get(knowledge_base_id))
            if code is None:
                return None, True
            if use_ws:
                ws = code
            else:
                kb = code
        if uri_list and has("uri"):
            uris = tuple(sorted({self._uri_codes[u] for u in uri_list if u in self._uri_codes}))
            if not uris:
                return None, True
        if require_access is not None and has("private"):
            access = require_access == "private"
        return (group, ws, kb, access, uris), False

    @staticmethod
    def _filter_dict(spec):
        group, ws, kb, access, uris = spec
        if ws < 0 and kb < 0 and access is None and uris is None:
            return None, group  # the pair code alone: the kernels' own group test
        out = {"group": group, "workspace": ws, "knowledge_base": kb}
        if access is not None:
            out["access"] = access
        if uris is not None:
            out["uris"] = uris
        return out, -1

    async def query(
        self,
        query: Union[str, List[str]],
        workspace_id: str,
        knowledge_base_id: str,
        table_name: str,
        topk: Optional[int] = None,
        topn: Optional[int] = None,
        uri_list: Optional[List[str]] = None,
        require_access: Optional[Literal["private", "public"]] = None,
        columns_to_select: Optional[List[str]] = None,
        distance_threshold: Optional[float] = None,
        rerank: bool = False,
    ) -> List[dict]:
        """``PGVector.query`` (pgvector.py:520-604): distance = LEAST over the query strings'
        embeddings of the cosine distance; filters; ``distance < threshold`` (0.8 whenever the
        argument is falsy); ascending; first ``topk`` (10 by default)."""
        if isinstance(query, str):
            query = [query]
        topk = topk if topk else DEFAULT_QUERY_TOP_K
        topn = topn if topn else DEFAULT_QUERY_TOP_N
        distance_threshold = distance_threshold if distance_threshold else DEFAULT_DISTANCE_THRESHOLD
        if topn > topk:
            raise ValueError(f"topn ({topn}) must be <= topk ({topk})")
        table = self.get_model(table_name)
        if not table.has_vector:
            raise AttributeError(f"table {table_name} has no vector column")
        if columns_to_select is None:
            columns_to_select = [c for c in table.columns if c != "vector"]
        start = time.perf_counter()
        q_embs = await self.embedding_func(query)
        q = np.asarray([self._to_list(e) for e in q_embs], dtype=np.float32)
        if q.ndim != 2 or q.shape[0] == 0:
            return []

        scored: List[dict] = []
        if table.index is not None and len(table.rows):
            spec, empty = self._row_filter(table, workspace_id, knowledge_base_id, uri_list, require_access)
            want_vec = "vector" in columns_to_select
            while not empty:
                while table.epoch & 1:  # a compaction is renumbering the rows
                    await asyncio.sleep(0.001)
                epoch = table.epoch
                many_uris = spec[4] is not None and len(spec[4]) > _MAX_FILTER_URIS
                if self._coalescer is not None and q.shape[0] <= _MAX_FOLD and not many_uris:
                    key = (table_name, spec, int(topk), int(q.shape[0]), float(distance_threshold))
                    pairs = await self._coalescer.submit(key, q)
                else:
                    pairs = await asyncio.to_thread(self._search, table, q, int(topk), spec, float(distance_threshold))
                vecs = None
                if want_vec and pairs:
                    vecs = await asyncio.to_thread(table.index.gather, table.to_device([p for p, _ in pairs]))
                if table.epoch != epoch:
                    continue  # positions were renumbered under this search: its answer names other rows now
                for j, (pos, dist) in enumerate(pairs):
                    row = table.rows[pos] if pos < len(table.rows) else None
                    if row is None:
                        continue  # deleted between the search and this look-up
                    payload = {}
                    for col in columns_to_select:
                        if col == "vector":
                            payload[col] = StoredVector(vecs[j].tolist())
                        elif col in table.colset:
                            payload[col] = row.get(col)
                    payload["distance"] = dist
                    scored.append(payload)
                break
        logger.info(f"[query] Retrieved {len(scored)} records from '{table_name}', elapsed={time.perf_counter() - start:.3f}s")
        return scored

    def _search_batch(self, key: tuple, queries: List[np.ndarray]) -> List[List[tuple]]:
        """Blocking part of coalesced ``query`` calls: requests with the same key in one launch."""
        table_name, spec, topk, fold, threshold = key
        index = self.tables[table_name].index
        filt, group = self._filter_dict(spec)
        dist, ids, cnt = index.search(np.concatenate(queries, axis=0), k=topk, fold=fold, group_filter=group, filter=filt,
                                      max_distance=threshold)
        self._note_search_copy(self.tables[table_name])
        return [[(int(ids[b][i]), float(dist[b][i])) for i in range(int(cnt[b]))] for b in range(len(queries))]

    def _host_mask(self, table: _Table, spec):
        """Row bitmask built on the host: only for a uri list longer than the device's list buffer holds."""
        group, ws, kb, access, uris = spec
        n = len(table.rows)
        keep = np.isin(table.uri_code[:n], np.asarray(uris, dtype=np.int32))
        if ws >= 0:
            keep &= table.ws_code[:n] == ws
        if kb >= 0:
            keep &= table.kb_code[:n] == kb
        if access is not None:
            keep &= table.private[:n] == access
        bits = np.packbits(table.device_order(keep), bitorder="little")  # the index tests bit (its own row position)
        pad = (-len(bits)) % 4
        if pad:
            bits = np.concatenate([bits, np.zeros(pad, dtype=np.uint8)])
        return bits.view(np.uint32)

    def _search(self, table: _Table, q: np.ndarray, topk: int, spec, threshold: float):
        """Blocking part of ``query``: one fused LEAST-fold search (or several for > 8 embeddings)."""
        index = table.index
        m = q.shape[0]
        if spec[4] is not None and len(spec[4]) > _MAX_FILTER_URIS:
            kw = {"group_filter": spec[0], "row_mask": self._host_mask(table, spec)}
        else:
            filt, group = self._filter_dict(spec)
            kw = {"group_filter": group, "filter": filt}
        if m <= _MAX_FOLD:
            dist, ids, cnt = index.search(q, k=topk, fold=m, max_distance=threshold, **kw)
            c = int(cnt[0])
            return [(int(ids[0][i]), float(dist[0][i])) for i in range(c)]
        # more embeddings than one launch folds: every row of the final top-k is inside the top-k of
        # the group of embeddings that realises its minimum, so per-row minima over the groups' lists
        # followed by a sort is exact
        best: Dict[int, float] = {}
        for s in range(0, m, _MAX_FOLD):
            part = q[s : s + _MAX_FOLD]
            dist, ids, cnt = index.search(part, k=topk, fold=part.shape[0], max_distance=threshold, **kw)
            for i in range(int(cnt[0])):
                pos, d = int(ids[0][i]), float(dist[0][i])
                if pos not in best or d < best[pos]:
                    best[pos] = d
        return sorted(best.items(), key=lambda t: (t[1], t[0]))[:topk]

    async def query_by_keys(
        self,
        key_value: List[str],
        workspace_id: str,
        knowledge_base_id: str,
        table_name: str,
        key_column: str = "documentKey",
        columns_to_select: Optional[List[str]] = None,
        additional_data_to_select: Optional[Dict[str, Any]] = None,
        additional_where_clause_list: Optional[Any] = None,
        order_by: Optional[List[Any]] = None,
        limit: Optional[int] = None,
    ) -> List[dict]:
        """Rows by key (pgvector.py:606-660).  Selecting ``"vector"`` returns the stored embedding
        (widened to float) as a ``StoredVector``."""
        extras = additional_data_to_select is not None or additional_where_clause_list is not None or order_by is not None
        if extras and not select_expr.is_native(additional_data_to_select, additional_where_clause_list, order_by):
            if self.delegate is not None:
                return await self.delegate.query_by_keys(
                    key_value, workspace_id, knowledge_base_id, table_name, key_column, columns_to_select,
                    additional_data_to_select, additional_where_clause_list, order_by, limit)
            raise NotImplementedError(
                "query_by_keys: SQL expression objects need a delegate store; this store evaluates the forms of "
                "hirag_prod_amd.select_expr (LeastCosineDistance, row predicates, Key / Desc / column names)")
        if extras:
            return await self._query_by_keys_with_expressions(
                key_value, workspace_id, knowledge_base_id, table_name, key_column, columns_to_select,
                additional_data_to_select or {}, additional_where_clause_list, order_by, limit)
        table = self.get_model(table_name)
        if key_column not in table.colset:
            raise AttributeError(f"table {table_name} has no column {key_column!r}")
        if columns_to_select is None:
            columns_to_select = [c for c in table.columns if c != "vector"]
        wanted = set(key_value) if key_value else None
        use_ws = bool(workspace_id) and "workspaceId" in table.colset
        use_kb = bool(knowledge_base_id) and "knowledgeBaseId" in table.colset
        hits: List[int] = []
        candidates = table.positions_where(key_column, wanted) if wanted is not None else None
        for pos in (candidates if candidates is not None else range(len(table.rows))):
            row = table.rows[pos]
            if row is None:
                continue
            if wanted is not None and row.get(key_column) not in wanted:
                continue
            if use_ws and row.get("workspaceId") != workspace_id:
                continue
            if use_kb and row.get("knowledgeBaseId") != knowledge_base_id:
                continue
            hits.append(pos)
            if limit is not None and len(hits) >= limit:
                break
        vecs = None
        if "vector" in columns_to_select and table.has_vector and hits and table.index is not None:
            vecs = await asyncio.to_thread(table.index.gather, table.to_device(hits))
        out: List[dict] = []
        for j, pos in enumerate(hits):
            row = table.rows[pos]
            rec = {}
            for col in columns_to_select:
                if col == "vector" and table.has_vector:
                    rec[col] = StoredVector(vecs[j].tolist()) if vecs is not None else None
                elif col in table.colset:
                    rec[col] = row.get(col)
            out.append(rec)
        return out

    async def _query_by_keys_with_expressions(self, key_value, workspace_id, knowledge_base_id, table_name, key_column,
                                              columns_to_select, extra_columns, where, order_by, limit) -> List[dict]:
        """``query_by_keys`` with extra selected columns, extra predicates and an ordering (pgvector.py:620-660: the
        statement gains select entities, ``.where``, ``.order_by`` and ``.limit`` in that order).  A
        ``LeastCosineDistance`` column is one pass of the distance-column kernel over the table
        (``search_service.py:100-113``); predicates and sort keys see the metadata row."""
        table = self.get_model(table_name)
        if key_column not in table.colset:
            raise AttributeError(f"table {table_name} has no column {key_column!r}")
        if columns_to_select is None:
            columns_to_select = [c for c in table.columns if c != "vector"]
        clauses = [] if where is None else (list(where) if isinstance(where, (list, tuple)) else [where])
        wanted = set(key_value) if key_value else None
        use_ws = bool(workspace_id) and "workspaceId" in table.colset
        use_kb = bool(knowledge_base_id) and "knowledgeBaseId" in table.colset
        while True:
            while table.epoch & 1:
                await asyncio.sleep(0.001)
            epoch = table.epoch
            candidates = table.positions_where(key_column, wanted) if wanted is not None else None
            hits: List[int] = []
            for pos in (candidates if candidates is not None else range(len(table.rows))):
                row = table.rows[pos]
                if row is None:
                    continue
                if wanted is not None and row.get(key_column) not in wanted:
                    continue
                if use_ws and row.get("workspaceId") != workspace_id:
                    continue
                if use_kb and row.get("knowledgeBaseId") != knowledge_base_id:
                    continue
                if any(not clause(row) for clause in clauses):
                    continue
                hits.append(pos)
            if order_by is not None:
                hits = select_expr.sort_rows(order_by, table.rows, hits)
            if limit is not None:
                hits = hits[:limit]
            computed: Dict[str, Any] = {}
            for name, expr in extra_columns.items():
                if isinstance(expr, select_expr.LeastCosineDistance):
                    col = None
                    if table.has_vector and table.index is not None and hits:
                        q = expr.embeddings
                        for s in range(0, q.shape[0], _MAX_DIST_FOLD):
                            part = await asyncio.to_thread(table.index.distances, q[s : s + _MAX_DIST_FOLD])
                            col = part if col is None else np.fmin(col, part)  # fmin: NaN only if both are NaN
                    computed[name] = None if col is None else table.host_order(col)
            if table.epoch == epoch:
                break
        vecs = None
        if "vector" in columns_to_select and table.has_vector and hits and table.index is not None:
            vecs = await asyncio.to_thread(table.index.gather, table.to_device(hits))
        out: List[dict] = []
        for j, pos in enumerate(hits):
            row = table.rows[pos]
            if row is None:  # deleted since the selection
                continue
            rec = {}
            for col in columns_to_select:
                if col == "vector" and table.has_vector:
                    rec[col] = StoredVector(vecs[j].tolist()) if vecs is not None else None
                elif col in table.colset:
                    rec[col] = row.get(col)
            for name, expr in extra_columns.items():
                if isinstance(expr, select_expr.LeastCosineDistance):
                    d = float(computed[name][pos]) if computed[name] is not None else float("nan")
                    rec[name] = None if math.isnan(d) else d
                else:
                    rec[name] = expr(row)
            out.append(rec)
        return out

    async def upsert_file_from_metadata(self, metadata: dict, table_name: str = "Files", mode: str = "append", **kwargs):
        """A Files row from a metadata dict plus keyword columns, then ``upsert_file`` (pgvector.py:505-518; the
        reference builds the row with a ``create_object`` helper its class does not define, so its own call raises --
        here the row is the dict itself, restricted to the table's columns by ``upsert_file``)."""
        if self.delegate is not None and table_name not in self.tables:
            return await self.delegate.upsert_file_from_metadata(metadata, table_name, mode, **kwargs)
        row = dict(metadata or {})
        row.update(kwargs)
        return await self.upsert_file(row, table_name, mode)

    async def get_existing_document_keys(self, uri: str, workspace_id: str, knowledge_base_id: str, table_name: str) -> List[str]:
        """``documentKey`` of every row with this uri / workspace / knowledge base (pgvector.py:662-683)."""
        table = self.get_model(table_name)
        candidates = table.positions_where("uri", [uri]) if "uri" in table.colset else None
        rows = table.rows if candidates is None else [table.rows[pos] for pos in candidates]
        return [
            row.get("documentKey")
            for row in rows
            if row is not None
            and row.get("uri") == uri
            and row.get("workspaceId") == workspace_id
            and row.get("knowledgeBaseId") == knowledge_base_id
            and row.get("documentKey")
        ]

    async def get_table(self, table_name: str) -> List[dict]:
        """Every live row with every column, vectors as lists (pgvector.py:685-702)."""
        table = self.get_model(table_name)
        live = table.rows.live_positions().tolist()
        vecs = None
        if table.has_vector and table.index is not None and live:
            vecs = await asyncio.to_thread(table.index.gather, table.to_device(live))
        out = []
        for j, pos in enumerate(live):
            rec = {col: table.rows[pos].get(col) for col in table.columns if col != "vector"}
            if table.has_vector:
                rec["vector"] = vecs[j].tolist() if vecs is not None else None
            out.append(rec)
        return out

    # -- the full distance column (cross-language search) ------------------------------------------
    async def query_distance_column(self, embeddings, workspace_id: str, knowledge_base_id: str,
                                    table_name: str = "Items", columns_to_select: Optional[List[str]] = None,
                                    column_name: str = "search_sentence_cosine_distance",
                                    max_distance: Optional[float] = None) -> List[dict]:
        """Every row of one knowledge base with ``LEAST_i(vector <=> e_i)`` as an extra column: what
        ``cross_language_search`` selects through ``additional_data_to_select``
        (``src/hirag_prod/cross_language_search/search_service.py:100-113``), without ordering or limit.
        With ``max_distance`` only the rows whose column is below it come back (the rows
        ``cross_language_search/functions.py:330-340`` keeps with 0.4), selected on the GPU, so the
        column never crosses to the host."""
        table = self.get_model(table_name)
        if columns_to_select is None:
            columns_to_select = [c for c in table.columns if c != "vector"]
        q = np.asarray([self._to_list(e) for e in embeddings], dtype=np.float32)
        if table.index is None or not len(table.rows) or q.size == 0:
            return []
        if max_distance is not None:
            spec, empty = self._row_filter(table, workspace_id, knowledge_base_id, None, None)
            if empty:
                return []
            filt, group = self._filter_dict(spec)
            while True:
                while table.epoch & 1:
                    await asyncio.sleep(0.001)
                epoch = table.epoch
                best: Dict[int, float] = {}
                for s in range(0, q.shape[0], _MAX_DIST_FOLD):  # LEAST < bound iff some group's LEAST < bound
                    dist, ids, _ = await asyncio.to_thread(lambda part=q[s : s + _MAX_DIST_FOLD]: table.index.range_search(
                        part, float(max_distance), group_filter=group, filter=filt))
                    for pos, d in zip(ids.tolist(), dist.tolist()):
                        if pos not in best or d < best[pos]:
                            best[pos] = d
                if table.epoch == epoch:
                    break
            out = []
            for pos in sorted(best):
                row = table.rows[pos]
                if row is None:
                    continue
                rec = {c: row.get(c) for c in columns_to_select if c in table.colset and c != "vector"}
                rec[column_name] = best[pos]
                out.append(rec)
            return out
        col = None
        for s in range(0, q.shape[0], _MAX_DIST_FOLD):
            part = await asyncio.to_thread(table.index.distances, q[s : s + _MAX_DIST_FOLD])
            col = part if col is None else np.fmin(col, part)  # fmin: NaN only if both are NaN
        col = table.host_order(col)
        out = []
        for pos, row in enumerate(table.rows):
            if row is None:
                continue
            if workspace_id and row.get("workspaceId") != workspace_id:
                continue
            if knowledge_base_id and row.get("knowledgeBaseId") != knowledge_base_id:
                continue
            rec = {c: row.get(c) for c in columns_to_select if c in table.colset and c != "vector"}
            d = float(col[pos])
            rec[column_name] = None if math.isnan(d) else d
            out.append(rec)
        return out

    async def prepare(self, table_name: Optional[str] = None) -> Dict[str, bool]:
        """After a bulk load of a ``vector`` (fp32) store: build the tables' search-side copies now rather than on the
        first batched query, so that even single queries stream half the bytes (``HvsIndex.prepare``).  Returns
        ``{table: ready}``; 16-bit stores need none and report False."""
        out: Dict[str, bool] = {}
        for name in ([table_name] if table_name else list(self.tables)):
            table = self.get_model(name)
            if table.index is not None and len(table.rows) and hasattr(table.index, "partition"):
                try:
                    await self.partition(name)
                except Exception as exc:  # noqa: BLE001 -- not enough device memory to move the rows: searches keep the per-row test
                    logger.warning("[partition] table '%s' stays in arrival order: %s", name, exc)
            if table.index is not None and hasattr(table.index, "prepare"):
                out[name] = bool(await asyncio.to_thread(table.index.prepare))
                self._note_search_copy(table)
        return out

    # -- compaction and snapshots (the table outlives deletes and restarts) --------------------------
    async def compact(self, table_name: Optional[str] = None) -> Dict[str, int]:
        """Reclaim the rows ``clean_table`` deleted: tombstoned rows leave HBM and the host lists, the
        survivors keep their order (a table in group order is put in that order again, which drops them as well).
        Returns ``{table: live rows}`` for the tables compacted."""
        names = [table_name] if table_name else list(self.tables)
        out: Dict[str, int] = {}
        async with self._lock:
            for name in names:
                table = self.get_model(name)
                n = len(table.rows)
                keep = table.rows.live_positions()
                if len(keep) != n:
                    if table.index is not None and (table.d2h is not None or table.part_rows):
                        await self._partition_table(table)
                    else:
                        table.epoch += 1  # odd: searches launched from here on wait, those under way will notice and run again
                        try:
                            if table.index is not None:
                                left = await asyncio.to_thread(table.index.compact, True)
                                if left != len(keep):
                                    raise RuntimeError(f"table {name}: {left} rows left on the device, {len(keep)} on the host")
                            self._compact_host_rows(table, keep)
                        finally:
                            table.epoch += 1
                out[name] = len(table.rows)
        return out

    @staticmethod
    def _compact_host_rows(table: _Table, keep: np.ndarray) -> None:
        n = len(table.rows)
        table.rows = table.rows.take(keep.tolist())
        for col in ("group", "ws_code", "kb_code", "uri_code", "private", "live"):
            setattr(table, col, getattr(table, col)[:n][keep].copy())
        table.pk_map = {tuple(row.get(c) for c in table.pk): pos for pos, row in enumerate(table.rows)}
        table.key_index.clear()

    async def partition(self, table_name: Optional[str] = None) -> Dict[str, int]:
        """Put the embedded tables (or one) in the order of their (workspaceId, knowledgeBaseId) pairs on the device, so
        that the WHERE on that pair, which every ``query`` carries (``pgvector.py:577-580``), selects a row range:
        no per-row test, no per-query compaction.  Deleted rows are dropped on the way (as ``compact`` does).  Rows
        added afterwards are found by the old per-row test until the next call; the store calls this by itself as a
        table grows (``auto_partition``).  Needs as much free device memory as the table holds.  Returns
        ``{table: rows in group order}``."""
        out: Dict[str, int] = {}
        async with self._lock:
            for name in ([table_name] if table_name else list(self.tables)):
                table = self.get_model(name)
                if table.index is not None and table.has_vector and hasattr(table.index, "partition"):
                    await self._partition_table(table)
                    out[name] = table.part_rows
        return out

    async def _partition_table(self, table: _Table) -> None:
        """(store lock held)  The index's ids are positions in ``table.rows`` and travel with the rows, so searches under
        way stay valid unless deleted rows leave the host lists too, which renumbers them: only then does the epoch move."""
        index = table.index
        n = len(table.rows)
        keep = table.rows.live_positions()
        renumber = len(keep) != n
        if renumber:
            table.epoch += 1
        try:
            await asyncio.to_thread(index.partition, False)
            ids = np.asarray(await asyncio.to_thread(index.ids), dtype=np.int64)  # device position -> position in table.rows
            if len(ids) != len(keep):
                raise RuntimeError(f"table {table.name}: {len(ids)} rows left on the device, {len(keep)} on the host")
            if renumber:
                rank = np.full(n, -1, dtype=np.int64)
                rank[keep] = np.arange(len(keep), dtype=np.int64)
                ids = rank[ids]
                if ids.min(initial=0) < 0:
                    raise RuntimeError(f"table {table.name}: the device kept a row the host has deleted")
                await asyncio.to_thread(index.set_ids, 0, ids)
                self._compact_host_rows(table, keep)
            if np.array_equal(ids, np.arange(len(ids), dtype=np.int64)):
                table.d2h = table.h2d = None
            else:
                table.d2h = ids
                table.h2d = np.empty(len(ids), dtype=np.int64)
                table.h2d[ids] = np.arange(len(ids), dtype=np.int64)
            table.part_rows = int(index.partition_info()[0]) if hasattr(index, "partition_info") else len(ids)
        finally:
            if renumber:
                table.epoch += 1

    @staticmethod
    def _json_default(v):
        if isinstance(v, datetime):
            return {"__datetime__": v.isoformat()}
        if isinstance(v, (np.integer,)):
            return int(v)
        if isinstance(v, (np.floating,)):
            return float(v)
        if isinstance(v, np.ndarray):
            return v.tolist()
        raise TypeError(f"cannot snapshot a value of type {type(v).__name__}")

    @staticmethod
    def _json_hook(d):
        if len(d) == 1 and "__datetime__" in d:
            return datetime.fromisoformat(d["__datetime__"])
        return d

    async def save(self, directory: str) -> None:
        """Write the whole store under ``directory``: ``store.json`` (settings and the filter-code
        dictionaries), per table ``<name>.rows.jsonl`` (metadata rows, ``null`` for a deleted row;
        ``datetime`` values are tagged, other non-JSON values are refused) and, for embedded tables,
        ``<name>.hvs`` (``snapshot.write_index``: the vectors exactly as stored)."""
        import json
        import os

        from .snapshot import write_index

        os.makedirs(directory, exist_ok=True)
        async with self._lock:
            def keys_by_code(mapping):
                keys = [None] * len(mapping)
                for key, code in mapping.items():
                    keys[code] = list(key) if isinstance(key, tuple) else key
                return keys

            meta = {"format": 1, "vector_type": self.vector_type, "embedding_dimension": self.embedding_dimension,
                    "workspaces": keys_by_code(self._ws_codes), "knowledge_bases": keys_by_code(self._kb_codes),
                    "uris": keys_by_code(self._uri_codes), "groups": keys_by_code(self._group_codes),
                    "tables": {name: {"rows": len(t.rows), "indexed": t.index is not None, "partitioned": bool(t.part_rows)}
                               for name, t in self.tables.items()}}
            for name, table in self.tables.items():
                with open(os.path.join(directory, f"{name}.rows.jsonl"), "w") as fh:
                    for row in table.rows:
                        fh.write(json.dumps(None if row is None else dict(row), default=self._json_default) + "\n")
                if table.index is not None:
                    await asyncio.to_thread(write_index, table.index, os.path.join(directory, f"{name}.hvs"))
            with open(os.path.join(directory, "store.json"), "w") as fh:  # last: its presence marks a complete snapshot
                json.dump(meta, fh)

    @classmethod
    async def load(cls, directory: str, embedding_func: Optional[Callable], **kwargs) -> "HipVectorStore":
        """Rebuild a store from ``save``'s files; ``kwargs`` as for the constructor (``vector_type`` and
        ``embedding_dimension`` come from the snapshot)."""
        import json
        import os

        from .snapshot import read_index

        with open(os.path.join(directory, "store.json")) as fh:
            meta = json.load(fh)
        if meta.get("format") != 1:
            raise ValueError(f"{directory}: unknown store snapshot format {meta.get('format')!r}")
        kwargs.pop("vector_type", None)
        kwargs.pop("embedding_dimension", None)
        store = cls(embedding_func, vector_type=meta["vector_type"], embedding_dimension=meta["embedding_dimension"], **kwargs)
        store._ws_codes = {k: i for i, k in enumerate(meta["workspaces"])}
        store._kb_codes = {k: i for i, k in enumerate(meta["knowledge_bases"])}
        store._uri_codes = {k: i for i, k in enumerate(meta["uris"])}
        store._group_codes = {tuple(k): i for i, k in enumerate(meta["groups"])}
        for name, info in meta["tables"].items():
            table = store.get_model(name)
            with open(os.path.join(directory, f"{name}.rows.jsonl")) as fh:
                table.rows = _RowStore(table.columns)
                table.rows.extend(json.loads(line, object_hook=cls._json_hook) for line in fh)
            if len(table.rows) != info["rows"]:
                raise ValueError(f"{directory}: {name}.rows.jsonl holds {len(table.rows)} rows, store.json says {info['rows']}")
            n = len(table.rows)
            table._grow(n)
            for pos, row in enumerate(table.rows):
                if row is None:
                    continue
                ws, kb = row.get("workspaceId"), row.get("knowledgeBaseId")
                if table.has_vector:
                    table.group[pos] = store._group_codes[(ws, kb)]
                table.ws_code[pos] = store._ws_codes[ws]
                table.kb_code[pos] = store._kb_codes[kb]
                table.uri_code[pos] = store._uri_codes[row.get("uri")]
                table.private[pos] = bool(row.get("private", False))
                table.live[pos] = True
                table.pk_map[tuple(row.get(c) for c in table.pk)] = pos
            if info["indexed"]:
                table.index = await asyncio.to_thread(
                    read_index, os.path.join(directory, f"{name}.hvs"), lambda dim, dtype: store._make_index(dim))
                if len(table.index) != n or table.index.dtype != store.dtype:
                    raise ValueError(f"{directory}: {name}.hvs does not match the store ({len(table.index)} rows, {table.index.dtype})")
                if n:
                    # the snapshot holds the rows in the index's order and their ids, which are positions in the metadata rows
                    ids = np.asarray(await asyncio.to_thread(table.index.ids), dtype=np.int64) if hasattr(table.index, "ids") else np.arange(n)
                    if not np.array_equal(ids, np.arange(n, dtype=np.int64)):
                        if not np.array_equal(np.sort(ids), np.arange(n, dtype=np.int64)):
                            raise ValueError(f"{directory}: the ids of {name}.hvs are not the positions of {name}.rows.jsonl")
                        table.d2h = ids
                        table.h2d = np.empty(n, dtype=np.int64)
                        table.h2d[ids] = np.arange(n, dtype=np.int64)
                    # the attribute columns are rebuilt from the metadata rows
                    await asyncio.to_thread(table.index.set_attributes, 0, table.device_order(table.ws_code[:n]),
                                            table.device_order(table.kb_code[:n]), table.device_order(table.uri_code[:n]),
                                            table.device_order(table.private[:n]))
        for name, info in meta["tables"].items():
            if info.get("partitioned") and store.tables[name].index is not None:
                await store.partition(name)  # the row ranges are not part of a snapshot; rows already in order do not move
        return store

    # -- graph side ----------------------------------------------------------------------------------
    @property
    def graph_store(self):
        if self._graph_store is None:
            from .graph_store import HipGraphStore

            self._graph_store = HipGraphStore(device=self.device, graph_factory=self._graph_factory)
        return self._graph_store

    def __getattr__(self, name):
        if name in ("upsert_graph", "query_node", "pagerank_top_chunks_with_reset"):
            delegate = self.__dict__.get("delegate")
            return getattr(delegate if delegate is not None else self.graph_store, name)
        raise AttributeError(name)

    async def health_check(self) -> Dict[str, Any]:
        from . import _lib

        return {
            "vdb": _lib.device_count() > 0,
            "rows": {name: int(t.live[: len(t.rows)].sum()) for name, t in self.tables.items()},
        }

    def close(self) -> None:
        for table in self.tables.values():
            if table.index is not None and hasattr(table.index, "close"):
                table.index.close()
                table.index = None
        if self._graph_store is not None:
            self._graph_store.close()


__all__ = ["BaseVDB", "HipVectorStore", "StoredVector"]
