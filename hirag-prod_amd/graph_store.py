"""The graph side of the reference's store and the fusion step that follows retrieval.

``HipGraphStore`` holds the ``Graph`` and ``Nodes`` tables the reference keeps in PostgreSQL
(``src/hirag_prod/storage/pgvector.py:190-344``) on the host and, per (workspace, knowledge base), an
``HvsGraph`` in HBM that is rebuilt only when rows changed; the reference rebuilds a networkx graph
from a ``SELECT`` on every query (``pgvector.py:395-412``).  ``pagerank_top_chunks_with_reset`` has the
arguments, filtering, ordering and empty-result cases of ``pgvector.py:381-450``.

``FusionQueryService`` mirrors the part of ``QueryService`` that follows the two recalls
(``src/hirag_prod/storage/query_service.py:246-402`` and ``441-509``): reset weights from the ranked
chunks and triplets, personalised PageRank, optional rerank through an injected (remote) reranker.
"""

from __future__ import annotations

import asyncio
import logging
import math
import time
from datetime import datetime
from types import SimpleNamespace
from typing import Any, Callable, Dict, List, Optional, Sequence, Tuple, Union

import numpy as np

logger = logging.getLogger(__name__)

# src/hirag_prod/configs/hi_rag_config.py:42-47
DEFAULT_QUERY_TOP_K = 10
DEFAULT_QUERY_TOP_N = 5
DEFAULT_LINK_TOP_K = 30
DEFAULT_PASSAGE_NODE_WEIGHT = 0.6
DEFAULT_PAGERANK_DAMPING = 0.5

GRAPH_COLUMNS = ("id", "source", "target", "uri", "workspaceId", "knowledgeBaseId", "documentId", "extractedTimestamp",
                 "createdAt", "createdBy", "updatedAt", "updatedBy")
NODE_COLUMNS = ("id", "node_id", "workspaceId", "knowledgeBaseId", "entityName", "entityType", "chunkIds", "uri", "documentId",
                "extractedTimestamp", "createdAt", "createdBy", "updatedAt", "updatedBy")


def is_chunk(node_id) -> bool:
    return str(node_id).startswith("chunk-")  # pgvector.py:202-203, 435-439


class _KnowledgeBaseGraph:
    """Rows of one (workspace, knowledge base) and their device graph."""

    def __init__(self):
        self.edges: Dict[Tuple[str, str], dict] = {}  # primary key (source, target) -> row, insertion order
        self.nodes: Dict[str, dict] = {}              # node_id -> Nodes row
        self.graph = None                             # HvsGraph, valid while not dirty
        self.index: Dict[str, int] = {}               # node id -> node number of the device graph
        self.names: List[str] = []
        self.dirty = True


class HipGraphStore:
    def __init__(self, device: int = 0, graph_factory: Optional[Callable[..., Any]] = None):
        self.device = device
        self._graph_factory = graph_factory
        self._kbs: Dict[Tuple[str, str], _KnowledgeBaseGraph] = {}
        self._lock = asyncio.Lock()

    def _make_graph(self, n_nodes, src, dst, chunk_flags):
        if self._graph_factory is not None:
            return self._graph_factory(n_nodes, src, dst, chunk_flags, self.device)
        from .graph import HvsGraph  # fails loudly when libhvs.so or the GPU is missing

        return HvsGraph(n_nodes, src, dst, chunk_flags, device=self.device)

    # -- ingest ----------------------------------------------------------------------------------
    async def upsert_graph(self, relations: List[Any], table_name: str = "Graph", mode: str = "append") -> Dict[str, int]:
        """pgvector.py:190-344: one Graph row per relation (first write wins on the primary key),
        one Nodes row per non-chunk endpoint whose ``chunkIds`` accumulate across calls."""
        now = datetime.now()
        n_edges = 0
        touched_nodes = set()
        async with self._lock:
            for rel in relations:
                props = dict(getattr(rel, "properties", None) or {})
                ws, kb = props.get("workspaceId"), props.get("knowledgeBaseId")
                source, target = rel.source, rel.target
                row = {**{k: v for k, v in props.items() if k in GRAPH_COLUMNS}, "id": props.get("id"), "source": source,
                       "target": target, "uri": props.get("uri", ""), "workspaceId": ws, "knowledgeBaseId": kb,
                       "documentId": props.get("documentId", "")}
                row.setdefault("createdAt", now)
                row["updatedAt"] = now
                state = self._kbs.setdefault((ws, kb), _KnowledgeBaseGraph())
                n_edges += 1
                if (source, target) not in state.edges:  # ON CONFLICT DO NOTHING
                    state.edges[(source, target)] = row
                    state.dirty = True
                chunk_id = props.get("chunkId", [])
                for node_id, name_hint in ((source, None if is_chunk(source) else props.get("source")),
                                           (target, None if is_chunk(target) else props.get("target"))):
                    if is_chunk(node_id):
                        continue
                    node = state.nodes.get(node_id)
                    if node is None:
                        node = {**{k: v for k, v in props.items() if k in NODE_COLUMNS}, "id": props.get("id"), "node_id": node_id,
                                "workspaceId": ws, "knowledgeBaseId": kb, "entityName": name_hint or node_id, "chunkIds": [],
                                "documentId": props.get("documentId", ""), "uri": props.get("uri", "")}
                        node.setdefault("createdAt", now)
                        node.setdefault("entityType", "entity")
                        state.nodes[node_id] = node
                    node["updatedAt"] = now
                    touched_nodes.add((ws, kb, node_id))
                    if chunk_id and chunk_id not in node["chunkIds"]:
                        node["chunkIds"].append(chunk_id)
        return {"edges": n_edges, "nodes": len(touched_nodes)}

    async def query_node(self, node_id: str, workspace_id: str, knowledge_base_id: str):
        """pgvector.py:346-379: an entity with ``metadata.chunkIds`` (deduplicated, first occurrence kept)."""
        state = self._kbs.get((workspace_id, knowledge_base_id))
        row = state.nodes.get(node_id) if state else None
        if not row:
            raise ValueError(f"Node not found: {node_id}")
        chunk_ids = list(dict.fromkeys(row.get("chunkIds") or []))
        meta = SimpleNamespace(entityType=row.get("entityType") or "UNKNOWN", description=[], chunkIds=chunk_ids,
                               documentId=row.get("documentId", ""), workspaceId=row.get("workspaceId", ""),
                               knowledgeBaseId=row.get("knowledgeBaseId", ""))
        return SimpleNamespace(id=node_id, page_content=row.get("entityName") or "", metadata=meta)

    async def clean_table(self, table_name: str, where: Dict[str, Any]) -> bool:
        """Delete the Graph / Nodes rows matching every ``where`` item (pgvector.py:452-473)."""
        if table_name not in ("Graph", "Nodes"):
            raise ValueError(f"No table found for table {table_name}")
        columns = GRAPH_COLUMNS if table_name == "Graph" else NODE_COLUMNS
        unknown = [k for k in where if k not in columns]
        if unknown:
            raise AttributeError(f"table {table_name} has no column {unknown[0]!r}")
        gone = False
        async with self._lock:
            for state in self._kbs.values():
                rows = state.edges if table_name == "Graph" else state.nodes
                hit = [key for key, row in rows.items() if all(row.get(k) == v for k, v in where.items())]
                for key in hit:
                    del rows[key]
                if hit:
                    gone = True
                    if table_name == "Graph":
                        state.dirty = True
        return gone

    # -- query -----------------------------------------------------------------------------------
    def _device_graph(self, state: _KnowledgeBaseGraph):
        if state.dirty or state.graph is None:
            index: Dict[str, int] = {}
            src: List[int] = []
            dst: List[int] = []
            for (s, t) in state.edges:  # node order of G.add_edge over the rows: first appearance, source first
                if not (s and t):
                    continue
                a = index.setdefault(s, len(index))
                b = index.setdefault(t, len(index))
                src.append(a)
                dst.append(b)
            if state.graph is not None and hasattr(state.graph, "close"):
                state.graph.close()
            state.graph = None
            state.index = index
            state.names = list(index)
            if index:
                flags = np.fromiter((is_chunk(n) for n in state.names), dtype=bool, count=len(index))
                state.graph = self._make_graph(len(index), np.asarray(src, dtype=np.int32), np.asarray(dst, dtype=np.int32), flags)
            state.dirty = False
        return state.graph

    def _pagerank(self, state: _KnowledgeBaseGraph, reset_weights, topk: int, alpha: float) -> List[Tuple[str, float]]:
        graph = self._device_graph(state)
        if graph is None:
            return []
        # seeds = the reset entries that name a node of this graph and carry a finite weight > 0 (pgvector.py:414-428),
        # normalised to sum 1; the sum runs in the mapping's order, as the reference's does
        known = [(state.index[node], w) for node, w in (reset_weights or {}).items() if node in state.index]
        nodes, weights = [], []
        for slot, w in known:
            try:
                val = float(w)
            except (TypeError, ValueError):
                continue
            if math.isfinite(val) and val > 0:
                nodes.append(slot)
                weights.append(val)
        total = 0.0
        for val in weights:
            total += val
        if total <= 0:
            return []
        graph.pagerank(nodes, [w / total for w in weights], alpha=alpha)
        top, scores = graph.top_chunks(topk)
        return [(state.names[int(j)], float(s)) for j, s in zip(top, scores)]

    async def pagerank_top_chunks_with_reset(self, workspace_id: str, knowledge_base_id: str, reset_weights: Dict[str, float],
                                             topk: int, alpha: float = 0.85) -> List[Tuple[str, float]]:
        if topk <= 0:
            return []
        start = time.perf_counter()
        async with self._lock:
            state = self._kbs.get((workspace_id, knowledge_base_id))
            if state is None or not state.edges:
                return []
            out = await asyncio.to_thread(self._pagerank, state, reset_weights, int(topk), float(alpha))
        logger.info("[pagerank_top_chunks_with_reset] nodes=%d, returned=%d, elapsed=%.3fs", len(state.index), len(out),
                    time.perf_counter() - start)
        return out

    def close(self) -> None:
        for state in self._kbs.values():
            if state.graph is not None and hasattr(state.graph, "close"):
                state.graph.close()
            state.graph = None
            state.dirty = True


async def apply_reranking(query, results: List[dict], topk: int, topn: int, reranker, rerank_with_time: bool = False,
                          key: str = "text") -> List[dict]:
    """src/hirag_prod/reranker/utils.py:6-26 with the reranker passed in (it is a remote service)."""
    if not results:
        return results
    have = len(results)
    head, keep = min(topk, have), min(topn, have)  # the reranker sees the first `head`, the caller gets `keep` of its answer
    if keep > head:
        raise ValueError(f"topn ({keep}) must be <= topk ({head})")
    return (await reranker.rerank(query=query, items=results[:head], key=key, rerank_with_time=rerank_with_time))[:keep]


def reset_weights(chunk_ids: Sequence[Optional[str]], triplets: Sequence[Dict[str, Any]], chunk_counts: Dict[str, int],
                  link_top_k: int, passage_node_weight: float) -> Tuple[Dict[str, float], Dict[str, float]]:
    """The personalisation vector of the fusion step (what src/hirag_prod/storage/query_service.py:290-379 arrives at),
    as arithmetic on two small arrays.  Returns ``(seeds, facts)``: ``seeds`` maps graph nodes to reset weights in the
    order the PageRank call receives them, ``facts`` the entity part alone (empty = no fact signal).

    Passages: the retrieved chunks in rank order carry ``1/rank``, moved to ``[0, passage_node_weight]`` by min-max
    scaling (``lo = 1/m``, ``hi = 1``; a single chunk scales by 1 and so gets 0); a key that occurs twice keeps its later
    rank.  Facts: an entity's mass is the sum of ``1/rank`` over the triplet ends that name it, added in rank order,
    divided by the number of chunks that mention the entity (``chunk_counts``; entities it does not list, and counts
    <= 0, divide by 1) and by the number of ends; the ``max(1, link_top_k)`` heaviest stay, ties in order of first
    appearance.  Only weights > 0 seed the walk; a node named on both sides gets the sum.
    """
    ranked = [c for c in chunk_ids if c]
    seeds: Dict[str, float] = {}
    if ranked:
        inv = 1.0 / np.arange(1, len(ranked) + 1, dtype=np.float64)
        lo, hi = inv[-1], inv[0]
        scaled = (inv - lo) / ((hi - lo) if hi > lo else 1.0) * passage_node_weight
        final = dict(zip(ranked, scaled.tolist()))  # later duplicates overwrite
        seeds = {c: w for c, w in final.items() if w > 0}

    slot_of: Dict[str, int] = {}
    ends: List[int] = []
    end_rank: List[int] = []
    for rank, rel in enumerate(triplets or ()):
        for ent in (rel.get("source"), rel.get("target")):
            if ent:
                ends.append(slot_of.setdefault(ent, len(slot_of)))
                end_rank.append(rank)
    facts: Dict[str, float] = {}
    if slot_of:
        mass = np.zeros(len(slot_of), dtype=np.float64)
        np.add.at(mass, ends, 1.0 / (np.asarray(end_rank, dtype=np.float64) + 1.0))  # unbuffered: added in rank order
        n_ends = np.bincount(ends, minlength=len(slot_of)).astype(np.float64)
        names = list(slot_of)
        listed = np.fromiter((n in chunk_counts for n in names), dtype=bool, count=len(names))
        mentions = np.fromiter((float(chunk_counts.get(n) or 0) for n in names), dtype=np.float64, count=len(names))
        mentions[mentions <= 0] = 1.0
        mass[listed] = (mass[listed] / mentions[listed]) / n_ends[listed]
        keep = np.argsort(-mass, kind="stable")[: max(1, link_top_k)]
        facts = {names[j]: float(mass[j]) for j in keep}
        for ent, w in facts.items():
            if w > 0:
                seeds[ent] = seeds.get(ent, 0.0) + w
    return seeds, facts


class FusionQueryService:
    """``vdb``: the store of the embedded tables (``HipVectorStore``); ``gdb``: the graph store (the
    same object when the vector store serves the graph calls itself)."""

    CHUNK_COLUMNS = ["text", "uri", "fileName", "private", "updatedAt", "documentKey", "extractedTimestamp"]  # storage_manager.py:178-186
    TRIPLET_COLUMNS = ["source", "target", "description", "fileName"]  # storage_manager.py:208

    def __init__(self, vdb, gdb=None, reranker=None, cluster_filter: Optional[Callable] = None):
        self.vdb = vdb
        self.gdb = gdb if gdb is not None else vdb
        self.reranker = reranker
        self.cluster_filter = cluster_filter

    async def recall_chunks(self, query, workspace_id, knowledge_base_id, topk=None, topn=None) -> Dict[str, Any]:
        chunks = await self.vdb.query(query=query, workspace_id=workspace_id, knowledge_base_id=knowledge_base_id, table_name="Chunks",
                                      topk=topk or DEFAULT_QUERY_TOP_K, topn=topn or DEFAULT_QUERY_TOP_N,
                                      columns_to_select=self.CHUNK_COLUMNS)
        return {"chunks": chunks, "chunk_ids": [c.get("documentKey") for c in chunks if c.get("documentKey")]}

    async def recall_triplets(self, query, workspace_id, knowledge_base_id, topk=None, topn=None) -> Dict[str, Any]:
        relations = await self.vdb.query(query=query, workspace_id=workspace_id, knowledge_base_id=knowledge_base_id,
                                         table_name="Triplets", topk=topk or DEFAULT_QUERY_TOP_K, topn=topn or DEFAULT_QUERY_TOP_N,
                                         columns_to_select=self.TRIPLET_COLUMNS)
        entity_ids = set()
        for rel in relations:
            for ent in (rel.get("source"), rel.get("target")):
                if ent:
                    entity_ids.add(ent)
        return {"relations": relations, "entity_ids": list(entity_ids)}

    async def recall_both(self, query, workspace_id, knowledge_base_id, topk=None, topn=None):
        """The two scans of one user query (Chunks and Triplets) issued together: each table's index
        works on its own HIP stream, so the launches overlap on the device.  The reference awaits them
        one after the other (query_service.py:418-424, 280-286)."""
        return await asyncio.gather(self.recall_chunks(query, workspace_id, knowledge_base_id, topk, topn),
                                    self.recall_triplets(query, workspace_id, knowledge_base_id, topk, topn))

    async def _entity_chunk_count(self, ent: str, workspace_id: str, knowledge_base_id: str) -> int:
        # query_service.py:321-338 reads ``node.metadata.chunk_ids``; the entity the reference's store
        # returns names that field ``chunkIds`` (schema/entity.py:6-11), so the count is 0 there and the
        # frequency penalty stays 1.  Both spellings are read the same way here, with the same outcome
        # for objects shaped like the reference's.
        try:
            node = await self.gdb.query_node(ent, workspace_id, knowledge_base_id)
        except Exception as e:
            logger.error("Failed to fetch entity chunk count: %s", e)
            return 0
        listed = getattr(getattr(node, "metadata", None), "chunk_ids", None)
        return len(listed) if isinstance(listed, list) else 0

    async def pagerank_chunks(self, query: Union[str, List[str]], query_chunks: List[Dict[str, Any]], workspace_id: str,
                              knowledge_base_id: str, topk: Optional[int] = None, topn: Optional[int] = None,
                              link_top_k: Optional[int] = None, passage_node_weight: Optional[float] = None,
                              damping: Optional[float] = None, triplet_recall: Optional[Dict[str, Any]] = None) -> Dict[str, Any]:
        """query_service.py:246-402.  ``triplet_recall`` may carry the result of ``recall_both``."""
        topk = topk or DEFAULT_QUERY_TOP_K
        topn = topn or DEFAULT_QUERY_TOP_N
        link_top_k = link_top_k or DEFAULT_LINK_TOP_K
        passage_node_weight = passage_node_weight or DEFAULT_PASSAGE_NODE_WEIGHT
        damping = damping or DEFAULT_PAGERANK_DAMPING
        if triplet_recall is None:
            triplet_recall = await self.recall_triplets(query, workspace_id, knowledge_base_id, topk, topn)
        entities = triplet_recall["entity_ids"]
        counts = await asyncio.gather(*[self._entity_chunk_count(e, workspace_id, knowledge_base_id) for e in entities]) if triplet_recall["relations"] else []
        seeds, facts = reset_weights([c.get("documentKey") for c in query_chunks], triplet_recall["relations"], dict(zip(entities, counts)),
                                     link_top_k, passage_node_weight)
        if not facts:  # no fact signal: the retrieval order stands (query_service.py:367-368)
            return {"pagerank": [], "query_top": query_chunks}

        ranked = await self.gdb.pagerank_top_chunks_with_reset(reset_weights=seeds, topk=topk, alpha=damping,
                                                               workspace_id=workspace_id, knowledge_base_id=knowledge_base_id)
        ids = [cid for cid, _ in ranked]
        rows = await self.vdb.query_by_keys(key_value=ids, workspace_id=workspace_id, knowledge_base_id=knowledge_base_id,
                                            table_name="Chunks", key_column="documentKey", columns_to_select=None) if ids else []
        by_id = {row.get("documentKey"): row for row in rows}
        rows = [by_id[cid] for cid in ids if cid in by_id]
        score_of = dict(ranked)
        for row in rows:
            row["pagerank_score"] = score_of.get(row.get("documentKey"), 0.0)
        return {"pagerank": rows, "query_top": query_chunks}

    async def query(self, workspace_id: str, knowledge_base_id: str, query: Union[str, List[str]], filter_by_clustering: bool = False,
                    strategy: str = "hybrid", topk: Optional[int] = None, topn: Optional[int] = None) -> Dict[str, Any]:
        """query_service.py:404-509 with both recalls issued together."""
        topk = topk or DEFAULT_QUERY_TOP_K
        topn = topn or DEFAULT_QUERY_TOP_N
        if strategy == "raw":
            return await self.recall_chunks(query, workspace_id, knowledge_base_id, topk, topn)
        if strategy in ("pagerank", "hybrid"):
            chunk_recall, triplet_recall = await self.recall_both(query, workspace_id, knowledge_base_id, topk, topn)
        else:
            chunk_recall, triplet_recall = await self.recall_chunks(query, workspace_id, knowledge_base_id, topk, topn), None
        chunks = chunk_recall.get("chunks", [])
        if not chunks:
            return {"chunks": [], "outliers": []}
        result = {"chunks": chunks, "outliers": []}
        if strategy in ("pagerank", "hybrid"):
            pr = await self.pagerank_chunks(query, chunks, workspace_id, knowledge_base_id, topk, topn, triplet_recall=triplet_recall)
            result["chunks"] = pr.get("pagerank") if pr.get("pagerank") else pr.get("query_top", [])
        if filter_by_clustering:
            if self.cluster_filter is None:
                raise NotImplementedError("filter_by_clustering needs a cluster_filter callable (the ward filter stays on the CPU)")
            result["chunks"] = await self.cluster_filter(workspace_id, knowledge_base_id, result["chunks"])
        if strategy in ("reranker", "hybrid") and self.reranker is not None:
            try:
                result["chunks"] = await apply_reranking(query, result["chunks"], topk, topn, self.reranker, rerank_with_time=True)
            except Exception as e:
                logger.error("Failed to rerank chunks in apply_strategy: %s", e)
        kept = {c.get("documentKey") for c in result["chunks"]}
        result["outliers"] = [c for c in chunks if c.get("documentKey") not in kept]
        return result
