"""The clustering filter that may follow the PageRank step: ward agglomeration of the (<= 10) ranked
chunks' stored vectors, then per cluster only the chunks with the latest ``extractedTimestamp`` stay.

Mirrors ``QueryService.apply_clustering`` and ``filter_chunks_by_cluster``
(``src/hirag_prod/storage/query_service.py:28-153``) over ``HierarchicalClustering``
(``src/hirag_prod/cluster/hierachical_cluster.py:42-46, 72``), i.e. scikit-learn's
``AgglomerativeClustering`` with the reference's configuration (``configs/hi_rag_config.py:49-54``:
3 clusters, ward linkage, "fixed").  At most ten vectors are clustered, so this step stays on the
host ("replicas only": there is nothing to shard); what comes from the GPU is the stored vectors
(``query_by_keys`` with the ``vector`` column -> ``hvs_gather``).  ``tests/golden/ward_*.npz`` hold labels
produced by the reference's own class for the same library calls.

One deliberate difference: when clustering is impossible (fewer chunks than clusters, no vectors)
the reference returns the tuple ``(chunks, [])`` from a function whose callers expect a list
(``query_service.py:119-120``); here the chunk list comes back unchanged.
"""

from __future__ import annotations

import logging
from datetime import datetime
from typing import Any, Dict, List, Optional

import numpy as np

logger = logging.getLogger(__name__)

# configs/hi_rag_config.py:49-54
CLUSTERING_N_CLUSTERS = 3
CLUSTERING_DISTANCE_THRESHOLD = 0.5
CLUSTERING_LINKAGE_METHOD = "ward"
CLUSTERING_N_TYPE = "fixed"


def ward_labels(feature_matrix, n_clusters: Optional[int] = CLUSTERING_N_CLUSTERS, distance_threshold: Optional[float] = None,
                linkage_method: str = CLUSTERING_LINKAGE_METHOD) -> np.ndarray:
    """``HierarchicalClustering(...).fit(X)[0]``: the labels of scikit-learn's agglomerative clustering."""
    if (n_clusters is None) == (distance_threshold is None):
        raise ValueError("Exactly one of n_clusters and distance_threshold must be set.")
    if linkage_method not in ("ward", "complete", "average", "single"):
        raise ValueError("linkage_method must be one of ['ward', 'complete', 'average', 'single']")
    if feature_matrix is None or len(feature_matrix) == 0:
        raise ValueError("feature_matrix cannot be None or empty")
    from sklearn.cluster import AgglomerativeClustering

    try:
        model = AgglomerativeClustering(n_clusters=n_clusters, distance_threshold=distance_threshold, linkage=linkage_method)
        return model.fit_predict(np.asarray(feature_matrix))
    except Exception as e:
        raise ValueError(f"Clustering failed: {e}")


class ClusterFilter:
    """``await ClusterFilter(vdb)(workspace_id, knowledge_base_id, chunks) -> chunks kept``; pass it to
    ``FusionQueryService(cluster_filter=...)``."""

    def __init__(self, vdb, n_type: str = CLUSTERING_N_TYPE, n_clusters: int = CLUSTERING_N_CLUSTERS,
                 distance_threshold: float = CLUSTERING_DISTANCE_THRESHOLD, linkage_method: str = CLUSTERING_LINKAGE_METHOD):
        self.vdb = vdb
        self.n_type = n_type
        self.n_clusters = n_clusters
        self.distance_threshold = distance_threshold
        self.linkage_method = linkage_method

    async def _stored_vectors(self, workspace_id: str, knowledge_base_id: str, chunk_ids: List[str]):
        """(ids that have a stored vector, in the caller's order; their vectors as one matrix) -- the rows come from the GPU
        table through ``query_by_keys`` / ``hvs_gather``"""
        rows = await self.vdb.query_by_keys(key_value=chunk_ids, workspace_id=workspace_id, knowledge_base_id=knowledge_base_id,
                                            table_name="Chunks", key_column="documentKey", columns_to_select=["documentKey", "vector"])
        stored = {r["documentKey"]: r["vector"] for r in rows if r.get("documentKey") and r.get("vector") is not None}
        ids = [cid for cid in chunk_ids if cid in stored]
        if not ids:
            return ids, None
        return ids, np.array([stored[cid].to_list() if hasattr(stored[cid], "to_list") else list(stored[cid]) for cid in ids])

    def _labels(self, features) -> np.ndarray:
        if self.n_type == "fixed":
            return ward_labels(features, n_clusters=self.n_clusters, linkage_method=self.linkage_method)
        return ward_labels(features, n_clusters=None, distance_threshold=self.distance_threshold, linkage_method=self.linkage_method)

    async def apply_clustering(self, workspace_id: str, knowledge_base_id: str, chunk_ids: List[str]) -> Dict[str, Any]:
        """The payload of ``QueryService.apply_clustering`` (query_service.py:28-102): ``clusters`` maps ``cluster_<label>`` to
        its chunk ids (labels in order of first appearance, ids in the caller's order); any failure gives the empty payload."""
        nothing = {"clusters": {}, "chunk_ids": [], "cluster_info": {}}
        if not chunk_ids:
            return nothing
        try:
            ids, features = await self._stored_vectors(workspace_id, knowledge_base_id, chunk_ids)
            if features is None:
                logger.warning("No valid embeddings found in chunks")
                return nothing
            labels = self._labels(features)
        except Exception as e:
            logger.error("Failed to apply clustering to chunks: %s", e)
            return nothing
        members = {f"cluster_{label}": [cid for cid, other in zip(ids, labels) if other == label] for label in dict.fromkeys(labels.tolist())}
        return {"clusters": members, "chunk_ids": ids, "cluster_info": {"n_clusters": len(members), "total_chunks": len(ids)}}

    async def __call__(self, workspace_id: str, knowledge_base_id: str, chunks: List[Dict[str, Any]]) -> List[Dict[str, Any]]:
        """``filter_chunks_by_cluster`` (query_service.py:104-153): of every cluster only the chunks carrying its newest
        ``extractedTimestamp`` stay (all of them on a tie, a missing stamp counts as the oldest)."""
        by_id = {c["documentKey"]: c for c in chunks if c.get("documentKey")}
        keys_in_order = [c["documentKey"] for c in chunks if c.get("documentKey")]
        clusters = (await self.apply_clustering(workspace_id, knowledge_base_id, keys_in_order)).get("clusters", {})
        if not clusters:
            return chunks
        kept: List[Dict[str, Any]] = []
        for keys in clusters.values():
            group = [by_id[key] for key in keys if key in by_id]
            stamps = [c.get("extractedTimestamp") or datetime.min for c in group]
            kept.extend(c for c, stamp in zip(group, stamps) if stamp == max(stamps))
        return kept
