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

    async def apply_clustering(self, workspace_id: str, knowledge_base_id: str, chunk_ids: List[str]) -> Dict[str, Any]:
        empty = {"clusters": {}, "chunk_ids": [], "cluster_info": {}}
        if not chunk_ids:
            return empty
        try:
            rows = await self.vdb.query_by_keys(key_value=chunk_ids, workspace_id=workspace_id, knowledge_base_id=knowledge_base_id,
                                                table_name="Chunks", key_column="documentKey",
                                                columns_to_select=["documentKey", "vector"])
            vectors = {}
            for row in rows:
                vec = row.get("vector")
                if row.get("documentKey") and vec is not None:
                    vectors[row["documentKey"]] = vec.to_list() if hasattr(vec, "to_list") else list(vec)
            valid_ids = [cid for cid in chunk_ids if vectors.get(cid) is not None]
            if not valid_ids:
                logger.warning("No valid embeddings found in chunks")
                return empty
            features = np.array([vectors[cid] for cid in valid_ids])
            if self.n_type == "fixed":
                labels = ward_labels(features, n_clusters=self.n_clusters, linkage_method=self.linkage_method)
            else:
                labels = ward_labels(features, n_clusters=None, distance_threshold=self.distance_threshold,
                                     linkage_method=self.linkage_method)
            clusters: Dict[str, List[str]] = {}
            for cid, label in zip(valid_ids, labels):
                clusters.setdefault(f"cluster_{label}", []).append(cid)
            return {"clusters": clusters, "chunk_ids": valid_ids,
                    "cluster_info": {"n_clusters": len(np.unique(labels)), "total_chunks": len(valid_ids)}}
        except Exception as e:
            logger.error("Failed to apply clustering to chunks: %s", e)
            return empty

    async def __call__(self, workspace_id: str, knowledge_base_id: str, chunks: List[Dict[str, Any]]) -> List[Dict[str, Any]]:
        chunk_ids = [c.get("documentKey") for c in chunks if c.get("documentKey")]
        clusters = (await self.apply_clustering(workspace_id, knowledge_base_id, chunk_ids)).get("clusters", {})
        if not clusters:
            return chunks
        by_id = {c.get("documentKey"): c for c in chunks if c.get("documentKey")}
        kept: List[Dict[str, Any]] = []
        for keys in clusters.values():
            latest: List[Dict[str, Any]] = []
            latest_ts = datetime.min
            for key in keys:
                chunk = by_id.get(key)
                if chunk:
                    ts = chunk.get("extractedTimestamp", datetime.min) or datetime.min
                    if not latest or ts > latest_ts:
                        latest, latest_ts = [chunk], ts
                    elif ts == latest_ts:
                        latest.append(chunk)
            kept.extend(latest)
        return kept
