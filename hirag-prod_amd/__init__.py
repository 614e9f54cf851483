"""MI355X-native dense-retrieval path of hirag-prod: exact cosine top-k and k-means assignment.

Import as ``hirag_prod_amd`` (the alias module at the repository root maps that name onto this
directory).  Public surface:

* ``HvsIndex``        -- one embedded table resident in HBM (``index.py``)
* ``HipVectorStore``  -- drop-in for the reference's ``BaseVDB`` / ``PGVector`` (``store.py``)
* ``HvsComm`` / ``ShardedSearch`` / ``sharded_kmeans_step`` -- one process per GPU over RCCL (``sharded.py``)
* ``HvsGraph``        -- knowledge graph in HBM + personalised PageRank (``graph.py``)
* ``HipGraphStore`` / ``FusionQueryService`` -- graph tables and the fusion step after retrieval (``graph_store.py``)
"""

from ._lib import HvsError, device_count, load_library
from .index import HvsIndex, merge_topk

__all__ = ["HvsError", "HvsIndex", "device_count", "load_library", "merge_topk"]


def __getattr__(name):
    # heavier modules are imported on first use
    if name in ("HipVectorStore", "BaseVDB"):
        from . import store

        return getattr(store, name)
    if name in ("HvsGraph", "PowerIterationFailedConvergence"):
        from . import graph

        return getattr(graph, name)
    if name in ("HipGraphStore", "FusionQueryService", "apply_reranking"):
        from . import graph_store

        return getattr(graph_store, name)
    if name in ("ClusterFilter", "ward_labels"):
        from . import cluster_filter

        return getattr(cluster_filter, name)
    if name in ("ShardedSearch", "HvsComm", "host_exchange", "sharded_kmeans_step", "shard_range"):
        from . import sharded

        return getattr(sharded, name)
    raise AttributeError(name)
