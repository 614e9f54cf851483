"""The vector-store interface of the reference, as this package implements it.

The reference declares the abstract base ``BaseVDB`` in
``src/hirag_prod/storage/base_vdb.py:5-62`` and injects an implementation through
``HiRAG.create(vdb=...)`` (``src/hirag_prod/hirag.py:457-471,534-536``).  When the reference package
is importable in the running interpreter its own class is used as the base, so that
``isinstance(store, BaseVDB)`` holds inside the reference; otherwise an abstract base with the same
five coroutine methods is declared here.  Table layouts (column names, primary keys) restate
``src/hirag_prod/schema/{chunk,item,triplets,file}.py`` as plain data.
"""

from __future__ import annotations

import abc
from typing import Callable, Dict, List, Tuple

try:  # pragma: no cover - only when the reference is installed next to this package
    from hirag_prod.storage.base_vdb import BaseVDB  # type: ignore
except Exception:  # the reference is absent (or its dependencies are): declare the same contract

    class BaseVDB(abc.ABC):
        """Five coroutines every vector store offers; argument meaning as in the reference."""

        embedding_func: Callable

        @abc.abstractmethod
        async def _init_vdb(self, *args, **kwargs):
            """Create the tables (called once by ``StorageManager.initialize``)."""

        @abc.abstractmethod
        async def upsert_texts(self, texts_to_upsert, properties_list, table_name, with_tokenization=False,
                               with_translation=False, mode="append"):
            """Embed ``texts_to_upsert`` and insert one row per text; existing primary keys win."""

        @abc.abstractmethod
        async def query(self, query, workspace_id, knowledge_base_id, table_name, topk=None, uri_list=None,
                        require_access=None, columns_to_select=None, distance_threshold=None, topn=None,
                        rerank=False):
            """Exact cosine top-k; returns one dict per hit with the selected columns and ``distance``."""

        @abc.abstractmethod
        async def query_by_keys(self, key_value, workspace_id, knowledge_base_id, table_name,
                                key_column="documentKey", columns_to_select=None, limit=None):
            """Rows whose ``key_column`` is in ``key_value``."""

        @abc.abstractmethod
        async def get_existing_document_keys(self, uri, workspace_id, knowledge_base_id, table_name):
            """``documentKey`` of every row of one file in one knowledge base."""


_COMMON_TAIL = ["extractedTimestamp", "createdAt", "createdBy", "updatedAt", "updatedBy"]

# table name -> (columns in declaration order, primary-key columns, has a vector column)
TABLES: Dict[str, Tuple[List[str], List[str], bool]] = {
    "Chunks": (
        ["id", "documentKey", "knowledgeBaseId", "workspaceId", "text", "fileName", "uri", "private", "type",
         "pageNumber", "documentId", "chunkIdx", "chunkType", "pageImageUrl", "pageWidth", "pageHeight", "headers",
         "children", "caption", "bbox", "vector"] + _COMMON_TAIL,
        ["documentKey", "knowledgeBaseId", "workspaceId"],
        True,
    ),
    "Items": (
        ["id", "documentKey", "knowledgeBaseId", "workspaceId", "text", "fileName", "uri", "private", "type",
         "pageNumber", "documentId", "chunkIdx", "chunkType", "pageImageUrl", "pageWidth", "pageHeight", "headers",
         "children", "caption", "bbox", "token_list", "token_start_index_list", "token_end_index_list", "translation",
         "translation_token_list", "translation_token_start_index_list", "translation_token_end_index_list",
         "vector"] + _COMMON_TAIL,
        ["documentKey", "knowledgeBaseId", "workspaceId"],
        True,
    ),
    "Triplets": (
        ["id", "source", "target", "description", "knowledgeBaseId", "workspaceId", "fileName", "uri", "documentId",
         "vector"] + _COMMON_TAIL,
        ["description", "knowledgeBaseId", "workspaceId"],
        True,
    ),
    "Files": (
        ["id", "documentKey", "knowledgeBaseId", "workspaceId", "text", "fileName", "uri", "private", "type",
         "pageNumber", "tableOfContents"] + _COMMON_TAIL,
        ["documentKey", "knowledgeBaseId", "workspaceId"],
        False,
    ),
}

# reference defaults: configs/hi_rag_config.py:42-43,56 and configs/envs.py:12-13
DEFAULT_QUERY_TOP_K = 10
DEFAULT_QUERY_TOP_N = 5
DEFAULT_DISTANCE_THRESHOLD = 0.8
