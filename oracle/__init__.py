"""CPU restatement of the reference's dense-retrieval path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``oracle/`` is imported by the
product package; only ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` may use it, and only as the checker.

Parity status: **parity unpinned** for the cosine scan.  The arithmetic of
the reference's path lives in the PostgreSQL *pgvector* extension (server
side, version not pinned by the reference; Python client ``pgvector==0.4.1``
in ``pyproject.toml:47``), which is absent from ``/root/reference`` and from
this image, and the reference's tests hold no golden vector for it.  The
restatement follows pgvector's published ``cosine_distance`` algorithm and the
reference's call site ``src/hirag_prod/storage/pgvector.py:520-604``; it is
anchored on (a) scikit-learn ``cosine_similarity`` -- the routine the
reference itself calls at ``src/hirag_prod/hirag.py:571-588`` and
``src/hirag_prod/cross_language_search/functions.py:93-100`` -- through the
committed fixtures under ``tests/golden/`` and (b) the importable
``hirag_prod.cluster`` module for the ward-clustering labels.
"""
