"""GPU tier: the BaseVDB drop-in end to end on the real HIP index, and a small-knowledge-base
workload of the shape the reference's 2wiki benchmark produces (its data are not in the tree)."""

import numpy as np
import pytest

from oracle import pgvector_oracle as po
from tests import store_cases

pytestmark = pytest.mark.gpu


def _store(vector_type, dim=None):
    import hirag_prod_amd.store as st

    return st.HipVectorStore.create(store_cases.make_embedding_func(dim or store_cases.DIM), vector_type=vector_type)


@pytest.mark.parametrize("vector_type,storage", [("halfvec", "f16"), ("vector", "f32"), ("bf16", "bf16")])
def test_store_basic_gpu(vector_type, storage):
    s = _store(vector_type)
    store_cases.run(store_cases.scenario_basic(s, storage))
    s.close()


def test_store_graph_and_fusion_gpu():
    s = _store("halfvec")
    store_cases.run(store_cases.scenario_graph_and_fusion(s, "f16"))
    s.close()


@pytest.mark.parametrize("vector_type,storage", [("halfvec", "f16"), ("vector", "f32")])
def test_store_load_rows_gpu(vector_type, storage):
    a, b = _store(vector_type), _store(vector_type)
    store_cases.run(store_cases.scenario_load_rows(a, b, storage))
    a.close()
    b.close()


@pytest.mark.parametrize("vector_type,storage", [("halfvec", "f16"), ("vector", "f32")])
def test_store_concurrent_queries_share_launches_gpu(vector_type, storage):
    s = _store(vector_type)
    store_cases.run(store_cases.scenario_concurrent_queries(s, storage))
    s.close()


@pytest.mark.parametrize("vector_type,storage", [("halfvec", "f16"), ("vector", "f32")])
def test_store_filters_reach_the_index_as_codes_gpu(vector_type, storage):
    s = _store(vector_type)
    store_cases.run(store_cases.scenario_no_host_mask(s, storage))
    s.close()


def test_store_failed_insert_and_concurrent_delete_compact_gpu():
    def make_add_fail(index, exc):
        real = index.add

        def add_once(*a, **kw):
            index.add = real
            raise exc

        index.add = add_once

    s = _store("halfvec")
    store_cases.run(store_cases.scenario_failed_add_and_races(s, "f16", make_add_fail))
    s.close()


def test_store_many_embeddings_gpu():
    s = _store("halfvec")
    store_cases.run(store_cases.scenario_many_embeddings(s, "f16"))
    s.close()


@pytest.mark.parametrize("vector_type,storage", [("halfvec", "f16"), ("vector", "f32"), ("bf16", "bf16")])
def test_store_compact_and_snapshot_gpu(tmp_path, vector_type, storage):
    import hirag_prod_amd.store as st

    async def load(directory):
        return await st.HipVectorStore.load(directory, store_cases.make_embedding_func())

    s = _store(vector_type)
    store_cases.run(store_cases.scenario_compact_and_snapshot(s, load, storage, str(tmp_path / "snap")))
    s.close()


def test_small_kb_workload():
    """8192 chunks + 32768 triplets, D = 1536, halfvec, 3 query variants, workspace/kb filter,
    threshold 0.8: both scans of one user query (Chunks, Triplets) against the float64 oracle."""
    dim = 1536
    s = _store("halfvec", dim)

    async def go():
        await s._init_vdb(embedding_dimension=dim)
        rng = np.random.default_rng(1)
        nc, nt = 8192, 32768
        # texts are only keys of the deterministic embedder; half the rows sit in another knowledge base
        chunk_props = [store_cases.chunk_props(i, kb="kb1" if i % 2 else "kb2") for i in range(nc)]
        chunk_texts = [p["text"] for p in chunk_props]
        await s.upsert_texts(chunk_texts, chunk_props, "Chunks")
        trip_props = [
            {"source": f"ent-{i}", "target": f"ent-{i + 1}", "description": f"ent-{i} relates to ent-{i + 1} ({i % 7})",
             "knowledgeBaseId": "kb1" if i % 2 else "kb2", "workspaceId": "ws1", "fileName": "f.pdf", "uri": "u", "documentId": "d"}
            for i in range(nt)
        ]
        trip_texts = [p["description"] for p in trip_props]
        await s.upsert_texts(trip_texts, trip_props, "Triplets")
        for t in range(5):
            # a query close to a stored chunk so that some rows pass the 0.8 threshold
            base = int(rng.integers(0, nc // 2)) * 2 + 1
            qs = [chunk_texts[base], f"paraphrase {t}", f"translation {t}"]
            res = await s.query(qs, "ws1", "kb1", "Chunks", topk=10, topn=5)
            keep = np.array([p["knowledgeBaseId"] == "kb1" for p in chunk_props])
            idx, dist = store_cases.expected_hits(chunk_texts, keep, qs, 10, 0.8, "f16", dim)
            store_cases.same_hits([r["documentKey"] for r in res], [chunk_props[i]["documentKey"] for i in idx], dist,
                                  [r["distance"] for r in res])
            assert len(res) >= 1 and res[0]["documentKey"] == chunk_props[base]["documentKey"]
            tq = [trip_texts[base], f"other {t}"]
            res = await s.query(tq, "ws1", "kb1", "Triplets", topk=10, topn=5,
                                columns_to_select=["source", "target", "description", "fileName"])
            keep = np.array([p["knowledgeBaseId"] == "kb1" for p in trip_props])
            idx, dist = store_cases.expected_hits(trip_texts, keep, tq, 10, 0.8, "f16", dim)
            store_cases.same_hits([r["description"] for r in res], [trip_props[i]["description"] for i in idx], dist)

    store_cases.run(go())
    s.close()


def test_search_copy_knob_and_log_line(caplog):
    """A `vector` (fp32) store builds a second copy of a table for the 16-bit matrix pipe on its first batched search and
    says so in one log line; `search_copy=False` keeps the table at its own size, and the answers are the same."""
    import logging

    import hirag_prod_amd.store as st

    async def go(search_copy):
        s = st.HipVectorStore.create(store_cases.make_embedding_func(), vector_type="vector", search_copy=search_copy)
        await s._init_vdb(embedding_dimension=store_cases.DIM)
        props = [store_cases.chunk_props(i) for i in range(400)]
        await s.upsert_texts([p["text"] for p in props], props, "Chunks")
        index = s.tables["Chunks"].index
        assert index.search_copy() == 0
        ready = await s.prepare("Chunks")
        held = index.search_copy()
        res = await s.query(["text ws1 kb1 11", "text ws1 kb1 12"], "ws1", "kb1", "Chunks", topk=10, distance_threshold=1.5)
        s.close()
        return ready["Chunks"], held, [(r["documentKey"], round(r["distance"], 5)) for r in res]

    with caplog.at_level(logging.INFO, logger="hirag_prod_amd.store"):
        ready, held, with_copy = store_cases.run(go(True))
    assert ready and held >= 400 * store_cases.DIM * 4
    assert sum("[search copy] table 'Chunks'" in r.getMessage() for r in caplog.records) == 1
    caplog.clear()
    with caplog.at_level(logging.INFO, logger="hirag_prod_amd.store"):
        ready, held, without = store_cases.run(go(False))
    assert not ready and held == 0
    assert not any("[search copy]" in r.getMessage() for r in caplog.records)
    assert [k for k, _ in with_copy] == [k for k, _ in without]
    # an index that had a copy gives the memory back when told to
    import hirag_prod_amd as hvs

    idx = hvs.HvsIndex(64, "f32")
    idx.add(np.random.default_rng(0).standard_normal((1000, 64)).astype(np.float32))
    assert idx.prepare() and idx.search_copy() > 0
    assert idx.search_copy(False) == 0 and not idx.prepare()
    assert idx.search_copy(True) == 0 and idx.prepare() and idx.search_copy() > 0
    idx.close()


@pytest.mark.parametrize("vector_type,storage", [("halfvec", "f16"), ("vector", "f32"), ("bf16", "bf16")])
def test_store_in_group_order_gpu(tmp_path, vector_type, storage):
    import hirag_prod_amd.store as st

    async def load(directory):
        return await st.HipVectorStore.load(directory, store_cases.make_embedding_func())

    s = _store(vector_type)
    store_cases.run(store_cases.scenario_partition(s, load, storage, str(tmp_path / "snap")))
    s.close()
