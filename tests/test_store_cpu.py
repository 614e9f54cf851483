"""CPU tier: host logic of the BaseVDB drop-in (metadata, filters, defaults, errors), with the
oracle-backed test index standing in for the GPU index."""

import pytest

from tests import store_cases
from tests.fake_graph import make_graph
from tests.fake_index import OracleIndex


def _store(vector_type):
    import hirag_prod_amd.store as st

    return st.HipVectorStore.create(store_cases.make_embedding_func(), vector_type=vector_type,
                                    index_factory=lambda dim, dtype, device: OracleIndex(dim, dtype, device),
                                    graph_factory=make_graph)


@pytest.mark.parametrize("vector_type,storage", [("halfvec", "f16"), ("vector", "f32")])
def test_store_basic(vector_type, storage):
    store_cases.run(store_cases.scenario_basic(_store(vector_type), storage))


def test_store_graph_and_fusion():
    store_cases.run(store_cases.scenario_graph_and_fusion(_store("halfvec"), "f16"))


def test_store_load_rows():
    store_cases.run(store_cases.scenario_load_rows(_store("halfvec"), _store("halfvec"), "f16"))


def test_store_concurrent_queries_share_launches():
    store_cases.run(store_cases.scenario_concurrent_queries(_store("halfvec"), "f16"))


def test_store_filters_reach_the_index_as_codes():
    store_cases.run(store_cases.scenario_no_host_mask(_store("halfvec"), "f16"))


def test_store_failed_insert_and_concurrent_delete_compact():
    def make_add_fail(index, exc):
        index.fail_next_add = exc

    store_cases.run(store_cases.scenario_failed_add_and_races(_store("halfvec"), "f16", make_add_fail))


def test_store_many_embeddings():
    store_cases.run(store_cases.scenario_many_embeddings(_store("halfvec"), "f16"))


@pytest.mark.parametrize("vector_type,storage", [("halfvec", "f16"), ("bf16", "bf16")])
def test_store_compact_and_snapshot(tmp_path, vector_type, storage):
    import hirag_prod_amd.store as st

    async def load(directory):
        return await st.HipVectorStore.load(directory, store_cases.make_embedding_func(),
                                            index_factory=lambda dim, dtype, device: OracleIndex(dim, dtype, device))

    store_cases.run(store_cases.scenario_compact_and_snapshot(_store(vector_type), load, storage, str(tmp_path / "snap")))


@pytest.mark.parametrize("vector_type,storage", [("halfvec", "f16"), ("vector", "f32")])
def test_store_in_group_order(tmp_path, vector_type, storage):
    import hirag_prod_amd.store as st

    async def load(directory):
        return await st.HipVectorStore.load(directory, store_cases.make_embedding_func(),
                                            index_factory=lambda dim, dtype, device: OracleIndex(dim, dtype, device))

    store_cases.run(store_cases.scenario_partition(_store(vector_type), load, storage, str(tmp_path / "snap")))


def test_store_is_a_basevdb_and_refuses_without_gpu():
    import hirag_prod_amd as hvs
    import hirag_prod_amd.store as st

    s = st.HipVectorStore.create(store_cases.make_embedding_func())
    assert isinstance(s, st.BaseVDB)
    if hvs.device_count() == 0:
        # the product path has no CPU fallback: creating the index must fail loudly
        with pytest.raises(hvs.HvsError):
            store_cases.run(s._init_vdb(embedding_dimension=8))
    with pytest.raises(ValueError):
        st.HipVectorStore(None, vector_type="int8")


def test_row_store_behaves_like_the_list_of_dicts_it_replaces():
    import json

    import hirag_prod_amd.store as st

    rows = st._RowStore(["documentKey", "text", "vector", "chunkIdx"])
    assert rows.names == ("documentKey", "text", "chunkIdx") and len(rows) == 0
    rows.append({"documentKey": "a", "text": "t0", "chunkIdx": 0, "not_a_column": 1})
    rows.append({"documentKey": "b"})  # a column the row came without reads None
    rows.append(None)  # a deleted row of a snapshot
    rows.extend([{"documentKey": "d", "chunkIdx": 3}])
    assert len(rows) == 4 and rows[2] is None
    r = rows[1]
    assert r["documentKey"] == "b" and r.get("text") is None and r.get("nope", 7) == 7 and "text" in r and "vector" not in r
    assert dict(rows[0]) == {"documentKey": "a", "text": "t0", "chunkIdx": 0}
    assert json.loads(json.dumps(dict(rows[3]))) == {"documentKey": "d", "text": None, "chunkIdx": 3}
    assert rows[-1]["documentKey"] == "d"
    with pytest.raises(IndexError):
        rows[4]
    with pytest.raises(KeyError):
        rows[0]["nope"]
    rows[0] = None  # delete: the view a caller still holds reads None from then on, the row itself is gone
    assert rows[0] is None and rows.column("text")[0] is None
    assert [None if x is None else x["documentKey"] for x in rows] == [None, "b", None, "d"]
    kept = rows.take([1, 3])
    assert len(kept) == 2 and [x["documentKey"] for x in kept] == ["b", "d"] and kept[1]["chunkIdx"] == 3
    rows[2] = {"documentKey": "c", "text": "again"}
    assert rows[2]["text"] == "again"


def test_env_names_of_the_reference_set_type_and_dimension(monkeypatch):
    """configs/envs.py:12-13 + schema/vector_config.py:8-11: USE_HALF_VEC picks the column type, EMBEDDING_DIMENSION its
    width; both are read when the store / its first index is made, not at import time."""
    import hirag_prod_amd.store as st

    made = []

    def factory(dim, dtype, device):
        made.append((dim, dtype))
        return OracleIndex(dim, dtype, device)

    monkeypatch.delenv("USE_HALF_VEC", raising=False)
    monkeypatch.delenv("EMBEDDING_DIMENSION", raising=False)
    assert st.HipVectorStore.create(None).vector_type == "halfvec"  # the reference's default
    for word, expect in (("false", "vector"), ("0", "vector"), ("True", "halfvec"), ("on", "halfvec")):
        monkeypatch.setenv("USE_HALF_VEC", word)
        assert st.HipVectorStore.create(None).vector_type == expect
        assert st.HipVectorStore.create(None, vector_type="bf16").vector_type == "bf16"  # an explicit argument wins
    monkeypatch.setenv("USE_HALF_VEC", "maybe")
    with pytest.raises(ValueError):
        st.HipVectorStore.create(None)

    monkeypatch.setenv("USE_HALF_VEC", "false")
    monkeypatch.setenv("EMBEDDING_DIMENSION", "24")
    s = st.HipVectorStore.create(store_cases.make_embedding_func(), index_factory=factory)
    store_cases.run(s._init_vdb())  # no argument: the environment's width
    assert s.embedding_dimension == 24 and made and all(m == (24, "f32") for m in made)
    s2 = st.HipVectorStore.create(store_cases.make_embedding_func(), index_factory=factory)
    store_cases.run(s2._init_vdb(embedding_dimension=16))  # StorageManager.initialize passes it explicitly
    assert s2.embedding_dimension == 16

    # a store whose first index is made by the first insert: the environment's width is the column's, and a vector of
    # another width is refused like PostgreSQL refuses it
    monkeypatch.setenv("EMBEDDING_DIMENSION", "5")
    s3 = st.HipVectorStore.create(store_cases.make_embedding_func(), index_factory=factory)
    with pytest.raises(ValueError, match="expected 5 dimensions"):
        store_cases.run(s3.upsert_texts(["a text"], [{"documentKey": "k", "workspaceId": "w", "knowledgeBaseId": "b"}], "Chunks"))


def test_the_store_reorders_a_table_once_its_tail_is_a_quarter_of_the_ordered_rows():
    """auto_partition: rows added since the last reordering >= partition_min_rows and >= partition_tail_fraction (0.25)
    of the ordered rows"""
    import hirag_prod_amd.store as st

    s = st.HipVectorStore.create(store_cases.make_embedding_func(), vector_type="halfvec", partition_min_rows=20,
                                 index_factory=lambda dim, dtype, device: OracleIndex(dim, dtype, device))
    assert s.partition_tail_fraction == 0.25

    async def go():
        await s._init_vdb(embedding_dimension=store_cases.DIM)
        chunks = s.tables["Chunks"]

        async def add(first, n):
            props = [store_cases.chunk_props(i, ws=f"ws{i % 3}") for i in range(first, first + n)]
            await s.upsert_texts([p["text"] for p in props], props, "Chunks")

        await add(0, 10)  # one workspace would need no order; three do, but 10 rows are below the minimum
        assert chunks.part_rows == 0
        await add(10, 90)
        assert chunks.part_rows == 100  # 100 rows behind none
        await add(100, 20)
        assert chunks.part_rows == 100  # 20 < 25 = a quarter of 100
        await add(120, 5)
        assert chunks.part_rows == 125  # 25 behind 100
        await add(125, 25)
        assert chunks.part_rows == 125  # 25 < 31
        res = await s.query("text ws1 kb1 130", "ws1", "kb1", "Chunks", topk=1, topn=1)
        assert res[0]["chunkIdx"] == 130  # a row of the tail

    store_cases.run(go())
    s.close()
