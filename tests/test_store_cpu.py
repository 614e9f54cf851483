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
