"""The PageRank oracle against golden vectors produced by networkx (the library the reference calls),
and the reset-weight arithmetic of the fusion step against hand-computed values."""

import glob
import os

import numpy as np
import pytest

from oracle import pagerank_oracle as pr


def load_case(path):
    z = np.load(path)
    n_chunks = int(z["n_chunks"])
    name = lambda i: f"chunk-{i}" if i < n_chunks else f"ent-{i - n_chunks}"
    edges = [(name(int(a)), name(int(b))) for a, b in zip(z["src"], z["dst"])]
    reset = {str(k): float(v) for k, v in zip(z["reset_nodes"], z["reset_weights"])}
    return z, edges, reset


@pytest.mark.parametrize("path", sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "pagerank_*.npz"))))
def test_oracle_matches_networkx_goldens(path):
    z, edges, reset = load_case(path)
    index = pr.node_index(edges)
    assert list(index) == [str(n) for n in z["nodes"]]  # same node order as the library's graph
    top = pr.pagerank_top_chunks_with_reset(edges, reset, int(z["topk"]), float(z["alpha"]))
    assert [n for n, _ in top] == [str(n) for n in z["top_nodes"]]
    np.testing.assert_allclose([s for _, s in top], z["top_scores"], rtol=0, atol=1e-15)
    p = np.zeros(len(index))
    for k, v in reset.items():
        if k in index:
            p[index[k]] = v
    src = np.array([index[s] for s, _ in edges])
    dst = np.array([index[t] for _, t in edges])
    scores, iters = pr.pagerank(len(index), src, dst, p, alpha=float(z["alpha"]))
    np.testing.assert_allclose(scores, z["scores"], rtol=0, atol=1e-15)
    assert abs(scores.sum() - 1.0) < 1e-9 and 1 < iters < 100


def test_degenerate_inputs():
    assert pr.pagerank_top_chunks_with_reset([], {"a": 1.0}, 5) == []
    assert pr.pagerank_top_chunks_with_reset([("chunk-1", "e")], {"a": 1.0}, 5) == []  # no seed on the graph
    assert pr.pagerank_top_chunks_with_reset([("chunk-1", "e")], {"e": float("nan"), "chunk-1": -1.0}, 5) == []
    assert pr.pagerank_top_chunks_with_reset([("chunk-1", "e")], {"e": 1.0}, 0) == []
    out = pr.pagerank_top_chunks_with_reset([("chunk-1", "e"), ("", "x"), ("e", "chunk-2")], {"e": 2.0}, 5, alpha=0.5)
    assert [n for n, _ in out] == ["chunk-1", "chunk-2"] and abs(out[0][1] - out[1][1]) < 1e-15  # tie: graph order


def test_fusion_reset_weights():
    chunks = ["chunk-a", "chunk-b", "chunk-c"]
    triplets = [{"source": "e1", "target": "e2"}, {"source": "e2", "target": "e3"}, {"source": "e1", "target": None}]
    reset, phrase = pr.fusion_reset_weights(chunks, triplets, ["e3", "e1", "e2"], {"e1": 4, "e2": 0}, link_top_k=2)
    # inverse rank 1, 1/2, 1/3 -> min-max -> x 0.6; the last chunk gets 0 and is left out of the reset
    assert reset["chunk-a"] == pytest.approx(0.6) and reset["chunk-b"] == pytest.approx(0.6 * (0.5 - 1 / 3) / (1 - 1 / 3))
    assert "chunk-c" not in reset
    # e1: (1 + 1/3) / 4 chunks / 2 occurrences; e2: (1 + 1/2) / 1 / 2; e3: 1/2 / 1 / 1; two largest kept
    assert phrase == {"e2": pytest.approx(0.75), "e3": pytest.approx(0.5)}
    assert reset["e2"] == pytest.approx(0.75) and "e1" not in reset
    reset, phrase = pr.fusion_reset_weights(chunks, [], [], {})
    assert phrase == {} and set(reset) == {"chunk-a", "chunk-b"}
    reset, _ = pr.fusion_reset_weights(["only"], [], [], {})
    assert reset == {}  # a single chunk: min == max -> weight 0


def test_product_reset_weights_equal_the_oracle_statement():
    """``graph_store.reset_weights`` (array arithmetic) against the oracle's statement of query_service.py:290-379:
    same keys in the same order, same float64 bits."""
    import random

    from hirag_prod_amd.graph_store import reset_weights

    rnd = random.Random(7)
    for case in range(300):
        m = rnd.randrange(0, 14)
        chunks = [f"chunk-{rnd.randrange(0, 10)}" if rnd.random() < 0.9 else None for _ in range(m)]  # duplicates and gaps
        ents = [f"e{j}" for j in range(rnd.randrange(1, 9))]
        triplets = [{"source": rnd.choice(ents + [None, ""]), "target": rnd.choice(ents + [None]), "description": "x"}
                    for _ in range(rnd.randrange(0, 12))]
        if case % 7 == 0 and chunks and triplets:
            triplets[0]["source"] = next((c for c in chunks if c), "chunk-0")  # a node seeded from both sides
        entity_ids = list({e for t in triplets for e in (t.get("source"), t.get("target")) if e})
        rnd.shuffle(entity_ids)
        counts = {e: rnd.choice([0, 0, 1, 2, 5, -1]) for e in entity_ids}
        k = rnd.choice([1, 2, 3, 30, 0])
        pw = rnd.choice([0.6, 0.05, 1.0])
        exp_reset, exp_phrase = pr.fusion_reset_weights([c for c in chunks if c], triplets, entity_ids, counts, link_top_k=k, passage_node_weight=pw)
        got_reset, got_phrase = reset_weights(chunks, triplets, counts, k, pw)
        assert list(got_phrase.items()) == list(exp_phrase.items()), case
        assert list(got_reset.items()) == list(exp_reset.items()), case
