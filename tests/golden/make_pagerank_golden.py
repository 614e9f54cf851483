"""Golden vectors for the personalised PageRank step, generated with networkx (a third-party
dependency of the reference, pinned there at 3.3; this image holds 3.4.2 with the same
``_pagerank_scipy`` routine).  Each case runs the library calls the reference makes in
``PGVector.pagerank_top_chunks_with_reset`` (``src/hirag_prod/storage/pgvector.py:406-441``): a DiGraph
from the rows, ``to_undirected()``, ``nx.pagerank(alpha, personalization)``, chunk nodes by descending
score.  Run from the repo root: ``python tests/golden/make_pagerank_golden.py``.
"""

import os

import networkx as nx
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))


def name_of(i, n_chunks):
    return f"chunk-{i}" if i < n_chunks else f"ent-{i - n_chunks}"


def run_case(tag, src, dst, n_chunks, reset, alpha, topk):
    edges = [(name_of(int(a), n_chunks), name_of(int(b), n_chunks)) for a, b in zip(src, dst)]
    g = nx.DiGraph()
    for s, t in edges:
        g.add_edge(s, t)
    total = sum(reset.values())
    pr = nx.pagerank(g.to_undirected(), alpha=alpha, personalization={k: v / total for k, v in reset.items()})
    nodes = list(pr)
    ranked = sorted(((n, s) for n, s in pr.items() if n.startswith("chunk-")), key=lambda t: t[1], reverse=True)[:topk]
    np.savez_compressed(
        os.path.join(HERE, f"pagerank_{tag}.npz"),
        src=np.asarray(src, np.int32), dst=np.asarray(dst, np.int32), n_chunks=np.int64(n_chunks),
        reset_nodes=np.array(list(reset), dtype=np.str_), reset_weights=np.array(list(reset.values()), np.float64),
        alpha=np.float64(alpha), topk=np.int64(topk),
        nodes=np.array(nodes, dtype=np.str_), scores=np.array([pr[n] for n in nodes], np.float64),
        top_nodes=np.array([n for n, _ in ranked], dtype=np.str_), top_scores=np.array([s for _, s in ranked], np.float64),
        networkx_version=np.array(nx.__version__),
    )
    print(tag, "nodes", len(nodes), "top", ranked[:3])


def main():
    # a hand-sized graph: duplicate and antiparallel rows, a self loop, a chunk-chunk edge
    n_chunks = 4
    src = [0, 4, 4, 5, 1, 5, 6, 2, 6, 6, 3, 0]
    dst = [4, 0, 5, 1, 5, 6, 2, 6, 6, 2, 6, 1]
    run_case("hand", src, dst, n_chunks, {"chunk-0": 0.6, "ent-1": 1.5, "ent-9": 3.0}, 0.5, 10)

    rng = np.random.default_rng(5)
    # chunk-entity membership + entity-entity relations, the shape the reference's graph has
    n_chunks, n_ent = 200, 120
    s1 = rng.integers(0, n_chunks, 900)
    d1 = n_chunks + rng.integers(0, n_ent, 900)
    s2 = n_chunks + rng.integers(0, n_ent, 400)
    d2 = n_chunks + rng.integers(0, n_ent, 400)
    src, dst = np.concatenate([s1, s2]), np.concatenate([d1, d2])
    reset = {f"chunk-{i}": 0.6 * (1.0 / (r + 1) - 0.1) / 0.9 for r, i in enumerate(rng.permutation(n_chunks)[:9])}
    reset.update({f"ent-{i}": float(w) for i, w in zip(rng.permutation(n_ent)[:30], rng.random(30))})
    run_case("bipartite_a50", src, dst, n_chunks, reset, 0.5, 10)
    run_case("bipartite_a85", src, dst, n_chunks, reset, 0.85, 25)

    # skewed degrees: a few hub entities touch most chunks
    n_chunks, n_ent = 4000, 1500
    hub = n_chunks + rng.integers(0, 8, 6000)
    s1 = rng.integers(0, n_chunks, 6000)
    s2 = rng.integers(0, n_chunks, 12000)
    d2 = n_chunks + (rng.zipf(1.6, 12000) % n_ent)
    s3 = n_chunks + rng.integers(0, n_ent, 5000)
    d3 = n_chunks + rng.integers(0, n_ent, 5000)
    src, dst = np.concatenate([s1, s2, s3]), np.concatenate([hub, d2, d3])
    reset = {f"ent-{i}": float(w) for i, w in zip(rng.permutation(n_ent)[:30], rng.random(30))}
    reset.update({f"chunk-{i}": 0.3 for i in rng.permutation(n_chunks)[:5]})
    run_case("hubs_a50", src, dst, n_chunks, reset, 0.5, 10)


if __name__ == "__main__":
    main()
