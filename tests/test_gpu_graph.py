"""GPU tier: the HIP graph build and personalised PageRank (through the C ABI) against the oracle and
the networkx golden vectors."""

import glob
import os

import numpy as np
import pytest

from oracle import pagerank_oracle as pro

pytestmark = pytest.mark.gpu

GOLDEN = sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "pagerank_*.npz")))


@pytest.fixture(scope="module")
def hvs():
    import hirag_prod_amd

    assert hirag_prod_amd.device_count() >= 1, "no GPU visible to libhvs"
    return hirag_prod_amd


def _top_from_scores(scores, is_chunk, k):
    nodes = np.flatnonzero(is_chunk)
    return nodes[np.argsort(-scores[nodes], kind="stable")][:k]


@pytest.mark.parametrize("path", GOLDEN)
def test_networkx_goldens(hvs, path):
    z = np.load(path)
    n_chunks = int(z["n_chunks"])
    name = lambda i: f"chunk-{i}" if i < n_chunks else f"ent-{i - n_chunks}"
    edges = [(name(int(a)), name(int(b))) for a, b in zip(z["src"], z["dst"])]
    index = pro.node_index(edges)
    names = list(index)
    assert names == [str(n) for n in z["nodes"]]
    src = np.array([index[s] for s, _ in edges], dtype=np.int32)
    dst = np.array([index[t] for _, t in edges], dtype=np.int32)
    g = hvs.HvsGraph(len(index), src, dst, [n.startswith("chunk-") for n in names])
    # structure: the symmetric simple adjacency the library's undirected graph has
    row_ptr, col = g.csr()
    e_ptr, e_col = pro.undirected_csr(len(index), src, dst)
    assert row_ptr.tolist() == e_ptr.tolist() and col.tolist() == e_col.tolist()
    seeds = [(index[str(k)], float(w)) for k, w in zip(z["reset_nodes"], z["reset_weights"]) if str(k) in index]
    iters, err = g.pagerank([s for s, _ in seeds], [w for _, w in seeds], alpha=float(z["alpha"]))
    scores = g.scores()
    # float64 throughout; only the order of additions differs from the library's
    np.testing.assert_allclose(scores, z["scores"], rtol=0, atol=1e-13)
    assert abs(scores.sum() - 1.0) < 1e-9 and err < len(index) * 1e-6 and 1 < iters < 100
    top, top_scores = g.top_chunks(int(z["topk"]))
    assert [names[j] for j in top] == [str(n) for n in z["top_nodes"]]
    np.testing.assert_allclose(top_scores, z["top_scores"], rtol=0, atol=1e-13)
    assert top_scores.tolist() == scores[top].tolist()
    g.close()


def test_skewed_graph_with_hubs_isolated_nodes_and_ties(hvs):
    rng = np.random.default_rng(21)
    n_chunks, n_ent, n_iso = 30000, 6000, 50
    n = n_chunks + n_ent + n_iso  # the last nodes have no edge: dangling mass goes to the seeds
    s1 = rng.integers(0, n_chunks, 20000)
    d1 = n_chunks + rng.integers(0, 3, 20000)  # three hubs with thousands of neighbours (block-per-row kernel)
    s2 = rng.integers(0, n_chunks, 90000)
    d2 = n_chunks + (rng.zipf(1.4, 90000) % n_ent)  # wave-per-row and sub-wave rows
    s3 = n_chunks + rng.integers(0, n_ent, 15000)
    d3 = n_chunks + rng.integers(0, n_ent, 15000)  # includes self loops and duplicates
    src = np.concatenate([s1, s2, s3]).astype(np.int32)
    dst = np.concatenate([d1, d2, d3]).astype(np.int32)
    is_chunk = np.arange(n) < n_chunks
    g = hvs.HvsGraph(n, src, dst, is_chunk)
    row_ptr, col = g.csr()
    e_ptr, e_col = pro.undirected_csr(n, src, dst)
    assert np.array_equal(row_ptr, e_ptr) and np.array_equal(col, e_col)
    deg = np.diff(row_ptr)
    assert deg.max() > 2048 and (deg == 0).sum() == g.n_dangling >= n_iso and ((deg > 16) & (deg <= 2048)).any()
    seeds = rng.permutation(n_chunks + n_ent)[:40]
    weights = rng.random(40) + 0.01
    p = np.zeros(n)
    p[seeds] = weights
    for alpha in (0.5, 0.85):
        iters, _ = g.pagerank(seeds, weights, alpha=alpha)
        exp, exp_iters = pro.pagerank(n, src, dst, p, alpha=alpha)
        got = g.scores()
        assert iters == exp_iters
        np.testing.assert_allclose(got, exp, rtol=0, atol=1e-13)
        for k in (1, 10, 1000):
            top, top_scores = g.top_chunks(k)
            want = _top_from_scores(got, is_chunk, k)  # ties (many chunks share a score) go to the lower node
            assert top.tolist() == want.tolist() and top_scores.tolist() == got[want].tolist()
            np.testing.assert_allclose(top_scores, exp[_top_from_scores(exp, is_chunk, k)], rtol=0, atol=1e-13)
    # reproducible to the bit, and a seed named twice counts with the sum of its weights
    g.pagerank(seeds, weights, alpha=0.5)
    a = g.scores()
    g.pagerank(np.concatenate([seeds, seeds[:3]]), np.concatenate([weights, weights[:3]]), alpha=0.5)
    b = g.scores()
    p2 = p.copy()
    p2[seeds[:3]] *= 2
    exp2, _ = pro.pagerank(n, src, dst, p2, alpha=0.5)
    np.testing.assert_allclose(b, exp2, rtol=0, atol=1e-13)
    g.pagerank(seeds, weights, alpha=0.5)
    assert np.array_equal(g.scores(), a)
    # more chunks asked for than exist / a range of the scores
    top, _ = g.top_chunks(n_chunks + 10)
    assert len(top) == n_chunks and len(set(top.tolist())) == n_chunks
    assert np.array_equal(g.scores(100, 50), a[100:150])
    g.close()


def test_errors_and_small_cases(hvs):
    # two nodes, one edge: x converges to the closed form
    g = hvs.HvsGraph(2, [0], [1], [True, False])
    g.pagerank([1], [3.0], alpha=0.5)
    exp, _ = pro.pagerank(2, np.array([0]), np.array([1]), np.array([0.0, 1.0]), alpha=0.5)
    np.testing.assert_allclose(g.scores(), exp, atol=1e-15)
    top, sc = g.top_chunks(5)
    assert top.tolist() == [0] and abs(sc[0] - exp[0]) < 1e-15
    with pytest.raises(hvs.HvsError):
        g.pagerank([2], [1.0])  # node outside the graph
    with pytest.raises(hvs.HvsError):
        g.pagerank([0], [float("nan")])
    with pytest.raises(hvs.HvsError):
        g.pagerank([0], [0.0])
    g.close()
    # no edges at all: every node dangling, the scores are the reset vector
    g = hvs.HvsGraph(3, [], [], [True, True, False])
    g.pagerank([1], [1.0], alpha=0.85)
    np.testing.assert_allclose(g.scores(), [0.0, 1.0, 0.0], atol=1e-12)
    g.close()
    with pytest.raises(hvs.HvsError):
        hvs.HvsGraph(3, [0, 5], [1, 2], [True, True, False])  # edge names node 5
    # a slowly mixing graph (a long path) does not converge in 5 iterations: the library's exception
    n = 2000
    g = hvs.HvsGraph(n, np.arange(n - 1), np.arange(1, n), np.ones(n, bool))
    with pytest.raises(hvs.PowerIterationFailedConvergence):
        g.pagerank([0], [1.0], alpha=0.999, max_iter=5, tol=1e-12)
    with pytest.raises(hvs.HvsError):
        hvs.HvsGraph(2, [0], [1], [True, False]).top_chunks(1)  # no PageRank computed yet
    g.close()
