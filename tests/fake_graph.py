"""CPU stand-in for ``HvsGraph`` built on the oracle, so the host logic of the graph store and the
fusion step is tested without a GPU (test infrastructure only)."""

import numpy as np

from oracle import pagerank_oracle as pr


class OracleGraph:
    def __init__(self, n_nodes, src, dst, is_chunk, device=0):
        self.n_nodes = int(n_nodes)
        self.src = np.asarray(src, dtype=np.int64)
        self.dst = np.asarray(dst, dtype=np.int64)
        self.is_chunk = np.asarray(is_chunk, dtype=bool)
        self._scores = None

    def pagerank(self, seed_nodes, seed_weights, alpha=0.85, max_iter=100, tol=1e-6):
        p = np.zeros(self.n_nodes)
        for j, w in zip(seed_nodes, seed_weights):
            p[j] += w
        self._scores, iters = pr.pagerank(self.n_nodes, self.src, self.dst, p, alpha=alpha, max_iter=max_iter, tol=tol)
        return iters, 0.0

    def scores(self, first=0, n=None):
        return self._scores[first : None if n is None else first + n].copy()

    def top_chunks(self, k):
        nodes = np.flatnonzero(self.is_chunk)
        order = nodes[np.argsort(-self._scores[nodes], kind="stable")][:k]
        return order.astype(np.int32), self._scores[order]

    def close(self):
        pass


def make_graph(n_nodes, src, dst, is_chunk, device=0):
    return OracleGraph(n_nodes, src, dst, is_chunk, device)
