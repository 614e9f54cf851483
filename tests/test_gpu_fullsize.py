"""GPU tier: the BASELINE configurations at their full per-GPU sizes, checked through properties that
do not need a CPU pass over the whole corpus: planted neighbours, agreement of the independent kernels
(batched MFMA search, streaming scan, full distance column + host selection), float64 recomputation of
the returned rows, shard-merge invariance, and sampled argmin checks for k-means."""

import numpy as np
import pytest

from oracle import pgvector_oracle as po
from tests import parity

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def hvs():
    import hirag_prod_amd

    return hirag_prod_amd


def _recheck(idx, q, dist, ids, storage, id_offset=0):
    """float64 distances of the returned rows (gathered back from HBM) against the returned distances."""
    for i in range(q.shape[0]):
        rows = idx.gather(ids[i] - id_offset)
        full = parity.full_distances(rows, q[i], storage)
        np.testing.assert_allclose(dist[i], full, atol=1e-5)
        assert np.all(np.diff(dist[i]) >= 0)


def _same_up_to_ties(d_a, i_a, d_b, i_b, tol=2e-6):
    """Two kernels add in different orders: equal distances within tol, equal ids except where
    neighbouring ranks tie within tol."""
    np.testing.assert_allclose(d_a, d_b, atol=tol)
    for r in range(d_a.shape[0]):
        assert set(i_a[r].tolist()) == set(i_b[r].tolist()) or abs(d_a[r][-1] - d_b[r][-1]) <= tol
        for p in np.nonzero(i_a[r] != i_b[r])[0]:
            where = np.nonzero(i_b[r] == i_a[r][p])[0]
            assert where.size == 0 or abs(d_b[r][where[0]] - d_a[r][p]) <= tol, (r, p)


def test_10m_x_1536_f32_top10(hvs):
    import torch

    n, d, k = 10_000_000, 1536, 10
    idx = hvs.HvsIndex(d, "f32")
    idx.fill_synthetic(n, seed=1234)
    q = po.synth_rows(4321, 0, 128, d)
    # planted neighbours: ten rows spread over the corpus = q[j] + 0.05 * noise, for four queries
    noise = po.synth_rows(99, 0, 40, d)
    planted_pos = {}
    for j in range(4):
        pos = np.array([(j * 10 + t) * 249_989 + 17 for t in range(10)], dtype=np.int64)
        idx.update(pos, q[j][None, :] + 0.05 * noise[j * 10 : j * 10 + 10])
        planted_pos[j] = pos
    dist, ids, cnt = idx.search(q, k=k, algo="mfma")
    assert np.all(cnt == k)
    for j in range(4):
        assert sorted(ids[j].tolist()) == sorted(planted_pos[j].tolist())
    _recheck(idx, q[:8], dist[:8], ids[:8], "f32")
    # "mfma" used the hi slabs of the split copy as a filter and scored 40 candidates per query exactly; the three
    # products over both slabs and the fp32 MFMA kernel give the same rows
    d1, i1, c1 = idx.search(q, k=k, algo="mfma_f32")
    _same_up_to_ties(d1, i1, dist, ids)
    d1, i1, c1 = idx.search(q, k=k, algo="mfma_split")
    _same_up_to_ties(d1, i1, dist, ids)
    # clouds of 60 near-copies around two more queries: the filter's bound proves nothing there, those two are searched
    # again as a batch of their own, and the answers are cloud rows all the same
    cloud = {}
    for j in (20, 21):
        pos = np.array([((j - 20) * 60 + t) * 77_773 + 5 for t in range(60)], dtype=np.int64)
        idx.update(pos, q[j][None, :] + 1e-4 * po.synth_rows(7 + j, 0, 60, d))
        cloud[j] = set(pos.tolist())
    dc, ic, cc = idx.search(q, k=k, algo="mfma")
    for j in (20, 21):
        assert cc[j] == k and set(ic[j].tolist()) <= cloud[j] and dc[j].max() < 1e-6
    for j in (0, 1, 2, 3, 50, 127):
        assert ic[j].tolist() == ids[j].tolist()
    # left to itself the library takes the filter route for a single query on this table too
    d1, i1, c1 = idx.search(q[9:10], k=k)
    _same_up_to_ties(d1, i1, dist[9:10], ids[9:10])
    # the streaming scan is an independent kernel: same rows, same order
    d2, i2, c2 = idx.search(q[:4], k=k, algo="scan")
    _same_up_to_ties(d2, i2, dist[:4], ids[:4])
    # third route: the full distance column, selected on the host
    col = idx.distances(q[5:6])
    best = np.argpartition(col, k)[:k]
    best = best[np.argsort(col[best], kind="stable")]
    parity.check_topk(dist[5], ids[5], cnt[5], col.astype(np.float64), k, label="column vs search")
    assert set(best.tolist()) == set(ids[5].tolist())
    # shard-merge invariance: three row ranges merged on the GPU give the single-range answer
    cuts = [0, 3_333_333, 7_000_001, n]
    parts = [idx.search(torch.from_numpy(q[:16]).cuda(), k=k, row_begin=cuts[s], row_end=cuts[s + 1], algo="mfma") for s in range(3)]
    md, mi, mc = hvs.merge_topk(torch.stack([p[0] for p in parts], 1), torch.stack([p[1] for p in parts], 1),
                                torch.stack([p[2] for p in parts], 1), k)
    whole = idx.search(torch.from_numpy(q[:16]).cuda(), k=k, algo="mfma")  # 16 embeddings: the same kernel as the parts
    _same_up_to_ties(md.cpu().numpy(), mi.cpu().numpy(), whole[0].cpu().numpy(), whole[1].cpu().numpy(), tol=0.0)  # same bits
    _same_up_to_ties(md.cpu().numpy(), mi.cpu().numpy(), dist[:16], ids[:16])  # and the 128-embedding batch's rows
    # k-means, K = 4096 (centroids = first rows): counts add up, sampled rows sit at their float64 argmin
    cent = torch.from_numpy(idx.gather(np.arange(4096))).cuda()
    assign, sums, counts = idx.kmeans_assign(cent)
    assert int(counts.sum().item()) == n and int(assign.min().item()) >= 0 and int(assign.max().item()) < 4096
    sample = np.random.default_rng(3).choice(n, 300, replace=False)
    rows = idx.gather(sample).astype(np.float64)
    c64 = cent.cpu().numpy().astype(np.float64)
    d2m = (c64 * c64).sum(1)[None, :] - 2.0 * rows @ c64.T
    got = assign.cpu().numpy()[sample]
    order = np.sort(d2m, axis=1)
    clear = (order[:, 1] - order[:, 0]) > 1e-5 * (rows * rows).sum(1)
    assert clear.sum() > 250
    np.testing.assert_array_equal(got[clear], d2m.argmin(1)[clear])
    a0 = got[0]
    members = np.nonzero(assign.cpu().numpy() == a0)[0]
    exp = idx.gather(members[:50_000]).astype(np.float64).sum(0) if len(members) <= 50_000 else None
    if exp is not None:
        np.testing.assert_allclose(sums[a0].cpu().numpy(), exp, rtol=1e-4, atol=1e-2)
    idx.close()


def test_cfg5_shard_bf16_top100_with_workspace_filter(hvs):
    n, d, k = 12_500_000, 768, 100
    first = 25_000_000  # this shard's first global row (rank 2 of 8)
    idx = hvs.HvsIndex(d, "bf16")
    idx.fill_synthetic(n, seed=1234, first_global_row=first, n_groups=64)
    q = po.synth_rows(4321, 0, 256, d)
    for gf in (-1, 5):
        dist, ids, cnt = idx.search(q, k=k, group_filter=gf, algo="mfma")
        assert np.all(cnt == k) and ids.min() >= first and ids.max() < first + n
        if gf >= 0:
            assert np.all(ids % 64 == gf)
        _recheck(idx, q[:3], dist[:3], ids[:3], "bf16", id_offset=first)
        d2, i2, c2 = idx.search(q[:2], k=k, group_filter=gf, algo="scan")
        np.testing.assert_array_equal(c2, cnt[:2])
        _same_up_to_ties(d2, i2, dist[:2], ids[:2])
        col = idx.distances(q[7:8]).astype(np.float64)
        keep = None if gf < 0 else ((first + np.arange(n)) % 64 == gf)
        parity.check_topk(dist[7], ids[7] - first, cnt[7], col, k, keep=keep, label=f"cfg5 column gf={gf}")
    idx.close()
