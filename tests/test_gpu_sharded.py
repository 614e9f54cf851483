"""GPU tier: the cross-shard merge kernel, with two shards of one corpus held as two indexes on the
same GPU (what two ranks would hold), merged exactly as ShardedSearch does after its all-gather."""

import numpy as np
import pytest
import torch

from oracle import pgvector_oracle as po
from tests import parity

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("k", [1, 10, 100])
def test_merge_of_two_shards_equals_one_corpus(k):
    import hirag_prod_amd as hvs
    from hirag_prod_amd.sharded import ShardedSearch, shard_range

    n, d, world = 30_001, 128, 3
    rows = po.synth_rows(7, 0, n, d)
    rows[5] = rows[n - 2]  # tie across shards
    q = po.synth_rows(8, 0, 40, d)
    q[0] = rows[5]
    qt = torch.from_numpy(q).cuda()
    shards = []
    for r in range(world):
        lo, hi = shard_range(n, r, world)
        idx = hvs.HvsIndex(d, "f32")
        idx.add(rows[lo:hi], ids=np.arange(lo, hi))
        shards.append(idx)
    parts = [s.search(qt, k=k, max_distance=1.05, algo="mfma" if i % 2 else "scan") for i, s in enumerate(shards)]
    gd = torch.stack([p[0] for p in parts], dim=1).contiguous()  # [nq, P, k]
    gi = torch.stack([p[1] for p in parts], dim=1).contiguous()
    gc = torch.stack([p[2] for p in parts], dim=1).contiguous()
    md, mi, mc = hvs.merge_topk(gd, gi, gc, k)
    torch.cuda.synchronize()
    md, mi, mc = md.cpu().numpy(), mi.cpu().numpy(), mc.cpu().numpy()
    for j in range(q.shape[0]):
        full = parity.full_distances(rows, q[j], "f32")
        parity.check_topk(md[j], mi[j], mc[j], full, k, threshold=1.05, label=f"merged k={k} q={j}")
    if k >= 2:
        assert mi[0][:2].tolist() == [5, n - 2]
    # world = 1: ShardedSearch is a pass-through
    one = ShardedSearch.for_index(shards[0])
    d1, i1, c1 = one.search(qt, k)
    assert torch.equal(i1, shards[0].search(qt, k=k)[1])
    for s in shards:
        s.close()
