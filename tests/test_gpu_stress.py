"""-m gpu: seeded random sweep over graph shapes / widths / modes (self loops, duplicate edges, isolated nodes, single-node
graphs, directed and symmetric edge sets) through the C ABI vs the oracle."""
import numpy as np
import pytest
import torch

from oracle import bookkeeping as obk
from oracle import modules as om
from oracle import ops as oops
from tests.util import close

pytestmark = pytest.mark.gpu


def weird_batch(seed):
    """Random batch with every irregularity the kernels must survive."""
    rng = np.random.RandomState(seed)
    G = int(rng.randint(1, 9))
    src, dst, batch, off = [], [], [], 0
    symmetric = bool(rng.rand() < 0.5)
    for g in range(G):
        n = int(rng.choice([1, 1, 2, 3, 5, 9, 17, 40]))
        m = int(rng.randint(0, 4 * n + 1)) if n > 1 else int(rng.randint(0, 2))
        for _ in range(m):
            a, b = int(rng.randint(0, n)), int(rng.randint(0, n))     # self loops and duplicates allowed
            src.append(a + off); dst.append(b + off)
            if symmetric and a != b:
                src.append(b + off); dst.append(a + off)
        batch += [g] * n
        off += n
    ei = torch.tensor([src, dst], dtype=torch.int64).reshape(2, -1)
    perm = torch.from_numpy(rng.permutation(ei.shape[1])).long()
    return ei[:, perm].contiguous(), torch.tensor(batch, dtype=torch.int64), off, G


@pytest.mark.parametrize("seed", range(24))
def test_random_aggregation_and_bookkeeping(dev, seed):
    from dp_gsat_amd.graph_index import BatchIndex
    from dp_gsat_amd.ops import masked_sum_aggregate, pna_aggregate
    import dp_gsat_amd as G
    ei, batch, N, ng = weird_batch(seed)
    E = ei.shape[1]
    rng = np.random.RandomState(1000 + seed)
    H = int(rng.choice([4, 8, 12, 32, 80, 128, 200, 256]))
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(N, H, generator=g)
    x[::3] = x[::3].relu()
    att = torch.rand(E, 1, generator=g)
    ix = BatchIndex(ei.to(dev), N)
    # integer bookkeeping, bit-exact
    rp, perm = obk.csr_by(ei[1], N)
    assert np.array_equal(ix.rowptr_dst.cpu().numpy().astype(np.int64), rp) and np.array_equal(ix.eid_by_dst.cpu().numpy().astype(np.int64), perm)
    und = obk.is_undirected(ei, N)
    assert ix.is_undirected == und
    if und and E:
        assert np.array_equal(ix.rev.cpu().numpy().astype(np.int64), obk.reverse_edge_perm(ei, N))
    seg = ix.graphs(batch.to(dev))
    assert np.array_equal(seg.node_ptr.cpu().numpy().astype(np.int64), obk.graph_ptr(batch))
    # GIN sum and PNA multi-aggregation, forward + backward
    aggr = ["mean", "min", "max", "std", "sum", "var"][: int(rng.randint(1, 7))]
    scal = [["identity"], ["identity", "amplification", "attenuation"], ["linear", "inverse_linear"]][int(rng.randint(0, 3))]
    avg = {"lin": 1.7, "log": 0.9}
    go1 = torch.randn(N, H, generator=g)
    go2 = torch.randn(N, len(scal) * len(aggr) * 2 * H, generator=g)
    ref = {}
    for dt in (torch.float32, torch.float64):
        xo, ao = x.to(dt).clone().requires_grad_(True), att.to(dt).clone().requires_grad_(True)
        o1 = oops.gin_aggregate(xo, ei, ao)
        o2 = oops.pna_aggregate(xo, ei, ao, aggr, scal, avg)
        torch.autograd.backward([o1, o2], [go1.to(dt), go2.to(dt)])
        ref[dt] = (o1, o2, xo.grad, ao.grad)
    xd, ad = x.to(dev).requires_grad_(True), att.to(dev).requires_grad_(True)
    d1 = masked_sum_aggregate(xd, ix, ad)
    d2 = pna_aggregate(xd, ix, ad, None, aggr, scal, avg) if H <= 256 else None
    torch.autograd.backward([d1, d2], [go1.to(dev), go2.to(dev)])
    r32, r64 = ref[torch.float32], ref[torch.float64]
    close(d1, r32[0], ref64=r64[0], what="gin out")
    close(d2, r32[1], ref64=r64[1], what="pna out")
    close(xd.grad, r32[2], 1e-4, ref64=r64[2], what="dx")
    close(ad.grad, r32[3], 1e-4, ref64=r64[3], what="datt")


@pytest.mark.parametrize("seed", range(12))
def test_random_gsat_step(dev, seed):
    import dp_gsat_amd as G
    from dp_gsat_amd.synth import Batch as NS
    ei, batch, N, ng = weird_batch(100 + seed)
    rng = np.random.RandomState(seed)
    H = int(rng.choice([8, 16, 32, 64]))
    backbone = ["GIN", "PNA"][seed % 2]
    edge = bool(rng.rand() < 0.5)
    if ei.shape[1] == 0 and edge:
        edge = False
    g = torch.Generator().manual_seed(seed)
    data = NS(x=torch.randn(N, 6, generator=g), edge_index=ei, batch=batch, edge_attr=None,
              y=torch.randint(0, 2, (ng, 1), generator=g).float(), num_graphs=ng)
    cfg = dict(model_name=backbone, n_layers=2, hidden_size=H, dropout_p=0.0, use_edge_attr=False,
               aggregators=["mean", "min", "max", "std"], scalers=bool(seed % 3 == 0), deg=torch.from_numpy(obk.deg_histogram(ei, N)))
    from tests.test_gpu_models import _mk_pair, _step
    pair = _mk_pair(G, backbone, cfg, 6, 0, H, edge, dev)
    # BatchNorm needs > 1 row in training mode; tiny batches run in eval mode
    training = N > 4
    _step(G, data, *pair, edge, H, dev, training)
