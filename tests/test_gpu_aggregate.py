"""-m gpu: masked sum aggregation (GIN/GINE) and pools through the C ABI vs the CPU oracle."""
import numpy as np
import pytest
import torch

from oracle import bookkeeping as obk
from oracle import ops as oops
from tests.graphs import random_batch, shuffle_edges

pytestmark = pytest.mark.gpu
TOL = 1e-4   # north_star: within 1e-4 on attention / embedding tensors


def _close(a, b, tol=TOL):
    a, b = a.detach().cpu().double(), b.detach().cpu().double()
    err = (a - b).abs().max().item() if a.numel() else 0.0
    scale = max(1.0, b.abs().max().item() if b.numel() else 1.0)
    assert err <= tol * scale, f"max abs err {err} (scale {scale})"


@pytest.mark.parametrize("H", [4, 16, 64, 80, 128, 256, 512])
@pytest.mark.parametrize("masked", [False, True])
def test_gin_aggregate_fwd_bwd(dev, H, masked):
    from dp_gsat_amd.graph_index import BatchIndex
    from dp_gsat_amd.ops import masked_sum_aggregate
    ei, batch, N = random_batch(1 + H, 9, 1, 40)
    ei = shuffle_edges(ei, 3)
    E = ei.shape[1]
    g = torch.Generator().manual_seed(H)
    x = torch.randn(N, H, generator=g)
    att = torch.rand(E, 1, generator=g) if masked else None
    go = torch.randn(N, H, generator=g)
    # oracle
    xo = x.clone().requires_grad_(True)
    ao = att.clone().requires_grad_(True) if masked else None
    oo = oops.gin_aggregate(xo, ei, ao)
    oo.backward(go)
    # HIP
    ix = BatchIndex(ei.to(dev), N)
    xd = x.to(dev).requires_grad_(True)
    ad = att.to(dev).requires_grad_(True) if masked else None
    od = masked_sum_aggregate(xd, ix, ad)
    od.backward(go.to(dev))
    _close(od, oo)
    _close(xd.grad, xo.grad)
    if masked:
        assert ad.grad.shape == att.shape
        _close(ad.grad, ao.grad)


@pytest.mark.parametrize("H", [16, 128])
def test_gine_aggregate_fwd_bwd(dev, H):
    from dp_gsat_amd.graph_index import BatchIndex
    from dp_gsat_amd.ops import masked_sum_aggregate
    ei, batch, N = random_batch(7, 6, 2, 30, undirected=False)
    E = ei.shape[1]
    g = torch.Generator().manual_seed(5)
    x, ee, att, go = torch.randn(N, H, generator=g), torch.randn(E, H, generator=g), torch.rand(E, 1, generator=g), torch.randn(N, H, generator=g)
    xo, eo, ao = (t.clone().requires_grad_(True) for t in (x, ee, att))
    oo = oops.gine_aggregate(xo, ei, eo, ao)
    oo.backward(go)
    ix = BatchIndex(ei.to(dev), N)
    xd, ed, ad = (t.to(dev).requires_grad_(True) for t in (x, ee, att))
    od = masked_sum_aggregate(xd, ix, ad, ed)
    od.backward(go.to(dev))
    _close(od, oo); _close(xd.grad, xo.grad); _close(ed.grad, eo.grad); _close(ad.grad, ao.grad)


def test_csr_and_rev_bit_exact(dev):
    from dp_gsat_amd.graph_index import BatchIndex
    for seed, und in [(0, True), (1, False), (2, True)]:
        ei, batch, N = random_batch(seed, 12, 1, 50, undirected=und)
        ei = shuffle_edges(ei, seed)
        ix = BatchIndex(ei.to(dev), N)
        rp, perm = obk.csr_by(ei[1], N)
        assert np.array_equal(ix.rowptr_dst.cpu().numpy().astype(np.int64), rp)
        assert np.array_equal(ix.eid_by_dst.cpu().numpy().astype(np.int64), perm)
        assert np.array_equal(ix.src_by_dst.cpu().numpy().astype(np.int64), ei[0].numpy()[perm])
        rp, perm = obk.csr_by(ei[0], N)
        assert np.array_equal(ix.rowptr_src.cpu().numpy().astype(np.int64), rp)
        assert np.array_equal(ix.eid_by_src.cpu().numpy().astype(np.int64), perm)
        assert ix.is_undirected == obk.is_undirected(ei, N) == und
        if und:
            assert np.array_equal(ix.rev.cpu().numpy().astype(np.int64), obk.reverse_edge_perm(ei, N))
        else:
            assert ix.rev is None
        seg = ix.graphs(batch.to(dev))
        assert np.array_equal(seg.node_ptr.cpu().numpy().astype(np.int64), obk.graph_ptr(batch))
        eptr, order, eg, _ = seg.edge_segments
        rp, perm = obk.csr_by(batch[ei[0]], seg.G)
        assert np.array_equal(eptr.cpu().numpy().astype(np.int64), rp)
        assert np.array_equal(order.cpu().numpy().astype(np.int64), perm)


@pytest.mark.parametrize("mean", [False, True])
def test_segment_pool(dev, mean):
    from dp_gsat_amd.graph_index import BatchIndex
    from dp_gsat_amd.ops import segment_pool
    ei, batch, N = random_batch(11, 7, 1, 33)
    H = 64
    x = torch.randn(N, H)
    go = torch.randn(7, H)
    xo = x.clone().requires_grad_(True)
    oo = (oops.global_mean_pool if mean else oops.global_add_pool)(xo, batch, 7)
    oo.backward(go)
    ix = BatchIndex(ei.to(dev), N)
    xd = x.to(dev).requires_grad_(True)
    od = segment_pool(xd, ix.graphs(batch.to(dev)), mean)
    od.backward(go.to(dev))
    _close(od, oo); _close(xd.grad, xo.grad)
