"""-m gpu: BASELINE config 5 at its FULL per-GPU share (the 8-GPU configuration's 1.25 M nodes / 12.5 M directed edges, 128 graphs,
H = 256, GIN + edge attention), i.e. exactly `bench.make_batch("c5", 128, seed)`.

The CPU oracle cannot finish at this size, so:
  * integer bookkeeping: size-independent exact properties (rowptr == bincount, permutations, involution, endpoint swap);
  * aggregation forward / backward at H = 256: the oracle's plain-PyTorch restatement evaluated with ATen on the GPU, fp32 and
    fp64 (an implementation independent of libgsat_hip), and bitwise run-to-run determinism;
  * extractor (H = 256, edge mode): graphs of a batch are independent (per-graph InstanceNorm), so the full batch's outputs on
    the rows of one graph must equal the oracle run on THAT GRAPH ALONE -- checked for three whole graphs taken out of the
    batch, forward and the embedding gradient (avoids a ~100 GB fp64 oracle evaluation).
Peak device memory ~170 GB of the 288 GB."""
import numpy as np
import pytest
import torch

from oracle import modules as om
from oracle import ops as oops
from tests.util import close

pytestmark = pytest.mark.gpu
H = 256


@pytest.fixture(scope="module")
def c5(dev):
    import bench
    import dp_gsat_amd as G
    host, _, _ = bench.make_batch("c5", bench.WORKLOADS["c5"]["graphs"], 0)
    assert host.num_nodes == 1_250_048 and host.num_graphs == 128 and host.num_edges == 12_499_968
    data = host.to(dev)
    ix = G.BatchIndex(data.edge_index, data.num_nodes)
    yield host, data, ix
    G.clear_cache()
    torch.cuda.empty_cache()


def test_c5_bookkeeping_properties(c5):
    host, data, ix = c5
    ei, N, E = data.edge_index, data.num_nodes, data.num_edges
    for rp, rows, eids, other, other_rows in ((ix.rowptr_dst, ei[1], ix.eid_by_dst, ix.src_by_dst, ei[0]),
                                             (ix.rowptr_src, ei[0], ix.eid_by_src, ix.dst_by_src, ei[1])):
        rp = rp.long()
        assert int(rp[0]) == 0 and int(rp[-1]) == E
        assert torch.equal(rp[1:] - rp[:-1], torch.bincount(rows, minlength=N))                  # rowptr == bincount
        e = eids.long()
        assert torch.equal(torch.sort(e)[0], torch.arange(E, device=ei.device))                   # a permutation of the edge ids
        assert torch.equal(rows[e], torch.repeat_interleave(torch.arange(N, device=ei.device), rp[1:] - rp[:-1]))
        assert torch.equal(other.long(), other_rows[e])
        seg = torch.repeat_interleave(torch.arange(N, device=ei.device), rp[1:] - rp[:-1])
        same_row = seg[1:] == seg[:-1]
        assert bool((e[1:][same_row] > e[:-1][same_row]).all())                                   # stable: edge ids ascend inside a row
    assert torch.equal(ix.eid_by_dst[ix.slot_dst_of_srcslot.long()], ix.eid_by_src)
    assert int(ix.chunk_ptr_dst[-1]) > 0 and int(ix.chunk_ptr_src[-1]) > 0                       # hub rows exist: chunked path
    assert ix.is_undirected
    rev = ix.rev.long()
    assert torch.equal(rev[rev], torch.arange(E, device=ei.device))                               # involution
    assert torch.equal(ei[0][rev], ei[1]) and torch.equal(ei[1][rev], ei[0])                      # endpoints swap
    seg = ix.graphs(data.batch, 128)
    npg = torch.bincount(data.batch, minlength=128)
    assert torch.equal(seg.node_ptr.long()[1:] - seg.node_ptr.long()[:-1], npg)
    eptr, order, eg, eg32 = seg.edge_segments
    assert torch.equal(eg, data.batch[ei[0]]) and torch.equal(eptr.long()[1:] - eptr.long()[:-1], torch.bincount(eg, minlength=128))
    assert torch.equal(eg[order.long()], torch.sort(eg, stable=True)[0])


def test_c5_aggregation_fwd_bwd_h256(c5):
    from dp_gsat_amd.ops import masked_sum_aggregate
    host, data, ix = c5
    dev, N, E = data.edge_index.device, data.num_nodes, data.num_edges
    g = torch.Generator(device=dev).manual_seed(5)
    x = torch.randn(N, H, device=dev, generator=g)
    att = torch.rand(E, 1, device=dev, generator=g)
    go = torch.randn(N, H, device=dev, generator=g)
    xd, ad = x.clone().requires_grad_(True), att.clone().requires_grad_(True)
    od = masked_sum_aggregate(xd, ix, ad)
    od.backward(go)
    out, dx, da = od.detach(), xd.grad, ad.grad
    xd2, ad2 = x.clone().requires_grad_(True), att.clone().requires_grad_(True)
    od2 = masked_sum_aggregate(xd2, ix, ad2)
    od2.backward(go)
    assert torch.equal(od2.detach(), out) and torch.equal(xd2.grad, dx) and torch.equal(ad2.grad, da)     # bitwise reproducible
    del od, od2, xd2, ad2
    got = (out, dx, da)
    for dt, tol in ((torch.float64, 1e-4), (torch.float32, 1e-4)):          # fp64 ATen is the yardstick; fp32 ATen (atomics) is looser
        xo, ao = x.to(dt).requires_grad_(True), att.to(dt).requires_grad_(True)
        oo = oops.gin_aggregate(xo, data.edge_index, ao)
        oo.backward(go.to(dt))
        for v, r, k in zip(got, (oo.detach(), xo.grad, ao.grad), ("out", "dx", "datt")):
            scale = max(1.0, float(r.abs().max()))
            err = float((v.to(dt) - r).abs().max())
            assert err <= tol * scale, f"{k} vs ATen {dt}: {err:.3e} > {tol * scale:.3e}"
        del xo, ao, oo
        torch.cuda.empty_cache()


def test_c5_extractor_h256_whole_graphs_out_of_the_batch(c5):
    import dp_gsat_amd as G
    host, data, ix = c5
    dev, N, E = data.edge_index.device, data.num_nodes, data.num_edges
    g = torch.Generator(device=dev).manual_seed(7)
    emb = torch.randn(N, H, device=dev, generator=g)
    u = torch.rand(E, 1, device=dev, generator=g).clamp_(1e-10, 1 - 1e-10)
    ga = torch.randn(E, 1, device=dev, generator=g)
    ext = G.ExtractorMLP(H, True).to(dev).eval()             # eval: no dropout; InstanceNorm uses batch statistics in eval too
    ed = emb.clone().requires_grad_(True)
    z, _ = ext.attend(ed, data.edge_index, data.batch)
    att = G.concrete_sample(z, 1.0, True, noise=u)
    att.backward(ga)
    demb = ed.grad
    # training mode (in-kernel Philox dropout): bitwise reproducible for a fixed seed, different for another seed
    ext.train()
    with torch.no_grad():
        a1 = ext.attend(emb, data.edge_index, data.batch, noise=u, seed=11)[1]
        a2 = ext.attend(emb, data.edge_index, data.batch, noise=u, seed=11)[1]
        a3 = ext.attend(emb, data.edge_index, data.batch, noise=u, seed=12)[1]
    assert torch.equal(a1, a2) and not torch.equal(a1, a3)
    assert bool(torch.isfinite(a1).all()) and float(a1.min()) >= 0.0 and float(a1.max()) <= 1.0
    del a1, a2, a3
    torch.cuda.empty_cache()
    batch_h, ei_h = host.batch, host.edge_index
    node_ptr = np.concatenate([[0], np.cumsum(np.bincount(batch_h.numpy(), minlength=128))])
    eg = batch_h[ei_h[0]]
    for gid in (0, 57, 127):
        n0, n1 = int(node_ptr[gid]), int(node_ptr[gid + 1])
        sel = (eg == gid).nonzero().view(-1)
        ei_g = (ei_h[:, sel] - n0).to(dev)
        assert int(ei_g.min()) >= 0 and int(ei_g.max()) < n1 - n0
        sel_d = sel.to(dev)
        ref = {}
        for dt in (torch.float32, torch.float64):
            o = om.ExtractorMLP(H, True).to(dev).to(dt).eval()
            o.load_state_dict({k: v.to(dt) for k, v in ext.state_dict().items()})
            e = emb[n0:n1].to(dt).clone().requires_grad_(True)
            zo = o(e, ei_g, torch.zeros(n1 - n0, dtype=torch.int64, device=dev))
            ao = oops.concrete_sample(zo, u[sel_d].to(dt), True)
            ao.backward(ga[sel_d].to(dt))
            ref[dt] = (zo.detach(), ao.detach(), e.grad)
        r32, r64 = ref[torch.float32], ref[torch.float64]
        close(z.detach()[sel_d], r32[0], ref64=r64[0], what=f"logits of graph {gid}")
        close(att.detach()[sel_d], r32[1], ref64=r64[1], what=f"att of graph {gid}")
        close(demb[n0:n1], r32[2], 1e-4, ref64=r64[2], what=f"demb of graph {gid}")
