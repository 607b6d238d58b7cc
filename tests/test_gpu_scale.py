"""-m gpu: sizes the CPU oracle cannot finish in seconds (power-law, ~1.5 M edges, H=256).  The checker is still the
oracle's plain-PyTorch restatement, evaluated with ATen on the GPU (fp32 and fp64) -- an implementation independent
of libgsat_hip -- plus size-independent properties (involution, checksums, determinism)."""
import pytest
import torch

from oracle import modules as om
from oracle import ops as oops
from tests.util import close

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def big(dev):
    from dp_gsat_amd import synth
    import dp_gsat_amd as G
    data = synth.powerlaw_batch(num_nodes=150_000, num_edges=1_500_000, num_graphs=16, seed=11, x_dim=8).to(dev)
    return data, G.BatchIndex(data.edge_index, data.num_nodes)


def test_bookkeeping_properties_at_scale(big):
    data, ix = big
    ei, N, E = data.edge_index, data.num_nodes, data.num_edges
    assert int(ix.chunk_ptr_dst[-1]) > 0                                     # hubs exist -> chunked rows are exercised
    rp = ix.rowptr_dst.long()
    assert rp[0] == 0 and rp[-1] == E and bool((rp[1:] >= rp[:-1]).all())
    assert torch.equal(rp[1:] - rp[:-1], torch.bincount(ei[1], minlength=N))
    assert torch.equal(torch.sort(ix.eid_by_dst.long())[0], torch.arange(E, device=ei.device))
    assert torch.equal(ei[1][ix.eid_by_dst.long()], torch.repeat_interleave(torch.arange(N, device=ei.device), rp[1:] - rp[:-1]))
    assert ix.is_undirected
    rev = ix.rev.long()
    assert torch.equal(rev[rev], torch.arange(E, device=ei.device))          # involution (duplicates included)
    assert torch.equal(ei[0][rev], ei[1]) and torch.equal(ei[1][rev], ei[0])


@pytest.mark.parametrize("H", [64, 256])
def test_masked_sum_aggregate_at_scale(big, H):
    from dp_gsat_amd.ops import masked_sum_aggregate
    data, ix = big
    dev, N, E = data.edge_index.device, data.num_nodes, data.num_edges
    g = torch.Generator(device=dev).manual_seed(H)
    x = torch.randn(N, H, device=dev, generator=g)
    att = torch.rand(E, 1, device=dev, generator=g)
    go = torch.randn(N, H, device=dev, generator=g)
    ref = {}
    for dt in (torch.float32, torch.float64):
        xo, ao = x.detach().to(dt).clone().requires_grad_(True), att.detach().to(dt).clone().requires_grad_(True)
        oo = oops.gin_aggregate(xo, data.edge_index, ao)
        oo.backward(go.to(dt))
        ref[dt] = (oo.detach(), xo.grad, ao.grad)
    xd, ad = x.detach().clone().requires_grad_(True), att.detach().clone().requires_grad_(True)
    od = masked_sum_aggregate(xd, ix, ad)
    od.backward(go)
    for i, k in enumerate(("out", "dx", "datt")):
        close((od, xd.grad, ad.grad)[i], ref[torch.float32][i], ref64=ref[torch.float64][i], what=k)
    od2 = masked_sum_aggregate(x, ix, att)
    assert torch.equal(od2, od.detach())                                      # bitwise reproducible
    # linearity: A(x1 + x2) == A(x1) + A(x2) up to rounding
    x2 = torch.randn(N, H, device=dev, generator=g)
    lhs = masked_sum_aggregate(x + x2, ix, att)
    rhs = od2 + masked_sum_aggregate(x2, ix, att)
    assert float((lhs - rhs).abs().max()) <= 1e-4 * max(1.0, float(lhs.abs().max()))


def test_extractor_at_scale(big):
    import dp_gsat_amd as G
    data, ix = big
    dev, N, E = data.edge_index.device, data.num_nodes, data.num_edges
    H = 64
    g = torch.Generator(device=dev).manual_seed(3)
    emb = torch.randn(N, H, device=dev, generator=g)
    u = torch.rand(E, 1, device=dev, generator=g).clamp_(1e-10, 1 - 1e-10)
    ga = torch.randn(E, 1, device=dev, generator=g)
    ext = G.ExtractorMLP(H, True).to(dev).train()
    seed = 99
    from dp_gsat_amd._lib import call, ptr, stream
    m1 = torch.empty(E, 4 * H, device=dev); m2 = torch.empty(E, H, device=dev)
    call("gsat_philox_keep_mask", seed, 1, E, 4 * H, 0.5, ptr(m1), stream())
    call("gsat_philox_keep_mask", seed, 2, E, H, 0.5, ptr(m2), stream())
    ref = {}
    for dt in (torch.float32, torch.float64):
        o = om.ExtractorMLP(H, True).to(dev).to(dt).train()
        o.load_state_dict({k: v.to(dt) for k, v in ext.state_dict().items()})
        e = emb.detach().to(dt).clone().requires_grad_(True)
        a = oops.concrete_sample(o(e, data.edge_index, data.batch, masks=[m1.to(dt), m2.to(dt)]), u.to(dt), True)
        a.backward(ga.to(dt))
        ref[dt] = dict(a=a.detach(), demb=e.grad, **{k: p.grad for k, p in o.named_parameters()})
    ed = emb.detach().clone().requires_grad_(True)
    _, a = ext.attend(ed, data.edge_index, data.batch, noise=u, seed=seed)      # in-kernel Philox dropout, sliced segments
    a.backward(ga)
    r32, r64 = ref[torch.float32], ref[torch.float64]
    close(a, r32["a"], ref64=r64["a"], what="att")
    close(ed.grad, r32["demb"], 1e-4, ref64=r64["demb"], what="demb")
    for k, p in ext.named_parameters():
        close(p.grad, r32[k], 1e-4, ref64=r64[k], what=k)
