"""-m gpu: extractor MLP (+ InstanceNorm, dropout masks, concrete sampler), lift / symmetrise / info loss
through the C ABI vs the CPU oracle."""
import numpy as np
import pytest
import torch

from oracle import bookkeeping as obk
from oracle import modules as om
from oracle import ops as oops
from tests.graphs import random_batch, shuffle_edges
from tests.util import close

pytestmark = pytest.mark.gpu


def _copy_params(dst, src):
    dst.load_state_dict(src.state_dict())


@pytest.mark.parametrize("H", [16, 64, 80, 128])
@pytest.mark.parametrize("edge_mode", [True, False])
@pytest.mark.parametrize("training", [True, False])
def test_extractor_fwd_bwd(dev, H, edge_mode, training):
    import dp_gsat_amd as G
    ei, batch, N = random_batch(H + 1, 10, 2, 40)
    ei = shuffle_edges(ei, 2)                     # rows of a graph are NOT contiguous -> exercises seg_order
    E = ei.shape[1]
    M = E if edge_mode else N
    g = torch.Generator().manual_seed(H)
    emb = torch.randn(N, H, generator=g)
    C1, C2 = (4 * H, H) if edge_mode else (2 * H, H)
    masks = [(torch.rand(M, C1, generator=g) > 0.5).float(), (torch.rand(M, C2, generator=g) > 0.5).float()]
    u = torch.rand(M, 1, generator=g).clamp_(1e-10, 1 - 1e-10)
    gz, ga = torch.randn(M, 1, generator=g), torch.randn(M, 1, generator=g)

    ref = {}
    oext = om.ExtractorMLP(H, edge_mode)
    for dt in (torch.float32, torch.float64):
        ext = om.ExtractorMLP(H, edge_mode).to(dt)
        ext.load_state_dict({k: v.to(dt) for k, v in oext.state_dict().items()})
        ext.train(training)
        e = emb.to(dt).clone().requires_grad_(True)
        z = ext(e, ei, batch, masks=[m.to(dt) for m in masks])
        a = oops.concrete_sample(z, u.to(dt), training)
        torch.autograd.backward([z, a], [gz.to(dt), ga.to(dt)])
        ref[dt] = dict(z=z, a=a, demb=e.grad, **{k: p.grad for k, p in ext.named_parameters()})

    ext = G.ExtractorMLP(H, edge_mode).to(dev)
    ext.load_state_dict(oext.state_dict())
    ext.train(training)
    ed = emb.to(dev).requires_grad_(True)
    z, a = ext.attend(ed, ei.to(dev), batch.to(dev), noise=u.to(dev) if training else None,
                      dropout_masks=[m.to(dev) for m in masks])
    torch.autograd.backward([z, a], [gz.to(dev), ga.to(dev)])
    r32, r64 = ref[torch.float32], ref[torch.float64]
    close(z, r32["z"], ref64=r64["z"], what="logits")
    close(a, r32["a"], ref64=r64["a"], what="att")
    close(ed.grad, r32["demb"], ref64=r64["demb"], what="demb")
    for k, p in ext.named_parameters():
        close(p.grad, r32[k], ref64=r64[k], what=k)


def test_extractor_philox_masks_match_explicit(dev):
    """The in-kernel Philox dropout equals running with the mask tensor the library reports for that seed."""
    import ctypes
    import dp_gsat_amd as G
    from dp_gsat_amd._lib import call, ptr, stream
    H = 32
    ei, batch, N = random_batch(3, 6, 3, 20)
    E = ei.shape[1]
    emb = torch.randn(N, H).to(dev)
    ext = G.ExtractorMLP(H, True).to(dev).train()
    seed = 1234567
    z1, _ = ext.attend(emb, ei.to(dev), batch.to(dev), seed=seed)
    m1 = torch.empty(E, 4 * H, device=dev); m2 = torch.empty(E, H, device=dev)
    call("gsat_philox_keep_mask", seed, 1, E, 4 * H, 0.5, ptr(m1), stream())
    call("gsat_philox_keep_mask", seed, 2, E, H, 0.5, ptr(m2), stream())
    z2, _ = ext.attend(emb, ei.to(dev), batch.to(dev), dropout_masks=[m1, m2])
    assert torch.equal(z1, z2)
    frac = m1.mean().item()
    assert 0.45 < frac < 0.55           # Bernoulli(keep = 0.5)
    z3, _ = ext.attend(emb, ei.to(dev), batch.to(dev), seed=seed)
    assert torch.equal(z1, z3)          # bitwise reproducible


def test_generic_mlp_matches_fused(dev):
    """BatchSequential path (Linear / InstanceNorm kernel / ReLU) == fused extractor pipeline in eval mode."""
    import dp_gsat_amd as G
    H = 32
    ei, batch, N = random_batch(4, 5, 3, 20)
    emb = torch.randn(N, H).to(dev)
    ext = G.ExtractorMLP(H, False).to(dev).eval()
    z_fused = ext(emb, ei.to(dev), batch.to(dev))
    z_generic = ext.mlp(emb, batch.to(dev))
    close(z_generic, z_fused, 1e-5)
    oext = om.ExtractorMLP(H, False).eval()
    oext.load_state_dict(ext.state_dict())
    close(z_generic, oext(emb.cpu(), ei, batch))


def test_instance_norm_bwd(dev):
    import dp_gsat_amd as G
    ei, batch, N = random_batch(8, 7, 1, 30)
    C = 48
    x = torch.randn(N, C)
    go = torch.randn(N, C)
    xo = x.clone().requires_grad_(True)
    yo = oops.instance_norm(xo, batch, 7)
    yo.backward(go)
    xd = x.to(dev).requires_grad_(True)
    yd = G.InstanceNorm(C)(xd, batch.to(dev))
    yd.backward(go.to(dev))
    close(yd, yo); close(xd.grad, xo.grad)


@pytest.mark.parametrize("undirected", [True, False])
def test_lift_symmetrise_info(dev, undirected):
    import dp_gsat_amd as G
    ei, batch, N = random_batch(21, 9, 2, 30, undirected=undirected)
    ei = shuffle_edges(ei, 5)
    E = ei.shape[1]
    g = torch.Generator().manual_seed(1)
    na = torch.rand(N, 1, generator=g); ea = torch.rand(E, 1, generator=g)
    ge = torch.randn(E, 1, generator=g)
    # lift
    no = na.clone().requires_grad_(True)
    lo = oops.lift_node_att_to_edge_att(no, ei); lo.backward(ge)
    nd = na.to(dev).requires_grad_(True)
    ld = G.lift_node_att_to_edge_att(nd, ei.to(dev)); ld.backward(ge.to(dev))
    close(ld, lo); close(nd.grad, no.grad)
    # symmetrise
    eo = ea.clone().requires_grad_(True)
    rev = torch.from_numpy(obk.reverse_edge_perm(ei, N)) if undirected else None
    so = oops.symmetrise(eo, rev); so.backward(ge)
    ed = ea.to(dev).requires_grad_(True)
    sd = G.symmetrise_edge_att(ed, ei.to(dev), N); sd.backward(ge.to(dev))
    close(sd, so); close(ed.grad, eo.grad)
    if undirected:
        assert torch.equal(sd[:, 0].cpu(), sd[:, 0].cpu()[rev])          # symmetric by construction
    # info loss: scalar r and tensor prior
    for r in (0.7, torch.rand(E, 1, generator=g) * 0.8 + 0.1):
        eo = ea.clone().requires_grad_(True)
        io = oops.info_loss(eo, r); io.backward()
        ed = ea.to(dev).requires_grad_(True)
        idv = G.info_loss(ed, r.to(dev) if isinstance(r, torch.Tensor) else r); idv.backward()
        close(idv, io, 1e-5); close(ed.grad, eo.grad, 1e-5)
    # info(att = r) ~ 0 (analytic identity)
    assert abs(G.info_loss(torch.full((E, 1), 0.7, device=dev), 0.7).item()) < 1e-5


def test_samplers(dev):
    import dp_gsat_amd as G
    g = torch.Generator().manual_seed(0)
    z = torch.randn(1000, 1, generator=g); u = torch.rand(1000, 1, generator=g).clamp_(1e-10, 1 - 1e-10)
    ga = torch.randn(1000, 1, generator=g)
    for fn_o, fn_d in ((lambda t: oops.concrete_sample(t, u, True), lambda t: G.concrete_sample(t, 1.0, True, u.to(dev))),
                       (lambda t: oops.concrete_sample(t, None, False), lambda t: G.concrete_sample(t, 1.0, False)),
                       (lambda t: oops.gumbel_sigmoid(t, u, 0.1), lambda t: G.gumbel_sigmoid(t, 0.1, 1e-10, u.to(dev)))):
        zo = z.clone().requires_grad_(True); ao = fn_o(zo); ao.backward(ga)
        zd = z.to(dev).requires_grad_(True); ad = fn_d(zd); ad.backward(ga.to(dev))
        close(ad, ao, 1e-5); close(zd.grad, zo.grad, 1e-4)


def test_reorder_like(dev):
    import dp_gsat_amd as G
    ei, batch, N = random_batch(2, 5, 3, 15)
    E = ei.shape[1]
    vals = torch.randn(E, 1)
    trans = ei.flip(0)
    out = G.reorder_like(trans.to(dev), ei.to(dev), vals.to(dev)).cpu()
    assert torch.equal(out, torch.from_numpy(obk.reorder_like(trans, ei, vals)))
    bad = ei.clone(); bad[1, 0] = (bad[1, 0] + 1) % N
    with pytest.raises(ValueError):
        G.reorder_like(bad.to(dev), ei.to(dev), vals.to(dev))


def test_in_kernel_concrete_noise(dev):
    """noise="philox": the concrete sampler's u is drawn inside the head kernel (no uniform_ launch).  The draw is
    exactly gsat_philox_noise(seed), so passing that tensor explicitly must give bitwise the same attention and gradients; the
    noise is uniform on (0, 1) and changes with the seed."""
    import dp_gsat_amd as G
    from dp_gsat_amd._lib import call, ptr, stream
    ei, batch, N = random_batch(3, 40, 5, 30)
    E = ei.shape[1]
    for edge_mode, M in ((True, E), (False, N)):
        ext = G.ExtractorMLP(32, edge_mode).to(dev).train()
        emb = torch.randn(N, 32, device=dev)
        ga = torch.randn(M, 1, device=dev)
        seed = 12345
        u = torch.empty(M, 1, device=dev)
        call("gsat_philox_noise", seed, M, ptr(u), stream())
        outs = []
        for noise in ("philox", u):
            e = emb.clone().requires_grad_(True)
            z, a = ext.attend(e, ei.to(dev), batch.to(dev), noise=noise, seed=seed)
            a.backward(ga)
            outs.append((z.detach(), a.detach(), e.grad))
        for x, y in zip(*outs):
            assert torch.equal(x, y)
        a2 = ext.attend(emb, ei.to(dev), batch.to(dev), noise="philox", seed=seed + 1)[1]
        assert not torch.equal(a2, outs[0][1])
        uu = torch.empty(200_000, device=dev)
        call("gsat_philox_noise", 7, uu.numel(), ptr(uu), stream())
        assert 0.0 < float(uu.min()) and float(uu.max()) < 1.0
        assert abs(float(uu.mean()) - 0.5) < 5e-3 and abs(float(uu.var()) - 1.0 / 12.0) < 2e-3
        hist = torch.histc(uu, bins=20, min=0.0, max=1.0) / uu.numel()
        assert float((hist - 0.05).abs().max()) < 3e-3


@pytest.mark.parametrize("edge_mode", [True, False])
@pytest.mark.parametrize("H", [32, 128, 320])
def test_extractor_mixed_segment_lengths(dev, H, edge_mode):
    """Batches mixing one-row, short and long (150-210 row) graphs in the unsliced statistics path, also at widths that are not a
    power of two and need several 64-channel column blocks (H = 320: C1 = 640 / 1280)."""
    import dp_gsat_amd as G
    from tests.graphs import random_batch as rb
    parts = [rb(7, 30, 2, 14), rb(8, 1, 150, 150), rb(9, 20, 3, 9), rb(10, 1, 210, 210), rb(11, 1, 1, 1)]
    eis, batches, off, goff = [], [], 0, 0
    for ei_p, b_p, n_p in parts:
        eis.append(ei_p + off)
        batches.append(b_p + goff)
        off += n_p
        goff += int(b_p.max()) + 1
    ei, batch, N = torch.cat(eis, dim=1), torch.cat(batches), off
    ei = shuffle_edges(ei, 4)
    E = ei.shape[1]
    M = E if edge_mode else N
    g = torch.Generator().manual_seed(H)
    emb = torch.randn(N, H, generator=g)
    C1, C2 = (4 * H, H) if edge_mode else (2 * H, H)
    masks = [(torch.rand(M, C1, generator=g) > 0.5).float(), (torch.rand(M, C2, generator=g) > 0.5).float()]
    u = torch.rand(M, 1, generator=g).clamp_(1e-10, 1 - 1e-10)
    ga = torch.randn(M, 1, generator=g)
    oext = om.ExtractorMLP(H, edge_mode)
    ref = {}
    for dt in (torch.float32, torch.float64):
        ext = om.ExtractorMLP(H, edge_mode).to(dt).train()
        ext.load_state_dict({k: v.to(dt) for k, v in oext.state_dict().items()})
        e = emb.to(dt).clone().requires_grad_(True)
        a = oops.concrete_sample(ext(e, ei, batch, masks=[m.to(dt) for m in masks]), u.to(dt), True)
        a.backward(ga.to(dt))
        ref[dt] = dict(a=a, demb=e.grad, **{k: p.grad for k, p in ext.named_parameters()})
    ext = G.ExtractorMLP(H, edge_mode).to(dev).train()
    ext.load_state_dict(oext.state_dict())
    ed = emb.to(dev).requires_grad_(True)
    _, a = ext.attend(ed, ei.to(dev), batch.to(dev), noise=u.to(dev), dropout_masks=[m.to(dev) for m in masks])
    a.backward(ga.to(dev))
    r32, r64 = ref[torch.float32], ref[torch.float64]
    close(a, r32["a"], ref64=r64["a"], what="att")
    close(ed.grad, r32["demb"], 1e-4, ref64=r64["demb"], what="demb")
    for k, p in ext.named_parameters():
        close(p.grad, r32[k], 1e-4, ref64=r64[k], what=k)
