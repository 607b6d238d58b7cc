"""-m gpu: PNA multi-aggregation fwd/bwd through the C ABI vs the CPU oracle."""
import pytest
import torch

from oracle import bookkeeping as obk
from oracle import ops as oops
from tests.graphs import random_batch, shuffle_edges
from tests.util import TOL, close

pytestmark = pytest.mark.gpu

ALL_AGG = ["mean", "min", "max", "std", "sum", "var"]


def std_conditioning(x, ei, att, N, kappa_ok=1e3):
    """Per-entry tolerance weights (>= 1) for dx [N,H] and datt [E].  var = E[m^2] - E[m]^2 of the message (the reference's formula,
    src/models/conv_layers.py:205-216) loses log10(kappa) digits to cancellation, kappa = E[m^2] / (var + 1e-5): any two fp32
    evaluations of std's gradient differ by ~6e-8 * kappa there (unless every message of the row is the same value, where (m - mean) = 0
    kills the term).  Up to kappa_ok that is inside the plain tolerance; beyond it the entries that receive the affected gradient -- the
    row itself, the sources of its in-edges, its in-edges' attention -- are compared at tolerance * kappa / kappa_ok."""
    src, dst = ei[0], ei[1]
    E = ei.shape[1]
    w = att.double().view(-1, 1) if att is not None else torch.ones(E, 1, dtype=torch.float64)
    xd = x.double()
    cnt = torch.zeros(N, 1, dtype=torch.float64).index_add_(0, dst, torch.ones(E, 1, dtype=torch.float64)).clamp(min=1)
    kap = torch.ones(N, x.shape[1], dtype=torch.float64)
    for m in (w * xd[src], w * xd[dst]):
        mean = torch.zeros_like(xd).index_add_(0, dst, m) / cnt
        msq = torch.zeros_like(xd).index_add_(0, dst, m * m) / cnt
        var = (msq - mean * mean).clamp(min=0)
        k = torch.where(var > 1e-9 * msq, msq / (var + 1e-5), torch.ones_like(var))
        kap = torch.maximum(kap, k / kappa_ok)
    w_dx = kap.clone()
    w_dx.scatter_reduce_(0, src.view(-1, 1).expand(-1, x.shape[1]), kap[dst], "amax")          # a source row receives the gradient of every row it feeds
    w_e = kap.amax(dim=1)[dst]
    return w_dx, w_e


def close_weighted(actual, r32, r64, weight, what):
    a, r32, r64 = actual.detach().cpu().double(), r32.double(), r64.double()
    weight = weight.view(a.shape)
    scale = max(1.0, r32.abs().max().item())
    allowed = TOL * scale + 4.0 * ((r32 - r64).abs() / weight).max().item()
    err = ((a - r64).abs() / weight).max().item()
    assert err <= allowed, f"{what}: max conditioning-weighted err {err:.3e} > allowed {allowed:.3e} (scale {scale:.3e}, worst weight {weight.max().item():.1f})"


@pytest.mark.parametrize("H", [8, 80, 128, 256, 320, 512])      # above 256: one launch per 256-channel chunk (320 = 256 + a 16-lane chunk)
@pytest.mark.parametrize("aggr,scalers", [
    (["mean", "min", "max", "std"], ["identity"]),                                   # src/configs/PNA-ogbg_mol.yml
    (["mean", "min", "max", "std", "sum"], ["identity"]),                            # src/configs/PNA-mutag.yml
    (ALL_AGG, ["identity", "amplification", "attenuation"]),                         # scalers: true
    (["sum", "var"], ["linear", "inverse_linear"]),
])
@pytest.mark.parametrize("masked", [True, False])
def test_pna_fwd_bwd(dev, H, aggr, scalers, masked):
    from dp_gsat_amd.graph_index import BatchIndex
    from dp_gsat_amd.ops import pna_aggregate
    ei, batch, N = random_batch(3 + H, 8, 1, 30)
    ei = shuffle_edges(ei, 1)
    E = ei.shape[1]
    g = torch.Generator().manual_seed(H + len(aggr))
    x = torch.randn(N, H, generator=g)
    x[::5] = x[::5].relu()          # exact zeros, as after ReLU: exercises the x_i == 0 arg rule
    att = torch.rand(E, 1, generator=g) if masked else None
    avg = oops.pna_avg_deg(torch.from_numpy(obk.deg_histogram(ei, N)))
    F = 2 * H
    go = torch.randn(N, len(scalers) * len(aggr) * F, generator=g)
    ref = {}
    for dt in (torch.float32, torch.float64):
        xo = x.to(dt).clone().requires_grad_(True)
        ao = att.to(dt).clone().requires_grad_(True) if masked else None
        oo = oops.pna_aggregate(xo, ei, ao, aggr, scalers, avg)
        oo.backward(go.to(dt))
        ref[dt] = (oo, xo.grad, ao.grad if masked else None)
    ix = BatchIndex(ei.to(dev), N)
    xd = x.to(dev).requires_grad_(True)
    ad = att.to(dev).requires_grad_(True) if masked else None
    od = pna_aggregate(xd, ix, ad, None, aggr, scalers, avg)
    od.backward(go.to(dev))
    r32, r64 = ref[torch.float32], ref[torch.float64]
    close(od, r32[0], ref64=r64[0], what="out")       # std of (nearly) identical messages: sqrt(1e-5 + rounding noise of var) in ANY fp32 evaluation
    w_dx, w_e = std_conditioning(x, ei, att, N)
    close_weighted(xd.grad, r32[1], r64[1], w_dx, "dx")
    if masked:
        close_weighted(ad.grad, r32[2], r64[2], w_e, "datt")


@pytest.mark.parametrize("H", [32, 80, 128, 384])
@pytest.mark.parametrize("aggr", [["mean", "min", "max", "std"], ["mean", "min", "max", "std", "sum"], ["max", "var", "mean"]])
def test_pna_with_edge_attr(dev, H, aggr):
    """message = att * [x_i || x_j || edge_emb] (PNA on spmotif / mnist: edge_attr present, src/models/conv_layers.py:168-171); the
    first two aggregator lists take the compile-time instantiations, the third the generic kernel."""
    from dp_gsat_amd.graph_index import BatchIndex
    from dp_gsat_amd.ops import pna_aggregate
    ei, batch, N = random_batch(5 + H, 6, 2, 25)
    E = ei.shape[1]
    g = torch.Generator().manual_seed(9)
    x, ee, att = torch.randn(N, H, generator=g), torch.randn(E, H, generator=g), torch.rand(E, 1, generator=g)
    x[::4] = x[::4].relu()
    scalers = ["identity"]
    avg = oops.pna_avg_deg(torch.from_numpy(obk.deg_histogram(ei, N)))
    go = torch.randn(N, len(aggr) * 3 * H, generator=g)
    ref = {}
    for dt in (torch.float32, torch.float64):
        xo, eo, ao = (t.to(dt).clone().requires_grad_(True) for t in (x, ee, att))
        oo = oops.pna_aggregate(xo, ei, ao, aggr, scalers, avg, eo)
        oo.backward(go.to(dt))
        ref[dt] = (oo, xo.grad, eo.grad, ao.grad)
    ix = BatchIndex(ei.to(dev), N)
    xd, ed, ad = (t.to(dev).requires_grad_(True) for t in (x, ee, att))
    od = pna_aggregate(xd, ix, ad, ed, aggr, scalers, avg)
    od.backward(go.to(dev))
    r32, r64 = ref[torch.float32], ref[torch.float64]
    close(od, r32[0], ref64=r64[0], what="out")
    close(xd.grad, r32[1], 1e-4, ref64=r64[1], what="dx")
    close(ed.grad, r32[2], 1e-4, ref64=r64[2], what="dedge")
    close(ad.grad, r32[3], 1e-4, ref64=r64[3], what="datt")


def test_pna_empty_rows_std(dev):
    """std of an empty neighbourhood is sqrt(1e-5), min/max/mean are 0 (src/models/conv_layers.py:215-216)."""
    from dp_gsat_amd.graph_index import BatchIndex
    from dp_gsat_amd.ops import pna_aggregate
    N, H = 5, 8
    ei = torch.tensor([[0, 1], [1, 0]], dtype=torch.int64)
    x = torch.randn(N, H)
    ix = BatchIndex(ei.to(dev), N)
    out = pna_aggregate(x.to(dev), ix, None, None, ["mean", "min", "max", "std"], ["identity"], {"lin": 1.0, "log": 1.0}).cpu()
    assert torch.all(out[2:, : 3 * 2 * H] == 0)
    assert torch.allclose(out[2:, 3 * 2 * H:], torch.full((3, 2 * H), 1e-5 ** 0.5))


@pytest.mark.parametrize("H", [64, 80, 128, 256])
@pytest.mark.parametrize("lds_budget", [0, 6144])            # default windows / tiny windows: most edges spill, hub rows overflow the LDS edge capacity
@pytest.mark.parametrize("aligned", [True, False])
def test_pna_tiled_backward(dev, monkeypatch, H, lds_budget, aligned):
    """The one-launch (LDS-resident) PNA backward vs the oracle and vs the two-pass backward: graph-aligned and fixed windows,
    windows much smaller than the graphs (spilled edges), a hub row longer than a window's edge capacity, isolated nodes; bitwise
    reproducible run to run."""
    from dp_gsat_amd.graph_index import BatchIndex
    from dp_gsat_amd.ops import pna_aggregate
    monkeypatch.setenv("GSAT_PNA_TILE_LDS", str(lds_budget))
    ei, batch, N = random_batch(11 + H, 24, 1, 40)
    hub = torch.arange(1, 200)                                 # a star inside graph 0's id range is not block-diagonal: fine, tiles are windows
    star = torch.stack([torch.cat([hub, torch.zeros_like(hub)]), torch.cat([torch.zeros_like(hub), hub])]) + (N - 200 if N > 400 else 0)
    ei = shuffle_edges(torch.cat([ei, star.clamp_(max=N - 1)], dim=1), 2)
    E = ei.shape[1]
    g = torch.Generator().manual_seed(H)
    x = torch.randn(N, H, generator=g)
    x[::5] = x[::5].relu()
    att = torch.rand(E, 1, generator=g)
    aggr = ["mean", "min", "max", "std", "sum"] if H == 80 else ["mean", "min", "max", "std"]
    avg = {"lin": 1.0, "log": 1.0}
    go = torch.randn(N, len(aggr) * 2 * H, generator=g)
    ref = {}
    for dt in (torch.float32, torch.float64):
        xo, ao = x.to(dt).clone().requires_grad_(True), att.to(dt).clone().requires_grad_(True)
        oops.pna_aggregate(xo, ei, ao, aggr, ["identity"], avg).backward(go.to(dt))
        ref[dt] = (xo.grad, ao.grad)

    def run(tiled):
        monkeypatch.setenv("GSAT_PNA_TILED", "1" if tiled else "0")
        ix = BatchIndex(ei.to(dev), N)
        if aligned:
            ix.graphs(batch.to(dev))
        if tiled and ix.pna_tiles(H):          # False when the LDS budget is too small for this width: ops falls back to the two-pass path
            tile_ptr, T, rows_nominal, rows_cap, edges_cap, spill = ix.pna_tiles(H)
            tp = tile_ptr[:, 0].cpu()
            assert torch.equal(tile_ptr[:, 1].cpu(), ix.rowptr_dst.cpu()[tp.long()]) and torch.equal(tile_ptr[:, 2].cpu(), ix.rowptr_src.cpu()[tp.long()])
            assert int(tp[0]) == 0 and int(tp[-1]) == N and bool((tp[1:] >= tp[:-1]).all()) and int((tp[1:] - tp[:-1]).max()) <= rows_cap
        xd, ad = x.to(dev).requires_grad_(True), att.to(dev).requires_grad_(True)
        pna_aggregate(xd, ix, ad, None, aggr, ["identity"], avg).backward(go.to(dev))
        return xd.grad, ad.grad

    dx, da = run(True)
    close(dx, ref[torch.float32][0], ref64=ref[torch.float64][0], what="dx")
    close(da, ref[torch.float32][1], ref64=ref[torch.float64][1], what="datt")
    dx2, da2 = run(True)
    assert torch.equal(dx, dx2) and torch.equal(da, da2)                      # bitwise reproducible
    dx0, da0 = run(False)                                                      # the two-pass backward: same sums, other order
    close(da, da0, 1e-5, what="datt tiled vs two-pass")
    close(dx, dx0, 1e-5, what="dx tiled vs two-pass")


@pytest.mark.parametrize("H", [64, 80, 128, 256])
@pytest.mark.parametrize("lds_budget", [0, 6144])            # default windows / tiny windows: most edges spill, hub rows overflow the LDS edge capacity
@pytest.mark.parametrize("aligned", [True, False])
def test_pna_node_attention_without_lift(dev, monkeypatch, H, lds_budget, aligned):
    """Node attention formed inside the aggregation kernels (gsat_pna_*_node_att, ops.LiftedAttention) vs the lifted [E, 1] tensor
    through the same kernels (forward output and dx bit-identical: the edge weight is the same product) and vs the oracle
    (example/gsat.py:112-117 + conv_layers.py:166-185); d node_att sums in CSR order instead of the lift kernel's, equal within fp32
    rounding; spilled edges, a hub row, isolated nodes, rows without in-edges; bitwise reproducible."""
    from dp_gsat_amd.graph_index import BatchIndex
    from dp_gsat_amd.ops import LiftedAttention, PnaAggregate, pna_aggregate
    monkeypatch.setenv("GSAT_PNA_TILE_LDS", str(lds_budget))
    monkeypatch.setattr("dp_gsat_amd.graph_index._HUBS_SEEN", [False])      # (an earlier test's hub rows would route this batch to the chunked path until its status lands)
    ei, batch, N = random_batch(17 + H, 24, 1, 40)
    hub = torch.arange(1, 200)
    star = torch.stack([torch.cat([hub, torch.zeros_like(hub)]), torch.cat([torch.zeros_like(hub), hub])]) + (N - 200 if N > 400 else 0)
    ei = shuffle_edges(torch.cat([ei, star.clamp_(max=N - 1)], dim=1), 2)
    g = torch.Generator().manual_seed(H + 1)
    x = torch.randn(N, H, generator=g)
    x[::5] = x[::5].relu()
    na = torch.rand(N, 1, generator=g)
    na[::7] = 0.0                                              # zero attention: every edge of the node carries weight 0
    aggr = ["mean", "min", "max", "std", "sum"] if H == 80 else ["mean", "min", "max", "std"]
    avg = {"lin": 1.0, "log": 1.0}
    go = torch.randn(N, len(aggr) * 2 * H, generator=g)
    ref = {}
    for dt in (torch.float32, torch.float64):
        xo, ao = x.to(dt).clone().requires_grad_(True), na.to(dt).clone().requires_grad_(True)
        out = oops.pna_aggregate(xo, ei, oops.lift_node_att_to_edge_att(ao, ei), aggr, ["identity"], avg)
        out.backward(go.to(dt))
        ref[dt] = (out.detach(), xo.grad, ao.grad)

    def run(lift_free):
        ix = BatchIndex(ei.to(dev), N)
        if aligned:
            ix.graphs(batch.to(dev))
        xd, ad = x.to(dev).requires_grad_(True), na.to(dev).requires_grad_(True)
        calls.clear()
        att = LiftedAttention(ad, ix)
        if not lift_free:
            att = att.edge()
        out = pna_aggregate(xd, ix, att, None, aggr, ["identity"], avg)
        out.backward(go.to(dev))
        return out.detach(), xd.grad, ad.grad, list(calls)

    from dp_gsat_amd import _lib
    calls, real = [], _lib.call
    monkeypatch.setattr("dp_gsat_amd.ops.call", lambda name, *a: (calls.append(name), real(name, *a))[1])
    tiles = bool(BatchIndex(ei.to(dev), N).pna_tiles(H))
    out, dx, dna, names = run(True)
    if tiles:                  # (a budget too small for the width: pna_aggregate writes the tensor out and takes the two-pass path)
        assert names == ["gsat_pna_fwd_node_att", "gsat_pna_bwd_tiled_node_att"], names            # no lift kernel, no [E] tensor
    close(out, ref[torch.float32][0], ref64=ref[torch.float64][0], what="out")
    close(dx, ref[torch.float32][1], ref64=ref[torch.float64][1], what="dx")
    close(dna, ref[torch.float32][2], ref64=ref[torch.float64][2], what="d node_att")
    out2, dx2, dna2, _ = run(True)
    assert torch.equal(out, out2) and torch.equal(dx, dx2) and torch.equal(dna, dna2)          # bitwise reproducible
    out0, dx0, dna0, names0 = run(False)
    assert "gsat_lift_fwd" in names0 and "gsat_lift_bwd" in names0
    assert torch.equal(out, out0)                                                                # same products, same order
    if tiles:
        assert torch.equal(dx, dx0)
    close(dna, dna0, 1e-5, what="d node_att: in-kernel vs lift backward")


def test_pna_tiled_backward_under_back_to_back_graph_replays(dev):
    """The whole per-batch pipeline (index build -> window build with its spill-source list -> PNA forward -> tiled backward)
    captured into one hipGraph and replayed back to back WITHOUT host syncs must reproduce the eager result bit for bit on every
    replay.  (Round 2 found hipMemsetAsync nodes racing with the neighbouring replay's kernels: the spill counter was reset out
    of order, overflowed its list and faulted; the library now zeroes with kernels only.)"""
    import dp_gsat_amd as G
    from dp_gsat_amd import synth
    from dp_gsat_amd.ops import pna_aggregate
    data = synth.molhiv_batch(512, seed=3).to(dev)
    N, E, H = data.num_nodes, data.num_edges, 128
    aggr, avg = ["mean", "min", "max", "std"], {"lin": 1.0, "log": 1.0}
    g = torch.Generator(device=dev).manual_seed(1)
    x = torch.randn(N, H, device=dev, generator=g).requires_grad_(True)
    att = torch.rand(E, 1, device=dev, generator=g).requires_grad_(True)
    go = torch.randn(N, 8 * H, device=dev, generator=g)
    dx_buf, da_buf = torch.zeros(N, H, device=dev), torch.zeros(E, 1, device=dev)

    def fn():
        G.clear_cache()
        ix = G.get_index(data.edge_index, N)
        ix.graphs(data.batch, data.num_graphs)
        x.grad = None
        att.grad = None
        pna_aggregate(x, ix, att, None, aggr, ["identity"], avg).backward(go)
        dx_buf.copy_(x.grad)
        da_buf.copy_(att.grad)
        held["spill"] = ix.pna_tiles(H)[5]          # [0] = number of listed spill sources, [1:] = the list (capacity N)

    held = {}
    fn()
    torch.cuda.synchronize()
    want_dx, want_da = dx_buf.clone(), da_buf.clone()
    G.set_sync_free(True)
    try:
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(3):
                fn()
        torch.cuda.current_stream().wait_stream(side)
        from tests.util import assert_no_memset_nodes, capture_with_dump
        graph, dot = capture_with_dump(fn)
        assert_no_memset_nodes(dot, "PNA pipeline")
        for r in range(3):
            dx_buf.zero_(); da_buf.zero_()
            for _ in range(25):
                graph.replay()
            torch.cuda.synchronize()
            assert torch.equal(dx_buf, want_dx) and torch.equal(da_buf, want_da), f"round {r}"
            # the counter doubles as the overflow record: k_pna_spill_rows keeps counting past the list's capacity (and drops the
            # entry), so a value above N would mean dropped spill rows, i.e. a silently wrong dx
            assert 0 <= int(held["spill"][0]) <= N, f"round {r}: spill list overflowed"
    finally:
        G.set_sync_free(False)
        G.clear_cache()


@pytest.mark.parametrize("H,Ho", [(64, 64), (128, 128), (128, 32), (256, 64)])
@pytest.mark.parametrize("mode", ["none", "edge", "node"])
def test_pna_conv_on_the_compact_aggregate(dev, monkeypatch, H, Ho, mode):
    """PNAConvSimple with one post_nn Linear as one autograd node on the compact aggregate (gsat_pna_fwd_compact + gsat_pna_post_fwd /
    _dw: the x_i half of [N, A*2*H] is rebuilt in the GEMM's operand loader) vs the two-op path (aggregate, then Linear) and vs the oracle
    (src/models/conv_layers.py:148-153,163-185): unit weights, an [E,1] attention tensor, lifted node attention; isolated nodes and rows
    without in-edges; bitwise reproducible."""
    from dp_gsat_amd.conv_layers import PNAConvSimple
    from dp_gsat_amd.graph_index import BatchIndex
    from dp_gsat_amd.ops import LiftedAttention
    monkeypatch.setattr("dp_gsat_amd.graph_index._HUBS_SEEN", [False])
    ei, batch, N = random_batch(5 + H + Ho, 40, 1, 40)
    ei = shuffle_edges(ei, 3)
    E = ei.shape[1]
    g = torch.Generator().manual_seed(H + Ho)
    x = torch.randn(N, H, generator=g)
    x[::5] = x[::5].relu()
    att = torch.rand(E, 1, generator=g)
    na = torch.rand(N, 1, generator=g)
    aggr = ["mean", "min", "max", "std"]
    deg = torch.bincount(torch.bincount(ei[1], minlength=N))
    conv = PNAConvSimple(2 * H, Ho, aggr, ["identity"], deg, post_layers=1)
    W, b = conv.post_nn[0].weight.detach().clone(), conv.post_nn[0].bias.detach().clone()
    go = torch.randn(N, Ho, generator=g)
    avg = {"lin": 1.0, "log": 1.0}
    ref = {}
    for dt in (torch.float32, torch.float64):
        xo = x.to(dt).clone().requires_grad_(True)
        ao = (att if mode == "edge" else na).to(dt).clone().requires_grad_(True)
        wo, bo = W.to(dt).clone().requires_grad_(True), b.to(dt).clone().requires_grad_(True)
        w_e = None if mode == "none" else (ao if mode == "edge" else oops.lift_node_att_to_edge_att(ao, ei))
        out = oops.pna_aggregate(xo, ei, w_e, aggr, ["identity"], avg) @ wo.t() + bo
        out.backward(go.to(dt))
        ref[dt] = (out.detach(), xo.grad, None if mode == "none" else ao.grad, wo.grad, bo.grad)

    conv = conv.to(dev)
    from dp_gsat_amd import _lib
    calls, real = [], _lib.call
    monkeypatch.setattr("dp_gsat_amd.ops.call", lambda name, *a: (calls.append(name), real(name, *a))[1])

    def run(compact):
        monkeypatch.setenv("GSAT_PNA_COMPACT", "1" if compact else "0")
        ix = BatchIndex(ei.to(dev), N)
        ix.graphs(batch.to(dev))
        xd = x.to(dev).requires_grad_(True)
        ad = None if mode == "none" else (att if mode == "edge" else na).to(dev).requires_grad_(True)
        w_e = None if mode == "none" else (ad if mode == "edge" else LiftedAttention(ad, ix))
        conv.zero_grad(set_to_none=True)
        calls.clear()
        out = conv(xd, ei.to(dev), None, edge_atten=w_e, index=ix)
        out.backward(go.to(dev))
        return (out.detach(), xd.grad, None if ad is None else ad.grad, conv.post_nn[0].weight.grad.clone(), conv.post_nn[0].bias.grad.clone()), list(calls)

    got, names = run(True)
    assert names[:2] == ["gsat_pna_fwd_compact", "gsat_pna_post_fwd"] and "gsat_pna_post_dw" in names and "gsat_pna_fwd" not in names, names
    for a, r32, r64, what in zip(got, ref[torch.float32], ref[torch.float64], ("out", "dx", "datt", "dW", "db")):
        if a is not None:
            close(a, r32, 3e-5, ref64=r64, what=what)          # (split-bf16 products: ~1e-5 of the result's scale)
    again, _ = run(True)
    for a, b_ in zip(got, again):
        assert a is None or torch.equal(a, b_)                                                     # bitwise reproducible
    two, names2 = run(False)
    assert "gsat_pna_fwd_compact" not in names2
    for a, b_, what in zip(got, two, ("out", "dx", "datt", "dW", "db")):
        if a is not None:
            close(a, b_, 3e-5, what=what + ": compact vs two-op path")


@pytest.mark.parametrize("mode", ["none", "edge", "node"])
@pytest.mark.parametrize("compact", [False, True])
def test_pna_residual_gradient_is_added_inside_the_aggregation_backward(dev, monkeypatch, mode, compact):
    """``conv(..., with_residual_input=True)`` hands x back as an identity output of the aggregation's autograd node; the layer's residual
    (src/models/pna.py:57-59) then sends its gradient THROUGH that node, whose tiled backward adds it in its own dx pass (dx_add): same
    gradients as the plain graph (x used twice, autograd adds), and the two paths' dx differ only by the order of one addition."""
    from dp_gsat_amd.conv_layers import PNAConvSimple
    from dp_gsat_amd.graph_index import BatchIndex
    from dp_gsat_amd.ops import LiftedAttention
    monkeypatch.setattr("dp_gsat_amd.graph_index._HUBS_SEEN", [False])
    monkeypatch.setenv("GSAT_PNA_COMPACT", "1" if compact else "0")
    H = 128
    ei, batch, N = random_batch(77, 40, 1, 40)
    ei = shuffle_edges(ei, 5)
    E = ei.shape[1]
    g = torch.Generator().manual_seed(9)
    x = torch.randn(N, H, generator=g)
    att, na = torch.rand(E, 1, generator=g), torch.rand(N, 1, generator=g)
    deg = torch.bincount(torch.bincount(ei[1], minlength=N))
    conv = PNAConvSimple(2 * H, H, ["mean", "min", "max", "std"], ["identity"], deg, post_layers=1).to(dev)
    go = torch.randn(N, H, generator=g).to(dev)
    gr = torch.randn(N, H, generator=g).to(dev)

    def run(fold):
        ix = BatchIndex(ei.to(dev), N)
        ix.graphs(batch.to(dev))
        xd = x.to(dev).requires_grad_(True)
        ad = None if mode == "none" else (att if mode == "edge" else na).to(dev).requires_grad_(True)
        w_e = None if mode == "none" else (ad if mode == "edge" else LiftedAttention(ad, ix))
        conv.zero_grad(set_to_none=True)
        if fold:
            h, x_res = conv(xd, ei.to(dev), None, edge_atten=w_e, index=ix, with_residual_input=True)
        else:
            h, x_res = conv(xd, ei.to(dev), None, edge_atten=w_e, index=ix), xd
        torch.autograd.backward([h, x_res * 1.5], [go, gr])
        return xd.grad, None if ad is None else ad.grad, conv.post_nn[0].weight.grad.clone()

    a, b = run(True), run(False)
    for u, v, what in zip(a, b, ("dx", "datt", "dW")):
        if u is not None:
            close(u, v, 1e-6, what=what)
    a2 = run(True)
    assert torch.equal(a[0], a2[0])


def test_layers_share_one_node_attention_gradient(dev, monkeypatch):
    """Three aggregation layers on one lifted node attention + another consumer of the same node_att (as the info loss is): the layers add
    their shares of d node_att inside their kernels into one buffer (no [N,1] add launch per layer), autograd adds the other consumer's
    gradient on top; equal to the per-layer buffers bit for bit in the layers' part (same kernels, a different place for the adds) within
    fp32 rounding of the sum order, and a second backward through the retained graph gives the same result again."""
    from dp_gsat_amd.graph_index import BatchIndex
    from dp_gsat_amd.ops import LiftedAttention, pna_aggregate
    monkeypatch.setattr("dp_gsat_amd.graph_index._HUBS_SEEN", [False])
    H, L = 128, 3
    ei, batch, N = random_batch(91, 40, 1, 40)
    ei = shuffle_edges(ei, 7)
    g = torch.Generator().manual_seed(3)
    xs = [torch.randn(N, H, generator=g) for _ in range(L)]
    na = torch.rand(N, 1, generator=g)
    gos = [torch.randn(N, 8 * H, generator=g) for _ in range(L)]
    aggr, avg = ["mean", "min", "max", "std"], {"lin": 1.0, "log": 1.0}

    def run(shared, twice=False):
        monkeypatch.setenv("GSAT_NODE_ATT_SHARED_GRAD", "1" if shared else "0")
        ix = BatchIndex(ei.to(dev), N)
        ix.graphs(batch.to(dev))
        leaf = na.to(dev).requires_grad_(True)
        node_att = leaf * 1.0                                   # a non-leaf, as the sampler's output is
        att = LiftedAttention(node_att, ix)
        outs = [pna_aggregate(x.to(dev), ix, att, None, aggr, ["identity"], avg) for x in xs]
        extra = (node_att * node_att).sum()                     # another consumer of node_att
        torch.autograd.backward(outs + [extra], [g_.to(dev) for g_ in gos] + [None], retain_graph=twice)
        first = leaf.grad.clone()
        if twice:
            leaf.grad = None
            torch.autograd.backward(outs + [extra], [g_.to(dev) for g_ in gos] + [None])
            assert torch.equal(first, leaf.grad)
        return first

    a, b = run(True), run(False)
    close(a, b, 1e-6, what="d node_att: shared buffer vs per-layer buffers")
    run(True, twice=True)
