"""-m gpu: the fused one-launch extractor (csrc/attn_fused.hip) against the staged pipeline it replaces (same C ABI entry point, `fused`
switch) and, through the module protocol, against the CPU oracle: whole-graph tiles, graphs larger than a tile ("big" graphs walked in
slabs), empty graphs, many tiny graphs per tile, ragged widths, edges whose endpoints lie in different graphs."""
import ctypes
import os

import numpy as np
import pytest
import torch

from oracle import modules as om
from oracle import ops as oops
from tests.graphs import random_batch, shuffle_edges
from tests.util import close

pytestmark = pytest.mark.gpu


def _sized_batch(sizes, seed, undirected=True, isolated=True):
    """Random trees + a few extra edges with the given node counts per graph (0 allowed: an empty graph)."""
    rng = np.random.RandomState(seed)
    src, dst, batch, off = [], [], [], 0
    for g, n in enumerate(sizes):
        es = set()
        for v in range(1, n):
            if isolated and rng.rand() < 0.04:
                continue
            es.add((int(rng.randint(0, v)), v))
        for _ in range(int(0.12 * n)):
            a, b = rng.randint(0, max(n, 1), size=2)
            if a != b:
                es.add((int(min(a, b)), int(max(a, b))))
        for (u, v) in sorted(es):
            src += [u + off, v + off] if undirected else [u + off]
            dst += [v + off, u + off] if undirected else [v + off]
        batch += [g] * n
        off += n
    ei = torch.tensor([src, dst], dtype=torch.int64).reshape(2, -1)
    return ei, torch.tensor(batch, dtype=torch.int64), off


def _run(dev, emb, params, ei, batch, G_, edge_mode, training, p, masks, u, fused, seed=7):
    """One gsat_attn_fwd call through the C ABI; returns every output / saved tensor."""
    from dp_gsat_amd import _lib
    from dp_gsat_amd._lib import call, ptr, stream
    from dp_gsat_amd.graph_index import BatchIndex
    from dp_gsat_amd.ops import _attn_args
    index = BatchIndex(ei.to(dev), emb.shape[0])
    seg = index.graphs(batch.to(dev), G_)
    N, H = emb.shape
    C1, C2 = params[0].shape[0], params[2].shape[0]
    M = index.E if edge_mode else N
    f32 = torch.float32
    P = torch.full((N, C1), float("nan"), dtype=f32, device=dev)
    Q = torch.full((N, C1), float("nan"), dtype=f32, device=dev) if edge_mode else None
    a1 = torch.full((M, C1), float("nan"), dtype=f32, device=dev)
    h2 = torch.full((M, C2), float("nan"), dtype=f32, device=dev)
    stats = torch.full((max(G_, 1) * (2 * C1 + 2 * C2),), float("nan"), dtype=f32, device=dev)
    logits = torch.full((M, 1), float("nan"), dtype=f32, device=dev)
    att = torch.full((M, 1), float("nan"), dtype=f32, device=dev)
    m1, m2 = masks if masks is not None else (None, None)
    args = _attn_args(emb, params, index, seg, edge_mode, training, p, seed, m1, m2, u, (P, Q, a1, h2, stats, logits, att), None, u is None and training)
    args.fused = 1 if fused else -1
    n = int(_lib.load().gsat_attn_fwd_workspace_bytes(ctypes.byref(args)))
    ws = torch.empty(max(n, 16), dtype=torch.uint8, device=dev)
    args.fwd_workspace, args.fwd_workspace_bytes = ptr(ws), n
    call("gsat_attn_fwd", ctypes.byref(args), stream())
    torch.cuda.synchronize()
    return dict(P=P, Q=Q, a1=a1, h2=h2, stats=stats, logits=logits, att=att)


def _params(H, edge_mode, dev, seed, C1=None, C2=None):
    g = torch.Generator().manual_seed(seed)
    C0 = 2 * H if edge_mode else H
    C1 = C1 or (4 * H if edge_mode else 2 * H)
    C2 = C2 or H
    mk = lambda *s: (torch.randn(*s, generator=g) / (s[-1] ** 0.5)).to(dev)
    return (mk(C1, C0), mk(C1) * 0.1, mk(C2, C1), mk(C2) * 0.1, mk(1, C2), mk(1) * 0.1)


CASES = {
    # name: graph sizes (nodes)
    "molecules": [25, 31, 12, 40, 96, 7, 18, 22, 64, 33, 29, 3, 2, 51],
    "big_graphs": [20, 165, 30, 417, 9, 129, 128, 40],
    "empty_and_tiny": [0, 1, 1, 0, 2, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 0, 3, 0],
    "one_graph": [50],
}


@pytest.mark.parametrize("case", list(CASES))
@pytest.mark.parametrize("H", [16, 64, 80, 128, 256])
@pytest.mark.parametrize("edge_mode", [False, True])
def test_fused_forward_equals_staged_pipeline(dev, case, H, edge_mode):
    sizes = CASES[case]
    ei, batch, N = _sized_batch(sizes, seed=H + len(sizes))
    ei = shuffle_edges(ei, 3)
    if edge_mode and ei.shape[1] == 0:
        pytest.skip("no edges")
    G_ = len(sizes)
    g = torch.Generator().manual_seed(5)
    emb = torch.randn(N, H, generator=g).to(dev)
    params = _params(H, edge_mode, dev, 11)
    M = ei.shape[1] if edge_mode else N
    C1, C2 = params[0].shape[0], params[2].shape[0]
    masks = ((torch.rand(M, C1, generator=g) > 0.5).float().to(dev), (torch.rand(M, C2, generator=g) > 0.5).float().to(dev))
    u = torch.rand(M, generator=g).clamp_(1e-6, 1 - 1e-6).to(dev)
    a = _run(dev, emb, params, ei, batch, G_, edge_mode, True, 0.5, masks, u, fused=True)
    b = _run(dev, emb, params, ei, batch, G_, edge_mode, True, 0.5, masks, u, fused=False)
    for k in ("P", "Q", "h2", "a1", "logits", "att"):
        if a[k] is None:
            continue
        assert not torch.isnan(a[k]).any(), f"{k}: the fused forward left rows unwritten"
        close(a[k], b[k], 5e-5, what=k)          # two fp32 summation orders, seen through 1/sigma of the InstanceNorms
    # statistics: mean | rstd of both layers for every graph, empty graphs included (mean 0, rstd 1/sqrt(eps))
    assert not torch.isnan(a["stats"]).any()
    sa, sb = a["stats"], b["stats"]
    n1 = G_ * C1
    close(sa[:n1], sb[:n1], 2e-5, what="mean1")
    close(sa[2 * n1:2 * n1 + G_ * C2], sb[2 * n1:2 * n1 + G_ * C2], 2e-5, what="mean2")
    # 1/sigma amplifies the rounding of a nearly constant channel (var ~ 1e-7 next to eps = 1e-5): compare the variances
    for lo, hi, what in ((n1, 2 * n1, "rstd1"), (2 * n1 + G_ * C2, 2 * n1 + 2 * G_ * C2, "rstd2")):
        va, vb = sa[lo:hi].double() ** -2, sb[lo:hi].double() ** -2
        close(va, vb, 2e-5, what=what)


def test_fused_forward_is_bitwise_reproducible_and_philox_matches_masks(dev):
    from dp_gsat_amd._lib import call, ptr, stream
    sizes = [25, 31, 140, 12, 40, 7, 18]
    ei, batch, N = _sized_batch(sizes, seed=2)
    H = 64
    emb = torch.randn(N, H).to(dev)
    for edge_mode in (False, True):
        params = _params(H, edge_mode, dev, 3)
        M = ei.shape[1] if edge_mode else N
        C1, C2 = params[0].shape[0], params[2].shape[0]
        a = _run(dev, emb, params, ei, batch, len(sizes), edge_mode, True, 0.5, None, None, fused=True, seed=99)
        b = _run(dev, emb, params, ei, batch, len(sizes), edge_mode, True, 0.5, None, None, fused=True, seed=99)
        for k in ("logits", "att", "h2", "a1"):
            assert torch.equal(a[k], b[k]), k
        m1 = torch.empty(M, C1, device=dev); m2 = torch.empty(M, C2, device=dev); u = torch.empty(M, device=dev)
        call("gsat_philox_keep_mask", 99, 1, M, C1, 0.5, ptr(m1), stream())
        call("gsat_philox_keep_mask", 99, 2, M, C2, 0.5, ptr(m2), stream())
        call("gsat_philox_noise", 99, M, ptr(u), stream())
        c = _run(dev, emb, params, ei, batch, len(sizes), edge_mode, True, 0.5, (m1, m2), u, fused=True, seed=99)
        for k in ("logits", "att", "h2", "a1"):
            assert torch.equal(a[k], c[k]), k


def test_fused_forward_with_edges_across_graphs(dev):
    """An edge whose destination lies in another graph (never produced by PyG collation, accepted by the reference's gathers): the tile
    computes that endpoint's projection on the side."""
    sizes = [20, 30, 25, 10]
    ei, batch, N = _sized_batch(sizes, seed=4)
    extra = torch.tensor([[3, 60, 22], [40, 5, 70]], dtype=torch.int64)        # source graph != destination graph
    ei = torch.cat([ei, extra], dim=1)
    H = 32
    emb = torch.randn(N, H).to(dev)
    params = _params(H, True, dev, 8)
    a = _run(dev, emb, params, ei, batch, len(sizes), True, False, 0.0, None, None, fused=True)
    b = _run(dev, emb, params, ei, batch, len(sizes), True, False, 0.0, None, None, fused=False)
    for k in ("h2", "a1", "logits", "att"):
        close(a[k], b[k], 2e-5, what=k)


@pytest.mark.parametrize("edge_mode", [False, True])
def test_fused_extractor_module_vs_oracle_with_big_graph(dev, edge_mode, monkeypatch):
    """Module protocol end to end (fused forward + backward) on a batch holding a graph larger than a tile."""
    import dp_gsat_amd as G
    monkeypatch.setenv("GSAT_ATTN_FUSED", "1")
    H = 64
    sizes = [30, 200, 12, 45, 0, 70]
    ei, batch, N = _sized_batch(sizes, seed=9)
    ei = shuffle_edges(ei, 1)
    M = ei.shape[1] if edge_mode else N
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
        ext.train()
        e = emb.to(dt).clone().requires_grad_(True)
        z = ext(e, ei, batch, masks=[m.to(dt) for m in masks])
        a = oops.concrete_sample(z, u.to(dt), True)
        torch.autograd.backward([z, a], [gz.to(dt), ga.to(dt)])
        ref[dt] = dict(z=z, a=a, demb=e.grad, **{k: p.grad for k, p in ext.named_parameters()})
    ext = G.ExtractorMLP(H, edge_mode).to(dev)
    ext.load_state_dict(oext.state_dict())
    ext.train()
    ed = emb.to(dev).requires_grad_(True)
    z, a = ext.attend(ed, ei.to(dev), batch.to(dev), noise=u.to(dev), dropout_masks=[m.to(dev) for m in masks])
    torch.autograd.backward([z, a], [gz.to(dev), ga.to(dev)])
    r32, r64 = ref[torch.float32], ref[torch.float64]
    close(z, r32["z"], ref64=r64["z"], what="logits")
    close(a, r32["a"], ref64=r64["a"], what="att")
    close(ed.grad, r32["demb"], ref64=r64["demb"], what="demb")
    for k, p in ext.named_parameters():
        close(p.grad, r32[k], ref64=r64[k], what=k)


def _fwd_bwd(dev, emb, params, ei, batch, G_, edge_mode, masks, u, dlogits, datt, fused_bwd, monkeypatch, p=0.5, training=True):
    """gsat_attn_fwd (staged) + gsat_attn_bwd through the C ABI; returns the gradients."""
    from dp_gsat_amd import _lib
    from dp_gsat_amd._lib import AttnGrads, call, ptr, stream
    from dp_gsat_amd.graph_index import BatchIndex
    from dp_gsat_amd.ops import _attn_args
    monkeypatch.setenv("GSAT_ATTN_BWD_FUSED", "1" if fused_bwd else "0")
    index = BatchIndex(ei.to(dev), emb.shape[0])
    seg = index.graphs(batch.to(dev), G_)
    N, H = emb.shape
    C1, C2 = params[0].shape[0], params[2].shape[0]
    M = index.E if edge_mode else N
    f32 = torch.float32
    mk = lambda *s: torch.empty(*s, dtype=f32, device=dev)
    bufs = (mk(N, C1), mk(N, C1) if edge_mode else None, mk(M, C1), mk(M, C2), mk(max(G_, 1) * (2 * C1 + 2 * C2)), mk(M, 1), mk(M, 1))
    m1, m2 = masks if masks is not None else (None, None)
    args = _attn_args(emb, params, index, seg, edge_mode, training, p, 7, m1, m2, u, bufs, None, u is None and training)
    args.fused = -1
    n = int(_lib.load().gsat_attn_fwd_workspace_bytes(ctypes.byref(args)))
    ws = torch.empty(max(n, 16), dtype=torch.uint8, device=dev)
    args.fwd_workspace, args.fwd_workspace_bytes = ptr(ws), n
    call("gsat_attn_fwd", ctypes.byref(args), stream())
    g = AttnGrads()
    g.dlogits, g.datt = ptr(dlogits), ptr(datt)
    if edge_mode:
        g.rowptr_src, g.eid_by_src = ptr(index.rowptr_src), ptr(index.eid_by_src)
        g.rowptr_dst, g.eid_by_dst = ptr(index.rowptr_dst), ptr(index.eid_by_dst)
        g.chunk_ptr_dst, g.chunk_ptr_src = (ptr(t) for t in index.long_rows)
    demb = torch.full_like(emb, float("nan"))
    grads = [torch.full_like(t, float("nan")) for t in params]
    g.demb = ptr(demb)
    g.dW1, g.db1, g.dW2, g.db2, g.dW3, g.db3 = (ptr(t) for t in grads)
    nb = int(_lib.load().gsat_attn_bwd_workspace_bytes(ctypes.byref(args)))
    wsb = torch.empty(nb, dtype=torch.uint8, device=dev)
    g.workspace, g.workspace_bytes = ptr(wsb), nb
    call("gsat_attn_bwd", ctypes.byref(args), ctypes.byref(g), stream())
    torch.cuda.synchronize()
    return dict(demb=demb, dW1=grads[0], db1=grads[1], dW2=grads[2], db2=grads[3], dW3=grads[4], db3=grads[5])


BWD_CASES = {
    "molecules": [25, 31, 12, 40, 96, 7, 18, 22, 64, 33, 29, 3, 2, 51, 17, 128, 1, 0, 45],
    "tiny": [0, 1, 1, 2, 1, 3, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 0, 5, 0],
    "big_graphs": [20, 165, 30, 417, 9, 129, 128, 40],
}


@pytest.mark.parametrize("case", list(BWD_CASES))
@pytest.mark.parametrize("H", [16, 64, 80, 128])
@pytest.mark.parametrize("edge_mode", [False, True])
@pytest.mark.parametrize("training", [True, False])
def test_fused_backward_equals_staged_backward(dev, case, H, edge_mode, training, monkeypatch):
    sizes = BWD_CASES[case]
    ei, batch, N = _sized_batch(sizes, seed=H + 1)
    ei = shuffle_edges(ei, 3)
    if edge_mode and (ei.shape[1] == 0 or 4 * H > 512):
        pytest.skip("no edges / layer-1 width beyond the fused backward")
    G_ = len(sizes)
    g = torch.Generator().manual_seed(5)
    emb = torch.randn(N, H, generator=g).to(dev)
    params = _params(H, edge_mode, dev, 11)
    M = ei.shape[1] if edge_mode else N
    C1, C2 = params[0].shape[0], params[2].shape[0]
    masks = ((torch.rand(M, C1, generator=g) > 0.5).float().to(dev), (torch.rand(M, C2, generator=g) > 0.5).float().to(dev)) if training else None
    u = torch.rand(M, generator=g).clamp_(1e-6, 1 - 1e-6).to(dev) if training else None
    dl, da = torch.randn(M, generator=g).to(dev), torch.randn(M, generator=g).to(dev)
    a = _fwd_bwd(dev, emb, params, ei, batch, G_, edge_mode, masks, u, dl, da, True, monkeypatch, training=training)
    b = _fwd_bwd(dev, emb, params, ei, batch, G_, edge_mode, masks, u, dl, da, False, monkeypatch, training=training)
    for k in a:
        assert not torch.isnan(a[k]).any(), f"{k}: unwritten entries"
        close(a[k], b[k], 1e-4, what=k)          # both paths run the split-bf16 policy; orders of summation differ


@pytest.mark.parametrize("case", ["molecules", "big_graphs"])
@pytest.mark.parametrize("H", [16, 64, 80, 128])
@pytest.mark.parametrize("edge_mode", [False, True])
def test_dual_gemm_backward_equals_separate_gemms(dev, case, H, edge_mode, monkeypatch):
    """The extractor backward's GEMM pairs (da1 / dW2, demb / dW1) through the dual tile kernel (csrc/dual_gemm.hip) against the separate
    split-bf16 GEMMs + slab reductions they replace: same precision policy, different summation orders."""
    sizes = BWD_CASES[case]
    ei, batch, N = _sized_batch(sizes, seed=H + 2)
    ei = shuffle_edges(ei, 4)
    if edge_mode and ei.shape[1] == 0:
        pytest.skip("no edges")
    G_ = len(sizes)
    g = torch.Generator().manual_seed(6)
    emb = torch.randn(N, H, generator=g).to(dev)
    params = _params(H, edge_mode, dev, 12)
    M = ei.shape[1] if edge_mode else N
    C1, C2 = params[0].shape[0], params[2].shape[0]
    masks = ((torch.rand(M, C1, generator=g) > 0.5).float().to(dev), (torch.rand(M, C2, generator=g) > 0.5).float().to(dev))
    u = torch.rand(M, generator=g).clamp_(1e-6, 1 - 1e-6).to(dev)
    dl, da = torch.randn(M, generator=g).to(dev), torch.randn(M, generator=g).to(dev)
    monkeypatch.setenv("GSAT_DUAL_GEMM", "1")
    a = _fwd_bwd(dev, emb, params, ei, batch, G_, edge_mode, masks, u, dl, da, False, monkeypatch)
    monkeypatch.setenv("GSAT_DUAL_GEMM", "0")
    b = _fwd_bwd(dev, emb, params, ei, batch, G_, edge_mode, masks, u, dl, da, False, monkeypatch)
    for k in a:
        assert not torch.isnan(a[k]).any(), f"{k}: unwritten entries"
        close(a[k], b[k], 1e-4, what=k)
    a2 = _fwd_bwd(dev, emb, params, ei, batch, G_, edge_mode, masks, u, dl, da, False, monkeypatch)          # (dual off twice: the harness is deterministic)
    assert all(torch.equal(a2[k], b[k]) for k in b)
    monkeypatch.setenv("GSAT_DUAL_GEMM", "1")
    c = _fwd_bwd(dev, emb, params, ei, batch, G_, edge_mode, masks, u, dl, da, False, monkeypatch)
    assert all(torch.equal(a[k], c[k]) for k in a), "the dual GEMM path must be bitwise reproducible"
