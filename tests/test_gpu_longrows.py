"""-m gpu: power-law rows (hubs split into 256-edge chunks) and few huge graphs, vs the oracle."""
import numpy as np
import pytest
import torch

from oracle import bookkeeping as obk
from oracle import modules as om
from oracle import ops as oops
from tests.util import close

pytestmark = pytest.mark.gpu


def hub_graph(seed, n=2500, hubs=(0, 5), hub_deg=(1500, 300), extra=2000):
    """Undirected multigraph-free graph where node 0 has ~1500 neighbours, node 5 ~300 (> and ~ the 256 chunk)."""
    rng = np.random.RandomState(seed)
    es = set()
    for h, d in zip(hubs, hub_deg):
        for v in rng.choice(np.arange(n), size=min(d, n - 1), replace=False):
            if v != h:
                es.add((min(h, int(v)), max(h, int(v))))
    while len(es) < sum(hub_deg) // 2 + extra:
        a, b = rng.randint(0, n, size=2)
        if a != b:
            es.add((int(min(a, b)), int(max(a, b))))
    und = np.array(sorted(es), dtype=np.int64).T
    ei = np.concatenate([und, und[::-1]], axis=1)
    perm = rng.permutation(ei.shape[1])
    return torch.from_numpy(ei[:, perm]).contiguous(), n


@pytest.mark.parametrize("H", [16, 64, 256, 512])
@pytest.mark.parametrize("gine", [False, True])
def test_sum_aggregate_with_hubs(dev, H, gine):
    from dp_gsat_amd.graph_index import BatchIndex
    from dp_gsat_amd.ops import masked_sum_aggregate
    ei, N = hub_graph(1)
    E = ei.shape[1]
    assert torch.bincount(ei[1]).max() > 3 * 256
    g = torch.Generator().manual_seed(H)
    x, att, go = torch.randn(N, H, generator=g), torch.rand(E, 1, generator=g), torch.randn(N, H, generator=g)
    ee = torch.randn(E, H, generator=g) if gine else None
    ref = {}
    for dt in (torch.float32, torch.float64):
        xo, ao = x.to(dt).clone().requires_grad_(True), att.to(dt).clone().requires_grad_(True)
        eo = ee.to(dt).clone().requires_grad_(True) if gine else None
        oo = oops.gine_aggregate(xo, ei, eo, ao) if gine else oops.gin_aggregate(xo, ei, ao)
        oo.backward(go.to(dt))
        ref[dt] = (oo, xo.grad, ao.grad, eo.grad if gine else None)
    ix = BatchIndex(ei.to(dev), N)
    assert int(ix.chunk_ptr_dst[-1]) >= 6 + 2        # node 0: >= 6 chunks, node 5: 2 chunks
    xd, ad = x.to(dev).requires_grad_(True), att.to(dev).requires_grad_(True)
    ed = ee.to(dev).requires_grad_(True) if gine else None
    od = masked_sum_aggregate(xd, ix, ad, ed)
    od.backward(go.to(dev))
    r32, r64 = ref[torch.float32], ref[torch.float64]
    close(od, r32[0], ref64=r64[0], what="out")
    close(xd.grad, r32[1], ref64=r64[1], what="dx")
    close(ad.grad, r32[2], ref64=r64[2], what="datt")
    if gine:
        close(ed.grad, r32[3], ref64=r64[3], what="dedge")
    # bitwise reproducible with chunked rows too
    od2 = masked_sum_aggregate(x.to(dev), ix, att.to(dev), ee.to(dev) if gine else None)
    assert torch.equal(od2, od.detach())


def test_gsat_step_on_powerlaw_two_graphs(dev):
    """C5-shaped: few large power-law graphs, GIN + edge attention (symmetrised), H=32."""
    import dp_gsat_amd as G
    from dp_gsat_amd import synth
    from tests.test_gpu_models import _mk_pair, _step
    data = synth.powerlaw_batch(num_nodes=3000, num_edges=30000, num_graphs=2, seed=3, x_dim=8)
    assert torch.bincount(data.edge_index[1]).max() > 256
    H = 32
    cfg = dict(model_name="GIN", n_layers=2, hidden_size=H, dropout_p=0.0)
    # duplicate edges exist in Chung-Lu sampling: symmetrisation pairs them in stable order on both sides
    pair = _mk_pair(G, "GIN", cfg, 8, 0, H, True, dev)
    _step(G, data, *pair, True, H, dev, True)


def test_pna_and_lift_with_hubs(dev):
    from dp_gsat_amd.graph_index import BatchIndex
    from dp_gsat_amd.ops import pna_aggregate
    import dp_gsat_amd as G
    ei, N = hub_graph(2, n=1200, hub_deg=(900, 280), extra=800)
    E = ei.shape[1]
    H = 16
    g = torch.Generator().manual_seed(0)
    x, na, go = torch.randn(N, H, generator=g), torch.rand(N, 1, generator=g), torch.randn(N, 8 * H, generator=g)
    ref = {}
    for dt in (torch.float32, torch.float64):
        xo, no = x.to(dt).clone().requires_grad_(True), na.to(dt).clone().requires_grad_(True)
        ea = oops.lift_node_att_to_edge_att(no, ei)
        oo = oops.pna_aggregate(xo, ei, ea, ["mean", "min", "max", "std"], ["identity"], {"lin": 1.0, "log": 1.0})
        oo.backward(go.to(dt))
        ref[dt] = (oo, xo.grad, no.grad)
    ix = BatchIndex(ei.to(dev), N)
    xd, nd = x.to(dev).requires_grad_(True), na.to(dev).requires_grad_(True)
    ead = G.lift_node_att_to_edge_att(nd, ei.to(dev))
    od = pna_aggregate(xd, ix, ead, None, ["mean", "min", "max", "std"], ["identity"], {"lin": 1.0, "log": 1.0})
    od.backward(go.to(dev))
    r32, r64 = ref[torch.float32], ref[torch.float64]
    close(od, r32[0], ref64=r64[0], what="out")
    close(xd.grad, r32[1], ref64=r64[1], what="dx")
    close(nd.grad, r32[2], ref64=r64[2], what="dnode_att")


def test_extractor_with_sliced_segments(dev):
    """Two graphs with > 4096 rows each: the segmented statistics run in row slices + ordered combine (Z > 1)."""
    import dp_gsat_amd as G
    from dp_gsat_amd import synth
    data = synth.powerlaw_batch(num_nodes=12000, num_edges=40000, num_graphs=2, seed=5, x_dim=8)
    H = 16
    for edge_mode in (True, False):
        M = data.num_edges if edge_mode else data.num_nodes
        assert M // 2 > 4096
        g = torch.Generator().manual_seed(1)
        emb = torch.randn(data.num_nodes, H, generator=g)
        C1 = 4 * H if edge_mode else 2 * H
        masks = [(torch.rand(M, C1, generator=g) > 0.5).float(), (torch.rand(M, H, generator=g) > 0.5).float()]
        u = torch.rand(M, 1, generator=g).clamp_(1e-10, 1 - 1e-10)
        ga = torch.randn(M, 1, generator=g)
        oext = om.ExtractorMLP(H, edge_mode).train()
        ref = {}
        for dt in (torch.float32, torch.float64):
            ext = om.ExtractorMLP(H, edge_mode).to(dt).train()
            ext.load_state_dict({k: v.to(dt) for k, v in oext.state_dict().items()})
            e = emb.to(dt).clone().requires_grad_(True)
            a = oops.concrete_sample(ext(e, data.edge_index, data.batch, masks=[m.to(dt) for m in masks]), u.to(dt), True)
            a.backward(ga.to(dt))
            ref[dt] = dict(a=a, demb=e.grad, **{k: p.grad for k, p in ext.named_parameters()})
        ext = G.ExtractorMLP(H, edge_mode).to(dev).train()
        ext.load_state_dict(oext.state_dict())
        ed = emb.to(dev).requires_grad_(True)
        _, a = ext.attend(ed, data.edge_index.to(dev), data.batch.to(dev), noise=u.to(dev), dropout_masks=[m.to(dev) for m in masks])
        a.backward(ga.to(dev))
        r32, r64 = ref[torch.float32], ref[torch.float64]
        close(a, r32["a"], ref64=r64["a"], what="att")
        close(ed.grad, r32["demb"], ref64=r64["demb"], what="demb")
        for k, p in ext.named_parameters():
            close(p.grad, r32[k], ref64=r64[k], what=k)


@pytest.mark.parametrize("hub_deg", [40, 3000, 4096, 4097, 20000])
def test_counting_csr_build_orders_long_rows(dev, hub_deg):
    """Batches below 2^20 keys build both CSRs by counting + per-row ordering (bookkeeping.hip): rows of <= 32 entries by one thread,
    longer ones by a workgroup (one bitonic chunk of 4096, or several chunks merged by rank).  Bit-exact vs a stable argsort."""
    from dp_gsat_amd.graph_index import BatchIndex
    rng = np.random.default_rng(hub_deg)
    N = hub_deg + 500
    leaves = rng.permutation(np.arange(1, N))[:hub_deg]
    src = np.concatenate([leaves, np.zeros(hub_deg, np.int64), rng.integers(0, N, 3000)])
    dst = np.concatenate([np.zeros(hub_deg, np.int64), leaves, rng.integers(0, N, 3000)])
    order = rng.permutation(src.size)              # hub edges scattered over the whole edge list
    ei = torch.from_numpy(np.stack([src[order], dst[order]]))
    ix = BatchIndex(ei.to(dev), N)
    s, d = ei[0].numpy(), ei[1].numpy()
    for rows, other, rp, col, eid in ((d, s, ix.rowptr_dst, ix.src_by_dst, ix.eid_by_dst), (s, d, ix.rowptr_src, ix.dst_by_src, ix.eid_by_src)):
        perm = np.argsort(rows, kind="stable")
        assert np.array_equal(rp.cpu().numpy(), np.concatenate([[0], np.cumsum(np.bincount(rows, minlength=N))]))
        assert np.array_equal(eid.cpu().numpy(), perm)
        assert np.array_equal(col.cpu().numpy(), other[perm])
    inv = np.empty(src.size, np.int64)
    inv[ix.eid_by_dst.cpu().numpy()] = np.arange(src.size)
    assert np.array_equal(ix.slot_dst_of_srcslot.cpu().numpy(), inv[ix.eid_by_src.cpu().numpy()])


@pytest.mark.parametrize("H", [16, 80, 128, 320])
@pytest.mark.parametrize("with_edge_attr", [False, True])
@pytest.mark.parametrize("aggr,scalers", [
    (["mean", "min", "max", "std"], ["identity"]),
    (["mean", "min", "max", "std", "sum"], ["identity"]),
    (["sum", "var", "max", "min"], ["identity", "amplification", "attenuation"]),
])
def test_pna_hub_rows_chunked(dev, H, with_edge_attr, aggr, scalers):
    """Rows of 900 and 280 in-edges (> GSAT_LONG_ROW_EDGES = 256) take the chunked path of the PNA kernels: per-chunk statistics and
    first min/max slots, folded per row in chunk order, per-edge gradients written chunk-parallel (gsat_pna_fwd_long / _bwd_long).
    Ties are frequent (ReLU zeros in x), so the first-occurrence rule across chunk boundaries is exercised."""
    from dp_gsat_amd.graph_index import BatchIndex
    from dp_gsat_amd.ops import pna_aggregate
    from tests.test_gpu_pna import close_weighted, std_conditioning
    ei, N = hub_graph(3 + H, n=1200, hub_deg=(900, 280), extra=800)
    E = ei.shape[1]
    g = torch.Generator().manual_seed(H)
    x = torch.randn(N, H, generator=g)
    x[::3] = x[::3].relu()
    att = torch.rand(E, 1, generator=g)
    ee = torch.randn(E, H, generator=g) if with_edge_attr else None
    avg = oops.pna_avg_deg(torch.from_numpy(obk.deg_histogram(ei, N)))
    go = torch.randn(N, len(scalers) * len(aggr) * (3 if with_edge_attr else 2) * H, generator=g)
    ref = {}
    for dt in (torch.float32, torch.float64):
        xo, ao = x.to(dt).clone().requires_grad_(True), att.to(dt).clone().requires_grad_(True)
        eo = ee.to(dt).clone().requires_grad_(True) if with_edge_attr else None
        oo = oops.pna_aggregate(xo, ei, ao, aggr, scalers, avg, eo)
        oo.backward(go.to(dt))
        ref[dt] = (oo, xo.grad, ao.grad, eo.grad if with_edge_attr else None)
    ix = BatchIndex(ei.to(dev), N)
    assert ix.long_rows[0] is not None
    xd, ad = x.to(dev).requires_grad_(True), att.to(dev).requires_grad_(True)
    ed = ee.to(dev).requires_grad_(True) if with_edge_attr else None
    od = pna_aggregate(xd, ix, ad, ed, aggr, scalers, avg)
    od.backward(go.to(dev))
    r32, r64 = ref[torch.float32], ref[torch.float64]
    close(od, r32[0], ref64=r64[0], what="out")
    w_dx, w_e = std_conditioning(x, ei, att, N)
    close_weighted(xd.grad, r32[1], r64[1], w_dx, "dx")
    close_weighted(ad.grad, r32[2], r64[2], w_e, "datt")
    if with_edge_attr:
        close(ed.grad, r32[3], 1e-4, ref64=r64[3], what="dedge")
    # run-to-run determinism of the chunked path
    xd2, ad2 = x.to(dev).requires_grad_(True), att.to(dev).requires_grad_(True)
    od2 = pna_aggregate(xd2, ix, ad2, ed.detach() if with_edge_attr else None, aggr, scalers, avg)
    od2.backward(go.to(dev))
    assert torch.equal(od, od2) and torch.equal(xd.grad, xd2.grad) and torch.equal(ad.grad, ad2.grad)


@pytest.mark.parametrize("pattern", ["runs", "random", "giant", "ragged_tail"])
def test_single_csr_counting_build(dev, pattern):
    """gsat_build_csr (the edges-by-graph order of the edge-mode extractor) through the C ABI, counting path: long runs of one key (one
    atomic per run of equal adjacent keys in a wave), unordered keys, a row of 3 chunks + a remainder (bitonic chunks merged by rank),
    a key count that does not fill the last wave.  Bit-exact vs a stable argsort."""
    from dp_gsat_amd._lib import call, ptr, stream
    from dp_gsat_amd.graph_index import call_size
    rng = np.random.default_rng(len(pattern))
    R = 300
    if pattern == "runs":
        rows = np.repeat(np.arange(R), rng.integers(0, 90, R))
    elif pattern == "random":
        rows = rng.integers(0, R, 20_000)
    elif pattern == "giant":
        rows = np.concatenate([rng.integers(0, R, 3000), np.full(3 * 4096 + 777, 17)])
        rows = rows[rng.permutation(rows.size)]
    else:
        rows = np.sort(rng.integers(0, R, 64 * 7 + 5))
    E = rows.size
    other = rng.integers(0, 1000, E)
    rows_d, other_d = torch.from_numpy(rows).to(dev), torch.from_numpy(other).to(dev)
    rowptr = torch.empty(R + 1, dtype=torch.int32, device=dev)
    other_sorted, perm = torch.empty(E, dtype=torch.int32, device=dev), torch.empty(E, dtype=torch.int32, device=dev)
    err = torch.zeros(1, dtype=torch.int32, device=dev)
    wb = max(call_size("gsat_csr_workspace_bytes", E, R), 256)
    ws = torch.empty(wb, dtype=torch.uint8, device=dev)
    call("gsat_build_csr", ptr(rows_d), ptr(other_d), E, R, ptr(rowptr), ptr(other_sorted), ptr(perm), ptr(err), ptr(ws), wb, stream())
    order = np.argsort(rows, kind="stable")
    assert int(err.item()) == 0
    assert np.array_equal(rowptr.cpu().numpy(), np.concatenate([[0], np.cumsum(np.bincount(rows, minlength=R))]))
    assert np.array_equal(perm.cpu().numpy(), order)
    assert np.array_equal(other_sorted.cpu().numpy(), other[order])
