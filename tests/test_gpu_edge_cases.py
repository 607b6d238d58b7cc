"""-m gpu: empty / ragged inputs, error behaviour, and BASELINE-size checks."""
import numpy as np
import pytest
import torch

from oracle import bookkeeping as obk
from oracle import modules as om
from oracle import ops as oops
from tests.util import close

pytestmark = pytest.mark.gpu


def test_empty_edge_set(dev):
    import dp_gsat_amd as G
    from dp_gsat_amd.ops import masked_sum_aggregate, pna_aggregate
    N, H = 5, 16
    ei = torch.zeros(2, 0, dtype=torch.int64, device=dev)
    batch = torch.tensor([0, 0, 1, 1, 1], device=dev)
    x = torch.randn(N, H, device=dev, requires_grad=True)
    ix = G.BatchIndex(ei, N)
    assert ix.is_undirected and ix.rowptr_dst.tolist() == [0] * (N + 1)
    out = masked_sum_aggregate(x, ix, torch.zeros(0, 1, device=dev))
    assert torch.equal(out, x)                                   # (1+eps) x_i, no messages
    out.sum().backward()
    assert torch.equal(x.grad, torch.ones_like(x))
    p = pna_aggregate(x.detach(), ix, None, None, ["mean", "min", "max", "std"], ["identity"], {"lin": 1.0, "log": 1.0})
    assert torch.all(p[:, : 6 * H] == 0) and torch.allclose(p[:, 6 * H:], torch.full((N, 2 * H), 1e-5 ** 0.5, device=dev))
    ext = G.ExtractorMLP(H, True).to(dev)
    z, a = ext.attend(x.detach(), ei, batch)
    assert z.shape == (0, 1) and a.shape == (0, 1)
    ext_n = G.ExtractorMLP(H, False).to(dev).eval()
    zn = ext_n(x.detach(), ei, batch)
    oext = om.ExtractorMLP(H, False).eval(); oext.load_state_dict(ext_n.state_dict())
    close(zn, oext(x.detach().cpu(), ei.cpu(), batch.cpu()))
    assert G.lift_node_att_to_edge_att(torch.rand(N, 1, device=dev), ei).shape == (0, 1)
    assert G.symmetrise_edge_att(torch.zeros(0, 1, device=dev), ei, N).shape == (0, 1)


def test_graphs_without_edges_and_isolated_nodes(dev):
    """Batch of 4 graphs where graph 1 is a single isolated node and graph 3 has two nodes and no edge: empty segments
    in the edge-mode InstanceNorm, empty CSR rows everywhere."""
    import dp_gsat_amd as G
    ei = torch.tensor([[0, 1, 1, 2, 4, 5], [1, 0, 2, 1, 5, 4]])
    batch = torch.tensor([0, 0, 0, 1, 2, 2, 3, 3])
    N, H = 8, 16
    g = torch.Generator().manual_seed(0)
    x = torch.randn(N, 6, generator=g)
    data = type("D", (), {})()
    cfg = dict(model_name="GIN", n_layers=2, hidden_size=H, dropout_p=0.0)
    oclf, oext = om.GIN(6, 0, 2, False, cfg), om.ExtractorMLP(H, True)
    clf = G.get_model(6, 0, 2, False, cfg, dev); clf.load_state_dict(oclf.state_dict())
    ext = G.ExtractorMLP(H, True).to(dev); ext.load_state_dict(oext.state_dict())
    oclf.eval(); oext.eval(); clf.eval(); ext.eval()
    emb_o = oclf.get_emb(x, ei, batch)
    z_o = oext(emb_o, ei, batch)
    emb = clf.get_emb(x.to(dev), ei.to(dev), batch.to(dev))
    z = ext(emb, ei.to(dev), batch.to(dev))
    close(emb, emb_o); close(z, z_o)
    logits_o = oclf(x, ei, batch, edge_atten=z_o.sigmoid())
    logits = clf(x.to(dev), ei.to(dev), batch.to(dev), edge_atten=z.sigmoid())
    assert logits.shape == (4, 1)
    close(logits, logits_o)


def test_error_behaviour(dev):
    import dp_gsat_amd as G
    from dp_gsat_amd._lib import GsatHipError
    from dp_gsat_amd.ops import masked_sum_aggregate
    ei = torch.tensor([[0, 1], [1, 0]], device=dev)
    ix = G.BatchIndex(ei, 2)
    with pytest.raises(GsatHipError, match="multiple of 4"):
        masked_sum_aggregate(torch.randn(2, 30, device=dev), ix)             # unsupported width fails loudly
    with pytest.raises(ValueError, match="rows"):
        masked_sum_aggregate(torch.randn(3, 16, device=dev), ix)
    with pytest.raises(ValueError, match="entries"):
        masked_sum_aggregate(torch.randn(2, 16, device=dev), ix, torch.rand(5, 1, device=dev))
    bad = G.BatchIndex(torch.tensor([[0, 7], [1, 0]], device=dev), 2)
    with pytest.raises(ValueError, match="outside"):
        bad.check()
    with pytest.raises(TypeError):
        masked_sum_aggregate(torch.randn(2, 16, device=dev, dtype=torch.float64), ix)
    seg = ix.graphs(torch.tensor([1, 0], device=dev), 2)
    with pytest.raises(ValueError, match="non-decreasing"):
        seg.check()
    gine = G.GINEConv(torch.nn.Linear(16, 16), edge_dim=None, in_channels=16).to(dev)
    with pytest.raises(ValueError, match="dimensionalities"):                # src/models/conv_layers.py:54-57
        gine(torch.randn(2, 16, device=dev), ei, edge_attr=torch.randn(2, 3, device=dev))


@pytest.mark.parametrize("workload", ["c1", "c2", "c3", "c4"])
def test_scope_a_at_baseline_size(dev, workload):
    """The bench's scope-A step at BASELINE.json's full batch sizes vs the oracle on the same seeded inputs
    (C1..C4 are small enough for the CPU oracle: a few seconds each)."""
    import bench
    import dp_gsat_amd as G
    wl = dict(bench.WORKLOADS[workload], key=workload)
    data, x_dim, e_dim = bench.make_batch(workload, wl["graphs"], 0)
    N, E, H, L = data.num_nodes, data.num_edges, wl["H"], wl["L"]
    edge = wl["edge_att"]
    g = torch.Generator().manual_seed(7)
    emb = torch.randn(N, H, generator=g)
    xs = [torch.randn(N, H, generator=g) for _ in range(L)]
    gine = data.edge_attr is not None
    ees = [torch.randn(E, H, generator=g) for _ in range(L)] if gine else None
    width = 8 * H if wl["backbone"] == "PNA" else H
    gouts = [torch.randn(N, width, generator=g) for _ in range(L)]
    M = E if edge else N
    C1 = 4 * H if edge else 2 * H
    u = torch.rand(M, 1, generator=g).clamp_(1e-10, 1 - 1e-10)
    masks = [(torch.rand(M, C1, generator=g) > 0.5).float(), (torch.rand(M, H, generator=g) > 0.5).float()]
    oext = om.ExtractorMLP(H, edge).train()

    def run_oracle(dt):
        ext = om.ExtractorMLP(H, edge).to(dt).train()
        ext.load_state_dict({k: v.to(dt) for k, v in oext.state_dict().items()})
        e = emb.to(dt).clone().requires_grad_(True)
        xl = [t.to(dt).clone().requires_grad_(True) for t in xs]
        att = oops.concrete_sample(ext(e, data.edge_index, data.batch, masks=[m.to(dt) for m in masks]), u.to(dt), True)
        if edge:
            rev = torch.from_numpy(obk.reverse_edge_perm(data.edge_index, N)) if obk.is_undirected(data.edge_index, N) else None
            ea = oops.symmetrise(att, rev)
        else:
            ea = oops.lift_node_att_to_edge_att(att, data.edge_index)
        outs = []
        for l in range(L):
            if wl["backbone"] == "PNA":
                outs.append(oops.pna_aggregate(xl[l], data.edge_index, ea, bench.PNA_AGGR, ["identity"], {"lin": 1.0, "log": 1.0}))
            elif gine:
                outs.append(oops.gine_aggregate(xl[l], data.edge_index, ees[l].to(dt), ea))
            else:
                outs.append(oops.gin_aggregate(xl[l], data.edge_index, ea))
        torch.autograd.backward(outs, [t.to(dt) for t in gouts])
        return dict(att=att, ea=ea, out0=outs[0], demb=e.grad, dx0=xl[0].grad, dW1=ext.feature_extractor[0].weight.grad,
                    dW3=ext.feature_extractor[8].weight.grad)

    r32, r64 = run_oracle(torch.float32), run_oracle(torch.float64)
    d = data.to(dev)
    ext = G.ExtractorMLP(H, edge).to(dev).train()
    ext.load_state_dict(oext.state_dict())
    e = emb.to(dev).requires_grad_(True)
    xl = [t.to(dev).requires_grad_(True) for t in xs]
    index = G.get_index(d.edge_index, N)
    _, att = ext.attend(e, d.edge_index, d.batch, noise=u.to(dev), dropout_masks=[m.to(dev) for m in masks])
    ea = G.symmetrise_edge_att(att, d.edge_index, N) if edge else G.lift_node_att_to_edge_att(att, d.edge_index)
    outs = []
    for l in range(L):
        if wl["backbone"] == "PNA":
            outs.append(G.ops.pna_aggregate(xl[l], index, ea, None, bench.PNA_AGGR, ["identity"], {"lin": 1.0, "log": 1.0}))
        else:
            outs.append(G.ops.masked_sum_aggregate(xl[l], index, ea, ees[l].to(dev) if gine else None))
    torch.autograd.backward(outs, [t.to(dev) for t in gouts])
    got = dict(att=att, ea=ea, out0=outs[0], demb=e.grad, dx0=xl[0].grad, dW1=ext.mlp.linears()[0].weight.grad, dW3=ext.mlp.linears()[2].weight.grad)
    # integer bookkeeping at full size: bit-exact
    rp, perm = obk.csr_by(data.edge_index[1], N)
    assert np.array_equal(index.rowptr_dst.cpu().numpy().astype(np.int64), rp)
    assert np.array_equal(index.eid_by_dst.cpu().numpy().astype(np.int64), perm)
    if edge:
        und = obk.is_undirected(data.edge_index, N)
        assert index.is_undirected == und
        if und:
            assert np.array_equal(index.rev.cpu().numpy().astype(np.int64), obk.reverse_edge_perm(data.edge_index, N))
            assert torch.equal(got["ea"][:, 0], got["ea"][index.rev.long(), 0])       # symmetrised attention is symmetric
    for k in ("att", "ea", "out0"):
        close(got[k], r32[k], ref64=r64[k], what=k)
    for k in ("demb", "dx0", "dW1", "dW3"):
        close(got[k], r32[k], 2e-4, ref64=r64[k], what=k)


def test_device_collation_matches_host_collation(dev):
    """PackedDataset.collate == concatenating the graphs on the host with node-offset edge_index (Batch.from_data_list)."""
    import dp_gsat_amd as G
    from types import SimpleNamespace as NS
    rng = np.random.RandomState(0)
    graphs = []
    for i in range(40):
        n = int(rng.randint(1, 12)); e = int(rng.randint(0, 20))
        graphs.append(NS(x=torch.randn(n, 5), edge_index=torch.from_numpy(rng.randint(0, n, size=(2, e))).long(),
                         y=torch.tensor([[float(i % 2)]]), edge_attr=torch.randn(e, 3), edge_label=torch.rand(e)))
    ds = G.PackedDataset.from_data_list(graphs, dev)
    for ids in ([3, 4, 5, 6], [39, 0, 17, 17, 2], list(range(40))):
        b = ds.collate(torch.tensor(ids, device=dev))
        off, xs, eis, bs, eas = 0, [], [], [], []
        for k, g in enumerate(ids):
            gr = graphs[g]
            xs.append(gr.x); eis.append(gr.edge_index + off); bs += [k] * gr.x.shape[0]; eas.append(gr.edge_attr); off += gr.x.shape[0]
        assert torch.equal(b.x.cpu(), torch.cat(xs)) and torch.equal(b.edge_index.cpu(), torch.cat(eis, dim=1))
        assert torch.equal(b.batch.cpu(), torch.tensor(bs)) and torch.equal(b.edge_attr.cpu(), torch.cat(eas))
        assert torch.equal(b.y.cpu(), torch.cat([graphs[g].y for g in ids])) and b.num_graphs == len(ids)


def test_line_graph_bit_exact(dev):
    """Device line-graph construction == the reference's pair loops, on the MUTAG fixture and the line-graph size recorded in the reference (comment at mutag_dual.py:385
    counts 451 808 dual edges for the whole dataset; here the first 128 kept graphs)."""
    import os
    import dp_gsat_amd as G
    z = np.load(os.path.join(os.path.dirname(__file__), "golden", "mutag128.npz"))
    ei = torch.from_numpy(z["edge_index"].astype(np.int64))
    batch = torch.from_numpy(z["batch"].astype(np.int64))
    N = batch.shape[0]
    order = torch.sort(ei[0], stable=True)[1]                      # source-sorted copy: group order == ascending source id
    ei = ei[:, order].contiguous()
    want = obk.line_graph_by_source(ei)
    dei, dbatch = G.line_graph(ei.to(dev), N, batch.to(dev))
    assert np.array_equal(dei.cpu().numpy(), want)                 # bit-exact, including the order of the dual edges
    assert torch.equal(dbatch.cpu(), batch[ei[0]])
    deg = torch.bincount(ei[0], minlength=N)
    assert dei.shape[1] == int((deg * (deg - 1)).sum())
    # unsorted edge list: same SET of dual edges (group order then follows ascending source id instead of first appearance)
    perm = torch.randperm(ei.shape[1], generator=torch.Generator().manual_seed(0))
    ei2 = ei[:, perm].contiguous()
    w2 = obk.line_graph_by_source(ei2)
    d2 = G.line_graph(ei2.to(dev), N)[0].cpu().numpy()
    key = lambda a: np.sort(a[0] * ei.shape[1] + a[1])
    assert np.array_equal(key(d2), key(w2))
