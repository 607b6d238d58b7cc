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


def test_bad_ids_are_reported_without_an_explicit_check(dev):
    """out-of-range ids and unsorted batch vectors must surface as ValueError on the normal call path (no explicit
    .check()), also for E <= 256, and must never reach a kernel unclamped (no fault, in eager and in sync-free mode)."""
    import dp_gsat_amd as G
    from dp_gsat_amd.ops import masked_sum_aggregate
    G.clear_cache()
    ei = torch.tensor([[0, 3, 1], [1, 0, 2]], device=dev)                   # id 3 == N: out of range, E <= 256
    x = torch.randn(3, 16, device=dev)
    with pytest.raises(ValueError, match="outside"):
        masked_sum_aggregate(x, G.get_index(ei, 3), torch.rand(3, 1, device=dev))
    # the index built from the bad ids is memory-safe: every stored id is inside [0, N)
    ix = G.BatchIndex(ei, 3)
    for t in (ix.src32, ix.dst32, ix.src_by_dst, ix.dst_by_src):
        assert int(t.min()) >= 0 and int(t.max()) < 3
    # a permuted batch vector through GSAT.forward_pass: ValueError, not silently wrong norms
    from dp_gsat_amd.synth import Batch
    from tests.graphs import random_batch
    ei3, batch3, n3 = random_batch(5, 3, 4, 9)
    bad_batch = batch3.flip(0).contiguous()                                    # same multiset of ids, wrong order
    bad = Batch(x=torch.randn(n3, 8), edge_index=ei3, batch=bad_batch, edge_attr=None, y=torch.ones(3, 1),
                num_graphs=3).to(dev)
    cfg = dict(model_name="GIN", n_layers=2, hidden_size=16, dropout_p=0.0, use_edge_attr=False)
    clf = G.get_model(8, 0, 2, False, cfg, dev)
    gsat = G.GSAT(clf, G.ExtractorMLP(16, True).to(dev), G.Criterion(2, False), None, learn_edge_att=True).train()
    G.clear_cache()
    with pytest.raises(ValueError, match="non-decreasing"):
        gsat.forward_pass(bad, 0, True)
    G.clear_cache()
    G.set_sync_free(True)
    try:
        with pytest.raises(ValueError, match="non-decreasing"):          # validated once, outside capture
            gsat.forward_pass(bad, 0, True)
        G.clear_cache()
        with pytest.raises(ValueError, match="outside"):
            masked_sum_aggregate(x, G.get_index(ei.clone(), 3), torch.rand(3, 1, device=dev))
    finally:
        G.set_sync_free(False)
        G.clear_cache()


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
        close(got[k], r32[k], 1e-4, ref64=r64[k], what=k)


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


def test_whole_mutag_file_reference_recorded_answers(dev):
    """The HIP bookkeeping against numbers the REFERENCE holds (not the oracle): whole Mutagenicity file, in-degree histogram
    [2401, 64058, 7570, 42140, 15319] and `edge 2k+1 reverses edge 2k` (data/mutag_dual/raw), and the dual-edge count
    recorded by the reference's author next to the pair loops: `# len dual_edges: 451808` (src/datasets/mutag_dual.py:385)."""
    import os
    import dp_gsat_amd as G
    from dp_gsat_amd.synth import mutag_full_topology
    ei, batch, kept = mutag_full_topology(os.path.join(os.path.dirname(__file__), "golden", "mutag_full.npz"))
    N, E = int(batch.shape[0]), int(ei.shape[1])
    assert (N, E) == (131488, 266894)
    ix = G.BatchIndex(ei.to(dev), N)
    indeg = (ix.rowptr_dst[1:] - ix.rowptr_dst[:-1]).long()
    assert torch.bincount(indeg).tolist() == [2401, 64058, 7570, 42140, 15319]
    assert int(ix.rowptr_dst[-1]) == E and int(ix.rowptr_src[-1]) == E
    assert ix.is_undirected
    rev = ix.rev.cpu().long()
    assert torch.equal(rev[0::2], torch.arange(1, E, 2)) and torch.equal(rev[1::2], torch.arange(0, E, 2))
    dei, dbatch = G.line_graph(ei.to(dev), N, batch.to(dev))
    assert tuple(dei.shape) == (2, 451808)
    # same dual edges as the reference's pair loops (the file is not source-sorted, so the reference's groups appear in order of
    # first appearance and the device's in ascending source id: compare as sets; the source-sorted copy is compared bit-exactly)
    key = lambda a: np.sort(a[0].astype(np.int64) * E + a[1])
    assert np.array_equal(key(dei.cpu().numpy()), key(obk.line_graph_by_source(ei)))
    order = torch.sort(ei[0], stable=True)[1]
    eis = ei[:, order].contiguous()
    assert np.array_equal(G.line_graph(eis.to(dev), N)[0].cpu().numpy(), obk.line_graph_by_source(eis))
    seg = ix.graphs(batch.to(dev), 4337)
    assert torch.equal(seg.node_ptr.cpu().long(), torch.from_numpy(obk.graph_ptr(batch, 4337)))


def test_undirected_line_graph_bit_exact(dev):
    """The ba_2motifs dual rule (src/datasets/ba_2motifs_dual.py:35-62) on the device vs the oracle's loop-by-loop restatement."""
    import dp_gsat_amd as G
    from dp_gsat_amd.synth import ba2motifs_batch
    from tests.graphs import random_batch, shuffle_edges
    d = ba2motifs_batch(num_graphs=64, seed=11)
    ei = shuffle_edges(d.edge_index, 3)                                 # edge order must not matter
    want = obk.line_graph_undirected(ei, d.batch, d.x, motif_start=20)
    got = G.line_graph_undirected(ei.to(dev), d.num_nodes, d.batch.to(dev), d.x.to(dev), motif_start=20)
    assert np.array_equal(got.edge_index.cpu().numpy(), want[0])
    assert np.array_equal(got.und_index.cpu().numpy(), want[1])
    assert np.array_equal(got.batch.cpu().numpy(), want[2])
    assert np.array_equal(got.x.cpu().numpy(), want[3])
    assert np.array_equal(got.node_label.cpu().numpy(), want[4])
    # und_of_edge: both directions of an edge map to the same dual node, whose endpoints are the edge's endpoints
    u = got.und_of_edge.cpu()
    und = got.und_index.cpu()
    lo, hi = torch.minimum(ei[0], ei[1]), torch.maximum(ei[0], ei[1])
    assert torch.equal(und[0][u], lo) and torch.equal(und[1][u], hi)
    # ragged molecule-like graphs with isolated nodes and duplicated edges (the dense adjacency of the reference dedupes them)
    ei2, b2, n2 = random_batch(21, 9, 2, 17)
    ei2 = torch.cat([ei2, ei2[:, :4]], dim=1)
    w2 = obk.line_graph_undirected(ei2, b2, None)
    g2 = G.line_graph_undirected(ei2.to(dev), n2, b2.to(dev))
    assert np.array_equal(g2.edge_index.cpu().numpy(), w2[0]) and np.array_equal(g2.und_index.cpu().numpy(), w2[1])
    # a self loop is no dual node (`node1 != node2`, :46) and does not disturb the others (the reference's loops would index
    # dual_dense[-1] through it; BA-2motifs has none, so that quirk is not reproduced)
    ei3 = torch.cat([ei2, torch.tensor([[1], [1]])], dim=1)
    g3 = G.line_graph_undirected(ei3.to(dev), n2, b2.to(dev))
    assert int(g3.und_of_edge[-1]) == -1 and torch.equal(g3.edge_index, g2.edge_index)
    with pytest.raises(ValueError, match="symmetric"):
        G.line_graph_undirected(torch.tensor([[0, 1], [1, 2]], device=dev), 3)
    empty = G.line_graph_undirected(torch.zeros(2, 0, dtype=torch.int64, device=dev), 3)
    assert empty.edge_index.shape == (2, 0) and empty.num_dual_nodes == 0


@pytest.mark.parametrize("seed", [0, 1, 2])
def test_reverse_permutation_from_csr_with_duplicates_and_self_loops(dev, seed):
    """Batches below 2^20 keys pair (s,d) copies with (d,s) copies through the CSRs (gsat_reverse_edge_perm_csr): k-th copy with k-th copy
    by edge id -- the pairing of the oracle's stable sort -- including multi-edges, self loops and a hub; one missing partner clears the flag."""
    from dp_gsat_amd.graph_index import BatchIndex
    rng = np.random.default_rng(seed)
    N = 60
    a, b = rng.integers(0, N, 400), rng.integers(0, N, 400)
    a, b = np.concatenate([a, a[:80], np.zeros(150, np.int64)]), np.concatenate([b, b[:80], rng.integers(1, N, 150)])     # repeats + hub 0
    loops = rng.integers(0, N, 20)
    src = np.concatenate([a, b, loops, loops[:5]])
    dst = np.concatenate([b, a, loops, loops[:5]])
    order = rng.permutation(src.size)
    ei = torch.from_numpy(np.stack([src[order], dst[order]]))
    assert obk.is_undirected(ei, N)
    ix = BatchIndex(ei.to(dev), N)
    rev = ix.rev
    assert rev is not None and np.array_equal(rev.cpu().numpy().astype(np.int64), obk.reverse_edge_perm(ei, N))
    cut = ei[:, 1:]                                        # one edge dropped: its partner has no partner any more (unless it was a loop)
    if cut[0, :].ne(cut[1, :]).all() or not obk.is_undirected(cut, N):
        assert BatchIndex(cut.contiguous().to(dev), N).rev is None


def test_pna_path_reports_bad_ids_late_and_learns_hubs_without_a_sync(dev):
    """The PNA ops never read the status words back inside a step (a host sync costs the 1 ms C3 step ~10 %): every batch queues an
    asynchronous copy instead, later batches harvest it.  Out-of-range ids therefore surface as ValueError one batch late, and the
    chunked hub path switches on once a batch with a long row has been seen."""
    import dp_gsat_amd as G
    from dp_gsat_amd import graph_index as gi
    from dp_gsat_amd.ops import pna_aggregate
    G.clear_cache()
    gi._HUBS_SEEN[0] = False
    aggr, sc, avg = ["mean", "min", "max", "std"], ["identity"], {"lin": 1.0, "log": 1.0}
    x = torch.randn(3, 16, device=dev)
    bad = G.BatchIndex(torch.tensor([[0, 3, 1], [1, 0, 2]], device=dev), 3)          # id 3 == N
    out = pna_aggregate(x, bad, None, None, aggr, sc, avg)                              # memory-safe (ids clamped), no error yet
    assert torch.isfinite(out).all()
    torch.cuda.synchronize()
    good = G.BatchIndex(torch.tensor([[0, 1], [1, 0]], device=dev), 3)
    with pytest.raises(ValueError, match="earlier batch"):
        for _ in range(3):                                                             # first use queues its own copy, the next ones harvest
            pna_aggregate(x, good, None, None, aggr, sc, avg)
            torch.cuda.synchronize()
    # strict mode (debugging): the same bad batch raises at once, before the first PNA launch, as the reference's gathers would
    G.clear_cache()
    G.set_strict(True)
    try:
        bad2 = G.BatchIndex(torch.tensor([[0, 3, 1], [1, 0, 2]], device=dev), 3)
        with pytest.raises(ValueError, match="outside"):
            pna_aggregate(x, bad2, None, None, aggr, sc, avg)
    finally:
        G.set_strict(False)
        G.clear_cache()
    # hubs: unknown -> plain kernels (correct), after the status has landed -> chunk lists
    hub = torch.stack([torch.arange(1, 400, device=dev), torch.zeros(399, dtype=torch.int64, device=dev)])
    ix = G.BatchIndex(hub.contiguous(), 400)
    assert ix.long_rows_nowait == (None, None)
    torch.cuda.synchronize()
    first = ix.long_rows_nowait
    assert first[0] is not None and first[1] is None and gi._HUBS_SEEN[0]
    ix2 = G.BatchIndex(hub.clone(), 400)
    assert ix2.long_rows_nowait[0] is not None                                          # hubs seen before: chunk lists right away
    xa = torch.randn(400, 16, device=dev)
    assert torch.equal(pna_aggregate(xa, ix2, None, None, aggr, sc, avg), pna_aggregate(xa, ix, None, None, aggr, sc, avg))
    gi._HUBS_SEEN[0] = False
    G.clear_cache()
