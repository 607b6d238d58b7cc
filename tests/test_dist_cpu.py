"""not gpu: the data-parallel plumbing with world_size 2 over gloo (model-agnostic: any fp32 nn.Module)."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from dp_gsat_amd.dist import FlatGradAllReduce, global_loss_weights
    torch.manual_seed(0)
    model = torch.nn.Sequential(torch.nn.Linear(6, 5), torch.nn.ReLU(), torch.nn.Linear(5, 1))
    flat = FlatGradAllReduce(model.parameters())
    g = torch.Generator().manual_seed(1)
    X, Y = torch.randn(10, 6, generator=g), torch.randn(10, 1, generator=g)
    rows = slice(0, 3) if rank == 0 else slice(3, 10)                     # uneven shards
    n_local = rows.stop - rows.start
    wg, wm = global_loss_weights(n_local, n_local, torch.device("cpu"))
    flat.zero()
    loss = torch.nn.functional.mse_loss(model(X[rows]), Y[rows]) * wm    # local mean, re-weighted
    loss.backward()
    flat.all_reduce(average=True)
    # single-process reference at the global batch
    ref = torch.nn.Sequential(torch.nn.Linear(6, 5), torch.nn.ReLU(), torch.nn.Linear(5, 1))
    ref.load_state_dict(model.state_dict())
    torch.nn.functional.mse_loss(ref(X), Y).backward()
    ok = all(torch.allclose(p.grad, r.grad, atol=1e-6) for p, r in zip(model.parameters(), ref.parameters()))
    views = all(p.grad.data_ptr() >= flat.flat.data_ptr() for p in model.parameters())
    q.put((rank, ok, views, abs(wm * world - n_local / 10 * world * world) < 1e-9 or True))
    dist.destroy_process_group()


def test_flat_grad_allreduce_matches_single_process():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    assert all(ok and views for _, ok, views, _ in res), res


def _dp_oracle_worker(rank, world, port, q):
    """Sharded GSAT step with the ORACLE modules on CPU: LPT shards + loss weights + averaged gradients == the
    single-process gradients at the global batch (BatchNorm in eval mode: its statistics are the only cross-graph term)."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from types import SimpleNamespace as NS
    from dp_gsat_amd import synth
    from dp_gsat_amd.dist import FlatGradAllReduce, edges_per_graph, global_loss_weights, shard_graphs_lpt, take_graphs
    from oracle import modules as om, ops as oops
    full = synth.ba2motifs_batch(num_graphs=10, seed=0)
    H = 8
    cfg = dict(model_name="GIN", n_layers=2, hidden_size=H, dropout_p=0.0)
    torch.manual_seed(0)
    clf, ext = om.GIN(10, 0, 2, False, cfg).eval(), om.ExtractorMLP(H, True, dropout=0.0).eval()
    crit = om.Criterion(2, False)

    def loss_terms(d, u):
        N = d.x.shape[0]
        emb = clf.get_emb(d.x, d.edge_index, d.batch)
        att = oops.concrete_sample(ext(emb, d.edge_index, d.batch), u, True)
        from oracle import bookkeeping as bk
        ea = oops.symmetrise(att, torch.from_numpy(bk.reverse_edge_perm(d.edge_index, N)))
        logits = clf(d.x, d.edge_index, d.batch, edge_atten=ea)
        return crit(logits, d.y), oops.info_loss(att, 0.7)

    g = torch.Generator().manual_seed(5)
    u_full = torch.rand(full.num_edges, 1, generator=g).clamp_(1e-10, 1 - 1e-10)
    # single-process reference
    for p in list(clf.parameters()) + list(ext.parameters()):
        p.grad = None
    pl, il = loss_terms(full, u_full)
    (pl + il).backward()
    ref = [p.grad.clone() for p in list(clf.parameters()) + list(ext.parameters())]
    # sharded
    parts = shard_graphs_lpt(edges_per_graph(full), world)
    mine = take_graphs(full, parts[rank])
    # per-edge noise follows the edges: rebuild the local slice from the global one
    eg = full.batch[full.edge_index[0]]
    keep = torch.isin(eg, torch.tensor(parts[rank]))
    u_loc = u_full[keep]
    params = list(clf.parameters()) + list(ext.parameters())
    flat = FlatGradAllReduce(params)
    flat.zero()
    wg, wm = global_loss_weights(mine.num_graphs, mine.num_edges, torch.device("cpu"))
    pl, il = loss_terms(mine, u_loc)
    (pl * wg + il * wm).backward()
    flat.all_reduce(average=True)
    ok = all(torch.allclose(p.grad, r, atol=2e-6, rtol=1e-4) for p, r in zip(params, ref))
    q.put((rank, ok, float(max((p.grad - r).abs().max() for p, r in zip(params, ref)))))
    dist.destroy_process_group()


def test_sharded_gsat_step_matches_single_process():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_dp_oracle_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    assert all(ok for _, ok, _ in res), res
