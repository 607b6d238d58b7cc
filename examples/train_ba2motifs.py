#!/usr/bin/env python3
"""End-to-end GSAT training on a synthetic BA-2motifs-style dataset with the MI355X hot path.

Every graph is a 20-node Barabasi-Albert tree plus a 5-node motif (house -> label 1, cycle -> label 0) attached by one edge
(the construction of src/datasets/ba_2motifs.py); the motif's edges are the ground-truth explanation.  The script follows the
reference's training loop (example/trainer.py:28-60: forward_pass, zero_grad, backward, Adam; accuracy and the ROC-AUC of the
learned edge attention against the motif edges) on top of:
  * dp_gsat_amd.PackedDataset -- the dataset lives in HBM, every batch is collated on the device from graph ids;
  * dp_gsat_amd.get_model / ExtractorMLP / GSAT -- the reference's module protocol over the HIP kernels;
  * data parallelism (optional): `python -m torch.distributed.run --nproc-per-node N examples/train_ba2motifs.py` shards every
    global batch over the ranks by edge count (LPT), re-weights the local losses and all-reduces one flat gradient buffer.

  python examples/train_ba2motifs.py --graphs 1000 --epochs 20
"""
import argparse
import os
import sys
from types import SimpleNamespace

import numpy as np
import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def make_graphs(num_graphs, seed):
    rng = np.random.RandomState(seed)
    graphs = []
    for _ in range(num_graphs):
        edges, deg = [(0, 1)], np.zeros(20)
        deg[0] = deg[1] = 1
        for v in range(2, 20):
            u = int(rng.choice(v, p=deg[:v] / deg[:v].sum()))
            edges.append((u, v)); deg[u] += 1; deg[v] += 1
        n_base = len(edges)
        house = rng.rand() < 0.5
        m = 20
        if house:
            edges += [(m, m + 1), (m + 1, m + 2), (m + 2, m + 3), (m + 3, m), (m + 4, m), (m + 4, m + 1)]
        else:
            edges += [(m + i, m + (i + 1) % 5) for i in range(5)]
        n_motif = len(edges) - n_base
        edges.append((int(rng.randint(0, 20)), m))
        und = np.asarray(edges, dtype=np.int64)
        ei = np.concatenate([und, und[:, ::-1]], axis=0).T                      # both directions
        lab = np.zeros(len(edges), dtype=np.float32)
        lab[n_base:n_base + n_motif] = 1.0
        graphs.append(SimpleNamespace(x=torch.full((25, 10), 0.1), edge_index=torch.from_numpy(np.ascontiguousarray(ei)),
                                      y=torch.tensor([[float(house)]]), edge_label=torch.from_numpy(np.concatenate([lab, lab])),
                                      edge_attr=None))
    return graphs


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--graphs", type=int, default=1000)
    ap.add_argument("--epochs", type=int, default=20)
    ap.add_argument("--batch-size", type=int, default=128, help="GLOBAL batch size")
    ap.add_argument("--hidden", type=int, default=64)
    ap.add_argument("--backbone", default="GIN", choices=["GIN", "PNA"])
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--backend", default="nccl", help="nccl (= RCCL, one GPU per rank) or gloo (to rehearse N ranks on one GPU)")
    args = ap.parse_args()

    world, rank = int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0"))
    dev = torch.device("cuda", int(os.environ.get("LOCAL_RANK", "0")) % max(torch.cuda.device_count(), 1))
    torch.cuda.set_device(dev)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(args.backend, rank=rank, world_size=world)
    import dp_gsat_amd as G
    from dp_gsat_amd.dist import FlatGradAllReduce, global_loss_weights, shard_graphs_lpt
    from sklearn.metrics import roc_auc_score

    torch.manual_seed(args.seed)                                  # same initial weights on every rank
    graphs = make_graphs(args.graphs, args.seed)
    n_train = int(0.8 * len(graphs))
    ds = G.PackedDataset.from_data_list(graphs, dev)
    edge_counts = ds.edge_counts.cpu().numpy()
    deg = torch.bincount(torch.cat([torch.bincount(g.edge_index[1], minlength=25) for g in graphs]), minlength=10)
    cfg = dict(model_name=args.backbone, n_layers=2, hidden_size=args.hidden, dropout_p=0.3, use_edge_attr=False,
               aggregators=["mean", "min", "max", "std", "sum"], scalers=False, deg=deg)
    clf = G.get_model(10, 0, 2, False, cfg, dev)
    ext = G.ExtractorMLP(args.hidden, True).to(dev)
    params = list(clf.parameters()) + list(ext.parameters())
    opt = torch.optim.Adam(params, lr=1e-3, weight_decay=3e-6)
    gsat = G.GSAT(clf, ext, G.Criterion(2, False), opt, learn_edge_att=True)
    flat = FlatGradAllReduce(params) if world > 1 else None
    torch.manual_seed(args.seed + 1000 * rank)                    # independent noise / dropout per rank from here on

    def evaluate(ids):
        gsat.eval()
        with torch.no_grad():
            b = ds.collate(torch.as_tensor(ids, device=dev))
            att, _, _, logits = gsat.forward_pass(b, 0, False)
        gsat.train()
        acc = float(((logits > 0).float() == b.y).float().mean())
        auc = roc_auc_score(b.edge_label.cpu().numpy(), att.view(-1).cpu().numpy())
        return acc, auc

    gen = np.random.RandomState(args.seed)                        # identical permutations on every rank
    for epoch in range(args.epochs):
        perm = gen.permutation(n_train)
        tot, nb = 0.0, 0
        for s in range(0, n_train, args.batch_size):
            ids = perm[s:s + args.batch_size]
            if world > 1:                                         # whole graphs to ranks, balanced by edge count
                ids = ids[shard_graphs_lpt(edge_counts[ids], world)[rank]]
            b = ds.collate(torch.as_tensor(ids, device=dev))
            w = global_loss_weights(b.num_graphs, b.edge_index.shape[1], dev) if world > 1 else None
            _, loss, ld, _ = gsat.forward_pass(b, epoch, True, loss_weights=w)
            if flat is not None:
                flat.zero_grad()
            else:
                opt.zero_grad(set_to_none=True)
            loss.backward()
            if flat is not None:
                flat.all_reduce(average=True, async_op=True)
                flat.wait()
            opt.step()
            tot, nb = tot + ld["loss"], nb + 1
        if rank == 0 and (epoch % 5 == 4 or epoch == args.epochs - 1):
            acc, auc = evaluate(np.arange(n_train, len(graphs)))
            print(f"epoch {epoch + 1:3d}  train loss {tot / nb:.4f}  test acc {acc:.3f}  attention ROC-AUC vs motif edges {auc:.3f}", flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
