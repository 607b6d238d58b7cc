"""Generates the committed fixtures under tests/golden/ (run in the BUILD container only).

  python tests/golden/make_golden.py

1. mutag128.npz  -- the real MUTAG topology: the first 128 graphs the reference keeps
   (`mask_log.txt` == 1, written by src/datasets/mutag.py:43-72), re-based to a collated batch, plus the
   whole-file facts SURVEY.md 8c lists (graph / node / edge counts, in-degree histogram, "edge 2k+1 is the
   reverse of edge 2k").  These are DATA files of the reference (data/mutag_dual/raw/*.txt); no reference
   source is read or copied.
2. oracle_*.npz -- inputs and outputs of the CPU oracle (oracle/) on small seeded cases.  They pin the
   oracle against silent drift (tests/test_oracle_golden.py) and give the GPU parity tests fixed targets
   (tests/test_gpu_golden.py).  They are NOT outputs of the reference: PyG / torch-scatter / torch-sparse
   are not installed, so the reference cannot run here (parity unpinned at that boundary, see oracle/__init__.py).
"""
import json
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
RAW = "/root/reference/data/mutag_dual/raw"


def make_mutag():
    A = np.loadtxt(f"{RAW}/Mutagenicity_A.txt", delimiter=",", dtype=np.int64) - 1            # [E,2] (row, col) 0-based
    gi = np.loadtxt(f"{RAW}/Mutagenicity_graph_indicator.txt", dtype=np.int64) - 1            # [N]
    nl = np.loadtxt(f"{RAW}/Mutagenicity_node_labels.txt", dtype=np.int64)                    # [N]
    gl = np.loadtxt(f"{RAW}/Mutagenicity_graph_labels.txt", dtype=np.int64)                   # [G]
    mask = np.loadtxt(f"{RAW}/mask_log.txt", dtype=np.int64)                                  # [G]
    N, E, G = len(gi), len(A), len(gl)
    indeg = np.bincount(A[:, 1], minlength=N)
    facts = dict(num_graphs=int(G), num_nodes=int(N), num_directed_edges=int(E), kept_graphs=int(mask.sum()),
                 in_degree_histogram=np.bincount(indeg).tolist(),
                 pairs_are_reverses=bool(np.array_equal(A[0::2, 0], A[1::2, 1]) and np.array_equal(A[0::2, 1], A[1::2, 0])),
                 duplicate_edges=int(E - len(np.unique(A[:, 0] * N + A[:, 1]))),
                 graph_indicator_sorted=bool(np.all(np.diff(gi) >= 0)))
    kept = np.flatnonzero(mask == 1)[:128]
    node_ptr = np.concatenate([[0], np.cumsum(np.bincount(gi, minlength=G))])
    new_id = -np.ones(N, dtype=np.int64)
    batch, labels, off = [], [], 0
    for b, g in enumerate(kept):
        n = node_ptr[g + 1] - node_ptr[g]
        new_id[node_ptr[g]:node_ptr[g + 1]] = np.arange(off, off + n)
        batch += [b] * n
        labels += nl[node_ptr[g]:node_ptr[g + 1]].tolist()
        off += n
    sel = (new_id[A[:, 0]] >= 0)
    assert np.array_equal(sel, new_id[A[:, 1]] >= 0)
    # the TU file lists (row, col); PyG reads it as edge_index = A.T (src/datasets/mutag.py:49)
    ei = np.stack([new_id[A[sel, 0]], new_id[A[sel, 1]]])
    np.savez_compressed(f"{HERE}/mutag128.npz", edge_index=ei.astype(np.int32), batch=np.asarray(batch, dtype=np.int32),
                        node_label=np.asarray(labels, dtype=np.int8), y=gl[kept].astype(np.int8))
    # 3. the WHOLE file's topology (all 4337 graphs, 266 894 directed edges) for the reference-held known answers:
    #    the in-degree histogram, "edge 2k+1 reverses edge 2k", and the dual-edge count the reference's author recorded
    #    next to the pair loops (`# len dual_edges: 451808`, src/datasets/mutag_dual.py:385).  Only the even edges are stored
    #    (the odd ones are their reverses, asserted above); ids are delta-coded so the npz stays small.
    assert facts["pairs_are_reverses"] and facts["graph_indicator_sorted"]
    even = A[0::2].astype(np.int64)
    np.savez_compressed(f"{HERE}/mutag_full.npz", even_src_delta=np.diff(even[:, 0], prepend=0).astype(np.int32),
                        even_dst_minus_src=(even[:, 1] - even[:, 0]).astype(np.int32),
                        nodes_per_graph=np.bincount(gi, minlength=G).astype(np.int32), kept_mask=mask.astype(np.int8))
    facts["line_graph_directed_dual_edges"] = 451808           # the reference's own recorded value (mutag_dual.py:385)
    facts["line_graph_rule"] = "pairs of directed edges sharing their FIRST endpoint, both orders (mutag_dual.py:345-377)"
    with open(f"{HERE}/mutag_facts.json", "w") as f:
        json.dump(facts, f, indent=1)
    print("mutag128:", ei.shape, off, facts)


def make_oracle_goldens():
    from oracle import bookkeeping as bk, modules as om, ops
    from tests.graphs import random_batch, shuffle_edges
    torch.manual_seed(0)
    # --- case A: GIN + edge attention (C2-like), training mode with pinned randomness ---------------------
    for name, backbone, learn_edge_att, H in (("gin_edge", "GIN", True, 16), ("pna_node", "PNA", False, 16)):
        ei, batch, N = random_batch(7 if backbone == "GIN" else 8, 6, 3, 14)
        ei = shuffle_edges(ei, 1)
        E, G = ei.shape[1], 6
        g = torch.Generator().manual_seed(3)
        x = torch.randn(N, 5, generator=g)
        y = torch.randint(0, 2, (G, 1), generator=g).float()
        cfg = dict(model_name=backbone, n_layers=2, hidden_size=H, dropout_p=0.0, use_edge_attr=False,
                   aggregators=["mean", "min", "max", "std"], scalers=False,
                   deg=torch.from_numpy(bk.deg_histogram(ei, N)))
        clf = (om.GIN if backbone == "GIN" else om.PNA)(5, 0, 2, False, cfg)
        ext = om.ExtractorMLP(H, learn_edge_att)
        M = E if learn_edge_att else N
        C1, C2 = (4 * H, H) if learn_edge_att else (2 * H, H)
        u = torch.rand(M, 1, generator=g).clamp_(1e-10, 1 - 1e-10)
        masks = [(torch.rand(M, C1, generator=g) > 0.5).float(), (torch.rand(M, C2, generator=g) > 0.5).float()]
        gsat = om.GSAT(clf, ext, om.Criterion(2, False), learn_edge_att=learn_edge_att).train()
        from types import SimpleNamespace as NS
        data = NS(x=x, edge_index=ei, batch=batch, edge_attr=None, y=y)
        edge_att, loss, loss_dict, clf_logits, aux = gsat.forward_pass(data, 12, True, u=u, masks=masks)
        loss.backward()
        out = dict(x=x, edge_index=ei, batch=batch, y=y, u=u, mask1=masks[0], mask2=masks[1], deg=cfg["deg"],
                   emb=aux["emb"], att_log_logits=aux["att_log_logits"], att=aux["att"], edge_att=edge_att,
                   clf_logits=clf_logits, loss=loss.detach().view(1), pred=torch.tensor([loss_dict["pred"]]),
                   info=torch.tensor([loss_dict["info"]]))
        for k, v in list(clf.state_dict().items()):
            out["clf." + k] = v
        for k, v in list(ext.state_dict().items()):
            out["ext." + k] = v
        for k, p in clf.named_parameters():
            out["grad.clf." + k] = p.grad if p.grad is not None else torch.zeros_like(p)
        for k, p in ext.named_parameters():
            out["grad.ext." + k] = p.grad
        np.savez_compressed(f"{HERE}/oracle_{name}.npz", **{k: v.detach().numpy() for k, v in out.items()})
        print(name, {k: round(v, 6) for k, v in loss_dict.items()})


if __name__ == "__main__":
    if os.path.isdir(RAW):
        make_mutag()
    else:
        print("reference data not present: keeping the committed mutag128.npz")
    make_oracle_goldens()
