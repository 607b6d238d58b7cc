"""not gpu: the oracle against its committed golden vectors and the reference's MUTAG data facts."""
import json
import os
from types import SimpleNamespace as NS

import numpy as np
import pytest
import torch

from oracle import bookkeeping as bk
from oracle import modules as om

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def load_case(name):
    z = np.load(os.path.join(GOLD, f"oracle_{name}.npz"))
    return {k: torch.from_numpy(z[k]) for k in z.files}


def build_oracle(case, backbone, learn_edge_att, H=16):
    cfg = dict(model_name=backbone, n_layers=2, hidden_size=H, dropout_p=0.0, use_edge_attr=False,
               aggregators=["mean", "min", "max", "std"], scalers=False, deg=case["deg"])
    clf = (om.GIN if backbone == "GIN" else om.PNA)(5, 0, 2, False, cfg)
    clf.load_state_dict({k[4:]: v for k, v in case.items() if k.startswith("clf.")})
    ext = om.ExtractorMLP(H, learn_edge_att)
    ext.load_state_dict({k[4:]: v for k, v in case.items() if k.startswith("ext.")})
    return cfg, clf, ext


@pytest.mark.parametrize("name,backbone,edge", [("gin_edge", "GIN", True), ("pna_node", "PNA", False)])
def test_oracle_reproduces_golden(name, backbone, edge):
    c = load_case(name)
    _, clf, ext = build_oracle(c, backbone, edge)
    gsat = om.GSAT(clf, ext, om.Criterion(2, False), learn_edge_att=edge).train()
    data = NS(x=c["x"], edge_index=c["edge_index"], batch=c["batch"], edge_attr=None, y=c["y"])
    edge_att, loss, ld, logits, aux = gsat.forward_pass(data, 12, True, u=c["u"], masks=[c["mask1"], c["mask2"]])
    loss.backward()
    for k, v in (("emb", aux["emb"]), ("att_log_logits", aux["att_log_logits"]), ("att", aux["att"]),
                 ("edge_att", edge_att), ("clf_logits", logits)):
        assert torch.allclose(v, c[k], atol=1e-6, rtol=1e-5), k
    assert abs(loss.item() - c["loss"].item()) < 1e-6
    for k, p in ext.named_parameters():
        assert torch.allclose(p.grad, c["grad.ext." + k], atol=1e-6, rtol=1e-4), k


def test_mutag_fixture_matches_reference_facts():
    """Facts of data/mutag_dual/raw/*.txt recorded by SURVEY.md 8c and re-derived by make_golden.py."""
    with open(os.path.join(GOLD, "mutag_facts.json")) as f:
        facts = json.load(f)
    assert facts["num_graphs"] == 4337 and facts["kept_graphs"] == 2951
    assert facts["num_nodes"] == 131488 and facts["num_directed_edges"] == 266894
    assert facts["in_degree_histogram"] == [2401, 64058, 7570, 42140, 15319]
    assert facts["pairs_are_reverses"] and facts["duplicate_edges"] == 0
    z = np.load(os.path.join(GOLD, "mutag128.npz"))
    ei, batch = z["edge_index"].astype(np.int64), z["batch"].astype(np.int64)
    N = len(batch)
    assert int(batch.max()) + 1 == 128 and np.all(np.diff(batch) >= 0)
    assert bk.is_undirected(ei, N)
    rev = bk.reverse_edge_perm(ei, N)
    E = ei.shape[1]
    assert np.array_equal(rev[0::2], np.arange(1, E, 2)) and np.array_equal(rev[1::2], np.arange(0, E, 2))
    assert np.array_equal(rev[rev], np.arange(E))                      # involution
    assert np.all(batch[ei[0]] == batch[ei[1]])                         # edges never cross graphs
    assert int(z["node_label"].max()) < 14


def test_oracle_on_whole_mutag_file_matches_reference_recorded_answers():
    """Reference-held known answers on the WHOLE Mutagenicity file (tests/golden/mutag_full.npz, from data/mutag_dual/raw):
    the in-degree histogram, `edge 2k+1 reverses edge 2k`, and the dual-edge count the reference's author recorded next to
    the pair loops -- `# len dual_edges: 451808`, src/datasets/mutag_dual.py:385.  This pins the oracle's line-graph rule
    (group by FIRST endpoint, both orders) and its bookkeeping to numbers that come from the reference, not from the oracle."""
    from dp_gsat_amd.synth import mutag_full_topology
    with open(os.path.join(GOLD, "mutag_facts.json")) as f:
        facts = json.load(f)
    ei, batch, kept = mutag_full_topology(os.path.join(GOLD, "mutag_full.npz"))
    ei, batch = ei.numpy(), batch.numpy()
    N, E = len(batch), ei.shape[1]
    assert (N, E, int(batch.max()) + 1, int(kept.sum())) == (131488, 266894, 4337, 2951)
    assert bk.deg_histogram(ei, N, minlength=0).tolist() == [2401, 64058, 7570, 42140, 15319]
    rowptr, perm = bk.csr_by(ei[1], N)
    assert np.bincount(np.diff(rowptr)).tolist() == facts["in_degree_histogram"]
    assert bk.is_undirected(ei, N)
    rev = bk.reverse_edge_perm(ei, N)
    assert np.array_equal(rev[0::2], np.arange(1, E, 2)) and np.array_equal(rev[1::2], np.arange(0, E, 2))
    dual = bk.line_graph_by_source(ei)
    assert dual.shape == (2, 451808) == (2, facts["line_graph_directed_dual_edges"])
    assert np.all(ei[0][dual[0]] == ei[0][dual[1]])                    # every dual edge joins two primal edges leaving one node


def test_oracle_undirected_line_graph_rule():
    """src/datasets/ba_2motifs_dual.py:35-62 on BA-2motifs-shaped graphs: E/2 dual nodes per graph, sum_v d_v (d_v - 1) dual
    edges, motif label on the 5 (cycle) or 6 (house) edges among nodes >= 20, features x[a] || x[b] with a < b."""
    from dp_gsat_amd.synth import ba2motifs_batch
    d = ba2motifs_batch(num_graphs=6, seed=3)
    ei, batch, x = d.edge_index.numpy(), d.batch.numpy(), d.x.numpy()
    dei, und, dbatch, dx, dlabel = bk.line_graph_undirected(ei, batch, x, motif_start=20)
    M = ei.shape[1] // 2
    assert und.shape == (2, M) and np.all(und[0] < und[1]) and len(dbatch) == M
    deg = np.bincount(ei[0], minlength=len(batch))
    assert dei.shape[1] == int((deg * (deg - 1)).sum())
    keys = und[0] * len(batch) + und[1]
    assert np.all(np.diff(keys) > 0)                                   # numbered in row-major (a, b) order
    per_graph = np.bincount(dbatch, weights=dlabel, minlength=6)
    assert set(per_graph.tolist()) <= {5.0, 6.0}
    assert np.array_equal(dx, np.concatenate([x[und[0]], x[und[1]]], axis=1))
    k = dei[0] * M + dei[1]
    assert np.all(np.diff(k) > 0)                                      # dense_to_sparse order, no duplicates
    shared = (und[0][dei[0]] == und[0][dei[1]]) | (und[0][dei[0]] == und[1][dei[1]]) | \
             (und[1][dei[0]] == und[0][dei[1]]) | (und[1][dei[0]] == und[1][dei[1]])
    assert shared.all()
