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
