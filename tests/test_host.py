"""not gpu: C-ABI library loads and exports every declared symbol; host-side protocol of the product package."""
import ctypes
import glob
import os
import re

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    names = set()
    for h in glob.glob(os.path.join(ROOT, "include", "*.h")):
        txt = re.sub(r"/\*.*?\*/", "", open(h).read(), flags=re.S)
        names |= set(re.findall(r"\b(gsat_[a-z0-9_]+)\s*\(", txt))
    return names


def test_library_exports_every_declared_symbol():
    import __graft_entry__ as ge
    ge.build()
    from dp_gsat_amd import _lib
    lib = ctypes.CDLL(_lib.LIB_PATH)
    decl = declared_symbols()
    assert len(decl) >= 25
    for name in sorted(decl):
        assert hasattr(lib, name), f"{name} declared in include/gsat_hip.h but not exported"
    assert decl == set(_lib.SIGNATURES), decl ^ set(_lib.SIGNATURES)
    assert _lib.load().gsat_abi_version() == 4


def test_ctypes_structs_match_header_layout():
    from dp_gsat_amd._lib import AttnArgs, AttnGrads
    # 3 i64, 5 i32, float, u64, 24 pointers, size_t, then noise_philox + fused (2 x i32), node_ptr (ABI 3)
    assert ctypes.sizeof(AttnArgs) == 3 * 8 + 5 * 4 + 4 + 8 + 24 * 8 + 8 + 8 + 8
    assert ctypes.sizeof(AttnGrads) == 16 * 8 + 8


def test_product_has_no_oracle_import_and_no_cpu_fallback():
    for path in glob.glob(os.path.join(ROOT, "dp_gsat_amd", "*.py")):
        src = open(path).read()
        assert not re.search(r"^\s*(from|import)\s+oracle", src, flags=re.M), path
    import dp_gsat_amd as G
    from dp_gsat_amd._lib import GsatHipError
    ei = torch.tensor([[0, 1], [1, 0]])
    with pytest.raises(GsatHipError):
        G.BatchIndex(ei, 2)                      # CPU tensors: fail loudly, never fall back
    with pytest.raises(GsatHipError):
        G.InstanceNorm(4)(torch.randn(3, 4), torch.zeros(3, dtype=torch.int64))


def test_state_dict_keys_match_reference_layout():
    import dp_gsat_amd as G
    from oracle import modules as om
    deg = torch.tensor([0, 5, 3, 2, 1, 0, 0, 0, 0, 0])
    for cfg, x_dim, e_dim in [
        (dict(model_name="GIN", n_layers=2, hidden_size=16, dropout_p=0.3), 7, 0),
        (dict(model_name="GIN", n_layers=2, hidden_size=16, dropout_p=0.3), 4, 1),                         # GINEConv
        (dict(model_name="GIN", n_layers=2, hidden_size=16, dropout_p=0.3, atom_encoder=True), 9, 3),
        (dict(model_name="PNA", n_layers=3, hidden_size=16, dropout_p=0.3, aggregators=["mean", "min", "max", "std"],
              scalers=False, deg=deg, use_edge_attr=False, atom_encoder=True), 9, 3),
        (dict(model_name="PNA", n_layers=2, hidden_size=16, dropout_p=0.3, aggregators=["mean", "min", "max", "std", "sum"],
              scalers=True, deg=deg), 14, 0),
    ]:
        mine = G.get_model(x_dim, e_dim, 2, False, cfg, "cpu")
        ref = (om.GIN if cfg["model_name"] == "GIN" else om.PNA)(x_dim, e_dim, 2, False, cfg)
        sd_m, sd_r = mine.state_dict(), ref.state_dict()
        assert list(sd_m.keys()) == list(sd_r.keys())
        assert all(sd_m[k].shape == sd_r[k].shape for k in sd_m)
        mine.load_state_dict(sd_r)
    sp_cfg = dict(model_name="SPMotifNet", n_layers=2, hidden_size=32)
    assert list(G.get_model(4, 1, 3, False, sp_cfg, "cpu").state_dict()) == list(om.SPMotifNet(4, 1, 3, False, sp_cfg).state_dict())
    # keys the reference's checkpoints carry
    keys = set(G.get_model(7, 0, 2, False, dict(model_name="GIN", n_layers=2, hidden_size=16, dropout_p=0.3), "cpu").state_dict())
    assert {"node_encoder.weight", "convs.0.eps", "convs.0.nn.0.weight", "convs.0.nn.1.running_mean",
            "convs.0.nn.1.num_batches_tracked", "convs.0.nn.3.bias", "fc_out.0.weight"} <= keys
    assert list(G.ExtractorMLP(16, True).state_dict()) == [f"feature_extractor.{i}.{p}" for i in (0, 4, 8) for p in ("weight", "bias")]
    dp = G.ExtractorMLP(16, {"learn_edge_att": False, "extractor_dropout_p": 0.5}, "primal")
    assert list(dp.state_dict())[0] == "primal_feature_extractor.0.weight" and dp.primal_learn_edge_att is False
    assert G.ExtractorMLP(16, True).feature_extractor[0].weight.shape == (64, 32)        # [4H, 2H]
    assert G.ExtractorMLP(16, False).feature_extractor[0].weight.shape == (32, 16)       # [2H, H]


def test_error_conventions():
    import dp_gsat_amd as G
    with pytest.raises(ValueError, match="Unknown model name"):
        G.get_model(3, 0, 2, False, {"model_name": "GCN"}, "cpu")                        # src/utils/get_model.py:15
    with pytest.raises(ValueError):
        G.ExtractorMLP(8, {"learn_edge_att": True, "extractor_dropout_p": 0.5}, "tertiary")
    assert G.get_r(10, 0.1, 0) == 0.9 and G.get_r(10, 0.1, 1000, final_r=0.7) == 0.7
    assert G.get_preds(torch.tensor([[2.0], [-1.0]]), False).tolist() == [[1.0], [0.0]]
    assert G.get_preds(torch.tensor([[0.1, 0.9, 0.0]]), False).tolist() == [1.0]
    crit = G.Criterion(2, False)
    assert torch.allclose(crit(torch.zeros(4, 1), torch.ones(4, 1)), torch.tensor(np.log(2.0), dtype=torch.float32))


def test_sharding_matches_oracle_bit_exact():
    from dp_gsat_amd.dist import shard_graphs_lpt, take_graphs, edges_per_graph
    from dp_gsat_amd import synth
    from oracle import bookkeeping as bk
    for seed, W in [(0, 2), (1, 4), (2, 8)]:
        e = np.random.RandomState(seed).randint(0, 300, size=61)
        assert shard_graphs_lpt(e, W) == bk.shard_graphs_lpt(e, W)
    b = synth.ba2motifs_batch(num_graphs=12, seed=0)
    epg = edges_per_graph(b)
    assert epg.sum() == b.num_edges
    parts = shard_graphs_lpt(epg, 3)
    subs = [take_graphs(b, p) for p in parts]
    assert sum(s.num_edges for s in subs) == b.num_edges and sum(s.num_nodes for s in subs) == b.num_nodes
    for s, p in zip(subs, parts):
        assert int(s.edge_index.max()) < s.num_nodes and s.num_graphs == len(p)
        assert torch.all(s.batch[s.edge_index[0]] == s.batch[s.edge_index[1]])
        assert torch.equal(s.y, b.y[torch.tensor(p)])


def test_synthetic_shapes():
    from dp_gsat_amd import synth
    b = synth.molhiv_batch(num_graphs=64, seed=0)
    assert b.x.dtype == torch.int64 and b.x.shape[1] == 9 and b.batch.shape[0] == b.num_nodes
    assert torch.bincount(b.edge_index[1], minlength=b.num_nodes).max() <= 4
    assert 35 < b.num_edges / 64 < 80
    c = synth.spmotif_batch(num_graphs=32, seed=0)
    assert c.edge_attr.shape == (c.num_edges, 1)
    m = synth.mutag_batch(os.path.join(ROOT, "tests", "golden", "mutag128.npz"))
    assert m.num_graphs == 128 and m.x.shape[1] == 14 and m.x.sum() == m.num_nodes
    p = synth.powerlaw_batch(num_nodes=4000, num_edges=40000, num_graphs=4, seed=0)
    assert torch.all(p.batch[p.edge_index[0]] == p.batch[p.edge_index[1]])
