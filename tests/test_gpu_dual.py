"""-m gpu: DP-GSAT orchestrator (src/run_gsat.py:189-281) vs the oracle restatement, node-attention mode (MUTAG configs)."""
from types import SimpleNamespace as NS

import pytest
import torch

from oracle import modules as om
from tests.graphs import line_graph, random_batch
from tests.util import close

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("epoch,training", [(3, True), (60, True), (60, False)])
def test_dual_forward_pass(dev, epoch, training):
    import dp_gsat_amd as G
    ei, batch, N = random_batch(5, 6, 3, 12, isolated=False)          # edges grouped by graph (not shuffled) -> sorted dual batch
    E = ei.shape[1]
    dei, dbatch = line_graph(ei, batch)
    Nd, Ed = E, dei.shape[1]
    H = 16
    g = torch.Generator().manual_seed(epoch)
    pdata = NS(x=torch.randn(N, 7, generator=g), edge_index=ei, batch=batch, edge_attr=None, y=torch.randint(0, 2, (6, 1), generator=g).float(),
               edge_label=(torch.rand(E, generator=g) > 0.7).float())
    ddata = NS(x=torch.randn(Nd, 9, generator=g), edge_index=dei, batch=dbatch, edge_attr=None, y=pdata.y.clone())
    cfg = dict(model_name="GIN", n_layers=2, hidden_size=H, dropout_p=0.0)
    mcfg = dict(pred_loss_coef=1, info_loss_coef=1, fix_r=False, decay_interval=10, decay_r=0.1, final_r=0.5)
    opc, odc = om.GIN(7, 0, 2, False, cfg), om.GIN(9, 0, 2, False, cfg)
    ope, ode = om.ExtractorMLP(H, False), om.ExtractorMLP(H, False)
    pu = torch.rand(N, 1, generator=g).clamp_(1e-10, 1 - 1e-10)
    dU = torch.rand(Nd, 1, generator=g)
    pm = [(torch.rand(N, 2 * H, generator=g) > 0.5).float(), (torch.rand(N, H, generator=g) > 0.5).float()]
    dm = [(torch.rand(Nd, 2 * H, generator=g) > 0.5).float(), (torch.rand(Nd, H, generator=g) > 0.5).float()]
    od = om.DualGSAT(opc, ope, odc, ode, mcfg, mcfg, False, False).train(training)
    o_att, o_loss, o_ld, o_logits = od.dual_forward_pass(pdata, ddata, epoch, training, pu, dU, pm, dm)

    pc = G.get_model(7, 0, 2, False, cfg, dev); pc.load_state_dict(opc.state_dict())
    dc = G.get_model(9, 0, 2, False, cfg, dev); dc.load_state_dict(odc.state_dict())
    shared = {"learn_edge_att": False, "extractor_dropout_p": 0.5}
    pe = G.ExtractorMLP(H, shared, "primal").to(dev); de = G.ExtractorMLP(H, shared, "dual").to(dev)
    pe.load_state_dict({"primal_" + k: v for k, v in ope.state_dict().items()})
    de.load_state_dict({"dual_" + k: v for k, v in ode.state_dict().items()})
    dg = G.DualGSAT(pc, pe, None, dc, de, None, mcfg, mcfg, False, False).train(training)
    tod = lambda d: NS(**{k: (v.to(dev) if isinstance(v, torch.Tensor) else v) for k, v in vars(d).items()})
    att, loss, ld, logits = dg.dual_forward_pass(tod(pdata), tod(ddata), epoch, training, pu.to(dev), dU.to(dev),
                                                 [m.to(dev) for m in pm], [m.to(dev) for m in dm])
    close(att, o_att, what="primal_edge_att")
    close(logits, o_logits, what="primal_clf_logits")
    close(loss, o_loss, what="loss")
    for k in ("loss", "pred", "info"):
        assert abs(ld[k] - o_ld[k]) < 1e-4 * max(1.0, abs(o_ld[k])), k
    if training:
        o_loss.backward(); loss.backward()
        for (k, p), (_, q) in zip(list(pc.named_parameters()) + list(de.named_parameters()), list(opc.named_parameters()) + list(ode.named_parameters())):
            if q.grad is not None:
                close(p.grad, q.grad, 1e-4, what=k)
