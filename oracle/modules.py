"""nn.Module restatement of the reference's backbones / extractor / GSAT step (ORACLE, test-only).

Module structure reproduces the reference's attribute names so that ``state_dict`` keys are the
reference's: GIN ``node_encoder, edge_encoder, convs.{i}.eps, convs.{i}.nn.{0,1,3},
convs.{i}.lin, fc_out.0``; PNA ``convs.{i}.post_nn.0, batch_norms.{i}.module, fc_out.{0,2,4}``;
extractor ``feature_extractor.{0,4,8}``.  Randomness (concrete noise ``u``, extractor dropout
masks) is always passed in explicitly; backbone dropout uses torch's generator (set p=0 for parity).
"""
from __future__ import annotations

from typing import Optional

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import bookkeeping as bk
from . import ops

# [3P] ogb 1.3.2 ogb/utils/features.py get_atom_feature_dims / get_bond_feature_dims
ATOM_FEATURE_DIMS = [119, 5, 12, 12, 10, 6, 6, 2, 2]
BOND_FEATURE_DIMS = [5, 6, 2]


class _CatEncoder(nn.Module):
    """[3P] ogb AtomEncoder / BondEncoder: sum of per-column nn.Embedding lookups, xavier_uniform init."""

    def __init__(self, emb_dim, dims, list_name):
        super().__init__()
        lst = nn.ModuleList()
        for d in dims:
            e = nn.Embedding(d, emb_dim)
            nn.init.xavier_uniform_(e.weight.data)
            lst.append(e)
        setattr(self, list_name, lst)
        self._list_name = list_name

    def forward(self, x):
        lst = getattr(self, self._list_name)
        out = 0
        for i in range(x.shape[1]):
            out = out + lst[i](x[:, i])
        return out


class AtomEncoder(_CatEncoder):
    def __init__(self, emb_dim):
        super().__init__(emb_dim, ATOM_FEATURE_DIMS, "atom_embedding_list")


class BondEncoder(_CatEncoder):
    def __init__(self, emb_dim):
        super().__init__(emb_dim, BOND_FEATURE_DIMS, "bond_embedding_list")


class InstanceNorm(nn.Module):
    """parameter-free placeholder so Sequential indices match (src/utils/get_model.py:64)."""

    def __init__(self, channels):
        super().__init__()
        self.channels = channels


class MLP(nn.Sequential):
    """src/utils/get_model.py:57-68 (+ BatchSequential :47-54)."""

    def __init__(self, channels, dropout, bias=True):
        m = []
        for i in range(1, len(channels)):
            m.append(nn.Linear(channels[i - 1], channels[i], bias))
            if i < len(channels) - 1:
                m.append(InstanceNorm(channels[i]))
                m.append(nn.ReLU())
                m.append(nn.Dropout(dropout))
        super().__init__(*m)
        self.p = dropout

    def weights(self):
        return [(mod.weight, mod.bias) for mod in self if isinstance(mod, nn.Linear)]

    def forward(self, inputs, batch, num_seg=None, masks=None):
        num_seg = int(batch.max()) + 1 if num_seg is None else num_seg
        return ops.mlp_forward(inputs, batch, num_seg, self.weights(), self.p, masks, self.training)


class ExtractorMLP(nn.Module):
    """example/gsat.py:120-139."""

    def __init__(self, hidden_size, learn_edge_att, dropout=0.5):
        super().__init__()
        self.learn_edge_att = learn_edge_att
        if learn_edge_att:
            self.feature_extractor = MLP([hidden_size * 2, hidden_size * 4, hidden_size, 1], dropout=dropout)
        else:
            self.feature_extractor = MLP([hidden_size * 1, hidden_size * 2, hidden_size, 1], dropout=dropout)

    def forward(self, emb, edge_index, batch, masks=None):
        G = int(batch.max()) + 1
        return ops.extractor_forward(emb, edge_index, batch, G, self.feature_extractor.weights(),
                                     self.learn_edge_att, self.feature_extractor.p, masks, self.training)


def _gin_mlp(i, o):
    return nn.Sequential(nn.Linear(i, o), nn.BatchNorm1d(o), nn.ReLU(inplace=True), nn.Linear(o, o))


class GINConv(nn.Module):
    """src/models/conv_layers.py:14-34 ; [3P] BaseGINConv(nn, eps=0, train_eps=False): eps is a buffer."""

    def __init__(self, nn_):
        super().__init__()
        self.nn = nn_
        self.register_buffer("eps", torch.Tensor([0.0]))

    def forward(self, x, edge_index, edge_attr=None, edge_atten=None):
        return self.nn(ops.gin_aggregate(x, edge_index, edge_atten, float(self.eps)))


class GINEConv(nn.Module):
    """src/models/conv_layers.py:37-66 ; [3P] BaseGINEConv(nn, edge_dim=H) owns lin = Linear(edge_dim, H)."""

    def __init__(self, nn_, edge_dim, in_channels):
        super().__init__()
        self.nn = nn_
        self.register_buffer("eps", torch.Tensor([0.0]))
        self.lin = nn.Linear(edge_dim, in_channels)

    def forward(self, x, edge_index, edge_attr=None, edge_atten=None):
        return self.nn(ops.gine_aggregate(x, edge_index, self.lin(edge_attr), edge_atten, float(self.eps)))


class GIN(nn.Module):
    """src/models/gin.py:12-81."""

    def __init__(self, x_dim, edge_attr_dim, num_class, multi_label, model_config):
        super().__init__()
        self.n_layers = model_config["n_layers"]
        H = model_config["hidden_size"]
        self.edge_attr_dim = edge_attr_dim
        self.dropout_p = model_config["dropout_p"]
        self.use_edge_attr = model_config.get("use_edge_attr", True)
        has_e = edge_attr_dim != 0 and self.use_edge_attr
        if model_config.get("atom_encoder", False):
            self.node_encoder = AtomEncoder(H)
            if has_e:
                self.edge_encoder = BondEncoder(H)
        else:
            self.node_encoder = nn.Linear(x_dim, H)
            if has_e:
                self.edge_encoder = nn.Linear(edge_attr_dim, H)
        self.convs = nn.ModuleList()
        for _ in range(self.n_layers):
            self.convs.append(GINEConv(_gin_mlp(H, H), H, H) if has_e else GINConv(_gin_mlp(H, H)))
        self.fc_out = nn.Sequential(nn.Linear(H, 1 if num_class == 2 and not multi_label else num_class))

    def get_emb(self, x, edge_index, batch, edge_attr=None, edge_atten=None):
        x = self.node_encoder(x)
        if edge_attr is not None and self.use_edge_attr:
            edge_attr = self.edge_encoder(edge_attr)
        for i in range(self.n_layers):
            x = self.convs[i](x, edge_index, edge_attr=edge_attr, edge_atten=edge_atten)
            x = F.relu(x)
            x = F.dropout(x, p=self.dropout_p, training=self.training)
        return x

    def get_pred_from_emb(self, emb, batch):
        return self.fc_out(ops.global_add_pool(emb, batch, int(batch.max()) + 1))

    def forward(self, x, edge_index, batch, edge_attr=None, edge_atten=None):
        return self.get_pred_from_emb(self.get_emb(x, edge_index, batch, edge_attr, edge_atten), batch)


class PNAConvSimple(nn.Module):
    """src/models/conv_layers.py:96-191."""

    def __init__(self, in_channels, out_channels, aggregators, scalers, deg, post_layers=1):
        super().__init__()
        self.aggregators, self.scalers = list(aggregators), list(scalers)
        self.avg_deg = ops.pna_avg_deg(deg)
        mods = [nn.Linear(len(aggregators) * len(scalers) * in_channels, out_channels)]
        for _ in range(post_layers - 1):
            mods += [nn.ReLU(), nn.Linear(out_channels, out_channels)]
        self.post_nn = nn.Sequential(*mods)

    def forward(self, x, edge_index, edge_attr=None, edge_atten=None):
        return self.post_nn(ops.pna_aggregate(x, edge_index, edge_atten, self.aggregators, self.scalers,
                                              self.avg_deg, edge_attr))


class _PyGBatchNorm(nn.Module):
    """[3P] torch_geometric.nn.BatchNorm: wrapper keeping nn.BatchNorm1d as ``.module``."""

    def __init__(self, c):
        super().__init__()
        self.module = nn.BatchNorm1d(c)

    def forward(self, x):
        return self.module(x)


class PNA(nn.Module):
    """src/models/pna.py:12-78."""

    def __init__(self, x_dim, edge_attr_dim, num_class, multi_label, model_config):
        super().__init__()
        H = model_config["hidden_size"]
        self.n_layers = model_config["n_layers"]
        self.dropout_p = model_config["dropout_p"]
        self.edge_attr_dim = edge_attr_dim
        use_e = model_config.get("use_edge_attr", True)
        if model_config.get("atom_encoder", False):
            self.node_encoder = AtomEncoder(H)
            if edge_attr_dim != 0 and use_e:
                self.edge_encoder = BondEncoder(H)
        else:
            self.node_encoder = nn.Linear(x_dim, H)
            if edge_attr_dim != 0 and use_e:
                self.edge_encoder = nn.Linear(edge_attr_dim, H)
        aggregators = model_config["aggregators"]
        scalers = ["identity", "amplification", "attenuation"] if model_config["scalers"] else ["identity"]
        deg = model_config["deg"]
        in_channels = (H * 2 if edge_attr_dim == 0 else H * 3) if use_e else H * 2
        self.convs, self.batch_norms = nn.ModuleList(), nn.ModuleList()
        for _ in range(self.n_layers):
            self.convs.append(PNAConvSimple(in_channels, H, aggregators, scalers, deg, post_layers=1))
            self.batch_norms.append(_PyGBatchNorm(H))
        self.fc_out = nn.Sequential(nn.Linear(H, H // 2), nn.ReLU(), nn.Linear(H // 2, H // 4), nn.ReLU(),
                                    nn.Linear(H // 4, 1 if num_class == 2 and not multi_label else num_class))

    def get_emb(self, x, edge_index, batch, edge_attr, edge_atten=None):
        x = self.node_encoder(x)
        if edge_attr is not None:
            edge_attr = self.edge_encoder(edge_attr)
        for conv, bn in zip(self.convs, self.batch_norms):
            h = F.relu(bn(conv(x, edge_index, edge_attr, edge_atten=edge_atten)))
            x = h + x
            x = F.dropout(x, self.dropout_p, training=self.training)
        return x

    def get_pred_from_emb(self, emb, batch):
        return self.fc_out(ops.global_mean_pool(emb, batch, int(batch.max()) + 1))

    def forward(self, x, edge_index, batch, edge_attr, edge_atten=None):
        return self.get_pred_from_emb(self.get_emb(x, edge_index, batch, edge_attr, edge_atten), batch)


class Criterion(nn.Module):
    """src/utils/get_model.py:19-34."""

    def __init__(self, num_class, multi_label):
        super().__init__()
        self.num_class, self.multi_label = num_class, multi_label

    def forward(self, logits, targets):
        if self.num_class == 2 and not self.multi_label:
            return F.binary_cross_entropy_with_logits(logits, targets.float())
        if self.num_class > 2 and not self.multi_label:
            return F.cross_entropy(logits, targets.long())
        is_labeled = targets == targets
        return F.binary_cross_entropy_with_logits(logits[is_labeled], targets[is_labeled].float())


class GSAT(nn.Module):
    """Vanilla GSAT step, example/gsat.py:12-117 (``forward_pass``), with explicit randomness."""

    def __init__(self, clf, extractor, criterion, learn_edge_att=True, final_r=0.7, decay_interval=10, decay_r=0.1):
        super().__init__()
        self.clf, self.extractor, self.criterion = clf, extractor, criterion
        self.learn_edge_att = learn_edge_att
        self.final_r, self.decay_interval, self.decay_r = final_r, decay_interval, decay_r

    def forward_pass(self, data, epoch, training, u=None, masks=None):
        N = data.x.shape[0]
        emb = self.clf.get_emb(data.x, data.edge_index, batch=data.batch, edge_attr=data.edge_attr)      # :75
        att_log_logits = self.extractor(emb, data.edge_index, data.batch, masks=masks)                  # :76
        att = ops.concrete_sample(att_log_logits, u, training)                                          # :77
        if self.learn_edge_att:
            if bk.is_undirected(data.edge_index, N):                                                    # :80
                rev = torch.from_numpy(bk.reverse_edge_perm(data.edge_index, N))
                edge_att = ops.symmetrise(att, rev)                                                     # :81-83
            else:
                edge_att = att
        else:
            edge_att = ops.lift_node_att_to_edge_att(att, data.edge_index)                              # :87
        clf_logits = self.clf(data.x, data.edge_index, data.batch, edge_attr=data.edge_attr, edge_atten=edge_att)   # :89
        pred_loss = self.criterion(clf_logits, data.y)                                                  # :28
        r = ops.get_r(self.decay_interval, self.decay_r, epoch, final_r=self.final_r)                   # :30
        info = ops.info_loss(att, r)                                                                    # :31
        loss = pred_loss + info
        loss_dict = {"loss": loss.item(), "pred": pred_loss.item(), "info": info.item()}
        return edge_att, loss, loss_dict, clf_logits, dict(emb=emb, att_log_logits=att_log_logits, att=att)


def f1_sparsity_loss(p_uv, y_uv, eps=1e-6):
    """src/run_gsat.py:151-180 (asserts / input() dropped)."""
    TP = (p_uv.view(-1) * y_uv.view(-1)).sum()
    P, G = p_uv.sum(), y_uv.sum()
    precision, recall = TP / (P + eps), TP / (G + eps)
    f1 = 2 * precision * recall / (precision + recall + eps)
    return (1 - f1) + p_uv.abs().mean()


class DualGSAT(nn.Module):
    """src/run_gsat.py:189-281 + 121-149 with explicit randomness; plotting / host copies dropped.
    The dual attention is Gumbel-sampled in eval mode too, as the reference does (:222); ``gumbel_noise_in_eval=False`` gives the
    noise-free sigmoid(logits / tau)."""

    def __init__(self, primal_clf, primal_extractor, dual_clf, dual_extractor, primal_cfg, dual_cfg,
                 primal_learn_edge_att, dual_learn_edge_att, gumbel_noise_in_eval=True):
        super().__init__()
        self.gumbel_noise_in_eval = gumbel_noise_in_eval
        self.primal_clf, self.primal_extractor, self.dual_clf, self.dual_extractor = primal_clf, primal_extractor, dual_clf, dual_extractor
        self.pc, self.dc = primal_cfg, dual_cfg
        self.primal_learn_edge_att, self.dual_learn_edge_att = primal_learn_edge_att, dual_learn_edge_att
        self.crit = Criterion(2, False)

    def _edge_att(self, att, data, learn_edge_att):
        N = data.x.shape[0]
        if learn_edge_att:
            if bk.is_undirected(data.edge_index, N):
                return ops.symmetrise(att, torch.from_numpy(bk.reverse_edge_perm(data.edge_index, N)))
            return att
        return ops.lift_node_att_to_edge_att(att, data.edge_index)

    def dual_forward_pass(self, primal_data, dual_data, epoch, training, primal_u=None, dual_U=None, primal_masks=None, dual_masks=None):
        pemb = self.primal_clf.get_emb(primal_data.x, primal_data.edge_index, batch=primal_data.batch, edge_attr=primal_data.edge_attr)   # :191
        plog = self.primal_extractor(pemb, primal_data.edge_index, primal_data.batch, masks=primal_masks)                                  # :199
        patt = ops.concrete_sample(plog, primal_u, training)                                                                               # :204
        demb = self.dual_clf.get_emb(dual_data.x, dual_data.edge_index, batch=dual_data.batch, edge_attr=dual_data.edge_attr)             # :208
        dlog = self.dual_extractor(demb, dual_data.edge_index, dual_data.batch, masks=dual_masks)                                          # :209
        datt = ops.gumbel_sigmoid(dlog, dual_U, tau=0.1) if (training or self.gumbel_noise_in_eval) else (dlog / 0.1).sigmoid()                                            # :222
        f1 = f1_sparsity_loss(datt, primal_data.edge_label.float())                                                                         # :226
        dual_edge_att = self._edge_att(datt, dual_data, self.dual_learn_edge_att)
        primal_edge_att = self._edge_att(patt, primal_data, self.primal_learn_edge_att)
        if epoch > 50:
            primal_edge_att = 0.3 * datt + (1 - 0.3) * primal_edge_att                                                                      # :253
        plogits = self.primal_clf(primal_data.x, primal_data.edge_index, primal_data.batch, edge_attr=primal_data.edge_attr, edge_atten=primal_edge_att)
        dlogits = self.dual_clf(dual_data.x, dual_data.edge_index, dual_data.batch, edge_attr=dual_data.edge_attr, edge_atten=dual_edge_att)
        ppred = self.crit(plogits, primal_data.y) * self.pc["pred_loss_coef"]
        dpred = self.crit(dlogits, dual_data.y) * self.dc["pred_loss_coef"]
        dual_r = self.dc.get("fix_r") or ops.get_r(self.dc["decay_interval"], self.dc["decay_r"], epoch, final_r=self.dc.get("final_r", 0.1), init_r=self.dc.get("init_r", 0.9))
        dinfo = ops.info_loss(dual_edge_att, dual_r) * self.dc["info_loss_coef"]
        pinfo = ops.info_loss(primal_edge_att, dlog.sigmoid().detach()) * self.pc["info_loss_coef"]                                         # :129-132
        loss = ppred + dpred + pinfo + dinfo + f1
        return primal_edge_att, loss, {"loss": (loss - f1).item(), "pred": dpred.item(), "info": dinfo.item()}, plogits


class LEConv(nn.Module):
    """src/models/conv_layers.py:69-92 ; [3P] PyG LEConv(in, out, bias=True): lin1/lin3 with bias, lin2 without, aggr add."""

    def __init__(self, in_channels, out_channels, bias=True):
        super().__init__()
        self.lin1 = nn.Linear(in_channels, out_channels, bias=bias)
        self.lin2 = nn.Linear(in_channels, out_channels, bias=False)
        self.lin3 = nn.Linear(in_channels, out_channels, bias=bias)

    def forward(self, x, edge_index, edge_weight=None, edge_atten=None):
        a, b = self.lin1(x), self.lin2(x)
        m = a[edge_index[0]] - b[edge_index[1]]                       # a_j - b_i
        if edge_weight is not None:
            m = m * edge_weight.view(-1, 1)
        if edge_atten is not None:
            m = m * edge_atten
        return ops.scatter_sum(m, edge_index[1], x.shape[0]) + self.lin3(x)


class SPMotifNet(nn.Module):
    """src/models/spmotif_gnn.py:9-87."""

    def __init__(self, x_dim, edge_attr_dim, num_class, multi_label, model_config):
        super().__init__()
        self.n_layers = model_config["n_layers"]
        H = model_config["hidden_size"]
        self.node_emb = nn.Linear(x_dim, H)
        self.convs = nn.ModuleList(LEConv(H, H) for _ in range(self.n_layers))
        self.relus = nn.ModuleList(nn.ReLU() for _ in range(self.n_layers))
        self.fc_out = nn.Sequential(nn.Linear(H, 2 * H), nn.ReLU(), nn.Linear(2 * H, num_class))
        self.conf_mlp = nn.Sequential(nn.Linear(H, 2 * H), nn.ReLU(), nn.Linear(2 * H, 3))
        self.cq = nn.Linear(3, 3)
        self.conf_fw = nn.Sequential(self.conf_mlp, self.cq)

    def get_emb(self, x, edge_index, batch, edge_attr, edge_atten=None):
        x = self.node_emb(x)
        for conv, relu in zip(self.convs, self.relus):
            x = relu(conv(x, edge_index, edge_weight=edge_attr, edge_atten=edge_atten))
        return x

    def get_pred_from_emb(self, emb, batch):
        return self.fc_out(ops.global_mean_pool(emb, batch, int(batch.max()) + 1))

    def forward(self, x, edge_index, batch, edge_attr, edge_atten=None):
        return self.get_pred_from_emb(self.get_emb(x, edge_index, batch, edge_attr, edge_atten), batch)
