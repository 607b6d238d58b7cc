"""GSAT step: extractor -> stochastic attention -> (symmetrise | lift) -> masked backbone pass -> loss.

``GSAT`` / ``ExtractorMLP`` keep the constructor and method signatures of the reference
(example/gsat.py:12-139; DP variants src/run_gsat.py:860-927) so that ``example/trainer.py`` and
``src/run_gsat.py`` can import them unchanged; see INTEGRATION.md.
"""
from __future__ import annotations

from typing import Optional

import torch
import torch.nn as nn

from .get_model import MLP
from .graph_index import get_index, sync_free
from .ops import ExtractorAttention, InfoLoss, Lift, LiftedAttention, Sample, Symmetrise, device_seed, edge_tensor, new_seed


class ExtractorMLP(nn.Module):
    """Per-edge (``learn_edge_att``) or per-node attention logits.

    Accepts both reference constructors:
      ``ExtractorMLP(hidden_size, learn_edge_att)``            (example/gsat.py:122)
      ``ExtractorMLP(hidden_size, shared_config, type)``       (src/run_gsat.py:890; type in {"primal","dual"})
    and both call forms ``extractor(emb, edge_index, batch)`` / ``extractor(emb, edge_index, batch, type)``.
    """

    def __init__(self, hidden_size, learn_edge_att_or_shared_config, type: Optional[str] = None):
        super().__init__()
        cfg = learn_edge_att_or_shared_config
        if isinstance(cfg, dict):                                       # DP-GSAT form
            if type not in ("primal", "dual"):
                raise ValueError("type must be 'primal' or 'dual'")
            learn_edge_att, dropout_p = cfg["learn_edge_att"], cfg["extractor_dropout_p"]
            prefix = type + "_"
        else:                                                           # vanilla form
            learn_edge_att, dropout_p, prefix = bool(cfg), 0.5, ""
        self._prefix = prefix
        self.type = type
        setattr(self, prefix + "learn_edge_att", learn_edge_att)
        if learn_edge_att:
            mlp = MLP([hidden_size * 2, hidden_size * 4, hidden_size, 1], dropout=dropout_p)
        else:
            mlp = MLP([hidden_size * 1, hidden_size * 2, hidden_size, 1], dropout=dropout_p)
        setattr(self, prefix + "feature_extractor", mlp)                # keys: [primal_|dual_]feature_extractor.{0,4,8}.*

    @property
    def mlp(self) -> MLP:
        return getattr(self, self._prefix + "feature_extractor")

    @property
    def edge_mode(self) -> bool:
        return bool(getattr(self, self._prefix + "learn_edge_att"))

    def attend(self, emb, edge_index, batch, noise=None, dropout_masks=None, seed=None):
        """(att_log_logits, att): logits and the sampled attention from ONE fused pipeline.
        ``noise``: uniform u in (0,1) per row -> concrete sample in training mode; None -> sigmoid(logits); the string
        "philox" -> concrete sample with u drawn inside the head kernel (Philox stream 4 of the step's seed, no noise tensor).
        ``dropout_masks``: optional explicit keep-masks [(M,C1),(M,C2)] (parity tests); default Philox(seed)."""
        index = get_index(edge_index, emb.shape[0])
        segments = index.graphs(batch)
        l1, l2, l3 = self.mlp.linears()
        m1, m2 = dropout_masks if dropout_masks is not None else (None, None)
        philox_noise = isinstance(noise, str)
        if philox_noise:
            if noise != "philox":
                raise ValueError("noise must be a tensor, None or 'philox'")
            noise = None
        seed_dev = None
        if seed is None:
            need = self.training and ((self.mlp.dropout_p > 0 and m1 is None) or philox_noise)
            if need and sync_free():
                # graph-capturable: the seed lives on the device and is redrawn by a graph-safe RNG op on every replay
                seed, seed_dev = 0, device_seed(emb.device)
            else:
                seed = new_seed() if need else 0
        return ExtractorAttention.apply(emb, l1.weight, l1.bias, l2.weight, l2.bias, l3.weight, l3.bias, index, segments,
                                        self.edge_mode, self.training, self.mlp.dropout_p, seed, m1, m2, noise, seed_dev, philox_noise)

    def forward(self, emb, edge_index, batch, type: Optional[str] = None, dropout_masks=None):
        if type is not None and self.type is not None and type != self.type:
            raise ValueError(f"extractor built for type={self.type!r} called with {type!r}")
        return self.attend(emb, edge_index, batch, None, dropout_masks)[0]


def get_r(decay_interval, decay_r, current_epoch, init_r=0.9, final_r=0.5):
    """example/gsat.py:105-110."""
    r = init_r - current_epoch // decay_interval * decay_r
    if r < final_r:
        r = final_r
    return r


def concrete_sample(att_log_logit, temp=1.0, training=True, noise=None):
    """src/run_gsat.py:877-885 / example/gsat.py:94-103.  ``noise`` = explicit u (parity), else drawn here."""
    if training:
        if noise is None:
            noise = torch.empty_like(att_log_logit).uniform_(1e-10, 1 - 1e-10)
        return Sample.apply(att_log_logit, noise, 1, temp, 0.0)
    return Sample.apply(att_log_logit, None, 0, 1.0, 0.0)


def gumbel_sigmoid(logits, tau=1.0, eps=1e-10, noise=None):
    """src/run_gsat.py:182-187."""
    if noise is None:
        noise = torch.rand_like(logits)
    return Sample.apply(logits, noise, 2, tau, eps)


def lift_node_att_to_edge_att(node_att, edge_index):
    """example/gsat.py:112-117.  Returns a lazy view (ops.LiftedAttention): the PNA aggregation forms node_att[src] * node_att[dst] at
    its mask load; any other use writes the [E, 1] tensor out on first touch."""
    return LiftedAttention(node_att, get_index(edge_index, node_att.shape[0]))


def symmetrise_edge_att(att, edge_index, num_nodes):
    """example/gsat.py:79-85: average with the reverse edge iff the edge set is symmetric."""
    index = get_index(edge_index, num_nodes)
    if sync_free():
        rev, flags = index.rev_and_flag                   # the symmetric / not-symmetric decision stays on the device
        return Symmetrise.apply(att, rev, flags)
    if index.is_undirected:
        return Symmetrise.apply(att, index.rev)
    return att


def info_loss(att, r):
    """example/gsat.py:31 ; src/run_gsat.py:127,132 (tensor prior allowed, detached)."""
    return InfoLoss.apply(edge_tensor(att), edge_tensor(r))


class GSAT(nn.Module):
    """example/gsat.py:12-117."""

    def __init__(self, clf, extractor, criterion, optimizer, learn_edge_att=True, final_r=0.7, decay_interval=10, decay_r=0.1):
        super().__init__()
        self.clf = clf
        self.extractor = extractor
        self.criterion = criterion
        self.optimizer = optimizer
        self.device = next(self.parameters()).device
        self.learn_edge_att = learn_edge_att
        self.final_r = final_r
        self.decay_interval = decay_interval
        self.decay_r = decay_r
        self.sync_loss_dict = True       # the reference calls .item() three times per step (example/gsat.py:34)

    def __loss__(self, att, clf_logits, clf_labels, epoch, loss_weights=None):
        pred_loss = self.criterion(clf_logits, clf_labels)
        r = self.get_r(self.decay_interval, self.decay_r, epoch, final_r=self.final_r)
        i_loss = info_loss(att, r)
        if loss_weights is None:
            loss = pred_loss + i_loss
        else:       # data-parallel shards: dist.global_loss_weights() makes the averaged gradients those of the global batch
            loss = pred_loss * loss_weights[0] + i_loss * loss_weights[1]
        if self.sync_loss_dict:
            vals = torch.stack([loss.detach(), pred_loss.detach(), i_loss.detach()]).tolist()   # one sync, not three
            loss_dict = {"loss": vals[0], "pred": vals[1], "info": vals[2]}
        else:
            loss_dict = {"loss": loss.detach(), "pred": pred_loss.detach(), "info": i_loss.detach()}
        return loss, loss_dict

    def forward_pass(self, data, epoch, training, noise=None, dropout_masks=None, loss_weights=None):
        """Returns (edge_att, loss, loss_dict, clf_logits) like the reference.  ``noise`` / ``dropout_masks``
        optionally pin the randomness (same-seed parity is impossible against torch's CPU generator); ``loss_weights``
        = (graph weight, attention-row weight) of a data-parallel shard (dp_gsat_amd.dist.global_loss_weights)."""
        N = data.x.shape[0]
        num_graphs = getattr(data, "num_graphs", None)
        if num_graphs is not None:      # PyG batches know their graph count: prime the segment cache without `batch.max()` (a sync)
            get_index(data.edge_index, N).graphs(data.batch, int(num_graphs))
        emb = self.clf.get_emb(data.x, data.edge_index, batch=data.batch, edge_attr=data.edge_attr)
        if training and noise is None:
            noise = "philox"            # u ~ U(1e-10, 1 - 1e-10) (example/gsat.py:96) drawn inside the head kernel: no uniform_ launch, no tensor
        _, att = self.extractor.attend(emb, data.edge_index, data.batch, noise if training else None, dropout_masks)
        if self.learn_edge_att:
            edge_att = symmetrise_edge_att(att, data.edge_index, N)
        else:
            edge_att = self.lift_node_att_to_edge_att(att, data.edge_index)
        clf_logits = self.clf(data.x, data.edge_index, data.batch, edge_attr=data.edge_attr, edge_atten=edge_att)
        loss, loss_dict = self.__loss__(att, clf_logits, data.y, epoch, loss_weights)
        return edge_att, loss, loss_dict, clf_logits

    @staticmethod
    def sampling(att_log_logit, training, noise=None):
        return concrete_sample(att_log_logit, 1.0, training, noise)

    get_r = staticmethod(get_r)
    lift_node_att_to_edge_att = staticmethod(lift_node_att_to_edge_att)
