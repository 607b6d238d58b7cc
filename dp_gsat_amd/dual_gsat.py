"""DP-GSAT ("dual/primal") step: a second GSAT on the line graph steers the primal attention.

Restates ``GSAT.dual_forward_pass / __loss__ / f1_sparsity_loss / dual_train_one_batch / dual_eval_one_batch``
of the fork (src/run_gsat.py:121-180, 189-428, 612-637) on top of the HIP operators.  Not reproduced (SURVEY App. C,
marked X): the blocking ``input()`` / ``plt.show()`` / seaborn heat-maps, the host copies that only feed them
(:262-274) and the ``NameError`` on ``old_primal_edge_att`` in edge-attention mode (the unused ``comb_att`` is dropped).
Like the reference (:222) the dual attention is Gumbel-sampled in eval mode as well; ``gumbel_noise_in_eval=False`` opts into the
deterministic sigmoid(logits / tau).  The reference's ``assert`` on the ranges of p_uv / y_uv in ``f1_sparsity_loss`` (a host
sync per step) is not reproduced: both are outputs of a sigmoid / 0-1 labels by construction.
"""
from __future__ import annotations

from typing import Optional

import torch
import torch.nn as nn

from .get_model import Criterion
from .ops import Sample
from .gsat import (concrete_sample, get_r, gumbel_sigmoid, info_loss, lift_node_att_to_edge_att,
                   symmetrise_edge_att)


def f1_sparsity_loss(p_uv, y_uv, eps=1e-6):
    """src/run_gsat.py:151-180: (1 - soft F1(p, y)) + mean|p|.  A handful of scalar reductions over [E]."""
    p, y = p_uv.view(-1), y_uv.view(-1)
    TP = (p * y).sum()
    P, G = p_uv.sum(), y_uv.sum()
    precision = TP / (P + eps)
    recall = TP / (G + eps)
    f1 = 2 * precision * recall / (precision + recall + eps)
    return (1 - f1) + p_uv.abs().mean()


class DualGSAT(nn.Module):
    """The fork's two-model GSAT.  ``*_method_config``: pred_loss_coef, info_loss_coef, fix_r, decay_interval,
    decay_r, final_r, init_r (src/run_gsat.py:75-108); ``*_learn_edge_att`` from shared_config."""

    def __init__(self, primal_clf, primal_extractor, primal_optimizer, dual_clf, dual_extractor, dual_optimizer,
                 primal_method_config, dual_method_config, primal_learn_edge_att, dual_learn_edge_att,
                 primal_num_class=2, primal_multi_label=False, dual_num_class=2, dual_multi_label=False,
                 mix_alpha: float = 0.3, mix_after_epoch: int = 50, gumbel_tau: float = 0.1,
                 gumbel_noise_in_eval: bool = True):
        super().__init__()
        self.primal_clf, self.primal_extractor, self.primal_optimizer = primal_clf, primal_extractor, primal_optimizer
        self.dual_clf, self.dual_extractor, self.dual_optimizer = dual_clf, dual_extractor, dual_optimizer
        self.primal_learn_edge_att, self.dual_learn_edge_att = primal_learn_edge_att, dual_learn_edge_att
        self.primal_criterion = Criterion(primal_num_class, primal_multi_label)
        self.dual_criterion = Criterion(dual_num_class, dual_multi_label)
        for side, cfg in (("primal", primal_method_config), ("dual", dual_method_config)):
            setattr(self, side + "_pred_loss_coef", cfg["pred_loss_coef"])
            setattr(self, side + "_info_loss_coef", cfg["info_loss_coef"])
            setattr(self, side + "_fix_r", cfg.get("fix_r", None))
            setattr(self, side + "_decay_interval", cfg.get("decay_interval", None))
            setattr(self, side + "_decay_r", cfg.get("decay_r", None))
            setattr(self, side + "_final_r", cfg.get("final_r", 0.1))
            setattr(self, side + "_init_r", cfg.get("init_r", 0.9))
        self.mix_alpha, self.mix_after_epoch = mix_alpha, mix_after_epoch
        self.gumbel_tau, self.gumbel_noise_in_eval = gumbel_tau, gumbel_noise_in_eval
        self.sync_loss_dict = True

    # -- src/run_gsat.py:121-149 -----------------------------------------------------------------------------------
    def __loss__(self, primal_att, dual_att, primal_clf_logits, dual_clf_logits, primal_clf_labels, dual_clf_labels,
                 dual_att_log_logits, epoch):
        primal_pred_loss = self.primal_criterion(primal_clf_logits, primal_clf_labels)
        dual_pred_loss = self.dual_criterion(dual_clf_logits, dual_clf_labels)
        dual_r = self.dual_fix_r if self.dual_fix_r else get_r(self.dual_decay_interval, self.dual_decay_r, epoch,
                                                                final_r=self.dual_final_r, init_r=self.dual_init_r)
        dual_info_loss = info_loss(dual_att, dual_r)
        primal_r = dual_att_log_logits.sigmoid().detach()                 # per-edge tensor prior (:129)
        primal_info_loss = info_loss(primal_att, primal_r)
        primal_pred_loss = primal_pred_loss * self.primal_pred_loss_coef
        primal_info_loss = primal_info_loss * self.primal_info_loss_coef
        dual_pred_loss = dual_pred_loss * self.dual_pred_loss_coef
        dual_info_loss = dual_info_loss * self.dual_info_loss_coef
        loss = primal_pred_loss + dual_pred_loss + primal_info_loss + dual_info_loss
        if self.sync_loss_dict:
            v = torch.stack([loss.detach(), dual_pred_loss.detach(), dual_info_loss.detach()]).tolist()
            loss_dict = {"loss": v[0], "pred": v[1], "info": v[2]}       # dual entries overwrite the primal ones (:145-146)
        else:
            loss_dict = {"loss": loss.detach(), "pred": dual_pred_loss.detach(), "info": dual_info_loss.detach()}
        return loss, loss_dict

    def _edge_att(self, att, data, learn_edge_att):
        if learn_edge_att:
            return symmetrise_edge_att(att, data.edge_index, data.x.shape[0])
        return lift_node_att_to_edge_att(att, data.edge_index)

    # -- src/run_gsat.py:189-428 -----------------------------------------------------------------------------------
    def dual_forward_pass(self, primal_data, dual_data, epoch, training, primal_noise=None, dual_noise=None,
                          primal_masks=None, dual_masks=None):
        from .graph_index import get_index
        for d in (primal_data, dual_data):
            if getattr(d, "num_graphs", None) is not None:      # prime the segment cache without `batch.max()` (a host sync)
                get_index(d.edge_index, d.x.shape[0]).graphs(d.batch, int(d.num_graphs))
        primal_emb = self.primal_clf.get_emb(primal_data.x, primal_data.edge_index, batch=primal_data.batch,
                                             edge_attr=primal_data.edge_attr)
        Mp = primal_data.edge_index.shape[1] if self.primal_learn_edge_att else primal_data.x.shape[0]
        if training and primal_noise is None:
            primal_noise = torch.empty(Mp, 1, device=primal_emb.device).uniform_(1e-10, 1 - 1e-10)
        _, primal_node_att = self.primal_extractor.attend(primal_emb, primal_data.edge_index, primal_data.batch,
                                                          primal_noise if training else None, primal_masks)
        dual_emb = self.dual_clf.get_emb(dual_data.x, dual_data.edge_index, batch=dual_data.batch, edge_attr=dual_data.edge_attr)
        dual_att_log_logits = self.dual_extractor(dual_emb, dual_data.edge_index, dual_data.batch, dropout_masks=dual_masks)
        if training or self.gumbel_noise_in_eval:
            dual_node_att = gumbel_sigmoid(dual_att_log_logits, tau=self.gumbel_tau, noise=dual_noise)       # :222
        else:
            dual_node_att = Sample.apply(dual_att_log_logits, None, 0, self.gumbel_tau, 0.0)   # sigmoid(logits / tau), no noise
        y_uv = primal_data.edge_label.float().to(dual_node_att.device)
        f1_loss = f1_sparsity_loss(dual_node_att, y_uv)                                                       # :226-227
        dual_edge_att = self._edge_att(dual_node_att, dual_data, self.dual_learn_edge_att)                     # :231-239
        primal_edge_att = self._edge_att(primal_node_att, primal_data, self.primal_learn_edge_att)             # :241-250
        if epoch > self.mix_after_epoch:                                                                      # :252-253
            primal_edge_att = self.mix_alpha * dual_node_att + (1 - self.mix_alpha) * primal_edge_att
        primal_clf_logits = self.primal_clf(primal_data.x, primal_data.edge_index, primal_data.batch,
                                            edge_attr=primal_data.edge_attr, edge_atten=primal_edge_att)
        dual_clf_logits = self.dual_clf(dual_data.x, dual_data.edge_index, dual_data.batch,
                                        edge_attr=dual_data.edge_attr, edge_atten=dual_edge_att)
        loss, loss_dict = self.__loss__(primal_edge_att, dual_edge_att, primal_clf_logits, dual_clf_logits,
                                        primal_data.y, dual_data.y, dual_att_log_logits, epoch)                # :276
        loss = loss + f1_loss                                                                                 # :281
        return primal_edge_att, loss, loss_dict, primal_clf_logits

    # -- src/run_gsat.py:612-637 -----------------------------------------------------------------------------------
    def dual_train_one_batch(self, primal_data, dual_data, epoch):
        for m in (self.primal_extractor, self.primal_clf, self.dual_extractor, self.dual_clf):
            m.train()
        att, loss, loss_dict, clf_logits = self.dual_forward_pass(primal_data, dual_data, epoch, training=True)
        self.primal_optimizer.zero_grad()
        self.dual_optimizer.zero_grad()
        loss.backward()
        self.primal_optimizer.step()
        self.dual_optimizer.step()
        return att.data.cpu().reshape(-1), loss_dict, clf_logits.data.cpu()

    @torch.no_grad()
    def dual_eval_one_batch(self, primal_data, dual_data, epoch):
        for m in (self.primal_extractor, self.primal_clf, self.dual_extractor, self.dual_clf):
            m.eval()
        att, loss, loss_dict, clf_logits = self.dual_forward_pass(primal_data, dual_data, epoch, training=False)
        return att.data.cpu().reshape(-1), loss_dict, clf_logits.data.cpu()
