"""PNA backbone with the reference's module layout (state_dict-compatible with src/models/pna.py)."""
from __future__ import annotations

import os

import torch.nn as nn
import torch.nn.functional as F

from .conv_layers import PNAConvSimple
from .encoders import AtomEncoder, BatchNorm, BondEncoder, Linear
from .graph_index import get_index
from .ops import segment_pool


class PNA(nn.Module):
    def __init__(self, x_dim, edge_attr_dim, num_class, multi_label, model_config):
        super().__init__()
        hidden = model_config["hidden_size"]
        self.n_layers = model_config["n_layers"]
        self.dropout_p = model_config["dropout_p"]
        self.edge_attr_dim = edge_attr_dim
        use_edge_attr = model_config.get("use_edge_attr", True)
        if model_config.get("atom_encoder", False):
            self.node_encoder = AtomEncoder(emb_dim=hidden)
            if edge_attr_dim != 0 and use_edge_attr:
                self.edge_encoder = BondEncoder(emb_dim=hidden)
        else:
            self.node_encoder = Linear(x_dim, hidden)
            if edge_attr_dim != 0 and use_edge_attr:
                self.edge_encoder = Linear(edge_attr_dim, hidden)
        aggregators = model_config["aggregators"]
        scalers = ["identity", "amplification", "attenuation"] if model_config["scalers"] else ["identity"]
        deg = model_config["deg"]
        if use_edge_attr:
            in_channels = hidden * 2 if edge_attr_dim == 0 else hidden * 3
        else:
            in_channels = hidden * 2
        self.convs = nn.ModuleList()
        self.batch_norms = nn.ModuleList()
        for _ in range(self.n_layers):
            self.convs.append(PNAConvSimple(in_channels=in_channels, out_channels=hidden, aggregators=aggregators,
                                            scalers=scalers, deg=deg, post_layers=1))
            self.batch_norms.append(BatchNorm(hidden))
        self.fc_out = nn.Sequential(nn.Linear(hidden, hidden // 2), nn.ReLU(),
                                    nn.Linear(hidden // 2, hidden // 4), nn.ReLU(),
                                    nn.Linear(hidden // 4, 1 if num_class == 2 and not multi_label else num_class))

    def pool(self, x, batch, index):
        return segment_pool(x, index.graphs(batch), mean=True)           # global_mean_pool

    def get_emb(self, x, edge_index, batch, edge_attr, edge_atten=None):
        index = get_index(edge_index, x.shape[0])
        if batch is not None:
            index.graphs(batch)        # register the batch vector first: one status read-back then validates ids AND order
        x = self.node_encoder(x)
        if edge_attr is not None:
            edge_attr = self.edge_encoder(edge_attr)
        for conv, batch_norm in zip(self.convs, self.batch_norms):
            # h = relu(BN(conv)); x = h + x; x = dropout(x)  (src/models/pna.py:57-59) -- one fused pass after the statistics
            # (the conv hands x back as an identity output of its autograd node: the residual's gradient is then added inside the
            #  aggregation backward, not by a separate [N,H] add per layer)
            if os.environ.get("GSAT_PNA_RESIDUAL_FOLD", "1") != "0":
                h, x_res = conv(x, edge_index, edge_attr, edge_atten=edge_atten, index=index, with_residual_input=True)
            else:
                h, x_res = conv(x, edge_index, edge_attr, edge_atten=edge_atten, index=index), x
            x = batch_norm(h, fused_relu=True, residual=x_res, dropout_p=self.dropout_p)
        return x

    def forward(self, x, edge_index, batch, edge_attr, edge_atten=None):
        emb = self.get_emb(x, edge_index, batch, edge_attr, edge_atten=edge_atten)
        return self.fc_out(self.pool(emb, batch, get_index(edge_index, emb.shape[0])))

    def get_pred_from_emb(self, emb, batch, edge_index=None):
        if edge_index is not None:
            return self.fc_out(self.pool(emb, batch, get_index(edge_index, emb.shape[0])))
        from .get_model import _SegmentCache
        from .ops import SegmentPool
        cache = getattr(self, "_pool_cache", None) or _SegmentCache()
        object.__setattr__(self, "_pool_cache", cache)
        sptr, _, _, G, _ = cache.get(batch)
        return self.fc_out(SegmentPool.apply(emb, sptr, G, True))
