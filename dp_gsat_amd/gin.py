"""GIN backbone with the reference's module layout (state_dict-compatible with src/models/gin.py)."""
from __future__ import annotations

import torch.nn as nn
import torch.nn.functional as F

from .conv_layers import GINConv, GINEConv
from .encoders import AtomEncoder, BatchNorm1d, BondEncoder, Linear
from .graph_index import get_index
from .ops import relu_dropout, segment_pool


class _UpdateMLP(nn.Sequential):
    """GIN's node update Linear -> BatchNorm1d -> ReLU -> Linear (src/models/gin.py:55-62) with the same four modules and state_dict
    keys; the ReLU is applied inside the BatchNorm kernels (forward and backward), module 2 stays as the placeholder it is upstream."""

    def forward(self, x):
        return self[3](self[1](self[0](x), fused_relu=True))


class GIN(nn.Module):
    def __init__(self, x_dim, edge_attr_dim, num_class, multi_label, model_config):
        super().__init__()
        self.n_layers = model_config["n_layers"]
        hidden = model_config["hidden_size"]
        self.edge_attr_dim = edge_attr_dim
        self.dropout_p = model_config["dropout_p"]
        self.use_edge_attr = model_config.get("use_edge_attr", True)
        with_edges = edge_attr_dim != 0 and self.use_edge_attr
        if model_config.get("atom_encoder", False):
            self.node_encoder = AtomEncoder(emb_dim=hidden)
            if with_edges:
                self.edge_encoder = BondEncoder(emb_dim=hidden)
        else:
            self.node_encoder = Linear(x_dim, hidden)
            if with_edges:
                self.edge_encoder = Linear(edge_attr_dim, hidden)
        self.convs = nn.ModuleList()
        self.relu = nn.ReLU()
        for _ in range(self.n_layers):
            if with_edges:
                self.convs.append(GINEConv(GIN.MLP(hidden, hidden), edge_dim=hidden, in_channels=hidden))
            else:
                self.convs.append(GINConv(GIN.MLP(hidden, hidden)))
        self.fc_out = nn.Sequential(nn.Linear(hidden, 1 if num_class == 2 and not multi_label else num_class))

    @staticmethod
    def MLP(in_channels: int, out_channels: int):
        return _UpdateMLP(Linear(in_channels, out_channels), BatchNorm1d(out_channels),
                          nn.ReLU(inplace=True), Linear(out_channels, out_channels))

    def pool(self, x, batch, index):
        return segment_pool(x, index.graphs(batch), mean=False)          # global_add_pool

    def get_emb(self, x, edge_index, batch, edge_attr=None, edge_atten=None):
        index = get_index(edge_index, x.shape[0])
        if batch is not None:
            index.graphs(batch)        # register the batch vector first: one status read-back then validates ids AND order
        x = self.node_encoder(x)
        if edge_attr is not None and self.use_edge_attr:
            edge_attr = self.edge_encoder(edge_attr)
        for conv in self.convs:
            x = conv(x, edge_index, edge_attr=edge_attr, edge_atten=edge_atten, index=index)
            x = relu_dropout(x, self.dropout_p, self.training)          # relu + dropout, one launch each way (src/models/gin.py:50-51)
        return x

    def forward(self, x, edge_index, batch, edge_attr=None, edge_atten=None):
        emb = self.get_emb(x, edge_index, batch, edge_attr=edge_attr, edge_atten=edge_atten)
        return self.fc_out(self.pool(emb, batch, get_index(edge_index, emb.shape[0])))

    def get_graph_emb(self, x, edge_index, batch, edge_attr=None, edge_atten=None):
        emb = self.get_emb(x, edge_index, batch, edge_attr, edge_atten)
        return self.pool(emb, batch, get_index(edge_index, emb.shape[0]))

    def get_pred_from_emb(self, emb, batch, edge_index=None):
        return self.fc_out(self._pool_from_batch(emb, batch, edge_index))

    def _pool_from_batch(self, emb, batch, edge_index):
        from .get_model import _SegmentCache
        if edge_index is not None:
            return self.pool(emb, batch, get_index(edge_index, emb.shape[0]))
        cache = getattr(self, "_pool_cache", None) or _SegmentCache()
        object.__setattr__(self, "_pool_cache", cache)
        sptr, _, _, G, _ = cache.get(batch)
        from .ops import SegmentPool
        return SegmentPool.apply(emb, sptr, G, False)
