"""SPMotifNet backbone (LEConv model of DIR) with the reference's module layout (src/models/spmotif_gnn.py)."""
from __future__ import annotations

import torch
import torch.nn as nn

from .conv_layers import LEConv
from .encoders import Linear
from .graph_index import get_index
from .ops import segment_pool


class SPMotifNet(nn.Module):
    def __init__(self, x_dim, edge_attr_dim, num_class, multi_label, model_config):
        super().__init__()
        self.n_layers = model_config["n_layers"]
        hidden = model_config["hidden_size"]
        self.edge_attr_dim = edge_attr_dim
        self.node_emb = Linear(x_dim, hidden)
        self.convs = nn.ModuleList(LEConv(in_channels=hidden, out_channels=hidden) for _ in range(self.n_layers))
        self.relus = nn.ModuleList(nn.ReLU() for _ in range(self.n_layers))
        self.fc_out = nn.Sequential(nn.Linear(hidden, 2 * hidden), nn.ReLU(), nn.Linear(2 * hidden, num_class))
        self.conf_mlp = nn.Sequential(nn.Linear(hidden, 2 * hidden), nn.ReLU(), nn.Linear(2 * hidden, 3))
        self.cq = nn.Linear(3, 3)
        self.conf_fw = nn.Sequential(self.conf_mlp, self.cq)

    def pool(self, x, batch, edge_index=None):
        if edge_index is not None:
            return segment_pool(x, get_index(edge_index, x.shape[0]).graphs(batch), mean=True)     # global_mean_pool
        from .get_model import _SegmentCache
        from .ops import SegmentPool
        cache = getattr(self, "_pool_cache", None) or _SegmentCache()
        object.__setattr__(self, "_pool_cache", cache)
        sptr, _, _, G, _ = cache.get(batch)
        return SegmentPool.apply(x, sptr, G, True)

    def get_node_reps(self, x, edge_index, edge_attr, batch, edge_atten):
        index = get_index(edge_index, x.shape[0])
        if batch is not None:
            index.graphs(batch)        # register the batch vector first: one status read-back then validates ids AND order
        x = self.node_emb(x)
        for conv, relu in zip(self.convs, self.relus):
            x = relu(conv(x=x, edge_index=edge_index, edge_weight=edge_attr, edge_atten=edge_atten, index=index))
        return x

    def forward(self, x, edge_index, batch, edge_attr, edge_atten=None):
        node_x = self.get_node_reps(x, edge_index, edge_attr, batch, edge_atten=edge_atten)
        return self.get_causal_pred(self.pool(node_x, batch, edge_index))

    def get_emb(self, x, edge_index, batch, edge_attr, edge_atten=None):
        return self.get_node_reps(x, edge_index, edge_attr, batch, edge_atten=edge_atten)

    def get_pred_from_emb(self, emb, batch, edge_index=None):
        return self.fc_out(self.pool(emb, batch, edge_index))

    def get_graph_rep(self, x, edge_index, edge_attr, batch, edge_atten):
        return self.pool(self.get_node_reps(x, edge_index, edge_attr, batch, edge_atten=edge_atten), batch, edge_index)

    def get_causal_pred(self, causal_graph_x):
        return self.fc_out(causal_graph_x)

    def get_conf_pred(self, conf_graph_x):
        return self.conf_fw(conf_graph_x)

    def get_comb_pred(self, causal_graph_x, conf_graph_x):
        return torch.sigmoid(self.conf_mlp(conf_graph_x).detach()) * self.fc_out(causal_graph_x)

    def reset_parameters(self):
        with torch.no_grad():
            for param in self.parameters():
                param.uniform_(-1.0, 1.0)
