"""Small helpers around the backbones: ogb categorical encoders, PyG BatchNorm wrapper, global pools."""
from __future__ import annotations

import torch
import torch.nn as nn

from .graph_index import get_index
from .ops import segment_pool

# [3P] ogb 1.3.2 get_atom_feature_dims() / get_bond_feature_dims()
ATOM_FEATURE_DIMS = [119, 5, 12, 12, 10, 6, 6, 2, 2]
BOND_FEATURE_DIMS = [5, 6, 2]


class AtomEncoder(nn.Module):
    """sum of 9 per-column embeddings; keys ``atom_embedding_list.{i}.weight`` as ogb's (src/models/gin.py:23)."""

    def __init__(self, emb_dim):
        super().__init__()
        self.atom_embedding_list = nn.ModuleList()
        for dim in ATOM_FEATURE_DIMS:
            emb = nn.Embedding(dim, emb_dim)
            nn.init.xavier_uniform_(emb.weight.data)
            self.atom_embedding_list.append(emb)

    def forward(self, x):
        out = self.atom_embedding_list[0](x[:, 0])
        for i in range(1, x.shape[1]):
            out = out + self.atom_embedding_list[i](x[:, i])
        return out


class BondEncoder(nn.Module):
    def __init__(self, emb_dim):
        super().__init__()
        self.bond_embedding_list = nn.ModuleList()
        for dim in BOND_FEATURE_DIMS:
            emb = nn.Embedding(dim, emb_dim)
            nn.init.xavier_uniform_(emb.weight.data)
            self.bond_embedding_list.append(emb)

    def forward(self, edge_attr):
        out = self.bond_embedding_list[0](edge_attr[:, 0])
        for i in range(1, edge_attr.shape[1]):
            out = out + self.bond_embedding_list[i](edge_attr[:, i])
        return out


class BatchNorm(nn.Module):
    """PyG's BatchNorm wrapper: keeps the BatchNorm1d as ``.module`` (keys ``batch_norms.{i}.module.*``)."""

    def __init__(self, in_channels):
        super().__init__()
        self.module = nn.BatchNorm1d(in_channels)

    def forward(self, x):
        return self.module(x)


def _segments(batch, edge_index=None, num_nodes=None):
    ix = get_index(edge_index, num_nodes) if edge_index is not None else None
    if ix is None:
        raise ValueError("pooling needs the batch's edge_index to find the cached index")
    return ix.graphs(batch)


def global_add_pool(x, batch, segments=None):
    return segment_pool(x, segments, mean=False)


def global_mean_pool(x, batch, segments=None):
    return segment_pool(x, segments, mean=True)
