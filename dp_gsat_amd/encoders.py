"""Small helpers around the backbones: ogb categorical encoders, PyG BatchNorm wrapper, global pools."""
from __future__ import annotations

import torch
import torch.nn as nn

from .graph_index import get_index
from .ops import EmbeddingSum, segment_pool

# [3P] ogb 1.3.2 get_atom_feature_dims() / get_bond_feature_dims()
ATOM_FEATURE_DIMS = [119, 5, 12, 12, 10, 6, 6, 2, 2]
BOND_FEATURE_DIMS = [5, 6, 2]


class _CategoricalEncoder(nn.Module):
    """Sum of per-column embeddings.  The ``nn.Embedding`` tables keep ogb's names (state_dict keys
    ``atom_embedding_list.{i}.weight`` / ``bond_embedding_list.{i}.weight``); the lookup-sum and its backward run in
    libgsat_hip (gather forward, one-hot MFMA GEMM backward)."""
    _list_name = ""
    _dims = ()

    def __init__(self, emb_dim):
        super().__init__()
        tables = nn.ModuleList()
        for dim in self._dims:
            emb = nn.Embedding(dim, emb_dim)
            nn.init.xavier_uniform_(emb.weight.data)
            tables.append(emb)
        setattr(self, self._list_name, tables)
        self._cache = {}

    def forward(self, x):
        tables = getattr(self, self._list_name)
        ncol = x.shape[1]
        W_all = torch.cat([tables[i].weight for i in range(ncol)], dim=0)
        return EmbeddingSum.apply(x, W_all, [tables[i].num_embeddings for i in range(ncol)], self._cache)


class AtomEncoder(_CategoricalEncoder):
    """ogb AtomEncoder (src/models/gin.py:23)."""
    _list_name, _dims = "atom_embedding_list", tuple(ATOM_FEATURE_DIMS)


class BondEncoder(_CategoricalEncoder):
    """ogb BondEncoder (src/models/gin.py:25)."""
    _list_name, _dims = "bond_embedding_list", tuple(BOND_FEATURE_DIMS)


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
