"""Small helpers around the backbones: ogb categorical encoders, PyG BatchNorm wrapper, global pools."""
from __future__ import annotations

import torch
import torch.nn as nn

from .graph_index import get_index
from .ops import BatchNormFn, EmbeddingSum, SyncBatchNormFn, linear, segment_pool

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


class Linear(nn.Linear):
    """nn.Linear (same parameters / keys) for node- and edge-row inputs: forward and dx stay library GEMMs, the weight
    gradient (a reduction over ~1e4..1e7 rows into a tiny matrix) uses the hand-written split-K MFMA GEMM."""

    def forward(self, x):
        return linear(x, self.weight, self.bias)


class BatchNorm1d(nn.BatchNorm1d):
    """nn.BatchNorm1d (same parameters / buffers / state_dict keys) whose arithmetic runs in libgsat_hip on ROCm tensors
    with 2-D fp32 input; anything else (CPU tensors in the host-protocol tests, odd shapes) uses torch's implementation.
    ``fused_relu`` applies ReLU inside the kernel (PNA).  ``sync_group`` (set by dp_gsat_amd.dist.sync_batchnorm) makes the
    training statistics span every rank of that process group, still on the HIP kernels."""
    sync_group = None          # None = per-rank statistics; a torch.distributed group (or True = default group) = global statistics

    def forward(self, x, fused_relu: bool = False, residual=None, dropout_p: float = 0.0):
        """``residual`` / ``dropout_p``: y = dropout_p(act(BN(x)) + residual) in the same kernels (the PNA layer tail)."""
        shape_ok = (x.is_cuda and x.dim() == 2 and x.dtype == torch.float32 and self.affine and x.shape[1] % 4 == 0
                    and (self.training or self.track_running_stats) and self.momentum is not None)
        hip_ok = shape_ok and x.shape[0] > 0
        p = float(dropout_p) if self.training else 0.0
        sync = None
        if (self.training or not self.track_running_stats) and self.sync_group is not None:
            import torch.distributed as dist
            if dist.is_available() and dist.is_initialized():
                group = None if self.sync_group is True else self.sync_group
                if dist.get_world_size(group) > 1:
                    # global statistics were asked for: EVERY rank must enter the collectives of SyncBatchNormFn -- also a rank whose shard
                    # has no rows (the kernels take N == 0: zero sums, no apply launch) -- and a shape the HIP path cannot take must not
                    # silently fall back to per-rank statistics
                    if not shape_ok:
                        raise ValueError("BatchNorm1d with sync_group needs a 2-D fp32 ROCm input with affine parameters and a channel count "
                                         f"that is a multiple of 4 (got {tuple(x.shape)}, {x.dtype}, device {x.device})")
                    sync = (group,)
        if not hip_ok and sync is None:
            y = super().forward(x)
            y = torch.relu(y) if fused_relu else y
            if residual is not None:
                y = y + residual
            return torch.nn.functional.dropout(y, p, training=True) if p > 0.0 else y
        training = self.training or not self.track_running_stats
        if self.training and self.track_running_stats and self.num_batches_tracked is not None:
            self.num_batches_tracked.add_(1)
        seed, seed_dev = 0, None
        if p > 0.0:
            from .graph_index import sync_free
            from .ops import device_seed, new_seed
            if sync_free():        # no host round trip inside a captured step: the seed word lives on the device
                seed_dev = device_seed(x.device)
            else:
                seed = new_seed()
        if sync is not None:
            return SyncBatchNormFn.apply(x, self.weight, self.bias, self.running_mean if self.track_running_stats else None,
                                         self.running_var if self.track_running_stats else None, self.momentum, self.eps, fused_relu,
                                         residual, p, seed, seed_dev, sync[0])
        return BatchNormFn.apply(x, self.weight, self.bias, self.running_mean if self.track_running_stats else None,
                                 self.running_var if self.track_running_stats else None, training, self.momentum, self.eps, fused_relu,
                                 residual, p, seed, seed_dev)


class BatchNorm(nn.Module):
    """PyG's BatchNorm wrapper: keeps the BatchNorm1d as ``.module`` (keys ``batch_norms.{i}.module.*``)."""

    def __init__(self, in_channels):
        super().__init__()
        self.module = BatchNorm1d(in_channels)

    def forward(self, x, fused_relu: bool = False, residual=None, dropout_p: float = 0.0):
        return self.module(x, fused_relu, residual, dropout_p)


def _segments(batch, edge_index=None, num_nodes=None):
    ix = get_index(edge_index, num_nodes) if edge_index is not None else None
    if ix is None:
        raise ValueError("pooling needs the batch's edge_index to find the cached index")
    return ix.graphs(batch)


def global_add_pool(x, batch, segments=None):
    return segment_pool(x, segments, mean=False)


def global_mean_pool(x, batch, segments=None):
    return segment_pool(x, segments, mean=True)
