"""Model factory, criterion and the attention MLP building blocks.

Mirrors the public names of the reference's ``src/utils/get_model.py`` (``get_model``, ``Criterion``,
``get_preds``, ``BatchSequential``, ``MLP``) and PyG's ``InstanceNorm`` so that ``state_dict`` keys and
call signatures are unchanged; the arithmetic runs in libgsat_hip.
"""
from __future__ import annotations

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import _lib
from ._lib import call, ptr, stream
from .graph_index import call_size
from .ops import InstanceNormFn


class _SegmentCache:
    """ptr / order / int32 ids of an arbitrary (not necessarily sorted) segment-id vector."""

    def __init__(self):
        self.key, self.val = None, None

    def get(self, batch: torch.Tensor, num_seg=None):
        key = (batch.data_ptr(), batch._version, int(batch.shape[0]), num_seg)
        if key == self.key:
            return self.val
        if batch.dtype != torch.int64 or batch.dim() != 1:
            raise ValueError("batch must be an int64 vector")
        if not batch.is_cuda:
            raise _lib.GsatHipError("InstanceNorm needs ROCm tensors (no CPU fallback)")
        b = batch.contiguous()
        n, dev = int(b.shape[0]), b.device
        G = (int(b.max().item()) + 1 if n else 0) if num_seg is None else int(num_seg)   # reference: batch.max()+1
        ws_bytes = max(call_size("gsat_csr_workspace_bytes", n, G), 256)
        ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
        sptr = torch.empty(G + 1, dtype=torch.int32, device=dev)
        order = torch.empty(max(n, 1), dtype=torch.int32, device=dev)[:n]
        err = torch.zeros(1, dtype=torch.int32, device=dev)
        call("gsat_build_csr", ptr(b), None, n, G, ptr(sptr), None, ptr(order), ptr(err), ptr(ws), ws_bytes, stream())
        seg32 = torch.empty(max(n, 1), dtype=torch.int32, device=dev)[:n]
        call("gsat_narrow_i64", ptr(b), n, ptr(seg32), stream())
        self.key, self.val = key, (sptr, order, seg32, G, b)
        return self.val


class InstanceNorm(nn.Module):
    """PyG ``InstanceNorm(C)`` as the reference uses it (src/utils/get_model.py:64): eps 1e-5, no affine,
    no running stats => parameter-free, batch statistics in train and eval."""

    def __init__(self, in_channels: int, eps: float = 1e-5):
        super().__init__()
        if abs(eps - 1e-5) > 1e-12:
            raise ValueError("the HIP InstanceNorm is built for eps = 1e-5")
        self.in_channels = in_channels
        self._segs = _SegmentCache()

    def forward(self, x, batch=None):
        if batch is None:
            batch = torch.zeros(x.shape[0], dtype=torch.int64, device=x.device)
        sptr, order, seg32, G, _ = self._segs.get(batch)
        return InstanceNormFn.apply(x, sptr, order, seg32, G)

    def extra_repr(self):
        return f"{self.in_channels}"


class BatchSequential(nn.Sequential):
    """nn.Sequential that routes ``batch`` into InstanceNorm (src/utils/get_model.py:47-54)."""

    def forward(self, inputs, batch):
        for module in self._modules.values():
            inputs = module(inputs, batch) if isinstance(module, InstanceNorm) else module(inputs)
        return inputs


class MLP(BatchSequential):
    """[Linear -> InstanceNorm -> ReLU -> Dropout] x (k-1) -> Linear   (src/utils/get_model.py:57-68)."""

    def __init__(self, channels, dropout, bias=True):
        layers = []
        for i in range(1, len(channels)):
            layers.append(nn.Linear(channels[i - 1], channels[i], bias))
            if i < len(channels) - 1:
                layers += [InstanceNorm(channels[i]), nn.ReLU(), nn.Dropout(dropout)]
        super().__init__(*layers)
        self.channels = list(channels)
        self.dropout_p = float(dropout)

    def linears(self):
        return [m for m in self if isinstance(m, nn.Linear)]


class Criterion(nn.Module):
    """src/utils/get_model.py:19-34."""

    def __init__(self, num_class, multi_label):
        super().__init__()
        self.num_class = num_class
        self.multi_label = multi_label

    def forward(self, logits, targets):
        if self.num_class == 2 and not self.multi_label:
            return F.binary_cross_entropy_with_logits(logits, targets.float())
        if self.num_class > 2 and not self.multi_label:
            return F.cross_entropy(logits, targets.long())
        is_labeled = targets == targets          # NaN marks unlabeled entries
        return F.binary_cross_entropy_with_logits(logits[is_labeled], targets[is_labeled].float())


def get_preds(logits, multi_label):
    """src/utils/get_model.py:37-44."""
    if multi_label or logits.shape[1] == 1:
        return (logits.sigmoid() > 0.5).float()
    return logits.argmax(dim=1).float()


def get_model(x_dim, edge_attr_dim, num_class, multi_label, model_config, device):
    """src/utils/get_model.py:7-16."""
    from .gin import GIN
    from .pna import PNA
    name = model_config["model_name"]
    if name == "GIN":
        model = GIN(x_dim, edge_attr_dim, num_class, multi_label, model_config)
    elif name == "PNA":
        model = PNA(x_dim, edge_attr_dim, num_class, multi_label, model_config)
    elif name == "SPMotifNet":
        from .spmotif_gnn import SPMotifNet
        model = SPMotifNet(x_dim, edge_attr_dim, num_class, multi_label, model_config)
    else:
        raise ValueError("[ERROR] Unknown model name!")
    return model.to(device)
