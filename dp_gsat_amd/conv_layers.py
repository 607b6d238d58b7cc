"""Conv operators with the reference's call signature ``conv(x, edge_index, edge_attr=None, edge_atten=None)``.

Message passing (gather, mask, reduce, and its backward) runs in the HIP kernels behind
:mod:`dp_gsat_amd.ops`; large dense node updates run on the split-bf16 MFMA GEMM of the same library, small ones on hipBLASLt.
"""
from __future__ import annotations

from typing import Dict, List

import torch
import torch.nn as nn

from .graph_index import get_index
from .ops import MaskedSumAggregate, masked_sum_aggregate, pna_aggregate, pna_conv


def _index_of(edge_index, x, index):
    return index if index is not None else get_index(edge_index, x.shape[0])


class GINConv(nn.Module):
    """out = nn((1+eps) x_i + sum_j a_ji x_j)   (src/models/conv_layers.py:14-34; eps is a 0 buffer)."""

    def __init__(self, nn_: nn.Module, eps: float = 0.0):
        super().__init__()
        self.nn = nn_
        self.register_buffer("eps", torch.Tensor([eps]))
        self._eps0 = float(eps)

    def forward(self, x, edge_index, edge_attr=None, edge_atten=None, size=None, index=None):
        out = masked_sum_aggregate(x, _index_of(edge_index, x, index), edge_atten, None, self._eps0)
        return self.nn(out)


class GINEConv(nn.Module):
    """out = nn((1+eps) x_i + sum_j a_ji relu(x_j + lin(e_ji)))   (src/models/conv_layers.py:37-66)."""

    def __init__(self, nn_: nn.Module, eps: float = 0.0, edge_dim=None, in_channels=None):
        super().__init__()
        self.nn = nn_
        self.register_buffer("eps", torch.Tensor([eps]))
        self._eps0 = float(eps)
        if in_channels is None:
            in_channels = next(m for m in nn_.modules() if isinstance(m, nn.Linear)).in_features
        from .encoders import Linear
        self.lin = Linear(edge_dim, in_channels) if edge_dim is not None else None

    def forward(self, x, edge_index, edge_attr=None, edge_atten=None, size=None, index=None):
        if self.lin is None and x.size(-1) != edge_attr.size(-1):
            raise ValueError("Node and edge feature dimensionalities do not match. Consider setting the "
                             "'edge_dim' attribute of 'GINEConv'")
        edge_emb = self.lin(edge_attr) if self.lin is not None else edge_attr
        out = masked_sum_aggregate(x, _index_of(edge_index, x, index), edge_atten, edge_emb, self._eps0)
        return self.nn(out)


class PNAConvSimple(nn.Module):
    """post_nn(scalers x aggregators of a_ji [x_i || x_j (|| e_ji)])   (src/models/conv_layers.py:96-191)."""

    def __init__(self, in_channels: int, out_channels: int, aggregators: List[str], scalers: List[str],
                 deg: torch.Tensor, post_layers: int = 1):
        super().__init__()
        self.in_channels, self.out_channels = in_channels, out_channels
        self.aggregators, self.scalers = list(aggregators), list(scalers)
        self.F_in, self.F_out = in_channels, out_channels
        deg = deg.to(torch.float)
        # statistics of the histogram VALUES, as the reference computes them (conv_layers.py:141-146)
        self.avg_deg: Dict[str, float] = {"lin": deg.mean().item(), "log": (deg + 1).log().mean().item(),
                                          "exp": deg.exp().mean().item()}
        from .encoders import Linear
        modules = [Linear(len(aggregators) * len(scalers) * self.F_in, self.F_out)]
        for _ in range(post_layers - 1):
            modules += [nn.ReLU(), Linear(self.F_out, self.F_out)]
        self.post_nn = nn.Sequential(*modules)

    def forward(self, x, edge_index, edge_attr=None, edge_atten=None, index=None, with_residual_input: bool = False):
        """``with_residual_input``: return ``(out, x_id)`` where ``x_id`` is ``x`` as an identity output of the aggregation's autograd node: a
        layer that also adds ``x`` as its residual (src/models/pna.py:57-59) uses ``x_id`` there, and that path's gradient is then added
        inside the aggregation backward instead of by a separate [N,H] add per layer."""
        index = _index_of(edge_index, x, index)
        x_id = x
        if len(self.post_nn) == 1:           # one Linear behind the aggregation: one autograd node on the compact aggregate when possible
            lin = self.post_nn[0]
            out = pna_conv(x, index, edge_atten, edge_attr, self.aggregators, self.scalers, self.avg_deg, lin.weight, lin.bias, with_residual_input)
            if out is not None:
                return out
        agg = pna_aggregate(x, index, edge_atten, edge_attr, self.aggregators, self.scalers, self.avg_deg, with_residual_input and x.is_cuda)
        if isinstance(agg, tuple):
            agg, x_id = agg
        if agg.shape[1] != self.post_nn[0].in_features:
            raise ValueError(f"PNAConvSimple was built for F_in={self.F_in} but the message is {agg.shape[1]} wide")
        out = self.post_nn(agg)
        return (out, x_id) if with_residual_input else out

    def __repr__(self):
        return f"{self.__class__.__name__}({self.in_channels}, {self.out_channels})"


class LEConv(nn.Module):
    """out_i = sum_j a_ji w_ji (lin1(x)_j - lin2(x)_i) + lin3(x)_i   (src/models/conv_layers.py:69-92; [3P] PyG LEConv:
    lin1 / lin3 with bias, lin2 without).  The message sum splits into a masked gather-sum of lin1(x) and a per-row weight
    total, so it reuses the GIN aggregation kernels; no [E,H] message is materialised."""

    def __init__(self, in_channels: int, out_channels: int, bias: bool = True):
        super().__init__()
        from .encoders import Linear
        self.in_channels, self.out_channels = in_channels, out_channels
        self.lin1 = Linear(in_channels, out_channels, bias=bias)
        self.lin2 = Linear(in_channels, out_channels, bias=False)
        self.lin3 = Linear(in_channels, out_channels, bias=bias)

    def forward(self, x, edge_index, edge_weight=None, edge_atten=None, index=None):
        index = _index_of(edge_index, x, index)
        a, b = self.lin1(x), self.lin2(x)
        w = None
        if edge_weight is not None:
            w = edge_weight.view(-1, 1)
        if edge_atten is not None:
            w = edge_atten.view(-1, 1) if w is None else w * edge_atten.view(-1, 1)
        agg = MaskedSumAggregate.apply(a, w, None, index, 0.0)                       # sum_j w_ji a_j
        ones = torch.ones(x.shape[0], 4, dtype=x.dtype, device=x.device)
        wsum = MaskedSumAggregate.apply(ones, w, None, index, 0.0)[:, :1]            # sum_j w_ji  (in-degree if w is None)
        return agg - b * wsum + self.lin3(x)
