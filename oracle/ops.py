"""Floating-point operators of the GSAT hot path in plain PyTorch (ORACLE, test-only).

Everything is differentiable through torch autograd; min/max use an explicit
arg-gather so that the backward follows torch-scatter's arg routing, not
``scatter_reduce``'s tie-splitting.  dtype follows the inputs (fp32 for parity
tests, fp64 to measure fp32 error).
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional, Sequence

import torch
import torch.nn.functional as F

Tensor = torch.Tensor


# ----------------------------------------------------------------------------------------------
# [3P] torch_scatter.scatter(src, index, 0, None, dim_size, reduce) -- SURVEY App. B
# call sites src/models/conv_layers.py:193-206
# ----------------------------------------------------------------------------------------------
def scatter_sum(src: Tensor, index: Tensor, dim_size: int) -> Tensor:
    out = src.new_zeros((dim_size,) + tuple(src.shape[1:]))
    return out.index_add(0, index, src)


def seg_count(index: Tensor, dim_size: int, dtype) -> Tensor:
    return torch.zeros(dim_size, dtype=dtype, device=index.device).index_add(
        0, index, torch.ones(index.shape[0], dtype=dtype, device=index.device))


def scatter_mean(src: Tensor, index: Tensor, dim_size: int) -> Tensor:
    cnt = seg_count(index, dim_size, src.dtype).clamp_(min=1)          # count.clamp_(1)
    return scatter_sum(src, index, dim_size) / cnt.view(-1, *([1] * (src.dim() - 1)))


def _scatter_arg(src: Tensor, index: Tensor, dim_size: int, reduce: str) -> Tensor:
    """arg[i,c] = FIRST edge slot e (in edge order) attaining the extremum of row i, or E if the
    row is empty (torch-scatter CPU: strict </> update keeps the first occurrence; empty rows
    keep arg == dim and are zero-filled)."""
    E = src.shape[0]
    with torch.no_grad():
        idx = index.view(-1, 1).expand_as(src)
        init = float("inf") if reduce == "min" else float("-inf")
        ext = torch.full((dim_size, src.shape[1]), init, dtype=src.dtype)
        ext = ext.scatter_reduce(0, idx, src, "amin" if reduce == "min" else "amax", include_self=True)
        hit = src == ext.index_select(0, index)
        pos = torch.arange(E).view(-1, 1).expand_as(src)
        cand = torch.where(hit, pos, torch.full_like(pos, E))
        arg = torch.full((dim_size, src.shape[1]), E, dtype=torch.long)
        arg = arg.scatter_reduce(0, idx, cand, "amin", include_self=True)
    return arg


def scatter_minmax(src: Tensor, index: Tensor, dim_size: int, reduce: str):
    """Value result; rows with no entries = 0; gradient goes to the arg element only."""
    E = src.shape[0]
    arg = _scatter_arg(src, index, dim_size, reduce)
    padded = torch.cat([src, src.new_zeros(1, src.shape[1])], dim=0)
    return padded.gather(0, arg), arg


def scatter(src: Tensor, index: Tensor, dim_size: int, reduce: str) -> Tensor:
    if reduce in ("sum", "add"):
        return scatter_sum(src, index, dim_size)
    if reduce == "mean":
        return scatter_mean(src, index, dim_size)
    if reduce in ("min", "max"):
        return scatter_minmax(src, index, dim_size, reduce)[0]
    raise ValueError(reduce)


# ----------------------------------------------------------------------------------------------
# [3P] torch_geometric.nn.InstanceNorm(C) with batch (src/utils/get_model.py:50-51,64)
# eps=1e-5, affine=False, track_running_stats=False => batch statistics in train AND eval.
# ----------------------------------------------------------------------------------------------
def instance_norm(x: Tensor, seg: Tensor, num_seg: int, eps: float = 1e-5) -> Tensor:
    norm = seg_count(seg, num_seg, x.dtype).clamp_(min=1).view(-1, 1)
    mean = scatter_sum(x, seg, num_seg) / norm
    xc = x - mean.index_select(0, seg)
    var = scatter_sum(xc * xc, seg, num_seg) / norm                     # biased
    return xc / (var + eps).sqrt().index_select(0, seg)


# ----------------------------------------------------------------------------------------------
# MLP / BatchSequential (src/utils/get_model.py:47-68)
# weights: [(W_k, b_k)] ; hidden layers: Linear -> InstanceNorm -> ReLU -> Dropout(p)
# dropout is applied with EXPLICIT keep-masks (float 0/1, already the Bernoulli draw); the
# 1/(1-p) scaling is nn.Dropout's.
# ----------------------------------------------------------------------------------------------
def mlp_forward(x: Tensor, seg: Tensor, num_seg: int, weights: Sequence, p: float = 0.0,
                masks: Optional[Sequence[Optional[Tensor]]] = None, training: bool = False,
                return_hidden: bool = False):
    h = x
    hidden = []
    n = len(weights)
    for k, (W, b) in enumerate(weights):
        h = F.linear(h, W, b)                                           # :61
        if k < n - 1:
            h = instance_norm(h, seg, num_seg)                          # :64
            h = torch.relu(h)                                           # :65
            if training and p > 0.0:                                    # :66
                m = masks[k] if masks is not None else None
                if m is None:
                    raise ValueError("oracle needs explicit dropout masks in training mode")
                h = h * m / (1.0 - p)
            hidden.append(h)
    return (h, hidden) if return_hidden else h


def extractor_forward(emb: Tensor, edge_index: Tensor, batch: Tensor, num_graphs: int, weights,
                      learn_edge_att: bool, p: float = 0.0, masks=None, training: bool = False) -> Tensor:
    """ExtractorMLP.forward (example/gsat.py:131-139; src/run_gsat.py:909-927).
    NOTE ``col, row = edge_index``: col = edge_index[0] = source, segments = batch[col]."""
    if learn_edge_att:
        col, row = edge_index[0], edge_index[1]
        f12 = torch.cat([emb[col], emb[row]], dim=-1)
        return mlp_forward(f12, batch[col], num_graphs, weights, p, masks, training)
    return mlp_forward(emb, batch, num_graphs, weights, p, masks, training)


# ----------------------------------------------------------------------------------------------
# samplers (example/gsat.py:94-103; src/run_gsat.py:182-187, 877-885)
# ----------------------------------------------------------------------------------------------
def concrete_sample(logits: Tensor, u: Optional[Tensor], training: bool, temp: float = 1.0) -> Tensor:
    if training:
        noise = torch.log(u) - torch.log(1.0 - u)
        return ((logits + noise) / temp).sigmoid()
    return logits.sigmoid()


def gumbel_sigmoid(logits: Tensor, U: Tensor, tau: float = 1.0, eps: float = 1e-10) -> Tensor:
    g = -torch.log(-torch.log(U + eps) + eps)
    return torch.sigmoid((logits + g) / tau)


def lift_node_att_to_edge_att(node_att: Tensor, edge_index: Tensor) -> Tensor:
    """example/gsat.py:112-117."""
    return node_att[edge_index[0]] * node_att[edge_index[1]]


def symmetrise(att: Tensor, rev: Optional[Tensor]) -> Tensor:
    """example/gsat.py:79-85: (att + att[rev]) / 2 when the edge set is symmetric, else att."""
    if rev is None:
        return att
    return (att + att[rev]) / 2


def get_r(decay_interval: int, decay_r: float, current_epoch: int, init_r: float = 0.9, final_r: float = 0.5) -> float:
    """example/gsat.py:105-110."""
    r = init_r - current_epoch // decay_interval * decay_r
    if r < final_r:
        r = final_r
    return r


def info_loss(att: Tensor, r) -> Tensor:
    """example/gsat.py:31 ; src/run_gsat.py:127,132 (r may be a per-edge tensor prior)."""
    return (att * torch.log(att / r + 1e-6) + (1 - att) * torch.log((1 - att) / (1 - r + 1e-6) + 1e-6)).mean()


# ----------------------------------------------------------------------------------------------
# conv operators (src/models/conv_layers.py)
# ----------------------------------------------------------------------------------------------
def gin_aggregate(x: Tensor, edge_index: Tensor, edge_atten: Optional[Tensor], eps: float = 0.0) -> Tensor:
    """GINConv.forward minus self.nn (src/models/conv_layers.py:14-34): sum_j a_e x_j + (1+eps) x_i."""
    msg = x[edge_index[0]]
    if edge_atten is not None:
        msg = msg * edge_atten
    out = scatter_sum(msg, edge_index[1], x.shape[0])
    return out + (1 + eps) * x


def gine_aggregate(x: Tensor, edge_index: Tensor, edge_emb: Tensor, edge_atten: Optional[Tensor], eps: float = 0.0) -> Tensor:
    """GINEConv.forward minus self.nn (src/models/conv_layers.py:37-66); ``edge_emb`` = self.lin(edge_attr)."""
    m = (x[edge_index[0]] + edge_emb).relu()
    if edge_atten is not None:
        m = m * edge_atten
    out = scatter_sum(m, edge_index[1], x.shape[0])
    return out + (1 + eps) * x


def pna_avg_deg(deg_hist: Tensor) -> Dict[str, float]:
    """src/models/conv_layers.py:140-146 -- statistics of the histogram VALUES (quirk R)."""
    deg = deg_hist.to(torch.float)
    return {"lin": deg.mean().item(), "log": (deg + 1).log().mean().item(), "exp": deg.exp().mean().item()}


def pna_aggregate(x: Tensor, edge_index: Tensor, edge_atten: Optional[Tensor], aggregators: List[str],
                  scalers: List[str], avg_deg: Dict[str, float], edge_emb: Optional[Tensor] = None) -> Tensor:
    """PNAConvSimple.message + aggregate (src/models/conv_layers.py:166-185, 193-259)."""
    src, dst = edge_index[0], edge_index[1]
    N = x.shape[0]
    parts = [x[dst], x[src]] + ([edge_emb] if edge_emb is not None else [])   # [x_i, x_j(, e)]  :168-171
    m = torch.cat(parts, dim=-1)
    if edge_atten is not None:
        m = m * edge_atten
    outs = []
    for a in aggregators:                                                   # :179
        if a == "sum":
            outs.append(scatter_sum(m, dst, N))
        elif a == "mean":
            outs.append(scatter_mean(m, dst, N))
        elif a in ("min", "max"):
            outs.append(scatter_minmax(m, dst, N, a)[0])
        elif a in ("var", "std"):
            mean = scatter_mean(m, dst, N)
            mean_sq = scatter_mean(m * m, dst, N)
            var = mean_sq - mean * mean                                     # :209-212
            outs.append(var if a == "var" else torch.sqrt(torch.relu(var) + 1e-5))   # :215-216
        else:
            raise ValueError(a)
    out = torch.cat(outs, dim=-1)
    deg = seg_count(dst, N, m.dtype).view(-1, 1)                            # :183 (unweighted)
    scaled = []
    for s in scalers:                                                       # :184
        if s == "identity":
            scaled.append(out)
        elif s == "amplification":
            scaled.append(out * (torch.log(deg + 1) / avg_deg["log"]))
        elif s == "attenuation":
            sc = avg_deg["log"] / torch.log(deg + 1)
            sc = torch.where(deg == 0, torch.ones_like(sc), sc)
            scaled.append(out * sc)
        elif s == "linear":
            scaled.append(out * (deg / avg_deg["lin"]))
        elif s == "inverse_linear":
            sc = avg_deg["lin"] / deg
            sc = torch.where(deg == 0, torch.ones_like(sc), sc)
            scaled.append(out * sc)
        else:
            raise ValueError(s)
    return torch.cat(scaled, dim=-1)


# ----------------------------------------------------------------------------------------------
# [3P] global pools (src/models/gin.py:34, src/models/pna.py:47)
# ----------------------------------------------------------------------------------------------
def global_add_pool(x: Tensor, batch: Tensor, num_graphs: int) -> Tensor:
    return scatter_sum(x, batch, num_graphs)


def global_mean_pool(x: Tensor, batch: Tensor, num_graphs: int) -> Tensor:
    return scatter_mean(x, batch, num_graphs)
