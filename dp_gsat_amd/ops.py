"""torch.autograd.Function wrappers around the C ABI (include/gsat_hip.h).

Every forward/backward here is one or two kernel launches on torch's current stream; PyTorch only
provides the device memory.  None of these has a CPU path.
"""
from __future__ import annotations

import os
from typing import Optional

import torch

from ._lib import call, ptr, stream
from .graph_index import BatchIndex


def _f32c(t: Optional[torch.Tensor]) -> Optional[torch.Tensor]:
    if t is None:
        return None
    if isinstance(t, LiftedAttention):
        # inside Function.forward autograd is off: writing the tensor out here would silently cut node_att from the graph
        raise TypeError("LiftedAttention reached an autograd Function: pass dp_gsat_amd.ops.edge_tensor(att) instead")
    if t.dtype != torch.float32:
        raise TypeError("dp_gsat_amd kernels compute in fp32; got " + str(t.dtype))
    return t.contiguous()


def _flat_att(att: Optional[torch.Tensor], E: int) -> Optional[torch.Tensor]:
    """edge_atten is [E,1] in the reference (src/models/conv_layers.py:31-32); kernels take [E]."""
    if att is None:
        return None
    if isinstance(att, LiftedAttention):
        raise TypeError("LiftedAttention must be written out (att.edge()) before it enters an autograd Function")
    if att.numel() != E:
        raise ValueError(f"edge_atten has {att.numel()} entries, expected {E}")
    return _f32c(att).view(-1)


class LiftedAttention:
    """``node_att[src] * node_att[dst]`` (example/gsat.py:112-117) that has not been written out: the PNA aggregation forms the product
    at its mask load (gsat_pna_*_node_att), so a node-attention step needs no [E] tensor and no lift kernels.  Every other consumer --
    a torch function, an operator, an attribute -- gets the [E, 1] tensor of ``Lift`` (built once, on first use, inside autograd)."""

    def __init__(self, node_att: torch.Tensor, index: BatchIndex):
        if node_att.numel() != index.N:
            raise ValueError("node attention must have one entry per node")
        self.node_att, self.index, self._edge = node_att, index, None
        # the layers that form the attention inside their kernels share ONE d node_att buffer: they all read `shared_node_att`, an identity
        # output of one autograd node (_SharedAttention) that only they consume; the first of their backward nodes to run writes the
        # buffer and hands it to autograd, the others add to it inside their own kernel and return nothing (no [N,1] add launch per
        # layer); _SharedAttention.backward runs after all of them, passes the sum on to node_att (where autograd may add other consumers'
        # gradients, e.g. the info loss) and clears the state for a second backward through a retained graph.
        self._dna = None
        self._shared = None

    @property
    def shared_node_att(self) -> torch.Tensor:
        if self._shared is None:
            self._shared = _SharedAttention.apply(self.node_att, self)
        return self._shared

    def _shared_dna(self, N, device):
        """(buffer, accumulate, what to return to autograd) for one layer's d node_att."""
        if self._dna is None or os.environ.get("GSAT_NODE_ATT_SHARED_GRAD", "1") == "0":
            buf = torch.empty(N, dtype=torch.float32, device=device)
            self._dna = buf if os.environ.get("GSAT_NODE_ATT_SHARED_GRAD", "1") != "0" else None
            return buf, 0, buf
        return self._dna, 1, None

    def edge(self) -> torch.Tensor:
        if self._edge is None:
            self._edge = Lift.apply(self.node_att, self.index)
        return self._edge

    @property
    def shape(self):
        return torch.Size((self.index.E, 1))

    def numel(self):
        return self.index.E

    @classmethod
    def __torch_function__(cls, func, types, args=(), kwargs=None):
        from torch.utils._pytree import tree_map
        unwrap = lambda a: a.edge() if isinstance(a, LiftedAttention) else a
        return func(*tree_map(unwrap, args), **tree_map(unwrap, kwargs or {}))

    def __getattr__(self, name):            # anything a tensor has and this view does not
        if name.startswith("__") and name.endswith("__"):
            raise AttributeError(name)
        return getattr(self.edge(), name)

    def __repr__(self):
        return f"LiftedAttention(nodes={self.index.N}, edges={self.index.E}, materialised={self._edge is not None})"


class _SharedAttention(torch.autograd.Function):
    """Identity on node_att whose output is consumed only by the aggregation layers that form the lifted attention in their kernels
    (LiftedAttention.shared_node_att): its backward sees their in-kernel sum as ONE gradient."""

    @staticmethod
    def forward(ctx, node_att, owner):
        ctx.owner = owner
        return node_att.view_as(node_att)

    @staticmethod
    def backward(ctx, grad):
        ctx.owner._dna = None
        return grad, None


def edge_tensor(att):
    """The [E, 1] tensor of an attention value: ``att.edge()`` for the lazy lifted view, ``att`` itself otherwise.  Call it (under autograd)
    before handing attention to a custom autograd Function."""
    return att.edge() if isinstance(att, LiftedAttention) else att


def _forward_operator(name):
    def op(self, *args):
        return getattr(self.edge(), name)(*args)
    op.__name__ = name
    return op


for _name in ("__mul__", "__rmul__", "__add__", "__radd__", "__sub__", "__rsub__", "__truediv__", "__rtruediv__", "__neg__", "__pow__",
              "__getitem__", "__len__", "__iter__", "__lt__", "__le__", "__gt__", "__ge__", "__matmul__", "__float__"):
    setattr(LiftedAttention, _name, _forward_operator(_name))


class MaskedSumAggregate(torch.autograd.Function):
    """out = (1+eps)*x + sum_{e: dst=i} att_e * msg_e   (GINConv / GINEConv message passing).

    replaces: GINConv.forward/message, GINEConv.forward/message (src/models/conv_layers.py:14-66)."""

    @staticmethod
    def forward(ctx, x, att, edge_emb, index: BatchIndex, self_coef: float):
        x = _f32c(x)
        attf = _flat_att(att, index.E)
        edge_emb = _f32c(edge_emb)
        N, H = x.shape
        if N != index.N:
            raise ValueError(f"x has {N} rows but the index was built for {index.N} nodes")
        if edge_emb is not None and tuple(edge_emb.shape) != (index.E, H):
            raise ValueError("Node and edge feature dimensionalities do not match. Consider setting the "
                             "'edge_dim' attribute of 'GINEConv'")     # src/models/conv_layers.py:54-57
        out = torch.empty_like(x)
        call("gsat_aggr_sum_fwd", ptr(x), None, ptr(attf), ptr(edge_emb), ptr(index.rowptr_dst),
             ptr(index.src_by_dst), ptr(index.eid_by_dst), N, index.E, H, float(self_coef), ptr(out),
             ptr(index.long_rows[0]), ptr(index.partial(H)) if index.long_rows[0] is not None else None, stream())
        ctx.save_for_backward(x, attf, edge_emb)
        ctx.index, ctx.self_coef = index, float(self_coef)
        ctx.att_shape = None if att is None else att.shape
        return out

    @staticmethod
    def backward(ctx, dout):
        x, attf, edge_emb = ctx.saved_tensors
        index = ctx.index
        dout = _f32c(dout)
        N, H = x.shape
        need_att = attf is not None and ctx.needs_input_grad[1]
        need_ee = edge_emb is not None and ctx.needs_input_grad[2]
        dx = torch.empty_like(x)
        datt = torch.empty(index.E, dtype=torch.float32, device=x.device) if need_att else None
        dee = torch.empty_like(edge_emb) if need_ee else None
        call("gsat_aggr_sum_bwd", ptr(x), ptr(attf), ptr(edge_emb), ptr(dout), ptr(index.rowptr_src),
             ptr(index.dst_by_src), ptr(index.eid_by_src), N, index.E, H, ctx.self_coef, ptr(dx), ptr(datt), ptr(dee),
             ptr(index.long_rows[1]), ptr(index.partial(H)) if index.long_rows[1] is not None else None, stream())
        return dx, (datt.view(ctx.att_shape) if need_att else None), dee, None, None


def masked_sum_aggregate(x, index, att=None, edge_emb=None, eps: float = 0.0):
    if isinstance(att, LiftedAttention):
        att = att.edge()                 # written out here, under autograd (inside Function.forward it would be cut from the graph)
    return MaskedSumAggregate.apply(x, att, edge_emb, index, 1.0 + eps)


class SegmentPool(torch.autograd.Function):
    """global_add_pool / global_mean_pool over a sorted ``batch`` (src/models/gin.py:34, pna.py:47)."""

    @staticmethod
    def forward(ctx, x, node_ptr, num_graphs: int, mean: bool):
        x = _f32c(x)
        H = x.shape[1]
        out = torch.empty(num_graphs, H, dtype=torch.float32, device=x.device)
        call("gsat_segment_pool_fwd", ptr(x), ptr(node_ptr), num_graphs, H, int(mean), ptr(out), stream())
        ctx.save_for_backward(node_ptr)
        ctx.n, ctx.mean = x.shape[0], bool(mean)
        return out

    @staticmethod
    def backward(ctx, dout):
        (node_ptr,) = ctx.saved_tensors
        dout = _f32c(dout)
        G, H = dout.shape
        dx = torch.empty(ctx.n, H, dtype=torch.float32, device=dout.device)
        if node_ptr.numel() and ctx.n:
            call("gsat_segment_pool_bwd", ptr(dout), ptr(node_ptr), G, H, int(ctx.mean), ptr(dx), stream())
        return dx, None, None, None


def segment_pool(x, segments, mean: bool):
    return SegmentPool.apply(x, segments.node_ptr, segments.G, mean)


AGGREGATOR_CODES = {"sum": 0, "mean": 1, "min": 2, "max": 3, "var": 4, "std": 5}
# the aggregator / scaler lists of the reference's PNA YAMLs (src/configs/PNA-*.yml): covered by the tiled backward
_FIXED_PNA = {((1, 2, 3, 5), (0,)): True, ((1, 2, 3, 5, 0), (0,)): True}
SCALER_CODES = {"identity": 0, "amplification": 1, "attenuation": 2, "linear": 3, "inverse_linear": 4}


class PnaAggregate(torch.autograd.Function):
    """[N, S*A*F] multi-aggregation of att_e * [x_i || x_j (|| edge_emb_e)].

    replaces: PNAConvSimple.message/aggregate (src/models/conv_layers.py:166-185, 193-259)."""

    @staticmethod
    def forward(ctx, x, att, edge_emb, index: BatchIndex, aggr_codes, scaler_codes, avg_lin: float, avg_log: float, node_att=None,
                passthrough: bool = False, lifted=None):
        # passthrough: ALSO return x itself (an identity output of this node).  A caller that feeds it to a second consumer of x -- the
        # layer's residual, src/models/pna.py:57-59 -- has that path's gradient arrive HERE, where the tiled backward adds it inside its
        # own dx pass (dx_add) instead of autograd launching one [N,H] add per layer.
        import ctypes
        ctx.set_materialize_grads(False)          # (out, x) outputs: an unused one arrives as None in backward
        x_in = x
        x = _f32c(x)
        ctx.passthrough = bool(passthrough)
        ctx.lifted = lifted
        if node_att is not None:
            out = PnaAggregate._forward_node_att(ctx, x, node_att, index, aggr_codes, scaler_codes, avg_lin, avg_log)
            return (out, x_in) if passthrough else out
        ctx.node_att = False
        attf = _flat_att(att, index.E)
        edge_emb = _f32c(edge_emb)
        N, H = x.shape
        if N != index.N:
            raise ValueError(f"x has {N} rows but the index was built for {index.N} nodes")
        if edge_emb is not None and tuple(edge_emb.shape) != (index.E, H):
            raise ValueError("edge_attr embedding must be [E, hidden]")
        A, S = len(aggr_codes), len(scaler_codes)
        F = (3 if edge_emb is not None else 2) * H
        a_arr = (ctypes.c_int32 * A)(*aggr_codes)
        s_arr = (ctypes.c_int32 * S)(*scaler_codes)
        out = torch.empty(N, S * A * F, dtype=torch.float32, device=x.device)
        hubs = index.long_rows_nowait[0]   # by-destination hub-chunk list (no host sync), None when no row is known / expected to be long
        if hubs is not None:
            call("gsat_pna_fwd_long", ptr(x), ptr(attf), ptr(edge_emb), ptr(index.rowptr_dst), ptr(index.src_by_dst), ptr(index.eid_by_dst),
                 N, index.E, H, a_arr, A, s_arr, S, float(avg_lin), float(avg_log), ptr(out), ptr(hubs),
                 ptr(index.pna_partial(H, edge_emb is not None)), stream())
        else:
            call("gsat_pna_fwd", ptr(x), ptr(attf), ptr(edge_emb), ptr(index.rowptr_dst), ptr(index.src_by_dst),
                 ptr(index.eid_by_dst), N, H, a_arr, A, s_arr, S, float(avg_lin), float(avg_log), ptr(out), stream())
        ctx.save_for_backward(x, attf, edge_emb)
        ctx.index = index
        ctx.cfg = (tuple(aggr_codes), tuple(scaler_codes), float(avg_lin), float(avg_log))
        ctx.att_shape = None if att is None else att.shape
        return (out, x_in) if passthrough else out

    @staticmethod
    def _forward_node_att(ctx, x, node_att, index, aggr_codes, scaler_codes, avg_lin, avg_log):
        """Edge weight = node_att[row] * node_att[source] formed inside the kernels (pna_aggregate checked that the tiled backward covers
        this batch)."""
        import ctypes
        N, H = x.shape
        if N != index.N or node_att.numel() != N:
            raise ValueError(f"x / node attention have {N} / {node_att.numel()} rows but the index was built for {index.N} nodes")
        na = _f32c(node_att).view(-1)
        A, S = len(aggr_codes), len(scaler_codes)
        a_arr = (ctypes.c_int32 * A)(*aggr_codes)
        s_arr = (ctypes.c_int32 * S)(*scaler_codes)
        out = torch.empty(N, S * A * 2 * H, dtype=torch.float32, device=x.device)
        call("gsat_pna_fwd_node_att", ptr(x), ptr(na), ptr(index.rowptr_dst), ptr(index.src_by_dst), N, H, a_arr, A, s_arr, S,
             float(avg_lin), float(avg_log), ptr(out), stream())
        ctx.save_for_backward(x, na)
        ctx.index, ctx.node_att = index, True
        ctx.cfg = (tuple(aggr_codes), tuple(scaler_codes), float(avg_lin), float(avg_log))
        ctx.att_shape = node_att.shape
        return out

    @staticmethod
    def _backward_node_att(ctx, dout, dx_add=None):
        import ctypes
        x, na = ctx.saved_tensors
        index = ctx.index
        aggr_codes, scaler_codes, _, _ = ctx.cfg
        dout = _f32c(dout)
        N, H = x.shape
        A, S = len(aggr_codes), len(scaler_codes)
        a_arr = (ctypes.c_int32 * A)(*aggr_codes)
        s_arr = (ctypes.c_int32 * S)(*scaler_codes)
        dev = x.device
        need_att = ctx.needs_input_grad[8]
        tile_ptr, T, rows_nominal, rows_cap, edges_cap, spill = index.pna_tiles(H)
        dmsg = torch.empty(max(index.E, 1), H, dtype=torch.float32, device=dev)[: index.E]
        dw = torch.empty(max(index.E, 1), dtype=torch.float32, device=dev) if need_att else None
        dna, acc, ret = None, 0, None
        if need_att:
            if ctx.lifted is not None:
                dna, acc, ret = ctx.lifted._shared_dna(N, dev)
            else:
                dna = ret = torch.empty(N, dtype=torch.float32, device=dev)
        dx = torch.empty_like(x)
        call("gsat_pna_bwd_tiled_node_att", ptr(x), ptr(na), ptr(dout), ptr(index.rowptr_dst), ptr(index.src_by_dst), ptr(tile_ptr), T,
             rows_nominal, rows_cap, edges_cap, ptr(index.rowptr_src), ptr(index.slot_dst_of_srcslot), N, index.E, H, a_arr, A, s_arr, S,
             ptr(spill[1:]), ptr(spill[:1]), ptr(dx), ptr(dmsg), ptr(dna), ptr(dw), ptr(dx_add), acc, stream())
        return dx, None, None, None, None, None, None, None, (ret.view(ctx.att_shape) if ret is not None else None), None, None

    @staticmethod
    def backward(ctx, dout, dx_add=None):
        import ctypes
        dx_add = None if dx_add is None else _f32c(dx_add)
        if dout is None:                       # only the identity output was used
            return dx_add, None, None, None, None, None, None, None, None, None, None
        if ctx.node_att:
            return PnaAggregate._backward_node_att(ctx, dout, dx_add)
        x, attf, edge_emb = ctx.saved_tensors
        index = ctx.index
        aggr_codes, scaler_codes, avg_lin, avg_log = ctx.cfg
        dout = _f32c(dout)
        N, H = x.shape
        A, S = len(aggr_codes), len(scaler_codes)
        a_arr = (ctypes.c_int32 * A)(*aggr_codes)
        s_arr = (ctypes.c_int32 * S)(*scaler_codes)
        need_att = attf is not None and ctx.needs_input_grad[1]
        need_ee = edge_emb is not None and ctx.needs_input_grad[2]
        dev = x.device
        dmsg = torch.empty(max(index.E, 1), H, dtype=torch.float32, device=dev)[: index.E]
        datt = torch.empty(index.E, dtype=torch.float32, device=dev) if need_att else None
        tiles = None
        hubs = index.long_rows_nowait[0]
        from .graph_index import sync_free
        # the tiled kernel walks a hub row with one lane group (correct, slow): batches known to hold hubs take the chunked two-pass path;
        # inside a captured step (no host knowledge of the degrees) the tiled kernel stays
        if (edge_emb is None and _FIXED_PNA.get((aggr_codes, scaler_codes)) and os.environ.get("GSAT_PNA_TILED", "1") != "0"
                and (hubs is None or sync_free())):
            tiles = index.pna_tiles(H) or None
        if tiles is not None:
            # one launch: per-edge gradient rows stay in LDS and are summed per source there (no [E,H] round trip through HBM)
            tile_ptr, T, rows_nominal, rows_cap, edges_cap, spill = tiles
            dx = torch.empty_like(x)
            call("gsat_pna_bwd_tiled", ptr(x), ptr(attf), ptr(dout), ptr(index.rowptr_dst), ptr(index.src_by_dst), ptr(index.eid_by_dst),
                 ptr(tile_ptr), T, rows_nominal, rows_cap, edges_cap, ptr(index.rowptr_src), ptr(index.slot_dst_of_srcslot), N, index.E, H,
                 a_arr, A, s_arr, S, ptr(spill[1:]), ptr(spill[:1]), ptr(dx), ptr(dmsg), ptr(datt), ptr(dx_add), stream())
            return dx, (datt.view(ctx.att_shape) if need_att else None), None, None, None, None, None, None, None, None, None
        dx_self = torch.empty_like(x)
        dee = torch.empty_like(edge_emb) if need_ee else None
        if hubs is not None:
            call("gsat_pna_bwd_long", ptr(x), ptr(attf), ptr(edge_emb), ptr(dout), ptr(index.rowptr_dst), ptr(index.src_by_dst),
                 ptr(index.eid_by_dst), N, index.E, H, a_arr, A, s_arr, S, avg_lin, avg_log, ptr(dx_self), ptr(dmsg), ptr(datt),
                 ptr(dee), ptr(hubs), ptr(index.pna_partial(H, edge_emb is not None)), stream())
        else:
            call("gsat_pna_bwd", ptr(x), ptr(attf), ptr(edge_emb), ptr(dout), ptr(index.rowptr_dst), ptr(index.src_by_dst),
                 ptr(index.eid_by_dst), N, H, a_arr, A, s_arr, S, avg_lin, avg_log, ptr(dx_self), ptr(dmsg), ptr(datt),
                 ptr(dee), stream())
        # second pass: sum the per-edge gradient rows of every source node through the inverted index
        dx = torch.empty_like(x)
        call("gsat_aggr_sum_fwd", ptr(dmsg), ptr(dx_self), None, None, ptr(index.rowptr_src),
             ptr(index.slot_dst_of_srcslot), None, N, index.E, H, 1.0, ptr(dx),
             ptr(index.long_rows[1]), ptr(index.partial(H)) if index.long_rows[1] is not None else None, stream())
        if dx_add is not None:
            dx = dx + dx_add
        return dx, (datt.view(ctx.att_shape) if need_att else None), dee, None, None, None, None, None, None, None, None


def pna_aggregate(x, index, att, edge_emb, aggregators, scalers, avg_deg, passthrough: bool = False):
    a = [AGGREGATOR_CODES[k] for k in aggregators]
    s = [SCALER_CODES[k] for k in scalers]
    if isinstance(att, LiftedAttention):
        # lifted node attention: formed inside the aggregation kernels when the batch takes the one-launch (tiled) backward -- no edge
        # features, the reference's aggregator set, no hub rows, rows of >= 64 channels; anything else gets the [E, 1] tensor
        if (edge_emb is None and att.index is index and _FIXED_PNA.get((tuple(a), tuple(s))) and index.long_rows_nowait[0] is None
                and os.environ.get("GSAT_PNA_TILED", "1") != "0" and os.environ.get("GSAT_NODE_ATT_LIFT", "0") != "1"
                and att._edge is None and index.pna_tiles(x.shape[1])):
            return PnaAggregate.apply(x, None, None, index, a, s, avg_deg["lin"], avg_deg["log"], att.shared_node_att, passthrough, att)
        att = att.edge()
    return PnaAggregate.apply(x, att, edge_emb, index, a, s, avg_deg["lin"], avg_deg["log"], None, passthrough)


# ------------------------------------------------------------------------------------------------
# attention extractor MLP (+ fused concrete sampler)
# ------------------------------------------------------------------------------------------------
def _attn_args(emb, params, index, segments, edge_mode, training, p, seed, mask1, mask2, u, bufs, seed_dev=None, noise_philox=False):
    from ._lib import AttnArgs
    W1, b1, W2, b2, W3, b3 = params
    N, H = emb.shape
    C1, C2 = W1.shape[0], W2.shape[0]
    a = AttnArgs()
    if edge_mode:
        eptr, order, _, eg32 = segments.edge_segments
        a.M, a.seg_ptr, a.seg_order, a.row_seg = index.E, ptr(eptr), ptr(order), ptr(eg32)
        a.src, a.dst = ptr(index.src32), ptr(index.dst32)
    else:
        a.M, a.seg_ptr, a.seg_order, a.row_seg = N, ptr(segments.node_ptr), None, ptr(segments.node_seg32)
        a.src, a.dst = None, None
    a.N, a.G, a.H, a.C1, a.C2 = N, segments.G, H, C1, C2
    a.edge_mode, a.training, a.p_drop, a.seed = int(edge_mode), int(training), float(p), int(seed) & ((1 << 64) - 1)
    a.W1, a.b1, a.W2, a.b2, a.W3, a.b3 = ptr(W1), ptr(b1), ptr(W2), ptr(b2), ptr(W3), ptr(b3)
    a.emb, a.mask1, a.mask2, a.u = ptr(emb), ptr(mask1), ptr(mask2), ptr(u)
    P, Q, a1, h2, stats, logits, att = bufs
    a.P, a.Q, a.a1, a.h2, a.stats, a.logits, a.att = ptr(P), ptr(Q), ptr(a1), ptr(h2), ptr(stats), ptr(logits), ptr(att)
    a.seed_dev = ptr(seed_dev)
    a.noise_philox = int(bool(noise_philox) and u is None)
    a.fused = 0              # library policy: one-launch forward for large node-mode batches (GSAT_ATTN_FUSED=1 / 0 forces / forbids it)
    a.node_ptr = ptr(segments.node_ptr)
    return a


class ExtractorAttention(torch.autograd.Function):
    """(logits, att) = sampler(MLP([emb[src] || emb[dst]] | emb, per-graph InstanceNorm)).

    replaces: ExtractorMLP.forward + MLP/BatchSequential/InstanceNorm + GSAT.sampling
    (example/gsat.py:94-103,131-139; src/utils/get_model.py:47-68; src/run_gsat.py:877-927)."""

    last_forward_kind = 0        # what the library chose for the latest forward (gsat_attn_fwd_kind): for logs and the bench line

    @staticmethod
    def forward(ctx, emb, W1, b1, W2, b2, W3, b3, index, segments, edge_mode, training, p, seed, mask1, mask2, u, seed_dev=None, noise_philox=False):
        import ctypes
        ctx.set_materialize_grads(False)          # an unused output (the logits, normally) arrives as None in backward, not as a zero-filled [M,1] tensor
        emb = _f32c(emb)
        params = tuple(_f32c(t) for t in (W1, b1, W2, b2, W3, b3))
        mask1, mask2 = _f32c(mask1), _f32c(mask2)
        u = None if u is None else _f32c(u).view(-1)
        N, H = emb.shape
        C1, C2 = params[0].shape[0], params[2].shape[0]
        C0 = 2 * H if edge_mode else H
        if params[0].shape[1] != C0 or params[2].shape[1] != C1 or tuple(params[4].shape) != (1, C2):
            raise ValueError("extractor weight shapes do not match the embedding width")
        if N != index.N:
            raise ValueError(f"emb has {N} rows but the index was built for {index.N} nodes")
        M = index.E if edge_mode else N
        for m_, c_ in ((mask1, C1), (mask2, C2)):
            if m_ is not None and tuple(m_.shape) != (M, c_):
                raise ValueError("dropout keep-mask has the wrong shape")
        if u is not None and u.numel() != M:
            raise ValueError("noise tensor has the wrong number of entries")
        dev, f32 = emb.device, torch.float32
        G = segments.G
        P = torch.empty(N, C1, dtype=f32, device=dev)
        Q = torch.empty(N, C1, dtype=f32, device=dev) if edge_mode else None
        a1 = torch.empty(M, C1, dtype=f32, device=dev)
        h2 = torch.empty(M, C2, dtype=f32, device=dev)
        stats = torch.empty(max(G, 1) * (2 * C1 + 2 * C2), dtype=f32, device=dev)
        logits = torch.empty(M, 1, dtype=f32, device=dev)
        att = torch.empty(M, 1, dtype=f32, device=dev)
        bufs = (P, Q, a1, h2, stats, logits, att)
        args = _attn_args(emb, params, index, segments, edge_mode, training, p, seed, mask1, mask2, u, bufs, seed_dev, noise_philox)
        from ._lib import load
        fws_bytes = int(load().gsat_attn_fwd_workspace_bytes(ctypes.byref(args)))
        ExtractorAttention.last_forward_kind = int(load().gsat_attn_fwd_kind(ctypes.byref(args)))      # 0 staged / 1 fused fp32 / 2 fused split-bf16 x 6
        if fws_bytes:
            fws = torch.empty(fws_bytes, dtype=torch.uint8, device=dev)
            args.fwd_workspace, args.fwd_workspace_bytes = ptr(fws), fws_bytes
        call("gsat_attn_fwd", ctypes.byref(args), stream())
        ctx.save_for_backward(emb, *params, P, Q if Q is not None else emb.new_empty(0), a1, h2, stats, att,
                              mask1 if mask1 is not None else emb.new_empty(0),
                              mask2 if mask2 is not None else emb.new_empty(0),
                              u if u is not None else emb.new_empty(0))
        ctx.meta = (index, segments, bool(edge_mode), bool(training), float(p), int(seed),
                    mask1 is not None, mask2 is not None, u is not None)
        ctx.seed_dev = seed_dev
        return logits, att

    @staticmethod
    def backward(ctx, dlogits, datt):
        import ctypes
        from ._lib import AttnGrads, load
        (emb, W1, b1, W2, b2, W3, b3, P, Q, a1, h2, stats, att, mask1, mask2, u) = ctx.saved_tensors
        index, segments, edge_mode, training, p, seed, has_m1, has_m2, has_u = ctx.meta
        params = (W1, b1, W2, b2, W3, b3)
        logits_dummy = att  # not read by the backward
        bufs = (P, Q if edge_mode else None, a1, h2, stats, logits_dummy, att)
        args = _attn_args(emb, params, index, segments, edge_mode, training, p, seed, mask1 if has_m1 else None,
                          mask2 if has_m2 else None, u if has_u else None, bufs, ctx.seed_dev)
        g = AttnGrads()
        dlogits = None if dlogits is None else _f32c(dlogits)
        datt = None if datt is None else _f32c(datt)
        g.dlogits, g.datt = ptr(dlogits), ptr(datt)
        if edge_mode:
            g.rowptr_src, g.eid_by_src = ptr(index.rowptr_src), ptr(index.eid_by_src)
            g.rowptr_dst, g.eid_by_dst = ptr(index.rowptr_dst), ptr(index.eid_by_dst)
            g.chunk_ptr_dst, g.chunk_ptr_src = (ptr(t) for t in index.long_rows)
        demb = torch.empty_like(emb)
        grads = [torch.empty_like(t) for t in params]
        g.demb = ptr(demb)
        g.dW1, g.db1, g.dW2, g.db2, g.dW3, g.db3 = (ptr(t) for t in grads)
        ws_bytes = int(load().gsat_attn_bwd_workspace_bytes(ctypes.byref(args)))
        ws = torch.empty(ws_bytes, dtype=torch.uint8, device=emb.device)
        g.workspace, g.workspace_bytes = ptr(ws), ws_bytes
        call("gsat_attn_bwd", ctypes.byref(args), ctypes.byref(g), stream())
        return (demb, *grads, None, None, None, None, None, None, None, None, None, None, None)


def new_seed() -> int:
    """63-bit Philox seed drawn from torch's CPU generator (follows torch.manual_seed, no device sync)."""
    return int(torch.empty((), dtype=torch.int64).random_().item())


_DEVICE_SEEDS = {}


def device_seed(device) -> torch.Tensor:
    """A fresh 1-element int64 seed tensor on ``device`` for a captured (sync-free) step: the next value of a device-resident counter stream
    (gsat_seed_next: one launch; torch's graph-safe ``random_()`` takes three).  The stream's base comes from torch's CPU generator the first
    time a (device, torch.initial_seed()) pair is seen, so ``torch.manual_seed`` restarts it; it must first be used outside a capture
    (warm-up runs do), otherwise this falls back to ``random_()``."""
    key = (str(device), torch.initial_seed())
    st = _DEVICE_SEEDS.get(key)
    if st is None:
        if torch.cuda.is_current_stream_capturing() or os.environ.get("GSAT_DEVICE_SEEDS", "1") == "0":
            return torch.empty(1, dtype=torch.int64, device=device).random_()
        st = torch.tensor([new_seed(), 0], dtype=torch.int64, device=device)
        _DEVICE_SEEDS.clear()
        _DEVICE_SEEDS[key] = st
    out = torch.empty(1, dtype=torch.int64, device=device)
    call("gsat_seed_next", ptr(st), ptr(out), stream())
    return out


class Sample(torch.autograd.Function):
    """sigmoid((logits + noise)/temp): concrete (mode 1) / Gumbel (mode 2) / none (mode 0)."""

    @staticmethod
    def forward(ctx, logits, noise, mode: int, temp: float, eps: float):
        z = _f32c(logits)
        nz = None if noise is None else _f32c(noise)
        att = torch.empty_like(z)
        call("gsat_sample_fwd", ptr(z), ptr(nz), int(mode), float(temp), float(eps), z.numel(), ptr(att), stream())
        ctx.save_for_backward(att)
        ctx.temp = float(temp)
        return att

    @staticmethod
    def backward(ctx, datt):
        (att,) = ctx.saved_tensors
        datt = _f32c(datt)
        dz = torch.empty_like(att)
        call("gsat_sample_bwd", ptr(att), ptr(datt), ctx.temp, att.numel(), ptr(dz), stream())
        return dz, None, None, None, None


class Lift(torch.autograd.Function):
    """edge_att = node_att[src] * node_att[dst]  (example/gsat.py:112-117)."""

    @staticmethod
    def forward(ctx, node_att, index: BatchIndex):
        a = _f32c(node_att)
        if a.numel() != index.N:
            raise ValueError("node attention must have one entry per node")
        out = torch.empty(index.E, 1, dtype=torch.float32, device=a.device)
        call("gsat_lift_fwd", ptr(a), ptr(index.src32), ptr(index.dst32), index.E, ptr(out), stream())
        ctx.save_for_backward(a)
        ctx.index = index
        return out

    @staticmethod
    def backward(ctx, dout):
        (a,) = ctx.saved_tensors
        ix = ctx.index
        dout = _f32c(dout)
        da = torch.empty_like(a)
        call("gsat_lift_bwd", ptr(a), ptr(dout), ptr(ix.rowptr_src), ptr(ix.dst_by_src), ptr(ix.eid_by_src),
             ptr(ix.rowptr_dst), ptr(ix.src_by_dst), ptr(ix.eid_by_dst), ix.N, ptr(da), stream())
        return da, None


class Symmetrise(torch.autograd.Function):
    """(att + att[rev]) / 2  (example/gsat.py:81-83); with ``flag`` (device int32) the `if is_undirected` of the
    reference is decided inside the kernel: flag[0] == 0 -> identity."""

    @staticmethod
    def forward(ctx, att, rev, flag=None):
        a = _f32c(att)
        out = torch.empty_like(a)
        call("gsat_symmetrise", ptr(a), ptr(rev), ptr(flag), a.numel(), ptr(out), stream())
        ctx.save_for_backward(rev, flag if flag is not None else rev.new_empty(0))
        ctx.has_flag = flag is not None
        return out

    @staticmethod
    def backward(ctx, dout):
        rev, flag = ctx.saved_tensors
        d = _f32c(dout)
        da = torch.empty_like(d)
        call("gsat_symmetrise", ptr(d), ptr(rev), ptr(flag if ctx.has_flag else None), d.numel(), ptr(da), stream())
        return da, None, None


class InfoLoss(torch.autograd.Function):
    """mean(att*log(att/r+1e-6) + (1-att)*log((1-att)/(1-r+1e-6)+1e-6)); r scalar or detached tensor prior."""

    @staticmethod
    def forward(ctx, att, r):
        a = _f32c(att)
        r_vec = _f32c(r.detach()) if isinstance(r, torch.Tensor) else None
        r_scalar = 0.0 if r_vec is not None else float(r)
        if r_vec is not None and r_vec.numel() != a.numel():
            raise ValueError("tensor prior must have one entry per attention value")
        out = torch.empty((), dtype=torch.float32, device=a.device)
        partial = torch.empty(1024, dtype=torch.float32, device=a.device)
        call("gsat_info_loss_fwd", ptr(a), ptr(r_vec), r_scalar, a.numel(), ptr(partial), ptr(out), stream())
        ctx.save_for_backward(a, r_vec if r_vec is not None else a.new_empty(0))
        ctx.r_scalar, ctx.has_vec = r_scalar, r_vec is not None
        return out

    @staticmethod
    def backward(ctx, gout):
        a, r_vec = ctx.saved_tensors
        gout = _f32c(gout)
        da = torch.empty_like(a)
        call("gsat_info_loss_bwd", ptr(a), ptr(r_vec if ctx.has_vec else None), ctx.r_scalar, ptr(gout), a.numel(), ptr(da), stream())
        return da, None


class InstanceNormFn(torch.autograd.Function):
    """per-graph InstanceNorm (eps 1e-5, no affine, batch statistics always) -- src/utils/get_model.py:50-51."""

    @staticmethod
    def forward(ctx, x, seg_ptr, seg_order, row_seg, G):
        x = _f32c(x)
        M, C = x.shape
        y = torch.empty_like(x)
        stats = torch.empty(max(G, 1) * 2 * C, dtype=torch.float32, device=x.device)
        call("gsat_instance_norm_fwd", ptr(x), ptr(seg_ptr), ptr(seg_order), ptr(row_seg), M, G, C, ptr(y), ptr(stats), stream())
        ctx.save_for_backward(y, stats, seg_ptr, row_seg, seg_order if seg_order is not None else seg_ptr.new_empty(0))
        ctx.G, ctx.has_order = G, seg_order is not None
        return y

    @staticmethod
    def backward(ctx, dy):
        y, stats, seg_ptr, row_seg, seg_order = ctx.saved_tensors
        dy = _f32c(dy)
        M, C = y.shape
        dx = torch.empty_like(y)
        ws = torch.empty(max(ctx.G, 1) * 2 * C, dtype=torch.float32, device=y.device)
        call("gsat_instance_norm_bwd", ptr(y), ptr(dy), ptr(stats), ptr(seg_ptr), ptr(seg_order if ctx.has_order else None),
             ptr(row_seg), M, ctx.G, C, ptr(dx), ptr(ws), stream())
        return dx, None, None, None, None


class EmbeddingSum(torch.autograd.Function):
    """out = sum_col W[offset(col) + x[:, col]]  (ogb AtomEncoder / BondEncoder); backward = O^T dout (one MFMA GEMM)."""

    @staticmethod
    def forward(ctx, x_idx, W_all, dims, onehot_cache):
        import ctypes
        W = _f32c(W_all)
        x_idx = x_idx.contiguous()
        if x_idx.dtype != torch.int64 or x_idx.dim() != 2 or x_idx.shape[1] != len(dims):
            raise ValueError("categorical features must be int64 [N, %d]" % len(dims))
        N, H = x_idx.shape[0], W.shape[1]
        d_arr = (ctypes.c_int32 * len(dims))(*dims)
        out = torch.empty(N, H, dtype=torch.float32, device=W.device)
        call("gsat_embsum_fwd", ptr(x_idx), d_arr, len(dims), ptr(W), N, H, ptr(out), stream())
        ctx.x_idx, ctx.dims, ctx.cache, ctx.R = x_idx, tuple(dims), onehot_cache, W.shape[0]
        return out

    @staticmethod
    def backward(ctx, dout):
        import ctypes
        from ._lib import load
        dout = _f32c(dout)
        x_idx, dims, R = ctx.x_idx, ctx.dims, ctx.R
        N, H = dout.shape
        Rp = (R + 3) // 4 * 4
        key = (x_idx.data_ptr(), x_idx._version, N)
        O = ctx.cache.get("O") if ctx.cache.get("key") == key else None
        if O is None:
            d_arr = (ctypes.c_int32 * len(dims))(*dims)
            O = torch.empty(N, Rp, dtype=torch.float32, device=dout.device)
            call("gsat_onehot_rows", ptr(x_idx), d_arr, len(dims), N, Rp, ptr(O), stream())
            ctx.cache["key"], ctx.cache["O"], ctx.cache["x"] = key, O, x_idx
        dW = torch.empty(Rp, H, dtype=torch.float32, device=dout.device)
        wsf = int(load().gsat_gemm_workspace_floats(1, Rp, H, N))
        ws = torch.empty(max(wsf, 1), dtype=torch.float32, device=dout.device)
        call("gsat_gemm_f32", 1, 0, Rp, H, N, ptr(O), Rp, ptr(dout), H, ptr(dW), H, None, 0, ptr(ws), wsf, stream())
        return None, dW[:R], None, None


class BatchNormFn(torch.autograd.Function):
    """BatchNorm1d over rows with optional fused ReLU (src/models/gin.py:58, src/models/pna.py:45,57) and, for PNA, the rest of
    the layer tail in the same passes: y = dropout_p(relu(BN(x)) + residual)  (src/models/pna.py:57-59).  The dropout mask is
    Philox stream 3 keyed by (seed, row, column); ``seed_dev`` (a 1-element int64 device tensor) replaces the host seed in
    sync-free / hipGraph mode."""

    @staticmethod
    def forward(ctx, x, weight, bias, running_mean, running_var, training, momentum, eps, relu, residual=None, dropout_p=0.0,
                seed=0, seed_dev=None):
        from ._lib import load
        x, weight, bias, residual = _f32c(x), _f32c(weight), _f32c(bias), _f32c(residual)
        N, C = x.shape
        if residual is not None and residual.shape != x.shape:
            raise ValueError("residual must have the shape of the normalised tensor")
        y = torch.empty_like(x)
        mean = torch.empty(C, dtype=torch.float32, device=x.device)
        rstd = torch.empty(C, dtype=torch.float32, device=x.device)
        ws = torch.empty(max(int(load().gsat_bn_workspace_floats(N, C)), 1), dtype=torch.float32, device=x.device)
        call("gsat_bn_act_fwd", ptr(x), ptr(weight), ptr(bias), ptr(running_mean), ptr(running_var), N, C, int(training),
             float(momentum), float(eps), int(relu), ptr(residual), float(dropout_p), int(seed), ptr(seed_dev), ptr(y), ptr(mean),
             ptr(rstd), ptr(ws), stream())
        ctx.save_for_backward(x, weight, bias, mean, rstd)
        ctx.flags = (bool(training), bool(relu), float(dropout_p), int(seed), residual is not None)
        ctx.seed_dev = seed_dev
        return y

    @staticmethod
    def backward(ctx, dy):
        from ._lib import load
        x, weight, bias, mean, rstd = ctx.saved_tensors
        training, relu, dropout_p, seed, has_res = ctx.flags
        dy = _f32c(dy)
        N, C = x.shape
        dx = torch.empty_like(x)
        dres = torch.empty_like(x) if has_res and ctx.needs_input_grad[9] else None
        dgamma, dbeta = torch.empty_like(weight), torch.empty_like(bias)
        ws = torch.empty(max(int(load().gsat_bn_workspace_floats(N, C)), 1), dtype=torch.float32, device=x.device)
        call("gsat_bn_act_bwd", ptr(x), ptr(dy), ptr(weight), ptr(bias), ptr(mean), ptr(rstd), N, C, int(training), int(relu),
             dropout_p, seed, ptr(ctx.seed_dev), ptr(dx), ptr(dres), ptr(dgamma), ptr(dbeta), ptr(ws), stream())
        return dx, dgamma, dbeta, None, None, None, None, None, None, dres, None, None, None


class SyncBatchNormFn(torch.autograd.Function):
    """BatchNorm1d (+ the fused PNA layer tail) whose batch statistics span every rank of ``group``: the HIP kernels produce the
    local column sums, torch.distributed all-reduces the [C] vectors (two small collectives forward, one backward), the HIP kernels
    apply them.  A sharded run then normalises exactly like the single-process model at the same global batch."""

    @staticmethod
    def forward(ctx, x, weight, bias, running_mean, running_var, momentum, eps, relu, residual, dropout_p, seed, seed_dev, group):
        import torch.distributed as dist
        from ._lib import load
        x, weight, bias, residual = _f32c(x), _f32c(weight), _f32c(bias), _f32c(residual)
        N, C = x.shape
        dev = x.device
        ws = torch.empty(max(int(load().gsat_bn_workspace_floats(max(N, 1), C)), 1), dtype=torch.float32, device=dev)
        stat = torch.empty(C + 1, dtype=torch.float32, device=dev)          # [ sum_c ... | row count ]
        call("gsat_bn_local_sum", ptr(x), None, N, C, ptr(stat), ptr(ws), stream())
        stat[C] = float(N)
        dist.all_reduce(stat, group=group)
        n_global = stat[C:C + 1].clone()
        mean = stat[:C] / n_global
        q = torch.empty(C, dtype=torch.float32, device=dev)
        call("gsat_bn_local_sum", ptr(x), ptr(mean), N, C, ptr(q), ptr(ws), stream())
        dist.all_reduce(q, group=group)
        var = q / n_global
        rstd = torch.rsqrt(var + eps)
        if running_mean is not None:
            with torch.no_grad():
                running_mean.mul_(1 - momentum).add_(mean, alpha=momentum)
                running_var.mul_(1 - momentum).add_(q / (n_global - 1).clamp_(min=1), alpha=momentum)     # unbiased, as torch
        y = torch.empty_like(x)
        call("gsat_bn_apply_fwd", ptr(x), ptr(weight), ptr(bias), ptr(mean), ptr(rstd), N, C, int(relu), ptr(residual), float(dropout_p),
             int(seed), ptr(seed_dev), ptr(y), stream())
        ctx.save_for_backward(x, weight, bias, mean, rstd, n_global)
        ctx.flags = (bool(relu), float(dropout_p), int(seed), residual is not None)
        ctx.seed_dev, ctx.group = seed_dev, group
        return y

    @staticmethod
    def backward(ctx, dy):
        import torch.distributed as dist
        from ._lib import load
        x, weight, bias, mean, rstd, n_global = ctx.saved_tensors
        relu, dropout_p, seed, has_res = ctx.flags
        dy = _f32c(dy)
        N, C = x.shape
        dev = x.device
        ws = torch.empty(max(int(load().gsat_bn_workspace_floats(max(N, 1), C)), 1), dtype=torch.float32, device=dev)
        sums = torch.empty(2, C, dtype=torch.float32, device=dev)
        call("gsat_bn_local_bwd_sums", ptr(x), ptr(dy), ptr(weight), ptr(bias), ptr(mean), ptr(rstd), N, C, int(relu), dropout_p, seed,
             ptr(ctx.seed_dev), ptr(sums[0]), ptr(sums[1]), ptr(ws), stream())
        dbeta, dgamma = sums[0].clone(), sums[1].clone()       # parameter gradients stay local: they are averaged with the rest
        dist.all_reduce(sums, group=ctx.group)
        dx = torch.empty_like(x)
        dres = torch.empty_like(x) if has_res and ctx.needs_input_grad[8] else None
        call("gsat_bn_apply_bwd", ptr(x), ptr(dy), ptr(weight), ptr(bias), ptr(mean), ptr(rstd), ptr(sums[0]), ptr(sums[1]),
             0, ptr(n_global), N, C, int(relu), dropout_p, seed, ptr(ctx.seed_dev), ptr(dx), ptr(dres), stream())
        return dx, dgamma, dbeta, None, None, None, None, None, dres, None, None, None, None


class ReluDropout(torch.autograd.Function):
    """dropout_p(relu(x)) in one launch, backward from the output alone (GIN layer tail, src/models/gin.py:49-52)."""

    @staticmethod
    def forward(ctx, x, p: float, seed: int, seed_dev):
        x = _f32c(x)
        N, C = x.shape
        y = torch.empty_like(x)
        call("gsat_relu_dropout_fwd", ptr(x), N, C, float(p), int(seed) & ((1 << 64) - 1), ptr(seed_dev), ptr(y), stream())
        ctx.save_for_backward(y)
        ctx.p = float(p)
        return y

    @staticmethod
    def backward(ctx, dy):
        (y,) = ctx.saved_tensors
        dy = _f32c(dy)
        dx = torch.empty_like(y)
        call("gsat_relu_dropout_bwd", ptr(y), ptr(dy), y.shape[0], y.shape[1], ctx.p, ptr(dx), stream())
        return dx, None, None, None


def relu_dropout(x, p: float, training: bool):
    """F.dropout(relu(x), p, training) -- HIP for 2-D fp32 ROCm tensors with a width that is a multiple of 4, torch otherwise."""
    p = float(p) if training else 0.0
    if not (x.is_cuda and x.dim() == 2 and x.dtype == torch.float32 and x.shape[1] % 4 == 0):
        return torch.nn.functional.dropout(torch.relu(x), p, training=True) if p > 0.0 else torch.relu(x)
    seed, seed_dev = 0, None
    if p > 0.0:
        from .graph_index import sync_free
        if sync_free():            # captured steps: the seed word lives on the device and is redrawn by a graph-safe RNG op
            seed_dev = device_seed(x.device)
        else:
            seed = new_seed()
    return ReluDropout.apply(x, p, seed, seed_dev)


def colsum(x):
    """Deterministic column sum of a 2-D fp32 device tensor (bias gradients)."""
    from ._lib import load
    x = _f32c(x)
    R, C = x.shape
    out = torch.empty(C, dtype=torch.float32, device=x.device)
    ws = torch.empty(max(int(load().gsat_colsum_workspace_floats(C)), 1), dtype=torch.float32, device=x.device)
    call("gsat_colsum", ptr(x), R, C, ptr(out), ptr(ws), stream())
    return out


def _gemm(a_t, b_t, M, N, K, A, lda, B, ldb, C, ldc, bias=None, split=True):
    from ._lib import load
    wsf = int(load().gsat_gemm_workspace_floats(int(a_t), M, N, K))
    ws = torch.empty(wsf, dtype=torch.float32, device=C.device) if wsf else None
    call("gsat_gemm_bf16x3" if split else "gsat_gemm_f32", int(a_t), int(b_t), M, N, K, ptr(A), lda, ptr(B), ldb, ptr(C), ldc,
         ptr(bias), 0, ptr(ws), wsf, stream())


# Linear products above this many FLOPs run on the split-bf16 MFMA path (gsat_gemm_bf16x3: 2-3x the fp32 library GEMM at the
# backbone's shapes); smaller ones stay on the library (hipBLASLt via torch), which is as fast there
_OWN_GEMM_FLOPS = float(os.environ.get("GSAT_OWN_GEMM_FLOPS", "2e9"))


class LinearFn(torch.autograd.Function):
    """y = x W^T + b for tall-skinny x (node / edge rows).  Large products -- forward, dx and the weight gradient, which
    reduces over ~1e5 rows into a tiny [out, in] matrix -- run on the hand-written MFMA GEMM; small ones on the library."""

    @staticmethod
    def forward(ctx, x, weight, bias):
        x, weight = _f32c(x), _f32c(weight)
        M, K = x.shape
        N = weight.shape[0]
        if 2.0 * M * N * K >= _OWN_GEMM_FLOPS and (bias is None or bias.data_ptr() % 16 == 0):
            y = torch.empty(M, N, dtype=torch.float32, device=x.device)
            _gemm(0, 1, M, N, K, x, K, weight, K, y, N, None if bias is None else _f32c(bias))
        else:
            y = torch.nn.functional.linear(x, weight, bias)
        ctx.save_for_backward(x, weight)
        ctx.has_bias = bias is not None
        return y

    @staticmethod
    def backward(ctx, dy):
        x, weight = ctx.saved_tensors
        dy = _f32c(dy)
        rows, n_out, n_in = x.shape[0], weight.shape[0], weight.shape[1]
        big = 2.0 * rows * n_out * n_in >= _OWN_GEMM_FLOPS
        dx = None
        if ctx.needs_input_grad[0]:
            if big:
                dx = torch.empty_like(x)
                _gemm(0, 0, rows, n_in, n_out, dy, n_out, weight, n_in, dx, n_in)
            else:
                dx = dy @ weight
        dw = None
        if ctx.needs_input_grad[1]:
            dw = torch.empty_like(weight)
            _gemm(1, 0, n_out, n_in, rows, dy, n_out, x, n_in, dw, n_in, split=big)
        db = colsum(dy) if ctx.has_bias and ctx.needs_input_grad[2] else None
        return dx, dw, db


class PnaConvFn(torch.autograd.Function):
    """PNAConvSimple.forward = post_nn(aggregate(message)) (src/models/conv_layers.py:148-153) as ONE autograd node on the compact
    aggregate: the forward writes the x_j parts [N, A*H] + four scalars per row (gsat_pna_fwd_compact) and the post_nn GEMM rebuilds the
    x_i columns in its operand loader (gsat_pna_post_fwd); the backward takes dW the same way (gsat_pna_post_dw), forms dAgg = dout W
    once and hands it to the one-launch aggregation backward.  Saves half of the [N, A*2*H] write + read per layer pass and half of the
    saved activation.  Edge weights: none, an [E] tensor, or node attention formed inside the kernels."""

    @staticmethod
    def forward(ctx, x, att, node_att, weight, bias, index: BatchIndex, aggr_codes, passthrough: bool = False, lifted=None):
        import ctypes
        ctx.set_materialize_grads(False)
        ctx.lifted = lifted
        x_in = x
        ctx.passthrough = bool(passthrough)
        x, weight = _f32c(x), _f32c(weight)
        bias = None if bias is None else _f32c(bias)
        N, H = x.shape
        A, Ho = len(aggr_codes), weight.shape[0]
        if N != index.N:
            raise ValueError(f"x has {N} rows but the index was built for {index.N} nodes")
        if node_att is not None:
            if node_att.numel() != N:
                raise ValueError("node attention must have one entry per node")
            w, eid = _f32c(node_att).view(-1), None
        elif att is not None:
            w, eid = _flat_att(att, index.E), index.eid_by_dst
        else:
            w, eid = None, None
        a_arr = (ctypes.c_int32 * A)(*aggr_codes)
        dev, f32 = x.device, torch.float32
        aggj = torch.empty(N, A * H, dtype=f32, device=dev)
        scal = torch.empty(N, 8, dtype=f32, device=dev)
        call("gsat_pna_fwd_compact", ptr(x), ptr(w), ptr(index.rowptr_dst), ptr(index.src_by_dst), ptr(eid), N, H, a_arr, A, ptr(aggj), ptr(scal),
             stream())
        out = torch.empty(N, Ho, dtype=f32, device=dev)
        call("gsat_pna_post_fwd", ptr(x), ptr(aggj), ptr(scal), N, H, A, ptr(weight), weight.shape[1], ptr(bias), Ho,
             ptr(out), stream())
        ctx.save_for_backward(x, w if w is not None else x.new_empty(0), aggj, scal, weight)
        ctx.index, ctx.aggr = index, tuple(aggr_codes)
        ctx.mode = "node" if node_att is not None else ("edge" if att is not None else "none")
        ctx.att_shape = node_att.shape if node_att is not None else (att.shape if att is not None else None)
        ctx.has_bias = bias is not None
        return (out, x_in) if passthrough else out

    @staticmethod
    def backward(ctx, dout, dx_add=None):
        import ctypes
        from ._lib import load
        dx_add = None if dx_add is None else _f32c(dx_add)
        if dout is None:
            return dx_add, None, None, None, None, None, None, None, None
        x, w, aggj, scal, weight = ctx.saved_tensors
        index, aggr_codes = ctx.index, ctx.aggr
        dout = _f32c(dout)
        N, H = x.shape
        A, Ho = len(aggr_codes), weight.shape[0]
        F = A * 2 * H
        dev, f32 = x.device, torch.float32
        a_arr = (ctypes.c_int32 * A)(*aggr_codes)
        s_arr = (ctypes.c_int32 * 1)(0)
        dW = db = None
        if ctx.needs_input_grad[3]:
            dW = torch.empty_like(weight)
            wsf = int(load().gsat_pna_post_dw_workspace_floats(N, H, A, Ho))
            ws = torch.empty(wsf, dtype=f32, device=dev) if wsf else None
            call("gsat_pna_post_dw", ptr(x), ptr(aggj), ptr(scal), N, H, A, ptr(dout), Ho, ptr(dW), ptr(ws), wsf, stream())
        if ctx.has_bias and ctx.needs_input_grad[4]:
            db = colsum(dout)
        need_x = ctx.needs_input_grad[0]
        need_att = ctx.needs_input_grad[2] if ctx.mode == "node" else (ctx.needs_input_grad[1] if ctx.mode == "edge" else False)
        if not (need_x or need_att):
            return None, None, None, dW, db, None, None, None, None
        dagg = torch.empty(N, F, dtype=f32, device=dev)
        _gemm(0, 0, N, F, Ho, dout, Ho, weight, F, dagg, F)
        tile_ptr, T, rows_nominal, rows_cap, edges_cap, spill = index.pna_tiles(H)
        dmsg = torch.empty(max(index.E, 1), H, dtype=f32, device=dev)[: index.E]
        dx = torch.empty_like(x)
        datt = dna = None
        if ctx.mode == "node":
            dna, acc, ret = None, 0, None
            if need_att:
                if ctx.lifted is not None:
                    dna, acc, ret = ctx.lifted._shared_dna(N, dev)
                else:
                    dna = ret = torch.empty(N, dtype=f32, device=dev)
            dw = torch.empty(max(index.E, 1), dtype=f32, device=dev) if need_att else None
            call("gsat_pna_bwd_tiled_node_att", ptr(x), ptr(w), ptr(dagg), ptr(index.rowptr_dst), ptr(index.src_by_dst), ptr(tile_ptr), T,
                 rows_nominal, rows_cap, edges_cap, ptr(index.rowptr_src), ptr(index.slot_dst_of_srcslot), N, index.E, H, a_arr, A, s_arr, 1,
                 ptr(spill[1:]), ptr(spill[:1]), ptr(dx), ptr(dmsg), ptr(dna), ptr(dw), ptr(dx_add), acc, stream())
            dna = ret.view(ctx.att_shape) if ret is not None else None
        else:
            datt = torch.empty(index.E, dtype=f32, device=dev) if need_att else None
            call("gsat_pna_bwd_tiled", ptr(x), ptr(w) if ctx.mode == "edge" else None, ptr(dagg), ptr(index.rowptr_dst), ptr(index.src_by_dst),
                 ptr(index.eid_by_dst), ptr(tile_ptr), T, rows_nominal, rows_cap, edges_cap, ptr(index.rowptr_src),
                 ptr(index.slot_dst_of_srcslot), N, index.E, H, a_arr, A, s_arr, 1, ptr(spill[1:]), ptr(spill[:1]), ptr(dx), ptr(dmsg),
                 ptr(datt), ptr(dx_add), stream())
            datt = datt.view(ctx.att_shape) if need_att else None
        return dx, datt, dna, dW, db, None, None, None, None


def pna_conv(x, index, att, edge_emb, aggregators, scalers, avg_deg, weight, bias, passthrough: bool = False):
    """post_nn[0](pna_aggregate(...)) for a PNAConvSimple whose post_nn is one Linear, on the compact aggregate -- opt-in
    (GSAT_PNA_COMPACT=1): measured on MI355X at C3 (profiles/r03_summary.md) the aggregation forward drops from 45 to 24 us per layer
    pass, but the post_nn GEMMs, whose split-bf16 staging is already vector-ALU bound, pay more for rebuilding the x_i columns (forward 65-70
    -> 87 us, weight gradient 79 -> 96 us) than the halved traffic returns: whole step 4.69 vs 4.55 ms.  Returns None when not taken (the
    caller runs the two ops)."""
    a = [AGGREGATOR_CODES[k] for k in aggregators]
    s = [SCALER_CODES[k] for k in scalers]
    N, H = x.shape
    if not (x.is_cuda and N > 0 and edge_emb is None and _FIXED_PNA.get((tuple(a), tuple(s))) and H % 64 == 0 and H <= 256
            and weight.shape[1] == len(a) * 2 * H and weight.shape[0] % 4 == 0 and os.environ.get("GSAT_PNA_COMPACT", "0") == "1"
            and os.environ.get("GSAT_PNA_TILED", "1") != "0" and index.long_rows_nowait[0] is None
            and (bias is None or bias.data_ptr() % 16 == 0) and index.pna_tiles(H)):
        return None
    node_att = lifted = None
    if isinstance(att, LiftedAttention):
        if att.index is index and att._edge is None and os.environ.get("GSAT_NODE_ATT_LIFT", "0") != "1":
            node_att, lifted, att = att.shared_node_att, att, None
        else:
            att = att.edge()
    return PnaConvFn.apply(x, att, node_att, weight, bias, index, a, passthrough, lifted)


def linear(x, weight, bias=None):
    """nn.functional.linear for 2-D fp32 ROCm inputs with a 4-aligned output width; anything else goes to torch.  An input width that is
    not a multiple of 4 (raw node / edge features: 10, 14, 1 columns) is zero-padded: the library's weight gradient for such a layer
    -- a [out, 10] result reduced over 1e4..1e5 rows by ONE workgroup -- took 65 us per backbone pass at C2, the split-K kernel takes 5."""
    if x.is_cuda and x.dim() == 2 and x.dtype == torch.float32 and weight.shape[0] % 4 == 0 and x.shape[0] > 0:
        rem = x.shape[1] % 4
        if rem:
            x = torch.nn.functional.pad(x, (0, 4 - rem))
            weight = torch.nn.functional.pad(weight, (0, 4 - rem))
        return LinearFn.apply(x, weight, bias)
    return torch.nn.functional.linear(x, weight, bias)
