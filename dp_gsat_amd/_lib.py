"""ctypes binding of libgsat_hip.so (the C ABI declared in include/gsat_hip.h).

There is deliberately NO fallback: if the library is missing or a tensor is not a contiguous
ROCm tensor, the call raises.  PyTorch is only used for device memory and streams.
"""
from __future__ import annotations

import ctypes
import os
from ctypes import c_char_p, c_float, c_int, c_int32, c_int64, c_size_t, c_uint64, c_void_p

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libgsat_hip.so")

P, I64, I32, F32, SZ, INT, U64 = c_void_p, c_int64, c_int32, c_float, c_size_t, c_int, c_uint64

# name -> (restype, argtypes); must list every symbol of include/gsat_hip.h (tests check this).
SIGNATURES = {
    "gsat_abi_version": (INT, []),
    "gsat_seed_next": (INT, [P, P, P]),
    "gsat_last_error": (c_char_p, []),
    "gsat_csr_workspace_bytes": (SZ, [I64, I64]),
    "gsat_rev_workspace_bytes": (SZ, [I64]),
    "gsat_build_csr": (INT, [P, P, I64, I64, P, P, P, P, P, SZ, P]),
    "gsat_reverse_edge_perm": (INT, [P, I64, I64, P, P, P, SZ, P]),
    "gsat_reverse_edge_perm_csr": (INT, [P, P, P, P, P, P, P, P, I64, I64, P, P, P]),
    "gsat_segment_ptr": (INT, [P, I64, I64, P, P, P]),
    "gsat_gather_i64": (INT, [P, I64, P, I64, P, P]),
    "gsat_row_chunks_workspace_bytes": (SZ, [I64]),
    "gsat_row_chunks": (INT, [P, I64, P, P, SZ, P]),
    "gsat_long_row_partial_floats": (SZ, [I64, I64]),
    "gsat_aggr_sum_fwd": (INT, [P, P, P, P, P, P, P, I64, I64, I64, F32, P, P, P, P]),
    "gsat_aggr_sum_bwd": (INT, [P, P, P, P, P, P, P, I64, I64, I64, F32, P, P, P, P, P, P]),
    "gsat_pna_fwd": (INT, [P, P, P, P, P, P, I64, I64, P, INT, P, INT, F32, F32, P, P]),
    "gsat_pna_bwd": (INT, [P, P, P, P, P, P, P, I64, I64, P, INT, P, INT, F32, F32, P, P, P, P, P]),
    "gsat_pna_long_row_floats": (SZ, [I64, I64, INT]),
    "gsat_pna_fwd_long": (INT, [P, P, P, P, P, P, I64, I64, I64, P, INT, P, INT, F32, F32, P, P, P, P]),
    "gsat_pna_bwd_long": (INT, [P, P, P, P, P, P, P, I64, I64, I64, P, INT, P, INT, F32, F32, P, P, P, P, P, P, P]),
    "gsat_pna_tile_plan": (INT, [I64, I64, P, P, P]),
    "gsat_pna_build_tiles": (INT, [P, P, P, P, P, I64, INT, INT, INT, P, P, P, P]),
    "gsat_pna_bwd_tiled": (INT, [P, P, P, P, P, P, P, I64, INT, INT, INT, P, P, I64, I64, I64, P, INT, P, INT, P, P, P, P, P, P, P]),
    "gsat_pna_fwd_compact": (INT, [P, P, P, P, P, I64, I64, P, INT, P, P, P]),
    "gsat_pna_post_fwd": (INT, [P, P, P, I64, I64, INT, P, I64, P, I64, P, P]),
    "gsat_pna_post_dw_workspace_floats": (SZ, [I64, I64, INT, I64]),
    "gsat_pna_post_dw": (INT, [P, P, P, I64, I64, INT, P, I64, P, P, SZ, P]),
    "gsat_pna_fwd_node_att": (INT, [P, P, P, P, I64, I64, P, INT, P, INT, F32, F32, P, P]),
    "gsat_pna_bwd_tiled_node_att": (INT, [P, P, P, P, P, P, I64, INT, INT, INT, P, P, I64, I64, I64, P, INT, P, INT, P, P, P, P, P, P, P, INT, P]),
    "gsat_csr_pair_workspace_bytes": (SZ, [I64, I64]),
    "gsat_build_csr_pair": (INT, [P, I64, I64] + [P] * 12 + [P, SZ, P]),
    "gsat_segment_ptr32": (INT, [P, I64, I64, P, P, P, P]),
    "gsat_bn_workspace_floats": (SZ, [I64, I64]),
    "gsat_bn_fwd": (INT, [P, P, P, P, P, I64, I64, INT, F32, F32, INT, P, P, P, P, P]),
    "gsat_bn_bwd": (INT, [P, P, P, P, P, P, I64, I64, INT, INT, P, P, P, P, P]),
    "gsat_bn_act_fwd": (INT, [P, P, P, P, P, I64, I64, INT, F32, F32, INT, P, F32, U64, P, P, P, P, P, P]),
    "gsat_bn_act_bwd": (INT, [P, P, P, P, P, P, I64, I64, INT, INT, F32, U64, P, P, P, P, P, P, P]),
    "gsat_bn_local_sum": (INT, [P, P, I64, I64, P, P, P]),
    "gsat_bn_apply_fwd": (INT, [P, P, P, P, P, I64, I64, INT, P, F32, U64, P, P, P]),
    "gsat_bn_local_bwd_sums": (INT, [P, P, P, P, P, P, I64, I64, INT, F32, U64, P, P, P, P, P]),
    "gsat_bn_apply_bwd": (INT, [P, P, P, P, P, P, P, P, I64, P, I64, I64, INT, F32, U64, P, P, P, P]),
    "gsat_relu_dropout_fwd": (INT, [P, I64, I64, F32, U64, P, P, P]),
    "gsat_relu_dropout_bwd": (INT, [P, P, I64, I64, F32, P, P]),
    "gsat_colsum_workspace_floats": (SZ, [I64]),
    "gsat_colsum": (INT, [P, I64, I64, P, P, P]),
    "gsat_embsum_fwd": (INT, [P, P, INT, P, I64, I64, P, P]),
    "gsat_onehot_rows": (INT, [P, P, INT, I64, I64, P, P]),
    "gsat_gemm_workspace_floats": (SZ, [INT, I64, I64, I64]),
    "gsat_gemm_f32": (INT, [INT, INT, I64, I64, I64, P, I64, P, I64, P, I64, P, INT, P, SZ, P]),
    "gsat_gemm_bf16x3": (INT, [INT, INT, I64, I64, I64, P, I64, P, I64, P, I64, P, INT, P, SZ, P]),
    "gsat_attn_fwd_workspace_bytes": (SZ, [P]),
    "gsat_attn_fwd_kind": (I32, [P]),
    "gsat_attn_bwd_workspace_bytes": (SZ, [P]),
    "gsat_attn_fwd": (INT, [P, P]),
    "gsat_attn_bwd": (INT, [P, P, P]),
    "gsat_instance_norm_fwd": (INT, [P, P, P, P, I64, I64, I64, P, P, P]),
    "gsat_instance_norm_bwd": (INT, [P, P, P, P, P, P, I64, I64, I64, P, P, P]),
    "gsat_philox_keep_mask": (INT, [U64, I32, I64, I64, F32, P, P]),
    "gsat_philox_noise": (INT, [U64, I64, P, P]),
    "gsat_sample_fwd": (INT, [P, P, INT, F32, F32, I64, P, P]),
    "gsat_sample_bwd": (INT, [P, P, F32, I64, P, P]),
    "gsat_lift_fwd": (INT, [P, P, P, I64, P, P]),
    "gsat_lift_bwd": (INT, [P, P, P, P, P, P, P, P, I64, P, P]),
    "gsat_symmetrise": (INT, [P, P, P, I64, P, P]),
    "gsat_info_loss_fwd": (INT, [P, P, F32, I64, P, P, P]),
    "gsat_info_loss_bwd": (INT, [P, P, F32, P, I64, P, P]),
    "gsat_collate": (INT, [P, I64, P, P, P, I64, P, P, I64, I64, P, P, P, P, P]),
    "gsat_line_graph_pair_counts": (INT, [P, I64, P, P]),
    "gsat_line_graph": (INT, [P, P, P, I64, I64, P, P]),
    "gsat_narrow_i64": (INT, [P, I64, P, P]),
    "gsat_und_edges_workspace_bytes": (SZ, [I64]),
    "gsat_und_edges": (INT, [P, I64, I64, P, P, P, P, P, P, P, P, SZ, P]),
    "gsat_und_line_graph_counts": (INT, [P, P, P, P, I64, P, P]),
    "gsat_und_line_graph": (INT, [P, P, P, P, P, I64, I64, P, P]),
    "gsat_segment_pool_fwd": (INT, [P, P, I64, I64, INT, P, P]),
    "gsat_segment_pool_bwd": (INT, [P, P, I64, I64, INT, P, P]),
}



class AttnArgs(ctypes.Structure):
    """mirror of `gsat_attn_args` (include/gsat_hip.h)."""
    _fields_ = [("M", I64), ("N", I64), ("G", I64), ("H", I32), ("C1", I32), ("C2", I32), ("edge_mode", I32),
                ("training", I32), ("p_drop", F32), ("seed", U64),
                ("src", P), ("dst", P), ("seg_ptr", P), ("seg_order", P), ("row_seg", P),
                ("W1", P), ("b1", P), ("W2", P), ("b2", P), ("W3", P), ("b3", P),
                ("emb", P), ("mask1", P), ("mask2", P), ("u", P),
                ("P", P), ("Q", P), ("a1", P), ("h2", P), ("stats", P), ("logits", P), ("att", P),
                ("fwd_workspace", P), ("fwd_workspace_bytes", SZ), ("seed_dev", P), ("noise_philox", I32), ("fused", I32),
                ("node_ptr", P)]


class AttnGrads(ctypes.Structure):
    """mirror of `gsat_attn_grads` (include/gsat_hip.h)."""
    _fields_ = [("dlogits", P), ("datt", P), ("rowptr_src", P), ("eid_by_src", P), ("rowptr_dst", P), ("eid_by_dst", P),
                ("chunk_ptr_src", P), ("chunk_ptr_dst", P),
                ("demb", P), ("dW1", P), ("db1", P), ("dW2", P), ("db2", P), ("dW3", P), ("db3", P),
                ("workspace", P), ("workspace_bytes", SZ)]


_lib = None


class GsatHipError(RuntimeError):
    pass


def load():
    """Load the shared library once; raise loudly if it is not built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise GsatHipError(
            f"{LIB_PATH} not found: the HIP extension is required (no CPU fallback). "
            "Build it with `python -c 'import __graft_entry__ as g; g.build()'` or `make -C dp_gsat_amd/csrc`.")
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def ptr(t):
    """Device pointer of a contiguous ROCm tensor (None -> NULL)."""
    if t is None:
        return None
    if not t.is_cuda:
        raise GsatHipError("dp_gsat_amd operators need ROCm (cuda) tensors: the HIP path has no CPU fallback")
    if not t.is_contiguous():
        raise GsatHipError("dp_gsat_amd operators need contiguous tensors")
    return t.data_ptr()


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)


def stream():
    """Raw handle of torch's current stream on the current device.  (torch.cuda.current_stream() builds a Stream object per call:
    ~10 us each, ~0.1 ms of a 1 ms C3 step.)"""
    if _raw_stream is not None:
        return _raw_stream(torch.cuda.current_device())
    return torch.cuda.current_stream().cuda_stream


def call(name, *args):
    lib = _lib if _lib is not None else load()
    rc = getattr(lib, name)(*args)
    if rc != 0:
        msg = lib.gsat_last_error()
        raise GsatHipError(f"{name} failed (code {rc}): {msg.decode() if msg else ''}")
    return rc
