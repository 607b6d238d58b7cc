// Fused attention-extractor kernels (attn_fused.hip): geometry + entry points shared with attn.hip.
#pragma once
#include "common.h"

namespace gsat {

constexpr int F_GT = 16;            // graphs per tile (LDS statistics arrays are [2][F_GT][width])

struct FusedGeom {
    int H, C1, C2, C2p;
    int CH, NCH;                    // layer-1 output channels per chunk (128 node / 64 edge), chunks
    int S1, S2;                     // float4 steps of a B fragment stream: layer 1 (H/8), layer 2 (CH/8)
    int NCB2;                       // 32-column blocks of layer 2
    int NRB, RM, RX;                // row blocks per row wave, MLP rows per tile (64*NRB), embedding rows per tile
    int LDX, LDU, LDT, LDH, SW;     // LDS row strides (floats) and the width of a statistics row
    int offU, offT, offS, offMeta;  // LDS offsets (floats)
    int lds_bytes;
    int pqg;                        // edge mode: P | Q of a tile's nodes go through global memory (no LDS image; tiles limited by edges, not by 64 nodes)
    int x6;                         // layer products as split-bf16 x 6 (hi / mid / lo planes, six MFMAs per 16 k) instead of fp32 MFMA
};

// a tile of whole graphs: MLP rows [row0, row0 + nrows) of graphs [g0, g0 + ng), their nodes [node0, node0 + nnodes); flags & 1: one graph
// larger than a tile ("big": walked in slabs)
struct FTile { int row0, nrows, g0, ng, node0, nnodes, flags, pad; };
size_t attn_plan_bytes(int64_t G);
bool attn_plan_ok(int64_t G);
int attn_plan_launch(hipStream_t stream, const int32_t* seg_ptr, const int32_t* node_ptr, int64_t G, int RM, int RX, FTile* tiles, int* counters);

bool fused_geometry(int H, int C1, int C2, bool edge, FusedGeom* out, bool pqg = false);
size_t fused_ws_bytes(const FusedGeom& g, int64_t G);
bool attn_fused_eligible(const gsat_attn_args* a, FusedGeom* g);
int attn_fused_fwd(hipStream_t stream, const gsat_attn_args* a, const FusedGeom& g);

// fused backward (attn_fused_bwd.hip): (dlogits, datt) -> dh1 [M, C1], dW2, dW3, db3 (db1 = db2 = 0) in one launch + one reduction
bool attn_fused_bwd_eligible(const gsat_attn_args* a);
size_t attn_fused_bwd_ws_bytes(const gsat_attn_args* a);
int attn_fused_bwd(hipStream_t stream, const gsat_attn_args* a, const gsat_attn_grads* gr, float* dh1, void* ws, size_t ws_bytes);

// dual split-bf16 tile GEMMs (dual_gemm.hip): OUT [R, NO] (+)= A W and DW [KA, KY] = A^T Y in one pass over the rows
bool dual_gemm_ok(int mode, int64_t R, int KA, int KY, int NO);
size_t dual_gemm_ws_bytes(int mode, int KA, int KY, int NO);
int dual_gemm(hipStream_t stream, int mode, int64_t R, int KA, int KY, int NO, const float* A, int lda, const float* Y, int ldy, const float* W,
              int ldw, float* OUT, int ldo, int accumulate, float* DW, int lddw, void* ws, size_t ws_bytes);

#ifdef __HIPCC__
// dropout keep-mask of 4 consecutive channels: explicit tensor, or Philox keyed by (seed, layer, row, column) -- the same draw as
// gsat_philox_keep_mask and the unfused kernels of attn.hip
__device__ __forceinline__ float4 keep4f(const float* mask, SeedRef sref, int layer, int row, int c, int C, float p, bool training) {
    if (!training || p <= 0.f) return make_float4(1.f, 1.f, 1.f, 1.f);
    if (mask) return ld4(mask + (size_t)row * C + c);
    return philox_keep4(sref.get(), layer, row, c, p);
}
#endif

}  // namespace gsat
