// Fused attention extractor, forward (example/gsat.py:94-103,131-139; src/utils/get_model.py:47-68):
//   gather -> Linear -> per-graph InstanceNorm -> ReLU -> Dropout -> Linear -> InstanceNorm -> ReLU -> Dropout -> Linear(.,1) -> concrete sample
// in ONE launch.  A 512-thread workgroup owns a TILE of whole graphs (<= 128 MLP rows, <= 16 graphs) and keeps every
// intermediate of the tile on chip:
//   * the tile's embedding rows sit in LDS; layer 1 runs as exact-fp32 MFMA (v_mfma_f32_32x32x2_f32) in chunks of 128 (node mode) /
//     64 (edge mode) output channels straight into an LDS tile; edge mode evaluates it on the tile's NODES (P | Q) and forms
//     h1[e] = P[src] + Q[dst] + b1 from LDS;
//   * the per-graph two-pass mean / variance is taken FROM LDS (a thread owns a channel of a graph: no cross-thread reduction, fixed
//     order), normalise + ReLU + Philox dropout rewrite the chunk in place, and the chunk feeds layer 2's MFMAs as their k-slice; the
//     [rows, C2] accumulators stay in registers across the chunks;
//   * layer 2's tile goes through the same statistics, then the C2 -> 1 dot, the sampler, and out.
// Weights are re-laid once per call into per-wave fragment streams (k_attn_prep) so that a wave's B operand is one coalesced 1 KB load per
// four MFMAs from L2, prefetched three steps ahead; nothing is weight-stationary, so tiles are dealt dynamically from a queue.
// A graph that does not fit a tile ("big": more rows than a tile holds) is walked by ONE workgroup in slabs, with P / Q and h2 staged
// through global memory (same-workgroup write -> barrier -> read) -- correct for any size, fast for the occasional large molecule; batches
// of huge graphs (C5) never come here (attn.hip picks the streaming pipeline from the average graph size).
// What reaches HBM: P (| Q), h2, the [G, C] statistics (all needed by the backward), logits, att -- and a1 only when the caller
// asks for it (the unfused backward).
#include "common.h"
#include "attn_fused.h"
#include <algorithm>
#include <cstdlib>

namespace gsat {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int FT = 512;             // threads per workgroup: 8 waves = 4 column waves x 2 row waves
constexpr int F_RMAX = 128;         // rows of the largest tile
constexpr float F_EPS = 1e-5f;


// ------------------------------------------------------------------------------------------------
// geometry (host): LDS layout in floats.  node: [X | T | stats | meta]; edge: [X | U | T | stats | meta]; the layer-2 tile H2 overlays
// everything in front of the statistics.
// ------------------------------------------------------------------------------------------------
static inline int ceil32(int v) { return (v + 31) / 32 * 32; }

bool fused_geometry(int H, int C1, int C2, bool edge, FusedGeom* out, bool pqg) {
    if (H < 8 || H > 256 || H % 8 || C1 < 4 || C1 % 4 || C2 < 4 || C2 % 4 || C2 > 256 || C1 > 2048) return false;
    for (int nrb = 2; nrb >= 1; --nrb) {
        FusedGeom g{};
        g.H = H; g.C1 = C1; g.C2 = C2; g.C2p = ceil32(C2);
        g.CH = edge ? 64 : 128;
        g.NCH = (C1 + g.CH - 1) / g.CH;
        g.S1 = H / 8; g.S2 = g.CH / 8;
        g.NCB2 = g.C2p / 32;
        g.NRB = nrb;
        g.RM = 64 * nrb;
        g.RX = edge ? (pqg ? g.RM : 64) : g.RM;
        g.pqg = (edge && pqg) ? 1 : 0;
        g.LDX = H + 4; g.LDU = 2 * g.CH + 4; g.LDT = g.CH + 4; g.LDH = g.C2p + 4;
        g.SW = std::max(g.CH, g.C2p);
        g.offU = g.RX * g.LDX;
        g.offT = g.offU + ((edge && !pqg) ? 64 * g.LDU : 0);
        g.offS = std::max(g.offT + g.RM * g.LDT, g.RM * g.LDH);
        const int stats = std::max(2 * F_GT * g.SW, 2 * std::max(C1, C2));      // small tiles: [2][GT][SW]; a big graph: [2][C]
        g.offMeta = g.offS + stats;
        g.lds_bytes = (g.offMeta + 32 + 4 * F_RMAX + 16 + 2 * g.C2p + g.NCH * g.CH) * 4;       // meta ints, then b2 | w3 | b1 (zero-padded)
        if (g.lds_bytes <= 160 * 1024) { *out = g; return true; }
    }
    return false;
}

size_t fused_ws_bytes(const FusedGeom& g, int64_t G) {
    size_t b = 256;                                                                 // counters
    b += align_up((size_t)(G + 1 + (G + 63) / 64 * 64) * sizeof(FTile), 256);       // compact tile list + the planner's sparse scratch
    b += align_up((size_t)g.NCH * 4 * g.S1 * 64 * 24, 256);                          // Wp1 (fp32 fragments, or 3 bf16 planes: 1.5 x)
    b += align_up((size_t)g.NCH * g.NCB2 * g.S2 * 64 * 24, 256);                     // Wp2
    return b;
}

// ------------------------------------------------------------------------------------------------
// prep launch: block 0 plans the tiles, the other blocks re-lay the weights into fragment streams
//   Wp1[(kc*4 + cb)*S1 + s][lane] = W1[col(kc,cb,lane&31)][koff + h*(H/2) + 4s .. +3]      (h = lane >> 5)
//   Wp2[(kc*NCB2 + cb)*S2 + s][lane] = W2[cb*32 + (lane&31)][kc*CH + h*(CH/2) + 4s .. +3]
// zero outside the matrices, so padded columns / k produce exact zeros.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ int wave_incl_scan(int v) {
    const int l = threadIdx.x & 63;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int t = __shfl_up(v, o, 64);
        if (l >= o) v += t;
    }
    return v;
}

// one superblock of <= 64 consecutive graphs, one wave; WRITE = false: count tiles only.  Greedy: a tile takes graphs while rows <= RM,
// nodes <= RX and graphs <= F_GT; a graph that alone exceeds a tile becomes a "big" tile of its own.
// Tiles of superblock `sb` go to sparse slots [sb*64, sb*64 + count); returns (#big, #small) tiles; a tile's `pad` word = its rank among the
// superblock's tiles of its kind (for the compaction pass).
__device__ __forceinline__ void plan_superblock(const int32_t* __restrict__ seg_ptr, const int32_t* __restrict__ node_ptr, int G, int sb, int RM,
                                                int RX, int* nbig, int* nsmall, FTile* __restrict__ sparse) {
    const int l = threadIdx.x & 63;
    const int g = sb * 64 + l;
    const int cnt = min(64, G - sb * 64);
    const int b_m = g < G ? seg_ptr[g] : 0, e_m = g < G ? seg_ptr[g + 1] : 0;
    const int b_x = g < G ? node_ptr[g] : 0, e_x = g < G ? node_ptr[g + 1] : 0;
    const int pm = wave_incl_scan(e_m - b_m), px = wave_incl_scan(e_x - b_x);
    int cur = 0, base_m = 0, base_x = 0, kb = 0, ks = 0;
    while (cur < cnt) {
        const bool ok = l >= cur && l < cnt && pm - base_m <= RM && px - base_x <= RX && l - cur < F_GT;
        const unsigned long long mask = __ballot(ok);
        int n = __popcll(mask);
        const bool big = n == 0;
        if (big) n = 1;
        const int last = cur + n - 1;                                   // wave-uniform
        const int nm = __builtin_amdgcn_readlane(pm, last), nx = __builtin_amdgcn_readlane(px, last);
        if (l == cur) {
            FTile t;
            t.row0 = b_m; t.nrows = nm - base_m; t.g0 = g; t.ng = n; t.node0 = b_x; t.nnodes = nx - base_x; t.flags = big ? 1 : 0;
            t.pad = big ? kb : ks;
            sparse[sb * 64 + kb + ks] = t;
        }
        if (big) ++kb; else ++ks;
        base_m = nm; base_x = nx; cur = last + 1;
    }
    *nbig = kb; *nsmall = ks;
}

constexpr int PLAN_T = 1024;
constexpr int PLAN_MAX_SB = 4096;

// Plans the tiles of a batch: called by ALL threads of ONE 1024-thread block.  counters[0] = tiles, [1] = 0 (queue head), [2] = big tiles.
__device__ void plan_tiles_block(const int32_t* __restrict__ seg_ptr, const int32_t* __restrict__ node_ptr, int G, int RM, int RX,
                                 FTile* __restrict__ tiles, int* __restrict__ counters) {
        __shared__ int sBig[PLAN_MAX_SB], sSmall[PLAN_MAX_SB], sBigCnt[PLAN_MAX_SB];
        __shared__ int sTot[2];
        const int nsb = (G + 63) / 64;
        const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
        FTile* const sparse = tiles + G + 1;                                  // [nsb * 64] scratch behind the compact list
        for (int sb = wave; sb < nsb; sb += PLAN_T / 64) {
            int nb, ns;
            plan_superblock(seg_ptr, node_ptr, G, sb, RM, RX, &nb, &ns, sparse);
            if (lane == 0) { sBig[sb] = nb; sSmall[sb] = ns; }
        }
        __syncthreads();
        if (wave == 0) {                      // exclusive scans of both count arrays (nsb <= 4096: 64 lanes x 64 entries)
            const int per = (nsb + 63) / 64;
            int tb = 0, ts = 0;
            for (int i = 0; i < per; ++i) { const int k = lane * per + i; if (k < nsb) { tb += sBig[k]; ts += sSmall[k]; } }
            const int ib = wave_incl_scan(tb), is = wave_incl_scan(ts);
            int ob = ib - tb, os = is - ts;
            for (int i = 0; i < per; ++i) {
                const int k = lane * per + i;
                if (k < nsb) { const int a = sBig[k], b = sSmall[k]; sBig[k] = ob; sBigCnt[k] = a; sSmall[k] = os; ob += a; os += b; }
            }
            if (lane == 63) { sTot[0] = ib; sTot[1] = is; }
        }
        __syncthreads();
        const int nbig = sTot[0];
        // compaction: big tiles lead the list (their workgroups start first), small tiles follow in batch order
        for (int i = threadIdx.x; i < nsb * 64; i += PLAN_T) {
            const int sb = i >> 6, k = i & 63;
            const int cb = sBigCnt[sb], ob = sBig[sb];
            const int cs = ((sb + 1 < nsb) ? sSmall[sb + 1] : sTot[1]) - sSmall[sb];
            if (k < cb + cs) {
                const FTile t = sparse[i];
                tiles[(t.flags & 1) ? ob + t.pad : nbig + sSmall[sb] + t.pad] = t;
            }
        }
        if (threadIdx.x == 0) { counters[0] = sTot[0] + sTot[1]; counters[1] = 0; counters[2] = sTot[0]; }
        if (threadIdx.x >= 16 && threadIdx.x < 48) counters[threadIdx.x] = 0;       // diagnostic stamp words (GSAT_FUSED_STAMPS builds)
}

__global__ __launch_bounds__(PLAN_T) void k_attn_plan(const int32_t* __restrict__ seg_ptr, const int32_t* __restrict__ node_ptr, int G, int RM, int RX,
                                                      FTile* __restrict__ tiles, int* __restrict__ counters) {
    plan_tiles_block(seg_ptr, node_ptr, G, RM, RX, tiles, counters);
}

__global__ __launch_bounds__(PLAN_T) void k_attn_prep(const float* __restrict__ W1, const float* __restrict__ W2, FusedGeom g, int edge,
                                                      const int32_t* __restrict__ seg_ptr, const int32_t* __restrict__ node_ptr, int G,
                                                      FTile* __restrict__ tiles, int* __restrict__ counters, float4* __restrict__ Wp1,
                                                      float4* __restrict__ Wp2) {
    if (blockIdx.x == 0) {
        plan_tiles_block(seg_ptr, node_ptr, G, g.RM, g.RX, tiles, counters);
        return;
    }
    const int K1 = edge ? 2 * g.H : g.H;
    if (g.x6) {
        // split-bf16 x 6: three planes per fragment (top 16 bits of x, of x - hi, of x - hi - mid: an exact 8 + 8 + 8-bit decomposition),
        // Wx[((stream*S + s)*3 + plane)*64 + lane] = 8 bf16: k = 16 s + 8 (lane >> 5) + j of column (lane & 31)
        const int S1x = g.H / 16, S2x = g.CH / 16;
        const int64_t m1 = (int64_t)g.NCH * 4 * S1x * 3 * 64, m2 = (int64_t)g.NCH * g.NCB2 * S2x * 3 * 64;
        uint4* const X1 = reinterpret_cast<uint4*>(Wp1);
        uint4* const X2 = reinterpret_cast<uint4*>(Wp2);
        for (int64_t i = (int64_t)(blockIdx.x - 1) * PLAN_T + threadIdx.x; i < m1 + m2; i += (int64_t)(gridDim.x - 1) * PLAN_T) {
            const bool first = i < m1;
            const int64_t j = first ? i : i - m1;
            const int lane = (int)(j & 63), c = lane & 31, h = lane >> 5;
            const int plane = (int)((j >> 6) % 3);
            const int64_t q = (j >> 6) / 3;
            const float* src = nullptr;
            if (first) {
                const int s = (int)(q % S1x), st = (int)(q / S1x), cb = st & 3, kc = st >> 2;
                const int k = 16 * s + 8 * h;
                int col, koff = 0;
                if (edge) { col = kc * 64 + (cb & 1) * 32 + c; koff = (cb >> 1) * g.H; }
                else col = kc * 128 + cb * 32 + c;
                if (col < g.C1) src = W1 + (size_t)col * K1 + koff + k;
            } else {
                const int s = (int)(q % S2x), st = (int)(q / S2x), cb = st % g.NCB2, kc = st / g.NCB2;
                const int col = cb * 32 + c, k = kc * g.CH + 16 * s + 8 * h;
                if (col < g.C2 && k < g.C1) src = W2 + (size_t)col * g.C1 + k;          // C1 % 8 may be 4: the tail is zero-filled below
            }
            unsigned w[4] = {0u, 0u, 0u, 0u};
            if (src) {
                float v[8];
                const int kk = first ? 0 : (int)(((j >> 6) / 3 / S2x) / g.NCB2) * g.CH + 16 * (int)(((j >> 6) / 3) % S2x) + 8 * h;
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = (first || kk + e < g.C1) ? src[e] : 0.f;
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    unsigned u = __builtin_bit_cast(unsigned, v[e]);
                    if (plane >= 1) { const float r1 = v[e] - __builtin_bit_cast(float, u & 0xFFFF0000u); u = __builtin_bit_cast(unsigned, r1);
                        if (plane == 2) { const float r2 = r1 - __builtin_bit_cast(float, u & 0xFFFF0000u); u = __builtin_bit_cast(unsigned, r2); } }
                    w[e >> 1] |= (u >> 16) << (16 * (e & 1));
                }
            }
            (first ? X1 : X2)[j] = make_uint4(w[0], w[1], w[2], w[3]);
        }
        return;
    }
    const int64_t n1 = (int64_t)g.NCH * 4 * g.S1 * 64, n2 = (int64_t)g.NCH * g.NCB2 * g.S2 * 64;
    for (int64_t i = (int64_t)(blockIdx.x - 1) * PLAN_T + threadIdx.x; i < n1 + n2; i += (int64_t)(gridDim.x - 1) * PLAN_T) {
        float4 v = f4zero();
        if (i < n1) {
            const int lane = (int)(i & 63), c = lane & 31, h = lane >> 5;
            const int s = (int)((i >> 6) % g.S1);
            const int st = (int)((i >> 6) / g.S1), cb = st & 3, kc = st >> 2;
            const int k = h * (g.H / 2) + 4 * s;
            int col, koff = 0;
            if (edge) { col = kc * 64 + (cb & 1) * 32 + c; koff = (cb >> 1) * g.H; }
            else col = kc * 128 + cb * 32 + c;
            if (col < g.C1) v = ld4(W1 + (size_t)col * K1 + koff + k);
            Wp1[i] = v;
        } else {
            const int64_t j = i - n1;
            const int lane = (int)(j & 63), c = lane & 31, h = lane >> 5;
            const int s = (int)((j >> 6) % g.S2);
            const int st = (int)((j >> 6) / g.S2), cb = st % g.NCB2, kc = st / g.NCB2;
            const int col = cb * 32 + c, k = kc * g.CH + h * (g.CH / 2) + 4 * s;
            if (col < g.C2 && k < g.C1) v = ld4(W2 + (size_t)col * g.C1 + k);
            Wp2[j] = v;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// MFMA core: acc[i] += A_i (32 rows in LDS, this lane's row and k-half already applied) x B (this wave's fragment stream in global)
// ------------------------------------------------------------------------------------------------
#define GSAT_MFMA(a, b, c) __builtin_amdgcn_mfma_f32_32x32x2f32((a), (b), (c), 0, 0, 0)

template <int NA>
__device__ __forceinline__ void mma_run(f32x16& acc0, f32x16& acc1, const float* a0, const float* a1, const float4* __restrict__ bp, const int S) {
    float4 b[4];
    b[0] = bp[0];
    b[1] = bp[64 * min(1, S - 1)];
    b[2] = bp[64 * min(2, S - 1)];
    for (int s = 0; s < S; s += 4) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if (s + j < S) {
                b[(j + 3) & 3] = bp[64 * min(s + j + 3, S - 1)];
                const float4 x0 = ld4(a0 + 4 * (s + j));
                float4 x1 = f4zero();
                if (NA > 1) x1 = ld4(a1 + 4 * (s + j));
                const float4 w = b[j];
                acc0 = GSAT_MFMA(x0.x, w.x, acc0);
                if (NA > 1) acc1 = GSAT_MFMA(x1.x, w.x, acc1);
                acc0 = GSAT_MFMA(x0.y, w.y, acc0);
                if (NA > 1) acc1 = GSAT_MFMA(x1.y, w.y, acc1);
                acc0 = GSAT_MFMA(x0.z, w.z, acc0);
                if (NA > 1) acc1 = GSAT_MFMA(x1.z, w.z, acc1);
                acc0 = GSAT_MFMA(x0.w, w.w, acc0);
                if (NA > 1) acc1 = GSAT_MFMA(x1.w, w.w, acc1);
            }
        }
    }
}

typedef __attribute__((ext_vector_type(8))) __bf16 fbf16x8;
typedef __attribute__((ext_vector_type(4))) unsigned fu32x4;
#define GSAT_MFMA_B(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_bf16((a), (b), (c), 0, 0, 0)

// x (8 fp32) -> three bf16 fragments whose sum is x exactly: top 16 bits of x, of x - hi, of x - hi - mid (truncation keeps every
// remainder exactly representable, so 8 + 8 + 8 mantissa bits are covered).  ~4.5 VALU ops per element.
__device__ __forceinline__ void split3(const float4 lo4, const float4 hi4, fbf16x8& p0, fbf16x8& p1, fbf16x8& p2) {
    const float x[8] = {lo4.x, lo4.y, lo4.z, lo4.w, hi4.x, hi4.y, hi4.z, hi4.w};
    fu32x4 a, b, c;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const unsigned u0 = __builtin_bit_cast(unsigned, x[2 * e]), u1 = __builtin_bit_cast(unsigned, x[2 * e + 1]);
        const float r0 = x[2 * e] - __builtin_bit_cast(float, u0 & 0xFFFF0000u), r1 = x[2 * e + 1] - __builtin_bit_cast(float, u1 & 0xFFFF0000u);
        const unsigned v0 = __builtin_bit_cast(unsigned, r0), v1 = __builtin_bit_cast(unsigned, r1);
        const float s0 = r0 - __builtin_bit_cast(float, v0 & 0xFFFF0000u), s1 = r1 - __builtin_bit_cast(float, v1 & 0xFFFF0000u);
        a[e] = __builtin_amdgcn_perm(u1, u0, 0x07060302u);
        b[e] = __builtin_amdgcn_perm(v1, v0, 0x07060302u);
        c[e] = __builtin_amdgcn_perm(__builtin_bit_cast(unsigned, s1), __builtin_bit_cast(unsigned, s0), 0x07060302u);
    }
    p0 = __builtin_bit_cast(fbf16x8, a); p1 = __builtin_bit_cast(fbf16x8, b); p2 = __builtin_bit_cast(fbf16x8, c);
}

// the same products on the bf16 matrix pipe: A (fp32 rows in LDS, this lane's row and its 8-k half applied) split in registers, B as three
// pre-split planes in the fragment stream; per 16 k six MFMAs (x0 y0 + x0 y1 + x1 y0 + x0 y2 + x2 y0 + x1 y1, small terms first) = 192
// cycles against 512 for eight fp32 MFMAs; the dropped terms are <= 2^-24 of |x||y|
template <int NA>
__device__ __forceinline__ void mma_run_x6(f32x16& acc0, f32x16& acc1, const float* a0, const float* a1, const uint4* __restrict__ bp, const int S) {
    uint4 b[3][3];
#pragma unroll
    for (int p = 0; p < 3; ++p) { b[0][p] = bp[p * 64]; b[1][p] = bp[(3 * min(1, S - 1) + p) * 64]; }
    for (int s = 0; s < S; s += 3) {
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            if (s + j < S) {
                const int sn = min(s + j + 2, S - 1);
#pragma unroll
                for (int p = 0; p < 3; ++p) b[(j + 2) % 3][p] = bp[(3 * sn + p) * 64];
                const fbf16x8 y0 = __builtin_bit_cast(fbf16x8, b[j][0]), y1 = __builtin_bit_cast(fbf16x8, b[j][1]), y2 = __builtin_bit_cast(fbf16x8, b[j][2]);
                fbf16x8 x0, x1, x2;
                split3(ld4(a0 + 16 * (s + j)), ld4(a0 + 16 * (s + j) + 4), x0, x1, x2);
                acc0 = GSAT_MFMA_B(x2, y0, acc0);
                acc0 = GSAT_MFMA_B(x0, y2, acc0);
                acc0 = GSAT_MFMA_B(x1, y1, acc0);
                acc0 = GSAT_MFMA_B(x1, y0, acc0);
                acc0 = GSAT_MFMA_B(x0, y1, acc0);
                acc0 = GSAT_MFMA_B(x0, y0, acc0);
                if (NA > 1) {
                    split3(ld4(a1 + 16 * (s + j)), ld4(a1 + 16 * (s + j) + 4), x0, x1, x2);
                    acc1 = GSAT_MFMA_B(x2, y0, acc1);
                    acc1 = GSAT_MFMA_B(x0, y2, acc1);
                    acc1 = GSAT_MFMA_B(x1, y1, acc1);
                    acc1 = GSAT_MFMA_B(x1, y0, acc1);
                    acc1 = GSAT_MFMA_B(x0, y1, acc1);
                    acc1 = GSAT_MFMA_B(x0, y0, acc1);
                }
            }
        }
    }
}

__device__ __forceinline__ void acc_zero(f32x16& a) {
#pragma unroll
    for (int r = 0; r < 16; ++r) a[r] = 0.f;
}

__device__ __forceinline__ float4 f4add(float4 a, float4 b) { return make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }

// one element of P (or Q) for a node outside the tile (malformed batches only: an edge whose endpoints lie in different graphs)
__device__ float4 slow_pq4(const float* __restrict__ emb_row, const float* __restrict__ W1, int K1, int koff, int col, int C1, int H) {
    float r[4] = {0.f, 0.f, 0.f, 0.f};
    for (int j = 0; j < 4; ++j) {
        if (col + j >= C1) break;
        const float* w = W1 + (size_t)(col + j) * K1 + koff;
        float acc = 0.f;
        for (int k = 0; k < H; ++k) acc = fmaf(emb_row[k], w[k], acc);
        r[j] = acc;
    }
    return make_float4(r[0], r[1], r[2], r[3]);
}

// Barrier for LDS hand-offs inside a tile: waits for this wave's LDS traffic only.  __syncthreads() also drains every outstanding global
// store (vmcnt(0)) in front of the barrier -- the a1 / P / h2 / statistics stores of a phase then cost an HBM write latency per barrier.
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// mean / 1/sigma of rows [b, e) of one LDS column (stride LD floats), two-pass exactly as PyG's InstanceNorm (mean, then the variance of
// the centred values); the first RC rows stay in registers between the passes, so a molecule-sized graph is read once, with all its
// loads in flight together.  Sequential summation order.
template <int RC>
__device__ __forceinline__ void column_stats(const float* col, const int LD, float bias, int b, int e, float* mean, float* rstd) {
    const float inv_n = 1.f / (float)max(e - b, 1);
    const float* p = col + b * LD;
    const int lastoff = max(e - 1 - b, 0) * LD;
    float v[RC];
    // every load is issued unconditionally (row offset clamped into the segment, no multiply per row) so that all RC are in flight
    // together; a predicated load per element compiles to a branch + wait per element (32 serial LDS round trips per column)
#pragma unroll
    for (int j = 0; j < RC; ++j) v[j] = p[min(j * LD, lastoff)];
#pragma unroll
    for (int j = 0; j < RC; ++j) v[j] = (b + j < e) ? v[j] + bias : 0.f;
    // four interleaved partial sums (rows j, j+4, ...): fixed order, a quarter of the dependent-add latency
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
#pragma unroll
    for (int j = 0; j < RC; j += 4) { s0 += v[j]; s1 += v[j + 1]; s2 += v[j + 2]; s3 += v[j + 3]; }
    float s;
    // longer graphs (edge mode: ~2 edges per atom): the rest in batches of 16 loads, again unconditional and all in flight
    const int n = e - b;
    for (int r0 = RC; r0 < n; r0 += 16) {
        float w[16];
#pragma unroll
        for (int j = 0; j < 16; ++j) w[j] = p[min((r0 + j) * LD, lastoff)];
#pragma unroll
        for (int j = 0; j < 16; j += 4) {
            s0 += (r0 + j < n) ? w[j] + bias : 0.f; s1 += (r0 + j + 1 < n) ? w[j + 1] + bias : 0.f;
            s2 += (r0 + j + 2 < n) ? w[j + 2] + bias : 0.f; s3 += (r0 + j + 3 < n) ? w[j + 3] + bias : 0.f;
        }
    }
    s = (s0 + s1) + (s2 + s3);
    const float mu = s * inv_n;
    float q0 = 0.f, q1 = 0.f, q2 = 0.f, q3 = 0.f;
#pragma unroll
    for (int j = 0; j < RC; j += 4) {
        const float d0 = (b + j < e) ? v[j] - mu : 0.f, d1 = (b + j + 1 < e) ? v[j + 1] - mu : 0.f;
        const float d2 = (b + j + 2 < e) ? v[j + 2] - mu : 0.f, d3 = (b + j + 3 < e) ? v[j + 3] - mu : 0.f;
        q0 = fmaf(d0, d0, q0); q1 = fmaf(d1, d1, q1); q2 = fmaf(d2, d2, q2); q3 = fmaf(d3, d3, q3);
    }
    for (int r0 = RC; r0 < n; r0 += 16) {
        float w[16];
#pragma unroll
        for (int j = 0; j < 16; ++j) w[j] = p[min((r0 + j) * LD, lastoff)];
#pragma unroll
        for (int j = 0; j < 16; j += 4) {
            const float d0 = (r0 + j < n) ? (w[j] + bias) - mu : 0.f, d1 = (r0 + j + 1 < n) ? (w[j + 1] + bias) - mu : 0.f;
            const float d2 = (r0 + j + 2 < n) ? (w[j + 2] + bias) - mu : 0.f, d3 = (r0 + j + 3 < n) ? (w[j + 3] + bias) - mu : 0.f;
            q0 = fmaf(d0, d0, q0); q1 = fmaf(d1, d1, q1); q2 = fmaf(d2, d2, q2); q3 = fmaf(d3, d3, q3);
        }
    }
    const float q = (q0 + q1) + (q2 + q3);
    *mean = mu;
    *rstd = 1.f / sqrtf(q * inv_n + F_EPS);
}

struct FusedArgs {
    const float *emb, *W1, *b1, *b2, *w3, *b3;
    const float4 *Wp1, *Wp2;
    const int32_t *src, *dst, *seg_ptr, *order;
    const float *mask1, *mask2, *u;
    float *P, *Q, *a1, *h2, *stats, *logits, *att;
    const FTile* tiles;
    int* counters;
    int64_t M, N, G;
    SeedRef seed;
    float p;
    int training, noise_philox;
    FusedGeom g;
};

template <bool EDGE, int NRB, int NCB2W, bool X6, bool PQG>
__global__ __launch_bounds__(FT, 2) void k_attn_fused_fwd(const FusedArgs A) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const FusedGeom& g = A.g;
    constexpr int CH = EDGE ? 64 : 128, LDT = CH + 4, LDU = 2 * CH + 4, S2 = CH / 8;      // compile-time strides: LDS offsets become immediates
    const int H = g.H, C1 = g.C1, C2 = g.C2, LDX = g.LDX, LDH = g.LDH, SW = g.SW;
    float* const X = lds;
    float* const U = lds + g.offU;
    float* const T = lds + g.offT;
    float* const H2 = lds;
    float* const sMean = lds + g.offS;
    int* const meta = reinterpret_cast<int*>(lds + g.offMeta);
    int* const sGptr = meta;                          // [F_GT + 1], tile-local row offsets of the tile's graphs
    int* const sRowG = meta + 32;                     // [RM] tile-local graph of a row
    int* const sRowId = sRowG + F_RMAX;               // [RM] global MLP row (edge id / node id)
    int* const sSrcL = sRowId + F_RMAX;               // [RM] edge mode: tile-local source node, or -(global + 1)
    int* const sDstL = sSrcL + F_RMAX;
    int* const sCtl = sDstL + F_RMAX;
    float* const sB2 = reinterpret_cast<float*>(sCtl + 16);     // [C2p] layer-2 bias, then [C2p] head weights: read by every tile
    float* const sW3 = sB2 + g.C2p;
    float* const sB1 = sW3 + g.C2p;                              // [NCH * CH] layer-1 bias
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int cw = wave & 3, rw = wave >> 2, c = lane & 31, h = lane >> 5;
    const int RM = 64 * NRB;
    const int64_t G = A.G;
    float* const mean1 = A.stats;
    float* const rstd1 = mean1 + (size_t)G * C1;
    float* const mean2 = rstd1 + (size_t)G * C1;
    float* const rstd2 = mean2 + (size_t)G * C2;
    const float sc = (A.training && A.p > 0.f) ? 1.f / (1.f - A.p) : 1.f;
    const bool train = A.training != 0;
    const int ntiles = A.counters[0];
    const int K1 = EDGE ? 2 * H : H;
    const int H4 = H >> 2, C24 = C2 >> 2;
    constexpr int CH4 = CH / 4;

    for (int i = tid; i < g.C2p; i += FT) { sB2[i] = i < C2 ? A.b2[i] : 0.f; sW3[i] = i < C2 ? A.w3[i] : 0.f; }
    for (int i = tid; i < g.NCH * g.CH; i += FT) sB1[i] = i < C1 ? A.b1[i] : 0.f;
#ifdef GSAT_FUSED_STAMPS
    long long stamp[14] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    long long tlast = __builtin_amdgcn_s_memtime();
#define STAMP(i) do { if (tid == 0) { const long long n_ = __builtin_amdgcn_s_memtime(); stamp[i] += n_ - tlast; tlast = n_; } } while (0)
#else
#define STAMP(i)
#endif
    // tiles are dealt round-robin (big graphs lead the list): no queue traffic, and the next descriptor is fetched a tile ahead
    FTile nxt = A.tiles[min((int)blockIdx.x, max(ntiles - 1, 0))];
    for (int t = blockIdx.x; t < ntiles; t += gridDim.x) {
        lds_barrier();
        const FTile tl = nxt;
        nxt = A.tiles[min(t + (int)gridDim.x, ntiles - 1)];
        const bool big = (tl.flags & 1) != 0;
        const int nslab_m = big ? (tl.nrows + RM - 1) / RM : 1;
        if (tid <= tl.ng) sGptr[tid] = A.seg_ptr[tl.g0 + tid] - tl.row0;
        if (tid == 0) sCtl[1] = 0;                     // set by the meta fill when an edge of the tile has an endpoint outside it

        // =========================== big graph, pass 0: P (| Q) of all its nodes -> global ===========================
        // (small tiles run the same GEMM1 code on their single slab inside the chunk loop below)
        auto load_x = [&](int x0, int nx) {            // rows [x0, x0 + nx) of emb -> X, zero-filled to a multiple of 32 rows
            // four 16-byte loads per thread in flight, issued unconditionally (row clamped, zeroed at the LDS store): a load under
            // `r < nx ? load : 0` compiles to a branch that waits for every load before the next one is issued
            const int fill = min((nx + 31) & ~31, EDGE ? g.RX : RM);
            for (int i0 = tid; i0 < fill * H4; i0 += 4 * FT) {
                float4 v[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int i = min(i0 + u * FT, fill * H4 - 1), r = i / H4, q = i - r * H4;
                    v[u] = ld4(A.emb + (size_t)(x0 + min(r, nx - 1)) * H + 4 * q);
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int i = i0 + u * FT, r = i / H4, q = i - r * H4;
                    if (i < fill * H4) st4(X + r * LDX + 4 * q, r < nx ? v[u] : f4zero());
                }
            }
        };
        // layer 1 of chunk kc on the rows in X -> LDS without the bias: node mode T[row][CH], edge mode U[node][P half | Q half]
        auto gemm1 = [&](int kc, int nx) {
            f32x16 acc0, acc1;
            acc_zero(acc0); acc_zero(acc1);
            const float4* bp = A.Wp1 + ((size_t)(kc * 4 + cw) * g.S1) * 64 + lane;
            const uint4* bx = reinterpret_cast<const uint4*>(A.Wp1) + ((size_t)(kc * 4 + cw) * (H / 16)) * 192 + lane;       // split-bf16 stream
            if (EDGE) {
                const int rb = rw;
                if (rb * 32 < nx) {
                    if (X6) mma_run_x6<1>(acc0, acc1, X + (rb * 32 + c) * LDX + 8 * h, nullptr, bx, H / 16);
                    else mma_run<1>(acc0, acc1, X + (rb * 32 + c) * LDX + h * (H / 2), nullptr, bp, g.S1);
                    float* const o = U + (rb * 32 + 4 * h) * LDU + cw * 32 + c;
#pragma unroll
                    for (int r = 0; r < 16; ++r) o[((r & 3) + 8 * (r >> 2)) * LDU] = acc0[r];
                }
            } else {
                const int rb0 = rw * NRB;
                const int na = min(NRB, (nx - rb0 * 32 + 31) / 32);
                if (na > 0) {
                    if (X6) {
                        const float* a0 = X + (rb0 * 32 + c) * LDX + 8 * h;
                        if (NRB > 1 && na > 1) mma_run_x6<2>(acc0, acc1, a0, a0 + 32 * LDX, bx, H / 16);
                        else mma_run_x6<1>(acc0, acc1, a0, nullptr, bx, H / 16);
                    } else {
                    const float* a0 = X + (rb0 * 32 + c) * LDX + h * (H / 2);
                    if (NRB > 1 && na > 1) mma_run<2>(acc0, acc1, a0, a0 + 32 * LDX, bp, g.S1);
                    else mma_run<1>(acc0, acc1, a0, nullptr, bp, g.S1);
                    }
                    float* const o = T + (rb0 * 32 + 4 * h) * LDT + cw * 32 + c;
#pragma unroll
                    for (int r = 0; r < 16; ++r) o[((r & 3) + 8 * (r >> 2)) * LDT] = acc0[r];
                    if (NRB > 1 && na > 1) {
#pragma unroll
                        for (int r = 0; r < 16; ++r) o[(32 + (r & 3) + 8 * (r >> 2)) * LDT] = acc1[r];
                    }
                }
            }
        };
        // edge mode, P | Q through global memory (g.pqg): layer 1 of chunk kc for the nx node rows in X straight from the accumulators to
        // P / Q (they are saved for the backward anyway), no LDS image -- a tile is then limited by its 128 EDGES instead of 64 nodes, and the
        // chunks run back to back as one MFMA stream (one exposed fragment latency per tile instead of one per chunk)
        auto gemm1_global = [&](int kc, int x0g, int nx) {
            f32x16 acc0, acc1;
            acc_zero(acc0); acc_zero(acc1);
            const float4* bp = A.Wp1 + ((size_t)(kc * 4 + cw) * g.S1) * 64 + lane;
            const uint4* bx = reinterpret_cast<const uint4*>(A.Wp1) + ((size_t)(kc * 4 + cw) * (H / 16)) * 192 + lane;
            // row blocks are dealt to the two row-waves alternately (block rw, then rw + 2): a tile of <= 64 nodes keeps both busy
            const int rbA = rw, rbB = rw + 2;
            const int na = (rbA * 32 < nx ? 1 : 0) + ((NRB > 1 && rbB * 32 < nx) ? 1 : 0);
            if (na <= 0) return;
            if (X6) {
                const float* a0 = X + (rbA * 32 + c) * LDX + 8 * h;
                if (na > 1) mma_run_x6<2>(acc0, acc1, a0, a0 + 64 * LDX, bx, H / 16);
                else mma_run_x6<1>(acc0, acc1, a0, nullptr, bx, H / 16);
            } else {
                const float* a0 = X + (rbA * 32 + c) * LDX + h * (H / 2);
                if (na > 1) mma_run<2>(acc0, acc1, a0, a0 + 64 * LDX, bp, g.S1);
                else mma_run<1>(acc0, acc1, a0, nullptr, bp, g.S1);
            }
            const int col128 = cw * 32 + c, ch = kc * CH + (col128 & (CH - 1));
            if (ch >= C1) return;
            // one 64-bit row pointer per block, 32-bit row offsets from an opaque copy of C1: otherwise the compiler hoists the 32 row
            // addresses (64 registers) out of the chunk loop and the kernel spills
            int c1o = C1;
            asm volatile("" : "+s"(c1o));
            float* const pA = (col128 >= CH ? A.Q : A.P) + ch + (size_t)(x0g + rbA * 32 + 4 * h) * C1;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int rr = (r & 3) + 8 * (r >> 2);
                if (rbA * 32 + 4 * h + rr < nx) pA[rr * c1o] = acc0[r];
            }
            if (na > 1) {
                float* const pB = pA + (size_t)64 * C1;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int rr = (r & 3) + 8 * (r >> 2);
                    if (rbB * 32 + 4 * h + rr < nx) pB[rr * c1o] = acc1[r];
                }
            }
        };
        // the chunk just computed -> P (| Q) in global memory, 16-byte rows from LDS (rows [0, nx) of the tile's nodes at global row x0)
        auto save_pq = [&](int kc, int x0, int nx) {
            if (EDGE) {
                for (int i = tid; i < nx * (2 * CH4); i += FT) {
                    const int r = i / (2 * CH4), q = i - r * (2 * CH4);
                    const int col = kc * CH + 4 * (q % CH4);
                    if (col < C1) st4((q < CH4 ? A.P : A.Q) + (size_t)(x0 + r) * C1 + col, ld4(U + r * LDU + 4 * q));
                }
            } else {
                for (int i = tid; i < nx * CH4; i += FT) {
                    const int r = i / CH4, q = i - r * CH4;
                    const int col = kc * CH + 4 * q;
                    if (col < C1) st4(A.P + (size_t)(x0 + r) * C1 + col, ld4(T + r * LDT + 4 * q));
                }
            }
        };
        if (big) {
            const int RXs = EDGE ? g.RX : RM;
            for (int x0 = 0; x0 < tl.nnodes; x0 += RXs) {
                const int nx = min(RXs, tl.nnodes - x0);
                __syncthreads();
                load_x(tl.node0 + x0, nx);
                __syncthreads();
                if (PQG) {
                    for (int kc = 0; kc < g.NCH; ++kc) gemm1_global(kc, tl.node0 + x0, nx);
                } else {
                for (int kc = 0; kc < g.NCH; ++kc) {
                    gemm1(kc, nx);
                    __syncthreads();
                    save_pq(kc, tl.node0 + x0, nx);
                    __syncthreads();
                }
                }
            }
            __syncthreads();          // P / Q of the whole graph are in global memory and visible to this workgroup
        }

        // column statistics of a big graph straight from global memory: src(row, col) -> value; results to lds stat arrays + global
        auto big_stats = [&](int C, auto&& value, float* bmean, float* brstd, float* gmean, float* grstd) {
            // 128 channels x 4 row slots per sweep; partial sums meet in LDS scratch (T is free here)
            float* const scratch = T;
            const int cc = tid & 127, slot = tid >> 7;
            const float inv_n = 1.f / (float)max(tl.nrows, 1);
            for (int c0 = 0; c0 < C; c0 += 128) {
                const int col = c0 + cc;
                float s = 0.f;
                if (col < C) for (int r = slot; r < tl.nrows; r += 4) s += value(r, col);
                __syncthreads();
                scratch[slot * 128 + cc] = s;
                __syncthreads();
                const float mu = ((scratch[cc] + scratch[128 + cc]) + (scratch[256 + cc] + scratch[384 + cc])) * inv_n;
                float v = 0.f;
                if (col < C) for (int r = slot; r < tl.nrows; r += 4) { const float d = value(r, col) - mu; v = fmaf(d, d, v); }
                __syncthreads();
                scratch[slot * 128 + cc] = v;
                __syncthreads();
                if (slot == 0 && col < C) {
                    const float var = ((scratch[cc] + scratch[128 + cc]) + (scratch[256 + cc] + scratch[384 + cc])) * inv_n;
                    const float rs = 1.f / sqrtf(var + F_EPS);
                    bmean[col] = mu; brstd[col] = rs;
                    gmean[(size_t)tl.g0 * C + col] = mu; grstd[(size_t)tl.g0 * C + col] = rs;
                }
            }
            __syncthreads();
        };
        float* const bMean = sMean;                 // big graph: [C] mean | [C] rstd of the layer in flight
        float* const bRstd = sMean + max(C1, C2);
        if (big) {
            if (EDGE) {
                big_stats(C1, [&](int r, int col) {
                    const int m = A.order ? A.order[tl.row0 + r] : tl.row0 + r;
                    return (A.P[(size_t)A.src[m] * C1 + col] + A.Q[(size_t)A.dst[m] * C1 + col]) + A.b1[col];
                }, bMean, bRstd, mean1, rstd1);
            } else {
                big_stats(C1, [&](int r, int col) { return A.P[(size_t)(tl.row0 + r) * C1 + col] + A.b1[col]; }, bMean, bRstd, mean1, rstd1);
            }
        }

        // ======================================= slabs of MLP rows (one for a small tile) =======================================
        for (int sl = 0; sl < nslab_m; ++sl) {
            const int r0 = sl * RM;                                 // first tile row of the slab
            const int nr = big ? min(RM, tl.nrows - r0) : tl.nrows;
            const int nrp = (nr + 31) & ~31;
            lds_barrier();
            if (!big) load_x(EDGE ? tl.node0 : tl.row0, EDGE ? tl.nnodes : tl.nrows);
            for (int r = tid; r < RM; r += FT) {
                if (r < nr) {
                    const int gr = tl.row0 + r0 + r;
                    const int m = EDGE ? (A.order ? A.order[gr] : gr) : gr;
                    sRowId[r] = m;
                    if (EDGE) {
                        const int s = A.src[m], d = A.dst[m];
                        const int sl_ = s - tl.node0, dl_ = d - tl.node0;
                        sSrcL[r] = (!big && (unsigned)sl_ < (unsigned)tl.nnodes) ? sl_ : -(s + 1);
                        sDstL[r] = (!big && (unsigned)dl_ < (unsigned)tl.nnodes) ? dl_ : -(d + 1);
                        if (!big && ((unsigned)sl_ >= (unsigned)tl.nnodes || (unsigned)dl_ >= (unsigned)tl.nnodes)) sCtl[1] = 1;      // (benign race: every writer stores 1)
                    }
                    int gl = 0;
                    if (!big) {
#pragma unroll
                        for (int k = 1; k < F_GT; ++k) gl += (k < tl.ng && sGptr[k] <= r) ? 1 : 0;       // graphs (empty ones included) that end at or before r
                    }
                    sRowG[r] = gl;
                }
            }
            f32x16 acc2[NCB2W][NRB];
#pragma unroll
            for (int j = 0; j < NCB2W; ++j)
#pragma unroll
                for (int i = 0; i < NRB; ++i) acc_zero(acc2[j][i]);
            lds_barrier();
            STAMP(0);
            if (PQG && !big) {          // P | Q of the tile's nodes, every chunk, straight to global memory; visible to this workgroup after the barrier
                for (int kc = 0; kc < g.NCH; ++kc) gemm1_global(kc, tl.node0, tl.nnodes);
                __syncthreads();
                STAMP(1);
            }

            for (int kc = 0; kc < g.NCH; ++kc) {
                // ---- layer 1, chunk kc -> T[row][CH] (pre-activation incl. bias) ----------------------------------------
                if (!big && !PQG) {
                    gemm1(kc, EDGE ? tl.nnodes : tl.nrows);
                    lds_barrier();
                    STAMP(1);
                    save_pq(kc, EDGE ? tl.node0 : tl.row0, EDGE ? tl.nnodes : tl.nrows);
                    STAMP(10);
                }
                if (PQG) {
                    // P[src] + Q[dst] from global memory (L2: written by this workgroup a moment ago), four rows x two loads in flight per
                    // thread, unconditional with clamped indices; an endpoint outside the tile (cross-graph edge) is recomputed
                    constexpr int GB = 2;
                    for (int i0 = tid; i0 < nrp * CH4; i0 += GB * FT) {
                        float4 pv[GB], qv[GB];
#pragma unroll
                        for (int u = 0; u < GB; ++u) {
                            const int i = min(i0 + u * FT, nrp * CH4 - 1), r = min(i / CH4, nr - 1), q = i - (i / CH4) * CH4;
                            const int col = min(kc * CH + 4 * q, C1 - 4);
                            const int sl_ = sSrcL[r], dl_ = sDstL[r];
                            const int sg = sl_ >= 0 ? tl.node0 + sl_ : -sl_ - 1, dg = dl_ >= 0 ? tl.node0 + dl_ : -dl_ - 1;
                            pv[u] = ld4(A.P + (size_t)sg * C1 + col);
                            qv[u] = ld4(A.Q + (size_t)dg * C1 + col);
                        }
#pragma unroll
                        for (int u = 0; u < GB; ++u) {
                            const int i = i0 + u * FT;
                            if (i < nrp * CH4) {
                                const int r = i / CH4, q = i - r * CH4, col = kc * CH + 4 * q;
                                st4(T + r * LDT + 4 * q, (r < nr && col < C1) ? f4add(pv[u], qv[u]) : f4zero());
                            }
                        }
                    }
                    if (!big && sCtl[1] != 0) {          // (rare) endpoints outside the tile: their P / Q rows belong to another workgroup -- recomputed
                        for (int i = tid; i < nr * CH4; i += FT) {                  // (same item <-> thread map as above: no barrier in between)
                            const int r = i / CH4, q = i - r * CH4, col = kc * CH + 4 * q;
                            const int sl_ = sSrcL[r], dl_ = sDstL[r];
                            if ((sl_ < 0 || dl_ < 0) && col < C1) {
                                const float4 p4 = sl_ >= 0 ? ld4(A.P + (size_t)(tl.node0 + sl_) * C1 + col) : slow_pq4(A.emb + (size_t)(-sl_ - 1) * H, A.W1, K1, 0, col, C1, H);
                                const float4 q4 = dl_ >= 0 ? ld4(A.Q + (size_t)(tl.node0 + dl_) * C1 + col) : slow_pq4(A.emb + (size_t)(-dl_ - 1) * H, A.W1, K1, H, col, C1, H);
                                st4(T + r * LDT + 4 * q, f4add(p4, q4));
                            }
                        }
                    }
                    lds_barrier();
                    STAMP(2);
                } else if (EDGE || big) {
                    for (int i = tid; i < nrp * CH4; i += FT) {
                        const int r = i / CH4, q = i - r * CH4;
                        float4 v = f4zero();
                        const int col = kc * CH + 4 * q;
                        if (r < nr && col < C1) {
                            if (EDGE) {
                                const int s = sSrcL[r], d = sDstL[r];
                                float4 p4, q4;
                                if (big) {
                                    p4 = ld4(A.P + (size_t)(-s - 1) * C1 + col);
                                    q4 = ld4(A.Q + (size_t)(-d - 1) * C1 + col);
                                } else {
                                    p4 = s >= 0 ? ld4(U + s * LDU + 4 * q) : slow_pq4(A.emb + (size_t)(-s - 1) * H, A.W1, K1, 0, col, C1, H);
                                    q4 = d >= 0 ? ld4(U + d * LDU + CH + 4 * q) : slow_pq4(A.emb + (size_t)(-d - 1) * H, A.W1, K1, H, col, C1, H);
                                }
                                v = f4add(p4, q4);
                            } else {
                                v = ld4(A.P + (size_t)sRowId[r] * C1 + col);
                            }
                        }
                        st4(T + r * LDT + 4 * q, v);
                    }
                    lds_barrier();
                    STAMP(2);
                }
                // ---- per-graph statistics of the chunk (small tiles; a big graph has them already) -------------------------
                if (!big) {
                    constexpr int nslot = FT / CH;
                    const int cc = tid % CH, slot = tid / CH;
                    const int col = kc * CH + cc;
                    const float bias = sB1[col];
                    for (int gl = slot; gl < tl.ng; gl += nslot) {
                        float mu, rs;
                        column_stats<32>(T + cc, LDT, bias, sGptr[gl], sGptr[gl + 1], &mu, &rs);
                        sMean[gl * SW + cc] = mu;
                        sMean[(F_GT + gl) * SW + cc] = rs;
                        if (col < C1) { mean1[(size_t)(tl.g0 + gl) * C1 + col] = mu; rstd1[(size_t)(tl.g0 + gl) * C1 + col] = rs; }
                    }
                    STAMP(11);
                    lds_barrier();
                    STAMP(3);
                }
                // ---- normalise + ReLU + dropout in place (a1 chunk) -----------------------------------------------------------
                constexpr int AB = (NRB == 2 && NCB2W == 2) ? 2 : 4;      // float4 items per thread in flight (registers: 64 accumulators stay live)
                for (int i0 = tid; i0 < nr * CH4; i0 += AB * FT) {
                    float4 hv[AB], mu[AB], rs[AB], bb[AB];
                    int mm[AB], rr[AB], qq[AB];
#pragma unroll
                    for (int u = 0; u < AB; ++u) {
                        const int i = min(i0 + u * FT, nr * CH4 - 1);          // clamped: loads stay unconditional, stores are guarded
                        rr[u] = i / CH4; qq[u] = i - rr[u] * CH4;
                        hv[u] = ld4(T + rr[u] * LDT + 4 * qq[u]);
                        bb[u] = ld4(sB1 + kc * CH + 4 * qq[u]);
                        mm[u] = sRowId[rr[u]];
                        if (big) { mu[u] = ld4(bMean + min(kc * CH + 4 * qq[u], C1 - 4)); rs[u] = ld4(bRstd + min(kc * CH + 4 * qq[u], C1 - 4)); }
                        else { const int gl = sRowG[rr[u]]; mu[u] = ld4(sMean + gl * SW + 4 * qq[u]); rs[u] = ld4(sMean + (F_GT + gl) * SW + 4 * qq[u]); }
                    }
#pragma unroll
                    for (int u = 0; u < AB; ++u) {
                        const int col = kc * CH + 4 * qq[u];
                        if (i0 + u * FT < nr * CH4 && col < C1) {
                            const float4 x = f4add(hv[u], bb[u]);
                            const float4 k = keep4f(A.mask1, A.seed, 1, mm[u], col, C1, A.p, train);
                            const float4 y = make_float4(fmaxf((x.x - mu[u].x) * rs[u].x, 0.f) * k.x * sc, fmaxf((x.y - mu[u].y) * rs[u].y, 0.f) * k.y * sc,
                                                         fmaxf((x.z - mu[u].z) * rs[u].z, 0.f) * k.z * sc, fmaxf((x.w - mu[u].w) * rs[u].w, 0.f) * k.w * sc);
                            st4(T + rr[u] * LDT + 4 * qq[u], y);
                            if (A.a1) st4(A.a1 + (size_t)mm[u] * C1 + col, y);
                        }
                    }
                }
                lds_barrier();
                STAMP(4);
                // ---- layer 2: acc2 += a1 chunk x W2[:, chunk] ---------------------------------------------------------------------
                {
                    const int rb0 = rw * NRB;
                    const int na = min(NRB, (nr - rb0 * 32 + 31) / 32);
                    if (na > 0) {
                        const float* a0 = T + (rb0 * 32 + c) * LDT + (X6 ? 8 * h : h * (CH / 2));
#pragma unroll
                        for (int j = 0; j < NCB2W; ++j) {
                            const int cb2 = cw + 4 * j;
                            if (cb2 < g.NCB2) {
                                if (X6) {
                                    const uint4* bx = reinterpret_cast<const uint4*>(A.Wp2) + ((size_t)(kc * g.NCB2 + cb2) * (CH / 16)) * 192 + lane;
                                    if (NRB > 1 && na > 1) mma_run_x6<2>(acc2[j][0], acc2[j][NRB - 1], a0, a0 + 32 * LDT, bx, CH / 16);
                                    else mma_run_x6<1>(acc2[j][0], acc2[j][NRB - 1], a0, nullptr, bx, CH / 16);
                                } else {
                                const float4* bp = A.Wp2 + ((size_t)(kc * g.NCB2 + cb2) * S2) * 64 + lane;
                                if (NRB > 1 && na > 1) mma_run<2>(acc2[j][0], acc2[j][NRB - 1], a0, a0 + 32 * LDT, bp, S2);
                                else mma_run<1>(acc2[j][0], acc2[j][NRB - 1], a0, nullptr, bp, S2);
                                }
                            }
                        }
                    }
                }
                lds_barrier();
                STAMP(5);
            }
            // ---- layer-2 tile -> LDS (overlays X / U / T), h2 -> global -------------------------------------------------------------
            {
                const int rb0 = rw * NRB;
#pragma unroll
                for (int j = 0; j < NCB2W; ++j) {
                    const int cb2 = cw + 4 * j;
                    if (cb2 < g.NCB2) {
#pragma unroll
                        for (int i = 0; i < NRB; ++i) {
                            if ((rb0 + i) * 32 < nrp) {
#pragma unroll
                                for (int r = 0; r < 16; ++r) {
                                    const int row = (rb0 + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                                    H2[row * LDH + cb2 * 32 + c] = acc2[j][i][r];
                                }
                            }
                        }
                    }
                }
            }
            lds_barrier();
            for (int i = tid; i < nr * C24; i += FT) {
                const int r = i / C24, q = i - r * C24;
                st4(A.h2 + (size_t)sRowId[r] * C2 + 4 * q, ld4(H2 + r * LDH + 4 * q));
            }
            STAMP(6);
            if (big) continue;             // the head of a big graph needs the statistics of ALL its slabs: below, from global h2
            // ---- layer-2 statistics ------------------------------------------------------------------------------------------------
            {
                const int cc = tid % g.C2p, slot = tid / g.C2p, nslot = FT / g.C2p;
                if (slot < nslot) {
                    const float bias = sB2[cc];
                    for (int gl = slot; gl < tl.ng; gl += nslot) {
                        float mu, rs;
                        column_stats<32>(H2 + cc, LDH, bias, sGptr[gl], sGptr[gl + 1], &mu, &rs);
                        sMean[gl * SW + cc] = mu;
                        sMean[(F_GT + gl) * SW + cc] = rs;
                        if (cc < C2) { mean2[(size_t)(tl.g0 + gl) * C2 + cc] = mu; rstd2[(size_t)(tl.g0 + gl) * C2 + cc] = rs; }
                    }
                }
            }
            lds_barrier();
            STAMP(7);
            // ---- head: C2 -> 1 dot, concrete sample ------------------------------------------------------------------------------
            {
                constexpr int TPR = FT / (64 * NRB);           // threads per row: 4 (128-row tiles) or 8
                const int r = tid / TPR, part = tid % TPR;
                float acc = 0.f;
                int m = 0;
                if (r < nr) {
                    m = sRowId[r];
                    const int gl = sRowG[r];
                    for (int q = part; q < C24; q += TPR) {
                        const float4 hv = f4add(ld4(H2 + r * LDH + 4 * q), ld4(sB2 + 4 * q));
                        const float4 mu = ld4(sMean + gl * SW + 4 * q), rs = ld4(sMean + (F_GT + gl) * SW + 4 * q);
                        const float4 k = keep4f(A.mask2, A.seed, 2, m, 4 * q, C2, A.p, train);
                        const float4 w = ld4(sW3 + 4 * q);
                        acc = fmaf(fmaxf((hv.x - mu.x) * rs.x, 0.f) * k.x * sc, w.x, acc);
                        acc = fmaf(fmaxf((hv.y - mu.y) * rs.y, 0.f) * k.y * sc, w.y, acc);
                        acc = fmaf(fmaxf((hv.z - mu.z) * rs.z, 0.f) * k.z * sc, w.z, acc);
                        acc = fmaf(fmaxf((hv.w - mu.w) * rs.w, 0.f) * k.w * sc, w.w, acc);
                    }
                }
                acc = group_sum<TPR>(acc);
                if (part == 0 && r < nr) {
                    const float z = acc + A.b3[0];
                    A.logits[m] = z;
                    if (A.att) {
                        float tt = z;
                        if (train && (A.u || A.noise_philox)) {
                            const float uu = A.u ? A.u[m] : philox_noise_u(A.seed.get(), m);
                            tt = z + (logf(uu) - logf(1.0f - uu));
                        }
                        A.att[m] = 1.f / (1.f + expf(-tt));
                    }
                }
            }
            STAMP(8);
        }
        if (big) {
            __syncthreads();          // h2 of the whole graph is in global memory
            big_stats(C2, [&](int r, int col) {
                const int gr = tl.row0 + r;
                const int m = EDGE ? (A.order ? A.order[gr] : gr) : gr;
                return A.h2[(size_t)m * C2 + col] + A.b2[col];
            }, bMean, bRstd, mean2, rstd2);
            constexpr int TPR = 4;
            for (int rbase = 0; rbase < tl.nrows; rbase += FT / TPR) {
                const int r = rbase + tid / TPR, part = tid % TPR;
                float acc = 0.f;
                int m = 0;
                if (r < tl.nrows) {
                    const int gr = tl.row0 + r;
                    m = EDGE ? (A.order ? A.order[gr] : gr) : gr;
                    for (int q = part; q < C24; q += TPR) {
                        const float4 hv = f4add(ld4(A.h2 + (size_t)m * C2 + 4 * q), ld4(A.b2 + 4 * q));
                        const float4 mu = ld4(bMean + 4 * q), rs = ld4(bRstd + 4 * q);
                        const float4 k = keep4f(A.mask2, A.seed, 2, m, 4 * q, C2, A.p, train);
                        const float4 w = ld4(A.w3 + 4 * q);
                        acc = fmaf(fmaxf((hv.x - mu.x) * rs.x, 0.f) * k.x * sc, w.x, acc);
                        acc = fmaf(fmaxf((hv.y - mu.y) * rs.y, 0.f) * k.y * sc, w.y, acc);
                        acc = fmaf(fmaxf((hv.z - mu.z) * rs.z, 0.f) * k.z * sc, w.z, acc);
                        acc = fmaf(fmaxf((hv.w - mu.w) * rs.w, 0.f) * k.w * sc, w.w, acc);
                    }
                }
                acc = group_sum<TPR>(acc);
                if (part == 0 && r < tl.nrows) {
                    const float z = acc + A.b3[0];
                    A.logits[m] = z;
                    if (A.att) {
                        float tt = z;
                        if (train && (A.u || A.noise_philox)) {
                            const float uu = A.u ? A.u[m] : philox_noise_u(A.seed.get(), m);
                            tt = z + (logf(uu) - logf(1.0f - uu));
                        }
                        A.att[m] = 1.f / (1.f + expf(-tt));
                    }
                }
            }
        }
    }
#ifdef GSAT_FUSED_STAMPS
    STAMP(9);
    if (tid == 0) {
        unsigned long long* o = reinterpret_cast<unsigned long long*>(A.counters + 16);
        for (int i = 0; i < 14; ++i) atomicAdd(o + i, (unsigned long long)stamp[i]);
        atomicAdd(o + 14, 1ull);
    }
#endif
}

template <bool EDGE, int NRB, int NCB2W, bool X6 = false, bool PQG = false>
static int launch_fused(hipStream_t stream, const FusedArgs& fa, int grid) {
    static size_t allowed = 64 * 1024;
    const size_t lds = (size_t)fa.g.lds_bytes;
    if (lds > allowed) {
        GSAT_CHECK_HIP(hipFuncSetAttribute((const void*)k_attn_fused_fwd<EDGE, NRB, NCB2W, X6, PQG>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        allowed = lds;
    }
    k_attn_fused_fwd<EDGE, NRB, NCB2W, X6, PQG><<<grid, FT, lds, stream>>>(fa);
    GSAT_LAUNCH_CHECK();
    return GSAT_OK;
}

size_t attn_plan_bytes(int64_t G) { return align_up((size_t)(G + 1 + (G + 63) / 64 * 64) * sizeof(FTile), 256); }
bool attn_plan_ok(int64_t G) { return G > 0 && G <= (int64_t)PLAN_MAX_SB * 64; }
int attn_plan_launch(hipStream_t stream, const int32_t* seg_ptr, const int32_t* node_ptr, int64_t G, int RM, int RX, FTile* tiles, int* counters) {
    k_attn_plan<<<1, PLAN_T, 0, stream>>>(seg_ptr, node_ptr, (int)G, RM, RX, tiles, counters);
    GSAT_LAUNCH_CHECK();
    return GSAT_OK;
}

bool attn_fused_eligible(const gsat_attn_args* a, FusedGeom* g) {
    // Policy (measured on MI355X, profiles/r03_summary.md).  args->fused / GSAT_ATTN_FUSED: 1 = take the one-launch forward whenever the
    // shapes allow it, -1 / "0" = never, 0 / unset = automatic: node mode with the split-bf16 x 6 products on a batch of at least 32768
    // rows (C3: 0.935 vs 0.949 ms per step).  Small batches and edge-mode batches stay on the staged pipeline: with one workgroup per CU
    // and barrier-separated phases a launch of few tiles loses (C1 / C2 / C4).
    const char* env = getenv("GSAT_ATTN_FUSED");                   // read per call: tests flip it inside one process
    const int want = a->fused != 0 ? a->fused : (env ? (atoi(env) != 0 ? 1 : -1) : 0);
    if (want < 0) return false;
    if (a->M <= 0 || a->G <= 0 || a->G > (int64_t)PLAN_MAX_SB * 64) return false;
    if (!a->edge_mode && a->seg_order) return false;
    if (a->edge_mode && !a->node_ptr) return false;
    {   // edge mode: P | Q of a tile's nodes through global memory (tiles limited by 128 EDGES, not 64 nodes) unless GSAT_ATTN_FUSED_PQG=0
        const char* ep = getenv("GSAT_ATTN_FUSED_PQG");
        const bool pqg = a->edge_mode && !(ep && atoi(ep) == 0);
        if (!(pqg && fused_geometry(a->H, a->C1, a->C2, true, g, true) && g->NRB == 2) && !fused_geometry(a->H, a->C1, a->C2, a->edge_mode != 0, g, false)) return false;
    }
    {   // split-bf16 x 6 products (GSAT_ATTN_FUSED_X6=0: exact fp32 MFMA): the common tile shape only, K a multiple of 16
        const char* ex = getenv("GSAT_ATTN_FUSED_X6");
        g->x6 = (!(ex && atoi(ex) == 0) && a->H % 16 == 0 && g->NRB == 2 && g->NCB2 <= 4) ? 1 : 0;
    }
    // batches of huge graphs (C5: ~10^5 rows per graph) belong to the streaming pipeline; an occasional large graph is walked in slabs here
    if (a->M > a->G * (int64_t)(2 * g->RM)) return false;
    if (want == 0 && (a->edge_mode || !g->x6 || a->M < 32768)) return false;
    return true;
}

int attn_fused_fwd(hipStream_t stream, const gsat_attn_args* a, const FusedGeom& g) {
    const size_t need = fused_ws_bytes(g, a->G);
    GSAT_REQUIRE(a->fwd_workspace && a->fwd_workspace_bytes >= need, GSAT_ERR_WORKSPACE, "gsat_attn_fwd: workspace %zu < %zu (fused path)",
                 a->fwd_workspace_bytes, need);
    char* w = static_cast<char*>(a->fwd_workspace);
    int* counters = reinterpret_cast<int*>(w); w += 256;
    FTile* tiles = reinterpret_cast<FTile*>(w); w += align_up((size_t)(a->G + 1 + (a->G + 63) / 64 * 64) * sizeof(FTile), 256);
    float4* Wp1 = reinterpret_cast<float4*>(w); w += align_up((size_t)g.NCH * 4 * g.S1 * 64 * 24, 256);
    float4* Wp2 = reinterpret_cast<float4*>(w);
    const int64_t nflt4 = (int64_t)g.NCH * 4 * g.S1 * 64 + (int64_t)g.NCH * g.NCB2 * g.S2 * 64;
    const int pack_blocks = (int)std::min<int64_t>(ceil_div(nflt4, PLAN_T), 512);
    const int32_t* node_ptr = a->edge_mode ? a->node_ptr : a->seg_ptr;
    k_attn_prep<<<1 + pack_blocks, PLAN_T, 0, stream>>>(a->W1, a->W2, g, a->edge_mode, a->seg_ptr, node_ptr, (int)a->G, tiles, counters, Wp1, Wp2);
    GSAT_LAUNCH_CHECK();
    FusedArgs fa{};
    fa.emb = a->emb; fa.W1 = a->W1; fa.b1 = a->b1; fa.b2 = a->b2; fa.w3 = a->W3; fa.b3 = a->b3;
    fa.Wp1 = Wp1; fa.Wp2 = Wp2;
    fa.src = a->src; fa.dst = a->dst; fa.seg_ptr = a->seg_ptr; fa.order = a->seg_order;
    fa.mask1 = a->mask1; fa.mask2 = a->mask2; fa.u = a->u;
    fa.P = a->P; fa.Q = a->Q; fa.a1 = a->a1; fa.h2 = a->h2; fa.stats = a->stats; fa.logits = a->logits; fa.att = a->att;
    fa.tiles = tiles; fa.counters = counters;
    fa.M = a->M; fa.N = a->N; fa.G = a->G;
    fa.seed = SeedRef{a->seed, a->seed_dev};
    fa.p = a->p_drop; fa.training = a->training; fa.noise_philox = a->noise_philox;
    fa.g = g;
    static int cus = 0;
    if (!cus) {
        int dev = 0;
        hipDeviceProp_t prop;
        GSAT_CHECK_HIP(hipGetDevice(&dev));
        GSAT_CHECK_HIP(hipGetDeviceProperties(&prop, dev));
        cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    }
    const int grid = (int)std::min<int64_t>(cus, a->G);         // one workgroup per CU pulls tiles from the queue (tiles <= graphs)
    const bool e = a->edge_mode != 0;
    const int ncbw = g.NCB2 > 4 ? 2 : 1;
#define GO(E, R, W) return launch_fused<E, R, W>(stream, fa, grid)
    if (e && g.pqg) {            // P | Q through global memory: NRB = 2 by construction
        if (g.x6) return launch_fused<true, 2, 1, true, true>(stream, fa, grid);
        if (ncbw == 2) return launch_fused<true, 2, 2, false, true>(stream, fa, grid);
        return launch_fused<true, 2, 1, false, true>(stream, fa, grid);
    }
    if (g.x6) { if (e) return launch_fused<true, 2, 1, true>(stream, fa, grid); return launch_fused<false, 2, 1, true>(stream, fa, grid); }
    if (e) { if (g.NRB == 2) { if (ncbw == 2) GO(true, 2, 2); GO(true, 2, 1); } if (ncbw == 2) GO(true, 1, 2); GO(true, 1, 1); }
    if (g.NRB == 2) { if (ncbw == 2) GO(false, 2, 2); GO(false, 2, 1); }
    if (ncbw == 2) GO(false, 1, 2);
    GO(false, 1, 1);
#undef GO
}

}  // namespace gsat
