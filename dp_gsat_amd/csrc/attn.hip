// Attention extractor: Linear -> per-graph InstanceNorm -> ReLU -> Dropout (x2) -> Linear(.,1) -> concrete
// sampler, forward and backward (example/gsat.py:94-103,131-139; src/utils/get_model.py:47-68).
//
// Staging (round-1 form; the ABI hides it):
//   * edge mode evaluates layer 1 on NODES: P = emb W1a^T, Q = emb W1b^T, h1_e = P[src_e]+Q[dst_e]+b1.
//     The [E,2H] concat, the E-row first GEMM and h1 itself are never materialised; the InstanceNorm
//     statistics kernel and the a1 producer recompute h1 from two gathered rows.
//   * dense contractions (P/Q, h2 = a1 W2^T and their gradients) go through the hand-written MFMA GEMMs of gemm.hip:
//     forward products exact fp32 (weight-stationary k_gemm_ws / tile kernel k_gemm_f32), backward products split-bf16;
//     the rest: segmented statistics, normalise + ReLU + Philox dropout, the head (C2 -> 1 dot + sampler, noise drawn
//     in the kernel when no tensor is given), InstanceNorm backward, deterministic column sums.
//   * all reductions run in a fixed order (no float atomics): results are bitwise reproducible.
#include "common.h"
#include "attn_fused.h"
#include <cstdlib>

namespace gsat {

constexpr float IN_EPS = 1e-5f;
constexpr int SB = 256;          // threads per block in the segmented kernels: 16 row slots x 16 lanes
#ifndef GSAT_SB_LANES
#define GSAT_SB_LANES 16
#endif
constexpr int SB_LANES = GSAT_SB_LANES;     // lanes per row slot, one float4 each -> 64 channels per block
constexpr int SB_SLOTS = SB / SB_LANES;
constexpr int SB_RC = 2;         // rows per slot kept in registers across the passes of the segmented kernels

// row-major C[M,N] = alpha(=1) * op(A) op(B) + beta(0|1) * C through the hand-written MFMA GEMM (gemm.hip).
// ta: A is given as [K,M]; tb: B is given as [N,K] (nn.Linear weight layout).
struct GemmWs { float* ptr; size_t floats; };
// `split_ok`: the product may run as split-bf16 (relative error ~1e-5 of |A||B|).  Only the BACKWARD products qualify: their
// results pass through linear maps only (rstd * (dy - mean(dy) - y mean(dy y)), further GEMMs), so the relative error stays
// 1e-5 of each gradient's own scale.  The forward products feed an InstanceNorm directly: 1/sigma of a nearly constant channel
// turns the same perturbation of h into ~3e2 x 1e-5 of the normalised value, so they stay exact fp32.
static int gemm_rm(hipStream_t stream, bool ta, bool tb, int64_t M, int64_t N, int64_t K, const float* A, int64_t lda, const float* B,
                   int64_t ldb, float beta, float* C, int64_t ldc, GemmWs ws = {nullptr, 0}, bool split_ok = false, SlabJob* defer = nullptr) {
    const char* env = getenv("GSAT_ATTN_BWD_SPLIT");                    // read per call: bench.py times both settings in one process
    const bool bwd_split = !(env && atoi(env) == 0);
    return gemm_f32(stream, ta, tb, M, N, K, A, lda, B, ldb, C, ldc, nullptr, beta != 0.f, ws.ptr, ws.floats, split_ok && bwd_split, defer);
}

// ------------------------------------------------------------------------------------------------
// pre-activation sources: dense rows (node mode / layer 2) or P[src]+Q[dst] (edge mode layer 1)
// ------------------------------------------------------------------------------------------------
template <bool EDGE>
struct PreAct {
    const float* P; const float* Q; const float* bias; const int32_t* src; const int32_t* dst; int C;
    __device__ __forceinline__ float4 load(int row, int c) const {
        float4 b = bias ? ld4(bias + c) : f4zero();
        float4 v;
        if (EDGE) {
            float4 p = ld4(P + (size_t)src[row] * C + c);
            float4 q = ld4(Q + (size_t)dst[row] * C + c);
            v = make_float4(p.x + q.x, p.y + q.y, p.z + q.z, p.w + q.w);
        } else {
            v = ld4(P + (size_t)row * C + c);
        }
        return make_float4(v.x + b.x, v.y + b.y, v.z + b.z, v.w + b.w);
    }
};

__device__ __forceinline__ float4 keep4(const float* mask, SeedRef sref, int layer, int row, int c, int C, float p, bool training) {
    if (!training || p <= 0.f) return make_float4(1.f, 1.f, 1.f, 1.f);
    if (mask) return ld4(mask + (size_t)row * C + c);
    return philox_keep4(sref.get(), layer, row, c, p);
}

// Sum a float4 per (slot, lane) over the 16 row slots in fixed order; result valid in slot 0.
__device__ __forceinline__ float4 slot_reduce(float4 v, float4 (*sm)[SB_LANES], int slot, int lane) {
    __syncthreads();
    sm[slot][lane] = v;
    __syncthreads();
    float4 r = f4zero();
    if (slot == 0) {
#pragma unroll
        for (int s = 0; s < SB_SLOTS; ++s) {
            float4 t = sm[s][lane];
            r.x += t.x; r.y += t.y; r.z += t.z; r.w += t.w;
        }
    }
    return r;
}

// rows [beg,end) of the z-th of Z equal slices of segment g (Z > 1: few huge graphs, blockIdx.z = slice)
__device__ __forceinline__ void seg_slice(const int32_t* __restrict__ seg_ptr, int g, int* beg, int* end, float* inv_n) {
    const int b = seg_ptr[g], e = seg_ptr[g + 1];
    *inv_n = 1.f / (float)max(e - b, 1);
    const int Z = gridDim.z, z = blockIdx.z;
    if (Z > 1) {
        const int per = (e - b + Z - 1) / Z;
        *beg = min(e, b + z * per);
        *end = min(e, *beg + per);
    } else {
        *beg = b; *end = e;
    }
}

// Z > 1 helpers for k_seg_stats: partial sums of h, then of (h - mean)^2, each followed by k_zcombine
template <bool EDGE>
__global__ __launch_bounds__(SB) void k_seg_partial(PreAct<EDGE> pre, const int32_t* __restrict__ seg_ptr, const int32_t* __restrict__ order,
                                                    const float* __restrict__ mean /* null: plain sum */, float* __restrict__ part) {
    __shared__ float4 sm[SB_SLOTS][SB_LANES];
    const int g = blockIdx.x;
    const int lane = threadIdx.x % SB_LANES, slot = threadIdx.x / SB_LANES;
    const int c = (blockIdx.y * SB_LANES + lane) * 4;
    const int C = pre.C;
    const bool on = c < C;
    int beg, end; float inv_n;
    seg_slice(seg_ptr, g, &beg, &end, &inv_n);
    float4 acc = f4zero();
    if (on) {
        const float4 mu = mean ? ld4(mean + (size_t)g * C + c) : f4zero();
        for (int r = beg + slot; r < end; r += SB_SLOTS) {
            float4 h = pre.load(order ? order[r] : r, c);
            if (mean) {
                float dx = h.x - mu.x, dy = h.y - mu.y, dz = h.z - mu.z, dw = h.w - mu.w;
                acc.x = fmaf(dx, dx, acc.x); acc.y = fmaf(dy, dy, acc.y); acc.z = fmaf(dz, dz, acc.z); acc.w = fmaf(dw, dw, acc.w);
            } else {
                acc.x += h.x; acc.y += h.y; acc.z += h.z; acc.w += h.w;
            }
        }
    }
    float4 tot = slot_reduce(acc, sm, slot, lane);
    if (slot == 0 && on) st4(part + ((size_t)g * gridDim.z + blockIdx.z) * C + c, tot);
}

// out[g,c] = f(sum_z part[g,z,c]):  mode 0 raw sum, 1 sum/n, 2 1/sqrt(sum/n + eps)
__global__ void k_zcombine(const float* __restrict__ part, const int32_t* __restrict__ seg_ptr, int G, int Z, int C, int mode,
                           float* __restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (int64_t)G * C) return;
    const int g = (int)(i / C), c = (int)(i % C);
    float acc = 0.f;
    for (int z = 0; z < Z; ++z) acc += part[((size_t)g * Z + z) * C + c];
    const float inv_n = 1.f / (float)max(seg_ptr[g + 1] - seg_ptr[g], 1);
    out[i] = mode == 0 ? acc : mode == 1 ? acc * inv_n : 1.f / sqrtf(acc * inv_n + IN_EPS);
}

// ------------------------------------------------------------------------------------------------
// per-graph, per-channel InstanceNorm statistics: grid (G, ceil(C/64)); two passes over the segment
// (mean, then variance of the centred values -- the reference's order), second pass is L2-hot.
// ------------------------------------------------------------------------------------------------
// APPLY: a third pass over the (now cache-hot) rows of the segment writes the layer's activation
//   a[m,c] = relu((h - mean) * rstd) * keep / (1-p)
// so the layer needs no separate normalise kernel (k_norm_apply) and no second trip of h through HBM.
template <bool EDGE, bool APPLY>
__global__ __launch_bounds__(SB) void k_seg_stats(PreAct<EDGE> pre, const int32_t* __restrict__ seg_ptr,
                                                  const int32_t* __restrict__ order, float* __restrict__ mean_out,
                                                  float* __restrict__ rstd_out, const float* __restrict__ mask = nullptr,
                                                  SeedRef seed = SeedRef{0, nullptr}, int layer = 0, float p = 0.f, int training = 0,
                                                  float* __restrict__ act_out = nullptr) {
    __shared__ float4 sm[SB_SLOTS][SB_LANES];
    __shared__ float4 bc[SB_LANES];
    const int g = blockIdx.x;
    const int lane = threadIdx.x % SB_LANES, slot = threadIdx.x / SB_LANES;
    const int c = (blockIdx.y * SB_LANES + lane) * 4;
    const int C = pre.C;
    const bool on = c < C;
    const int beg = seg_ptr[g], end = seg_ptr[g + 1];
    const float inv_n = 1.f / (float)max(end - beg, 1);
    // the first SB_RC rows of every slot stay in registers across the passes (segments of <= 32 rows -- most molecule graphs -- make ONE
    // trip to memory instead of three dependent ones); longer segments re-read their tail, L2-hot, as before.  Same summation order.
    float4 hc[SB_RC];
    int mc[SB_RC];
#pragma unroll
    for (int i = 0; i < SB_RC; ++i) {
        const int r = beg + slot + i * SB_SLOTS;
        mc[i] = (r < end) ? (order ? order[r] : r) : -1;
    }
#pragma unroll
    for (int i = 0; i < SB_RC; ++i) hc[i] = (on && mc[i] >= 0) ? pre.load(mc[i], c) : f4zero();
    const int rest = beg + slot + SB_RC * SB_SLOTS;
    float4 acc = f4zero();
    if (on) {
#pragma unroll
        for (int i = 0; i < SB_RC; ++i)
            if (mc[i] >= 0) { acc.x += hc[i].x; acc.y += hc[i].y; acc.z += hc[i].z; acc.w += hc[i].w; }
        for (int r = rest; r < end; r += SB_SLOTS) {
            float4 h = pre.load(order ? order[r] : r, c);
            acc.x += h.x; acc.y += h.y; acc.z += h.z; acc.w += h.w;
        }
    }
    float4 tot = slot_reduce(acc, sm, slot, lane);
    if (slot == 0) bc[lane] = make_float4(tot.x * inv_n, tot.y * inv_n, tot.z * inv_n, tot.w * inv_n);
    __syncthreads();
    const float4 mu = bc[lane];
    acc = f4zero();
    if (on) {
#pragma unroll
        for (int i = 0; i < SB_RC; ++i)
            if (mc[i] >= 0) {
                float dx = hc[i].x - mu.x, dy = hc[i].y - mu.y, dz = hc[i].z - mu.z, dw = hc[i].w - mu.w;
                acc.x = fmaf(dx, dx, acc.x); acc.y = fmaf(dy, dy, acc.y); acc.z = fmaf(dz, dz, acc.z); acc.w = fmaf(dw, dw, acc.w);
            }
        for (int r = rest; r < end; r += SB_SLOTS) {
            float4 h = pre.load(order ? order[r] : r, c);
            float dx = h.x - mu.x, dy = h.y - mu.y, dz = h.z - mu.z, dw = h.w - mu.w;
            acc.x = fmaf(dx, dx, acc.x); acc.y = fmaf(dy, dy, acc.y); acc.z = fmaf(dz, dz, acc.z); acc.w = fmaf(dw, dw, acc.w);
        }
    }
    tot = slot_reduce(acc, sm, slot, lane);
    float4 rs = make_float4(1.f / sqrtf(tot.x * inv_n + IN_EPS), 1.f / sqrtf(tot.y * inv_n + IN_EPS),
                            1.f / sqrtf(tot.z * inv_n + IN_EPS), 1.f / sqrtf(tot.w * inv_n + IN_EPS));
    if (slot == 0 && on) {
        st4(mean_out + (size_t)g * C + c, mu);
        st4(rstd_out + (size_t)g * C + c, rs);
    }
    if (APPLY) {
        if (slot == 0) bc[lane] = rs;              // every thread took `mu` out of bc before slot_reduce's barriers
        __syncthreads();
        rs = bc[lane];
        const float sc = (training && p > 0.f) ? 1.f / (1.f - p) : 1.f;
        if (on) {
#pragma unroll
            for (int i = 0; i < SB_RC; ++i)
                if (mc[i] >= 0) {
                    const int m = mc[i];
                    const float4 h = hc[i];
                    const float4 k = keep4(mask, seed, layer, m, c, C, p, training != 0);
                    st4(act_out + (size_t)m * C + c,
                        make_float4(fmaxf((h.x - mu.x) * rs.x, 0.f) * k.x * sc, fmaxf((h.y - mu.y) * rs.y, 0.f) * k.y * sc,
                                    fmaxf((h.z - mu.z) * rs.z, 0.f) * k.z * sc, fmaxf((h.w - mu.w) * rs.w, 0.f) * k.w * sc));
                }
            for (int r = rest; r < end; r += SB_SLOTS) {
                const int m = order ? order[r] : r;
                const float4 h = pre.load(m, c);
                const float4 k = keep4(mask, seed, layer, m, c, C, p, training != 0);
                st4(act_out + (size_t)m * C + c,
                    make_float4(fmaxf((h.x - mu.x) * rs.x, 0.f) * k.x * sc, fmaxf((h.y - mu.y) * rs.y, 0.f) * k.y * sc,
                                fmaxf((h.z - mu.z) * rs.z, 0.f) * k.z * sc, fmaxf((h.w - mu.w) * rs.w, 0.f) * k.w * sc));
            }
        }
    }
}

// a[m,c] = relu((h - mean) * rstd) * keep / (1-p)     (elementwise over [M, C]); RELU=false: plain norm
template <bool EDGE, bool RELU>
__global__ void k_norm_apply(PreAct<EDGE> pre, const int32_t* __restrict__ row_seg, const float* __restrict__ mean,
                             const float* __restrict__ rstd, const float* __restrict__ mask, SeedRef seed, int layer,
                             float p, int training, int64_t M, float* __restrict__ out) {
    const int C = pre.C, C4 = C >> 2;
    const float sc = (training && p > 0.f) ? 1.f / (1.f - p) : 1.f;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < M * C4; i += (int64_t)gridDim.x * blockDim.x) {
        const int m = (int)(i / C4), c = (int)(i % C4) * 4;
        const int g = row_seg[m];
        float4 h = pre.load(m, c);
        float4 mu = ld4(mean + (size_t)g * C + c), rs = ld4(rstd + (size_t)g * C + c);
        float4 y = make_float4((h.x - mu.x) * rs.x, (h.y - mu.y) * rs.y, (h.z - mu.z) * rs.z, (h.w - mu.w) * rs.w);
        if (RELU) {
            float4 k = keep4(mask, seed, layer, m, c, C, p, training != 0);
            y = make_float4(fmaxf(y.x, 0.f) * k.x * sc, fmaxf(y.y, 0.f) * k.y * sc, fmaxf(y.z, 0.f) * k.z * sc, fmaxf(y.w, 0.f) * k.w * sc);
        }
        st4(out + (size_t)m * C + c, y);
    }
}

// head: z[m] = sum_c relu(norm(h2+b2))*keep/(1-p)*w3[c] + b3 ; att = sigmoid(z (+ logit noise))
template <int LPR>
__global__ __launch_bounds__(256) void k_head_fwd(const float* __restrict__ h2, const float* __restrict__ b2,
                                                  const int32_t* __restrict__ row_seg, const float* __restrict__ mean,
                                                  const float* __restrict__ rstd, const float* __restrict__ mask, SeedRef seed,
                                                  float p, int training, const float* __restrict__ w3, const float* __restrict__ b3,
                                                  const float* __restrict__ u, int noise_philox, int64_t M, int C, float* __restrict__ logits,
                                                  float* __restrict__ att) {
    const int lane = threadIdx.x % LPR;
    const float sc = (training && p > 0.f) ? 1.f / (1.f - p) : 1.f;
    PreAct<false> pre{h2, nullptr, b2, nullptr, nullptr, C};
    for (int64_t m = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) / LPR; m < M; m += (int64_t)gridDim.x * blockDim.x / LPR) {
        const int g = row_seg[m];
        float acc = 0.f;
        for (int c = lane * 4; c < C; c += LPR * 4) {
            float4 h = pre.load((int)m, c);
            float4 mu = ld4(mean + (size_t)g * C + c), rs = ld4(rstd + (size_t)g * C + c);
            float4 k = keep4(mask, seed, 2, (int)m, c, C, p, training != 0);
            float4 w = ld4(w3 + c);
            acc = fmaf(fmaxf((h.x - mu.x) * rs.x, 0.f) * k.x * sc, w.x, acc);
            acc = fmaf(fmaxf((h.y - mu.y) * rs.y, 0.f) * k.y * sc, w.y, acc);
            acc = fmaf(fmaxf((h.z - mu.z) * rs.z, 0.f) * k.z * sc, w.z, acc);
            acc = fmaf(fmaxf((h.w - mu.w) * rs.w, 0.f) * k.w * sc, w.w, acc);
        }
        acc = group_sum<LPR>(acc);
        if (lane == 0) {
            const float z = acc + b3[0];
            logits[m] = z;
            if (att) {
                float t = z;
                if (training && (u || noise_philox)) {          // concrete sample; the noise is drawn here when the caller passed none
                    const float uu = u ? u[m] : philox_noise_u(seed.get(), (int)m);
                    t = z + (logf(uu) - logf(1.0f - uu));
                }
                att[m] = 1.f / (1.f + expf(-t));
            }
        }
    }
}

// dz[m] = dlogits[m] + datt[m] * att[m] * (1 - att[m]); the same launch zeroes the two bias gradients that are identically 0
__global__ __launch_bounds__(256) void k_dz(const float* __restrict__ dlogits, const float* __restrict__ datt, const float* __restrict__ att,
                                            int64_t M, float* __restrict__ dz, float* __restrict__ zero_a, int na, float* __restrict__ zero_b, int nb,
                                            float* __restrict__ block_sum = nullptr /* [gridDim.x]: sum of the block's 256 dz values */) {
    int64_t m = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t total = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = m; i < na; i += total) zero_a[i] = 0.f;
    for (int64_t i = m; i < nb; i += total) zero_b[i] = 0.f;
    float v = 0.f;
    if (m < M) {
        v = dlogits ? dlogits[m] : 0.f;
        if (datt) { float a = att[m]; v = fmaf(datt[m], a * (1.f - a), v); }
        dz[m] = v;
    }
    if (block_sum) {                      // fixed order: 64-lane butterfly, then the four waves in order
        __shared__ float ws[4];
        const float w = group_sum<64>(v);
        if ((threadIdx.x & 63) == 0) ws[threadIdx.x >> 6] = w;
        __syncthreads();
        if (threadIdx.x == 0) block_sum[blockIdx.x] = ((ws[0] + ws[1]) + ws[2]) + ws[3];
    }
}

// Backward statistics of layer 2 (through the head): per (graph, channel)
//   S1 = mean_r dy2 , S2 = mean_r dy2*yhat2 , dw3 partial = sum_r dz * a2
// APPLY (unsliced segments only): a second pass over the cache-hot rows writes dh2 = rstd2 * (dy2 - S1 - yhat2 * S2), i.e. k_dh2
// dz of row m from the two upstream gradients (what k_dz writes): dlogits[m] + datt[m] * att[m] * (1 - att[m])
struct DzSrc {
    const float* dz;                 // precomputed (sliced segments), or NULL: evaluate from the three vectors below
    const float *dlogits, *datt, *att;
    __device__ __forceinline__ float get(int m) const {
        if (dz) return dz[m];
        float v = dlogits ? dlogits[m] : 0.f;
        if (datt) { const float a = att[m]; v = fmaf(datt[m], a * (1.f - a), v); }
        return v;
    }
};

template <bool APPLY>
__global__ __launch_bounds__(SB) void k_head_bwd_stats(const float* __restrict__ h2, const float* __restrict__ b2,
                                                       const int32_t* __restrict__ seg_ptr, const int32_t* __restrict__ order,
                                                       const float* __restrict__ mean, const float* __restrict__ rstd,
                                                       const float* __restrict__ mask, SeedRef seed, float p, int training,
                                                       const float* __restrict__ w3, const DzSrc dzs, int C,
                                                       float* __restrict__ S1, float* __restrict__ S2, float* __restrict__ dw3p,
                                                       float* __restrict__ dh2 = nullptr) {
    __shared__ float4 sm[SB_SLOTS][SB_LANES];
    __shared__ float4 bc1[SB_LANES], bc2[SB_LANES];
    const int g = blockIdx.x;
    const int lane = threadIdx.x % SB_LANES, slot = threadIdx.x / SB_LANES;
    const int c = (blockIdx.y * SB_LANES + lane) * 4;
    const bool on = c < C;
    int beg, end; float inv_n;
    seg_slice(seg_ptr, g, &beg, &end, &inv_n);
    if (gridDim.z > 1) inv_n = 1.f;                         // raw partial sums, scaled by k_zcombine
    const size_t orow = gridDim.z > 1 ? (size_t)g * gridDim.z + blockIdx.z : (size_t)g;
    const float sc = (training && p > 0.f) ? 1.f / (1.f - p) : 1.f;
    PreAct<false> pre{h2, nullptr, b2, nullptr, nullptr, C};
    float4 a1 = f4zero(), a2 = f4zero(), a3 = f4zero();
    if (on) {
        const float4 mu = ld4(mean + (size_t)g * C + c), rs = ld4(rstd + (size_t)g * C + c), w = ld4(w3 + c);
        for (int r = beg + slot; r < end; r += SB_SLOTS) {
            const int m = order ? order[r] : r;
            const float d = dzs.get(m);
            float4 h = pre.load(m, c);
            float4 k = keep4(mask, seed, 2, m, c, C, p, training != 0);
            float4 y = make_float4((h.x - mu.x) * rs.x, (h.y - mu.y) * rs.y, (h.z - mu.z) * rs.z, (h.w - mu.w) * rs.w);
            float4 dy = make_float4(y.x > 0.f ? d * w.x * k.x * sc : 0.f, y.y > 0.f ? d * w.y * k.y * sc : 0.f,
                                    y.z > 0.f ? d * w.z * k.z * sc : 0.f, y.w > 0.f ? d * w.w * k.w * sc : 0.f);
            a1.x += dy.x; a1.y += dy.y; a1.z += dy.z; a1.w += dy.w;
            a2.x = fmaf(dy.x, y.x, a2.x); a2.y = fmaf(dy.y, y.y, a2.y); a2.z = fmaf(dy.z, y.z, a2.z); a2.w = fmaf(dy.w, y.w, a2.w);
            a3.x = fmaf(d, fmaxf(y.x, 0.f) * k.x * sc, a3.x); a3.y = fmaf(d, fmaxf(y.y, 0.f) * k.y * sc, a3.y);
            a3.z = fmaf(d, fmaxf(y.z, 0.f) * k.z * sc, a3.z); a3.w = fmaf(d, fmaxf(y.w, 0.f) * k.w * sc, a3.w);
        }
    }
    float4 t1 = slot_reduce(a1, sm, slot, lane);
    float4 t2 = slot_reduce(a2, sm, slot, lane);
    float4 t3 = slot_reduce(a3, sm, slot, lane);
    t1 = make_float4(t1.x * inv_n, t1.y * inv_n, t1.z * inv_n, t1.w * inv_n);
    t2 = make_float4(t2.x * inv_n, t2.y * inv_n, t2.z * inv_n, t2.w * inv_n);
    if (slot == 0 && on) {
        st4(S1 + orow * C + c, t1);
        st4(S2 + orow * C + c, t2);
        st4(dw3p + orow * C + c, t3);
    }
    if (APPLY) {
        if (slot == 0) { bc1[lane] = t1; bc2[lane] = t2; }
        __syncthreads();
        if (!on) return;
        const float4 s1 = bc1[lane], s2 = bc2[lane];
        const float4 mu = ld4(mean + (size_t)g * C + c), rs = ld4(rstd + (size_t)g * C + c), w = ld4(w3 + c);
        for (int r = beg + slot; r < end; r += SB_SLOTS) {
            const int m = order ? order[r] : r;
            const float d = dzs.get(m);
            const float4 h = pre.load(m, c);
            const float4 k = keep4(mask, seed, 2, m, c, C, p, training != 0);
            const float4 y = make_float4((h.x - mu.x) * rs.x, (h.y - mu.y) * rs.y, (h.z - mu.z) * rs.z, (h.w - mu.w) * rs.w);
            const float4 dy = make_float4(y.x > 0.f ? d * w.x * k.x * sc : 0.f, y.y > 0.f ? d * w.y * k.y * sc : 0.f,
                                          y.z > 0.f ? d * w.z * k.z * sc : 0.f, y.w > 0.f ? d * w.w * k.w * sc : 0.f);
            st4(dh2 + (size_t)m * C + c, make_float4(rs.x * (dy.x - s1.x - y.x * s2.x), rs.y * (dy.y - s1.y - y.y * s2.y),
                                                      rs.z * (dy.z - s1.z - y.z * s2.z), rs.w * (dy.w - s1.w - y.w * s2.w)));
        }
    }
}

// dh2[m,c] = rstd2 * (dy2 - S1 - yhat2 * S2)
__global__ void k_dh2(const float* __restrict__ h2, const float* __restrict__ b2, const int32_t* __restrict__ row_seg,
                      const float* __restrict__ mean, const float* __restrict__ rstd, const float* __restrict__ mask, SeedRef seed,
                      float p, int training, const float* __restrict__ w3, const float* __restrict__ dz,
                      const float* __restrict__ S1, const float* __restrict__ S2, int64_t M, int C, float* __restrict__ dh2) {
    const int C4 = C >> 2;
    const float sc = (training && p > 0.f) ? 1.f / (1.f - p) : 1.f;
    PreAct<false> pre{h2, nullptr, b2, nullptr, nullptr, C};
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < M * C4; i += (int64_t)gridDim.x * blockDim.x) {
        const int m = (int)(i / C4), c = (int)(i % C4) * 4;
        const int g = row_seg[m];
        const float d = dz[m];
        float4 h = pre.load(m, c);
        float4 mu = ld4(mean + (size_t)g * C + c), rs = ld4(rstd + (size_t)g * C + c), w = ld4(w3 + c);
        float4 k = keep4(mask, seed, 2, m, c, C, p, training != 0);
        float4 s1 = ld4(S1 + (size_t)g * C + c), s2 = ld4(S2 + (size_t)g * C + c);
        float4 y = make_float4((h.x - mu.x) * rs.x, (h.y - mu.y) * rs.y, (h.z - mu.z) * rs.z, (h.w - mu.w) * rs.w);
        float4 dy = make_float4(y.x > 0.f ? d * w.x * k.x * sc : 0.f, y.y > 0.f ? d * w.y * k.y * sc : 0.f,
                                y.z > 0.f ? d * w.z * k.z * sc : 0.f, y.w > 0.f ? d * w.w * k.w * sc : 0.f);
        st4(dh2 + (size_t)m * C + c, make_float4(rs.x * (dy.x - s1.x - y.x * s2.x), rs.y * (dy.y - s1.y - y.y * s2.y),
                                                  rs.z * (dy.z - s1.z - y.z * s2.z), rs.w * (dy.w - s1.w - y.w * s2.w)));
    }
}

// Backward statistics of layer 1 from (da1, a1):  dy1 = da1*[a1>0]*sc ; yhat1*[a1>0] = a1/sc
// APPLY (unsliced segments only): a second pass writes dh1 in place over da1, i.e. k_dh1
template <bool APPLY, bool EDGE>
__global__ __launch_bounds__(SB) void k_l1_bwd_stats(float* __restrict__ da1, const float* __restrict__ a1,
                                                     const int32_t* __restrict__ seg_ptr, const int32_t* __restrict__ order,
                                                     float sc, int C, float* __restrict__ S1, float* __restrict__ S2,
                                                     PreAct<EDGE> pre = PreAct<EDGE>{}, const float* __restrict__ mean = nullptr,
                                                     const float* __restrict__ rstd = nullptr) {
    __shared__ float4 sm[SB_SLOTS][SB_LANES];
    __shared__ float4 bc1[SB_LANES], bc2[SB_LANES];
    const int g = blockIdx.x;
    const int lane = threadIdx.x % SB_LANES, slot = threadIdx.x / SB_LANES;
    const int c = (blockIdx.y * SB_LANES + lane) * 4;
    const bool on = c < C;
    int beg, end; float inv_n;
    seg_slice(seg_ptr, g, &beg, &end, &inv_n);
    if (gridDim.z > 1) inv_n = 1.f;                         // raw partial sums, scaled by k_zcombine
    const size_t orow = gridDim.z > 1 ? (size_t)g * gridDim.z + blockIdx.z : (size_t)g;
    const float inv_sc = 1.f / sc;
    float4 s1 = f4zero(), s2 = f4zero();
    // da1 / a1 (and, for APPLY, the pre-activation) of the first SB_RC rows per slot stay in registers (see k_seg_stats)
    float4 dcache[SB_RC], acache[SB_RC], hcache[SB_RC];
    int mc[SB_RC];
#pragma unroll
    for (int i = 0; i < SB_RC; ++i) {
        const int r = beg + slot + i * SB_SLOTS;
        mc[i] = (r < end && on) ? (order ? order[r] : r) : -1;
    }
#pragma unroll
    for (int i = 0; i < SB_RC; ++i) {
        dcache[i] = mc[i] >= 0 ? ld4(da1 + (size_t)mc[i] * C + c) : f4zero();
        acache[i] = mc[i] >= 0 ? ld4(a1 + (size_t)mc[i] * C + c) : f4zero();
        if (APPLY) hcache[i] = mc[i] >= 0 ? pre.load(mc[i], c) : f4zero();
    }
    const int rest = beg + slot + SB_RC * SB_SLOTS;
#define GSAT_L1_DY(D4, A4)                                                                                                         \
    const float4 dy = make_float4((A4).x > 0.f ? (D4).x * sc : 0.f, (A4).y > 0.f ? (D4).y * sc : 0.f,                                \
                                  (A4).z > 0.f ? (D4).z * sc : 0.f, (A4).w > 0.f ? (D4).w * sc : 0.f);
#define GSAT_L1_ACC(D4, A4)                                                                                                        \
    {                                                                                                                              \
        GSAT_L1_DY(D4, A4)                                                                                                         \
        s1.x += dy.x; s1.y += dy.y; s1.z += dy.z; s1.w += dy.w;                                                                    \
        s2.x = fmaf(dy.x, (A4).x * inv_sc, s2.x); s2.y = fmaf(dy.y, (A4).y * inv_sc, s2.y);                                          \
        s2.z = fmaf(dy.z, (A4).z * inv_sc, s2.z); s2.w = fmaf(dy.w, (A4).w * inv_sc, s2.w);                                          \
    }
    if (on) {
#pragma unroll
        for (int i = 0; i < SB_RC; ++i)
            if (mc[i] >= 0) GSAT_L1_ACC(dcache[i], acache[i])
        for (int r = rest; r < end; r += SB_SLOTS) {
            const int m = order ? order[r] : r;
            const float4 d = ld4(da1 + (size_t)m * C + c), a = ld4(a1 + (size_t)m * C + c);
            GSAT_L1_ACC(d, a)
        }
    }
#undef GSAT_L1_ACC
    float4 t1 = slot_reduce(s1, sm, slot, lane);
    float4 t2 = slot_reduce(s2, sm, slot, lane);
    t1 = make_float4(t1.x * inv_n, t1.y * inv_n, t1.z * inv_n, t1.w * inv_n);
    t2 = make_float4(t2.x * inv_n, t2.y * inv_n, t2.z * inv_n, t2.w * inv_n);
    if (slot == 0 && on) {
        st4(S1 + orow * C + c, t1);
        st4(S2 + orow * C + c, t2);
    }
    if (APPLY) {
        if (slot == 0) { bc1[lane] = t1; bc2[lane] = t2; }
        __syncthreads();
        if (!on) return;
        const float4 q1 = bc1[lane], q2 = bc2[lane];
        const float4 mu = ld4(mean + (size_t)g * C + c), rs = ld4(rstd + (size_t)g * C + c);
#define GSAT_L1_OUT(M_, H4, D4, A4)                                                                                                \
        {                                                                                                                          \
            const float4 y = make_float4(((H4).x - mu.x) * rs.x, ((H4).y - mu.y) * rs.y, ((H4).z - mu.z) * rs.z, ((H4).w - mu.w) * rs.w); \
            GSAT_L1_DY(D4, A4)                                                                                                     \
            st4(da1 + (size_t)(M_) * C + c, make_float4(rs.x * (dy.x - q1.x - y.x * q2.x), rs.y * (dy.y - q1.y - y.y * q2.y),        \
                                                         rs.z * (dy.z - q1.z - y.z * q2.z), rs.w * (dy.w - q1.w - y.w * q2.w)));     \
        }
#pragma unroll
        for (int i = 0; i < SB_RC; ++i)
            if (mc[i] >= 0) GSAT_L1_OUT(mc[i], hcache[i], dcache[i], acache[i])
        for (int r = rest; r < end; r += SB_SLOTS) {
            const int m = order ? order[r] : r;
            const float4 h = pre.load(m, c);
            const float4 d = ld4(da1 + (size_t)m * C + c), a = ld4(a1 + (size_t)m * C + c);
            GSAT_L1_OUT(m, h, d, a)
        }
#undef GSAT_L1_OUT
    }
#undef GSAT_L1_DY
}

// dh1[m,c] = rstd1 * (dy1 - S1 - yhat1*S2), written in place over da1 ; yhat1 recomputed from the pre-activation
template <bool EDGE>
__global__ void k_dh1(PreAct<EDGE> pre, const int32_t* __restrict__ row_seg, const float* __restrict__ mean,
                      const float* __restrict__ rstd, const float* __restrict__ a1, float sc, const float* __restrict__ S1,
                      const float* __restrict__ S2, int64_t M, float* __restrict__ da1_inout) {
    const int C = pre.C, C4 = C >> 2;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < M * C4; i += (int64_t)gridDim.x * blockDim.x) {
        const int m = (int)(i / C4), c = (int)(i % C4) * 4;
        const int g = row_seg[m];
        float4 h = pre.load(m, c);
        float4 mu = ld4(mean + (size_t)g * C + c), rs = ld4(rstd + (size_t)g * C + c);
        float4 y = make_float4((h.x - mu.x) * rs.x, (h.y - mu.y) * rs.y, (h.z - mu.z) * rs.z, (h.w - mu.w) * rs.w);
        float4 d = ld4(da1_inout + (size_t)m * C + c), a = ld4(a1 + (size_t)m * C + c);
        float4 dy = make_float4(a.x > 0.f ? d.x * sc : 0.f, a.y > 0.f ? d.y * sc : 0.f, a.z > 0.f ? d.z * sc : 0.f, a.w > 0.f ? d.w * sc : 0.f);
        float4 s1 = ld4(S1 + (size_t)g * C + c), s2 = ld4(S2 + (size_t)g * C + c);
        st4(da1_inout + (size_t)m * C + c, make_float4(rs.x * (dy.x - s1.x - y.x * s2.x), rs.y * (dy.y - s1.y - y.y * s2.y),
                                                        rs.z * (dy.z - s1.z - y.z * s2.z), rs.w * (dy.w - s1.w - y.w * s2.w)));
    }
}

// generic InstanceNorm backward from saved y: dx = rstd * (dy - mean(dy) - y*mean(dy*y))
__global__ __launch_bounds__(SB) void k_in_bwd_stats(const float* __restrict__ y, const float* __restrict__ dy,
                                                     const int32_t* __restrict__ seg_ptr, const int32_t* __restrict__ order, int C,
                                                     float* __restrict__ S1, float* __restrict__ S2) {
    __shared__ float4 sm[SB_SLOTS][SB_LANES];
    const int g = blockIdx.x;
    const int lane = threadIdx.x % SB_LANES, slot = threadIdx.x / SB_LANES;
    const int c = (blockIdx.y * SB_LANES + lane) * 4;
    const bool on = c < C;
    int beg, end; float inv_n;
    seg_slice(seg_ptr, g, &beg, &end, &inv_n);
    if (gridDim.z > 1) inv_n = 1.f;                         // raw partial sums, scaled by k_zcombine
    const size_t orow = gridDim.z > 1 ? (size_t)g * gridDim.z + blockIdx.z : (size_t)g;
    float4 s1 = f4zero(), s2 = f4zero();
    if (on)
        for (int r = beg + slot; r < end; r += SB_SLOTS) {
            const int m = order ? order[r] : r;
            float4 d = ld4(dy + (size_t)m * C + c), v = ld4(y + (size_t)m * C + c);
            s1.x += d.x; s1.y += d.y; s1.z += d.z; s1.w += d.w;
            s2.x = fmaf(d.x, v.x, s2.x); s2.y = fmaf(d.y, v.y, s2.y); s2.z = fmaf(d.z, v.z, s2.z); s2.w = fmaf(d.w, v.w, s2.w);
        }
    float4 t1 = slot_reduce(s1, sm, slot, lane);
    float4 t2 = slot_reduce(s2, sm, slot, lane);
    if (slot == 0 && on) {
        st4(S1 + orow * C + c, make_float4(t1.x * inv_n, t1.y * inv_n, t1.z * inv_n, t1.w * inv_n));
        st4(S2 + orow * C + c, make_float4(t2.x * inv_n, t2.y * inv_n, t2.z * inv_n, t2.w * inv_n));
    }
}

__global__ void k_in_bwd_apply(const float* __restrict__ y, const float* __restrict__ dy, const int32_t* __restrict__ row_seg,
                               const float* __restrict__ rstd, const float* __restrict__ S1, const float* __restrict__ S2,
                               int64_t M, int C, float* __restrict__ dx) {
    const int C4 = C >> 2;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < M * C4; i += (int64_t)gridDim.x * blockDim.x) {
        const int m = (int)(i / C4), c = (int)(i % C4) * 4;
        const int g = row_seg[m];
        float4 v = ld4(y + (size_t)m * C + c), d = ld4(dy + (size_t)m * C + c);
        float4 rs = ld4(rstd + (size_t)g * C + c), s1 = ld4(S1 + (size_t)g * C + c), s2 = ld4(S2 + (size_t)g * C + c);
        st4(dx + (size_t)m * C + c, make_float4(rs.x * (d.x - s1.x - v.x * s2.x), rs.y * (d.y - s1.y - v.y * s2.y),
                                                 rs.z * (d.z - s1.z - v.z * s2.z), rs.w * (d.w - s1.w - v.w * s2.w)));
    }
}

// Deterministic column sum: out[rb, c] = sum over the rb-th row block; grid (ceil(C/64), RB)
__global__ __launch_bounds__(SB) void k_colsum(const float* __restrict__ x, int64_t R, int C, int64_t rows_per_block,
                                               float* __restrict__ out) {
    __shared__ float4 sm[SB_SLOTS][SB_LANES];
    const int lane = threadIdx.x % SB_LANES, slot = threadIdx.x / SB_LANES;
    const int c = (blockIdx.x * SB_LANES + lane) * 4;
    const int64_t beg = (int64_t)blockIdx.y * rows_per_block, end = min(R, beg + rows_per_block);
    float4 acc = f4zero();
    if (c + 3 < C) {
        for (int64_t r = beg + slot; r < end; r += SB_SLOTS) {
            float4 v = ld4(x + (size_t)r * C + c);
            acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
        }
    } else if (c < C) {       // ragged tail (C not a multiple of 4, e.g. C == 1)
        for (int64_t r = beg + slot; r < end; r += SB_SLOTS) {
            const float* p = x + (size_t)r * C + c;
            acc.x += p[0];
            if (c + 1 < C) acc.y += p[1];
            if (c + 2 < C) acc.z += p[2];
        }
    }
    float4 t = slot_reduce(acc, sm, slot, lane);
    if (slot == 0 && c < C) {
        float* o = out + (size_t)blockIdx.y * C + c;
        o[0] = t.x;
        if (c + 1 < C) o[1] = t.y;
        if (c + 2 < C) o[2] = t.z;
        if (c + 3 < C) o[3] = t.w;
    }
}

// Tail of the head's backward for unsliced segments, ONE launch instead of four column-sum launches: blocks [0, ceil(C/64)):
// dW3[c] = sum over the graphs of the per-graph partials of k_head_bwd_stats; last block: db3 = sum of k_dz's per-block sums.
// 64 row slots x 16 lanes per block, eight rows in flight per slot; fixed summation order.
constexpr int FIN_T = 1024, FIN_SLOTS = FIN_T / SB_LANES;
__global__ __launch_bounds__(FIN_T) void k_head_bwd_finish(const float* __restrict__ dw3p, const float* __restrict__ dz_part, int64_t G, int ndz, int C,
                                                           float* __restrict__ dW3, float* __restrict__ db3) {
    __shared__ float4 sm[FIN_SLOTS][SB_LANES];
    const int lane = threadIdx.x % SB_LANES, slot = threadIdx.x / SB_LANES;
    const int ctiles = (C + 4 * SB_LANES - 1) / (4 * SB_LANES);
    float4 acc = f4zero();
    if ((int)blockIdx.x < ctiles) {
        const int c = (blockIdx.x * SB_LANES + lane) * 4;
        if (c < C) {
            int64_t g = slot;
            for (; g + 7 * FIN_SLOTS < G; g += 8 * FIN_SLOTS) {
                float4 v[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) v[j] = ld4(dw3p + (size_t)(g + j * FIN_SLOTS) * C + c);
#pragma unroll
                for (int j = 0; j < 8; ++j) { acc.x += v[j].x; acc.y += v[j].y; acc.z += v[j].z; acc.w += v[j].w; }
            }
            for (; g < G; g += FIN_SLOTS) { const float4 v = ld4(dw3p + (size_t)g * C + c); acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w; }
        }
    } else {
        for (int i = threadIdx.x; i < ndz; i += FIN_T) acc.x += dz_part[i];
    }
    sm[slot][lane] = acc;
    __syncthreads();
    if (slot == 0) {
        float4 r = f4zero();
        for (int s_ = 0; s_ < FIN_SLOTS; ++s_) { const float4 t = sm[s_][lane]; r.x += t.x; r.y += t.y; r.z += t.z; r.w += t.w; }
        if ((int)blockIdx.x < ctiles) {
            const int c = (blockIdx.x * SB_LANES + lane) * 4;
            if (c < C) st4(dW3 + c, r);
        } else {
            sm[0][lane] = r;
        }
    }
    if ((int)blockIdx.x >= ctiles) {
        __syncthreads();
        if (threadIdx.x == 0) {
            float r = 0.f;
            for (int l = 0; l < SB_LANES; ++l) r += sm[0][l].x;
            db3[0] = r;
        }
    }
}

__global__ void k_philox_noise(uint64_t seed, int64_t M, float* __restrict__ u) {
    const int64_t m = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (m < M) u[m] = philox_noise_u(seed, (int)m);
}

__global__ void k_philox_mask(uint64_t seed, int layer, int64_t M, int C, float p, float* __restrict__ keep) {
    const int C4 = C >> 2;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < M * C4; i += (int64_t)gridDim.x * blockDim.x) {
        const int m = (int)(i / C4), c = (int)(i % C4) * 4;
        st4(keep + (size_t)m * C + c, keep4(nullptr, SeedRef{seed, nullptr}, layer, m, c, C, p, true));
    }
}

static inline int ew_blocks(int64_t work_items) {
    int64_t nb = ceil_div(work_items, 256);
    if (nb > 256 * 32) nb = 256 * 32;
    return (int)std::max<int64_t>(nb, 1);
}

static int colsum(hipStream_t stream, const float* x, int64_t R, int C, float* out, float* scratch /* [256*C] */) {
    if (C <= 0) return GSAT_OK;
    const int ctiles = (int)ceil_div(C, 4 * SB_LANES);
    if (R <= 0) { GSAT_CHECK_HIP(gsat::zero_async(out, sizeof(float) * C, stream)); return GSAT_OK; }
    int64_t RB = std::min<int64_t>(256, ceil_div(R, 64));
    if (RB <= 1) {
        k_colsum<<<dim3(ctiles, 1), SB, 0, stream>>>(x, R, C, R, out);
    } else {
        int64_t rpb = ceil_div(R, RB);
        RB = ceil_div(R, rpb);
        k_colsum<<<dim3(ctiles, (unsigned)RB), SB, 0, stream>>>(x, R, C, rpb, scratch);
        k_colsum<<<dim3(ctiles, 1), SB, 0, stream>>>(scratch, RB, C, RB, out);
    }
    GSAT_LAUNCH_CHECK();
    return GSAT_OK;
}

// number of row slices per segment for the segmented statistics: 1 for batches of small graphs, up to 64 when a
// graph has tens of thousands of rows (C5); must agree between the workspace queries and the launches.
static inline int seg_slices(int64_t rows, int64_t G) {
    const int64_t avg = rows / std::max<int64_t>(G, 1);
    return (int)std::max<int64_t>(1, std::min<int64_t>(64, ceil_div(avg, 4096)));
}

template <bool EDGE>
static int launch_seg_stats(hipStream_t stream, PreAct<EDGE> pre, const int32_t* seg_ptr, const int32_t* order, int64_t G, int Z,
                            float* mean, float* rstd, float* part) {
    if (G <= 0) return GSAT_OK;
    const int C = pre.C;
    const dim3 grid((unsigned)G, (unsigned)ceil_div(C, 4 * SB_LANES), (unsigned)Z);
    if (Z == 1) {
        k_seg_stats<EDGE, false><<<grid, SB, 0, stream>>>(pre, seg_ptr, order, mean, rstd);
    } else {
        GSAT_REQUIRE(part, GSAT_ERR_WORKSPACE, "segmented statistics need a workspace for sliced segments");
        const unsigned cb = (unsigned)ceil_div(G * C, 256);
        k_seg_partial<EDGE><<<grid, SB, 0, stream>>>(pre, seg_ptr, order, nullptr, part);
        k_zcombine<<<cb, 256, 0, stream>>>(part, seg_ptr, (int)G, Z, C, 1, mean);
        k_seg_partial<EDGE><<<grid, SB, 0, stream>>>(pre, seg_ptr, order, mean, part);
        k_zcombine<<<cb, 256, 0, stream>>>(part, seg_ptr, (int)G, Z, C, 2, rstd);
    }
    GSAT_LAUNCH_CHECK();
    return GSAT_OK;
}

// split-K slabs of the weight-gradient GEMMs: dW2, dW1 (edge mode: its two halves).  Each keeps its own region, because their ordered
// sums are deferred to ONE launch at the end of the backward (slab_reduce_jobs)
static size_t attn_gemm_ws_w2(const gsat_attn_args* a) { return align_up(gemm_workspace_floats(a->C2, a->C1, a->M, true), 64); }
static size_t attn_gemm_ws_w1(const gsat_attn_args* a) { return align_up(gemm_workspace_floats(a->C1, a->H, a->N, true), 64); }
static size_t attn_gemm_ws_floats(const gsat_attn_args* a) { return attn_gemm_ws_w2(a) + (a->edge_mode ? 2 : 1) * attn_gemm_ws_w1(a); }

static int check_args(const gsat_attn_args* a, const char* who) {
    GSAT_REQUIRE(a, GSAT_ERR_ARG, "%s: null args", who);
    GSAT_REQUIRE(a->M >= 0 && a->N >= 0 && a->G >= 0 && a->M < (1ll << 31) && a->N < (1ll << 31), GSAT_ERR_ARG, "%s: bad extents", who);
    GSAT_REQUIRE(a->H > 0 && a->C1 > 0 && a->C2 > 0 && a->H % 4 == 0 && a->C1 % 4 == 0 && a->C2 % 4 == 0, GSAT_ERR_UNSUPPORTED,
                 "%s: H, C1, C2 must be positive multiples of 4 (got %d, %d, %d)", who, a->H, a->C1, a->C2);
    GSAT_REQUIRE(a->p_drop >= 0.f && a->p_drop < 1.f, GSAT_ERR_ARG, "%s: dropout p must be in [0,1)", who);
    if (a->M == 0) return GSAT_OK;
    GSAT_REQUIRE(a->emb && a->W1 && a->b1 && a->W2 && a->b2 && a->W3 && a->b3 && a->seg_ptr && a->row_seg, GSAT_ERR_ARG, "%s: null input", who);
    GSAT_REQUIRE(a->P && a->h2 && a->stats && a->logits, GSAT_ERR_ARG, "%s: null output/saved buffer", who);
    if (a->edge_mode) GSAT_REQUIRE(a->src && a->dst && a->Q, GSAT_ERR_ARG, "%s: edge mode needs src, dst, Q", who);
    else GSAT_REQUIRE(a->M == a->N, GSAT_ERR_ARG, "%s: node mode needs M == N", who);
    return GSAT_OK;
}

}  // namespace gsat

using namespace gsat;

extern "C" {

size_t gsat_colsum_workspace_floats(int64_t C) { return (size_t)256 * (size_t)(C > 0 ? C : 0); }

int gsat_colsum(const float* x, int64_t R, int64_t C, float* out, float* workspace, void* stream) {
    GSAT_REQUIRE(R >= 0 && C >= 0 && C < (1ll << 31), GSAT_ERR_ARG, "gsat_colsum: bad extents");
    if (C == 0) return GSAT_OK;
    GSAT_REQUIRE(out && (x || R == 0) && workspace, GSAT_ERR_ARG, "gsat_colsum: null pointer");
    return colsum((hipStream_t)stream, x, R, (int)C, out, workspace);
}


int gsat_attn_fwd(const gsat_attn_args* a, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    int rc = check_args(a, "gsat_attn_fwd");
    if (rc) return rc;
    if (a->M == 0) return GSAT_OK;
    {
        FusedGeom fg;
        if (attn_fused_eligible(a, &fg)) return attn_fused_fwd(stream, a, fg);        // one launch, whole graphs per workgroup (attn_fused.hip)
    }
    GSAT_REQUIRE(a->a1, GSAT_ERR_ARG, "gsat_attn_fwd: the staged pipeline needs the a1 buffer");
    const int64_t M = a->M, N = a->N, G = a->G;
    const int H = a->H, C1 = a->C1, C2 = a->C2;
    float* mean1 = a->stats;
    float* rstd1 = mean1 + (size_t)G * C1;
    float* mean2 = rstd1 + (size_t)G * C1;
    float* rstd2 = mean2 + (size_t)G * C2;
    // ---- layer 1 on nodes -------------------------------------------------------------------
    if (a->edge_mode) {
        if ((rc = gemm_rm(stream, false, true, N, C1, H, a->emb, H, a->W1, 2 * H, 0.f, a->P, C1))) return rc;        // P = emb W1[:, :H]^T
        if ((rc = gemm_rm(stream, false, true, N, C1, H, a->emb, H, a->W1 + H, 2 * H, 0.f, a->Q, C1))) return rc;    // Q = emb W1[:, H:]^T
    } else {
        if ((rc = gemm_rm(stream, false, true, N, C1, H, a->emb, H, a->W1, H, 0.f, a->P, C1))) return rc;
    }
    const int Z = seg_slices(M, G);
    float* part = nullptr;
    if (Z > 1) {
        const size_t need = (size_t)G * Z * std::max(C1, C2) * sizeof(float);
        GSAT_REQUIRE(a->fwd_workspace && a->fwd_workspace_bytes >= need, GSAT_ERR_WORKSPACE, "gsat_attn_fwd: workspace %zu < %zu", a->fwd_workspace_bytes, need);
        part = static_cast<float*>(a->fwd_workspace);
    }
    const dim3 sgrid((unsigned)G, (unsigned)ceil_div(C1, 4 * SB_LANES), 1);
    const SeedRef sref{a->seed, a->seed_dev};
    if (a->edge_mode) {
        PreAct<true> pre{a->P, a->Q, a->b1, a->src, a->dst, C1};
        if (Z == 1 && G > 0) {        // statistics and activation in one launch: the rows of the segment are re-read from cache
            k_seg_stats<true, true><<<sgrid, SB, 0, stream>>>(pre, a->seg_ptr, a->seg_order, mean1, rstd1, a->mask1, sref, 1, a->p_drop, a->training, a->a1);
        } else {
            if ((rc = launch_seg_stats<true>(stream, pre, a->seg_ptr, a->seg_order, G, Z, mean1, rstd1, part))) return rc;
            k_norm_apply<true, true><<<ew_blocks(M * (C1 / 4)), 256, 0, stream>>>(pre, a->row_seg, mean1, rstd1, a->mask1, sref, 1, a->p_drop, a->training, M, a->a1);
        }
    } else {
        PreAct<false> pre{a->P, nullptr, a->b1, nullptr, nullptr, C1};
        if (Z == 1 && G > 0) {
            k_seg_stats<false, true><<<sgrid, SB, 0, stream>>>(pre, a->seg_ptr, a->seg_order, mean1, rstd1, a->mask1, sref, 1, a->p_drop, a->training, a->a1);
        } else {
            if ((rc = launch_seg_stats<false>(stream, pre, a->seg_ptr, a->seg_order, G, Z, mean1, rstd1, part))) return rc;
            k_norm_apply<false, true><<<ew_blocks(M * (C1 / 4)), 256, 0, stream>>>(pre, a->row_seg, mean1, rstd1, a->mask1, sref, 1, a->p_drop, a->training, M, a->a1);
        }
    }
    GSAT_LAUNCH_CHECK();
    // ---- layer 2 ----------------------------------------------------------------------------
    if ((rc = gemm_rm(stream, false, true, M, C2, C1, a->a1, C1, a->W2, C1, 0.f, a->h2, C2))) return rc;
    {
        PreAct<false> pre{a->h2, nullptr, a->b2, nullptr, nullptr, C2};
        if ((rc = launch_seg_stats<false>(stream, pre, a->seg_ptr, a->seg_order, G, Z, mean2, rstd2, part))) return rc;
    }
    // ---- head + sampler ---------------------------------------------------------------------
    const int q = C2 / 4;
    const int lpr = q <= 4 ? 4 : q <= 8 ? 8 : q <= 16 ? 16 : q <= 32 ? 32 : 64;
    const int nb = (int)std::min<int64_t>(ceil_div(M, 256 / lpr), 256 * 32);
#define HEAD(L) k_head_fwd<L><<<nb, 256, 0, stream>>>(a->h2, a->b2, a->row_seg, mean2, rstd2, a->mask2, SeedRef{a->seed, a->seed_dev}, a->p_drop, a->training, a->W3, a->b3, a->u, a->noise_philox, M, C2, a->logits, a->att)
    switch (lpr) { case 4: HEAD(4); break; case 8: HEAD(8); break; case 16: HEAD(16); break; case 32: HEAD(32); break; default: HEAD(64); break; }
#undef HEAD
    GSAT_LAUNCH_CHECK();
    return GSAT_OK;
}

size_t gsat_attn_fwd_workspace_bytes(const gsat_attn_args* a) {
    if (!a) return 0;
    FusedGeom fg;
    if (attn_fused_eligible(a, &fg)) return fused_ws_bytes(fg, a->G);
    const int Z = seg_slices(a->M, a->G);
    return Z > 1 ? (size_t)a->G * Z * std::max(a->C1, a->C2) * sizeof(float) : 0;
}

int gsat_attn_fwd_kind(const gsat_attn_args* a) {
    if (!a) return 0;
    FusedGeom fg;
    if (!attn_fused_eligible(a, &fg)) return 0;
    return fg.x6 ? 2 : 1;
}

size_t gsat_attn_bwd_workspace_bytes(const gsat_attn_args* a) {
    if (!a) return 0;
    const size_t M = (size_t)a->M, N = (size_t)a->N, G = (size_t)a->G, C1 = a->C1, C2 = a->C2;
    const size_t cmax = C1 > C2 ? C1 : C2;
    size_t b = 0;
    b += align_up((M > G ? M : G) * 4, 256);   // dz (or its per-graph sums)
    b += align_up(M * C2 * 4, 256);            // dh2
    b += align_up(M * C1 * 4, 256);            // da1 -> dh1
    b += 2 * align_up(G * C1 * 4, 256);        // S1', S2'
    b += 3 * align_up(G * C2 * 4, 256);        // S1, S2, dw3 partial
    b += align_up(256 * cmax * 4, 256);        // column-sum scratch
    { const size_t Z = seg_slices(a->M, a->G); if (Z > 1) b += 3 * align_up(G * Z * cmax * 4, 256); }   // sliced-segment partials
    if (a->edge_mode) b += 2 * align_up(N * C1 * 4, 256) + align_up(gsat_long_row_partial_floats(a->M, a->C1) * 4, 256);   // dP, dQ, hub partials
    b += align_up(attn_gemm_ws_floats(a) * 4, 256);          // split-K slabs of the weight-gradient GEMMs
    if (attn_fused_bwd_eligible(a)) b += align_up(attn_fused_bwd_ws_bytes(a), 256);      // tiles, W2 fragment stream, per-workgroup partials
    b += align_up(std::max(dual_gemm_ws_bytes(1, a->C2, a->C1, a->C1), dual_gemm_ws_bytes(2, a->C1, a->H, a->H)), 256);      // dual GEMMs: fragment stream + partial slabs
    return b + 1024;
}

int gsat_attn_bwd(const gsat_attn_args* a, const gsat_attn_grads* gr, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    int rc = check_args(a, "gsat_attn_bwd");
    if (rc) return rc;
    GSAT_REQUIRE(gr && gr->demb && gr->dW1 && gr->db1 && gr->dW2 && gr->db2 && gr->dW3 && gr->db3, GSAT_ERR_ARG, "gsat_attn_bwd: null gradient output");
    const int64_t M = a->M, N = a->N, G = a->G;
    const int H = a->H, C1 = a->C1, C2 = a->C2;
    const int C0 = a->edge_mode ? 2 * H : H;
    if (M == 0) {
        GSAT_CHECK_HIP(gsat::zero_async(gr->demb, sizeof(float) * N * H, stream));
        GSAT_CHECK_HIP(gsat::zero_async(gr->dW1, sizeof(float) * C1 * C0, stream));
        GSAT_CHECK_HIP(gsat::zero_async(gr->db1, sizeof(float) * C1, stream));
        GSAT_CHECK_HIP(gsat::zero_async(gr->dW2, sizeof(float) * C2 * C1, stream));
        GSAT_CHECK_HIP(gsat::zero_async(gr->db2, sizeof(float) * C2, stream));
        GSAT_CHECK_HIP(gsat::zero_async(gr->dW3, sizeof(float) * C2, stream));
        GSAT_CHECK_HIP(gsat::zero_async(gr->db3, sizeof(float), stream));
        return GSAT_OK;
    }
    GSAT_REQUIRE(gr->dlogits || gr->datt, GSAT_ERR_ARG, "gsat_attn_bwd: need dlogits and/or datt");
    GSAT_REQUIRE(!gr->datt || a->att, GSAT_ERR_ARG, "gsat_attn_bwd: datt given but att was not saved");
    if (a->edge_mode) GSAT_REQUIRE(gr->rowptr_src && gr->eid_by_src && gr->rowptr_dst && gr->eid_by_dst, GSAT_ERR_ARG, "gsat_attn_bwd: edge mode needs both CSRs");
    Arena ar(gr->workspace, gr->workspace_bytes);
    float* dz = ar.take<float>(std::max<int64_t>(M, G));
    float* dh2 = ar.take<float>((size_t)M * C2);
    float* da1 = ar.take<float>((size_t)M * C1);
    float* S1p = ar.take<float>((size_t)G * C1);
    float* S2p = ar.take<float>((size_t)G * C1);
    float* S1 = ar.take<float>((size_t)G * C2);
    float* S2 = ar.take<float>((size_t)G * C2);
    float* dw3p = ar.take<float>((size_t)G * C2);
    float* scratch = ar.take<float>((size_t)256 * std::max(C1, C2));
    const int Z = seg_slices(M, G);
    float *zp1 = nullptr, *zp2 = nullptr, *zp3 = nullptr;
    if (Z > 1) {
        const size_t n = (size_t)G * Z * std::max(C1, C2);
        zp1 = ar.take<float>(n); zp2 = ar.take<float>(n); zp3 = ar.take<float>(n);
    }
    float *dP = nullptr, *dQ = nullptr;
    float* lpart = nullptr;
    if (a->edge_mode) {
        dP = ar.take<float>((size_t)N * C1);
        dQ = ar.take<float>((size_t)N * C1);
        lpart = ar.take<float>(gsat_long_row_partial_floats(M, C1));
    }
    GemmWs gws{nullptr, attn_gemm_ws_w2(a)}, gws1{nullptr, attn_gemm_ws_w1(a)}, gws1b{nullptr, a->edge_mode ? attn_gemm_ws_w1(a) : 0};
    gws.ptr = ar.take<float>(gws.floats);
    gws1.ptr = ar.take<float>(gws1.floats);
    gws1b.ptr = ar.take<float>(gws1b.floats);
    SlabJob jobs[3] = {};                        // pending ordered sums of dW2, dW1 (| its two halves): one launch at the end
    const size_t dws_bytes = std::max(dual_gemm_ws_bytes(1, C2, C1, C1), dual_gemm_ws_bytes(2, C1, H, H));
    char* dws = ar.take<char>(dws_bytes);
    const bool fused_bwd = attn_fused_bwd_eligible(a) && seg_slices(M, G) == 1;
    const size_t fws_bytes = fused_bwd ? attn_fused_bwd_ws_bytes(a) : 0;
    char* fws = fused_bwd ? ar.take<char>(fws_bytes) : nullptr;
    GSAT_REQUIRE(ar.ok(), GSAT_ERR_WORKSPACE, "gsat_attn_bwd: workspace %zu < %zu", gr->workspace_bytes, ar.off);
    float* mean1 = a->stats;
    float* rstd1 = mean1 + (size_t)G * C1;
    float* mean2 = rstd1 + (size_t)G * C1;
    float* rstd2 = mean2 + (size_t)G * C2;
    const float sc = (a->training && a->p_drop > 0.f) ? 1.f / (1.f - a->p_drop) : 1.f;

    // b1 and b2 sit in front of an InstanceNorm, which removes any per-channel constant of its segment: their
    // gradients are exactly zero (the reference's autograd returns ~1e-7 rounding noise of a cancelling sum).
    const dim3 g2((unsigned)G, (unsigned)ceil_div(C2, 4 * SB_LANES), (unsigned)Z), g1((unsigned)G, (unsigned)ceil_div(C1, 4 * SB_LANES), (unsigned)Z);
    if (fused_bwd) {          // one launch from (dlogits, datt) down to dh1 (in the da1 buffer), dW2 / dW3 / db3 included (attn_fused_bwd.hip)
        if ((rc = attn_fused_bwd(stream, a, gr, da1, fws, fws_bytes))) return rc;
    } else {
    // env GSAT_ATTN_BWD_MERGED=0: the round-2 sequence (two-stage column sums for db3 and dW3) also for unsliced segments (A/B switch)
    const char* env_m = getenv("GSAT_ATTN_BWD_MERGED");
    const bool merged = Z == 1 && G > 0 && !(env_m && atoi(env_m) == 0);
    const unsigned dzb = (unsigned)ceil_div(M, 256);
    float* dzpart = scratch;                         // [dzb] per-block sums of dz; `scratch` holds 256 * max(C1, C2) >= M / 256 floats for M < 2^24 C
    const bool part_ok = (size_t)dzb <= (size_t)256 * std::max(C1, C2);
    // b1 and b2 sit in front of an InstanceNorm, which removes any per-channel constant of its segment: their
    // gradients are exactly zero (the reference's autograd returns ~1e-7 rounding noise of a cancelling sum).
    k_dz<<<dzb, 256, 0, stream>>>(gr->dlogits, gr->datt, a->att, M, dz, gr->db1, (int)C1, gr->db2, (int)C2, (merged && part_ok) ? dzpart : nullptr);
    GSAT_LAUNCH_CHECK();
    if (!(merged && part_ok) && (rc = colsum(stream, dz, M, 1, gr->db3, scratch))) return rc;
    // ---- through the head and the second InstanceNorm ------------------------------------------
    if (Z == 1)          // statistics and dh2 in one launch
        k_head_bwd_stats<true><<<g2, SB, 0, stream>>>(a->h2, a->b2, a->seg_ptr, a->seg_order, mean2, rstd2, a->mask2, SeedRef{a->seed, a->seed_dev}, a->p_drop,
                                                      a->training, a->W3, DzSrc{dz, nullptr, nullptr, nullptr}, C2, S1, S2, dw3p, dh2);
    else
        k_head_bwd_stats<false><<<g2, SB, 0, stream>>>(a->h2, a->b2, a->seg_ptr, a->seg_order, mean2, rstd2, a->mask2, SeedRef{a->seed, a->seed_dev}, a->p_drop,
                                                       a->training, a->W3, DzSrc{dz, nullptr, nullptr, nullptr}, C2, zp1, zp2, zp3);
    if (merged && part_ok)       // dW3 and db3 from the partials, one launch (instead of two two-stage column sums)
        k_head_bwd_finish<<<(unsigned)ceil_div(C2, 4 * SB_LANES) + 1, FIN_T, 0, stream>>>(dw3p, dzpart, G, (int)dzb, C2, gr->dW3, gr->db3);
    if (Z > 1) {
        const unsigned cb = (unsigned)ceil_div(G * C2, 256);
        k_zcombine<<<cb, 256, 0, stream>>>(zp1, a->seg_ptr, (int)G, Z, C2, 1, S1);
        k_zcombine<<<cb, 256, 0, stream>>>(zp2, a->seg_ptr, (int)G, Z, C2, 1, S2);
        k_zcombine<<<cb, 256, 0, stream>>>(zp3, a->seg_ptr, (int)G, Z, C2, 0, dw3p);
    }
    GSAT_LAUNCH_CHECK();
    if (!(merged && part_ok) && (rc = colsum(stream, dw3p, G, C2, gr->dW3, scratch))) return rc;
    if (Z > 1) {
        k_dh2<<<ew_blocks(M * (C2 / 4)), 256, 0, stream>>>(a->h2, a->b2, a->row_seg, mean2, rstd2, a->mask2, SeedRef{a->seed, a->seed_dev}, a->p_drop, a->training,
                                                           a->W3, dz, S1, S2, M, C2, dh2);
        GSAT_LAUNCH_CHECK();
    }
    // dW2[C2,C1] = dh2^T a1 ; da1[M,C1] = dh2 W2
    if (dual_gemm_ok(1, M, C2, C1, C1)) {            // both products in one pass over dh2 (dual_gemm.hip)
        if ((rc = dual_gemm(stream, 1, M, C2, C1, C1, dh2, C2, a->a1, C1, a->W2, C1, da1, C1, 0, gr->dW2, C1, dws, dws_bytes))) return rc;
    } else {
        if ((rc = gemm_rm(stream, true, false, C2, C1, M, dh2, C2, a->a1, C1, 0.f, gr->dW2, C1, gws, true, &jobs[0]))) return rc;
        if ((rc = gemm_rm(stream, false, false, M, C1, C2, dh2, C2, a->W2, C1, 0.f, da1, C1, GemmWs{nullptr, 0}, true))) return rc;
    }
    // ---- through ReLU/dropout and the first InstanceNorm ---------------------------------------
    if (Z == 1) {        // statistics and dh1 (in place over da1) in one launch
        if (a->edge_mode) {
            PreAct<true> pre1{a->P, a->Q, a->b1, a->src, a->dst, C1};
            k_l1_bwd_stats<true, true><<<g1, SB, 0, stream>>>(da1, a->a1, a->seg_ptr, a->seg_order, sc, C1, S1p, S2p, pre1, mean1, rstd1);
        } else {
            PreAct<false> pre1{a->P, nullptr, a->b1, nullptr, nullptr, C1};
            k_l1_bwd_stats<true, false><<<g1, SB, 0, stream>>>(da1, a->a1, a->seg_ptr, a->seg_order, sc, C1, S1p, S2p, pre1, mean1, rstd1);
        }
    } else {
        k_l1_bwd_stats<false, false><<<g1, SB, 0, stream>>>(da1, a->a1, a->seg_ptr, a->seg_order, sc, C1, zp1, zp2);
    }
    if (Z > 1) {
        const unsigned cb = (unsigned)ceil_div(G * C1, 256);
        k_zcombine<<<cb, 256, 0, stream>>>(zp1, a->seg_ptr, (int)G, Z, C1, 1, S1p);
        k_zcombine<<<cb, 256, 0, stream>>>(zp2, a->seg_ptr, (int)G, Z, C1, 1, S2p);
    }
    GSAT_LAUNCH_CHECK();
    }       // staged path
    if (a->edge_mode) {
        if (Z > 1) {
            PreAct<true> pre{a->P, a->Q, a->b1, a->src, a->dst, C1};
            k_dh1<true><<<ew_blocks(M * (C1 / 4)), 256, 0, stream>>>(pre, a->row_seg, mean1, rstd1, a->a1, sc, S1p, S2p, M, da1);
            GSAT_LAUNCH_CHECK();
        }
        // dP[n,:] = sum over out-edges of n of dh1[e,:], dQ[n,:] = sum over in-edges (gather-sum, hub rows chunked)
        if ((rc = aggr_sum_fwd_impl(stream, da1, nullptr, nullptr, nullptr, gr->rowptr_src, gr->eid_by_src, nullptr, N, M, C1, 0.f, dP,
                                    gr->chunk_ptr_src, lpart))) return rc;
        if ((rc = aggr_sum_fwd_impl(stream, da1, nullptr, nullptr, nullptr, gr->rowptr_dst, gr->eid_by_dst, nullptr, N, M, C1, 0.f, dQ,
                                    gr->chunk_ptr_dst, lpart))) return rc;
        // demb = dP W1a + dQ W1b ; dW1[:, :H] = dP^T emb ; dW1[:, H:] = dQ^T emb
        if (dual_gemm_ok(2, N, C1, H, H)) {
            if ((rc = dual_gemm(stream, 2, N, C1, H, H, dP, C1, a->emb, H, a->W1, 2 * H, gr->demb, H, 0, gr->dW1, 2 * H, dws, dws_bytes))) return rc;
            if ((rc = dual_gemm(stream, 2, N, C1, H, H, dQ, C1, a->emb, H, a->W1 + H, 2 * H, gr->demb, H, 1, gr->dW1 + H, 2 * H, dws, dws_bytes))) return rc;
        } else {
        if ((rc = gemm_rm(stream, false, false, N, H, C1, dP, C1, a->W1, 2 * H, 0.f, gr->demb, H, GemmWs{nullptr, 0}, true))) return rc;
        if ((rc = gemm_rm(stream, false, false, N, H, C1, dQ, C1, a->W1 + H, 2 * H, 1.f, gr->demb, H, GemmWs{nullptr, 0}, true))) return rc;
        if ((rc = gemm_rm(stream, true, false, C1, H, N, dP, C1, a->emb, H, 0.f, gr->dW1, 2 * H, gws1, true, &jobs[1]))) return rc;
        if ((rc = gemm_rm(stream, true, false, C1, H, N, dQ, C1, a->emb, H, 0.f, gr->dW1 + H, 2 * H, gws1b, true, &jobs[2]))) return rc;
        }
    } else {
        if (Z > 1) {
            PreAct<false> pre{a->P, nullptr, a->b1, nullptr, nullptr, C1};
            k_dh1<false><<<ew_blocks(M * (C1 / 4)), 256, 0, stream>>>(pre, a->row_seg, mean1, rstd1, a->a1, sc, S1p, S2p, M, da1);
            GSAT_LAUNCH_CHECK();
        }
        if (dual_gemm_ok(2, N, C1, H, H)) {
            if ((rc = dual_gemm(stream, 2, N, C1, H, H, da1, C1, a->emb, H, a->W1, H, gr->demb, H, 0, gr->dW1, H, dws, dws_bytes))) return rc;
        } else {
        if ((rc = gemm_rm(stream, false, false, N, H, C1, da1, C1, a->W1, H, 0.f, gr->demb, H, GemmWs{nullptr, 0}, true))) return rc;
        if ((rc = gemm_rm(stream, true, false, C1, H, N, da1, C1, a->emb, H, 0.f, gr->dW1, H, gws1, true, &jobs[1]))) return rc;
        }
    }
    return slab_reduce_jobs(stream, jobs, 3);
}

int gsat_instance_norm_fwd(const float* x, const int32_t* seg_ptr, const int32_t* seg_order, const int32_t* row_seg, int64_t M,
                           int64_t G, int64_t C, float* y, float* stats, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    GSAT_REQUIRE(M >= 0 && G >= 0 && C > 0 && C % 4 == 0 && M < (1ll << 31), GSAT_ERR_UNSUPPORTED, "gsat_instance_norm_fwd: C must be a positive multiple of 4");
    if (M == 0 || G == 0) return GSAT_OK;
    GSAT_REQUIRE(x && seg_ptr && row_seg && y && stats, GSAT_ERR_ARG, "gsat_instance_norm_fwd: null pointer");
    float* mean = stats;
    float* rstd = stats + (size_t)G * C;
    PreAct<false> pre{x, nullptr, nullptr, nullptr, nullptr, (int)C};
    k_seg_stats<false, false><<<dim3((unsigned)G, (unsigned)ceil_div(C, 4 * SB_LANES), 1), SB, 0, stream>>>(pre, seg_ptr, seg_order, mean, rstd);
    k_norm_apply<false, false><<<ew_blocks(M * (C / 4)), 256, 0, stream>>>(pre, row_seg, mean, rstd, nullptr, SeedRef{0, nullptr}, 0, 0.f, 0, M, y);
    GSAT_LAUNCH_CHECK();
    return GSAT_OK;
}

int gsat_instance_norm_bwd(const float* y, const float* dy, const float* stats, const int32_t* seg_ptr, const int32_t* seg_order,
                           const int32_t* row_seg, int64_t M, int64_t G, int64_t C, float* dx, float* workspace, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    GSAT_REQUIRE(M >= 0 && G >= 0 && C > 0 && C % 4 == 0 && M < (1ll << 31), GSAT_ERR_UNSUPPORTED, "gsat_instance_norm_bwd: C must be a positive multiple of 4");
    if (M == 0 || G == 0) return GSAT_OK;
    GSAT_REQUIRE(y && dy && stats && seg_ptr && row_seg && dx && workspace, GSAT_ERR_ARG, "gsat_instance_norm_bwd: null pointer");
    float* S1 = workspace;
    float* S2 = workspace + (size_t)G * C;
    k_in_bwd_stats<<<dim3((unsigned)G, (unsigned)ceil_div(C, 4 * SB_LANES), 1), SB, 0, stream>>>(y, dy, seg_ptr, seg_order, (int)C, S1, S2);
    k_in_bwd_apply<<<ew_blocks(M * (C / 4)), 256, 0, stream>>>(y, dy, row_seg, stats + (size_t)G * C, S1, S2, M, (int)C, dx);
    GSAT_LAUNCH_CHECK();
    return GSAT_OK;
}

int gsat_philox_noise(uint64_t seed, int64_t M, float* u, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    GSAT_REQUIRE(M >= 0 && M < (1ll << 31), GSAT_ERR_ARG, "gsat_philox_noise: bad M");
    if (M == 0) return GSAT_OK;
    GSAT_REQUIRE(u, GSAT_ERR_ARG, "gsat_philox_noise: null pointer");
    k_philox_noise<<<(unsigned)ceil_div(M, 256), 256, 0, stream>>>(seed, M, u);
    GSAT_LAUNCH_CHECK();
    return GSAT_OK;
}

int gsat_philox_keep_mask(uint64_t seed, int32_t layer, int64_t M, int64_t C, float p_drop, float* keep, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    GSAT_REQUIRE(M >= 0 && C > 0 && C % 4 == 0 && M < (1ll << 31), GSAT_ERR_UNSUPPORTED, "gsat_philox_keep_mask: C must be a positive multiple of 4");
    if (M == 0) return GSAT_OK;
    GSAT_REQUIRE(keep, GSAT_ERR_ARG, "gsat_philox_keep_mask: null pointer");
    k_philox_mask<<<ew_blocks(M * (C / 4)), 256, 0, stream>>>(seed, layer, M, (int)C, p_drop, keep);
    GSAT_LAUNCH_CHECK();
    return GSAT_OK;
}

}  // extern "C"
