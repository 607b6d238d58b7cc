// BatchNorm1d over node rows (GIN `nn.1`, PNA `batch_norms.{i}` -- src/models/gin.py:58, src/models/pna.py:45,57),
// training and eval, with an optional fused ReLU (PNA: relu(batch_norm(conv(...)))).
// gsat_bn_act_* add the rest of a PNA layer tail to the same passes: y = dropout(relu(BN(x)) + residual)
// (src/models/pna.py:57-59), dropout mask from Philox (stream 3, keyed by row / column), recomputed in the backward.
// Statistics are two-pass (mean, then variance of the centred values) with two-stage, fixed-order column sums, so
// results are bitwise reproducible; running statistics follow torch (momentum update, unbiased running_var).
#include "common.h"
#include <cstdlib>

namespace gsat {

constexpr int NB = 256, NL = 16, NS = 16;     // 16 row slots x 16 lanes (float4) = 64 channels per block
constexpr int BN_DROPOUT_STREAM = 3;       // 1 and 2 are the extractor layers

struct Tail {                 // what follows act(BN(x)): + residual, then inverted dropout
    const float* residual;    // [N, C] or NULL
    float p;                  // drop probability (0 = no dropout)
    SeedRef seed;
    __device__ __forceinline__ float4 keep_scale(int64_t row, int c) const {
        if (p <= 0.f) return make_float4(1.f, 1.f, 1.f, 1.f);
        const float s = 1.f / (1.f - p);
        float4 k = philox_keep4(seed.get(), BN_DROPOUT_STREAM, (int)row, c, p);
        return make_float4(k.x * s, k.y * s, k.z * s, k.w * s);
    }
};

__device__ __forceinline__ float4 nslot_reduce(float4 v, float4 (*sm)[NL], int slot, int lane) {
    __syncthreads();
    sm[slot][lane] = v;
    __syncthreads();
    float4 r = f4zero();
    if (slot == 0) {
#pragma unroll
        for (int s = 0; s < NS; ++s) { float4 t = sm[s][lane]; r.x += t.x; r.y += t.y; r.z += t.z; r.w += t.w; }
    }
    return r;
}

// column sum of a [RB, C] partial array by one block (16 row slots x 16 lanes), fixed order; valid in slot 0
__device__ __forceinline__ float4 block_colsum(const float* __restrict__ part, int RB, int ld, int c, bool on, float4 (*sm)[NL], int slot, int lane) {
    float4 a = f4zero();
    if (on)
        for (int r0 = slot; r0 < RB; r0 += 8 * NS) {             // eight rows in flight (clamped, masked afterwards), summed in row order
            float4 t[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) t[j] = ld4(part + (size_t)min(r0 + j * NS, RB - 1) * ld + c);
#pragma unroll
            for (int j = 0; j < 8; ++j)
                if (r0 + j * NS < RB) { a.x += t[j].x; a.y += t[j].y; a.z += t[j].z; a.w += t[j].w; }
        }
    return nslot_reduce(a, sm, slot, lane);
}

// MODE 0: part[rb,c] = sum_rows x ; MODE 1: sum_rows (x - mean)^2
template <int MODE>
__global__ __launch_bounds__(NB) void k_bn_partial(const float* __restrict__ x, int64_t N, int C, int64_t rows_per_block,
                                                   const float* __restrict__ mean, float* __restrict__ part) {
    __shared__ float4 sm[NS][NL];
    const int lane = threadIdx.x % NL, slot = threadIdx.x / NL;
    const int c = (blockIdx.x * NL + lane) * 4;
    const bool on = c < C;
    const int64_t beg = (int64_t)blockIdx.y * rows_per_block, end = min(N, beg + rows_per_block);
    const float4 mu = (MODE == 1 && on) ? ld4(mean + c) : f4zero();
    float4 acc = f4zero();
    if (on)
        for (int64_t r = beg + slot; r < end; r += NS) {
            float4 v = ld4(x + (size_t)r * C + c);
            if (MODE == 0) { acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w; }
            else {
                float a = v.x - mu.x, b = v.y - mu.y, d = v.z - mu.z, e = v.w - mu.w;
                acc.x = fmaf(a, a, acc.x); acc.y = fmaf(b, b, acc.y); acc.z = fmaf(d, d, acc.z); acc.w = fmaf(e, e, acc.w);
            }
        }
    float4 t = nslot_reduce(acc, sm, slot, lane);
    if (slot == 0 && on) st4(part + (size_t)blockIdx.y * C + c, t);
}

// One-pass training statistics (round 3): a block takes its rows' mean, then the sum of squares centred on THAT mean (the second sweep re-reads
// the block's ~200 rows from L2), and leaves (mean_b, M2_b) per channel; k_bn_finalize_chan combines the blocks pairwise-exactly (Chan et al.:
// delta = mean_b - mean; mean += delta n_b / n; M2 += M2_b + delta^2 n_a n_b / n) in fixed order.  As accurate as the two-pass scheme it
// replaces (no E[x^2] - mean^2 cancellation), with one sweep over HBM and three launches (partial, finalize, apply) instead of five.
__global__ __launch_bounds__(NB) void k_bn_partial_chan(const float* __restrict__ x, int64_t N, int C, int64_t rows_per_block, float* __restrict__ part) {
    __shared__ float4 sm[NS][NL];
    __shared__ float4 smean[NL];
    const int lane = threadIdx.x % NL, slot = threadIdx.x / NL;
    const int c = (blockIdx.x * NL + lane) * 4;
    const bool on = c < C;
    const int64_t beg = (int64_t)blockIdx.y * rows_per_block, end = min(N, beg + rows_per_block);
    const float inv = 1.f / (float)max<int64_t>(end - beg, 1);
    float4 acc = f4zero();
    if (on)
        for (int64_t r = beg + slot; r < end; r += NS) { const float4 v = ld4(x + (size_t)r * C + c); acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w; }
    float4 t = nslot_reduce(acc, sm, slot, lane);
    if (slot == 0) smean[lane] = make_float4(t.x * inv, t.y * inv, t.z * inv, t.w * inv);
    __syncthreads();
    const float4 mu = smean[lane];
    acc = f4zero();
    if (on)
        for (int64_t r = beg + slot; r < end; r += NS) {
            const float4 v = ld4(x + (size_t)r * C + c);
            const float a = v.x - mu.x, b = v.y - mu.y, d = v.z - mu.z, e = v.w - mu.w;
            acc.x = fmaf(a, a, acc.x); acc.y = fmaf(b, b, acc.y); acc.z = fmaf(d, d, acc.z); acc.w = fmaf(e, e, acc.w);
        }
    t = nslot_reduce(acc, sm, slot, lane);
    if (slot == 0 && on) {
        st4(part + (size_t)blockIdx.y * 2 * C + c, mu);
        st4(part + (size_t)blockIdx.y * 2 * C + C + c, t);
    }
}

struct Chan4 { float4 mean, m2; float n; };
__device__ __forceinline__ void chan_add(Chan4& a, float4 mb, float4 m2b, float nb) {
    if (nb <= 0.f) return;
    const float tot = a.n + nb, w = nb / tot, u = a.n * w;          // u = n_a n_b / n
    const float dx = mb.x - a.mean.x, dy = mb.y - a.mean.y, dz = mb.z - a.mean.z, dw = mb.w - a.mean.w;
    a.mean = make_float4(fmaf(dx, w, a.mean.x), fmaf(dy, w, a.mean.y), fmaf(dz, w, a.mean.z), fmaf(dw, w, a.mean.w));
    a.m2 = make_float4(a.m2.x + m2b.x + dx * dx * u, a.m2.y + m2b.y + dy * dy * u, a.m2.z + m2b.z + dz * dz * u, a.m2.w + m2b.w + dw * dw * u);
    a.n = tot;
}
__global__ __launch_bounds__(NB) void k_bn_finalize_chan(const float* __restrict__ part, int RB, int64_t N, int64_t rows_per_block, int C, float eps,
                                                         float momentum, float* __restrict__ mean, float* __restrict__ rstd,
                                                         float* __restrict__ running_mean, float* __restrict__ running_var) {
    __shared__ float4 smu[NS][NL], sm2[NS][NL];
    __shared__ float sn[NS];
    const int lane = threadIdx.x % NL, slot = threadIdx.x / NL;
    const int c = (blockIdx.x * NL + lane) * 4;
    const bool on = c < C;
    Chan4 a{f4zero(), f4zero(), 0.f};
    for (int r0 = slot; r0 < RB; r0 += 8 * NS) {                 // blocks slot, slot + 16, ... in order; eight records in flight (clamped loads)
        float4 mb[8], qb[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int r = min(r0 + j * NS, RB - 1);
            mb[j] = on ? ld4(part + (size_t)r * 2 * C + c) : f4zero();
            qb[j] = on ? ld4(part + (size_t)r * 2 * C + C + c) : f4zero();
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int r = r0 + j * NS;
            if (r < RB) chan_add(a, mb[j], qb[j], (float)(min(N, (int64_t)(r + 1) * rows_per_block) - (int64_t)r * rows_per_block));
        }
    }
    smu[slot][lane] = a.mean; sm2[slot][lane] = a.m2;
    if (lane == 0) sn[slot] = a.n;
    __syncthreads();
    if (slot != 0 || !on) return;
    for (int s = 1; s < NS; ++s) chan_add(a, smu[s][lane], sm2[s][lane], sn[s]);          // then the sixteen slot results, in slot order
    const float inv = 1.f / (float)N;
    st4(mean + c, a.mean);
    st4(rstd + c, make_float4(1.f / sqrtf(a.m2.x * inv + eps), 1.f / sqrtf(a.m2.y * inv + eps), 1.f / sqrtf(a.m2.z * inv + eps), 1.f / sqrtf(a.m2.w * inv + eps)));
    if (running_mean) {
        const float ub = N > 1 ? 1.f / (float)(N - 1) : inv;
        const float4 rm = ld4(running_mean + c), rv = ld4(running_var + c);
        st4(running_mean + c, make_float4((1.f - momentum) * rm.x + momentum * a.mean.x, (1.f - momentum) * rm.y + momentum * a.mean.y,
                                          (1.f - momentum) * rm.z + momentum * a.mean.z, (1.f - momentum) * rm.w + momentum * a.mean.w));
        st4(running_var + c, make_float4((1.f - momentum) * rv.x + momentum * a.m2.x * ub, (1.f - momentum) * rv.y + momentum * a.m2.y * ub,
                                         (1.f - momentum) * rv.z + momentum * a.m2.z * ub, (1.f - momentum) * rv.w + momentum * a.m2.w * ub));
    }
}

// STAGE 0: mean = colsum(part)/N ; STAGE 1: rstd = 1/sqrt(colsum(part)/N + eps) and the running-statistics update
template <int STAGE>
__global__ __launch_bounds__(NB) void k_bn_finalize(const float* __restrict__ part, int RB, int64_t N, int C, float eps, float momentum,
                                                    float* __restrict__ mean, float* __restrict__ rstd, float* __restrict__ running_mean,
                                                    float* __restrict__ running_var) {
    __shared__ float4 sm[NS][NL];
    const int lane = threadIdx.x % NL, slot = threadIdx.x / NL;
    const int c = (blockIdx.x * NL + lane) * 4;
    const bool on = c < C;
    float4 t = block_colsum(part, RB, C, c, on, sm, slot, lane);
    if (slot != 0 || !on) return;
    const float inv = 1.f / (float)N;
    if (STAGE == 0) {
        st4(mean + c, make_float4(t.x * inv, t.y * inv, t.z * inv, t.w * inv));
    } else {
        st4(rstd + c, make_float4(1.f / sqrtf(t.x * inv + eps), 1.f / sqrtf(t.y * inv + eps), 1.f / sqrtf(t.z * inv + eps), 1.f / sqrtf(t.w * inv + eps)));
        if (running_mean) {
            const float ub = N > 1 ? 1.f / (float)(N - 1) : inv;
            float4 m = ld4(mean + c), rm = ld4(running_mean + c), rv = ld4(running_var + c);
            st4(running_mean + c, make_float4((1.f - momentum) * rm.x + momentum * m.x, (1.f - momentum) * rm.y + momentum * m.y,
                                              (1.f - momentum) * rm.z + momentum * m.z, (1.f - momentum) * rm.w + momentum * m.w));
            st4(running_var + c, make_float4((1.f - momentum) * rv.x + momentum * t.x * ub, (1.f - momentum) * rv.y + momentum * t.y * ub,
                                             (1.f - momentum) * rv.z + momentum * t.z * ub, (1.f - momentum) * rv.w + momentum * t.w * ub));
        }
    }
}

// out[c] = colsum(part): the local (per-rank) sum, handed to the caller for the cross-rank reduction
__global__ __launch_bounds__(NB) void k_bn_sum_finalize(const float* __restrict__ part, int RB, int C, float* __restrict__ out) {
    __shared__ float4 sm[NS][NL];
    const int lane = threadIdx.x % NL, slot = threadIdx.x / NL;
    const int c = (blockIdx.x * NL + lane) * 4;
    const bool on = c < C;
    float4 t = block_colsum(part, RB, C, c, on, sm, slot, lane);
    if (slot == 0 && on) st4(out + c, t);
}

__global__ void k_bn_eval_stats(const float* __restrict__ running_mean, const float* __restrict__ running_var, int C, float eps,
                                float* __restrict__ mean, float* __restrict__ rstd) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c < C) { mean[c] = running_mean[c]; rstd[c] = 1.f / sqrtf(running_var[c] + eps); }
}

__global__ void k_bn_apply(const float* __restrict__ x, const float* __restrict__ mean, const float* __restrict__ rstd,
                           const float* __restrict__ gamma, const float* __restrict__ beta, int64_t N, int C, int relu, Tail tail, float* __restrict__ y) {
    const int C4 = C >> 2;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < N * C4; i += (int64_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % C4) * 4;
        float4 v = ld4(x + i * 4), mu = ld4(mean + c), rs = ld4(rstd + c), g = ld4(gamma + c), b = ld4(beta + c);
        float4 o = make_float4(fmaf((v.x - mu.x) * rs.x, g.x, b.x), fmaf((v.y - mu.y) * rs.y, g.y, b.y),
                               fmaf((v.z - mu.z) * rs.z, g.z, b.z), fmaf((v.w - mu.w) * rs.w, g.w, b.w));
        if (relu) o = make_float4(fmaxf(o.x, 0.f), fmaxf(o.y, 0.f), fmaxf(o.z, 0.f), fmaxf(o.w, 0.f));
        if (tail.residual) { const float4 r = ld4(tail.residual + i * 4); o = make_float4(o.x + r.x, o.y + r.y, o.z + r.z, o.w + r.w); }
        if (tail.p > 0.f) { const float4 k = tail.keep_scale(i / C4, c); o = make_float4(o.x * k.x, o.y * k.y, o.z * k.z, o.w * k.w); }
        st4(y + i * 4, o);
    }
}

// backward partial sums: part[rb, 0:C] = sum dy' , part[rb, C:2C] = sum dy' * xhat ; dy' = dy * [y > 0] when relu
__global__ __launch_bounds__(NB) void k_bn_bwd_partial(const float* __restrict__ x, const float* __restrict__ dy, const float* __restrict__ mean,
                                                       const float* __restrict__ rstd, const float* __restrict__ gamma,
                                                       const float* __restrict__ beta, int64_t N, int C, int relu, int64_t rows_per_block,
                                                       Tail tail, float* __restrict__ part) {
    __shared__ float4 sm[NS][NL];
    const int lane = threadIdx.x % NL, slot = threadIdx.x / NL;
    const int c = (blockIdx.x * NL + lane) * 4;
    const bool on = c < C;
    const int64_t beg = (int64_t)blockIdx.y * rows_per_block, end = min(N, beg + rows_per_block);
    float4 a1 = f4zero(), a2 = f4zero();
    if (on) {
        const float4 mu = ld4(mean + c), rs = ld4(rstd + c), g = ld4(gamma + c), b = ld4(beta + c);
        for (int64_t r = beg + slot; r < end; r += NS) {
            float4 v = ld4(x + (size_t)r * C + c), d = ld4(dy + (size_t)r * C + c);
            if (tail.p > 0.f) { const float4 k = tail.keep_scale(r, c); d = make_float4(d.x * k.x, d.y * k.y, d.z * k.z, d.w * k.w); }
            float4 xh = make_float4((v.x - mu.x) * rs.x, (v.y - mu.y) * rs.y, (v.z - mu.z) * rs.z, (v.w - mu.w) * rs.w);
            if (relu) {
                d.x = fmaf(xh.x, g.x, b.x) > 0.f ? d.x : 0.f; d.y = fmaf(xh.y, g.y, b.y) > 0.f ? d.y : 0.f;
                d.z = fmaf(xh.z, g.z, b.z) > 0.f ? d.z : 0.f; d.w = fmaf(xh.w, g.w, b.w) > 0.f ? d.w : 0.f;
            }
            a1.x += d.x; a1.y += d.y; a1.z += d.z; a1.w += d.w;
            a2.x = fmaf(d.x, xh.x, a2.x); a2.y = fmaf(d.y, xh.y, a2.y); a2.z = fmaf(d.z, xh.z, a2.z); a2.w = fmaf(d.w, xh.w, a2.w);
        }
    }
    float4 t1 = nslot_reduce(a1, sm, slot, lane);
    float4 t2 = nslot_reduce(a2, sm, slot, lane);
    if (slot == 0 && on) {
        st4(part + (size_t)blockIdx.y * 2 * C + c, t1);
        st4(part + (size_t)blockIdx.y * 2 * C + C + c, t2);
    }
}

__global__ __launch_bounds__(NB) void k_bn_bwd_finalize(const float* __restrict__ part, int RB, int C, float* __restrict__ dbeta,
                                                        float* __restrict__ dgamma) {
    __shared__ float4 sm[NS][NL];
    const int lane = threadIdx.x % NL, slot = threadIdx.x / NL;
    const int c = (blockIdx.x * NL + lane) * 4;
    const bool on = c < C;
    float4 t1 = block_colsum(part, RB, 2 * C, c, on, sm, slot, lane);
    float4 t2 = block_colsum(part + C, RB, 2 * C, c, on, sm, slot, lane);
    if (slot == 0 && on) { st4(dbeta + c, t1); st4(dgamma + c, t2); }
}

// training: dx = gamma*rstd*(dy' - dbeta/N - xhat*dgamma/N) ; eval: dx = gamma*rstd*dy'
__global__ void k_bn_bwd_apply(const float* __restrict__ x, const float* __restrict__ dy, const float* __restrict__ mean,
                               const float* __restrict__ rstd, const float* __restrict__ gamma, const float* __restrict__ beta,
                               const float* __restrict__ dbeta, const float* __restrict__ dgamma, int64_t N, int C, int relu, int training,
                               Tail tail, float* __restrict__ dx, float* __restrict__ dres, float inv_n, const float* __restrict__ rows_dev) {
    const int C4 = C >> 2;
    if (rows_dev) inv_n = 1.f / *rows_dev;            // global row count of a sharded batch, still on the device
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < N * C4; i += (int64_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % C4) * 4;
        float4 v = ld4(x + i * 4), d = ld4(dy + i * 4), mu = ld4(mean + c), rs = ld4(rstd + c), g = ld4(gamma + c), b = ld4(beta + c);
        if (tail.p > 0.f) { const float4 k = tail.keep_scale(i / C4, c); d = make_float4(d.x * k.x, d.y * k.y, d.z * k.z, d.w * k.w); }
        if (dres) st4(dres + i * 4, d);                  // gradient of the residual branch: d(out)/d(residual) = dropout only
        float4 xh = make_float4((v.x - mu.x) * rs.x, (v.y - mu.y) * rs.y, (v.z - mu.z) * rs.z, (v.w - mu.w) * rs.w);
        if (relu) {
            d.x = fmaf(xh.x, g.x, b.x) > 0.f ? d.x : 0.f; d.y = fmaf(xh.y, g.y, b.y) > 0.f ? d.y : 0.f;
            d.z = fmaf(xh.z, g.z, b.z) > 0.f ? d.z : 0.f; d.w = fmaf(xh.w, g.w, b.w) > 0.f ? d.w : 0.f;
        }
        float4 o;
        if (training) {
            float4 db = ld4(dbeta + c), dg = ld4(dgamma + c);
            o = make_float4(g.x * rs.x * (d.x - db.x * inv_n - xh.x * dg.x * inv_n), g.y * rs.y * (d.y - db.y * inv_n - xh.y * dg.y * inv_n),
                            g.z * rs.z * (d.z - db.z * inv_n - xh.z * dg.z * inv_n), g.w * rs.w * (d.w - db.w * inv_n - xh.w * dg.w * inv_n));
        } else {
            o = make_float4(g.x * rs.x * d.x, g.y * rs.y * d.y, g.z * rs.z * d.z, g.w * rs.w * d.w);
        }
        st4(dx + i * 4, o);
    }
}

static inline void row_blocks(int64_t N, int64_t* RB, int64_t* rpb) {
    int64_t rb = std::min<int64_t>(256, std::max<int64_t>(1, ceil_div(N, 64)));
    *rpb = ceil_div(N, rb);
    *RB = ceil_div(N, *rpb);
}
static inline int ew_grid(int64_t items) { return (int)std::max<int64_t>(1, std::min<int64_t>(ceil_div(items, 256), 256 * 32)); }

}  // namespace gsat

using namespace gsat;

extern "C" {

size_t gsat_bn_workspace_floats(int64_t N, int64_t C) {
    int64_t RB, rpb;
    row_blocks(N > 0 ? N : 1, &RB, &rpb);
    return (size_t)RB * (size_t)C * 2;
}

int gsat_bn_act_fwd(const float* x, const float* gamma, const float* beta, float* running_mean, float* running_var, int64_t N, int64_t C,
                    int training, float momentum, float eps, int relu, const float* residual, float dropout_p, uint64_t seed,
                    const uint64_t* seed_dev, float* y, float* save_mean, float* save_rstd, float* workspace, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    GSAT_REQUIRE(N >= 0 && C > 0 && C % 4 == 0 && N < (1ll << 31), GSAT_ERR_UNSUPPORTED, "gsat_bn_fwd: C must be a positive multiple of 4");
    GSAT_REQUIRE(dropout_p >= 0.f && dropout_p < 1.f, GSAT_ERR_ARG, "gsat_bn_act_fwd: dropout_p must be in [0, 1)");
    if (N == 0) return GSAT_OK;
    GSAT_REQUIRE(x && gamma && beta && y && save_mean && save_rstd, GSAT_ERR_ARG, "gsat_bn_fwd: null pointer");
    if (training) {
        GSAT_REQUIRE(workspace, GSAT_ERR_WORKSPACE, "gsat_bn_fwd: workspace required in training mode");
        int64_t RB, rpb;
        row_blocks(N, &RB, &rpb);
        float* part0 = workspace;
        float* part1 = workspace + (size_t)RB * C;
        const dim3 grid((unsigned)ceil_div(C, 64), (unsigned)RB);
        const unsigned ct = (unsigned)ceil_div(C, 64);
        static const int one_pass = getenv("GSAT_BN_ONE_PASS") ? atoi(getenv("GSAT_BN_ONE_PASS")) : 1;
        if (one_pass) {
            k_bn_partial_chan<<<grid, NB, 0, stream>>>(x, N, (int)C, rpb, workspace);
            k_bn_finalize_chan<<<ct, NB, 0, stream>>>(workspace, (int)RB, N, rpb, (int)C, eps, momentum, save_mean, save_rstd, running_mean, running_var);
        } else {
        k_bn_partial<0><<<grid, NB, 0, stream>>>(x, N, (int)C, rpb, nullptr, part0);
        k_bn_finalize<0><<<ct, NB, 0, stream>>>(part0, (int)RB, N, (int)C, eps, momentum, save_mean, save_rstd, running_mean, running_var);
        k_bn_partial<1><<<grid, NB, 0, stream>>>(x, N, (int)C, rpb, save_mean, part1);
        k_bn_finalize<1><<<ct, NB, 0, stream>>>(part1, (int)RB, N, (int)C, eps, momentum, save_mean, save_rstd, running_mean, running_var);
        }
    } else {
        GSAT_REQUIRE(running_mean && running_var, GSAT_ERR_ARG, "gsat_bn_fwd: eval mode needs running statistics");
        k_bn_eval_stats<<<(unsigned)ceil_div(C, 256), 256, 0, stream>>>(running_mean, running_var, (int)C, eps, save_mean, save_rstd);
    }
    const Tail tail{residual, dropout_p, SeedRef{seed, seed_dev}};
    k_bn_apply<<<ew_grid(N * (C / 4)), 256, 0, stream>>>(x, save_mean, save_rstd, gamma, beta, N, (int)C, relu, tail, y);
    GSAT_LAUNCH_CHECK();
    return GSAT_OK;
}

int gsat_bn_fwd(const float* x, const float* gamma, const float* beta, float* running_mean, float* running_var, int64_t N, int64_t C,
                int training, float momentum, float eps, int relu, float* y, float* save_mean, float* save_rstd, float* workspace,
                void* stream_) {
    return gsat_bn_act_fwd(x, gamma, beta, running_mean, running_var, N, C, training, momentum, eps, relu, nullptr, 0.f, 0, nullptr, y,
                           save_mean, save_rstd, workspace, stream_);
}

int gsat_bn_act_bwd(const float* x, const float* dy, const float* gamma, const float* beta, const float* save_mean, const float* save_rstd,
                    int64_t N, int64_t C, int training, int relu, float dropout_p, uint64_t seed, const uint64_t* seed_dev, float* dx,
                    float* dresidual, float* dgamma, float* dbeta, float* workspace, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    GSAT_REQUIRE(N >= 0 && C > 0 && C % 4 == 0 && N < (1ll << 31), GSAT_ERR_UNSUPPORTED, "gsat_bn_bwd: C must be a positive multiple of 4");
    GSAT_REQUIRE(dgamma && dbeta, GSAT_ERR_ARG, "gsat_bn_bwd: null gradient output");
    GSAT_REQUIRE(dropout_p >= 0.f && dropout_p < 1.f, GSAT_ERR_ARG, "gsat_bn_act_bwd: dropout_p must be in [0, 1)");
    if (N == 0) {
        GSAT_CHECK_HIP(gsat::zero_async(dgamma, sizeof(float) * C, stream));
        GSAT_CHECK_HIP(gsat::zero_async(dbeta, sizeof(float) * C, stream));
        return GSAT_OK;
    }
    GSAT_REQUIRE(x && dy && gamma && beta && save_mean && save_rstd && dx && workspace, GSAT_ERR_ARG, "gsat_bn_bwd: null pointer");
    int64_t RB, rpb;
    row_blocks(N, &RB, &rpb);
    const Tail tail{nullptr, dropout_p, SeedRef{seed, seed_dev}};
    k_bn_bwd_partial<<<dim3((unsigned)ceil_div(C, 64), (unsigned)RB), NB, 0, stream>>>(x, dy, save_mean, save_rstd, gamma, beta, N, (int)C, relu, rpb, tail, workspace);
    k_bn_bwd_finalize<<<(unsigned)ceil_div(C, 64), NB, 0, stream>>>(workspace, (int)RB, (int)C, dbeta, dgamma);
    k_bn_bwd_apply<<<ew_grid(N * (C / 4)), 256, 0, stream>>>(x, dy, save_mean, save_rstd, gamma, beta, dbeta, dgamma, N, (int)C, relu, training, tail, dx, dresidual,
                                                              1.f / (float)N, nullptr);
    GSAT_LAUNCH_CHECK();
    return GSAT_OK;
}

int gsat_bn_bwd(const float* x, const float* dy, const float* gamma, const float* beta, const float* save_mean, const float* save_rstd,
                int64_t N, int64_t C, int training, int relu, float* dx, float* dgamma, float* dbeta, float* workspace, void* stream_) {
    return gsat_bn_act_bwd(x, dy, gamma, beta, save_mean, save_rstd, N, C, training, relu, 0.f, 0, nullptr, dx, nullptr, dgamma, dbeta,
                           workspace, stream_);
}

/*
 * Data-parallel BatchNorm in three local steps with the cross-rank reductions left to the caller (one tiny all-reduce each):
 *   gsat_bn_local_sum:   out[c] = sum_rows x[r,c]  (centre == NULL)  or  sum_rows (x[r,c] - centre[c])^2
 *   gsat_bn_apply_fwd:   y = dropout_p( [relu](gamma (x - mean) rstd + beta) + residual ) with GIVEN (global) mean / rstd
 *   gsat_bn_local_bwd_sums / gsat_bn_apply_bwd: local (sum dy', sum dy' xhat), then dx with the GLOBAL sums and row count
 */
int gsat_bn_local_sum(const float* x, const float* centre, int64_t N, int64_t C, float* out, float* workspace, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    GSAT_REQUIRE(N >= 0 && C > 0 && C % 4 == 0 && N < (1ll << 31) && out, GSAT_ERR_ARG, "gsat_bn_local_sum: bad argument");
    if (N == 0) { GSAT_CHECK_HIP(gsat::zero_async(out, sizeof(float) * C, stream)); return GSAT_OK; }
    GSAT_REQUIRE(x && workspace, GSAT_ERR_ARG, "gsat_bn_local_sum: null pointer");
    int64_t RB, rpb;
    row_blocks(N, &RB, &rpb);
    const dim3 grid((unsigned)ceil_div(C, 64), (unsigned)RB);
    if (centre) k_bn_partial<1><<<grid, NB, 0, stream>>>(x, N, (int)C, rpb, centre, workspace);
    else k_bn_partial<0><<<grid, NB, 0, stream>>>(x, N, (int)C, rpb, nullptr, workspace);
    k_bn_sum_finalize<<<(unsigned)ceil_div(C, 64), NB, 0, stream>>>(workspace, (int)RB, (int)C, out);
    GSAT_LAUNCH_CHECK();
    return GSAT_OK;
}

int gsat_bn_apply_fwd(const float* x, const float* gamma, const float* beta, const float* mean, const float* rstd, int64_t N, int64_t C,
                      int relu, const float* residual, float dropout_p, uint64_t seed, const uint64_t* seed_dev, float* y, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    GSAT_REQUIRE(N >= 0 && C > 0 && C % 4 == 0 && N < (1ll << 31), GSAT_ERR_ARG, "gsat_bn_apply_fwd: bad extents");
    GSAT_REQUIRE(dropout_p >= 0.f && dropout_p < 1.f, GSAT_ERR_ARG, "gsat_bn_apply_fwd: dropout_p must be in [0, 1)");
    if (N == 0) return GSAT_OK;
    GSAT_REQUIRE(x && gamma && beta && mean && rstd && y, GSAT_ERR_ARG, "gsat_bn_apply_fwd: null pointer");
    const Tail tail{residual, dropout_p, SeedRef{seed, seed_dev}};
    k_bn_apply<<<ew_grid(N * (C / 4)), 256, 0, stream>>>(x, mean, rstd, gamma, beta, N, (int)C, relu, tail, y);
    GSAT_LAUNCH_CHECK();
    return GSAT_OK;
}

// ---- GIN layer tail: y = dropout_p(relu(x)) (src/models/gin.py:49-52), one launch forward, one backward (from y alone) ----------------
constexpr int RELU_DROPOUT_STREAM = 5;
__global__ void k_relu_dropout_fwd(const float* __restrict__ x, int64_t N, int C, float p, SeedRef seed, float* __restrict__ y) {
    const int C4 = C >> 2;
    const float s = p > 0.f ? 1.f / (1.f - p) : 1.f;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < N * C4; i += (int64_t)gridDim.x * blockDim.x) {
        const float4 v = ld4(x + i * 4);
        float4 o = make_float4(fmaxf(v.x, 0.f), fmaxf(v.y, 0.f), fmaxf(v.z, 0.f), fmaxf(v.w, 0.f));
        if (p > 0.f) {
            const float4 k = philox_keep4(seed.get(), RELU_DROPOUT_STREAM, (int)(i / C4), (int)(i % C4) * 4, p);
            o = make_float4(o.x * k.x * s, o.y * k.y * s, o.z * k.z * s, o.w * k.w * s);
        }
        st4(y + i * 4, o);
    }
}
// y > 0 <=> x > 0 and kept, so dx = dy / (1 - p) there and 0 elsewhere: no mask recomputation
__global__ void k_relu_dropout_bwd(const float* __restrict__ y, const float* __restrict__ dy, int64_t n4, float s, float* __restrict__ dx) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
        const float4 v = ld4(y + i * 4), d = ld4(dy + i * 4);
        st4(dx + i * 4, make_float4(v.x > 0.f ? d.x * s : 0.f, v.y > 0.f ? d.y * s : 0.f, v.z > 0.f ? d.z * s : 0.f, v.w > 0.f ? d.w * s : 0.f));
    }
}

int gsat_relu_dropout_fwd(const float* x, int64_t N, int64_t C, float dropout_p, uint64_t seed, const uint64_t* seed_dev, float* y, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    GSAT_REQUIRE(N >= 0 && C > 0 && C % 4 == 0 && N < (1ll << 31), GSAT_ERR_ARG, "gsat_relu_dropout_fwd: bad extents (C must be a multiple of 4)");
    GSAT_REQUIRE(dropout_p >= 0.f && dropout_p < 1.f, GSAT_ERR_ARG, "gsat_relu_dropout_fwd: dropout_p must be in [0, 1)");
    if (N == 0) return GSAT_OK;
    GSAT_REQUIRE(x && y, GSAT_ERR_ARG, "gsat_relu_dropout_fwd: null pointer");
    k_relu_dropout_fwd<<<ew_grid(N * (C / 4)), 256, 0, stream>>>(x, N, (int)C, dropout_p, SeedRef{seed, seed_dev}, y);
    GSAT_LAUNCH_CHECK();
    return GSAT_OK;
}

int gsat_relu_dropout_bwd(const float* y, const float* dy, int64_t N, int64_t C, float dropout_p, float* dx, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    GSAT_REQUIRE(N >= 0 && C > 0 && C % 4 == 0 && N < (1ll << 31), GSAT_ERR_ARG, "gsat_relu_dropout_bwd: bad extents (C must be a multiple of 4)");
    GSAT_REQUIRE(dropout_p >= 0.f && dropout_p < 1.f, GSAT_ERR_ARG, "gsat_relu_dropout_bwd: dropout_p must be in [0, 1)");
    if (N == 0) return GSAT_OK;
    GSAT_REQUIRE(y && dy && dx, GSAT_ERR_ARG, "gsat_relu_dropout_bwd: null pointer");
    k_relu_dropout_bwd<<<ew_grid(N * (C / 4)), 256, 0, stream>>>(y, dy, N * (C / 4), dropout_p > 0.f ? 1.f / (1.f - dropout_p) : 1.f, dx);
    GSAT_LAUNCH_CHECK();
    return GSAT_OK;
}

int gsat_bn_local_bwd_sums(const float* x, const float* dy, const float* gamma, const float* beta, const float* mean, const float* rstd,
                           int64_t N, int64_t C, int relu, float dropout_p, uint64_t seed, const uint64_t* seed_dev, float* sum_dy,
                           float* sum_dy_xhat, float* workspace, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    GSAT_REQUIRE(N >= 0 && C > 0 && C % 4 == 0 && N < (1ll << 31) && sum_dy && sum_dy_xhat, GSAT_ERR_ARG, "gsat_bn_local_bwd_sums: bad argument");
    if (N == 0) {
        GSAT_CHECK_HIP(gsat::zero_async(sum_dy, sizeof(float) * C, stream));
        GSAT_CHECK_HIP(gsat::zero_async(sum_dy_xhat, sizeof(float) * C, stream));
        return GSAT_OK;
    }
    GSAT_REQUIRE(x && dy && gamma && beta && mean && rstd && workspace, GSAT_ERR_ARG, "gsat_bn_local_bwd_sums: null pointer");
    int64_t RB, rpb;
    row_blocks(N, &RB, &rpb);
    const Tail tail{nullptr, dropout_p, SeedRef{seed, seed_dev}};
    k_bn_bwd_partial<<<dim3((unsigned)ceil_div(C, 64), (unsigned)RB), NB, 0, stream>>>(x, dy, mean, rstd, gamma, beta, N, (int)C, relu, rpb, tail, workspace);
    k_bn_bwd_finalize<<<(unsigned)ceil_div(C, 64), NB, 0, stream>>>(workspace, (int)RB, (int)C, sum_dy, sum_dy_xhat);
    GSAT_LAUNCH_CHECK();
    return GSAT_OK;
}

int gsat_bn_apply_bwd(const float* x, const float* dy, const float* gamma, const float* beta, const float* mean, const float* rstd,
                      const float* sum_dy, const float* sum_dy_xhat, int64_t global_rows, const float* global_rows_dev, int64_t N, int64_t C,
                      int relu, float dropout_p, uint64_t seed, const uint64_t* seed_dev, float* dx, float* dresidual, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    GSAT_REQUIRE(N >= 0 && C > 0 && C % 4 == 0 && N < (1ll << 31) && (global_rows_dev || global_rows >= N), GSAT_ERR_ARG, "gsat_bn_apply_bwd: bad extents");
    if (N == 0) return GSAT_OK;
    GSAT_REQUIRE(x && dy && gamma && beta && mean && rstd && sum_dy && sum_dy_xhat && dx, GSAT_ERR_ARG, "gsat_bn_apply_bwd: null pointer");
    const Tail tail{nullptr, dropout_p, SeedRef{seed, seed_dev}};
    k_bn_bwd_apply<<<ew_grid(N * (C / 4)), 256, 0, stream>>>(x, dy, mean, rstd, gamma, beta, sum_dy, sum_dy_xhat, N, (int)C, relu, 1, tail, dx, dresidual,
                                                              global_rows > 0 ? 1.f / (float)global_rows : 0.f, global_rows_dev);
    GSAT_LAUNCH_CHECK();
    return GSAT_OK;
}

}  // extern "C"
