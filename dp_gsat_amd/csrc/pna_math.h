// Per-channel aggregate arithmetic of PNAConvSimple (src/models/conv_layers.py:193-259) shared by the aggregation kernels (pna.hip) and
// the GEMM operand loaders that synthesise the x_i half of the aggregate (gemm.hip): one definition, identical bits.
#pragma once
#include "common.h"

namespace gsat {

constexpr int AGG_SUM = GSAT_AGG_SUM, AGG_MEAN = GSAT_AGG_MEAN, AGG_MIN = GSAT_AGG_MIN, AGG_MAX = GSAT_AGG_MAX, AGG_VAR = GSAT_AGG_VAR;

struct Acc4 {   // running per-channel statistics of one float4 column slice of the message
    float4 s, q, mn, mx;
    __device__ __forceinline__ void init() {
        s = f4zero(); q = f4zero();
        mn = make_float4(INFINITY, INFINITY, INFINITY, INFINITY);
        mx = make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
    }
    __device__ __forceinline__ void add(float4 m) {
        s.x += m.x; s.y += m.y; s.z += m.z; s.w += m.w;
        q.x = fmaf(m.x, m.x, q.x); q.y = fmaf(m.y, m.y, q.y); q.z = fmaf(m.z, m.z, q.z); q.w = fmaf(m.w, m.w, q.w);
        mn.x = fminf(mn.x, m.x); mn.y = fminf(mn.y, m.y); mn.z = fminf(mn.z, m.z); mn.w = fminf(mn.w, m.w);
        mx.x = fmaxf(mx.x, m.x); mx.y = fmaxf(mx.y, m.y); mx.z = fmaxf(mx.z, m.z); mx.w = fmaxf(mx.w, m.w);
    }
};

// inv_n = 1 / max(count, 1), computed once per row: the per-channel means multiply by it (the reference divides; the two differ by at
// most one ulp when the count is not a power of two) and the std uses the hardware square root (1 ulp) -- an IEEE division and a
// correctly rounded sqrt per channel and aggregator were ~330 of the ~450 vector instructions of a row
__device__ __forceinline__ float agg_value(int a, float s, float q, float mn, float mx, float cnt, float inv_n) {
    switch (a) {
        case AGG_SUM: return s;
        case AGG_MEAN: return s * inv_n;
        case AGG_MIN: return cnt > 0.f ? mn : 0.f;
        case AGG_MAX: return cnt > 0.f ? mx : 0.f;
        default: {
            float mean = s * inv_n, msq = q * inv_n;
            float var = msq - mean * mean;
            return a == AGG_VAR ? var : __builtin_amdgcn_sqrtf(fmaxf(var, 0.f) + 1e-5f);
        }
    }
}

__device__ __forceinline__ float4 agg_value4(int a, const Acc4& c, float cnt) {
    const float inv_n = 1.f / fmaxf(cnt, 1.f);
    return make_float4(agg_value(a, c.s.x, c.q.x, c.mn.x, c.mx.x, cnt, inv_n), agg_value(a, c.s.y, c.q.y, c.mn.y, c.mx.y, cnt, inv_n),
                       agg_value(a, c.s.z, c.q.z, c.mn.z, c.mx.z, cnt, inv_n), agg_value(a, c.s.w, c.q.w, c.mn.w, c.mx.w, cnt, inv_n));
}

__device__ __forceinline__ float4 f4scale(float a, float4 v) { return make_float4(a * v.x, a * v.y, a * v.z, a * v.w); }

// statistics of (att_k * xi) over the row from the scalar statistics of att
__device__ __forceinline__ Acc4 self_stats(float4 xi, float sa, float sa2, float amin, float amax) {
    Acc4 r;
    r.s = f4scale(sa, xi);
    r.q = make_float4(xi.x * xi.x * sa2, xi.y * xi.y * sa2, xi.z * xi.z * sa2, xi.w * xi.w * sa2);
    r.mn = make_float4(xi.x >= 0.f ? xi.x * amin : xi.x * amax, xi.y >= 0.f ? xi.y * amin : xi.y * amax,
                       xi.z >= 0.f ? xi.z * amin : xi.z * amax, xi.w >= 0.f ? xi.w * amin : xi.w * amax);
    r.mx = make_float4(xi.x >= 0.f ? xi.x * amax : xi.x * amin, xi.y >= 0.f ? xi.y * amax : xi.y * amin,
                       xi.z >= 0.f ? xi.z * amax : xi.z * amin, xi.w >= 0.f ? xi.w * amax : xi.w * amin);
    return r;
}


// The PNA aggregate [N, NAGG * 2 * H] (aggregator-major, [x_i part | x_j part] per aggregator) as a GEMM operand that is never written out:
// the x_j parts come from the compact aggregate aggj [N, NAGG * H]; the x_i parts are recomputed from x [N, H] and a coefficient pair per
// row and aggregator.  Every in-edge of row i carries att_e * x_i, so with n = max(in-degree, 1):
//   mean = x_i (sum a / n)      min = x_i >= 0 ? x_i min a : x_i max a      max = the other way      (min a, max a stored as 0 for a row
//   std  = sqrt(relu(x_i^2 (sum a^2 / n) - mean^2) + 1e-5)      sum = x_i sum a                         without in-edges: aggregate 0)
// scal[row] = 8 floats (sum a / n, sum a^2 / n, min a | 0, max a | 0, sum a, 0, 0, 0), written by the compact forward (pna_scal_store); an
// aggregator reads ONE aligned pair of it (pna_coef_offset).  Same arithmetic as self_stats() / agg_value() up to the order of two
// multiplications.
struct PnaVirt {
    const float* x;
    const float* aggj;
    const float* scal;
    int H, NAGG;
};
__device__ __forceinline__ void pna_scal_store(float* scal, int row, float sa, float sa2, float amin, float amax, float cnt) {
    const float inv_n = 1.f / fmaxf(cnt, 1.f);
    st4(scal + (size_t)row * 8, make_float4(sa * inv_n, sa2 * inv_n, cnt > 0.f ? amin : 0.f, cnt > 0.f ? amax : 0.f));
    st4(scal + (size_t)row * 8 + 4, make_float4(sa, 0.f, 0.f, 0.f));
}
// aggregator slot `a` (0 mean, 1 min, 2 max, 3 std, 4 sum: the fixed order of the reference's YAMLs)
__device__ __forceinline__ int pna_coef_offset(int a) { return (a == 1 || a == 2) ? 2 : (a == 4 ? 4 : 0); }
__device__ __forceinline__ float pna_self(int a, float xi, float2 cf) {
    switch (a) {
        case 1: return xi >= 0.f ? xi * cf.x : xi * cf.y;
        case 2: return xi >= 0.f ? xi * cf.y : xi * cf.x;
        case 3: { const float mean = xi * cf.x; return __builtin_amdgcn_sqrtf(fmaxf(xi * xi * cf.y - mean * mean, 0.f) + 1e-5f); }
        default: return xi * cf.x;          // mean (sum a / n), sum (sum a)
    }
}

}  // namespace gsat
