// PNAConvSimple message passing: one pass over the in-edges of a destination row produces every
// requested aggregator (sum / mean / min / max / var / std) and scaler of the message
//   m_e = att_e * [ x_i || x_j (|| edge_emb_e) ]            (src/models/conv_layers.py:166-185)
// without materialising m ([E, 2H] or [E, 3H]) and without one scatter pass per aggregator.
// The x_i third of the message is the same vector for every in-edge of row i, so its aggregates
// follow from four scalar statistics of att over the row (sum, sum of squares, min, max).
//
// Backward is two atomics-free passes: (1) per destination row, recompute the row statistics,
// route gradients (first-occurrence arg for min/max, as torch-scatter's CPU kernel) and store the
// per-edge gradient row in CSR slot order; (2) per source row, sum those rows through the
// inverted index (gsat_aggr_sum_fwd with an identity weight).  Both are bitwise reproducible.
#include "common.h"

namespace gsat {

constexpr int PNA_BLOCK = 256;
constexpr int AGG_SUM = 0, AGG_MEAN = 1, AGG_MIN = 2, AGG_MAX = 3, AGG_VAR = 4, AGG_STD = 5;
constexpr int SC_ID = 0, SC_AMP = 1, SC_ATT = 2, SC_LIN = 3, SC_INVLIN = 4;

struct PnaCfg {
    int A, S;
    int aggr[8];
    int scal[8];
    float avg_lin, avg_log;
};

__device__ __forceinline__ int pna_xcd_remap(int b, int nb) {
    int q = nb >> 3, r = nb & 7, x = b & 7, i = b >> 3;
    return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + i;
}

__device__ __forceinline__ float scaler_factor(int s, float deg, float avg_lin, float avg_log) {
    switch (s) {
        case SC_AMP: return logf(deg + 1.f) / avg_log;
        case SC_ATT: return deg == 0.f ? 1.f : avg_log / logf(deg + 1.f);
        case SC_LIN: return deg / avg_lin;
        case SC_INVLIN: return deg == 0.f ? 1.f : avg_lin / deg;
        default: return 1.f;
    }
}

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

__device__ __forceinline__ float agg_value(int a, float s, float q, float mn, float mx, float cnt) {
    const float n = fmaxf(cnt, 1.f);
    switch (a) {
        case AGG_SUM: return s;
        case AGG_MEAN: return s / n;
        case AGG_MIN: return cnt > 0.f ? mn : 0.f;
        case AGG_MAX: return cnt > 0.f ? mx : 0.f;
        default: {
            float mean = s / n, msq = q / n;
            float var = msq - mean * mean;
            return a == AGG_VAR ? var : sqrtf(fmaxf(var, 0.f) + 1e-5f);
        }
    }
}

__device__ __forceinline__ float4 agg_value4(int a, const Acc4& c, float cnt) {
    return make_float4(agg_value(a, c.s.x, c.q.x, c.mn.x, c.mx.x, cnt), agg_value(a, c.s.y, c.q.y, c.mn.y, c.mx.y, cnt),
                       agg_value(a, c.s.z, c.q.z, c.mn.z, c.mx.z, cnt), agg_value(a, c.s.w, c.q.w, c.mn.w, c.mx.w, cnt));
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

template <int LPR, bool HAS_EE>
__global__ __launch_bounds__(PNA_BLOCK) void k_pna_fwd(
    const float* __restrict__ x, const float* __restrict__ att, const float* __restrict__ edge_emb,
    const int32_t* __restrict__ rowptr, const int32_t* __restrict__ col, const int32_t* __restrict__ eid,
    int num_rows, int H, PnaCfg cfg, float* __restrict__ out, int rows_per_group) {
    constexpr int GPB = PNA_BLOCK / LPR;
    const int lane = threadIdx.x % LPR;
    const int c = lane * 4;
    const bool on = c < H;
    const int parts = HAS_EE ? 3 : 2;
    const int F = parts * H;
    const size_t out_stride = (size_t)cfg.S * cfg.A * F;
    const int grp = pna_xcd_remap(blockIdx.x, gridDim.x) * GPB + threadIdx.x / LPR;
    int row = grp * rows_per_group;
    const int row_end = min(num_rows, row + rows_per_group);
    for (; row < row_end; ++row) {
        const int beg = rowptr[row], end = rowptr[row + 1];
        const float cnt = (float)(end - beg);
        float sa = 0.f, sa2 = 0.f, amin = INFINITY, amax = -INFINITY;
        Acc4 aj, ae;
        aj.init(); ae.init();
        const float4 xi = on ? ld4(x + (size_t)row * H + c) : f4zero();
        for (int k = beg; k < end; ++k) {
            const int j = col[k];
            const int e = (att != nullptr || HAS_EE) ? eid[k] : 0;
            const float w = att ? att[e] : 1.f;
            sa += w; sa2 = fmaf(w, w, sa2); amin = fminf(amin, w); amax = fmaxf(amax, w);
            if (on) {
                aj.add(f4scale(w, ld4(x + (size_t)j * H + c)));
                if (HAS_EE) ae.add(f4scale(w, ld4(edge_emb + (size_t)e * H + c)));
            }
        }
        if (!on) continue;
        const Acc4 ai = self_stats(xi, sa, sa2, amin, amax);
        float* orow = out + (size_t)row * out_stride;
        for (int s = 0; s < cfg.S; ++s) {
            const float f = scaler_factor(cfg.scal[s], cnt, cfg.avg_lin, cfg.avg_log);
            for (int a = 0; a < cfg.A; ++a) {
                float* o = orow + (size_t)(s * cfg.A + a) * F + c;
                st4(o, f4scale(f, agg_value4(cfg.aggr[a], ai, cnt)));
                st4(o + H, f4scale(f, agg_value4(cfg.aggr[a], aj, cnt)));
                if (HAS_EE) st4(o + 2 * H, f4scale(f, agg_value4(cfg.aggr[a], ae, cnt)));
            }
        }
    }
}

// per-channel gradient routing coefficients:  d m_k = P + Q*m_k + gmin*[k==argmin] + gmax*[k==argmax]
struct Coef4 { float4 P, Q, gmin, gmax; };

__device__ __forceinline__ void coef_scalar(float gs, float gm, float gmn, float gmx, float gv, float gsd, float s, float q,
                                            float cnt, float* P, float* Q) {
    const float n = fmaxf(cnt, 1.f);
    const float mean = s / n, msq = q / n;
    const float var = msq - mean * mean;
    const float sd = sqrtf(fmaxf(var, 0.f) + 1e-5f);
    const float gvt = gv + (var > 0.f ? gsd / (2.f * sd) : 0.f);
    *P = gs + gm / n - 2.f * mean * gvt / n;
    *Q = 2.f * gvt / n;
    (void)gmn; (void)gmx;
}

template <int LPR, bool HAS_EE>
__global__ __launch_bounds__(PNA_BLOCK) void k_pna_bwd_dst(
    const float* __restrict__ x, const float* __restrict__ att, const float* __restrict__ edge_emb,
    const float* __restrict__ dout, const int32_t* __restrict__ rowptr, const int32_t* __restrict__ col,
    const int32_t* __restrict__ eid, int num_rows, int H, PnaCfg cfg, float* __restrict__ dx_self,
    float* __restrict__ dmsg, float* __restrict__ datt, float* __restrict__ dedge, int rows_per_group) {
    constexpr int GPB = PNA_BLOCK / LPR;
    const int lane = threadIdx.x % LPR;
    const int c = lane * 4;
    const bool on = c < H;
    const int parts = HAS_EE ? 3 : 2;
    const int F = parts * H;
    const size_t out_stride = (size_t)cfg.S * cfg.A * F;
    const int grp = pna_xcd_remap(blockIdx.x, gridDim.x) * GPB + threadIdx.x / LPR;
    int row = grp * rows_per_group;
    const int row_end = min(num_rows, row + rows_per_group);
    for (; row < row_end; ++row) {
        const int beg = rowptr[row], end = rowptr[row + 1];
        const float cnt = (float)(end - beg);
        if (end == beg) {          // no in-edges: every aggregate is a constant of x
            if (on) st4(dx_self + (size_t)row * H + c, f4zero());
            continue;
        }
        // ---- fold the scalers: per aggregator KIND, per message part, the upstream gradient ----
        float4 g[6][HAS_EE ? 3 : 2];
#pragma unroll
        for (int kd = 0; kd < 6; ++kd)
#pragma unroll
            for (int p = 0; p < parts; ++p) g[kd][p] = f4zero();
        if (on) {
            const float* drow = dout + (size_t)row * out_stride;
            for (int s = 0; s < cfg.S; ++s) {
                const float f = scaler_factor(cfg.scal[s], cnt, cfg.avg_lin, cfg.avg_log);
                for (int a = 0; a < cfg.A; ++a) {
                    const float* d = drow + (size_t)(s * cfg.A + a) * F + c;
                    const int kd = cfg.aggr[a];
#pragma unroll
                    for (int kk = 0; kk < 6; ++kk) {
                        if (kk == kd) {
#pragma unroll
                            for (int p = 0; p < parts; ++p) {
                                float4 v = ld4(d + p * H);
                                g[kk][p].x = fmaf(f, v.x, g[kk][p].x); g[kk][p].y = fmaf(f, v.y, g[kk][p].y);
                                g[kk][p].z = fmaf(f, v.z, g[kk][p].z); g[kk][p].w = fmaf(f, v.w, g[kk][p].w);
                            }
                        }
                    }
                }
            }
        }
        // ---- pass 1: row statistics and first-occurrence args ---------------------------------
        float sa = 0.f, sa2 = 0.f, amin = INFINITY, amax = -INFINITY;
        int kmin_a = beg, kmax_a = beg;
        Acc4 aj, ae;
        aj.init(); ae.init();
        int4 jmin = make_int4(beg, beg, beg, beg), jmax = jmin, emin = jmin, emax = jmin;
        const float4 xi = on ? ld4(x + (size_t)row * H + c) : f4zero();
        for (int k = beg; k < end; ++k) {
            const int j = col[k];
            const int e = (att != nullptr || HAS_EE) ? eid[k] : 0;
            const float w = att ? att[e] : 1.f;
            sa += w; sa2 = fmaf(w, w, sa2);
            if (w < amin) { amin = w; kmin_a = k; }
            if (w > amax) { amax = w; kmax_a = k; }
            if (on) {
                float4 m = f4scale(w, ld4(x + (size_t)j * H + c));
                if (m.x < aj.mn.x) jmin.x = k; if (m.y < aj.mn.y) jmin.y = k; if (m.z < aj.mn.z) jmin.z = k; if (m.w < aj.mn.w) jmin.w = k;
                if (m.x > aj.mx.x) jmax.x = k; if (m.y > aj.mx.y) jmax.y = k; if (m.z > aj.mx.z) jmax.z = k; if (m.w > aj.mx.w) jmax.w = k;
                aj.add(m);
                if (HAS_EE) {
                    float4 me = f4scale(w, ld4(edge_emb + (size_t)e * H + c));
                    if (me.x < ae.mn.x) emin.x = k; if (me.y < ae.mn.y) emin.y = k; if (me.z < ae.mn.z) emin.z = k; if (me.w < ae.mn.w) emin.w = k;
                    if (me.x > ae.mx.x) emax.x = k; if (me.y > ae.mx.y) emax.y = k; if (me.z > ae.mx.z) emax.z = k; if (me.w > ae.mx.w) emax.w = k;
                    ae.add(me);
                }
            }
        }
        const Acc4 ai = self_stats(xi, sa, sa2, amin, amax);
        // x_i part: arg of att*x_i is the arg-min/max of att by the sign of x_i (first slot when x_i == 0)
        int4 imin, imax;
        imin.x = xi.x > 0.f ? kmin_a : (xi.x < 0.f ? kmax_a : beg); imax.x = xi.x > 0.f ? kmax_a : (xi.x < 0.f ? kmin_a : beg);
        imin.y = xi.y > 0.f ? kmin_a : (xi.y < 0.f ? kmax_a : beg); imax.y = xi.y > 0.f ? kmax_a : (xi.y < 0.f ? kmin_a : beg);
        imin.z = xi.z > 0.f ? kmin_a : (xi.z < 0.f ? kmax_a : beg); imax.z = xi.z > 0.f ? kmax_a : (xi.z < 0.f ? kmin_a : beg);
        imin.w = xi.w > 0.f ? kmin_a : (xi.w < 0.f ? kmax_a : beg); imax.w = xi.w > 0.f ? kmax_a : (xi.w < 0.f ? kmin_a : beg);
        // ---- coefficients ---------------------------------------------------------------------
        float4 Pi, Qi, Pj, Qj, Pe = f4zero(), Qe = f4zero();
#define GSAT_COEF(P, Q, ACC, PART)                                                                                   \
        coef_scalar(g[0][PART].x, g[1][PART].x, 0, 0, g[4][PART].x, g[5][PART].x, ACC.s.x, ACC.q.x, cnt, &P.x, &Q.x); \
        coef_scalar(g[0][PART].y, g[1][PART].y, 0, 0, g[4][PART].y, g[5][PART].y, ACC.s.y, ACC.q.y, cnt, &P.y, &Q.y); \
        coef_scalar(g[0][PART].z, g[1][PART].z, 0, 0, g[4][PART].z, g[5][PART].z, ACC.s.z, ACC.q.z, cnt, &P.z, &Q.z); \
        coef_scalar(g[0][PART].w, g[1][PART].w, 0, 0, g[4][PART].w, g[5][PART].w, ACC.s.w, ACC.q.w, cnt, &P.w, &Q.w);
        GSAT_COEF(Pi, Qi, ai, 0)
        GSAT_COEF(Pj, Qj, aj, 1)
        if (HAS_EE) { GSAT_COEF(Pe, Qe, ae, 2) }
#undef GSAT_COEF
        // ---- pass 2: per-edge gradients -------------------------------------------------------
        float4 dxi = f4zero();
        for (int k = beg; k < end; ++k) {
            const int j = col[k];
            const int e = eid[k];
            const float w = att ? att[e] : 1.f;
            float da = 0.f;
            if (on) {
                // x_j part
                const float4 xj = ld4(x + (size_t)j * H + c);
                float4 dm;
                dm.x = fmaf(Qj.x, w * xj.x, Pj.x) + (k == jmin.x ? g[2][1].x : 0.f) + (k == jmax.x ? g[3][1].x : 0.f);
                dm.y = fmaf(Qj.y, w * xj.y, Pj.y) + (k == jmin.y ? g[2][1].y : 0.f) + (k == jmax.y ? g[3][1].y : 0.f);
                dm.z = fmaf(Qj.z, w * xj.z, Pj.z) + (k == jmin.z ? g[2][1].z : 0.f) + (k == jmax.z ? g[3][1].z : 0.f);
                dm.w = fmaf(Qj.w, w * xj.w, Pj.w) + (k == jmin.w ? g[2][1].w : 0.f) + (k == jmax.w ? g[3][1].w : 0.f);
                st4(dmsg + (size_t)k * H + c, f4scale(w, dm));
                da += f4dot(dm, xj);
                // x_i part
                float4 di;
                di.x = fmaf(Qi.x, w * xi.x, Pi.x) + (k == imin.x ? g[2][0].x : 0.f) + (k == imax.x ? g[3][0].x : 0.f);
                di.y = fmaf(Qi.y, w * xi.y, Pi.y) + (k == imin.y ? g[2][0].y : 0.f) + (k == imax.y ? g[3][0].y : 0.f);
                di.z = fmaf(Qi.z, w * xi.z, Pi.z) + (k == imin.z ? g[2][0].z : 0.f) + (k == imax.z ? g[3][0].z : 0.f);
                di.w = fmaf(Qi.w, w * xi.w, Pi.w) + (k == imin.w ? g[2][0].w : 0.f) + (k == imax.w ? g[3][0].w : 0.f);
                dxi = f4fma(w, di, dxi);
                da += f4dot(di, xi);
                if (HAS_EE) {
                    const float4 ee = ld4(edge_emb + (size_t)e * H + c);
                    float4 de;
                    de.x = fmaf(Qe.x, w * ee.x, Pe.x) + (k == emin.x ? g[2][2].x : 0.f) + (k == emax.x ? g[3][2].x : 0.f);
                    de.y = fmaf(Qe.y, w * ee.y, Pe.y) + (k == emin.y ? g[2][2].y : 0.f) + (k == emax.y ? g[3][2].y : 0.f);
                    de.z = fmaf(Qe.z, w * ee.z, Pe.z) + (k == emin.z ? g[2][2].z : 0.f) + (k == emax.z ? g[3][2].z : 0.f);
                    de.w = fmaf(Qe.w, w * ee.w, Pe.w) + (k == emin.w ? g[2][2].w : 0.f) + (k == emax.w ? g[3][2].w : 0.f);
                    if (dedge) st4(dedge + (size_t)e * H + c, f4scale(w, de));
                    da += f4dot(de, ee);
                }
            }
            if (datt) {
                da = group_sum<LPR>(da);
                if (lane == 0) datt[e] = da;
            }
        }
        if (on) st4(dx_self + (size_t)row * H + c, dxi);
    }
}

static inline int pna_lpr(int64_t H) {
    if (H <= 0 || H % 4 != 0 || H > 256) return 0;
    int64_t q = H / 4;
    return q <= 4 ? 4 : q <= 8 ? 8 : q <= 16 ? 16 : q <= 32 ? 32 : 64;
}

static inline void pna_grid(int64_t N, int lpr, int* nb, int* rpg) {
    const int gpb = PNA_BLOCK / lpr;
    int64_t b = ceil_div(N, gpb);
    if (b > 256 * 16) b = 256 * 16;
    if (b < 1) b = 1;
    *nb = (int)b;
    *rpg = (int)std::max<int64_t>(1, ceil_div(N, b * gpb));
}

static int make_cfg(const int32_t* aggr, int A, const int32_t* scal, int S, float avg_lin, float avg_log, PnaCfg* cfg) {
    GSAT_REQUIRE(aggr && scal && A >= 1 && A <= 8 && S >= 1 && S <= 8, GSAT_ERR_ARG, "pna: need 1..8 aggregators and scalers");
    cfg->A = A; cfg->S = S; cfg->avg_lin = avg_lin; cfg->avg_log = avg_log;
    for (int i = 0; i < 8; ++i) { cfg->aggr[i] = 0; cfg->scal[i] = 0; }
    for (int i = 0; i < A; ++i) {
        GSAT_REQUIRE(aggr[i] >= 0 && aggr[i] <= 5, GSAT_ERR_ARG, "pna: unknown aggregator code %d", aggr[i]);
        cfg->aggr[i] = aggr[i];
    }
    for (int i = 0; i < S; ++i) {
        GSAT_REQUIRE(scal[i] >= 0 && scal[i] <= 4, GSAT_ERR_ARG, "pna: unknown scaler code %d", scal[i]);
        cfg->scal[i] = scal[i];
    }
    return GSAT_OK;
}

}  // namespace gsat

using namespace gsat;

#define GSAT_LPR_DISPATCH(lpr, CALL) \
    switch (lpr) { case 4: CALL(4); break; case 8: CALL(8); break; case 16: CALL(16); break; case 32: CALL(32); break; default: CALL(64); break; }

extern "C" {

int gsat_pna_fwd(const float* x, const float* att, const float* edge_emb, const int32_t* rowptr, const int32_t* col,
                 const int32_t* eid, int64_t N, int64_t H, const int32_t* aggregators, int A, const int32_t* scalers, int S,
                 float avg_deg_lin, float avg_deg_log, float* out, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    GSAT_REQUIRE(N >= 0 && N < (1ll << 31), GSAT_ERR_ARG, "gsat_pna_fwd: bad N");
    PnaCfg cfg;
    int rc = make_cfg(aggregators, A, scalers, S, avg_deg_lin, avg_deg_log, &cfg);
    if (rc) return rc;
    if (N == 0) return GSAT_OK;
    const int lpr = pna_lpr(H);
    GSAT_REQUIRE(lpr, GSAT_ERR_UNSUPPORTED, "gsat_pna_fwd: H=%lld must be a multiple of 4 and <= 256", (long long)H);
    GSAT_REQUIRE(x && rowptr && col && out && eid, GSAT_ERR_ARG, "gsat_pna_fwd: null pointer");
    int nb, rpg;
    pna_grid(N, lpr, &nb, &rpg);
#define CALL(L)                                                                                                              \
    do {                                                                                                                     \
        if (edge_emb) k_pna_fwd<L, true><<<nb, PNA_BLOCK, 0, stream>>>(x, att, edge_emb, rowptr, col, eid, (int)N, (int)H, cfg, out, rpg); \
        else k_pna_fwd<L, false><<<nb, PNA_BLOCK, 0, stream>>>(x, att, edge_emb, rowptr, col, eid, (int)N, (int)H, cfg, out, rpg);         \
    } while (0)
    GSAT_LPR_DISPATCH(lpr, CALL);
#undef CALL
    GSAT_LAUNCH_CHECK();
    return GSAT_OK;
}

int gsat_pna_bwd(const float* x, const float* att, const float* edge_emb, const float* dout, const int32_t* rowptr,
                 const int32_t* col, const int32_t* eid, int64_t N, int64_t H, const int32_t* aggregators, int A,
                 const int32_t* scalers, int S, float avg_deg_lin, float avg_deg_log, float* dx_self, float* dmsg,
                 float* datt, float* dedge_emb, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    GSAT_REQUIRE(N >= 0 && N < (1ll << 31), GSAT_ERR_ARG, "gsat_pna_bwd: bad N");
    PnaCfg cfg;
    int rc = make_cfg(aggregators, A, scalers, S, avg_deg_lin, avg_deg_log, &cfg);
    if (rc) return rc;
    if (N == 0) return GSAT_OK;
    const int lpr = pna_lpr(H);
    GSAT_REQUIRE(lpr, GSAT_ERR_UNSUPPORTED, "gsat_pna_bwd: H=%lld must be a multiple of 4 and <= 256", (long long)H);
    GSAT_REQUIRE(x && dout && rowptr && col && eid && dx_self && dmsg, GSAT_ERR_ARG, "gsat_pna_bwd: null pointer");
    int nb, rpg;
    pna_grid(N, lpr, &nb, &rpg);
#define CALL(L)                                                                                                              \
    do {                                                                                                                     \
        if (edge_emb) k_pna_bwd_dst<L, true><<<nb, PNA_BLOCK, 0, stream>>>(x, att, edge_emb, dout, rowptr, col, eid, (int)N, (int)H, cfg, dx_self, dmsg, datt, dedge_emb, rpg); \
        else k_pna_bwd_dst<L, false><<<nb, PNA_BLOCK, 0, stream>>>(x, att, edge_emb, dout, rowptr, col, eid, (int)N, (int)H, cfg, dx_self, dmsg, datt, dedge_emb, rpg);         \
    } while (0)
    GSAT_LPR_DISPATCH(lpr, CALL);
#undef CALL
    GSAT_LAUNCH_CHECK();
    return GSAT_OK;
}

}  // extern "C"
