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
#include "pna_math.h"

namespace gsat {

constexpr int PNA_BLOCK = 128;
constexpr int SC_AMP = GSAT_SCALE_AMPLIFICATION, SC_ATT = GSAT_SCALE_ATTENUATION, SC_LIN = GSAT_SCALE_LINEAR, SC_INVLIN = GSAT_SCALE_INVERSE_LINEAR;

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

// ---- long (hub) rows ----------------------------------------------------------------------------------------------------------
// A row with more than GSAT_LONG_ROW_EDGES in-edges would keep ONE lane group busy for its whole length (a 1e5-edge hub: ~25 ms).
// With the index's hub-chunk list (chunk_ptr, gsat_row_chunks) the row kernels treat it in three steps instead:
//   k_pna_chunk_stats   one lane group per 256-edge chunk: running statistics of the chunk (sum, sum of squares, min, max and their
//                       FIRST slots) per channel and the chunk's attention statistics -> record[item]
//   k_pna_fwd / k_pna_bwd_dst   the row's lane group folds its chunks' records in chunk order (strict comparisons keep the first
//                       occurrence across chunks, sums are added chunk by chunk: reproducible); the backward then stores the row's
//                       routing coefficients in the record slot of its first chunk
//   k_pna_chunk_apply   (backward) one lane group per chunk: the per-edge gradients of its 256 edges from that record
// record = 6 planes of H per gathered part (x_j, edge_emb) + 8 scalars.
constexpr int CH = GSAT_LONG_ROW_EDGES;

__host__ __device__ __forceinline__ size_t pna_rec_floats(int H, int gparts) { return (size_t)6 * gparts * H + 8; }
static inline int64_t pna_max_chunks(int64_t E) { return 2 * (E / CH) + 1; }

__device__ __forceinline__ void pna_chunk_item(const int32_t* __restrict__ chunk_ptr, const int32_t* __restrict__ rowptr, int num_rows,
                                               int item, int* row, int* beg, int* end) {
    int lo = 0, hi = num_rows;                   // invariant: chunk_ptr[lo] <= item < chunk_ptr[hi]
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (chunk_ptr[mid] <= item) lo = mid; else hi = mid;
    }
    *row = lo;
    *beg = rowptr[lo] + (item - chunk_ptr[lo]) * CH;
    *end = min(rowptr[lo + 1], *beg + CH);
}

__device__ __forceinline__ void st4i(float* p, int4 v) { *reinterpret_cast<int4*>(p) = v; }
__device__ __forceinline__ int4 ld4i(const float* p) { return *reinterpret_cast<const int4*>(p); }

// strict-comparison update of (min, max, first slots) by one message
__device__ __forceinline__ void arg_update(const Acc4& a, float4 m, int kk, int4& jmin, int4& jmax) {
    if (m.x < a.mn.x) jmin.x = kk; if (m.y < a.mn.y) jmin.y = kk; if (m.z < a.mn.z) jmin.z = kk; if (m.w < a.mn.w) jmin.w = kk;
    if (m.x > a.mx.x) jmax.x = kk; if (m.y > a.mx.y) jmax.y = kk; if (m.z > a.mx.z) jmax.z = kk; if (m.w > a.mx.w) jmax.w = kk;
}

// fold a chunk's (statistics, first slots) into the row's: later chunks win only on strictly smaller / larger values
__device__ __forceinline__ void rec_combine(Acc4& a, int4& jmin, int4& jmax, const float* __restrict__ r, int H, int c) {
    const float4 s = ld4(r + c), q = ld4(r + H + c), mn = ld4(r + 2 * H + c), mx = ld4(r + 3 * H + c);
    const int4 pmin = ld4i(r + 4 * H + c), pmax = ld4i(r + 5 * H + c);
    a.s.x += s.x; a.s.y += s.y; a.s.z += s.z; a.s.w += s.w;
    a.q.x += q.x; a.q.y += q.y; a.q.z += q.z; a.q.w += q.w;
    if (mn.x < a.mn.x) { a.mn.x = mn.x; jmin.x = pmin.x; } if (mn.y < a.mn.y) { a.mn.y = mn.y; jmin.y = pmin.y; }
    if (mn.z < a.mn.z) { a.mn.z = mn.z; jmin.z = pmin.z; } if (mn.w < a.mn.w) { a.mn.w = mn.w; jmin.w = pmin.w; }
    if (mx.x > a.mx.x) { a.mx.x = mx.x; jmax.x = pmax.x; } if (mx.y > a.mx.y) { a.mx.y = mx.y; jmax.y = pmax.y; }
    if (mx.z > a.mx.z) { a.mx.z = mx.z; jmax.z = pmax.z; } if (mx.w > a.mx.w) { a.mx.w = mx.w; jmax.w = pmax.w; }
}

template <int LPR, bool HAS_EE>
__global__ __launch_bounds__(PNA_BLOCK) void k_pna_chunk_stats(
    const float* __restrict__ x, const float* __restrict__ att, const float* __restrict__ edge_emb, const int32_t* __restrict__ rowptr,
    const int32_t* __restrict__ col, const int32_t* __restrict__ eid, int num_rows, int H, const int32_t* __restrict__ chunk_ptr,
    float* __restrict__ partial, int c0, int Hc) {
    const LaneGroups<LPR, PNA_BLOCK> lg;
    constexpr int GPB = LaneGroups<LPR, PNA_BLOCK>::GPB;
    if (lg.grp < 0) return;
    const int lane = lg.lane, c = c0 + lane * 4;
    const bool on = lane * 4 < Hc;
    constexpr int GP = HAS_EE ? 2 : 1;
    const size_t rec = pna_rec_floats(H, GP);
    const int total = chunk_ptr[num_rows];
    for (int item = blockIdx.x * GPB + lg.grp; item < total; item += gridDim.x * GPB) {
        int row, beg, end;
        pna_chunk_item(chunk_ptr, rowptr, num_rows, item, &row, &beg, &end);
        float sa = 0.f, sa2 = 0.f, amin = INFINITY, amax = -INFINITY, wfirst = 1.f;
        int kmin_a = beg, kmax_a = beg;
        Acc4 aj, ae;
        aj.init(); ae.init();
        int4 jmin = make_int4(beg, beg, beg, beg), jmax = jmin, emin = jmin, emax = jmin;
        for (int k = beg; k < end; k += 4) {
            const int nb = min(4, end - k);
            int j[4], e[4];
            float w[4];
            float4 xv[4], ev[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) { j[u] = u < nb ? col[k + u] : 0; e[u] = u < nb ? eid[k + u] : 0; }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                w[u] = (u < nb && att) ? att[e[u]] : 1.f;
                xv[u] = (u < nb && on) ? ld4(x + (size_t)j[u] * H + c) : f4zero();
                if (HAS_EE) ev[u] = (u < nb && on) ? ld4(edge_emb + (size_t)e[u] * H + c) : f4zero();
            }
            if (k == beg) wfirst = w[0];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                if (u >= nb) continue;
                const int kk = k + u;
                sa += w[u]; sa2 = fmaf(w[u], w[u], sa2);
                if (w[u] < amin) { amin = w[u]; kmin_a = kk; }
                if (w[u] > amax) { amax = w[u]; kmax_a = kk; }
                if (on) {
                    const float4 m = f4scale(w[u], xv[u]);
                    arg_update(aj, m, kk, jmin, jmax);
                    aj.add(m);
                    if (HAS_EE) {
                        const float4 me = f4scale(w[u], ev[u]);
                        arg_update(ae, me, kk, emin, emax);
                        ae.add(me);
                    }
                }
            }
        }
        float* r = partial + (size_t)item * rec;
        if (on) {
            st4(r + c, aj.s); st4(r + H + c, aj.q); st4(r + 2 * H + c, aj.mn); st4(r + 3 * H + c, aj.mx);
            st4i(r + 4 * H + c, jmin); st4i(r + 5 * H + c, jmax);
            if (HAS_EE) {
                float* re = r + 6 * H;
                st4(re + c, ae.s); st4(re + H + c, ae.q); st4(re + 2 * H + c, ae.mn); st4(re + 3 * H + c, ae.mx);
                st4i(re + 4 * H + c, emin); st4i(re + 5 * H + c, emax);
            }
        }
        if (lane == 0) {
            float* sc = r + (size_t)6 * GP * H;
            st4(sc, make_float4(sa, sa2, amin, amax));
            st4(sc + 4, make_float4(__int_as_float(kmin_a), __int_as_float(kmax_a), wfirst, 0.f));
        }
    }
}

// backward of the long rows' edges from the row record written by k_pna_bwd_dst (same per-edge arithmetic as its pass 2)
template <int LPR, bool HAS_EE>
__global__ __launch_bounds__(PNA_BLOCK) void k_pna_chunk_apply(
    const float* __restrict__ x, const float* __restrict__ att, const float* __restrict__ edge_emb, const int32_t* __restrict__ rowptr,
    const int32_t* __restrict__ col, const int32_t* __restrict__ eid, int num_rows, int H, const int32_t* __restrict__ chunk_ptr,
    const float* __restrict__ partial, float* __restrict__ dmsg, float* __restrict__ datt, float* __restrict__ dedge, int c0, int Hc) {
    const LaneGroups<LPR, PNA_BLOCK> lg;
    constexpr int GPB = LaneGroups<LPR, PNA_BLOCK>::GPB;
    if (lg.grp < 0) return;
    const int lane = lg.lane, c = c0 + lane * 4;
    const bool on = lane * 4 < Hc;
    constexpr int GP = HAS_EE ? 2 : 1;
    const size_t rec = pna_rec_floats(H, GP);
    const int total = chunk_ptr[num_rows];
    for (int item = blockIdx.x * GPB + lg.grp; item < total; item += gridDim.x * GPB) {
        int row, beg, end;
        pna_chunk_item(chunk_ptr, rowptr, num_rows, item, &row, &beg, &end);
        const float* r = partial + (size_t)chunk_ptr[row] * rec;
        float4 Pj = f4zero(), Qj = f4zero(), gmn_j = f4zero(), gmx_j = f4zero(), Pe = f4zero(), Qe = f4zero(), gmn_e = f4zero(), gmx_e = f4zero();
        int4 jmin = make_int4(-1, -1, -1, -1), jmax = jmin, emin = jmin, emax = jmin;
        if (on) {
            Pj = ld4(r + c); Qj = ld4(r + H + c); gmn_j = ld4(r + 2 * H + c); gmx_j = ld4(r + 3 * H + c);
            jmin = ld4i(r + 4 * H + c); jmax = ld4i(r + 5 * H + c);
            if (HAS_EE) {
                const float* re = r + 6 * H;
                Pe = ld4(re + c); Qe = ld4(re + H + c); gmn_e = ld4(re + 2 * H + c); gmx_e = ld4(re + 3 * H + c);
                emin = ld4i(re + 4 * H + c); emax = ld4i(re + 5 * H + c);
            }
        }
        const float4 ts = ld4(r + (size_t)6 * GP * H), ks = ld4(r + (size_t)6 * GP * H + 4);
        const float t1 = ts.x, t2 = ts.y, tmn = ts.z, tmx = ts.w;
        const int kmin_a = __float_as_int(ks.x), kmax_a = __float_as_int(ks.y);
        for (int kb = beg; kb < end; kb += 4) {
            const int nb = min(4, end - kb);
            int j4[4], e4[4];
            float w4[4];
            float4 x4[4], ee4[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) { j4[u] = u < nb ? col[kb + u] : 0; e4[u] = u < nb ? eid[kb + u] : 0; }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                w4[u] = (u < nb && att) ? att[e4[u]] : 1.f;
                x4[u] = (u < nb && on) ? ld4(x + (size_t)j4[u] * H + c) : f4zero();
                if (HAS_EE) ee4[u] = (u < nb && on) ? ld4(edge_emb + (size_t)e4[u] * H + c) : f4zero();
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                if (u >= nb) continue;
                const int k = kb + u, e = e4[u];
                const float w = w4[u];
                const float4 xj = x4[u];
                float da = 0.f;
                if (on) {
                    float4 dm;
                    dm.x = fmaf(Qj.x, w * xj.x, Pj.x) + (k == jmin.x ? gmn_j.x : 0.f) + (k == jmax.x ? gmx_j.x : 0.f);
                    dm.y = fmaf(Qj.y, w * xj.y, Pj.y) + (k == jmin.y ? gmn_j.y : 0.f) + (k == jmax.y ? gmx_j.y : 0.f);
                    dm.z = fmaf(Qj.z, w * xj.z, Pj.z) + (k == jmin.z ? gmn_j.z : 0.f) + (k == jmax.z ? gmx_j.z : 0.f);
                    dm.w = fmaf(Qj.w, w * xj.w, Pj.w) + (k == jmin.w ? gmn_j.w : 0.f) + (k == jmax.w ? gmx_j.w : 0.f);
                    st4(dmsg + (size_t)k * H + c, f4scale(w, dm));
                    da += f4dot(dm, xj);
                    if (HAS_EE) {
                        const float4 ee = ee4[u];
                        float4 de;
                        de.x = fmaf(Qe.x, w * ee.x, Pe.x) + (k == emin.x ? gmn_e.x : 0.f) + (k == emax.x ? gmx_e.x : 0.f);
                        de.y = fmaf(Qe.y, w * ee.y, Pe.y) + (k == emin.y ? gmn_e.y : 0.f) + (k == emax.y ? gmx_e.y : 0.f);
                        de.z = fmaf(Qe.z, w * ee.z, Pe.z) + (k == emin.z ? gmn_e.z : 0.f) + (k == emax.z ? gmx_e.z : 0.f);
                        de.w = fmaf(Qe.w, w * ee.w, Pe.w) + (k == emin.w ? gmn_e.w : 0.f) + (k == emax.w ? gmx_e.w : 0.f);
                        if (dedge) st4(dedge + (size_t)e * H + c, f4scale(w, de));
                        da += f4dot(de, ee);
                    }
                }
                if (datt) {
                    da = group_sum<LPR>(da);
                    if (lane == 0) {
                        da += t1 + w * t2 + (k == kmin_a ? tmn : 0.f) + (k == kmax_a ? tmx : 0.f);     // the x_i part, summed over the lanes by the row's group
                        datt[e] = c0 ? datt[e] + da : da;
                    }
                }
            }
        }
    }
}

// LONG = false: every row of the lane group's span except, when a hub-chunk list is given, the rows of more than GSAT_LONG_ROW_EDGES
// in-edges; LONG = true: exactly those rows (one lane group per row, found through the chunk list), their statistics folded from the
// chunk records.  Two instantiations rather than a branch: the fold's registers would cost the common kernel a wave of occupancy.
template <int LPR, bool HAS_EE, int NAGG, bool LONG>
__global__ __launch_bounds__(PNA_BLOCK) void k_pna_fwd(
    const float* __restrict__ x, const float* __restrict__ att, const float* __restrict__ edge_emb,
    const int32_t* __restrict__ rowptr, const int32_t* __restrict__ col, const int32_t* __restrict__ eid,
    int num_rows, int H, PnaCfg cfg, float* __restrict__ out, int rows_per_group, int c0, int Hc,
    const int32_t* __restrict__ chunk_ptr, const float* __restrict__ partial, float* __restrict__ scal_out) {
    // scal_out != NULL (fixed aggregator sets, no edge features): COMPACT output -- `out` is [N, NAGG * H] with the x_j parts only and
    // scal_out[row] = four coefficients of the row's edge weights; the x_i parts are a closed form of x_i and those four (pna_math.h: PnaVirt), which the
    // post_nn GEMM's operand loader recomputes instead of reading them back.
    // channels [c0, c0 + Hc) of rows that are H wide: widths above 256 run as one launch per 256-channel chunk
    const LaneGroups<LPR, PNA_BLOCK> lg;
    constexpr int GPB = LaneGroups<LPR, PNA_BLOCK>::GPB;
    const int lane = lg.lane;
    const int c = c0 + lane * 4;
    const bool on = lane * 4 < Hc;
    const int parts = HAS_EE ? 3 : 2;
    const int F = parts * H;
    const size_t out_stride = (size_t)cfg.S * cfg.A * F;
    if (lg.grp < 0) return;
    const int grp = pna_xcd_remap(blockIdx.x, gridDim.x) * GPB + lg.grp;
    int next_row = grp * rows_per_group;
    const int row_end = min(num_rows, next_row + rows_per_group);
    int item = blockIdx.x * GPB + lg.grp;
    const int total = LONG ? chunk_ptr[num_rows] : 0;
    for (;;) {
        int row;
        if (LONG) {                 // the lane group that draws a row's FIRST chunk owns the row
            if (item >= total) break;
            int b_, e_;
            pna_chunk_item(chunk_ptr, rowptr, num_rows, item, &row, &b_, &e_);
            const bool first = item == chunk_ptr[row];
            item += gridDim.x * GPB;
            if (!first) continue;
        } else {
            if (next_row >= row_end) break;
            row = next_row++;
        }
        const int beg = rowptr[row], end = rowptr[row + 1];
        if (!LONG && chunk_ptr != nullptr && end - beg > CH) continue;
        const float cnt = (float)(end - beg);
        float sa = 0.f, sa2 = 0.f, amin = INFINITY, amax = -INFINITY;
        Acc4 aj, ae;
        aj.init(); ae.init();
        const float4 xi = on ? ld4(x + (size_t)row * H + c) : f4zero();
        // node attention (att != NULL, eid == NULL): att holds one value per NODE and the edge weight is att[row] * att[source], the
        // lifted attention of example/gsat.py:112-117 formed at the load (same product, same bits, no [E] tensor, no edge ids)
        const bool natt = !HAS_EE && !LONG && att != nullptr && eid == nullptr;
        const float na_row = natt ? att[row] : 1.f;
        if (LONG) {                 // hub row: fold the records k_pna_chunk_stats wrote for its chunks, in chunk order
            constexpr int GP = HAS_EE ? 2 : 1;
            const size_t rec = pna_rec_floats(H, GP);
            int4 d0, d1;
            for (int p = chunk_ptr[row]; p < chunk_ptr[row + 1]; ++p) {
                const float* r = partial + (size_t)p * rec;
                if (on) {
                    rec_combine(aj, d0, d1, r, H, c);
                    if (HAS_EE) rec_combine(ae, d0, d1, r + 6 * H, H, c);
                }
                const float4 sc = ld4(r + (size_t)6 * GP * H);
                sa += sc.x; sa2 += sc.y; amin = fminf(amin, sc.z); amax = fmaxf(amax, sc.w);
            }
        }
        for (int k = beg; k < (LONG ? beg : end); k += 4) {       // batches of 4 in-edges: all index / att / row loads issued together
            const int nb = min(4, end - k);
            int j[4], e[4];
            float w[4];
            float4 xv[4], ev[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                j[u] = u < nb ? col[k + u] : 0;
                e[u] = (u < nb && !natt && (att != nullptr || HAS_EE)) ? eid[k + u] : 0;
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                w[u] = (u < nb && att) ? att[natt ? j[u] : e[u]] * na_row : 1.f;
                xv[u] = (u < nb && on) ? ld4(x + (size_t)j[u] * H + c) : f4zero();
                if (HAS_EE) ev[u] = (u < nb && on) ? ld4(edge_emb + (size_t)e[u] * H + c) : f4zero();
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                if (u < nb) {
                    sa += w[u]; sa2 = fmaf(w[u], w[u], sa2); amin = fminf(amin, w[u]); amax = fmaxf(amax, w[u]);
                    if (on) {
                        aj.add(f4scale(w[u], xv[u]));
                        if (HAS_EE) ae.add(f4scale(w[u], ev[u]));
                    }
                }
            }
        }
        if (NAGG && !HAS_EE && scal_out) {
            if (lane == 0 && c0 == 0) pna_scal_store(scal_out, row, sa, sa2, amin, amax, cnt);
            if (!on) continue;
            float* o = out + (size_t)row * ((size_t)(NAGG ? NAGG : 1) * H) + c;
            constexpr int kAggC[5] = {AGG_MEAN, AGG_MIN, AGG_MAX, GSAT_AGG_STD, AGG_SUM};
#pragma unroll
            for (int a = 0; a < (NAGG ? NAGG : 1); ++a) st4(o + (size_t)a * H, agg_value4(kAggC[a], aj, cnt));
            continue;
        }
        if (!on) continue;
        const Acc4 ai = self_stats(xi, sa, sa2, amin, amax);
        float* orow = out + (size_t)row * out_stride;
        if (NAGG) {        // (mean, min, max, std[, sum]; identity): NAGG * parts fixed segments, order [aggregator][part]
            float* o = orow + c;
            constexpr int kAgg[5] = {AGG_MEAN, AGG_MIN, AGG_MAX, GSAT_AGG_STD, AGG_SUM};
#pragma unroll
            for (int a = 0; a < NAGG; ++a) {
                st4_nt(o + (size_t)(a * parts) * H, agg_value4(kAgg[a], ai, cnt));
                st4_nt(o + (size_t)(a * parts + 1) * H, agg_value4(kAgg[a], aj, cnt));
                if (HAS_EE) st4_nt(o + (size_t)(a * parts + 2) * H, agg_value4(kAgg[a], ae, cnt));
            }
            continue;
        }
        for (int s = 0; s < cfg.S; ++s) {
            const float f = scaler_factor(cfg.scal[s], cnt, cfg.avg_lin, cfg.avg_log);
            for (int a = 0; a < cfg.A; ++a) {
                float* o = orow + (size_t)(s * cfg.A + a) * F + c;
                st4_nt(o, f4scale(f, agg_value4(cfg.aggr[a], ai, cnt)));
                st4_nt(o + H, f4scale(f, agg_value4(cfg.aggr[a], aj, cnt)));
                if (HAS_EE) st4_nt(o + 2 * H, f4scale(f, agg_value4(cfg.aggr[a], ae, cnt)));
            }
        }
    }
}

// NAGG = 4 | 5: the configurations of the reference's PNA YAMLs (aggregators mean,min,max,std[,sum]; scaler identity) with the
// segment loop resolved at compile time; the generic instantiation covers every other aggregator / scaler list.
template <int LPR, bool HAS_EE, int NAGG, bool LONG>
__global__ __launch_bounds__(PNA_BLOCK, HAS_EE ? 2 : 4) void k_pna_bwd_dst(
    const float* __restrict__ x, const float* __restrict__ att, const float* __restrict__ edge_emb,
    const float* __restrict__ dout, const int32_t* __restrict__ rowptr, const int32_t* __restrict__ col,
    const int32_t* __restrict__ eid, int num_rows, int H, PnaCfg cfg, float* __restrict__ dx_self,
    float* __restrict__ dmsg, float* __restrict__ datt, float* __restrict__ dedge, int rows_per_group, int c0, int Hc,
    const int32_t* __restrict__ chunk_ptr, float* __restrict__ partial) {
    // channels [c0, c0 + Hc) of H-wide rows (see k_pna_fwd); chunks after the first ADD their share of datt (launches are stream-ordered)
    const LaneGroups<LPR, PNA_BLOCK> lg;
    constexpr int GPB = LaneGroups<LPR, PNA_BLOCK>::GPB;
    // The upstream gradient row (S*A*parts segments of H floats, 64 % of the kernel's bytes) is fetched by LDS-DMA at
    // the top of the row, so its HBM latency overlaps the index -> att -> x_j dependency chain without holding VGPRs.
    constexpr int MAXSEG = NAGG ? NAGG * (HAS_EE ? 3 : 2) : 8;
    __shared__ float4 stage[MAXSEG][PNA_BLOCK];
    const int lane = lg.lane;
    const int c = c0 + lane * 4;
    const bool on = lane * 4 < Hc;
    const int parts = HAS_EE ? 3 : 2;
    const int F = parts * H;
    const size_t out_stride = (size_t)cfg.S * cfg.A * F;
    const int nseg = cfg.S * cfg.A * parts;
    const bool staged = NAGG != 0 || nseg <= MAXSEG;
    const int wave_base = (threadIdx.x >> 6) << 6;
    if (lg.grp < 0) return;
    const int grp = pna_xcd_remap(blockIdx.x, gridDim.x) * GPB + lg.grp;
    int next_row = grp * rows_per_group;
    const int row_end = min(num_rows, next_row + rows_per_group);
    int item = blockIdx.x * GPB + lg.grp;
    const int total = LONG ? chunk_ptr[num_rows] : 0;
    for (;;) {
        int row;
        if (LONG) {                 // see k_pna_fwd
            if (item >= total) break;
            int b_, e_;
            pna_chunk_item(chunk_ptr, rowptr, num_rows, item, &row, &b_, &e_);
            const bool first = item == chunk_ptr[row];
            item += gridDim.x * GPB;
            if (!first) continue;
        } else {
            if (next_row >= row_end) break;
            row = next_row++;
        }
        const int beg = rowptr[row], end = rowptr[row + 1];
        if (!LONG && chunk_ptr != nullptr && end - beg > CH) continue;
        const float cnt = (float)(end - beg);
        if (end == beg) {          // no in-edges: every aggregate is a constant of x
            if (on) st4(dx_self + (size_t)row * H + c, f4zero());
            continue;
        }
        if (staged && on) {
            const float* g0 = dout + (size_t)row * out_stride + c;
            for (int sg = 0; sg < nseg; ++sg)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(g0 + (size_t)sg * H),
                                                 (__attribute__((address_space(3))) void*)&stage[sg][wave_base], 16, 0, 0);
        }
        // ---- pass 1: row statistics and first-occurrence args ---------------------------------
        float sa = 0.f, sa2 = 0.f, amin = INFINITY, amax = -INFINITY;
        int kmin_a = beg, kmax_a = beg;
        Acc4 aj, ae;
        aj.init(); ae.init();
        int4 jmin = make_int4(beg, beg, beg, beg), jmax = jmin, emin = jmin, emax = jmin;
        const float4 xi = on ? ld4(x + (size_t)row * H + c) : f4zero();
        int lj[4] = {0, 0, 0, 0}, le[4] = {0, 0, 0, 0};      // indices / weights of the last batch: rows with <= 4 in-edges
        float lw[4] = {1.f, 1.f, 1.f, 1.f};                   // (all of a molecule graph) skip the second index + att round trip
        float wfirst = 1.f;                                   // att of the row's first slot (arg of att*x_i where x_i == 0)
        constexpr int GP = HAS_EE ? 2 : 1;
        const size_t rec = pna_rec_floats(H, GP);
        if (LONG) {                 // hub row: statistics and first slots from its chunks' records (k_pna_chunk_stats), in chunk order
            for (int p = chunk_ptr[row]; p < chunk_ptr[row + 1]; ++p) {
                const float* r = partial + (size_t)p * rec;
                if (on) {
                    rec_combine(aj, jmin, jmax, r, H, c);
                    if (HAS_EE) rec_combine(ae, emin, emax, r + 6 * H, H, c);
                }
                const float4 sc = ld4(r + (size_t)6 * GP * H), ks = ld4(r + (size_t)6 * GP * H + 4);
                sa += sc.x; sa2 += sc.y;
                if (sc.z < amin) { amin = sc.z; kmin_a = __float_as_int(ks.x); }
                if (sc.w > amax) { amax = sc.w; kmax_a = __float_as_int(ks.y); }
                if (p == chunk_ptr[row]) wfirst = ks.z;
            }
        }
        for (int k = beg; k < (LONG ? beg : end); k += 4) {
            const int nb = min(4, end - k);
            int j[4], e[4];
            float w[4];
            float4 xv[4], ev[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                j[u] = u < nb ? col[k + u] : 0;
                e[u] = u < nb ? eid[k + u] : 0;
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                w[u] = (u < nb && att) ? att[e[u]] : 1.f;
                xv[u] = (u < nb && on) ? ld4(x + (size_t)j[u] * H + c) : f4zero();
                if (HAS_EE) ev[u] = (u < nb && on) ? ld4(edge_emb + (size_t)e[u] * H + c) : f4zero();
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) { lj[u] = j[u]; le[u] = e[u]; lw[u] = w[u]; }
            if (k == beg) wfirst = w[0];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                if (u < nb) {
                    const int kk = k + u;
                    sa += w[u]; sa2 = fmaf(w[u], w[u], sa2);
                    if (w[u] < amin) { amin = w[u]; kmin_a = kk; }
                    if (w[u] > amax) { amax = w[u]; kmax_a = kk; }
                    if (on) {
                        float4 m = f4scale(w[u], xv[u]);
                        if (m.x < aj.mn.x) jmin.x = kk; if (m.y < aj.mn.y) jmin.y = kk; if (m.z < aj.mn.z) jmin.z = kk; if (m.w < aj.mn.w) jmin.w = kk;
                        if (m.x > aj.mx.x) jmax.x = kk; if (m.y > aj.mx.y) jmax.y = kk; if (m.z > aj.mx.z) jmax.z = kk; if (m.w > aj.mx.w) jmax.w = kk;
                        aj.add(m);
                        if (HAS_EE) {
                            float4 me = f4scale(w[u], ev[u]);
                            if (me.x < ae.mn.x) emin.x = kk; if (me.y < ae.mn.y) emin.y = kk; if (me.z < ae.mn.z) emin.z = kk; if (me.w < ae.mn.w) emin.w = kk;
                            if (me.x > ae.mx.x) emax.x = kk; if (me.y > ae.mx.y) emax.y = kk; if (me.z > ae.mx.z) emax.z = kk; if (me.w > ae.mx.w) emax.w = kk;
                            ae.add(me);
                        }
                    }
                }
            }
        }
        const Acc4 ai = self_stats(xi, sa, sa2, amin, amax);
        // ---- fold scalers + aggregators straight into the routing coefficients of each message part ----
        //   d m_k = P + Q*m_k + gmin*[k==argmin] + gmax*[k==argmax]
        float t1 = 0.f, t2 = 0.f, tmn = 0.f, tmx = 0.f;
        float4 Pi = f4zero(), Qi = f4zero(), Pj = f4zero(), Qj = f4zero(), Pe = f4zero(), Qe = f4zero();
        float4 gmn_i = f4zero(), gmx_i = f4zero(), gmn_j = f4zero(), gmx_j = f4zero(), gmn_e = f4zero(), gmx_e = f4zero();
        if (on) {
            const float n = fmaxf(cnt, 1.f), inv_n = 1.f / n;
            const float* drow = dout + (size_t)row * out_stride + c;
            if (staged) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // the LDS-DMA of this row has landed
#define GSAT_FOLD(ACC, PART, P, Q, GMN, GMX)                                                                           \
            {                                                                                                          \
                const float4 mean = f4scale(inv_n, ACC.s);                                                             \
                float4 var = make_float4(ACC.q.x * inv_n - mean.x * mean.x, ACC.q.y * inv_n - mean.y * mean.y,         \
                                         ACC.q.z * inv_n - mean.z * mean.z, ACC.q.w * inv_n - mean.w * mean.w);        \
                const float4 hs = make_float4(var.x > 0.f ? 0.5f / sqrtf(var.x + 1e-5f) : 0.f, var.y > 0.f ? 0.5f / sqrtf(var.y + 1e-5f) : 0.f, \
                                              var.z > 0.f ? 0.5f / sqrtf(var.z + 1e-5f) : 0.f, var.w > 0.f ? 0.5f / sqrtf(var.w + 1e-5f) : 0.f); \
                float4 p0 = f4zero(), gv = f4zero();                                                                   \
                if (NAGG) {                                                                                            \
                    p0 = f4scale(inv_n, stage[0 * parts + (PART)][threadIdx.x]);                                       \
                    if (NAGG == 5) { const float4 vsum = stage[4 * parts + (PART)][threadIdx.x];                       \
                                     p0 = make_float4(p0.x + vsum.x, p0.y + vsum.y, p0.z + vsum.z, p0.w + vsum.w); }   \
                    GMN = stage[1 * parts + (PART)][threadIdx.x];                                                      \
                    GMX = stage[2 * parts + (PART)][threadIdx.x];                                                      \
                    const float4 vs = stage[3 * parts + (PART)][threadIdx.x];                                          \
                    gv = make_float4(vs.x * hs.x, vs.y * hs.y, vs.z * hs.z, vs.w * hs.w);                              \
                } else                                                                                                 \
                for (int s = 0; s < cfg.S; ++s) {                                                                      \
                    const float f = scaler_factor(cfg.scal[s], cnt, cfg.avg_lin, cfg.avg_log);                         \
                    for (int a = 0; a < cfg.A; ++a) {                                                                  \
                        const int sg_ = (s * cfg.A + a) * parts + (PART);                                              \
                        const float4 v = f4scale(f, staged ? stage[sg_][threadIdx.x] : ld4(drow + (size_t)sg_ * H));   \
                        switch (cfg.aggr[a]) {                                                                         \
                            case AGG_SUM: p0.x += v.x; p0.y += v.y; p0.z += v.z; p0.w += v.w; break;                   \
                            case AGG_MEAN: p0 = f4fma(inv_n, v, p0); break;                                            \
                            case AGG_MIN: GMN.x += v.x; GMN.y += v.y; GMN.z += v.z; GMN.w += v.w; break;               \
                            case AGG_MAX: GMX.x += v.x; GMX.y += v.y; GMX.z += v.z; GMX.w += v.w; break;               \
                            case AGG_VAR: gv.x += v.x; gv.y += v.y; gv.z += v.z; gv.w += v.w; break;                   \
                            default: gv.x = fmaf(v.x, hs.x, gv.x); gv.y = fmaf(v.y, hs.y, gv.y);                       \
                                     gv.z = fmaf(v.z, hs.z, gv.z); gv.w = fmaf(v.w, hs.w, gv.w); break;                \
                        }                                                                                              \
                    }                                                                                                  \
                }                                                                                                      \
                Q = f4scale(2.f * inv_n, gv);                                                                          \
                P = make_float4(p0.x - mean.x * Q.x, p0.y - mean.y * Q.y, p0.z - mean.z * Q.z, p0.w - mean.w * Q.w);   \
            }
            GSAT_FOLD(ai, 0, Pi, Qi, gmn_i, gmx_i)
        // ---- x_i part in closed form: every in-edge carries the same vector x_i, so its gradient needs no per-edge pass.
        //   d(att_k x_i) = Pi + Qi*att_k*x_i + gmn_i*[k == argmin] + gmx_i*[k == argmax],  arg of att*x_i = arg-min/max of att by
        //   the sign of x_i (first slot where x_i == 0)
        //   dx_i    = sum_k att_k * d(att_k x_i) = Pi*sum(att) + Qi*x_i*sum(att^2) + gmn_i*att[argmin] + gmx_i*att[argmax]
        //   datt_k += <d(att_k x_i), x_i> = t1 + att_k*t2 + [k == kmin_a]*tmn + [k == kmax_a]*tmx   (per-lane partial sums)
            {
#define GSAT_SELF(C)                                                                                                   \
            {                                                                                                          \
                const float xc = xi.C;                                                                                 \
                const float wlo = xc > 0.f ? amin : (xc < 0.f ? amax : wfirst), whi = xc > 0.f ? amax : (xc < 0.f ? amin : wfirst); \
                dxi.C = Pi.C * sa + Qi.C * xc * sa2 + gmn_i.C * wlo + gmx_i.C * whi;                                   \
                t1 = fmaf(Pi.C, xc, t1);                                                                               \
                t2 = fmaf(Qi.C * xc, xc, t2);                                                                          \
                tmn += xc > 0.f ? gmn_i.C * xc : (xc < 0.f ? gmx_i.C * xc : 0.f);                                      \
                tmx += xc > 0.f ? gmx_i.C * xc : (xc < 0.f ? gmn_i.C * xc : 0.f);                                      \
            }
            float4 dxi;
            GSAT_SELF(x) GSAT_SELF(y) GSAT_SELF(z) GSAT_SELF(w)
#undef GSAT_SELF
            st4(dx_self + (size_t)row * H + c, dxi);
            }
            GSAT_FOLD(aj, 1, Pj, Qj, gmn_j, gmx_j)
            if (HAS_EE) { GSAT_FOLD(ae, 2, Pe, Qe, gmn_e, gmx_e) }
#undef GSAT_FOLD
        }
        if (LONG) {                 // the per-edge pass of a hub row runs chunk-parallel in k_pna_chunk_apply, from this record
            float* r = partial + (size_t)chunk_ptr[row] * rec;          // the records of its chunks have been consumed above
            if (on) {
                st4(r + c, Pj); st4(r + H + c, Qj); st4(r + 2 * H + c, gmn_j); st4(r + 3 * H + c, gmx_j);
                st4i(r + 4 * H + c, jmin); st4i(r + 5 * H + c, jmax);
                if (HAS_EE) {
                    float* re = r + 6 * H;
                    st4(re + c, Pe); st4(re + H + c, Qe); st4(re + 2 * H + c, gmn_e); st4(re + 3 * H + c, gmx_e);
                    st4i(re + 4 * H + c, emin); st4i(re + 5 * H + c, emax);
                }
            }
            const float s1 = group_sum<LPR>(t1), s2 = group_sum<LPR>(t2), smn = group_sum<LPR>(tmn), smx = group_sum<LPR>(tmx);
            if (lane == 0) {
                st4(r + (size_t)6 * GP * H, make_float4(s1, s2, smn, smx));
                st4(r + (size_t)6 * GP * H + 4, make_float4(__int_as_float(kmin_a), __int_as_float(kmax_a), 0.f, 0.f));
            }
            continue;
        }
        // ---- pass 2: per-edge gradients -------------------------------------------------------
        for (int kb = beg; kb < end; kb += 4) {
            const int nb = min(4, end - kb);
            int j4[4], e4[4];
            float w4[4];
            float4 x4[4], ee4[4];
            if (end - beg <= 4) {
#pragma unroll
                for (int u = 0; u < 4; ++u) { j4[u] = lj[u]; e4[u] = le[u]; w4[u] = lw[u]; }
            } else {
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    j4[u] = u < nb ? col[kb + u] : 0;
                    e4[u] = u < nb ? eid[kb + u] : 0;
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) w4[u] = (u < nb && att) ? att[e4[u]] : 1.f;
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                x4[u] = (u < nb && on) ? ld4(x + (size_t)j4[u] * H + c) : f4zero();      // L1/L2-hot: pass 1 just gathered these rows
                if (HAS_EE) ee4[u] = (u < nb && on) ? ld4(edge_emb + (size_t)e4[u] * H + c) : f4zero();
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                if (u >= nb) continue;
                const int k = kb + u, e = e4[u];
                const float w = w4[u];
                const float4 xj = x4[u];
                float da = 0.f;
                if (on) {
                    float4 dm;
                    dm.x = fmaf(Qj.x, w * xj.x, Pj.x) + (k == jmin.x ? gmn_j.x : 0.f) + (k == jmax.x ? gmx_j.x : 0.f);
                    dm.y = fmaf(Qj.y, w * xj.y, Pj.y) + (k == jmin.y ? gmn_j.y : 0.f) + (k == jmax.y ? gmx_j.y : 0.f);
                    dm.z = fmaf(Qj.z, w * xj.z, Pj.z) + (k == jmin.z ? gmn_j.z : 0.f) + (k == jmax.z ? gmx_j.z : 0.f);
                    dm.w = fmaf(Qj.w, w * xj.w, Pj.w) + (k == jmin.w ? gmn_j.w : 0.f) + (k == jmax.w ? gmx_j.w : 0.f);
                    st4(dmsg + (size_t)k * H + c, f4scale(w, dm));
                    da += f4dot(dm, xj);
                    da += t1 + w * t2 + (k == kmin_a ? tmn : 0.f) + (k == kmax_a ? tmx : 0.f);
                    if (HAS_EE) {
                        const float4 ee = ee4[u];
                        float4 de;
                        de.x = fmaf(Qe.x, w * ee.x, Pe.x) + (k == emin.x ? gmn_e.x : 0.f) + (k == emax.x ? gmx_e.x : 0.f);
                        de.y = fmaf(Qe.y, w * ee.y, Pe.y) + (k == emin.y ? gmn_e.y : 0.f) + (k == emax.y ? gmx_e.y : 0.f);
                        de.z = fmaf(Qe.z, w * ee.z, Pe.z) + (k == emin.z ? gmn_e.z : 0.f) + (k == emax.z ? gmx_e.z : 0.f);
                        de.w = fmaf(Qe.w, w * ee.w, Pe.w) + (k == emin.w ? gmn_e.w : 0.f) + (k == emax.w ? gmx_e.w : 0.f);
                        if (dedge) st4(dedge + (size_t)e * H + c, f4scale(w, de));
                        da += f4dot(de, ee);
                    }
                }
                if (datt) {
                    da = group_sum<LPR>(da);
                    if (lane == 0) datt[e] = c0 ? datt[e] + da : da;
                }
            }
        }
    }
}

// ---- tiled backward: one workgroup owns a window of destination rows and their in-edges -------------------------------
// The two-pass backward above writes every per-edge gradient row to HBM (dmsg [E,H]) and reads it back in the per-source sum:
// 8*E*H bytes of traffic for nothing but a transpose.  Batched graphs are block-diagonal and molecule-sized, so a window of
// consecutive rows (aligned to graph starts where a graph start is near, see k_pna_tiles) contains both endpoints of nearly
// all of its edges.  One workgroup per window, three dependent memory round trips in all:
//   trip 1  the window descriptor (row / slot bounds of both CSRs)
//   trip 2  the window's indices -> LDS, and each lane group's first upstream gradient row -> registers
//   trip 3  the gathered x_j row of every edge -> LDS by LDS-DMA with a per-lane source address, and the attention of every edge
//   rows    a lane group per destination row: statistics and first-occurrence args from LDS; as soon as the fold has consumed the
//           gradient row the next one is requested into the same registers; the per-edge gradient row is written IN PLACE over
//           the gathered row in LDS; dx_self -> dx
//   sums    a lane group per source row: dx[j] += its edges' rows in by-source slot order, indices and rows straight from LDS
// Edges whose source lies outside the window (or beyond the LDS edge capacity: hub rows) spill their row to dmsg and are
// added by k_pna_bwd_spill afterwards, in by-source slot order as well, so the result is bitwise reproducible.
constexpr int TILE_BLOCK = 512;

// desc[t] = (first row of window t, rowptr_dst[row], rowptr_src[row], 0); desc[T] closes the last window
__global__ void k_pna_tiles(const int32_t* __restrict__ node_ptr, const int32_t* __restrict__ node_seg, const int32_t* __restrict__ rowptr,
                            const int32_t* __restrict__ rowptr_src, int N, int TN, int slack, int T, int4* __restrict__ desc,
                            int32_t* __restrict__ spill_count) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t == 0 && spill_count) *spill_count = 0;          // the list is filled by the NEXT launch (k_pna_spill_rows): no separate zeroing launch
    if (t > T) return;
    const int64_t p = (int64_t)t * TN;
    int start = N;
    if (t < T && p < N) {
        start = (int)p;
        if (node_seg) {             // pull the window start back to the start of the graph that contains row p, at most `slack` rows
            const int g0 = node_ptr[node_seg[p]];
            start = min((int)p, max(g0, (int)p - slack));      // never past p, whatever node_ptr holds for an unsorted batch vector: windows stay <= RCAP rows
        }
    }
    desc[t] = make_int4(start, rowptr[start], rowptr_src[start], 0);
}

// NATT: `att` is the NODE attention [N] and the edge weight att[row] * att[source] (no eid, no [E] attention tensor); `datt` is then
// d node_att [N]: the destination's share of an edge (d w * att[source]) is summed in the row's own loop, the source's share
// (d w * att[row]) is kept per slot in LDS (in the space of the unused edge ids) and summed in the per-source pass next to the rows;
// spilled edges leave theirs in dw[slot] for k_pna_bwd_spill.
template <int LPR, int NAGG, bool NATT>
__global__ __launch_bounds__(TILE_BLOCK, 4) void k_pna_bwd_tile(
    const float* __restrict__ x, const float* __restrict__ att, const float* __restrict__ dout, const int32_t* __restrict__ rowptr,
    const int32_t* __restrict__ col, const int32_t* __restrict__ eid, const int4* __restrict__ desc,
    const int32_t* __restrict__ rowptr_src, const int32_t* __restrict__ slot_map, int H, int TE, int RCAP, float* dx,
    float* __restrict__ dmsg, float* datt, float* __restrict__ dw, const float* __restrict__ dx_add, int datt_acc) {
    constexpr int GPB = LaneGroups<LPR, TILE_BLOCK>::GPB;     // lane groups (rows in flight) per workgroup
    constexpr int RPW = 64 / LPR;                // rows covered by one wave-instruction
    constexpr int NSEG = NAGG * 2;
    extern __shared__ float4 smem4[];
    float4* erow = smem4;                                        // [TE][LPR] gathered x_j row, then the edge's gradient row
    int* s_col = reinterpret_cast<int*>(erow + (size_t)TE * LPR);
    int* s_eid = s_col + TE;
    float* s_w = reinterpret_cast<float*>(s_eid);                // NATT: [TE] the edge weight att[row] * att[source]; once its row is done, the source's share of d node_att
    float* s_att = reinterpret_cast<float*>(s_eid + TE);
    int* s_slot = reinterpret_cast<int*>(s_att + TE);            // [TE] by-source slots of the window's sources -> window-local by-destination slot
    int* s_rp = s_slot + TE;                                     // [RCAP+1] by-destination row pointers relative to the window's first slot
    int* s_rps = s_rp + RCAP + 1;                                // [RCAP+1] by-source row pointers relative to the window's first slot
    float* s_na = reinterpret_cast<float*>(s_rps + RCAP + 1);    // [RCAP+1] NATT: node attention of the window's rows
    float* s_dna = s_na + RCAP + 1;                              // [RCAP+1] NATT: the rows' share (as destinations) of d node_att
    const int tid = threadIdx.x, wave = tid >> 6;
    const LaneGroups<LPR, TILE_BLOCK> lg;
    const int grp = lg.grp < 0 ? GPB : lg.grp;          // idle lanes (LPR = 20: lanes 60-63) behave like a group beyond every row
    const int lane = lg.lane, c = lane * 4;
    const bool on = c < H && lg.grp >= 0;
    const int t = pna_xcd_remap(blockIdx.x, gridDim.x);
    const int4 d0 = desc[t], d1 = desc[t + 1];
    const int n0 = d0.x, n1 = d1.x;
    if (n1 <= n0) return;
    const int nr = n1 - n0;                                      // <= RCAP by construction of the windows
    const int k0 = d0.y, ne = min(d1.y - k0, TE);
    const int s0 = d0.z, ns = d1.z - s0, nsl = min(ns, TE);
    // ---- trip 2 ------------------------------------------------------------------------------------------
    float4 dcur[NSEG];
    float4 xi_cur = f4zero();
    const size_t out_stride = (size_t)NSEG * H;
    int row = lg.grp >= 0 ? n0 + grp : n1;               // idle lanes own no row
#pragma unroll
    for (int sg = 0; sg < NSEG; ++sg) dcur[sg] = f4zero();
    if (row < n1 && on) {
        const float* g0 = dout + (size_t)row * out_stride + c;
#pragma unroll
        for (int sg = 0; sg < NSEG; ++sg) dcur[sg] = ld4(g0 + (size_t)sg * H);
        xi_cur = ld4(x + (size_t)row * H + c);
    }
    for (int i = tid; i <= nr; i += TILE_BLOCK) {
        s_rp[i] = rowptr[n0 + i] - k0;
        s_rps[i] = rowptr_src[n0 + i] - s0;
        if (NATT) { s_na[i] = att[min(n0 + i, n1 - 1)]; s_dna[i] = 0.f; }
    }
    for (int i = tid; i < ne; i += TILE_BLOCK) {
        s_col[i] = col[k0 + i];
        if (!NATT) s_eid[i] = eid[k0 + i];
    }
    for (int i = tid; i < nsl; i += TILE_BLOCK) s_slot[i] = slot_map[s0 + i] - k0;
    __syncthreads();
    // ---- trip 3 ------------------------------------------------------------------------------------------
    for (int base = 0; base < ne; base += GPB) {
        const int i = base + grp;
        if (i < ne && on)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(x + (size_t)s_col[i] * H + c),
                                             (__attribute__((address_space(3))) void*)(erow + (size_t)(base + wave * RPW) * LPR), 16, 0, 0);
    }
    for (int i = tid; i < ne; i += TILE_BLOCK) {
        if (NATT) {              // att[source] for the gradient, the edge weight for everything else: the slot's row by bisection of the row pointers
            int lo = 0, hi = nr;
            while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (s_rp[mid] <= i) lo = mid; else hi = mid; }
            const float a = att[s_col[i]];
            s_att[i] = a;
            s_w[i] = a * s_na[lo];
        } else {
            s_att[i] = att ? att[s_eid[i]] : 1.f;
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    // ---- destination rows ----------------------------------------------------------------------------------
    for (; row < n1; row += GPB) {
        const int nrow = row + GPB;
        const int beg = s_rp[row - n0], end = s_rp[row - n0 + 1];
        const float cnt = (float)(end - beg);
        // pass 1: statistics and first-occurrence args (empty rows fall through with zero statistics)
        float sa = 0.f, sa2 = 0.f, amin = INFINITY, amax = -INFINITY;
        int kmin_a = beg, kmax_a = beg;
        Acc4 aj;
        aj.init();
        int4 jmin = make_int4(beg, beg, beg, beg), jmax = jmin;
        float wfirst = 1.f;
        for (int k = beg; k < end; ++k) {
            float w;
            float4 xv;
            if (k < ne) {
                w = NATT ? s_w[k] : s_att[k];
                xv = on ? erow[(size_t)k * LPR + lane] : f4zero();
            } else {                                             // beyond the LDS edge capacity (hub rows): straight from memory
                w = NATT ? att[col[k0 + k]] * s_na[row - n0] : (att ? att[eid[k0 + k]] : 1.f);
                xv = on ? ld4(x + (size_t)col[k0 + k] * H + c) : f4zero();
            }
            if (k == beg) wfirst = w;
            sa += w; sa2 = fmaf(w, w, sa2);
            if (w < amin) { amin = w; kmin_a = k; }
            if (w > amax) { amax = w; kmax_a = k; }
            const float4 m = f4scale(w, xv);
            if (m.x < aj.mn.x) jmin.x = k; if (m.y < aj.mn.y) jmin.y = k; if (m.z < aj.mn.z) jmin.z = k; if (m.w < aj.mn.w) jmin.w = k;
            if (m.x > aj.mx.x) jmax.x = k; if (m.y > aj.mx.y) jmax.y = k; if (m.z > aj.mx.z) jmax.z = k; if (m.w > aj.mx.w) jmax.w = k;
            aj.add(m);
        }
        const float4 xi = xi_cur;
        const Acc4 ai = self_stats(xi, sa, sa2, amin, amax);
        const float n = fmaxf(cnt, 1.f), inv_n = 1.f / n;
        float t1 = 0.f, t2 = 0.f, tmn = 0.f, tmx = 0.f;
        float4 Pi, Qi, Pj, Qj, gmn_i, gmx_i, gmn_j, gmx_j;
#define GSAT_TFOLD(ACC, PART, P, Q, GMN, GMX)                                                                              \
        {                                                                                                                  \
            const float4 mean = f4scale(inv_n, ACC.s);                                                                     \
            const float4 var = make_float4(ACC.q.x * inv_n - mean.x * mean.x, ACC.q.y * inv_n - mean.y * mean.y,           \
                                           ACC.q.z * inv_n - mean.z * mean.z, ACC.q.w * inv_n - mean.w * mean.w);          \
            const float4 hs = make_float4(var.x > 0.f ? 0.5f * __builtin_amdgcn_rsqf(var.x + 1e-5f) : 0.f,                 \
                                          var.y > 0.f ? 0.5f * __builtin_amdgcn_rsqf(var.y + 1e-5f) : 0.f,                 \
                                          var.z > 0.f ? 0.5f * __builtin_amdgcn_rsqf(var.z + 1e-5f) : 0.f,                 \
                                          var.w > 0.f ? 0.5f * __builtin_amdgcn_rsqf(var.w + 1e-5f) : 0.f);                \
            float4 p0 = f4scale(inv_n, dcur[0 * 2 + (PART)]);                                                              \
            if (NAGG == 5) { const float4 vsum = dcur[(NAGG == 5 ? 4 : 0) * 2 + (PART)];                                   \
                             p0 = make_float4(p0.x + vsum.x, p0.y + vsum.y, p0.z + vsum.z, p0.w + vsum.w); }               \
            GMN = dcur[1 * 2 + (PART)];                                                                                    \
            GMX = dcur[2 * 2 + (PART)];                                                                                    \
            const float4 vs = dcur[3 * 2 + (PART)];                                                                        \
            const float4 gv = make_float4(vs.x * hs.x, vs.y * hs.y, vs.z * hs.z, vs.w * hs.w);                             \
            Q = f4scale(2.f * inv_n, gv);                                                                                  \
            P = make_float4(p0.x - mean.x * Q.x, p0.y - mean.y * Q.y, p0.z - mean.z * Q.z, p0.w - mean.w * Q.w);           \
        }
        GSAT_TFOLD(ai, 0, Pi, Qi, gmn_i, gmx_i)
        GSAT_TFOLD(aj, 1, Pj, Qj, gmn_j, gmx_j)
#undef GSAT_TFOLD
        // the gradient row is consumed: request the next one into the same registers, in flight under the rest of this row
        if (nrow < n1 && on) {
            const float* g0 = dout + (size_t)nrow * out_stride + c;
#pragma unroll
            for (int sg = 0; sg < NSEG; ++sg) dcur[sg] = ld4(g0 + (size_t)sg * H);
            xi_cur = ld4(x + (size_t)nrow * H + c);
        }
        if (on) {      // x_i third in closed form (see k_pna_bwd_dst); an empty row gets exact zeros (P = Q = 0 * finite, sums 0)
            float4 dxi;
#define GSAT_TSELF(C)                                                                                                      \
            {                                                                                                              \
                const float xc = xi.C;                                                                                     \
                const float wlo = xc > 0.f ? amin : (xc < 0.f ? amax : wfirst), whi = xc > 0.f ? amax : (xc < 0.f ? amin : wfirst); \
                dxi.C = Pi.C * sa + Qi.C * xc * sa2 + gmn_i.C * wlo + gmx_i.C * whi;                                       \
                t1 = fmaf(Pi.C, xc, t1);                                                                                   \
                t2 = fmaf(Qi.C * xc, xc, t2);                                                                              \
                tmn += xc > 0.f ? gmn_i.C * xc : (xc < 0.f ? gmx_i.C * xc : 0.f);                                          \
                tmx += xc > 0.f ? gmx_i.C * xc : (xc < 0.f ? gmn_i.C * xc : 0.f);                                          \
            }
            GSAT_TSELF(x) GSAT_TSELF(y) GSAT_TSELF(z) GSAT_TSELF(w)
#undef GSAT_TSELF
            if (end == beg) dxi = f4zero();                      // amin/amax are +-inf on an empty row: no in-edges, no gradient
            st4(dx + (size_t)row * H + c, dxi);
        }
        // pass 2: per-edge gradient rows, in place over the gathered rows
        for (int k = beg; k < end; ++k) {
            float w;
            float4 xj;
            int e = 0, j;
            if (k < ne) {
                w = NATT ? s_w[k] : s_att[k]; j = s_col[k];
                if (!NATT) e = s_eid[k];
                xj = on ? erow[(size_t)k * LPR + lane] : f4zero();
            } else {
                j = col[k0 + k];
                if (!NATT) e = eid[k0 + k];
                w = NATT ? att[j] * s_na[row - n0] : (att ? att[e] : 1.f);
                xj = on ? ld4(x + (size_t)j * H + c) : f4zero();
            }
            const bool in_lds = k < ne && j >= n0 && j < n1;
            float da = 0.f;
            if (on) {
                float4 dm;
                dm.x = fmaf(Qj.x, w * xj.x, Pj.x) + (k == jmin.x ? gmn_j.x : 0.f) + (k == jmax.x ? gmx_j.x : 0.f);
                dm.y = fmaf(Qj.y, w * xj.y, Pj.y) + (k == jmin.y ? gmn_j.y : 0.f) + (k == jmax.y ? gmx_j.y : 0.f);
                dm.z = fmaf(Qj.z, w * xj.z, Pj.z) + (k == jmin.z ? gmn_j.z : 0.f) + (k == jmax.z ? gmx_j.z : 0.f);
                dm.w = fmaf(Qj.w, w * xj.w, Pj.w) + (k == jmin.w ? gmn_j.w : 0.f) + (k == jmax.w ? gmx_j.w : 0.f);
                const float4 o = f4scale(w, dm);
                if (in_lds) erow[(size_t)k * LPR + lane] = o; else st4(dmsg + (size_t)(k0 + k) * H + c, o);
                da = f4dot(dm, xj) + t1 + w * t2 + (k == kmin_a ? tmn : 0.f) + (k == kmax_a ? tmx : 0.f);
            }
            if (datt) {
                da = group_sum<LPR>(da);
                if (NATT) {
                    if (lane == 0) {
                        s_dna[row - n0] = fmaf(da, k < ne ? s_att[k] : att[j], s_dna[row - n0]);      // destination share: d w * att[source]
                        const float share = da * s_na[row - n0];                                      // source share: d w * att[row]
                        if (in_lds) s_w[k] = share; else dw[k0 + k] = share;                          // (the slot's weight is not read again)
                    }
                } else if (lane == 0) datt[e] = da;
            }
        }
    }
    __syncthreads();
    // ---- per-source sums over the rows kept in LDS (same lane group <-> row map as above: dx[j] is this thread's own store) ----
    for (int j = lg.grp >= 0 ? n0 + grp : n1; j < n1; j += GPB) {
        if (!on) continue;
        const int sb = s_rps[j - n0], se = s_rps[j - n0 + 1];
        float4 acc = ld4(dx + (size_t)j * H + c);
        // NATT, datt_acc: d node_att is shared by every layer that used the attention: this launch adds its share to what is there
        const float dprev = (NATT && datt && datt_acc && lane == 0) ? datt[j] : 0.f;
        if (dx_add) {              // a gradient that reaches x by another path (the layer's residual): added here instead of by a separate kernel
            const float4 r = ld4(dx_add + (size_t)j * H + c);
            acc.x += r.x; acc.y += r.y; acc.z += r.z; acc.w += r.w;
        }
        float sw = 0.f;
        for (int s_ = sb; s_ < se; ++s_) {
            const int kl = s_ < nsl ? s_slot[s_] : slot_map[s0 + s_] - k0;
            if (kl >= 0 && kl < ne) {
                const float4 v = erow[(size_t)kl * LPR + lane];
                acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
                if (NATT) sw += s_w[kl];
            }
        }
        st4(dx + (size_t)j * H + c, acc);
        if (NATT && datt && lane == 0) datt[j] = (s_dna[j - n0] + sw) + dprev;      // destination share (this lane's own LDS store of the row loop) + source share
    }
}

// Which sources have an edge that k_pna_bwd_tile will spill is a property of the windows and the two CSRs alone, so it is found
// ONCE per batch (gsat_pna_build_tiles): spill_rows[0 .. spill_count) lists those sources (order irrelevant: each is independent).
// An edge stays in LDS iff its by-destination slot lies among the first TE slots of its SOURCE's window.
__global__ void k_pna_spill_rows(const int4* __restrict__ desc, int TN, int TE, const int32_t* __restrict__ rowptr_src,
                                 const int32_t* __restrict__ slot_map, int num_rows, int32_t* __restrict__ spill_rows,
                                 int32_t* __restrict__ spill_count) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= num_rows) return;
    const int sb = rowptr_src[j], se = rowptr_src[j + 1];
    if (sb == se) return;
    int t = j / TN;                                   // window starts are pulled back by at most TN rows: j is in window t or t + 1
    if (j >= desc[t + 1].x) ++t;
    const int k0 = desc[t].y, ne = min(desc[t + 1].y - k0, TE);
    bool flag = false;
    for (int s_ = sb; s_ < se && !flag; ++s_) {
        const int k = slot_map[s_];
        flag = k < k0 || k >= k0 + ne;
    }
    if (flag) {
        const int idx = atomicAdd(spill_count, 1);
        if (idx < num_rows) spill_rows[idx] = j;            // cannot overflow as long as the counter starts at 0; stay in bounds regardless.  The counter keeps
                                                            // counting, so *spill_count > num_rows IS the overflow record (tests assert it never happens)
    }
}

// dx[j] += the rows of source j that k_pna_bwd_tile spilled to dmsg, in by-source slot order; one lane group per listed source
template <int LPR>
__global__ __launch_bounds__(256) void k_pna_bwd_spill(const float* __restrict__ dmsg, const int4* __restrict__ desc, int TN, int TE,
                                                       const int32_t* __restrict__ rowptr_src, const int32_t* __restrict__ slot_map,
                                                       const int32_t* __restrict__ spill_rows, const int32_t* __restrict__ spill_count,
                                                       int num_rows, int H, float* __restrict__ dx, const float* __restrict__ dw,
                                                       float* __restrict__ dna) {
    const LaneGroups<LPR, 256> lg;
    constexpr int GPB = LaneGroups<LPR, 256>::GPB;
    const int lane = lg.lane, c = lane * 4;
    const int count = min(*spill_count, num_rows);
    if (lg.grp < 0) return;
    for (int i = blockIdx.x * GPB + lg.grp; i < count; i += gridDim.x * GPB) {
        const int j = spill_rows[i];
        const int sb = rowptr_src[j], se = rowptr_src[j + 1];
        int t = j / TN;
        if (j >= desc[t + 1].x) ++t;
        const int k0 = desc[t].y, ne = min(desc[t + 1].y - k0, TE);
        if (c >= H) continue;
        float4 acc = ld4(dx + (size_t)j * H + c);
        float sw = 0.f;
        for (int s_ = sb; s_ < se; ++s_) {
            const int k = slot_map[s_];
            if (k >= k0 && k < k0 + ne) continue;
            const float4 v = ld4(dmsg + (size_t)k * H + c);
            acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
            if (dna) sw += dw[k];
        }
        st4(dx + (size_t)j * H + c, acc);
        if (dna && lane == 0) dna[j] += sw;                       // node attention: the source's share of its spilled edges
    }
}

static inline int pna_lpr(int64_t H) {
    if (H <= 0 || H % 4 != 0 || H > 256) return 0;
    int64_t q = H / 4;
    // 80 channels (the hidden size of every PNA YAML of the reference): 20-lane groups, three rows per wave instead of two
    return q <= 4 ? 4 : q <= 8 ? 8 : q <= 16 ? 16 : q <= 20 ? 20 : q <= 32 ? 32 : 64;
}

constexpr int PNA_CHUNK = 256;       // channels per launch of the row kernels (64 lanes x float4)
static inline bool pna_width_ok(int64_t H) { return H > 0 && H % 4 == 0 && H <= 512; }

static inline void pna_grid(int64_t N, int lpr, int* nb, int* rpg) {
    const int gpb = (PNA_BLOCK / 64) * (64 / lpr);
    int64_t b = ceil_div(N, gpb);
    if (b > 256 * 64) b = 256 * 64;          // one row per lane group up to 16k blocks: shorter per-wave dependency chains
    if (b < 1) b = 1;
    *nb = (int)b;
    *rpg = (int)std::max<int64_t>(1, ceil_div(N, b * gpb));
}

// 4 / 5 for the aggregator lists (mean,min,max,std) / (mean,min,max,std,sum) with the identity scaler, else 0
static inline int fixed_aggregators(const PnaCfg& cfg) {
    if (cfg.S != 1 || cfg.scal[0] != 0 || (cfg.A != 4 && cfg.A != 5)) return 0;
    if (cfg.aggr[0] != AGG_MEAN || cfg.aggr[1] != AGG_MIN || cfg.aggr[2] != AGG_MAX || cfg.aggr[3] != GSAT_AGG_STD) return 0;
    if (cfg.A == 5 && cfg.aggr[4] != AGG_SUM) return 0;
    return cfg.A;
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

template <int L, int NA, bool NATT>
static hipError_t pna_tile_allow_lds(size_t lds) {       // raise the kernel's dynamic-LDS limit once per size (not a stream operation)
    static size_t allowed = 0;
    if (lds <= allowed) return hipSuccess;
    hipError_t e = hipFuncSetAttribute((const void*)k_pna_bwd_tile<L, NA, NATT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e == hipSuccess) allowed = lds;
    return e;
}

static size_t pna_tile_lds_bytes(int lpr, int edges_cap, int rows_cap) {
    return (size_t)edges_cap * ((size_t)lpr * 16 + 16) + (size_t)(rows_cap + 1) * 16;      // rows + 4 per-slot words, 2 row pointers + node attention and its gradient per row
}

}  // namespace gsat

using namespace gsat;

#define GSAT_LPR_DISPATCH(lpr, CALL) \
    switch (lpr) { case 4: CALL(4); break; case 8: CALL(8); break; case 16: CALL(16); break; case 20: CALL(20); break; case 32: CALL(32); break; default: CALL(64); break; }

extern "C" {

static int pna_fwd_impl(const char* who, const float* x, const float* att, const float* edge_emb, const int32_t* rowptr, const int32_t* col,
                        const int32_t* eid, int64_t N, int64_t E, int64_t H, const int32_t* aggregators, int A, const int32_t* scalers, int S,
                        float avg_deg_lin, float avg_deg_log, float* out, const int32_t* chunk_ptr, float* partial, hipStream_t stream,
                        float* scal_out = nullptr) {
    GSAT_REQUIRE(N >= 0 && N < (1ll << 31), GSAT_ERR_ARG, "%s: bad N", who);
    PnaCfg cfg;
    int rc = make_cfg(aggregators, A, scalers, S, avg_deg_lin, avg_deg_log, &cfg);
    if (rc) return rc;
    if (N == 0) return GSAT_OK;
    GSAT_REQUIRE(pna_width_ok(H), GSAT_ERR_UNSUPPORTED, "%s: H=%lld must be a multiple of 4 and <= 512", who, (long long)H);
    GSAT_REQUIRE(x && rowptr && out, GSAT_ERR_ARG, "%s: null pointer", who);   /* col / eid may be NULL when E == 0 */
    GSAT_REQUIRE(chunk_ptr == nullptr || partial != nullptr, GSAT_ERR_ARG, "%s: chunk_ptr needs the record workspace", who);
    if (E <= CH) chunk_ptr = nullptr;                      // no row can be long
    const int nagg = fixed_aggregators(cfg);
    GSAT_REQUIRE(!scal_out || (nagg && !edge_emb && !chunk_ptr), GSAT_ERR_UNSUPPORTED,
                 "%s: the compact output needs (mean,min,max,std[,sum]) with the identity scaler, no edge features, no hub chunks", who);
    for (int c0 = 0; c0 < (int)H; c0 += PNA_CHUNK) {
        const int Hc = std::min<int>(PNA_CHUNK, (int)H - c0);
        const int lpr = pna_lpr(Hc);
        int nb, rpg;
        pna_grid(N, lpr, &nb, &rpg);
        const int gpb = (PNA_BLOCK / 64) * (64 / lpr);
        const int cb = chunk_ptr ? (int)std::min<int64_t>(ceil_div(pna_max_chunks(E), gpb), 256 * 16) : 0;
#define GO(L, EE, NA)                                                                                                                       \
    do {                                                                                                                                    \
        k_pna_fwd<L, EE, NA, false><<<nb, PNA_BLOCK, 0, stream>>>(x, att, edge_emb, rowptr, col, eid, (int)N, (int)H, cfg, out, rpg, c0, Hc, chunk_ptr, partial, scal_out); \
        if (cb) k_pna_fwd<L, EE, NA, true><<<cb, PNA_BLOCK, 0, stream>>>(x, att, edge_emb, rowptr, col, eid, (int)N, (int)H, cfg, out, rpg, c0, Hc, chunk_ptr, partial, scal_out); \
    } while (0)
#define CALL(L)                                                                                                              \
    do {                                                                                                                     \
        if (edge_emb) {                                                                                                      \
            if (cb) k_pna_chunk_stats<L, true><<<cb, PNA_BLOCK, 0, stream>>>(x, att, edge_emb, rowptr, col, eid, (int)N, (int)H, chunk_ptr, partial, c0, Hc); \
            if (nagg == 4) GO(L, true, 4); else if (nagg == 5) GO(L, true, 5); else GO(L, true, 0);                          \
        } else {                                                                                                             \
            if (cb) k_pna_chunk_stats<L, false><<<cb, PNA_BLOCK, 0, stream>>>(x, att, edge_emb, rowptr, col, eid, (int)N, (int)H, chunk_ptr, partial, c0, Hc); \
            if (nagg == 4) GO(L, false, 4); else if (nagg == 5) GO(L, false, 5); else GO(L, false, 0);                       \
        }                                                                                                                    \
    } while (0)
        GSAT_LPR_DISPATCH(lpr, CALL);
#undef CALL
#undef GO
        GSAT_LAUNCH_CHECK();
    }
    return GSAT_OK;
}

int gsat_pna_fwd(const float* x, const float* att, const float* edge_emb, const int32_t* rowptr, const int32_t* col,
                 const int32_t* eid, int64_t N, int64_t H, const int32_t* aggregators, int A, const int32_t* scalers, int S,
                 float avg_deg_lin, float avg_deg_log, float* out, void* stream_) {
    return pna_fwd_impl("gsat_pna_fwd", x, att, edge_emb, rowptr, col, eid, N, 0, H, aggregators, A, scalers, S, avg_deg_lin, avg_deg_log, out,
                        nullptr, nullptr, (hipStream_t)stream_);
}

int gsat_pna_fwd_node_att(const float* x, const float* node_att, const int32_t* rowptr, const int32_t* col, int64_t N, int64_t H,
                          const int32_t* aggregators, int A, const int32_t* scalers, int S, float avg_deg_lin, float avg_deg_log, float* out,
                          void* stream_) {
    GSAT_REQUIRE(node_att, GSAT_ERR_ARG, "gsat_pna_fwd_node_att: null node attention");
    return pna_fwd_impl("gsat_pna_fwd_node_att", x, node_att, nullptr, rowptr, col, /*eid=*/nullptr, N, 0, H, aggregators, A, scalers, S,
                        avg_deg_lin, avg_deg_log, out, nullptr, nullptr, (hipStream_t)stream_);
}

int gsat_pna_fwd_compact(const float* x, const float* att, const int32_t* rowptr, const int32_t* col, const int32_t* eid, int64_t N,
                         int64_t H, const int32_t* aggregators, int A, float* aggj, float* scal, void* stream_) {
    GSAT_REQUIRE(aggj && scal && ((uintptr_t)scal % 16 == 0), GSAT_ERR_ARG, "gsat_pna_fwd_compact: null / misaligned output");
    const int32_t ident = 0;
    return pna_fwd_impl("gsat_pna_fwd_compact", x, att, nullptr, rowptr, col, eid, N, 0, H, aggregators, A, &ident, 1, 1.f, 1.f, aggj, nullptr,
                        nullptr, (hipStream_t)stream_, scal);
}

size_t gsat_pna_long_row_floats(int64_t num_edges, int64_t H, int has_edge_emb) {
    return (size_t)pna_max_chunks(num_edges > 0 ? num_edges : 0) * pna_rec_floats((int)H, has_edge_emb ? 2 : 1);
}

int gsat_pna_fwd_long(const float* x, const float* att, const float* edge_emb, const int32_t* rowptr, const int32_t* col,
                      const int32_t* eid, int64_t N, int64_t E, int64_t H, const int32_t* aggregators, int A, const int32_t* scalers, int S,
                      float avg_deg_lin, float avg_deg_log, float* out, const int32_t* chunk_ptr, float* partial, void* stream_) {
    GSAT_REQUIRE(E >= 0 && E < (1ll << 31), GSAT_ERR_ARG, "gsat_pna_fwd_long: bad E");
    return pna_fwd_impl("gsat_pna_fwd_long", x, att, edge_emb, rowptr, col, eid, N, E, H, aggregators, A, scalers, S, avg_deg_lin, avg_deg_log,
                        out, chunk_ptr, partial, (hipStream_t)stream_);
}

static int pna_bwd_impl(const char* who, const float* x, const float* att, const float* edge_emb, const float* dout, const int32_t* rowptr,
                        const int32_t* col, const int32_t* eid, int64_t N, int64_t E, int64_t H, const int32_t* aggregators, int A,
                        const int32_t* scalers, int S, float avg_deg_lin, float avg_deg_log, float* dx_self, float* dmsg, float* datt,
                        float* dedge_emb, const int32_t* chunk_ptr, float* partial, hipStream_t stream) {
    GSAT_REQUIRE(N >= 0 && N < (1ll << 31), GSAT_ERR_ARG, "%s: bad N", who);
    PnaCfg cfg;
    int rc = make_cfg(aggregators, A, scalers, S, avg_deg_lin, avg_deg_log, &cfg);
    if (rc) return rc;
    if (N == 0) return GSAT_OK;
    GSAT_REQUIRE(pna_width_ok(H), GSAT_ERR_UNSUPPORTED, "%s: H=%lld must be a multiple of 4 and <= 512", who, (long long)H);
    GSAT_REQUIRE(x && dout && rowptr && dx_self, GSAT_ERR_ARG, "%s: null pointer", who);   /* col / eid / dmsg may be NULL when E == 0 */
    GSAT_REQUIRE(chunk_ptr == nullptr || partial != nullptr, GSAT_ERR_ARG, "%s: chunk_ptr needs the record workspace", who);
    if (E <= CH) chunk_ptr = nullptr;                      // no row can be long
    const int nagg = fixed_aggregators(cfg);
    for (int c0 = 0; c0 < (int)H; c0 += PNA_CHUNK) {
        const int Hc = std::min<int>(PNA_CHUNK, (int)H - c0);
        const int lpr = pna_lpr(Hc);
        int nb, rpg;
        pna_grid(N, lpr, &nb, &rpg);
        const int gpb = (PNA_BLOCK / 64) * (64 / lpr);
        const int cb = chunk_ptr ? (int)std::min<int64_t>(ceil_div(pna_max_chunks(E), gpb), 256 * 16) : 0;
#define GO(L, EE, NA)                                                                                                                       \
    do {                                                                                                                                    \
        k_pna_bwd_dst<L, EE, NA, false><<<nb, PNA_BLOCK, 0, stream>>>(x, att, edge_emb, dout, rowptr, col, eid, (int)N, (int)H, cfg, dx_self, dmsg, datt, dedge_emb, rpg, c0, Hc, chunk_ptr, partial); \
        if (cb) k_pna_bwd_dst<L, EE, NA, true><<<cb, PNA_BLOCK, 0, stream>>>(x, att, edge_emb, dout, rowptr, col, eid, (int)N, (int)H, cfg, dx_self, dmsg, datt, dedge_emb, rpg, c0, Hc, chunk_ptr, partial); \
    } while (0)
#define CALL(L)                                                                                                              \
    do {                                                                                                                     \
        if (edge_emb) {                                                                                                      \
            if (cb) k_pna_chunk_stats<L, true><<<cb, PNA_BLOCK, 0, stream>>>(x, att, edge_emb, rowptr, col, eid, (int)N, (int)H, chunk_ptr, partial, c0, Hc); \
            if (nagg == 4) GO(L, true, 4); else if (nagg == 5) GO(L, true, 5); else GO(L, true, 0);                          \
            if (cb) k_pna_chunk_apply<L, true><<<cb, PNA_BLOCK, 0, stream>>>(x, att, edge_emb, rowptr, col, eid, (int)N, (int)H, chunk_ptr, partial, dmsg, datt, dedge_emb, c0, Hc); \
        } else {                                                                                                             \
            if (cb) k_pna_chunk_stats<L, false><<<cb, PNA_BLOCK, 0, stream>>>(x, att, edge_emb, rowptr, col, eid, (int)N, (int)H, chunk_ptr, partial, c0, Hc); \
            if (nagg == 4) GO(L, false, 4); else if (nagg == 5) GO(L, false, 5); else GO(L, false, 0);                       \
            if (cb) k_pna_chunk_apply<L, false><<<cb, PNA_BLOCK, 0, stream>>>(x, att, edge_emb, rowptr, col, eid, (int)N, (int)H, chunk_ptr, partial, dmsg, datt, dedge_emb, c0, Hc); \
        }                                                                                                                    \
    } while (0)
        GSAT_LPR_DISPATCH(lpr, CALL);
#undef CALL
#undef GO
        GSAT_LAUNCH_CHECK();
    }
    return GSAT_OK;
}

int gsat_pna_bwd(const float* x, const float* att, const float* edge_emb, const float* dout, const int32_t* rowptr,
                 const int32_t* col, const int32_t* eid, int64_t N, int64_t H, const int32_t* aggregators, int A,
                 const int32_t* scalers, int S, float avg_deg_lin, float avg_deg_log, float* dx_self, float* dmsg,
                 float* datt, float* dedge_emb, void* stream_) {
    return pna_bwd_impl("gsat_pna_bwd", x, att, edge_emb, dout, rowptr, col, eid, N, 0, H, aggregators, A, scalers, S, avg_deg_lin, avg_deg_log,
                        dx_self, dmsg, datt, dedge_emb, nullptr, nullptr, (hipStream_t)stream_);
}

int gsat_pna_bwd_long(const float* x, const float* att, const float* edge_emb, const float* dout, const int32_t* rowptr,
                      const int32_t* col, const int32_t* eid, int64_t N, int64_t E, int64_t H, const int32_t* aggregators, int A,
                      const int32_t* scalers, int S, float avg_deg_lin, float avg_deg_log, float* dx_self, float* dmsg,
                      float* datt, float* dedge_emb, const int32_t* chunk_ptr, float* partial, void* stream_) {
    GSAT_REQUIRE(E >= 0 && E < (1ll << 31), GSAT_ERR_ARG, "gsat_pna_bwd_long: bad E");
    return pna_bwd_impl("gsat_pna_bwd_long", x, att, edge_emb, dout, rowptr, col, eid, N, E, H, aggregators, A, scalers, S, avg_deg_lin,
                        avg_deg_log, dx_self, dmsg, datt, dedge_emb, chunk_ptr, partial, (hipStream_t)stream_);
}

// ---- tiled backward (see k_pna_bwd_tile) ------------------------------------------------------------------------------
int gsat_pna_tile_plan(int64_t H, int64_t lds_budget_bytes, int32_t* rows_nominal, int32_t* rows_slack, int32_t* edges_cap) {
    const int lpr = pna_lpr(H);
    GSAT_REQUIRE(lpr, GSAT_ERR_UNSUPPORTED, "gsat_pna_tile_plan: H=%lld must be a multiple of 4 and <= 256", (long long)H);
    GSAT_REQUIRE(rows_nominal && rows_slack && edges_cap, GSAT_ERR_ARG, "gsat_pna_tile_plan: null pointer");
    if (lds_budget_bytes <= 0) lds_budget_bytes = 80 * 1024;      // one workgroup per CU: measured best at C3 (profiles/r02_pna_bwd_probe.txt)
    GSAT_REQUIRE(lds_budget_bytes <= 160 * 1024, GSAT_ERR_ARG, "gsat_pna_tile_plan: LDS budget above 160 KiB");
    int64_t te = (lds_budget_bytes - 64) / ((int64_t)lpr * 16 + 16 + 8);     // + 16 bytes per window row, rows <= te / 2 (pna_tile_lds_bytes)
    te = te / 8 * 8;
    GSAT_REQUIRE(te >= 16, GSAT_ERR_UNSUPPORTED, "gsat_pna_tile_plan: LDS budget too small for H=%lld", (long long)H);
    int64_t half = std::max<int64_t>(4, te / 4 / 4 * 4);        // window = [nominal, nominal + slack] rows, ~2 in-edges per row
    *rows_nominal = (int32_t)half;
    *rows_slack = (int32_t)half;
    *edges_cap = (int32_t)te;
    return GSAT_OK;
}

int gsat_pna_build_tiles(const int32_t* node_ptr, const int32_t* node_seg, const int32_t* rowptr, const int32_t* rowptr_src,
                         const int32_t* slot_dst_of_srcslot, int64_t N, int rows_nominal, int rows_slack, int edges_cap, int32_t* tile_desc,
                         int32_t* spill_rows, int32_t* spill_count, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    GSAT_REQUIRE(N >= 0 && N < (1ll << 31) && rows_nominal > 0 && rows_slack >= 0 && rows_slack <= rows_nominal && tile_desc && rowptr && rowptr_src,
                 GSAT_ERR_ARG, "gsat_pna_build_tiles: bad argument");
    GSAT_REQUIRE(((uintptr_t)tile_desc & 15) == 0, GSAT_ERR_ARG, "gsat_pna_build_tiles: tile_desc must be 16-byte aligned");
    GSAT_REQUIRE((node_ptr == nullptr) == (node_seg == nullptr), GSAT_ERR_ARG, "gsat_pna_build_tiles: node_ptr and node_seg go together");
    const int64_t T = ceil_div(N, rows_nominal);
    k_pna_tiles<<<(int)ceil_div(T + 1, 256), 256, 0, stream>>>(node_ptr, node_seg, rowptr, rowptr_src, (int)N, rows_nominal, node_seg ? rows_slack : 0, (int)T,
                                                               (int4*)tile_desc, spill_count);
    GSAT_LAUNCH_CHECK();
    GSAT_REQUIRE(spill_rows && spill_count && edges_cap > 0, GSAT_ERR_ARG, "gsat_pna_build_tiles: null spill list");
    if (N > 0 && slot_dst_of_srcslot) {
        k_pna_spill_rows<<<(int)ceil_div(N, 256), 256, 0, stream>>>((const int4*)tile_desc, rows_nominal, edges_cap, rowptr_src, slot_dst_of_srcslot,
                                                                   (int)N, spill_rows, spill_count);
        GSAT_LAUNCH_CHECK();
    }
    return GSAT_OK;
}

static int pna_bwd_tiled_impl(const char* who, bool natt, const float* x, const float* att, const float* dout, const int32_t* rowptr,
                              const int32_t* col, const int32_t* eid, const int32_t* tile_desc, int64_t num_tiles, int rows_nominal,
                              int rows_cap, int edges_cap, const int32_t* rowptr_src, const int32_t* slot_dst_of_srcslot, int64_t N,
                              int64_t E, int64_t H, const int32_t* aggregators, int A, const int32_t* scalers, int S,
                              const int32_t* spill_rows, const int32_t* spill_count, float* dx, float* dmsg, float* datt, float* dw,
                              const float* dx_add, int datt_acc, hipStream_t stream) {
    GSAT_REQUIRE(N >= 0 && N < (1ll << 31) && E >= 0 && num_tiles >= 0, GSAT_ERR_ARG, "%s: bad extents", who);
    PnaCfg cfg;
    int rc = make_cfg(aggregators, A, scalers, S, 1.f, 1.f, &cfg);
    if (rc) return rc;
    const int nagg = fixed_aggregators(cfg);
    GSAT_REQUIRE(nagg, GSAT_ERR_UNSUPPORTED, "%s: only (mean,min,max,std[,sum]) with the identity scaler", who);
    if (N == 0) return GSAT_OK;
    const int lpr = pna_lpr(H);
    GSAT_REQUIRE(lpr, GSAT_ERR_UNSUPPORTED, "%s: H=%lld must be a multiple of 4 and <= 256", who, (long long)H);
    GSAT_REQUIRE(x && dout && rowptr && tile_desc && rowptr_src && dx && rows_nominal > 0 && rows_cap >= rows_nominal && rows_cap <= 2 * rows_nominal && edges_cap > 0, GSAT_ERR_ARG, "%s: null pointer", who);
    GSAT_REQUIRE(E == 0 || (col && (natt || eid) && slot_dst_of_srcslot && dmsg), GSAT_ERR_ARG, "%s: null edge arrays", who);
    GSAT_REQUIRE(!natt || (att && (!datt || dw || E == 0)), GSAT_ERR_ARG, "%s: node attention needs node_att and, for its gradient, the [E] scratch dw", who);
    const size_t lds = pna_tile_lds_bytes(lpr, edges_cap, rows_cap);
    GSAT_REQUIRE(lds <= 160 * 1024, GSAT_ERR_ARG, "%s: tile needs %zu bytes of LDS", who, lds);
#define GO(L, NA, NT)                                                                                                        \
    do {                                                                                                                     \
        GSAT_CHECK_HIP((pna_tile_allow_lds<L, NA, NT>(lds)));                                                                \
        k_pna_bwd_tile<L, NA, NT><<<(int)num_tiles, TILE_BLOCK, lds, stream>>>(x, att, dout, rowptr, col, eid, (const int4*)tile_desc,     \
                                                                          rowptr_src, slot_dst_of_srcslot, (int)H, edges_cap, rows_cap, dx, dmsg, datt, dw, dx_add, datt_acc); \
    } while (0)
#define CALL(L) do { if (natt) { if (nagg == 4) GO(L, 4, true); else GO(L, 5, true); } else { if (nagg == 4) GO(L, 4, false); else GO(L, 5, false); } } while (0)
    if (num_tiles > 0) { GSAT_LPR_DISPATCH(lpr, CALL); }
#undef CALL
#undef GO
    GSAT_LAUNCH_CHECK();
    if (E > 0) {
        GSAT_REQUIRE(spill_rows && spill_count, GSAT_ERR_ARG, "%s: null spill list (gsat_pna_build_tiles)", who);
        const int nb = (int)std::min<int64_t>(std::max<int64_t>(ceil_div(N, 8 * 4 * (64 / lpr)), 1), 256 * 8);     // sized for ~1/8 of the sources
        const float* dw_in = (natt && datt) ? dw : nullptr;
        float* dna = (natt && datt) ? datt : nullptr;
#define CALL(L) k_pna_bwd_spill<L><<<nb, 256, 0, stream>>>(dmsg, (const int4*)tile_desc, rows_nominal, edges_cap, rowptr_src, slot_dst_of_srcslot, spill_rows, spill_count, (int)N, (int)H, dx, dw_in, dna)
        GSAT_LPR_DISPATCH(lpr, CALL);
#undef CALL
        GSAT_LAUNCH_CHECK();
    }
    return GSAT_OK;
}

int gsat_pna_bwd_tiled(const float* x, const float* att, const float* dout, const int32_t* rowptr, const int32_t* col,
                       const int32_t* eid, const int32_t* tile_desc, int64_t num_tiles, int rows_nominal, int rows_cap, int edges_cap,
                       const int32_t* rowptr_src, const int32_t* slot_dst_of_srcslot, int64_t N, int64_t E, int64_t H,
                       const int32_t* aggregators, int A, const int32_t* scalers, int S, const int32_t* spill_rows,
                       const int32_t* spill_count, float* dx, float* dmsg, float* datt, const float* dx_add, void* stream_) {
    return pna_bwd_tiled_impl("gsat_pna_bwd_tiled", false, x, att, dout, rowptr, col, eid, tile_desc, num_tiles, rows_nominal, rows_cap, edges_cap,
                              rowptr_src, slot_dst_of_srcslot, N, E, H, aggregators, A, scalers, S, spill_rows, spill_count, dx, dmsg, datt,
                              nullptr, dx_add, 0, (hipStream_t)stream_);
}

int gsat_pna_bwd_tiled_node_att(const float* x, const float* node_att, const float* dout, const int32_t* rowptr, const int32_t* col,
                                const int32_t* tile_desc, int64_t num_tiles, int rows_nominal, int rows_cap, int edges_cap,
                                const int32_t* rowptr_src, const int32_t* slot_dst_of_srcslot, int64_t N, int64_t E, int64_t H,
                                const int32_t* aggregators, int A, const int32_t* scalers, int S, const int32_t* spill_rows,
                                const int32_t* spill_count, float* dx, float* dmsg, float* dnode_att, float* dw, const float* dx_add,
                                int accumulate_dnode_att, void* stream_) {
    return pna_bwd_tiled_impl("gsat_pna_bwd_tiled_node_att", true, x, node_att, dout, rowptr, col, nullptr, tile_desc, num_tiles, rows_nominal,
                              rows_cap, edges_cap, rowptr_src, slot_dst_of_srcslot, N, E, H, aggregators, A, scalers, S, spill_rows, spill_count,
                              dx, dmsg, dnode_att, dw, dx_add, accumulate_dnode_att, (hipStream_t)stream_);
}

}  // extern "C"
