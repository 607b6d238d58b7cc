// Masked sum aggregation (GINConv / GINEConv message passing) and segment pools.
//
// HBM-bound gather kernels.  Mapping: a group of LPR lanes owns one destination row, each lane
// carries NV float4 column slices (H <= LPR*4*NV), so one row read is a single coalesced
// LPR*16-byte segment and 64/LPR rows are in flight per wavefront.  No message tensor [E,H] is ever
// materialised and there are no atomics: rows are summed in CSR order (edge-id order inside a row),
// so results are bitwise reproducible.  Blocks are remapped so that every XCD walks one
// contiguous span of rows (neighbouring rows share gathered lines in that XCD's L2).
#include "common.h"

namespace gsat {

constexpr int AGG_BLOCK = 256;
constexpr int CH = GSAT_LONG_ROW_EDGES;   // rows with more in-edges than this are split into CH-edge chunks

// Logical block id such that the blocks resident on one XCD (b % 8 equal) cover a contiguous range.
__device__ __forceinline__ int xcd_remap(int b, int nb) {
    int q = nb >> 3, r = nb & 7, x = b & 7, i = b >> 3;
    return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + i;
}

template <int LPR, int NV, bool GINE>
__global__ __launch_bounds__(AGG_BLOCK) void k_aggr_sum_fwd(
    const float* __restrict__ x, const float* __restrict__ self_rows, const float* __restrict__ att,
    const float* __restrict__ edge_emb, const int32_t* __restrict__ rowptr, const int32_t* __restrict__ col,
    const int32_t* __restrict__ eid, int num_rows, int H, float self_coef, float* __restrict__ out, int rows_per_group,
    const int32_t* __restrict__ chunk_ptr, const float* __restrict__ partial) {
    constexpr int GPB = AGG_BLOCK / LPR;
    const int lane = threadIdx.x % LPR;
    const int grp = xcd_remap(blockIdx.x, gridDim.x) * GPB + threadIdx.x / LPR;
    int row = grp * rows_per_group;
    const int row_end = min(num_rows, row + rows_per_group);
    for (; row < row_end; ++row) {
        const int beg = rowptr[row], end = rowptr[row + 1];
        float4 acc[NV];
#pragma unroll
        for (int v = 0; v < NV; ++v) {
            int c = (v * LPR + lane) * 4;
            acc[v] = f4zero();
            if (c < H && self_coef != 0.f) {
                float4 s = ld4(self_rows + (size_t)row * H + c);
                acc[v] = make_float4(self_coef * s.x, self_coef * s.y, self_coef * s.z, self_coef * s.w);
            }
        }
        if (chunk_ptr != nullptr && end - beg > CH) {      // long row: add the chunk partials in chunk order
            for (int p = chunk_ptr[row]; p < chunk_ptr[row + 1]; ++p) {
#pragma unroll
                for (int v = 0; v < NV; ++v) {
                    int c = (v * LPR + lane) * 4;
                    if (c < H) {
                        float4 t = ld4(partial + (size_t)p * H + c);
                        acc[v].x += t.x; acc[v].y += t.y; acc[v].z += t.z; acc[v].w += t.w;
                    }
                }
            }
#pragma unroll
            for (int v = 0; v < NV; ++v) {
                int c = (v * LPR + lane) * 4;
                if (c < H) st4(out + (size_t)row * H + c, acc[v]);
            }
            continue;
        }
        int k = beg;
        for (; k + 2 <= end; k += 2) {   // two gathered rows in flight per group
            int j0 = col[k], j1 = col[k + 1];
            int e0 = 0, e1 = 0;
            if (att != nullptr || GINE) { e0 = eid[k]; e1 = eid[k + 1]; }
            float w0 = att ? att[e0] : 1.f, w1 = att ? att[e1] : 1.f;
#pragma unroll
            for (int v = 0; v < NV; ++v) {
                int c = (v * LPR + lane) * 4;
                if (c < H) {
                    float4 a = ld4(x + (size_t)j0 * H + c);
                    float4 b = ld4(x + (size_t)j1 * H + c);
                    if (GINE) {
                        float4 ea = ld4(edge_emb + (size_t)e0 * H + c);
                        float4 eb = ld4(edge_emb + (size_t)e1 * H + c);
                        a = make_float4(fmaxf(a.x + ea.x, 0.f), fmaxf(a.y + ea.y, 0.f), fmaxf(a.z + ea.z, 0.f), fmaxf(a.w + ea.w, 0.f));
                        b = make_float4(fmaxf(b.x + eb.x, 0.f), fmaxf(b.y + eb.y, 0.f), fmaxf(b.z + eb.z, 0.f), fmaxf(b.w + eb.w, 0.f));
                    }
                    acc[v] = f4fma(w0, a, acc[v]);
                    acc[v] = f4fma(w1, b, acc[v]);
                }
            }
        }
        if (k < end) {
            int j0 = col[k];
            int e0 = (att != nullptr || GINE) ? eid[k] : 0;
            float w0 = att ? att[e0] : 1.f;
#pragma unroll
            for (int v = 0; v < NV; ++v) {
                int c = (v * LPR + lane) * 4;
                if (c < H) {
                    float4 a = ld4(x + (size_t)j0 * H + c);
                    if (GINE) {
                        float4 ea = ld4(edge_emb + (size_t)e0 * H + c);
                        a = make_float4(fmaxf(a.x + ea.x, 0.f), fmaxf(a.y + ea.y, 0.f), fmaxf(a.z + ea.z, 0.f), fmaxf(a.w + ea.w, 0.f));
                    }
                    acc[v] = f4fma(w0, a, acc[v]);
                }
            }
        }
#pragma unroll
        for (int v = 0; v < NV; ++v) {
            int c = (v * LPR + lane) * 4;
            if (c < H) st4(out + (size_t)row * H + c, acc[v]);
        }
    }
}

// Transposed pass: group owns SOURCE row j, walks its out-edges, gathers dout[dst].
template <int LPR, int NV, bool GINE>
__global__ __launch_bounds__(AGG_BLOCK) void k_aggr_sum_bwd(
    const float* __restrict__ x, const float* __restrict__ att, const float* __restrict__ edge_emb,
    const float* __restrict__ dout, const int32_t* __restrict__ rowptr, const int32_t* __restrict__ dstc,
    const int32_t* __restrict__ eid, int num_rows, int H, float self_coef, float* __restrict__ dx,
    float* __restrict__ datt, float* __restrict__ dedge, int rows_per_group, const int32_t* __restrict__ chunk_ptr,
    const float* __restrict__ partial) {
    constexpr int GPB = AGG_BLOCK / LPR;
    const int lane = threadIdx.x % LPR;
    const int grp = xcd_remap(blockIdx.x, gridDim.x) * GPB + threadIdx.x / LPR;
    int row = grp * rows_per_group;
    const int row_end = min(num_rows, row + rows_per_group);
    for (; row < row_end; ++row) {
        const int beg = rowptr[row], end = rowptr[row + 1];
        if (chunk_ptr != nullptr && end - beg > CH) {      // long row: per-edge outputs were written by the chunk kernel
#pragma unroll
            for (int v = 0; v < NV; ++v) {
                int c = (v * LPR + lane) * 4;
                if (c < H) {
                    float4 g = ld4(dout + (size_t)row * H + c);
                    float4 a = make_float4(self_coef * g.x, self_coef * g.y, self_coef * g.z, self_coef * g.w);
                    for (int p = chunk_ptr[row]; p < chunk_ptr[row + 1]; ++p) {
                        float4 t = ld4(partial + (size_t)p * H + c);
                        a.x += t.x; a.y += t.y; a.z += t.z; a.w += t.w;
                    }
                    st4(dx + (size_t)row * H + c, a);
                }
            }
            continue;
        }
        float4 acc[NV], xj[NV];
#pragma unroll
        for (int v = 0; v < NV; ++v) {
            int c = (v * LPR + lane) * 4;
            acc[v] = f4zero();
            xj[v] = f4zero();
            if (c < H) {
                float4 g = ld4(dout + (size_t)row * H + c);
                acc[v] = make_float4(self_coef * g.x, self_coef * g.y, self_coef * g.z, self_coef * g.w);
                xj[v] = ld4(x + (size_t)row * H + c);
            }
        }
        for (int k = beg; k < end; ++k) {
            const int i = dstc[k], e = eid[k];
            const float w = att ? att[e] : 1.f;
            float dot = 0.f;
#pragma unroll
            for (int v = 0; v < NV; ++v) {
                int c = (v * LPR + lane) * 4;
                if (c < H) {
                    float4 g = ld4(dout + (size_t)i * H + c);
                    if (GINE) {
                        float4 ea = ld4(edge_emb + (size_t)e * H + c);
                        float4 pre = make_float4(xj[v].x + ea.x, xj[v].y + ea.y, xj[v].z + ea.z, xj[v].w + ea.w);
                        float4 m = make_float4(fmaxf(pre.x, 0.f), fmaxf(pre.y, 0.f), fmaxf(pre.z, 0.f), fmaxf(pre.w, 0.f));
                        dot += f4dot(m, g);
                        g = make_float4(pre.x > 0.f ? g.x : 0.f, pre.y > 0.f ? g.y : 0.f, pre.z > 0.f ? g.z : 0.f, pre.w > 0.f ? g.w : 0.f);
                        if (dedge) st4(dedge + (size_t)e * H + c, make_float4(w * g.x, w * g.y, w * g.z, w * g.w));
                    } else {
                        dot += f4dot(xj[v], g);
                    }
                    acc[v] = f4fma(w, g, acc[v]);
                }
            }
            if (datt) {
                dot = group_sum<LPR>(dot);
                if (lane == 0) datt[e] = dot;
            }
        }
#pragma unroll
        for (int v = 0; v < NV; ++v) {
            int c = (v * LPR + lane) * 4;
            if (c < H) st4(dx + (size_t)row * H + c, acc[v]);
        }
    }
}


// item i of the chunk list -> (row, first slot, last slot): row = the r with chunk_ptr[r] <= i < chunk_ptr[r+1]
__device__ __forceinline__ void chunk_item(const int32_t* __restrict__ chunk_ptr, const int32_t* __restrict__ rowptr, int num_rows,
                                           int item, int* row, int* beg, int* end) {
    int lo = 0, hi = num_rows;                   // invariant: chunk_ptr[lo] <= item < chunk_ptr[hi]
    while (hi - lo > 1) {
        int mid = (lo + hi) >> 1;
        if (chunk_ptr[mid] <= item) lo = mid; else hi = mid;
    }
    *row = lo;
    *beg = rowptr[lo] + (item - chunk_ptr[lo]) * CH;
    *end = min(rowptr[lo + 1], *beg + CH);
}

// Long rows, forward: one lane group sums one CH-edge chunk into partial[item,:] (no self term).
template <int LPR, int NV, bool GINE>
__global__ __launch_bounds__(AGG_BLOCK) void k_aggr_chunk_fwd(
    const float* __restrict__ x, const float* __restrict__ att, const float* __restrict__ edge_emb,
    const int32_t* __restrict__ rowptr, const int32_t* __restrict__ col, const int32_t* __restrict__ eid, int num_rows, int H,
    const int32_t* __restrict__ chunk_ptr, float* __restrict__ partial) {
    constexpr int GPB = AGG_BLOCK / LPR;
    const int lane = threadIdx.x % LPR;
    const int total = chunk_ptr[num_rows];
    for (int item = blockIdx.x * GPB + threadIdx.x / LPR; item < total; item += gridDim.x * GPB) {
        int row, beg, end;
        chunk_item(chunk_ptr, rowptr, num_rows, item, &row, &beg, &end);
        float4 acc[NV];
#pragma unroll
        for (int v = 0; v < NV; ++v) acc[v] = f4zero();
        int k = beg;
        for (; k + 4 <= end; k += 4) {             // four gathered rows in flight (hub chunks are latency-bound otherwise)
            int j[4], e[4];
            float w[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                j[u] = col[k + u];
                e[u] = (att != nullptr || GINE) ? eid[k + u] : 0;
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) w[u] = att ? att[e[u]] : 1.f;
#pragma unroll
            for (int v = 0; v < NV; ++v) {
                int c = (v * LPR + lane) * 4;
                if (c < H) {
                    float4 a[4];
#pragma unroll
                    for (int u = 0; u < 4; ++u) a[u] = ld4(x + (size_t)j[u] * H + c);
                    if (GINE) {
#pragma unroll
                        for (int u = 0; u < 4; ++u) {
                            float4 ea = ld4(edge_emb + (size_t)e[u] * H + c);
                            a[u] = make_float4(fmaxf(a[u].x + ea.x, 0.f), fmaxf(a[u].y + ea.y, 0.f), fmaxf(a[u].z + ea.z, 0.f), fmaxf(a[u].w + ea.w, 0.f));
                        }
                    }
#pragma unroll
                    for (int u = 0; u < 4; ++u) acc[v] = f4fma(w[u], a[u], acc[v]);
                }
            }
        }
        for (; k < end; ++k) {
            const int j = col[k];
            const int e = (att != nullptr || GINE) ? eid[k] : 0;
            const float w = att ? att[e] : 1.f;
#pragma unroll
            for (int v = 0; v < NV; ++v) {
                int c = (v * LPR + lane) * 4;
                if (c < H) {
                    float4 a = ld4(x + (size_t)j * H + c);
                    if (GINE) {
                        float4 ea = ld4(edge_emb + (size_t)e * H + c);
                        a = make_float4(fmaxf(a.x + ea.x, 0.f), fmaxf(a.y + ea.y, 0.f), fmaxf(a.z + ea.z, 0.f), fmaxf(a.w + ea.w, 0.f));
                    }
                    acc[v] = f4fma(w, a, acc[v]);
                }
            }
        }
#pragma unroll
        for (int v = 0; v < NV; ++v) {
            int c = (v * LPR + lane) * 4;
            if (c < H) st4(partial + (size_t)item * H + c, acc[v]);
        }
    }
}

// Long rows, backward: per-edge datt / dedge_emb are final; the dx contribution of the chunk goes to partial[item,:].
template <int LPR, int NV, bool GINE>
__global__ __launch_bounds__(AGG_BLOCK) void k_aggr_chunk_bwd(
    const float* __restrict__ x, const float* __restrict__ att, const float* __restrict__ edge_emb, const float* __restrict__ dout,
    const int32_t* __restrict__ rowptr, const int32_t* __restrict__ dstc, const int32_t* __restrict__ eid, int num_rows, int H,
    float* __restrict__ datt, float* __restrict__ dedge, const int32_t* __restrict__ chunk_ptr, float* __restrict__ partial) {
    constexpr int GPB = AGG_BLOCK / LPR;
    const int lane = threadIdx.x % LPR;
    const int total = chunk_ptr[num_rows];
    for (int item = blockIdx.x * GPB + threadIdx.x / LPR; item < total; item += gridDim.x * GPB) {
        int row, beg, end;
        chunk_item(chunk_ptr, rowptr, num_rows, item, &row, &beg, &end);
        float4 acc[NV], xj[NV];
#pragma unroll
        for (int v = 0; v < NV; ++v) {
            int c = (v * LPR + lane) * 4;
            acc[v] = f4zero();
            xj[v] = c < H ? ld4(x + (size_t)row * H + c) : f4zero();
        }
        for (int k = beg; k < end; ++k) {
            const int i = dstc[k], e = eid[k];
            const float w = att ? att[e] : 1.f;
            float dot = 0.f;
#pragma unroll
            for (int v = 0; v < NV; ++v) {
                int c = (v * LPR + lane) * 4;
                if (c < H) {
                    float4 g = ld4(dout + (size_t)i * H + c);
                    if (GINE) {
                        float4 ea = ld4(edge_emb + (size_t)e * H + c);
                        float4 pre = make_float4(xj[v].x + ea.x, xj[v].y + ea.y, xj[v].z + ea.z, xj[v].w + ea.w);
                        float4 m = make_float4(fmaxf(pre.x, 0.f), fmaxf(pre.y, 0.f), fmaxf(pre.z, 0.f), fmaxf(pre.w, 0.f));
                        dot += f4dot(m, g);
                        g = make_float4(pre.x > 0.f ? g.x : 0.f, pre.y > 0.f ? g.y : 0.f, pre.z > 0.f ? g.z : 0.f, pre.w > 0.f ? g.w : 0.f);
                        if (dedge) st4(dedge + (size_t)e * H + c, make_float4(w * g.x, w * g.y, w * g.z, w * g.w));
                    } else {
                        dot += f4dot(xj[v], g);
                    }
                    acc[v] = f4fma(w, g, acc[v]);
                }
            }
            if (datt) {
                dot = group_sum<LPR>(dot);
                if (lane == 0) datt[e] = dot;
            }
        }
#pragma unroll
        for (int v = 0; v < NV; ++v) {
            int c = (v * LPR + lane) * 4;
            if (c < H) st4(partial + (size_t)item * H + c, acc[v]);
        }
    }
}

template <int LPR, int NV>
__global__ __launch_bounds__(AGG_BLOCK) void k_segment_pool_fwd(const float* __restrict__ x, const int32_t* __restrict__ ptr,
                                                                 int num_seg, int H, int mean, float* __restrict__ out) {
    constexpr int GPB = AGG_BLOCK / LPR;
    const int lane = threadIdx.x % LPR;
    for (int g = blockIdx.x * GPB + threadIdx.x / LPR; g < num_seg; g += gridDim.x * GPB) {
        const int beg = ptr[g], end = ptr[g + 1];
        const float scale = mean ? 1.f / (float)max(end - beg, 1) : 1.f;
        float4 acc[NV];
#pragma unroll
        for (int v = 0; v < NV; ++v) acc[v] = f4zero();
        for (int r = beg; r < end; ++r) {
#pragma unroll
            for (int v = 0; v < NV; ++v) {
                int c = (v * LPR + lane) * 4;
                if (c < H) {
                    float4 a = ld4(x + (size_t)r * H + c);
                    acc[v].x += a.x; acc[v].y += a.y; acc[v].z += a.z; acc[v].w += a.w;
                }
            }
        }
#pragma unroll
        for (int v = 0; v < NV; ++v) {
            int c = (v * LPR + lane) * 4;
            if (c < H) {
                if (mean) acc[v] = make_float4(acc[v].x / (float)max(end - beg, 1), acc[v].y / (float)max(end - beg, 1),
                                               acc[v].z / (float)max(end - beg, 1), acc[v].w / (float)max(end - beg, 1));
                st4(out + (size_t)g * H + c, acc[v]);
            }
        }
        (void)scale;
    }
}

template <int LPR, int NV>
__global__ __launch_bounds__(AGG_BLOCK) void k_segment_pool_bwd(const float* __restrict__ dout, const int32_t* __restrict__ ptr,
                                                                 int num_seg, int H, int mean, float* __restrict__ dx) {
    // blockIdx.x = segment; rows of the segment are spread over the groups of gridDim.y blocks
    constexpr int GPB = AGG_BLOCK / LPR;
    const int lane = threadIdx.x % LPR;
    const int g = blockIdx.x;
    const int beg = ptr[g], end = ptr[g + 1];
    const float cnt = (float)max(end - beg, 1);
    float4 gv[NV];
#pragma unroll
    for (int v = 0; v < NV; ++v) {
        int c = (v * LPR + lane) * 4;
        gv[v] = f4zero();
        if (c < H) {
            gv[v] = ld4(dout + (size_t)g * H + c);
            if (mean) gv[v] = make_float4(gv[v].x / cnt, gv[v].y / cnt, gv[v].z / cnt, gv[v].w / cnt);
        }
    }
    for (int r = beg + blockIdx.y * GPB + threadIdx.x / LPR; r < end; r += gridDim.y * GPB) {
#pragma unroll
        for (int v = 0; v < NV; ++v) {
            int c = (v * LPR + lane) * 4;
            if (c < H) st4(dx + (size_t)r * H + c, gv[v]);
        }
    }
}

struct RowGeom { int lpr, nv; };
static inline bool row_geom(int64_t H, RowGeom* g) {
    if (H <= 0 || H % 4 != 0 || H > 2048) return false;
    int64_t q = H / 4;
    if (q <= 4) *g = {4, 1};
    else if (q <= 8) *g = {8, 1};
    else if (q <= 16) *g = {16, 1};
    else if (q <= 32) *g = {32, 1};
    else if (q <= 64) *g = {64, 1};
    else if (q <= 128) *g = {64, 2};
    else if (q <= 256) *g = {64, 4};
    else *g = {64, 8};
    return true;
}

// Dispatch a kernel template over the supported (LPR, NV) geometries.
#define GSAT_ROW_DISPATCH(geom, CALL)                     \
    do {                                                  \
        if ((geom).nv == 1) {                             \
            switch ((geom).lpr) {                         \
                case 4: CALL(4, 1); break;                \
                case 8: CALL(8, 1); break;                \
                case 16: CALL(16, 1); break;              \
                case 32: CALL(32, 1); break;              \
                default: CALL(64, 1); break;              \
            }                                             \
        } else if ((geom).nv == 2) { CALL(64, 2); }       \
        else if ((geom).nv == 4) { CALL(64, 4); }         \
        else { CALL(64, 8); }                             \
    } while (0)

static inline void span_grid(int64_t num_rows, int lpr, int* nblocks, int* rows_per_group) {
    const int gpb = AGG_BLOCK / lpr;
    int64_t groups_needed = num_rows;                       // one row per group if the grid allows it
    int64_t nb = ceil_div(groups_needed, gpb);
    const int64_t cap = 256 * 64;                            // one row per lane group up to 16k blocks (shorter per-wave chains)
    if (nb > cap) nb = cap;
    if (nb < 1) nb = 1;
    *nblocks = (int)nb;
    *rows_per_group = (int)ceil_div(num_rows, nb * gpb);
    if (*rows_per_group < 1) *rows_per_group = 1;
}


// upper bound on the number of long-row chunks of a CSR with E entries: every long row has > CH entries
static inline int64_t max_chunks(int64_t E) { return 2 * (E / CH) + 1; }

int aggr_sum_fwd_impl(hipStream_t stream, const float* x, const float* self_rows, const float* att, const float* edge_emb,
                      const int32_t* rowptr, const int32_t* col, const int32_t* eid, int64_t N, int64_t E, int64_t H, float self_coef,
                      float* out, const int32_t* chunk_ptr, float* partial) {
    GSAT_REQUIRE(N >= 0 && N < (1ll << 31), GSAT_ERR_ARG, "gsat_aggr_sum_fwd: bad N");
    if (N == 0) return GSAT_OK;
    RowGeom g;
    GSAT_REQUIRE(row_geom(H, &g), GSAT_ERR_UNSUPPORTED, "gsat_aggr_sum_fwd: H=%lld must be a multiple of 4 and <= 2048", (long long)H);
    GSAT_REQUIRE((x || (self_rows && E == 0)) && rowptr && out && (col || E == 0), GSAT_ERR_ARG, "gsat_aggr_sum_fwd: null pointer");   // an edge-row source with E == 0 rows is empty, hence NULL
    GSAT_REQUIRE((att == nullptr && edge_emb == nullptr) || eid || E == 0, GSAT_ERR_ARG, "gsat_aggr_sum_fwd: eid required with att/edge_emb");
    GSAT_REQUIRE(chunk_ptr == nullptr || partial != nullptr, GSAT_ERR_ARG, "gsat_aggr_sum_fwd: chunk_ptr needs a partial-sum workspace");
    if (E <= CH) chunk_ptr = nullptr;                      // no row can be long
    if (!self_rows) self_rows = x;
    int nb, rpg;
    span_grid(N, g.lpr, &nb, &rpg);
    const int cb = chunk_ptr ? (int)std::min<int64_t>(ceil_div(max_chunks(E), AGG_BLOCK / g.lpr), 256 * 16) : 0;
#define CALL(L, V)                                                                                               \
    do {                                                                                                         \
        if (edge_emb) {                                                                                          \
            if (cb) k_aggr_chunk_fwd<L, V, true><<<cb, AGG_BLOCK, 0, stream>>>(x, att, edge_emb, rowptr, col, eid, (int)N, (int)H, chunk_ptr, partial); \
            k_aggr_sum_fwd<L, V, true><<<nb, AGG_BLOCK, 0, stream>>>(x, self_rows, att, edge_emb, rowptr, col, eid, (int)N, (int)H, self_coef, out, rpg, chunk_ptr, partial); \
        } else {                                                                                                 \
            if (cb) k_aggr_chunk_fwd<L, V, false><<<cb, AGG_BLOCK, 0, stream>>>(x, att, edge_emb, rowptr, col, eid, (int)N, (int)H, chunk_ptr, partial); \
            k_aggr_sum_fwd<L, V, false><<<nb, AGG_BLOCK, 0, stream>>>(x, self_rows, att, edge_emb, rowptr, col, eid, (int)N, (int)H, self_coef, out, rpg, chunk_ptr, partial); \
        }                                                                                                        \
    } while (0)
    GSAT_ROW_DISPATCH(g, CALL);
#undef CALL
    GSAT_LAUNCH_CHECK();
    return GSAT_OK;
}

}  // namespace gsat

using namespace gsat;

extern "C" {

size_t gsat_long_row_partial_floats(int64_t num_edges, int64_t H) { return (size_t)max_chunks(num_edges) * (size_t)H; }

int gsat_aggr_sum_fwd(const float* x, const float* self_rows, const float* att, const float* edge_emb,
                      const int32_t* rowptr, const int32_t* col, const int32_t* eid, int64_t N, int64_t E, int64_t H,
                      float self_coef, float* out, const int32_t* chunk_ptr, float* partial, void* stream_) {
    return aggr_sum_fwd_impl((hipStream_t)stream_, x, self_rows, att, edge_emb, rowptr, col, eid, N, E, H, self_coef, out, chunk_ptr, partial);
}

int gsat_aggr_sum_bwd(const float* x, const float* att, const float* edge_emb, const float* dout,
                      const int32_t* rowptr_src, const int32_t* dst_sorted, const int32_t* eid_src, int64_t N, int64_t E,
                      int64_t H, float self_coef, float* dx, float* datt, float* dedge_emb, const int32_t* chunk_ptr,
                      float* partial, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    GSAT_REQUIRE(N >= 0 && N < (1ll << 31), GSAT_ERR_ARG, "gsat_aggr_sum_bwd: bad N");
    if (N == 0) return GSAT_OK;
    RowGeom g;
    GSAT_REQUIRE(row_geom(H, &g), GSAT_ERR_UNSUPPORTED, "gsat_aggr_sum_bwd: H=%lld must be a multiple of 4 and <= 2048", (long long)H);
    GSAT_REQUIRE(x && dout && rowptr_src && dx && ((dst_sorted && eid_src) || E == 0), GSAT_ERR_ARG, "gsat_aggr_sum_bwd: null pointer");
    GSAT_REQUIRE(chunk_ptr == nullptr || partial != nullptr, GSAT_ERR_ARG, "gsat_aggr_sum_bwd: chunk_ptr needs a partial-sum workspace");
    if (E <= CH) chunk_ptr = nullptr;
    int nb, rpg;
    span_grid(N, g.lpr, &nb, &rpg);
    const int cb = chunk_ptr ? (int)std::min<int64_t>(ceil_div(max_chunks(E), AGG_BLOCK / g.lpr), 256 * 16) : 0;
#define CALL(L, V)                                                                                               \
    do {                                                                                                         \
        if (edge_emb) {                                                                                          \
            if (cb) k_aggr_chunk_bwd<L, V, true><<<cb, AGG_BLOCK, 0, stream>>>(x, att, edge_emb, dout, rowptr_src, dst_sorted, eid_src, (int)N, (int)H, datt, dedge_emb, chunk_ptr, partial); \
            k_aggr_sum_bwd<L, V, true><<<nb, AGG_BLOCK, 0, stream>>>(x, att, edge_emb, dout, rowptr_src, dst_sorted, eid_src, (int)N, (int)H, self_coef, dx, datt, dedge_emb, rpg, chunk_ptr, partial); \
        } else {                                                                                                 \
            if (cb) k_aggr_chunk_bwd<L, V, false><<<cb, AGG_BLOCK, 0, stream>>>(x, att, edge_emb, dout, rowptr_src, dst_sorted, eid_src, (int)N, (int)H, datt, dedge_emb, chunk_ptr, partial); \
            k_aggr_sum_bwd<L, V, false><<<nb, AGG_BLOCK, 0, stream>>>(x, att, edge_emb, dout, rowptr_src, dst_sorted, eid_src, (int)N, (int)H, self_coef, dx, datt, dedge_emb, rpg, chunk_ptr, partial); \
        }                                                                                                        \
    } while (0)
    GSAT_ROW_DISPATCH(g, CALL);
#undef CALL
    GSAT_LAUNCH_CHECK();
    return GSAT_OK;
}

int gsat_segment_pool_fwd(const float* x, const int32_t* ptr, int64_t G, int64_t H, int mean, float* out, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    GSAT_REQUIRE(G >= 0 && G < (1ll << 31), GSAT_ERR_ARG, "gsat_segment_pool_fwd: bad G");
    if (G == 0) return GSAT_OK;
    RowGeom g;
    GSAT_REQUIRE(row_geom(H, &g), GSAT_ERR_UNSUPPORTED, "gsat_segment_pool_fwd: H=%lld must be a multiple of 4 and <= 2048", (long long)H);
    GSAT_REQUIRE(x && ptr && out, GSAT_ERR_ARG, "gsat_segment_pool_fwd: null pointer");
    int nb = (int)std::min<int64_t>(ceil_div(G, AGG_BLOCK / g.lpr), 4096);
#define CALL(L, V) k_segment_pool_fwd<L, V><<<nb, AGG_BLOCK, 0, stream>>>(x, ptr, (int)G, (int)H, mean, out)
    GSAT_ROW_DISPATCH(g, CALL);
#undef CALL
    GSAT_LAUNCH_CHECK();
    return GSAT_OK;
}

int gsat_segment_pool_bwd(const float* dout, const int32_t* ptr, int64_t G, int64_t H, int mean, float* dx, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    GSAT_REQUIRE(G >= 0 && G < (1ll << 31), GSAT_ERR_ARG, "gsat_segment_pool_bwd: bad G");
    if (G == 0) return GSAT_OK;
    RowGeom g;
    GSAT_REQUIRE(row_geom(H, &g), GSAT_ERR_UNSUPPORTED, "gsat_segment_pool_bwd: H=%lld must be a multiple of 4 and <= 2048", (long long)H);
    GSAT_REQUIRE(dout && ptr && dx, GSAT_ERR_ARG, "gsat_segment_pool_bwd: null pointer");
    dim3 grid((unsigned)G, 4);
#define CALL(L, V) k_segment_pool_bwd<L, V><<<grid, AGG_BLOCK, 0, stream>>>(dout, ptr, (int)G, (int)H, mean, dx)
    GSAT_ROW_DISPATCH(g, CALL);
#undef CALL
    GSAT_LAUNCH_CHECK();
    return GSAT_OK;
}

}  // extern "C"
