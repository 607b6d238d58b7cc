// Line graph with one dual node per UNDIRECTED primal edge -- the rule of the reference's ba_2motifs dual dataset
// (src/datasets/ba_2motifs_dual.py:35-62): undirected edges are numbered in row-major order of their (smaller, larger)
// endpoint pair, and two dual nodes are adjacent (both directions) when the primal edges share an endpoint; the dual
// edge list comes out in (i, j) row-major order, as dense_to_sparse of the reference's dual adjacency matrix does.
// On a collated batch (node ids ordered by graph) the global numbering equals the per-graph numberings laid end to end.
//
// The reference does this with O(n^2) Python loops over dense adjacency matrices per graph; here it is one radix sort
// of the directed edge keys plus three integer passes.  Results are bit-exact against oracle/bookkeeping.py.
#include "common.h"
#include <rocprim/rocprim.hpp>

namespace gsat {

__global__ void k_lg_keys(const int64_t* __restrict__ ei, int64_t E, int64_t N, uint64_t* __restrict__ keys, int32_t* __restrict__ ids,
                          int32_t* status) {
    int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= E) return;
    int64_t s = ei[e], d = ei[E + e];
    if (s < 0 || s >= N) { atomicAdd(&status[2], 1); s = N - 1; }
    if (d < 0 || d >= N) { atomicAdd(&status[2], 1); d = N - 1; }
    keys[e] = (uint64_t)s * (uint64_t)N + (uint64_t)d;
    ids[e] = (int32_t)e;
}

__device__ __forceinline__ int64_t lg_lower_bound(const uint64_t* __restrict__ a, int64_t n, uint64_t v) {
    int64_t lo = 0, hi = n;
    while (lo < hi) {
        const int64_t mid = (lo + hi) >> 1;
        if (a[mid] < v) lo = mid + 1; else hi = mid;
    }
    return lo;
}

// flag[k] = 1 iff sorted slot k is the numbering slot of its undirected edge: s < d, first copy of that (s, d).
// Also the by-source row pointers of the sorted list and the symmetry check (every (s, d) needs its (d, s)).
__global__ void k_lg_flags(const uint64_t* __restrict__ ks, int64_t E, int64_t N, int32_t* __restrict__ flag, int32_t* __restrict__ rowptr,
                           int32_t* status) {
    int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k <= N) rowptr[k] = (int32_t)lg_lower_bound(ks, E, (uint64_t)k * (uint64_t)N);
    if (k >= E) return;
    const uint64_t key = ks[k];
    const uint64_t s = key / (uint64_t)N, d = key % (uint64_t)N;
    const bool first = k == 0 || ks[k - 1] != key;
    flag[k] = (s < d && first) ? 1 : 0;
    if (s != d) {
        const uint64_t rk = d * (uint64_t)N + s;
        const int64_t lb = lg_lower_bound(ks, E, rk);
        if (lb >= E || ks[lb] != rk) atomicAdd(&status[1], 1);          // reverse edge missing: the rule needs a symmetric edge set
    }
}

// und_of_slot[k] = id of the undirected edge of sorted slot k (-1 for self loops); und_of_edge by original edge id;
// endpoints of every undirected edge; counts[i] = number of dual neighbours of dual node i.
__global__ void k_lg_und(const uint64_t* __restrict__ ks, const int32_t* __restrict__ perm, const int32_t* __restrict__ rank,
                         const int32_t* __restrict__ flag, int64_t E, int64_t N, int32_t* __restrict__ und_of_slot,
                         int32_t* __restrict__ und_of_edge, int32_t* __restrict__ und_src, int32_t* __restrict__ und_dst) {
    int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= E) return;
    const uint64_t key = ks[k];
    const uint64_t s = key / (uint64_t)N, d = key % (uint64_t)N;
    int32_t u = -1;
    if (s != d) {
        const uint64_t ck = s < d ? key : d * (uint64_t)N + s;
        const int64_t lb = lg_lower_bound(ks, E, ck);
        if (lb < E && ks[lb] == ck) u = rank[lb];                       // exclusive scan at the first copy of the canonical slot
    }
    und_of_slot[k] = u;
    und_of_edge[perm[k]] = u;
    if (flag[k]) { und_src[rank[k]] = (int32_t)s; und_dst[rank[k]] = (int32_t)d; }
}

// distinct, non-self incident undirected edges of node v other than `self`: visits them in ascending id order
template <class F>
__device__ __forceinline__ void lg_visit_row(const int32_t* __restrict__ rowptr, const int32_t* __restrict__ und_of_slot, int v, int self, F f) {
    int prev = -1;
    for (int k = rowptr[v]; k < rowptr[v + 1]; ++k) {
        const int u = und_of_slot[k];
        if (u < 0 || u == self || u == prev) continue;                  // self loop, the edge itself, duplicate copy
        prev = u;
        f(u);
    }
}

__global__ void k_lg_counts(const int32_t* __restrict__ rowptr, const int32_t* __restrict__ und_of_slot, const int32_t* __restrict__ und_src,
                            const int32_t* __restrict__ und_dst, int64_t M, int64_t* __restrict__ counts) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= M) return;
    int64_t c = 0;
    lg_visit_row(rowptr, und_of_slot, und_src[i], (int)i, [&](int) { ++c; });
    lg_visit_row(rowptr, und_of_slot, und_dst[i], (int)i, [&](int) { ++c; });
    counts[i] = c;
}

// dual node i = (a, b): its neighbours are the other edges at a and at b, each list ascending -> one two-way merge
__global__ void k_lg_fill(const int32_t* __restrict__ rowptr, const int32_t* __restrict__ und_of_slot, const int32_t* __restrict__ und_src,
                          const int32_t* __restrict__ und_dst, const int64_t* __restrict__ dual_ptr, int64_t M, int64_t total,
                          int64_t* __restrict__ dual_ei) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= M) return;
    const int a = und_src[i], b = und_dst[i];
    int ka = rowptr[a], ea = rowptr[a + 1], kb = rowptr[b], eb = rowptr[b + 1];
    int64_t o = dual_ptr[i];
    const int64_t o_end = dual_ptr[i + 1];
    int last = -1;
    auto next = [&](int& k, int e) -> int {                             // next admissible id of a row, or INT_MAX at its end
        while (k < e) {
            const int u = und_of_slot[k];
            if (u >= 0 && u != (int)i && u != last) return u;
            ++k;
        }
        return 0x7fffffff;
    };
    while (o < o_end) {
        const int ua = next(ka, ea), ub = next(kb, eb);
        const int u = ua < ub ? ua : ub;
        if (u == 0x7fffffff) break;
        if (ua <= ub) ++ka;
        if (ub <= ua) ++kb;
        last = u;
        dual_ei[o] = i;
        dual_ei[total + o] = u;
        ++o;
    }
}

__global__ void k_lg_total(const int32_t* __restrict__ rank, const int32_t* __restrict__ flag, int64_t E, int32_t* status) {
    if (blockIdx.x == 0 && threadIdx.x == 0) status[0] = E > 0 ? rank[E - 1] + flag[E - 1] : 0;
}

static size_t lg_sort_temp(int64_t n) {
    size_t tb = 0;
    uint64_t* k = nullptr;
    int32_t* v = nullptr;
    (void)rocprim::radix_sort_pairs(nullptr, tb, k, k, v, v, (size_t)(n > 0 ? n : 1), 0, 64u, (hipStream_t)0);
    return align_up(tb, 256) + 256;
}
static size_t lg_scan_temp(int64_t n) {
    size_t tb = 0;
    int32_t* p = nullptr;
    (void)rocprim::exclusive_scan(nullptr, tb, p, p, 0, (size_t)(n > 0 ? n : 1), rocprim::plus<int>(), (hipStream_t)0);
    return align_up(tb, 256) + 256;
}

}  // namespace gsat

using namespace gsat;

extern "C" {

size_t gsat_und_edges_workspace_bytes(int64_t E) {
    const size_t e = (size_t)(E > 0 ? E : 1);
    return align_up(e * 8, 256) + 3 * align_up(e * 4, 256) + lg_sort_temp(E) + lg_scan_temp(E);
}

int gsat_und_edges(const int64_t* edge_index, int64_t E, int64_t N, uint64_t* sorted_keys, int32_t* rowptr, int32_t* und_of_slot,
                   int32_t* und_of_edge, int32_t* und_src, int32_t* und_dst, int32_t* status, void* workspace, size_t ws_bytes,
                   void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    GSAT_REQUIRE(E >= 0 && N >= 0 && status && rowptr, GSAT_ERR_ARG, "gsat_und_edges: bad argument");
    GSAT_REQUIRE(E < (1ll << 31) && N < (1ll << 31), GSAT_ERR_UNSUPPORTED, "gsat_und_edges: >2^31 entries");
    GSAT_CHECK_HIP(gsat::zero_async(status, 4 * sizeof(int32_t), stream));
    if (E == 0) {
        GSAT_CHECK_HIP(gsat::zero_async(rowptr, (size_t)(N + 1) * sizeof(int32_t), stream));
        return GSAT_OK;
    }
    GSAT_REQUIRE(edge_index && sorted_keys && und_of_slot && und_of_edge && und_src && und_dst && N > 0, GSAT_ERR_ARG,
                 "gsat_und_edges: null pointer");
    Arena ar(workspace, ws_bytes);
    uint64_t* kin = ar.take<uint64_t>(E);
    int32_t* ids = ar.take<int32_t>(E);
    int32_t* perm = ar.take<int32_t>(E);
    int32_t* flag = ar.take<int32_t>(E);
    size_t tb = lg_sort_temp(E), tc = lg_scan_temp(E);
    char* temp = ar.take<char>(tb);
    char* temp2 = ar.take<char>(tc);
    GSAT_REQUIRE(ar.ok() && temp && temp2, GSAT_ERR_WORKSPACE, "gsat_und_edges: workspace %zu < %zu", ws_bytes, ar.off);
    const int B = 256;
    k_lg_keys<<<ceil_div(E, B), B, 0, stream>>>(edge_index, E, N, kin, ids, status);
    GSAT_LAUNCH_CHECK();
    int bits = 1;
    while (bits < 64 && (((uint64_t)N * (uint64_t)N - 1) >> bits) != 0) ++bits;
    GSAT_CHECK_HIP(rocprim::radix_sort_pairs(temp, tb, kin, sorted_keys, ids, perm, (size_t)E, 0, (unsigned)bits, stream));
    k_lg_flags<<<ceil_div(std::max<int64_t>(E, N + 1), B), B, 0, stream>>>(sorted_keys, E, N, flag, rowptr, status);
    GSAT_LAUNCH_CHECK();
    int32_t* rank = ids;                                                  // the unsorted ids are dead after the sort
    GSAT_CHECK_HIP(rocprim::exclusive_scan(temp2, tc, flag, rank, 0, (size_t)E, rocprim::plus<int>(), stream));
    k_lg_total<<<1, 64, 0, stream>>>(rank, flag, E, status);
    GSAT_LAUNCH_CHECK();
    k_lg_und<<<ceil_div(E, B), B, 0, stream>>>(sorted_keys, perm, rank, flag, E, N, und_of_slot, und_of_edge, und_src, und_dst);
    GSAT_LAUNCH_CHECK();
    return GSAT_OK;
}

int gsat_und_line_graph_counts(const int32_t* rowptr, const int32_t* und_of_slot, const int32_t* und_src, const int32_t* und_dst,
                               int64_t M, int64_t* counts, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    GSAT_REQUIRE(M >= 0, GSAT_ERR_ARG, "gsat_und_line_graph_counts: bad M");
    if (M == 0) return GSAT_OK;
    GSAT_REQUIRE(rowptr && und_of_slot && und_src && und_dst && counts, GSAT_ERR_ARG, "gsat_und_line_graph_counts: null pointer");
    k_lg_counts<<<ceil_div(M, 256), 256, 0, stream>>>(rowptr, und_of_slot, und_src, und_dst, M, counts);
    GSAT_LAUNCH_CHECK();
    return GSAT_OK;
}

int gsat_und_line_graph(const int32_t* rowptr, const int32_t* und_of_slot, const int32_t* und_src, const int32_t* und_dst,
                        const int64_t* dual_ptr, int64_t M, int64_t total, int64_t* dual_edge_index, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    GSAT_REQUIRE(M >= 0 && total >= 0, GSAT_ERR_ARG, "gsat_und_line_graph: bad extents");
    if (M == 0 || total == 0) return GSAT_OK;
    GSAT_REQUIRE(rowptr && und_of_slot && und_src && und_dst && dual_ptr && dual_edge_index, GSAT_ERR_ARG, "gsat_und_line_graph: null pointer");
    k_lg_fill<<<ceil_div(M, 256), 256, 0, stream>>>(rowptr, und_of_slot, und_src, und_dst, dual_ptr, M, total, dual_edge_index);
    GSAT_LAUNCH_CHECK();
    return GSAT_OK;
}

}  // extern "C"
