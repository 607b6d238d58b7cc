// Small per-edge / per-node operators of the GSAT step: concrete & Gumbel-sigmoid samplers,
// node->edge lift, reverse-edge symmetrisation and the KL "info" loss, forward and backward.
// All are single coalesced passes (4-12 B/edge); reductions are two-stage and order-fixed.
#include "common.h"

namespace gsat {

constexpr int EB = 256;

// mode 0: sigmoid(z/temp); 1: concrete  z + log u - log(1-u); 2: gumbel  z - log(-log(U+eps)+eps)
__global__ void k_sample_fwd(const float* __restrict__ z, const float* __restrict__ noise, int mode, float temp, float eps,
                             int64_t M, float* __restrict__ att) {
    int64_t m = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (m >= M) return;
    float t = z[m];
    if (mode == 1) { float u = noise[m]; t += logf(u) - logf(1.0f - u); }
    else if (mode == 2) { float u = noise[m]; t += -logf(-logf(u + eps) + eps); }
    att[m] = 1.f / (1.f + expf(-t / temp));
}

__global__ void k_sample_bwd(const float* __restrict__ att, const float* __restrict__ datt, float temp, int64_t M, float* __restrict__ dz) {
    int64_t m = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (m >= M) return;
    float a = att[m];
    dz[m] = datt[m] * a * (1.f - a) / temp;
}

__global__ void k_lift_fwd(const float* __restrict__ a, const int32_t* __restrict__ src, const int32_t* __restrict__ dst, int64_t E,
                           float* __restrict__ out) {
    int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e < E) out[e] = a[src[e]] * a[dst[e]];
}

// da[n] = sum_{k in out(n)} dout[eid]*a[dst] + sum_{k in in(n)} dout[eid]*a[src]   (out-edges first, CSR order)
__global__ void k_lift_bwd(const float* __restrict__ a, const float* __restrict__ dout, const int32_t* __restrict__ rp_src,
                           const int32_t* __restrict__ dst_by_src, const int32_t* __restrict__ eid_by_src,
                           const int32_t* __restrict__ rp_dst, const int32_t* __restrict__ src_by_dst,
                           const int32_t* __restrict__ eid_by_dst, int64_t N, float* __restrict__ da) {
    int64_t n = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= N) return;
    float acc = 0.f;
    for (int k = rp_src[n]; k < rp_src[n + 1]; ++k) acc = fmaf(dout[eid_by_src[k]], a[dst_by_src[k]], acc);
    for (int k = rp_dst[n]; k < rp_dst[n + 1]; ++k) acc = fmaf(dout[eid_by_dst[k]], a[src_by_dst[k]], acc);
    da[n] = acc;
}

// out[k] = (a[k] + a[rev[k]]) / 2 ; rev is an involution so the same kernel is its own backward
__global__ void k_symmetrise(const float* __restrict__ a, const int32_t* __restrict__ rev, const int32_t* __restrict__ flag, int64_t E,
                             float* __restrict__ out) {
    int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= E) return;
    const bool sym = flag == nullptr || flag[0] != 0;       // device-side `if is_undirected(...)`: no host round trip
    out[e] = sym ? (a[e] + a[rev[e]]) / 2.f : a[e];
}

__device__ __forceinline__ float info_term(float a, float r) {
    return a * logf(a / r + 1e-6f) + (1.f - a) * logf((1.f - a) / (1.f - r + 1e-6f) + 1e-6f);
}
__device__ __forceinline__ float info_dterm(float a, float r) {
    const float q = 1.f - r + 1e-6f;
    const float t1 = a / r + 1e-6f, t2 = (1.f - a) / q + 1e-6f;
    return logf(t1) + a / (r * t1) - logf(t2) - (1.f - a) / (q * t2);
}

// block partial sums of the info-loss terms; partial[b] in block order
__global__ __launch_bounds__(EB) void k_info_partial(const float* __restrict__ att, const float* __restrict__ r_vec, float r_scalar,
                                                     int64_t M, int64_t per_block, float* __restrict__ partial) {
    __shared__ float sm[EB];
    const int64_t beg = (int64_t)blockIdx.x * per_block, end = min(M, beg + per_block);
    float acc = 0.f;
    for (int64_t m = beg + threadIdx.x; m < end; m += EB) acc += info_term(att[m], r_vec ? r_vec[m] : r_scalar);
    sm[threadIdx.x] = acc;
    __syncthreads();
    for (int s = EB / 2; s > 0; s >>= 1) {
        if (threadIdx.x < s) sm[threadIdx.x] += sm[threadIdx.x + s];
        __syncthreads();
    }
    if (threadIdx.x == 0) partial[blockIdx.x] = sm[0];
}

__global__ __launch_bounds__(EB) void k_info_final(const float* __restrict__ partial, int nb, float inv_m, float* __restrict__ out) {
    __shared__ float sm[EB];
    float acc = 0.f;
    for (int i = threadIdx.x; i < nb; i += EB) acc += partial[i];
    sm[threadIdx.x] = acc;
    __syncthreads();
    for (int s = EB / 2; s > 0; s >>= 1) {
        if (threadIdx.x < s) sm[threadIdx.x] += sm[threadIdx.x + s];
        __syncthreads();
    }
    if (threadIdx.x == 0) out[0] = sm[0] * inv_m;
}

__global__ void k_info_bwd(const float* __restrict__ att, const float* __restrict__ r_vec, float r_scalar, const float* __restrict__ gout,
                           int64_t M, float inv_m, float* __restrict__ datt) {
    int64_t m = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (m < M) datt[m] = gout[0] * inv_m * info_dterm(att[m], r_vec ? r_vec[m] : r_scalar);
}

__global__ void k_narrow(const int64_t* __restrict__ in, int64_t n, int32_t* __restrict__ out) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = (int32_t)in[i];
}

}  // namespace gsat

using namespace gsat;
#define GRID1(n) (unsigned)ceil_div((n), EB), EB, 0, stream

extern "C" {

int gsat_sample_fwd(const float* logits, const float* noise, int mode, float temp, float eps, int64_t M, float* att, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    GSAT_REQUIRE(M >= 0 && mode >= 0 && mode <= 2 && temp > 0.f, GSAT_ERR_ARG, "gsat_sample_fwd: bad argument");
    if (M == 0) return GSAT_OK;
    GSAT_REQUIRE(logits && att && (mode == 0 || noise), GSAT_ERR_ARG, "gsat_sample_fwd: null pointer");
    k_sample_fwd<<<GRID1(M)>>>(logits, noise, mode, temp, eps, M, att);
    GSAT_LAUNCH_CHECK();
    return GSAT_OK;
}

int gsat_sample_bwd(const float* att, const float* datt, float temp, int64_t M, float* dlogits, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    GSAT_REQUIRE(M >= 0 && temp > 0.f, GSAT_ERR_ARG, "gsat_sample_bwd: bad argument");
    if (M == 0) return GSAT_OK;
    GSAT_REQUIRE(att && datt && dlogits, GSAT_ERR_ARG, "gsat_sample_bwd: null pointer");
    k_sample_bwd<<<GRID1(M)>>>(att, datt, temp, M, dlogits);
    GSAT_LAUNCH_CHECK();
    return GSAT_OK;
}

int gsat_lift_fwd(const float* node_att, const int32_t* src, const int32_t* dst, int64_t E, float* edge_att, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    GSAT_REQUIRE(E >= 0, GSAT_ERR_ARG, "gsat_lift_fwd: bad E");
    if (E == 0) return GSAT_OK;
    GSAT_REQUIRE(node_att && src && dst && edge_att, GSAT_ERR_ARG, "gsat_lift_fwd: null pointer");
    k_lift_fwd<<<GRID1(E)>>>(node_att, src, dst, E, edge_att);
    GSAT_LAUNCH_CHECK();
    return GSAT_OK;
}

int gsat_lift_bwd(const float* node_att, const float* dedge_att, const int32_t* rowptr_src, const int32_t* dst_by_src,
                  const int32_t* eid_by_src, const int32_t* rowptr_dst, const int32_t* src_by_dst, const int32_t* eid_by_dst,
                  int64_t N, float* dnode_att, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    GSAT_REQUIRE(N >= 0, GSAT_ERR_ARG, "gsat_lift_bwd: bad N");
    if (N == 0) return GSAT_OK;
    GSAT_REQUIRE(node_att && dedge_att && rowptr_src && dst_by_src && eid_by_src && rowptr_dst && src_by_dst && eid_by_dst && dnode_att,
                 GSAT_ERR_ARG, "gsat_lift_bwd: null pointer");
    k_lift_bwd<<<GRID1(N)>>>(node_att, dedge_att, rowptr_src, dst_by_src, eid_by_src, rowptr_dst, src_by_dst, eid_by_dst, N, dnode_att);
    GSAT_LAUNCH_CHECK();
    return GSAT_OK;
}

int gsat_symmetrise(const float* att, const int32_t* rev, const int32_t* undirected_flag, int64_t E, float* out, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    GSAT_REQUIRE(E >= 0, GSAT_ERR_ARG, "gsat_symmetrise: bad E");
    if (E == 0) return GSAT_OK;
    GSAT_REQUIRE(att && rev && out, GSAT_ERR_ARG, "gsat_symmetrise: null pointer");
    k_symmetrise<<<GRID1(E)>>>(att, rev, undirected_flag, E, out);
    GSAT_LAUNCH_CHECK();
    return GSAT_OK;
}

int gsat_info_loss_fwd(const float* att, const float* r_vec, float r_scalar, int64_t M, float* partial /* [1024] */, float* out,
                       void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    GSAT_REQUIRE(M > 0 && att && partial && out, GSAT_ERR_ARG, "gsat_info_loss_fwd: bad argument (mean over an empty set)");
    int nb = (int)std::min<int64_t>(1024, ceil_div(M, EB * 4));
    int64_t per_block = ceil_div(M, nb);
    nb = (int)ceil_div(M, per_block);
    k_info_partial<<<nb, EB, 0, stream>>>(att, r_vec, r_scalar, M, per_block, partial);
    k_info_final<<<1, EB, 0, stream>>>(partial, nb, 1.f / (float)M, out);
    GSAT_LAUNCH_CHECK();
    return GSAT_OK;
}

int gsat_info_loss_bwd(const float* att, const float* r_vec, float r_scalar, const float* gout, int64_t M, float* datt, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    GSAT_REQUIRE(M > 0 && att && gout && datt, GSAT_ERR_ARG, "gsat_info_loss_bwd: bad argument");
    k_info_bwd<<<GRID1(M)>>>(att, r_vec, r_scalar, gout, M, 1.f / (float)M, datt);
    GSAT_LAUNCH_CHECK();
    return GSAT_OK;
}

int gsat_narrow_i64(const int64_t* in, int64_t n, int32_t* out, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    GSAT_REQUIRE(n >= 0, GSAT_ERR_ARG, "gsat_narrow_i64: bad n");
    if (n == 0) return GSAT_OK;
    GSAT_REQUIRE(in && out, GSAT_ERR_ARG, "gsat_narrow_i64: null pointer");
    k_narrow<<<GRID1(n)>>>(in, n, out);
    GSAT_LAUNCH_CHECK();
    return GSAT_OK;
}

}  // extern "C"

// ------------------------------------------------------------------------------------------------
// Device-side batch assembly (PyG Batch.from_data_list, src/utils/get_data_loaders.py:130-145): the dataset stays packed
// in HBM (all graphs' nodes / edges concatenated, local node ids); a batch is a list of graph ids.
// ------------------------------------------------------------------------------------------------
namespace gsat {

__device__ __forceinline__ int seg_of(const int64_t* __restrict__ ptr, int nseg, int64_t i) {   // ptr[s] <= i < ptr[s+1]
    int lo = 0, hi = nseg;
    while (hi - lo > 1) {
        int mid = (lo + hi) >> 1;
        if (ptr[mid] <= i) lo = mid; else hi = mid;
    }
    return lo;
}

__global__ void k_collate_nodes(const int64_t* __restrict__ ids, const int64_t* __restrict__ node_ptr_all,
                                const int64_t* __restrict__ out_node_ptr, int G, int64_t N, int64_t* __restrict__ batch,
                                int64_t* __restrict__ node_src_row) {
    int64_t n = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= N) return;
    const int g = seg_of(out_node_ptr, G, n);
    batch[n] = g;
    node_src_row[n] = node_ptr_all[ids[g]] + (n - out_node_ptr[g]);
}

__global__ void k_collate_edges(const int64_t* __restrict__ ids, const int64_t* __restrict__ edge_ptr_all,
                                const int64_t* __restrict__ out_edge_ptr, const int64_t* __restrict__ out_node_ptr,
                                const int64_t* __restrict__ edge_local_all, int64_t E_all, int G, int64_t E,
                                int64_t* __restrict__ edge_index, int64_t* __restrict__ edge_src_slot) {
    int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= E) return;
    const int g = seg_of(out_edge_ptr, G, e);
    const int64_t slot = edge_ptr_all[ids[g]] + (e - out_edge_ptr[g]);
    const int64_t off = out_node_ptr[g];
    edge_src_slot[e] = slot;
    edge_index[e] = edge_local_all[slot] + off;
    edge_index[E + e] = edge_local_all[E_all + slot] + off;
}

}  // namespace gsat

extern "C" {

int gsat_collate(const int64_t* graph_ids, int64_t num_graphs, const int64_t* node_ptr_all, const int64_t* edge_ptr_all,
                 const int64_t* edge_local_all, int64_t num_edges_all, const int64_t* out_node_ptr, const int64_t* out_edge_ptr,
                 int64_t N, int64_t E, int64_t* batch, int64_t* node_src_row, int64_t* edge_index, int64_t* edge_src_slot, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    GSAT_REQUIRE(num_graphs >= 0 && num_graphs < (1ll << 31) && N >= 0 && E >= 0, GSAT_ERR_ARG, "gsat_collate: bad extents");
    if (num_graphs == 0) return GSAT_OK;
    GSAT_REQUIRE(graph_ids && node_ptr_all && edge_ptr_all && out_node_ptr && out_edge_ptr, GSAT_ERR_ARG, "gsat_collate: null pointer");
    if (N > 0) {
        GSAT_REQUIRE(batch && node_src_row, GSAT_ERR_ARG, "gsat_collate: null node outputs");
        gsat::k_collate_nodes<<<GRID1(N)>>>(graph_ids, node_ptr_all, out_node_ptr, (int)num_graphs, N, batch, node_src_row);
    }
    if (E > 0) {
        GSAT_REQUIRE(edge_local_all && edge_index && edge_src_slot, GSAT_ERR_ARG, "gsat_collate: null edge outputs");
        gsat::k_collate_edges<<<GRID1(E)>>>(graph_ids, edge_ptr_all, out_edge_ptr, out_node_ptr, edge_local_all, num_edges_all, (int)num_graphs, E,
                                            edge_index, edge_src_slot);
    }
    GSAT_LAUNCH_CHECK();
    return GSAT_OK;
}

}  // extern "C"

// ------------------------------------------------------------------------------------------------
// Line ("dual") graph of the fork: one dual node per DIRECTED primal edge; two dual nodes are joined, in both
// directions, when their primal edges leave the same node (src/datasets/mutag_dual.py:345-377, `group_by_first`).
// The reference builds it with O(sum deg^2) Python loops per dataset; here one kernel over the by-source CSR.
// Output order = the reference's: source nodes ascending, then pairs (i < j) in edge order, (e_i,e_j) then (e_j,e_i).
// ------------------------------------------------------------------------------------------------
namespace gsat {

__global__ void k_pair_counts(const int32_t* __restrict__ rowptr, int64_t N, int64_t* __restrict__ counts) {
    int64_t n = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= N) return;
    const int64_t d = rowptr[n + 1] - rowptr[n];
    counts[n] = d * (d - 1) / 2;                       // unordered pairs of out-edges of node n
}

__global__ void k_line_graph(const int32_t* __restrict__ rowptr, const int32_t* __restrict__ eid, const int64_t* __restrict__ pair_ptr,
                             int64_t N, int64_t num_pairs, int64_t* __restrict__ dual_edge_index) {
    int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= num_pairs) return;
    int64_t lo = 0, hi = N;                             // pair_ptr[lo] <= t < pair_ptr[hi]
    while (hi - lo > 1) {
        int64_t mid = (lo + hi) >> 1;
        if (pair_ptr[mid] <= t) lo = mid; else hi = mid;
    }
    const int beg = rowptr[lo];
    const int64_t d = rowptr[lo + 1] - beg;
    int64_t r = t - pair_ptr[lo];                       // r-th pair (i < j) in row-major order of the upper triangle
    int64_t i = 0;
    // rows of the upper triangle have d-1, d-2, ... entries: find i with prefix(i) <= r < prefix(i+1)
    {
        // prefix(i) = i*d - i*(i+1)/2 ; solve by a short binary search (d < 2^31)
        int64_t a = 0, b = d - 1;
        while (b - a > 1) {
            int64_t m = (a + b) >> 1;
            if (m * d - m * (m + 1) / 2 <= r) a = m; else b = m;
        }
        i = a;
    }
    const int64_t j = i + 1 + (r - (i * d - i * (i + 1) / 2));
    const int64_t e1 = eid[beg + i], e2 = eid[beg + j];
    const int64_t E2 = 2 * num_pairs;
    dual_edge_index[2 * t] = e1;          dual_edge_index[E2 + 2 * t] = e2;
    dual_edge_index[2 * t + 1] = e2;      dual_edge_index[E2 + 2 * t + 1] = e1;
}

}  // namespace gsat

extern "C" {

int gsat_line_graph_pair_counts(const int32_t* rowptr_src, int64_t N, int64_t* counts, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    GSAT_REQUIRE(N >= 0, GSAT_ERR_ARG, "gsat_line_graph_pair_counts: bad N");
    if (N == 0) return GSAT_OK;
    GSAT_REQUIRE(rowptr_src && counts, GSAT_ERR_ARG, "gsat_line_graph_pair_counts: null pointer");
    gsat::k_pair_counts<<<GRID1(N)>>>(rowptr_src, N, counts);
    GSAT_LAUNCH_CHECK();
    return GSAT_OK;
}

int gsat_line_graph(const int32_t* rowptr_src, const int32_t* eid_by_src, const int64_t* pair_ptr, int64_t N, int64_t num_pairs,
                    int64_t* dual_edge_index, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    GSAT_REQUIRE(N >= 0 && num_pairs >= 0, GSAT_ERR_ARG, "gsat_line_graph: bad extents");
    if (num_pairs == 0) return GSAT_OK;
    GSAT_REQUIRE(rowptr_src && eid_by_src && pair_ptr && dual_edge_index, GSAT_ERR_ARG, "gsat_line_graph: null pointer");
    gsat::k_line_graph<<<GRID1(num_pairs)>>>(rowptr_src, eid_by_src, pair_ptr, N, num_pairs, dual_edge_index);
    GSAT_LAUNCH_CHECK();
    return GSAT_OK;
}

}  // extern "C"
