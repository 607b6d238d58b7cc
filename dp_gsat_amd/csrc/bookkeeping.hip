// Integer edge bookkeeping: COO -> CSR, reverse-edge permutation, segment pointers.
// Results are bit-exact against oracle/bookkeeping.py (stable orders everywhere).
// One-off per batch (cached by the host), so the sort itself is rocPRIM's radix sort; the kernels
// around it are plain coalesced integer passes.
#include "common.h"
#include <rocprim/rocprim.hpp>

namespace gsat {

constexpr int64_t GSAT_COUNTING_MAX_KEYS = 1 << 20;     // rocPRIM's own switch from merge sort to Onesweep

static thread_local char g_err[512] = {0};
void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
const char* get_error() { return g_err; }

// next seed of a device-resident stream: state = (base, counter); out = splitmix64(base + counter * golden), 63 bits.  One launch per
// seed inside a captured step (torch's graph-safe random_() costs three: the draw and two fills of its Philox offset words).
__global__ void k_seed_next(uint64_t* __restrict__ state, uint64_t* __restrict__ out) {
    const uint64_t c = state[1] + 1;
    state[1] = c;
    uint64_t z = state[0] + c * 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    z ^= z >> 31;
    out[0] = z & 0x7FFFFFFFFFFFFFFFull;
}

__global__ void k_zero_words(uint32_t* __restrict__ p, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = 0u;
}
__global__ void k_zero_words2(uint32_t* __restrict__ p, size_t n, uint32_t* __restrict__ q, size_t m) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n + m; i += (size_t)gridDim.x * blockDim.x) {
        if (i < n) p[i] = 0u; else q[i - n] = 0u;
    }
}
hipError_t zero2_async(void* p, size_t bytes_p, void* q, size_t bytes_q, hipStream_t stream) {       // two regions, one launch
    const size_t n = bytes_p / 4, m = bytes_q / 4;
    if (n + m == 0) return hipSuccess;
    const unsigned blocks = (unsigned)std::min<size_t>((n + m + 255) / 256, 256 * 16);
    k_zero_words2<<<blocks, 256, 0, stream>>>(static_cast<uint32_t*>(p), n, static_cast<uint32_t*>(q), m);
    return hipGetLastError();
}
hipError_t zero_async(void* p, size_t bytes, hipStream_t stream) {
    const size_t n = bytes / 4;
    if (n == 0) return hipSuccess;
    const unsigned blocks = (unsigned)std::min<size_t>((n + 255) / 256, 256 * 16);
    k_zero_words<<<blocks, 256, 0, stream>>>(static_cast<uint32_t*>(p), n);
    return hipGetLastError();
}

static int bits_for(uint64_t max_value) {   // number of low bits that can be set in [0, max_value]
    int b = 1;
    while (b < 64 && (max_value >> b) != 0) ++b;
    return b;
}

__global__ void k_make_row_keys(const int64_t* __restrict__ rows, int64_t E, int64_t num_rows,
                                uint32_t* __restrict__ keys, int32_t* __restrict__ ids, int32_t* err) {
    int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= E) return;
    int64_t r = rows[e];
    if (r < 0 || r >= num_rows) { atomicAdd(err, 1); r = num_rows - 1; }
    keys[e] = (uint32_t)r;
    ids[e] = (int32_t)e;
}

// ptr[r] = first sorted position whose key is >= r  (r in [0, num_rows])
template <class K>
__global__ void k_lower_bounds(const K* __restrict__ sorted, int64_t n, int64_t num_rows, int32_t* __restrict__ ptr) {
    int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r > num_rows) return;
    int64_t lo = 0, hi = n;
    while (lo < hi) {
        int64_t mid = (lo + hi) >> 1;
        if ((int64_t)sorted[mid] < r) lo = mid + 1; else hi = mid;
    }
    ptr[r] = (int32_t)lo;
}

__global__ void k_gather_narrow(const int64_t* __restrict__ src, const int32_t* __restrict__ perm, int64_t n,
                                int32_t* __restrict__ out) {
    int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k < n) out[k] = (int32_t)src[perm[k]];
}

__global__ void k_make_edge_keys2(const int64_t* __restrict__ ei, int64_t E, int64_t N, int key_bits, uint64_t* __restrict__ k,
                                  int32_t* __restrict__ ids) {
    int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= E) return;
    const uint64_t s = (uint64_t)ei[e], d = (uint64_t)ei[E + e];
    k[e] = s * (uint64_t)N + d;
    k[E + e] = (d * (uint64_t)N + s) | (1ull << key_bits);
    ids[e] = (int32_t)e;
    ids[E + e] = (int32_t)e;
}

__global__ void k_pair_reverse2(const uint64_t* __restrict__ sorted, const int32_t* __restrict__ pq, int64_t E, int key_bits,
                                int32_t* __restrict__ rev, int32_t* __restrict__ flags) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= E) return;
    rev[pq[E + i]] = pq[i];
    if (sorted[i] != (sorted[E + i] & ~(1ull << key_bits))) atomicAdd(&flags[1], 1);
}

// ---- reverse-edge permutation from the two CSRs (no sort) ---------------------------------------------------------------------
// The k-th copy (by edge id) of (s,d) pairs with the k-th copy of (d,s) -- the pairing of the stable sort above.  Copies of (s,d) sit
// in the out-row of s and in the in-row of d, both in edge-id order: count the earlier ones in the SHORTER row, then walk the shorter
// of in-row(s) / out-row(d) to the k-th copy of (d,s).  Hub-leaf edges cost the leaf's degree; 3 launches instead of a 2E-key sort.
__global__ void k_rev_from_csr(const int32_t* __restrict__ src32, const int32_t* __restrict__ dst32, const int32_t* __restrict__ rp_dst,
                               const int32_t* __restrict__ src_by_dst, const int32_t* __restrict__ eid_by_dst,
                               const int32_t* __restrict__ rp_src, const int32_t* __restrict__ dst_by_src,
                               const int32_t* __restrict__ eid_by_src, int64_t E, int32_t* __restrict__ rev, int32_t* __restrict__ flags) {
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= E) return;
    const int s = src32[e], d = dst32[e];
    const int os_b = rp_src[s], os_e = rp_src[s + 1], id_b = rp_dst[d], id_e = rp_dst[d + 1];
    int k = 0;
    if (os_e - os_b <= id_e - id_b) {
        for (int p = os_b; p < os_e && eid_by_src[p] < e; ++p) k += dst_by_src[p] == d;
    } else {
        for (int p = id_b; p < id_e && eid_by_dst[p] < e; ++p) k += src_by_dst[p] == s;
    }
    const int is_b = rp_dst[s], is_e = rp_dst[s + 1], od_b = rp_src[d], od_e = rp_src[d + 1];
    int found = -1;
    if (is_e - is_b <= od_e - od_b) {
        for (int p = is_b; p < is_e; ++p)
            if (src_by_dst[p] == d && k-- == 0) { found = eid_by_dst[p]; break; }
    } else {
        for (int p = od_b; p < od_e; ++p)
            if (dst_by_src[p] == s && k-- == 0) { found = eid_by_src[p]; break; }
    }
    if (found < 0) { atomicAdd(&flags[1], 1); found = (int)e; }        // no partner: not symmetric; rev stays in bounds
    rev[e] = found;
}

__global__ void k_finish_flags(int32_t* flags) { flags[0] = flags[1] == 0 ? 1 : 0; }

__global__ void k_check_sorted(const int64_t* __restrict__ ids, int64_t n, int64_t num_seg, int32_t* flags) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    int64_t v = ids[i];
    bool bad = v < 0 || v >= num_seg || (i > 0 && ids[i - 1] > v);
    if (bad) atomicAdd(flags, 1);
}

// order / range check, int32 copy of the ids and the segment pointers in one launch (thread i < n: element i; thread r <= num_seg:
// boundary r)
__global__ void k_segments(const int64_t* __restrict__ ids, int64_t n, int64_t num_seg, int32_t* flags, int32_t* __restrict__ ids32,
                           int32_t* __restrict__ ptr) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) {
        int64_t v = ids[i];
        bool bad = v < 0 || v >= num_seg || (i > 0 && ids[i - 1] > v);
        if (bad) atomicAdd(flags, 1);
        ids32[i] = (int32_t)(v < 0 ? 0 : (v >= num_seg ? num_seg - 1 : v));      // clamped: row_seg lookups stay in bounds
    }
    if (i <= num_seg) {
        int64_t lo = 0, hi = n;
        while (lo < hi) {
            int64_t mid = (lo + hi) >> 1;
            if (ids[mid] < i) lo = mid + 1; else hi = mid;
        }
        ptr[i] = (int32_t)lo;
    }
}

__global__ void k_gather_i64(const int64_t* __restrict__ table, int64_t table_len, const int64_t* __restrict__ index, int64_t n,
                             int64_t* __restrict__ out) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    int64_t j = index[i];
    j = j < 0 ? 0 : (j >= table_len ? table_len - 1 : j);       // clamped like the CSR builders (range errors are counted there)
    out[i] = table[j];
}

struct ChunkCount {      // number of GSAT_LONG_ROW_EDGES-sized chunks of row r (0 for short rows and for r == num_rows)
    const int32_t* rowptr;
    int num_rows;
    __host__ __device__ int operator()(int r) const {
        if (r >= num_rows) return 0;
        const int deg = rowptr[r + 1] - rowptr[r];
        return deg > GSAT_LONG_ROW_EDGES ? (deg + GSAT_LONG_ROW_EDGES - 1) / GSAT_LONG_ROW_EDGES : 0;
    }
};

static size_t chunk_scan_temp_bytes(int64_t n) {
    size_t tb = 0;
    auto in = rocprim::make_transform_iterator(rocprim::make_counting_iterator<int>(0), ChunkCount{nullptr, 0});
    int32_t* out = nullptr;
    (void)rocprim::exclusive_scan(nullptr, tb, in, out, 0, (size_t)(n + 1), rocprim::plus<int>(), (hipStream_t)0);
    return align_up(tb, 256) + 256;
}

// ---- both CSRs of a batch from ONE sort ----------------------------------------------------------------------------
// keys[i] = dst[i] for i < E (by-destination half), N + src[i - E] for i >= E (by-source half); ids = edge id.  A stable sort
// leaves the halves in [0,E) and [E,2E), each ordered by (row, edge id) -- the same order two separate sorts produce -- in
// half the launches (the builds are chains of ~5 us dependent launches, not bandwidth).
__global__ void k_make_pair_keys(const int64_t* __restrict__ ei, int64_t E, int64_t N, uint32_t* __restrict__ keys,
                                 int32_t* __restrict__ ids, int32_t* __restrict__ src32, int32_t* __restrict__ dst32, int32_t* err) {
    int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= E) return;
    int64_t s = ei[e], d = ei[E + e];
    if (s < 0 || s >= N) { atomicAdd(err, 1); s = N - 1; }
    if (d < 0 || d >= N) { atomicAdd(err, 1); d = N - 1; }
    keys[e] = (uint32_t)d;
    keys[E + e] = (uint32_t)(N + s);
    ids[e] = (int32_t)e;
    ids[E + e] = (int32_t)e;
    if (src32) src32[e] = (int32_t)s;        // the clamped ids: every consumer stays in bounds even if the host never reads err[0]
    if (dst32) dst32[e] = (int32_t)d;
}

__global__ void k_pair_rowptrs(const uint32_t* __restrict__ sorted, int64_t E, int64_t N, int32_t* __restrict__ rowptr_dst,
                               int32_t* __restrict__ rowptr_src) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i > 2 * N + 1) return;
    const bool second = i > N;
    const int64_t r = second ? i - (N + 1) : i;                 // row of its half, in [0, N]
    const int64_t target = second ? N + r : r;
    int64_t lo = 0, hi = 2 * E;
    while (lo < hi) {
        int64_t mid = (lo + hi) >> 1;
        if ((int64_t)sorted[mid] < target) lo = mid + 1; else hi = mid;
    }
    if (second) rowptr_src[r] = (int32_t)(lo - E); else rowptr_dst[r] = (int32_t)lo;
}

__global__ void k_pair_gather(const int64_t* __restrict__ ei, const int32_t* __restrict__ perm, int64_t E, int64_t num_nodes,
                              const int32_t* __restrict__ rowptr_dst, int32_t* __restrict__ src_by_dst, int32_t* __restrict__ eid_by_dst,
                              int32_t* __restrict__ dst_by_src, int32_t* __restrict__ eid_by_src, int32_t* __restrict__ slot_dst_of_srcslot) {
    int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= 2 * E) return;
    const int32_t e = perm[k];
    // out-of-range ids are clamped to N-1 exactly as k_make_pair_keys clamped their sort keys (and counted them in err_flag[0]),
    // so the CSRs stay self-consistent and memory-safe whether or not the host reads the counter
    if (k < E) {
        const int64_t sraw = ei[e];
        eid_by_dst[k] = e;
        src_by_dst[k] = (int32_t)((sraw < 0 || sraw >= num_nodes) ? num_nodes - 1 : sraw);
        return;
    }
    const int64_t draw = ei[E + e];
    const int32_t d = (int32_t)((draw < 0 || draw >= num_nodes) ? num_nodes - 1 : draw);
    eid_by_src[k - E] = e;
    dst_by_src[k - E] = d;
    // slot of edge e in the by-destination CSR: row d holds its edge ids in ascending order in perm[rowptr_dst[d] .. rowptr_dst[d+1])
    int lo = rowptr_dst[d], hi = rowptr_dst[d + 1];
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (perm[mid] < e) lo = mid + 1; else hi = mid;
    }
    slot_dst_of_srcslot[k - E] = lo;
}

// ---- the same two CSRs by counting instead of sorting (batches up to GSAT_COUNTING_MAX_KEYS keys) ---------------------------------
// Below 1 M items rocPRIM's radix_sort_pairs is an iterated merge sort: 16 dependent launches (~80 us) for a molecule batch's 2e5 keys,
// the largest single piece of the per-batch bookkeeping (its Onesweep alternative zeroes its histograms with hipMemsetAsync, which a
// captured graph on ROCm 7.2 does not order against the neighbouring replay).  The keys are row numbers, so:
//   count   cnt[key]++                                   (integer atomics: the totals do not depend on the order)
//   scan    first[key] = exclusive sum of cnt            (rocPRIM, two launches)
//   place   slot[first[key] + --cnt[key]] = edge id      (arbitrary order inside a row)
//   order   every row's edge ids ascending               (= the stable order of the sort): one thread per ENTRY counts the smaller ids
//           of its row (rows of <= 1024 entries); longer rows are listed and ordered by a workgroup each: bitonic network in LDS per
//           4096-entry chunk, chunks merged by rank
// The result is the unique (row, edge id) order, bit-identical to the sort's, in 7 launches.  gsat_build_csr (one CSR; the edge-mode
// extractor's edges-by-graph order) runs the same steps on its single key column.
constexpr int COUNT_ELEM_ROW = 1024;
constexpr int COUNT_CHUNK = 4096;
constexpr int COUNT_LONG_BLOCK = 256;

__global__ void k_pair_count(const int64_t* __restrict__ ei, int64_t E, int64_t N, int32_t* __restrict__ cnt,
                             int32_t* __restrict__ src32, int32_t* __restrict__ dst32, int32_t* err) {
    int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= E) return;
    int64_t s = ei[e], d = ei[E + e];
    if (s < 0 || s >= N) { atomicAdd(err, 1); s = N - 1; }
    if (d < 0 || d >= N) { atomicAdd(err, 1); d = N - 1; }
    atomicAdd(&cnt[d], 1);
    atomicAdd(&cnt[N + s], 1);
    if (src32) src32[e] = (int32_t)s;
    if (dst32) dst32[e] = (int32_t)d;
}

__global__ void k_pair_place(const int64_t* __restrict__ ei, int64_t E, int64_t N, const int64_t* __restrict__ first,
                             int32_t* __restrict__ cnt, int32_t* __restrict__ slot, int32_t* __restrict__ rowkey) {
    int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= E) return;
    int64_t s = ei[e], d = ei[E + e];
    s = (s < 0 || s >= N) ? N - 1 : s;
    d = (d < 0 || d >= N) ? N - 1 : d;
    const int pd = (int32_t)first[d] + atomicSub(&cnt[d], 1) - 1, ps = (int32_t)first[N + s] + atomicSub(&cnt[N + s], 1) - 1;
    slot[pd] = (int32_t)e; rowkey[pd] = (int32_t)d;
    slot[ps] = (int32_t)e; rowkey[ps] = (int32_t)(N + s);
}

// single key column (gsat_build_csr).  Its typical input -- the graph id of every edge of a collated batch -- is long runs of one key:
// a wave folds each run of equal ADJACENT keys into one atomic (41 same-address atomics per graph serialise in L2 otherwise: 14 -> 4 us
// per launch at C4).  Equal keys that are not adjacent simply issue separate atomics, so any input works.
struct KeyRun { bool head; int head_lane, len, offset; };
__device__ __forceinline__ KeyRun key_run(int key, bool valid) {
    const int lane = threadIdx.x & 63;
    const int prev = __shfl_up(key, 1, 64);
    const unsigned long long vmask = __ballot(valid);                 // a prefix of the wave: only the tail of the grid is invalid
    const bool head = valid && (lane == 0 || prev != key);
    const unsigned long long heads = __ballot(head);
    KeyRun r;
    r.head = head;
    const unsigned long long upto = lane == 63 ? ~0ull : ((1ull << (lane + 1)) - 1);
    r.head_lane = valid ? 63 - __clzll(heads & upto) : 0;
    const unsigned long long above = r.head_lane == 63 ? 0ull : (heads & ~((1ull << (r.head_lane + 1)) - 1));
    const int next = above ? __ffsll((long long)above) - 1 : (vmask ? 64 - __clzll(vmask) : 0);
    r.len = next - r.head_lane;
    r.offset = lane - r.head_lane;
    return r;
}

__global__ void k_rows_count(const int64_t* __restrict__ rows, int64_t E, int64_t R, int32_t* __restrict__ cnt, int32_t* err) {
    int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const bool valid = e < E;
    int64_t r = valid ? rows[e] : 0;
    if (valid && (r < 0 || r >= R)) { atomicAdd(err, 1); r = R - 1; }
    const KeyRun run = key_run((int)r, valid);
    if (run.head) atomicAdd(&cnt[r], run.len);
}

__global__ void k_rows_place(const int64_t* __restrict__ rows, int64_t E, int64_t R, const int64_t* __restrict__ first,
                             int32_t* __restrict__ cnt, int32_t* __restrict__ slot, int32_t* __restrict__ rowkey) {
    int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const bool valid = e < E;
    int64_t r = valid ? rows[e] : 0;
    r = (r < 0 || r >= R) ? R - 1 : r;
    const KeyRun run = key_run((int)r, valid);
    int old = run.head ? atomicSub(&cnt[r], run.len) : 0;
    old = __shfl(old, run.head_lane, 64);
    if (!valid) return;
    const int p = (int32_t)first[r] + old - 1 - run.offset;
    slot[p] = (int32_t)e; rowkey[p] = (int32_t)r;
}

// thread i <= R (R key rows + the closing entry): row pointers (PAIR: of both CSRs, with the hub-chunk pointers), long rows listed;
// thread i < total: entry i takes its place in its row = the number of smaller edge ids in that row
template <bool PAIR>
__global__ void k_count_order(const int64_t* __restrict__ first, int64_t R, int64_t total, const int32_t* __restrict__ slot,
                              const int32_t* __restrict__ rowkey, int32_t* __restrict__ perm, int64_t E, int64_t N,
                              int32_t* __restrict__ rowptr_dst, int32_t* __restrict__ rowptr_src, int32_t* __restrict__ chunk_ptr_dst,
                              int32_t* __restrict__ chunk_ptr_src, int32_t* __restrict__ status, int32_t* __restrict__ long_rows,
                              int32_t* __restrict__ long_count) {
    const int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k <= R) {
        const int64_t f = first[k];
        const int beg = (int32_t)f;
        if (PAIR) {
            const int chunks = (int32_t)(f >> 32);
            if (k <= N) { rowptr_dst[k] = beg; chunk_ptr_dst[k] = chunks; }
            if (k == N) status[1] = chunks;              // hub-chunk totals next to the range-error counter: one host read gets all
            if (k >= N) {
                const int c0 = (int32_t)(first[N] >> 32);
                rowptr_src[k - N] = beg - (int32_t)E;
                chunk_ptr_src[k - N] = chunks - c0;
                if (k == 2 * N) status[2] = chunks - c0;
            }
        } else {
            rowptr_dst[k] = beg;
        }
        if (k < R && (int32_t)first[k + 1] - beg > COUNT_ELEM_ROW)
            long_rows[atomicAdd(long_count, 1)] = (int32_t)k;    // order of the list is irrelevant: every listed row is ordered on its own
    }
    if (k < total) {
        const int r = rowkey[k];
        const int beg = (int32_t)first[r], end = (int32_t)first[r + 1];
        if (end - beg > COUNT_ELEM_ROW) return;
        const int v = slot[k];
        int rank = 0;
        for (int j = beg; j < end; ++j) rank += slot[j] < v;
        perm[beg + rank] = v;
    }
}

__global__ __launch_bounds__(COUNT_LONG_BLOCK) void k_pair_order_long(const int64_t* __restrict__ first, int32_t* __restrict__ slot,
                                                                       int32_t* __restrict__ perm, const int32_t* __restrict__ long_rows,
                                                                       const int32_t* __restrict__ long_count) {
    __shared__ int32_t buf[COUNT_CHUNK];
    const int nlong = *long_count;
    for (int q = blockIdx.x; q < nlong; q += gridDim.x) {
        const int k = long_rows[q];
        const int beg = (int32_t)first[k], end = (int32_t)first[k + 1], deg = end - beg;
        const int nchunks = (deg + COUNT_CHUNK - 1) / COUNT_CHUNK;
        for (int c = 0; c < nchunks; ++c) {
            const int cb = beg + c * COUNT_CHUNK, len = min(COUNT_CHUNK, end - cb);
            int p2 = 64;
            while (p2 < len) p2 <<= 1;
            for (int i = threadIdx.x; i < p2; i += COUNT_LONG_BLOCK) buf[i] = i < len ? slot[cb + i] : 0x7fffffff;
            __syncthreads();
            for (int size = 2; size <= p2; size <<= 1)
                for (int stride = size >> 1; stride > 0; stride >>= 1) {
                    for (int t = threadIdx.x; t < (p2 >> 1); t += COUNT_LONG_BLOCK) {
                        const int lo = 2 * t - (t & (stride - 1)), hi = lo + stride;
                        const bool up = (lo & size) == 0;
                        const int a = buf[lo], b = buf[hi];
                        if ((a > b) == up) { buf[lo] = b; buf[hi] = a; }
                    }
                    __syncthreads();
                }
            // a single chunk is final; several chunks go back to slot[] chunk-sorted for the rank merge below
            int32_t* outp = nchunks == 1 ? perm : slot;
            for (int i = threadIdx.x; i < len; i += COUNT_LONG_BLOCK) outp[cb + i] = buf[i];
            __syncthreads();
        }
        if (nchunks == 1) continue;
        // edge ids are distinct: final position = own position in its chunk + the number of smaller ids in every other chunk
        for (int i = threadIdx.x; i < deg; i += COUNT_LONG_BLOCK) {
            const int v = slot[beg + i], own = i / COUNT_CHUNK;
            int r = i - own * COUNT_CHUNK;
            for (int c = 0; c < nchunks; ++c) {
                if (c == own) continue;
                const int32_t* base = slot + beg + c * COUNT_CHUNK;
                int lo = 0, hi = min(COUNT_CHUNK, deg - c * COUNT_CHUNK);
                while (lo < hi) {
                    const int mid = (lo + hi) >> 1;
                    if (base[mid] < v) lo = mid + 1; else hi = mid;
                }
                r += lo;
            }
            perm[beg + r] = v;
        }
        __syncthreads();
    }
}

// scanned value of key row k: entries in the low word, hub chunks (ChunkCount's rule) in the high word -- one scan yields the row
// pointers and the chunk pointers of both CSRs
struct CountAndChunks {
    const int32_t* cnt;
    __host__ __device__ int64_t operator()(int k) const {
        const int deg = cnt[k];
        const int64_t chunks = deg > GSAT_LONG_ROW_EDGES ? (deg + GSAT_LONG_ROW_EDGES - 1) / GSAT_LONG_ROW_EDGES : 0;
        return (int64_t)deg + (chunks << 32);
    }
};

static size_t count_scan_items_temp_bytes(int64_t items) {
    size_t tb = 0;
    auto in = rocprim::make_transform_iterator(rocprim::make_counting_iterator<int>(0), CountAndChunks{nullptr});
    int64_t* p = nullptr;
    (void)rocprim::exclusive_scan(nullptr, tb, in, p, (int64_t)0, (size_t)std::max<int64_t>(items, 1), rocprim::plus<int64_t>(), (hipStream_t)0);
    return align_up(tb, 256) + 256;
}
static size_t count_scan_temp_bytes(int64_t n) { return count_scan_items_temp_bytes(2 * n + 1); }

static bool counting_build(int64_t E) {
    static const int on = [] { const char* e = getenv("GSAT_CSR_COUNTING"); return e ? atoi(e) : 1; }();
    return on && 2 * E <= GSAT_COUNTING_MAX_KEYS;
}

struct ChunkCount2 {     // ChunkCount over the rows of both CSRs laid end to end: [0, N] by-destination, [N+1, 2N+1] by-source
    const int32_t* rp_dst;
    const int32_t* rp_src;
    int num_rows;
    __host__ __device__ int operator()(int i) const {
        const int32_t* rp = i > num_rows ? rp_src : rp_dst;
        const int r = i > num_rows ? i - (num_rows + 1) : i;
        if (r >= num_rows) return 0;
        const int deg = rp[r + 1] - rp[r];
        return deg > GSAT_LONG_ROW_EDGES ? (deg + GSAT_LONG_ROW_EDGES - 1) / GSAT_LONG_ROW_EDGES : 0;
    }
};

__global__ void k_split_chunk_ptrs(const int32_t* __restrict__ scan, int64_t N, int32_t* __restrict__ chunk_ptr_dst,
                                   int32_t* __restrict__ chunk_ptr_src, int32_t* __restrict__ status) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i > N) return;
    const int32_t d = scan[i], s = scan[N + 1 + i] - scan[N + 1];
    chunk_ptr_dst[i] = d;
    chunk_ptr_src[i] = s;
    if (i == N) { status[1] = d; status[2] = s; }      // hub-chunk totals next to the range-error counter: one host read gets all
}

static size_t chunk2_scan_temp_bytes(int64_t n) {
    size_t tb = 0;
    auto in = rocprim::make_transform_iterator(rocprim::make_counting_iterator<int>(0), ChunkCount2{nullptr, nullptr, 0});
    int32_t* out = nullptr;
    (void)rocprim::exclusive_scan(nullptr, tb, in, out, 0, (size_t)(2 * n + 2), rocprim::plus<int>(), (hipStream_t)0);
    return align_up(tb, 256) + 256;
}

template <class K>
static size_t sort_temp_bytes(int64_t n) {
    size_t tb = 0;
    K* k = nullptr;
    int32_t* v = nullptr;
    (void)rocprim::radix_sort_pairs(nullptr, tb, k, k, v, v, (size_t)(n > 0 ? n : 1), 0, (unsigned)(8 * sizeof(K)), (hipStream_t)0);
    return align_up(tb, 256) + 256;
}

}  // namespace gsat

using namespace gsat;

extern "C" {

int gsat_abi_version(void) { return GSAT_ABI_VERSION; }

int gsat_seed_next(uint64_t* state, uint64_t* out, void* stream) {
    GSAT_REQUIRE(state && out, GSAT_ERR_ARG, "gsat_seed_next: null pointer");
    gsat::k_seed_next<<<1, 1, 0, (hipStream_t)stream>>>(state, out);
    GSAT_LAUNCH_CHECK();
    return GSAT_OK;
}
const char* gsat_last_error(void) { return gsat::get_error(); }

size_t gsat_csr_workspace_bytes(int64_t E, int64_t num_rows) {
    size_t e = (size_t)(E > 0 ? E : 1), r = (size_t)(num_rows > 0 ? num_rows + 2 : 2);
    return 3 * align_up(e * 4, 256) + 3 * align_up(r * 4, 256) + std::max(sort_temp_bytes<uint32_t>(E), count_scan_items_temp_bytes(num_rows + 1));
}

size_t gsat_rev_workspace_bytes(int64_t E) {
    size_t e2 = (size_t)(E > 0 ? 2 * E : 2);
    return 2 * align_up(e2 * 8, 256) + 2 * align_up(e2 * 4, 256) + sort_temp_bytes<uint64_t>(2 * E);
}

int gsat_build_csr(const int64_t* rows, const int64_t* other, int64_t E, int64_t num_rows, int32_t* rowptr,
                   int32_t* other_sorted, int32_t* perm, int32_t* err_flag, void* workspace, size_t ws_bytes,
                   void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    GSAT_REQUIRE(E >= 0 && num_rows >= 0 && rowptr && err_flag, GSAT_ERR_ARG, "gsat_build_csr: bad argument");
    GSAT_REQUIRE(E < (1ll << 31) && num_rows < (1ll << 31), GSAT_ERR_UNSUPPORTED, "gsat_build_csr: >2^31 entries");
    if (E == 0) {
        GSAT_CHECK_HIP(gsat::zero_async(rowptr, (size_t)(num_rows + 1) * sizeof(int32_t), stream));
        return GSAT_OK;
    }
    GSAT_REQUIRE(rows && perm && num_rows > 0, GSAT_ERR_ARG, "gsat_build_csr: null rows/perm");
    Arena ar(workspace, ws_bytes);
    uint32_t* keys_in = ar.take<uint32_t>(E);
    uint32_t* keys_out = ar.take<uint32_t>(E);
    int32_t* ids = ar.take<int32_t>(E);
    int32_t* cnt = ar.take<int32_t>(num_rows + 2);       // counting build: [0, R) row counts, [R] = 0 closes the scan, [R+1] = long rows listed
    int64_t* first = ar.take<int64_t>(num_rows + 2);
    size_t tb = std::max(sort_temp_bytes<uint32_t>(E), count_scan_items_temp_bytes(num_rows + 1));
    char* temp = ar.take<char>(tb);
    GSAT_REQUIRE(ar.ok() && temp, GSAT_ERR_WORKSPACE, "gsat_build_csr: workspace %zu < %zu", ws_bytes, ar.off);

    const int B = 256;
    if (counting_build(E / 2)) {                         // E keys <= 2^20: count / scan / place / order (see the pair build)
        int32_t* slot = ids;
        int32_t* rowkey = reinterpret_cast<int32_t*>(keys_out);
        int32_t* long_rows = reinterpret_cast<int32_t*>(keys_in);
        GSAT_CHECK_HIP(gsat::zero_async(cnt, (size_t)(num_rows + 2) * sizeof(int32_t), stream));
        k_rows_count<<<ceil_div(E, B), B, 0, stream>>>(rows, E, num_rows, cnt, err_flag);
        GSAT_LAUNCH_CHECK();
        size_t ts = count_scan_items_temp_bytes(num_rows + 1);
        auto cin = rocprim::make_transform_iterator(rocprim::make_counting_iterator<int>(0), CountAndChunks{cnt});
        GSAT_CHECK_HIP(rocprim::exclusive_scan(temp, ts, cin, first, (int64_t)0, (size_t)(num_rows + 1), rocprim::plus<int64_t>(), stream));
        k_rows_place<<<ceil_div(E, B), B, 0, stream>>>(rows, E, num_rows, first, cnt, slot, rowkey);
        GSAT_LAUNCH_CHECK();
        k_count_order<false><<<ceil_div(std::max<int64_t>(E, num_rows + 1), B), B, 0, stream>>>(
            first, num_rows, E, slot, rowkey, perm, E, num_rows, rowptr, nullptr, nullptr, nullptr, nullptr, long_rows, cnt + num_rows + 1);
        GSAT_LAUNCH_CHECK();
        k_pair_order_long<<<256, COUNT_LONG_BLOCK, 0, stream>>>(first, slot, perm, long_rows, cnt + num_rows + 1);
        GSAT_LAUNCH_CHECK();
    } else {
        k_make_row_keys<<<ceil_div(E, B), B, 0, stream>>>(rows, E, num_rows, keys_in, ids, err_flag);
        GSAT_LAUNCH_CHECK();
        int end_bit = bits_for((uint64_t)(num_rows - 1));
        GSAT_CHECK_HIP(rocprim::radix_sort_pairs(temp, tb, keys_in, keys_out, ids, perm, (size_t)E, 0, (unsigned)end_bit, stream));
        k_lower_bounds<uint32_t><<<ceil_div(num_rows + 1, B), B, 0, stream>>>(keys_out, E, num_rows, rowptr);
        GSAT_LAUNCH_CHECK();
    }
    if (other && other_sorted) {
        k_gather_narrow<<<ceil_div(E, B), B, 0, stream>>>(other, perm, E, other_sorted);
        GSAT_LAUNCH_CHECK();
    }
    return GSAT_OK;
}

int gsat_reverse_edge_perm(const int64_t* edge_index, int64_t E, int64_t N, int32_t* rev, int32_t* flags,
                           void* workspace, size_t ws_bytes, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    GSAT_REQUIRE(E >= 0 && N >= 0 && flags, GSAT_ERR_ARG, "gsat_reverse_edge_perm: bad argument");
    GSAT_REQUIRE(E < (1ll << 30) && N < (1ll << 31), GSAT_ERR_UNSUPPORTED, "gsat_reverse_edge_perm: >2^30 edges");
    GSAT_CHECK_HIP(gsat::zero_async(flags, 2 * sizeof(int32_t), stream));
    if (E == 0) {
        k_finish_flags<<<1, 1, 0, stream>>>(flags);
        GSAT_LAUNCH_CHECK();
        return GSAT_OK;
    }
    GSAT_REQUIRE(edge_index && rev && N > 0, GSAT_ERR_ARG, "gsat_reverse_edge_perm: null pointer");
    // one stable sort of 2E keys: [0,E) the edge keys src*N+dst, [E,2E) the transposed keys dst*N+src with a half bit on top;
    // sorted position i of the first half pairs with position i of the second half (same launches as ONE sort)
    Arena ar(workspace, ws_bytes);
    uint64_t* kin = ar.take<uint64_t>(2 * E);
    uint64_t* kout = ar.take<uint64_t>(2 * E);
    int32_t* ids = ar.take<int32_t>(2 * E);
    int32_t* pq = ar.take<int32_t>(2 * E);
    size_t tb = sort_temp_bytes<uint64_t>(2 * E);
    char* temp = ar.take<char>(tb);
    GSAT_REQUIRE(ar.ok() && temp, GSAT_ERR_WORKSPACE, "gsat_reverse_edge_perm: workspace %zu < %zu", ws_bytes, ar.off);
    const int B = 256;
    const int key_bits = bits_for((uint64_t)N * (uint64_t)N - 1);
    GSAT_REQUIRE(key_bits < 63, GSAT_ERR_UNSUPPORTED, "gsat_reverse_edge_perm: N too large");
    k_make_edge_keys2<<<ceil_div(E, B), B, 0, stream>>>(edge_index, E, N, key_bits, kin, ids);
    GSAT_LAUNCH_CHECK();
    GSAT_CHECK_HIP(rocprim::radix_sort_pairs(temp, tb, kin, kout, ids, pq, (size_t)(2 * E), 0, (unsigned)(key_bits + 1), stream));
    k_pair_reverse2<<<ceil_div(E, B), B, 0, stream>>>(kout, pq, E, key_bits, rev, flags);
    GSAT_LAUNCH_CHECK();
    k_finish_flags<<<1, 1, 0, stream>>>(flags);
    GSAT_LAUNCH_CHECK();
    return GSAT_OK;
}

int gsat_reverse_edge_perm_csr(const int32_t* src32, const int32_t* dst32, const int32_t* rowptr_dst, const int32_t* src_by_dst,
                               const int32_t* eid_by_dst, const int32_t* rowptr_src, const int32_t* dst_by_src, const int32_t* eid_by_src,
                               int64_t E, int64_t N, int32_t* rev, int32_t* flags, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    GSAT_REQUIRE(E >= 0 && N >= 0 && flags, GSAT_ERR_ARG, "gsat_reverse_edge_perm_csr: bad argument");
    GSAT_REQUIRE(E < (1ll << 31), GSAT_ERR_UNSUPPORTED, "gsat_reverse_edge_perm_csr: >2^31 edges");
    GSAT_CHECK_HIP(gsat::zero_async(flags, 2 * sizeof(int32_t), stream));
    if (E > 0) {
        GSAT_REQUIRE(src32 && dst32 && rowptr_dst && src_by_dst && eid_by_dst && rowptr_src && dst_by_src && eid_by_src && rev && N > 0,
                     GSAT_ERR_ARG, "gsat_reverse_edge_perm_csr: null pointer");
        k_rev_from_csr<<<ceil_div(E, 256), 256, 0, stream>>>(src32, dst32, rowptr_dst, src_by_dst, eid_by_dst, rowptr_src, dst_by_src, eid_by_src,
                                                             E, rev, flags);
        GSAT_LAUNCH_CHECK();
    }
    k_finish_flags<<<1, 1, 0, stream>>>(flags);
    GSAT_LAUNCH_CHECK();
    return GSAT_OK;
}

size_t gsat_csr_pair_workspace_bytes(int64_t E, int64_t N) {
    const size_t e2 = (size_t)(E > 0 ? 2 * E : 2), n2 = (size_t)(N > 0 ? 2 * N + 2 : 2);
    return 4 * align_up(e2 * 4, 256) + align_up((size_t)(E > 0 ? E : 1) * 4, 256) + 4 * align_up(n2 * 4, 256) +
           std::max(sort_temp_bytes<uint32_t>(2 * E), count_scan_temp_bytes(N)) + chunk2_scan_temp_bytes(N);
}

int gsat_build_csr_pair(const int64_t* edge_index, int64_t E, int64_t N, int32_t* rowptr_dst, int32_t* src_by_dst, int32_t* eid_by_dst,
                        int32_t* rowptr_src, int32_t* dst_by_src, int32_t* eid_by_src, int32_t* slot_dst_of_srcslot,
                        int32_t* chunk_ptr_dst, int32_t* chunk_ptr_src, int32_t* src32, int32_t* dst32, int32_t* err_flag,
                        void* workspace, size_t ws_bytes, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    GSAT_REQUIRE(E >= 0 && N >= 0 && rowptr_dst && rowptr_src && chunk_ptr_dst && chunk_ptr_src && err_flag, GSAT_ERR_ARG,
                 "gsat_build_csr_pair: bad argument");
    GSAT_REQUIRE(2 * E < (1ll << 31) && 2 * N + 2 < (1ll << 31), GSAT_ERR_UNSUPPORTED, "gsat_build_csr_pair: >2^30 entries");
    if (E == 0) {
        GSAT_CHECK_HIP(gsat::zero_async(err_flag, 4 * sizeof(int32_t), stream));
        GSAT_CHECK_HIP(gsat::zero_async(rowptr_dst, (size_t)(N + 1) * sizeof(int32_t), stream));
        GSAT_CHECK_HIP(gsat::zero_async(rowptr_src, (size_t)(N + 1) * sizeof(int32_t), stream));
        GSAT_CHECK_HIP(gsat::zero_async(chunk_ptr_dst, (size_t)(N + 1) * sizeof(int32_t), stream));
        GSAT_CHECK_HIP(gsat::zero_async(chunk_ptr_src, (size_t)(N + 1) * sizeof(int32_t), stream));
        return GSAT_OK;
    }
    GSAT_REQUIRE(edge_index && src_by_dst && eid_by_dst && dst_by_src && eid_by_src && slot_dst_of_srcslot && N > 0, GSAT_ERR_ARG,
                 "gsat_build_csr_pair: null pointer");
    Arena ar(workspace, ws_bytes);
    uint32_t* keys_in = ar.take<uint32_t>(2 * E);
    uint32_t* keys_out = ar.take<uint32_t>(2 * E);
    int32_t* ids = ar.take<int32_t>(2 * E);
    int32_t* perm = ar.take<int32_t>(2 * E);
    int32_t* scan = ar.take<int32_t>(2 * N + 2);
    int32_t* cnt = ar.take<int32_t>(2 * N + 2);          // counting build: [0, 2N) row counts, [2N] = 0 closes the scan, [2N+1] = long rows listed
    int64_t* first = ar.take<int64_t>(2 * N + 2);          // (entries, hub chunks) before each key row
    const bool counting = counting_build(E);
    size_t tb = std::max(sort_temp_bytes<uint32_t>(2 * E), count_scan_temp_bytes(N)), tc = chunk2_scan_temp_bytes(N);
    char* temp = ar.take<char>(tb);
    char* temp2 = ar.take<char>(tc);
    GSAT_REQUIRE(ar.ok() && temp && temp2, GSAT_ERR_WORKSPACE, "gsat_build_csr_pair: workspace %zu < %zu", ws_bytes, ar.off);
    const int B = 256;
    if (counting) {
        int32_t* slot = ids;                             // [2E] edge ids grouped by row, unordered inside a row
        int32_t* long_rows = reinterpret_cast<int32_t*>(keys_in);      // [rows longer than COUNT_ELEM_ROW: fewer than 2E / 1024]
        GSAT_CHECK_HIP(zero2_async(cnt, (size_t)(2 * N + 2) * sizeof(int32_t), err_flag, 4 * sizeof(int32_t), stream));      // counts + the four status words
        k_pair_count<<<ceil_div(E, B), B, 0, stream>>>(edge_index, E, N, cnt, src32, dst32, err_flag);
        GSAT_LAUNCH_CHECK();
        size_t ts = count_scan_temp_bytes(N);
        auto cin = rocprim::make_transform_iterator(rocprim::make_counting_iterator<int>(0), CountAndChunks{cnt});
        GSAT_CHECK_HIP(rocprim::exclusive_scan(temp, ts, cin, first, (int64_t)0, (size_t)(2 * N + 1), rocprim::plus<int64_t>(), stream));
        int32_t* rowkey = reinterpret_cast<int32_t*>(keys_out);
        k_pair_place<<<ceil_div(E, B), B, 0, stream>>>(edge_index, E, N, first, cnt, slot, rowkey);
        GSAT_LAUNCH_CHECK();
        k_count_order<true><<<ceil_div(std::max<int64_t>(2 * E, 2 * N + 1), B), B, 0, stream>>>(
            first, 2 * N, 2 * E, slot, rowkey, perm, E, N, rowptr_dst, rowptr_src, chunk_ptr_dst, chunk_ptr_src, err_flag, long_rows, cnt + 2 * N + 1);
        GSAT_LAUNCH_CHECK();
        k_pair_order_long<<<256, COUNT_LONG_BLOCK, 0, stream>>>(first, slot, perm, long_rows, cnt + 2 * N + 1);
        GSAT_LAUNCH_CHECK();
    } else {
        GSAT_CHECK_HIP(gsat::zero_async(err_flag, 4 * sizeof(int32_t), stream));
        k_make_pair_keys<<<ceil_div(E, B), B, 0, stream>>>(edge_index, E, N, keys_in, ids, src32, dst32, err_flag);
        GSAT_LAUNCH_CHECK();
        const int end_bit = bits_for((uint64_t)(2 * N - 1));
        GSAT_CHECK_HIP(rocprim::radix_sort_pairs(temp, tb, keys_in, keys_out, ids, perm, (size_t)(2 * E), 0, (unsigned)end_bit, stream));
        k_pair_rowptrs<<<ceil_div(2 * N + 2, B), B, 0, stream>>>(keys_out, E, N, rowptr_dst, rowptr_src);
        GSAT_LAUNCH_CHECK();
    }
    k_pair_gather<<<ceil_div(2 * E, B), B, 0, stream>>>(edge_index, perm, E, N, rowptr_dst, src_by_dst, eid_by_dst, dst_by_src, eid_by_src,
                                                        slot_dst_of_srcslot);
    GSAT_LAUNCH_CHECK();
    if (counting) return GSAT_OK;                        // chunk pointers came out of the same scan
    auto in = rocprim::make_transform_iterator(rocprim::make_counting_iterator<int>(0), ChunkCount2{rowptr_dst, rowptr_src, (int)N});
    GSAT_CHECK_HIP(rocprim::exclusive_scan(temp2, tc, in, scan, 0, (size_t)(2 * N + 2), rocprim::plus<int>(), stream));
    k_split_chunk_ptrs<<<ceil_div(N + 1, B), B, 0, stream>>>(scan, N, chunk_ptr_dst, chunk_ptr_src, err_flag);
    GSAT_LAUNCH_CHECK();
    return GSAT_OK;
}

size_t gsat_row_chunks_workspace_bytes(int64_t num_rows) { return chunk_scan_temp_bytes(num_rows); }

int gsat_row_chunks(const int32_t* rowptr, int64_t num_rows, int32_t* chunk_ptr, void* workspace, size_t ws_bytes, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    GSAT_REQUIRE(num_rows >= 0 && num_rows < (1ll << 31) - 1 && rowptr && chunk_ptr, GSAT_ERR_ARG, "gsat_row_chunks: bad argument");
    size_t tb = chunk_scan_temp_bytes(num_rows);
    GSAT_REQUIRE(workspace && ws_bytes >= tb, GSAT_ERR_WORKSPACE, "gsat_row_chunks: workspace %zu < %zu", ws_bytes, tb);
    auto in = rocprim::make_transform_iterator(rocprim::make_counting_iterator<int>(0), ChunkCount{rowptr, (int)num_rows});
    GSAT_CHECK_HIP(rocprim::exclusive_scan(workspace, tb, in, chunk_ptr, 0, (size_t)(num_rows + 1), rocprim::plus<int>(), stream));
    return GSAT_OK;
}

int gsat_segment_ptr(const int64_t* seg_ids, int64_t n, int64_t num_seg, int32_t* ptr, int32_t* flags, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    GSAT_REQUIRE(n >= 0 && num_seg >= 0 && ptr && flags, GSAT_ERR_ARG, "gsat_segment_ptr: bad argument");
    GSAT_REQUIRE(n < (1ll << 31), GSAT_ERR_UNSUPPORTED, "gsat_segment_ptr: >2^31 rows");
    const int B = 256;
    GSAT_CHECK_HIP(gsat::zero_async(flags, sizeof(int32_t), stream));
    if (n > 0) {
        GSAT_REQUIRE(seg_ids, GSAT_ERR_ARG, "gsat_segment_ptr: null ids");
        k_check_sorted<<<ceil_div(n, B), B, 0, stream>>>(seg_ids, n, num_seg, flags);
        GSAT_LAUNCH_CHECK();
    }
    k_lower_bounds<int64_t><<<ceil_div(num_seg + 1, B), B, 0, stream>>>(seg_ids, n, num_seg, ptr);
    GSAT_LAUNCH_CHECK();
    return GSAT_OK;
}

int gsat_segment_ptr32(const int64_t* seg_ids, int64_t n, int64_t num_seg, int32_t* ptr, int32_t* seg_ids32, int32_t* flags, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    GSAT_REQUIRE(n >= 0 && num_seg >= 0 && ptr && flags, GSAT_ERR_ARG, "gsat_segment_ptr32: bad argument");
    GSAT_REQUIRE(n < (1ll << 31) && num_seg < (1ll << 31), GSAT_ERR_UNSUPPORTED, "gsat_segment_ptr32: >2^31 rows");
    GSAT_REQUIRE(n == 0 || (seg_ids && seg_ids32), GSAT_ERR_ARG, "gsat_segment_ptr32: null ids");
    const int B = 256;
    k_segments<<<ceil_div(std::max<int64_t>(n, num_seg + 1), B), B, 0, stream>>>(seg_ids, n, num_seg, flags, seg_ids32, ptr);
    GSAT_LAUNCH_CHECK();
    return GSAT_OK;
}

int gsat_gather_i64(const int64_t* table, int64_t table_len, const int64_t* index, int64_t n, int64_t* out, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    GSAT_REQUIRE(n >= 0 && table_len >= 0, GSAT_ERR_ARG, "gsat_gather_i64: bad n");
    if (n == 0) return GSAT_OK;
    GSAT_REQUIRE(table && index && out && table_len > 0, GSAT_ERR_ARG, "gsat_gather_i64: null pointer / empty table");
    k_gather_i64<<<ceil_div(n, 256), 256, 0, stream>>>(table, table_len, index, n, out);
    GSAT_LAUNCH_CHECK();
    return GSAT_OK;
}

}  // extern "C"
