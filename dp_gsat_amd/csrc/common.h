// Shared host/device helpers for libgsat_hip (gfx950 / CDNA4 only).
#pragma once
#include <cstring>
#include <cstdint>
#include <cstdio>
#include <cstdarg>
#include <hip/hip_runtime.h>

#include "../../include/gsat_hip.h"

namespace gsat {

// ---- error plumbing: no exceptions cross the C ABI ------------------------------------------
void set_error(const char* fmt, ...);
const char* get_error();

#define GSAT_CHECK_HIP(expr)                                                                  \
    do {                                                                                      \
        hipError_t _e = (expr);                                                               \
        if (_e != hipSuccess) {                                                               \
            gsat::set_error("%s:%d: %s -> %s", __FILE__, __LINE__, #expr, hipGetErrorString(_e)); \
            return GSAT_ERR_HIP;                                                              \
        }                                                                                     \
    } while (0)

#define GSAT_REQUIRE(cond, code, ...)                                                         \
    do {                                                                                      \
        if (!(cond)) {                                                                        \
            gsat::set_error(__VA_ARGS__);                                                     \
            return (code);                                                                    \
        }                                                                                     \
    } while (0)

#define GSAT_LAUNCH_CHECK() GSAT_CHECK_HIP(hipGetLastError())

static inline int64_t ceil_div(int64_t a, int64_t b) { return (a + b - 1) / b; }

// Zero `bytes` (a multiple of 4) of device memory with a KERNEL.  hipMemsetAsync is not used anywhere in the library: inside a
// captured hipGraph its memset nodes were observed (ROCm 7.2, back-to-back replays of one graph) to run out of order with the
// kernels of the neighbouring replay -- a counter reset by one raced with the previous replay's readers and writers.
hipError_t zero_async(void* p, size_t bytes, hipStream_t stream);
static inline size_t align_up(size_t a, size_t b) { return (a + b - 1) / b * b; }

// Bump allocator over a caller-provided workspace (the library never allocates device memory).
struct Arena {
    char* base;
    size_t cap, off;
    Arena(void* p, size_t bytes) : base(static_cast<char*>(p)), cap(bytes), off(0) {}
    template <class T> T* take(size_t n) {
        size_t b = align_up(n * sizeof(T), 256);
        if (base == nullptr || off + b > cap) { off += b; return nullptr; }
        T* r = reinterpret_cast<T*>(base + off);
        off += b;
        return r;
    }
    bool ok() const { return base != nullptr && off <= cap; }
};

// ---- fp32 MFMA GEMM (gemm.hip): C[M,N] (+)= op(A) op(B) (+ bias) --------------------------------
// `defer` != NULL: a split-K product leaves the ordered sum of its partial slabs to the caller (slab_reduce_jobs: several products' sums
// in ONE launch); defer->nslab == 0 afterwards means nothing is pending (no split, or the sum was done here)
struct SlabJob { const float* ws; int nslab; size_t slab; int M, N; int64_t ldc; int accumulate; float* out; };
int gemm_f32(hipStream_t stream, bool a_t, bool b_t, int64_t M, int64_t N, int64_t K, const float* A, int64_t lda, const float* B,
             int64_t ldb, float* C, int64_t ldc, const float* bias, bool accumulate, float* ws, size_t ws_floats, bool allow_split = false,
             SlabJob* defer = nullptr);
int slab_reduce_jobs(hipStream_t stream, const SlabJob* jobs, int n);
size_t gemm_workspace_floats(int64_t M, int64_t N, int64_t K, bool reduce_rows);

// ---- gather-sum over a CSR (aggregate.hip), reused by the extractor backward -------------------------
int aggr_sum_fwd_impl(hipStream_t stream, const float* x, const float* self_rows, const float* att, const float* edge_emb,
                      const int32_t* rowptr, const int32_t* col, const int32_t* eid, int64_t N, int64_t E, int64_t H, float self_coef,
                      float* out, const int32_t* chunk_ptr, float* partial);

// ---- device helpers -------------------------------------------------------------------------
#ifdef __HIPCC__
constexpr int WAVE = 64;

// sum over the `width` consecutive lanes of an aligned lane group (width = power of two <= 64)
// One cross-lane step of a reduction as a DPP operand modifier of the add (no LDS crossbar round trip): v + v[lane permuted by CTRL]
template <int CTRL> __device__ __forceinline__ float dpp_add(float v) {
    const int t = __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true);
    return v + __builtin_bit_cast(float, t);
}

// sum over the WIDTH consecutive lanes of an aligned lane group; power-of-two widths leave the total in EVERY lane of the group.
// Steps inside a 16-lane row are DPP adds (quad permutes, half-row and row mirrors: ~1 VALU op each); the 16 <-> 16 step is one
// ds_swizzle, the 32 <-> 32 step one ds_bpermute -- the five dependent ds_bpermute round trips of a shuffle butterfly were the
// longest latency chain of the per-edge loops (PNA backward 76 -> ~70 us at C3).
template <int WIDTH> __device__ __forceinline__ float group_sum(float v) {
    if constexpr ((WIDTH & (WIDTH - 1)) == 0) {
        if constexpr (WIDTH >= 2) v = dpp_add<0xB1>(v);            // quad_perm [1,0,3,2]: lane ^ 1
        if constexpr (WIDTH >= 4) v = dpp_add<0x4E>(v);            // quad_perm [2,3,0,1]: lane ^ 2
        if constexpr (WIDTH >= 8) v = dpp_add<0x141>(v);           // row_half_mirror: the other quad of the 8-lane half row
        if constexpr (WIDTH >= 16) v = dpp_add<0x140>(v);          // row_mirror: the other half of the 16-lane row
        if constexpr (WIDTH >= 32) v += __builtin_bit_cast(float, __builtin_amdgcn_ds_swizzle(__builtin_bit_cast(int, v), 0x401F));   // lane ^ 16
        if constexpr (WIDTH >= 64) v += __shfl_xor(v, 32, 64);
    } else {
        // lane groups that are not a power of two wide (20 lanes = 80 channels, three groups per wave): a shift-down tree over the
        // group's lanes; only lane 0 of the group holds the full sum (which is all the callers read)
        const int l = (int)(threadIdx.x & 63) % WIDTH;
#pragma unroll
        for (int o = 16; o > 0; o >>= 1) {
            const float t = __shfl_down(v, o, 64);
            if (l + o < WIDTH) v += t;
        }
    }
    return v;
}

// Lane-group geometry shared by the row kernels: LPR lanes (one float4 each) own a row; a wave carries 64 / LPR groups and the
// lanes beyond the last whole group idle (LPR = 20: three rows per wave, lanes 60-63 idle).  Groups never straddle a wave.
template <int LPR, int BLOCK> struct LaneGroups {
    static constexpr int GPW = 64 / LPR, GPB = (BLOCK / 64) * GPW;
    int grp, lane;       // group within the block (or -1 for an idle lane), lane within the group
    __device__ __forceinline__ LaneGroups() {
        const int wl = threadIdx.x & 63, sub = wl / LPR;
        lane = wl % LPR;
        grp = sub < GPW ? (int)(threadIdx.x >> 6) * GPW + sub : -1;
    }
};

__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ void st4(float* p, float4 v) { *reinterpret_cast<float4*>(p) = v; }
// streaming (non-temporal) 16-byte store for outputs that are written once and not re-read by this kernel
__device__ __forceinline__ void st4_nt(float* p, float4 v) {
    typedef float v4f __attribute__((ext_vector_type(4)));
    v4f t = {v.x, v.y, v.z, v.w};
    __builtin_nontemporal_store(t, reinterpret_cast<v4f*>(p));
}
__device__ __forceinline__ float4 f4zero() { return make_float4(0.f, 0.f, 0.f, 0.f); }
__device__ __forceinline__ float4 f4fma(float a, float4 x, float4 acc) {
    acc.x = fmaf(a, x.x, acc.x); acc.y = fmaf(a, x.y, acc.y);
    acc.z = fmaf(a, x.z, acc.z); acc.w = fmaf(a, x.w, acc.w);
    return acc;
}
__device__ __forceinline__ float f4dot(float4 a, float4 b) {
    return fmaf(a.x, b.x, fmaf(a.y, b.y, fmaf(a.z, b.z, a.w * b.w)));
}

// Philox4x32-10 counter RNG: one 128-bit draw per (row, 4-column group); keep-mask decisions are
// reproducible between forward and backward from (seed, stream id, row, column) alone.
__device__ __forceinline__ uint4 philox4x32(uint64_t seed, uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3) {
    uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        uint64_t p0 = (uint64_t)0xD2511F53u * c0;
        uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
        uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
        uint32_t n1 = (uint32_t)p1;
        uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
        uint32_t n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    return make_uint4(c0, c1, c2, c3);
}

// u in [1e-10, 1 - 1e-10] of the concrete sampler for attention row `row` (Philox stream 4): what `torch.empty_like(z).uniform_(1e-10,
// 1 - 1e-10)` draws in the reference (example/gsat.py:96), at 23-bit resolution
__device__ __forceinline__ float philox_noise_u(uint64_t seed, int row) {
    const uint4 r = philox4x32(seed, (uint32_t)row, 0u, 4u, 0x5A17u);
    // 23 random bits + a half step: u in [2^-24, 1 - 2^-24], every value exactly representable (1 - 1e-10 rounds to 1.0f in fp32)
    return ((float)(r.x >> 9) + 0.5f) * (1.0f / 8388608.0f);
}

// Philox seed by value, or read from device memory (`dev` != NULL): a captured hipGraph then draws a new dropout mask on
// every replay because the host (or a graph-safe RNG op) rewrites that word between replays.
struct SeedRef {
    uint64_t value;
    const uint64_t* dev;
    __device__ __forceinline__ uint64_t get() const { return dev ? *dev : value; }
};

// keep[row, c..c+3] in {0,1} of the dropout stream `layer` (extractor layers 1 / 2, backbone layer tails 3)
__device__ __forceinline__ float4 philox_keep4(uint64_t seed, int layer, int row, int c, float p) {
    uint4 r = philox4x32(seed, (uint32_t)row, (uint32_t)(c >> 2), (uint32_t)layer, 0x5A17u);
    const float k = 1.0f / 16777216.0f;
    return make_float4((float)(r.x >> 8) * k >= p ? 1.f : 0.f, (float)(r.y >> 8) * k >= p ? 1.f : 0.f,
                       (float)(r.z >> 8) * k >= p ? 1.f : 0.f, (float)(r.w >> 8) * k >= p ? 1.f : 0.f);
}
#endif

}  // namespace gsat
