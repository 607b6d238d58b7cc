// "Dual" split-bf16 tile GEMMs of the extractor backward: for a tall row block A [R, KA] the two products that share it
//     OUT [R, NO]  (+)= A  W        (W [KA, NO] row-major: the next layer's weight)
//     DW  [KA, KY]   = A^T Y        (Y [R, KY]: the activations that met A in the forward)
// in ONE pass over the rows, instead of two k_gemm_bf16x3 launches + a slab reduction each (attn.hip: da1 / dW2 and demb / dW1).
// Why it pays: these shapes are R ~ 5e4..1e7 rows against 64..512 columns, so both products are memory-bound and A is read twice by the
// staged pair; the tile kernel of gemm.hip also pays a prologue / epilogue per 128 x 128 tile with only 4..8 k-slabs in between.
// Here a persistent 512-thread workgroup per CU walks 128-row tiles (static round-robin: fixed summation order, bitwise reproducible):
//   * rows are loaded with 16-byte coalesced loads, split into bf16 hi | lo planes in LDS (row-major, 16-byte row pad);
//   * the row product reads A fragments with ds_read_b128 and W as a pre-split fragment stream from L2 (k_dual_pack);
//   * the weight-gradient product takes BOTH operands k-major out of the same row-major planes with ds_read_b64_tr_b16;
//   * hi*hi + hi*lo + lo*hi on v_mfma_f32_32x32x16_bf16, fp32 accumulate (the precision policy of the staged backward);
//   * the 64-column chunk planes are double-buffered: the next chunk's rows are in flight (registers) under the current chunk's MFMAs;
//   * DW accumulates per workgroup over its tiles and leaves as one partial slab; k_dual_reduce sums the slabs in workgroup order.
// MODE 1 (A resident, KA <= 128; NO = KY chunked by 64):  da1 = dh2 W2, dW2 = dh2^T a1.
// MODE 2 (Y resident, KY = NO <= 128; KA chunked by 64):  demb (+)= dh1 W1, dW1 = dh1^T emb   (edge mode: dP / dQ halves).
#include "common.h"
#include "attn_fused.h"
#include <algorithm>
#include <cstdlib>

namespace gsat {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(8))) short s16x8;

constexpr int DT = 512;
constexpr int D_RM = 128;                 // rows per tile
constexpr int D_CH = 64;                  // chunk width
constexpr int D_SBC = D_CH * 2 + 16;      // bytes per row of a chunk plane
constexpr int D_MAXCH = 8;                // chunks (the chunked extent is at most 512)

__device__ __forceinline__ unsigned short d_bf16_bits(float x) { const __bf16 h = (__bf16)x; return __builtin_bit_cast(unsigned short, h); }
__device__ __forceinline__ float d_bf16_val(unsigned short b) { return __builtin_bit_cast(float, (unsigned)b << 16); }
__device__ __forceinline__ void d_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
#define D_MFMA(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_bf16((a), (b), (c), 0, 0, 0)

__device__ __forceinline__ bf16x8 d_tr_frag(const unsigned char* plane, int SB, int r0, int c0, int lane) {
    const int q = (lane & 15) >> 2, p = lane & 3;
    const unsigned char* a = plane + (r0 + q) * SB + (c0 + 4 * p) * 2;
    const s16x4 v0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(a));
    const s16x4 v1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(a + 4 * SB));
    const s16x8 v = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
    return __builtin_bit_cast(bf16x8, v);
}

// 4 floats -> 4 bf16 hi + 4 bf16 lo, stored as two 8-byte words at plane[row][col .. col + 3]
__device__ __forceinline__ void d_store_split(unsigned char* hi, unsigned char* lo, int off, float4 v) {
    const unsigned short h0 = d_bf16_bits(v.x), h1 = d_bf16_bits(v.y), h2 = d_bf16_bits(v.z), h3 = d_bf16_bits(v.w);
    const unsigned short l0 = d_bf16_bits(v.x - d_bf16_val(h0)), l1 = d_bf16_bits(v.y - d_bf16_val(h1));
    const unsigned short l2 = d_bf16_bits(v.z - d_bf16_val(h2)), l3 = d_bf16_bits(v.w - d_bf16_val(h3));
    *reinterpret_cast<uint2*>(hi + off) = make_uint2(h0 | (unsigned)h1 << 16, h2 | (unsigned)h3 << 16);
    *reinterpret_cast<uint2*>(lo + off) = make_uint2(l0 | (unsigned)l1 << 16, l2 | (unsigned)l3 << 16);
}

struct DualArgs {
    const float *A, *Y, *W;
    float *OUT, *part;
    const uint4* Wq;
    int64_t R;
    int KA, KY, NO;                  // extents; MODE 1: NO == KY (chunked), MODE 2: KY == NO (resident), KA chunked
    int lda, ldy, ldo;
    int KAp, KYp;                    // padded to 32
    int SBR;                         // bytes per row of the resident plane
    int nch, S;                      // chunks, k-steps of the row product per chunk
    int ncb;                         // MODE 2: 32-column blocks of OUT
    int accumulate;                  // OUT += (edge mode's second half)
    int ldpart;                      // row stride of a partial slab (floats)
    int ntiles;
    int offC0, offC1;                // byte offsets of the two chunk buffers (each hi | lo)
};

// W [KA, NO] (row stride ldw) -> split-bf16 B-operand streams.
//   MODE 1: Wq[(((c*2 + cb)*S + s)*2 + plane)*64 + lane]: k = 16 s + 8 (lane>>5) + j,        col = c*64 + cb*32 + (lane&31),  S = KAp/16
//   MODE 2: Wq[(((c*ncb + cb)*4 + s)*2 + plane)*64 + lane]: k = c*64 + 16 s + 8 (lane>>5) + j, col = cb*32 + (lane&31)
__global__ void k_dual_pack(const float* __restrict__ W, int ldw, int KA, int NO, int mode, int nch, int ncb, int S, uint4* __restrict__ Wq) {
    const int64_t n = (int64_t)nch * ncb * S * 64;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const int lane = (int)(i & 63), c32 = lane & 31, h = lane >> 5;
        const int s = (int)((i >> 6) % S);
        const int st = (int)((i >> 6) / S), cb = st % ncb, c = st / ncb;
        const int col = mode == 1 ? c * 64 + cb * 32 + c32 : cb * 32 + c32;
        const int k0 = mode == 1 ? 16 * s + 8 * h : c * 64 + 16 * s + 8 * h;
        unsigned short hi[8], lo[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int k = k0 + j;
            const float v = (k < KA && col < NO) ? W[(size_t)k * ldw + col] : 0.f;
            hi[j] = d_bf16_bits(v);
            lo[j] = d_bf16_bits(v - d_bf16_val(hi[j]));
        }
        const size_t o = ((size_t)(st * S + s) * 2) * 64 + lane;
        Wq[o] = make_uint4(hi[0] | (unsigned)hi[1] << 16, hi[2] | (unsigned)hi[3] << 16, hi[4] | (unsigned)hi[5] << 16, hi[6] | (unsigned)hi[7] << 16);
        Wq[o + 64] = make_uint4(lo[0] | (unsigned)lo[1] << 16, lo[2] | (unsigned)lo[3] << 16, lo[4] | (unsigned)lo[5] << 16, lo[6] | (unsigned)lo[7] << 16);
    }
}

#ifdef GSAT_FUSED_STAMPS
__device__ unsigned long long g_dual_stamps[2][8];
#define DSTAMP(i) do { if (tid == 0) { const long long n_ = __builtin_amdgcn_s_memtime(); dst_[i] += n_ - dtl_; dtl_ = n_; } } while (0)
#else
#define DSTAMP(i)
#endif

// LDS: resident planes [128][SBR] hi | lo, chunk planes [128][D_SBC] hi | lo, the chunk's weight fragments (fragment order, shared by the
// waves).  Everything the NEXT step needs (rows of the next chunk, its weight fragments, at a tile's last chunk also the next tile's
// resident rows and first chunk) is fetched into registers while the current chunk's MFMAs run, and written to LDS behind a barrier:
// the only exposed memory latency is the kernel's very first fill.
template <int MODE>
__global__ __launch_bounds__(DT, 2) void k_dual_gemm(const DualArgs P) {
    extern __shared__ __attribute__((aligned(16))) unsigned char dsm[];
    unsigned char* const Rhi = dsm;
    unsigned char* const Rlo = dsm + D_RM * P.SBR;
    unsigned char* const Chi = dsm + P.offC0;
    unsigned char* const Clo = Chi + D_RM * D_SBC;
    uint4* const Wl = reinterpret_cast<uint4*>(dsm + P.offC1);           // [ncb][S][2][64] fragments of the chunk
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int SBR = P.SBR;
    const float* const Cg = MODE == 1 ? P.Y : P.A;
    const int ldc = MODE == 1 ? P.ldy : P.lda, Cext = MODE == 1 ? P.KY : P.KA;
    const float* const Rg = MODE == 1 ? P.A : P.Y;
    const int ldr = MODE == 1 ? P.lda : P.ldy, Rext = MODE == 1 ? P.KA : P.KY, Rextp = MODE == 1 ? P.KAp : P.KYp;
    const int W4 = Rextp >> 2;                          // float4 per resident row: 128 * W4 / 512 <= 8 per thread
    const int wfrag = P.ncb * P.S * 2 * 64;             // uint4 per chunk of the weight stream (<= 2048: 4 per thread)
    f32x16 accW[D_MAXCH];
    for (int k = 0; k < D_MAXCH; ++k)
#pragma unroll
        for (int r = 0; r < 16; ++r) accW[k][r] = 0.f;
    // Prefetch loads are UNCONDITIONAL (addresses clamped into the matrix, the out-of-range lanes are zeroed when the registers are
    // written to LDS): a load under `cond ? load : 0` compiles to a branch with `s_waitcnt vmcnt(0)` at its merge, i.e. no prefetch at all.
    // The staging registers are plain scalars (arrays captured by the lambdas of an earlier version stayed in scratch, which also
    // serialises: a scratch store needs the loaded value at once).
    float4 rq0 = f4zero(), rq1 = f4zero(), rq2 = f4zero(), rq3 = f4zero();
    uint4 wq0 = make_uint4(0u, 0u, 0u, 0u), wq1 = wq0, wq2 = wq0, wq3 = wq0;
    int64_t pf_row0 = 0;
    int pf_c = 0;
#define D_ROWS_LOAD(ROW0, C)                                                                                                   \
    {                                                                                                                          \
        pf_row0 = (ROW0); pf_c = (C);                                                                                          \
        const int q_ = tid & 15, r_ = tid >> 4;                                                                                \
        const int col_ = min(pf_c * D_CH + 4 * q_, Cext - 4);                                                                  \
        rq0 = ld4(Cg + (size_t)min(pf_row0 + r_, P.R - 1) * ldc + col_);                                                       \
        rq1 = ld4(Cg + (size_t)min(pf_row0 + r_ + 32, P.R - 1) * ldc + col_);                                                  \
        rq2 = ld4(Cg + (size_t)min(pf_row0 + r_ + 64, P.R - 1) * ldc + col_);                                                  \
        rq3 = ld4(Cg + (size_t)min(pf_row0 + r_ + 96, P.R - 1) * ldc + col_);                                                  \
    }
#define D_ROWS_STORE()                                                                                                         \
    {                                                                                                                          \
        const int q_ = tid & 15, r_ = tid >> 4;                                                                                \
        const bool cin_ = pf_c * D_CH + 4 * q_ < Cext;                                                                         \
        d_store_split(Chi, Clo, r_ * D_SBC + q_ * 8, (cin_ && pf_row0 + r_ < P.R) ? rq0 : f4zero());                           \
        d_store_split(Chi, Clo, (r_ + 32) * D_SBC + q_ * 8, (cin_ && pf_row0 + r_ + 32 < P.R) ? rq1 : f4zero());               \
        d_store_split(Chi, Clo, (r_ + 64) * D_SBC + q_ * 8, (cin_ && pf_row0 + r_ + 64 < P.R) ? rq2 : f4zero());               \
        d_store_split(Chi, Clo, (r_ + 96) * D_SBC + q_ * 8, (cin_ && pf_row0 + r_ + 96 < P.R) ? rq3 : f4zero());               \
    }
#define D_W_LOAD(C)                                                                                                            \
    {                                                                                                                          \
        const uint4* wp_ = P.Wq + (size_t)(C) * wfrag;                                                                         \
        wq0 = wp_[min(tid, wfrag - 1)]; wq1 = wp_[min(tid + DT, wfrag - 1)];                                                   \
        wq2 = wp_[min(tid + 2 * DT, wfrag - 1)]; wq3 = wp_[min(tid + 3 * DT, wfrag - 1)];                                      \
    }
#define D_W_STORE()                                                                                                            \
    {                                                                                                                          \
        if (tid < wfrag) Wl[tid] = wq0;                                                                                        \
        if (tid + DT < wfrag) Wl[tid + DT] = wq1;                                                                              \
        if (tid + 2 * DT < wfrag) Wl[tid + 2 * DT] = wq2;                                                                      \
        if (tid + 3 * DT < wfrag) Wl[tid + 3 * DT] = wq3;                                                                      \
    }
    // resident rows of a tile -> planes, two batches of four 16-byte loads per thread
    auto res_fill = [&](int64_t row0) {
#pragma unroll 1
        for (int hb = 0; hb < 2; ++hb) {
            float4 v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int i = min(tid + (hb * 4 + u) * DT, D_RM * W4 - 1), r = i / W4, q = i - r * W4;
                const int64_t row = min(row0 + r, P.R - 1);
                v[u] = ld4(Rg + (size_t)row * ldr + min(4 * q, Rext - 4));
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int i = tid + (hb * 4 + u) * DT, r = i / W4, q = i - r * W4;
                const bool in = row0 + r < P.R && 4 * q < Rext;
                if (i < D_RM * W4) d_store_split(Rhi, Rlo, r * SBR + q * 8, in ? v[u] : f4zero());
            }
        }
    };
#ifdef GSAT_FUSED_STAMPS
    long long dst_[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    long long dtl_ = __builtin_amdgcn_s_memtime();
#endif
    if ((int)blockIdx.x < P.ntiles) {
        const int64_t row0 = (int64_t)blockIdx.x * D_RM;
        D_ROWS_LOAD(row0, 0) D_W_LOAD(0)
        res_fill(row0);
        D_ROWS_STORE() D_W_STORE()
    }
    d_barrier();
    DSTAMP(0);
    for (int t = blockIdx.x; t < P.ntiles; t += gridDim.x) {
        const int64_t row0 = (int64_t)t * D_RM;
        const int tn = t + (int)gridDim.x;
        f32x16 accO[2];
#pragma unroll
        for (int r = 0; r < 16; ++r) { accO[0][r] = 0.f; accO[1][r] = 0.f; }
        for (int c = 0; c < P.nch; ++c) {
            f32x16 accC = accW[c];                                  // (scratch) before the prefetches: vmcnt retires in order
            const bool last = c + 1 == P.nch;
            if (!last) { D_ROWS_LOAD(row0, c + 1) D_W_LOAD(c + 1) }
            else if (tn < P.ntiles) { D_ROWS_LOAD((int64_t)tn * D_RM, 0) D_W_LOAD(0) }      // (the next tile's resident rows are loaded at the seam)
            if (MODE == 1) {
                // DW[:, chunk] += A^T Y[:, chunk]: (KAp / 32) x 2 tiles, one per wave; k = the 128 tile rows
                {
                    const int ib = wave >> 1, jb = wave & 1;
                    if (ib * 32 < P.KAp) {
                        const int gq = lane >> 4, hh = gq >> 1, c16 = (gq & 1) * 16;
#pragma unroll 1
                        for (int s = 0; s < D_RM / 16; ++s) {
                            const int r0 = 16 * s + 8 * hh;
                            const bf16x8 ah = d_tr_frag(Rhi, SBR, r0, ib * 32 + c16, lane);
                            const bf16x8 al = d_tr_frag(Rlo, SBR, r0, ib * 32 + c16, lane);
                            const bf16x8 bh = d_tr_frag(Chi, D_SBC, r0, jb * 32 + c16, lane);
                            const bf16x8 bl = d_tr_frag(Clo, D_SBC, r0, jb * 32 + c16, lane);
                            accC = D_MFMA(al, bh, accC);
                            accC = D_MFMA(ah, bl, accC);
                            accC = D_MFMA(ah, bh, accC);
                        }
                    }
                }
                // OUT[:, chunk] = A W[:, chunk]: 4 row blocks x 2 column blocks, one per wave; k = KA; weight fragments from LDS
                {
                    const int rb = wave >> 1, cb = wave & 1;
                    f32x16 acc;
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
                    const unsigned char* ah = Rhi + (rb * 32 + (lane & 31)) * SBR + (lane >> 5) * 16;
                    const unsigned char* al = ah + D_RM * SBR;
                    const uint4* wl = Wl + (size_t)(cb * P.S) * 128 + lane;
                    for (int s = 0; s < P.S; ++s) {
                        const bf16x8 xh = *reinterpret_cast<const bf16x8*>(ah + s * 32);
                        const bf16x8 xl = *reinterpret_cast<const bf16x8*>(al + s * 32);
                        const bf16x8 bh = __builtin_bit_cast(bf16x8, wl[s * 128]), bl = __builtin_bit_cast(bf16x8, wl[s * 128 + 64]);
                        acc = D_MFMA(xl, bh, acc);
                        acc = D_MFMA(xh, bl, acc);
                        acc = D_MFMA(xh, bh, acc);
                    }
                    const int col = c * D_CH + cb * 32 + (lane & 31);
                    if (col < P.NO) {
                        float* const ob = P.OUT + (size_t)(row0 + rb * 32 + 4 * (lane >> 5)) * P.ldo + col;
                        const int64_t rlim = P.R - (row0 + rb * 32 + 4 * (lane >> 5));           // rows left below this lane's first row
                        if (P.accumulate) {
#pragma unroll
                            for (int r = 0; r < 16; ++r) { const int dr = (r & 3) + 8 * (r >> 2); if (dr < rlim) ob[(size_t)dr * P.ldo] += acc[r]; }
                        } else {
#pragma unroll
                            for (int r = 0; r < 16; ++r) { const int dr = (r & 3) + 8 * (r >> 2); if (dr < rlim) ob[(size_t)dr * P.ldo] = acc[r]; }
                        }
                    }
                }
            } else {
                // OUT += A[:, chunk] W[chunk, :]: 4 row blocks x ncb column blocks, two per wave; k = 64
                {
                    const int rb = wave >> 1;
                    const unsigned char* ah = Chi + (rb * 32 + (lane & 31)) * D_SBC + (lane >> 5) * 16;
                    const unsigned char* al = ah + D_RM * D_SBC;
#pragma unroll
                    for (int u = 0; u < 2; ++u) {
                        const int cb = 2 * (wave & 1) + u;
                        if (cb < P.ncb) {
                            const uint4* wl = Wl + (size_t)(cb * 4) * 128 + lane;
                            f32x16 acc = accO[u];
#pragma unroll
                            for (int s = 0; s < 4; ++s) {
                                const bf16x8 xh = *reinterpret_cast<const bf16x8*>(ah + s * 32);
                                const bf16x8 xl = *reinterpret_cast<const bf16x8*>(al + s * 32);
                                const bf16x8 bh = __builtin_bit_cast(bf16x8, wl[s * 128]), bl = __builtin_bit_cast(bf16x8, wl[s * 128 + 64]);
                                acc = D_MFMA(xl, bh, acc);
                                acc = D_MFMA(xh, bl, acc);
                                acc = D_MFMA(xh, bh, acc);
                            }
                            accO[u] = acc;
                        }
                    }
                }
                // DW[chunk, :] += A[:, chunk]^T Y: 2 x (KYp / 32) tiles, one per wave; k = the 128 tile rows
                {
                    const int ib = wave >> 2, jb = wave & 3;
                    if (jb * 32 < P.KYp) {
                        const int gq = lane >> 4, hh = gq >> 1, c16 = (gq & 1) * 16;
#pragma unroll 1
                        for (int s = 0; s < D_RM / 16; ++s) {
                            const int r0 = 16 * s + 8 * hh;
                            const bf16x8 ah = d_tr_frag(Chi, D_SBC, r0, ib * 32 + c16, lane);
                            const bf16x8 al = d_tr_frag(Clo, D_SBC, r0, ib * 32 + c16, lane);
                            const bf16x8 bh = d_tr_frag(Rhi, SBR, r0, jb * 32 + c16, lane);
                            const bf16x8 bl = d_tr_frag(Rlo, SBR, r0, jb * 32 + c16, lane);
                            accC = D_MFMA(al, bh, accC);
                            accC = D_MFMA(ah, bl, accC);
                            accC = D_MFMA(ah, bh, accC);
                        }
                    }
                }
                if (last) {                                      // the tile's OUT rows from the accumulators
                    const int rb = wave >> 1;
#pragma unroll
                    for (int u = 0; u < 2; ++u) {
                        const int cb = 2 * (wave & 1) + u;
                        const int col = cb * 32 + (lane & 31);
                        if (cb < P.ncb && col < P.NO) {
                            float* const ob = P.OUT + (size_t)(row0 + rb * 32 + 4 * (lane >> 5)) * P.ldo + col;
                            const int64_t rlim = P.R - (row0 + rb * 32 + 4 * (lane >> 5));
                            if (P.accumulate) {
#pragma unroll
                                for (int r = 0; r < 16; ++r) { const int dr = (r & 3) + 8 * (r >> 2); if (dr < rlim) ob[(size_t)dr * P.ldo] += accO[u][r]; }
                            } else {
#pragma unroll
                                for (int r = 0; r < 16; ++r) { const int dr = (r & 3) + 8 * (r >> 2); if (dr < rlim) ob[(size_t)dr * P.ldo] = accO[u][r]; }
                            }
                        }
                    }
                }
            }
            accW[c] = accC;
            DSTAMP(1);
            d_barrier();                                        // every wave is done with the chunk planes, the weights (and, at `last`, the resident planes)
            if (!last) { D_ROWS_STORE() D_W_STORE() }
            else if (tn < P.ntiles) { D_ROWS_STORE() D_W_STORE() res_fill((int64_t)tn * D_RM); }
            d_barrier();
            DSTAMP(2);
        }
    }
    DSTAMP(3);
    // this workgroup's partial of DW: [KAp, ldpart]; MODE 1: rows = KA blocks (wave >> 1), columns = chunk*64 + (wave & 1)*32;
    // MODE 2: rows = chunk*64 + (wave >> 2)*32, columns = (wave & 3)*32
    float* const pw = P.part + (size_t)blockIdx.x * P.KAp * P.ldpart;
    for (int c = 0; c < P.nch; ++c) {
        const f32x16 acc = accW[c];
        const int rbase = MODE == 1 ? (wave >> 1) * 32 : c * D_CH + (wave >> 2) * 32;
        const int cbase = MODE == 1 ? c * D_CH + (wave & 1) * 32 : (wave & 3) * 32;
        const bool live = MODE == 1 ? (wave >> 1) * 32 < P.KAp : (wave & 3) * 32 < P.KYp;
        if (live) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = rbase + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                if (row < P.KAp) pw[(size_t)row * P.ldpart + cbase + (lane & 31)] = acc[r];
            }
        }
    }
#ifdef GSAT_FUSED_STAMPS
    DSTAMP(4);
    if (tid == 0) { for (int i = 0; i < 6; ++i) atomicAdd(&g_dual_stamps[MODE - 1][i], (unsigned long long)dst_[i]); atomicAdd(&g_dual_stamps[MODE - 1][7], 1ull); }
#endif
}

// DW[k, j..j+3] = sum over the workgroup slabs (fixed order): four lanes share a float4 of the output (lane q takes slabs q, q + 4, ...,
// two partial sums in flight), 16-byte loads, then a fixed tree over the quad
__global__ void k_dual_reduce(const float* __restrict__ part, int nslab, size_t slab_stride, int rows, int cols, int ldpart, float* __restrict__ out,
                              int ldout) {
    const int64_t tg = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int q = (int)(tg & 3);
    const int64_t i4 = tg >> 2;
    const int c4 = cols >> 2;
    const bool live = i4 < (int64_t)rows * c4;
    float4 a = f4zero(), b = f4zero();
    int r = 0, c = 0;
    if (live) {
        r = (int)(i4 / c4); c = (int)(i4 % c4) * 4;
        const float* p0 = part + (size_t)r * ldpart + c;
        int s = q;
        for (; s + 4 < nslab; s += 8) {
            const float4 v0 = ld4(p0 + (size_t)s * slab_stride), v1 = ld4(p0 + (size_t)(s + 4) * slab_stride);
            a.x += v0.x; a.y += v0.y; a.z += v0.z; a.w += v0.w;
            b.x += v1.x; b.y += v1.y; b.z += v1.z; b.w += v1.w;
        }
        if (s < nslab) { const float4 v0 = ld4(p0 + (size_t)s * slab_stride); a.x += v0.x; a.y += v0.y; a.z += v0.z; a.w += v0.w; }
    }
    float4 v = make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w);
    v.x += __shfl_xor(v.x, 1, 64); v.y += __shfl_xor(v.y, 1, 64); v.z += __shfl_xor(v.z, 1, 64); v.w += __shfl_xor(v.w, 1, 64);
    v.x += __shfl_xor(v.x, 2, 64); v.y += __shfl_xor(v.y, 2, 64); v.z += __shfl_xor(v.z, 2, 64); v.w += __shfl_xor(v.w, 2, 64);
    if (live && q == 0) {
        float* o = out + (size_t)r * ldout + c;
        if ((ldout & 3) == 0 && ((uintptr_t)out & 15) == 0) st4(o, v);
        else { o[0] = v.x; o[1] = v.y; o[2] = v.z; o[3] = v.w; }
    }
}

static int dual_nwg() {
    static int cus = 0;
    if (!cus) {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0) cus = prop.multiProcessorCount;
        else cus = 256;
    }
    return cus;
}

bool dual_gemm_ok(int mode, int64_t R, int KA, int KY, int NO) {
    // opt-in (GSAT_DUAL_GEMM=1): measured on MI355X the one-pass kernels do not beat the GEMM pairs they replace yet (C3: 94 / 79 us + 13 us
    // reductions against 58 + 40 us): one 143 KB workgroup per CU leaves its row loads, weight-fragment loads and MFMAs un-overlapped
    // (profiles/r03_summary.md has the phase stamps)
    const char* env = getenv("GSAT_DUAL_GEMM");
    if (!(env && atoi(env) != 0)) return false;
    if (R <= 0 || KA % 4 || KY % 4 || NO % 4) return false;
    if (mode == 1) return KA <= 128 && NO == KY && KY <= D_MAXCH * D_CH;
    return KY <= 128 && NO == KY && KA <= D_MAXCH * D_CH;
}

size_t dual_gemm_ws_bytes(int mode, int KA, int KY, int NO) {
    const int KAp = (KA + 31) / 32 * 32, KYp = (KY + 31) / 32 * 32;
    const int nch = mode == 1 ? (KY + D_CH - 1) / D_CH : (KA + D_CH - 1) / D_CH;
    const int ncb = mode == 1 ? 2 : KYp / 32, S = mode == 1 ? KAp / 16 : 4;
    const int ldpart = mode == 1 ? nch * D_CH : KYp;
    const int rows = mode == 1 ? KAp : nch * D_CH;
    return align_up((size_t)nch * ncb * S * 2 * 64 * 16, 256) + align_up((size_t)dual_nwg() * rows * ldpart * 4, 256);
}

// OUT [R, NO] (+)= A W ; DW [KA, KY] = A^T Y      (see the header comment for the two modes)
int dual_gemm(hipStream_t stream, int mode, int64_t R, int KA, int KY, int NO, const float* A, int lda, const float* Y, int ldy, const float* W,
              int ldw, float* OUT, int ldo, int accumulate, float* DW, int lddw, void* ws, size_t ws_bytes) {
    GSAT_REQUIRE(dual_gemm_ok(mode, R, KA, KY, NO), GSAT_ERR_UNSUPPORTED, "dual_gemm: unsupported shape");
    GSAT_REQUIRE(ws && ws_bytes >= dual_gemm_ws_bytes(mode, KA, KY, NO), GSAT_ERR_WORKSPACE, "dual_gemm: workspace too small");
    GSAT_REQUIRE(lda % 4 == 0 && ldy % 4 == 0 && ((uintptr_t)A % 16 == 0) && ((uintptr_t)Y % 16 == 0), GSAT_ERR_ARG, "dual_gemm: alignment");
    DualArgs P{};
    P.A = A; P.Y = Y; P.W = W; P.OUT = OUT; P.R = R; P.KA = KA; P.KY = KY; P.NO = NO; P.lda = lda; P.ldy = ldy; P.ldo = ldo;
    P.KAp = (KA + 31) / 32 * 32; P.KYp = (KY + 31) / 32 * 32;
    P.nch = mode == 1 ? (KY + D_CH - 1) / D_CH : (KA + D_CH - 1) / D_CH;
    P.ncb = mode == 1 ? 2 : P.KYp / 32;
    P.S = mode == 1 ? P.KAp / 16 : 4;
    P.accumulate = accumulate;
    P.ldpart = mode == 1 ? P.nch * D_CH : P.KYp;
    const int prow = mode == 1 ? P.KAp : P.nch * D_CH;           // rows of a partial slab
    const int resw = mode == 1 ? P.KAp : P.KYp;
    P.SBR = resw * 2 + 16;
    P.offC0 = 2 * D_RM * P.SBR;                                  // chunk planes
    P.offC1 = P.offC0 + 2 * D_RM * D_SBC;                        // the chunk's weight fragments: ncb * S * 2 * 64 uint4 (<= 32 KB)
    const size_t lds = (size_t)P.offC1 + (size_t)P.ncb * P.S * 2 * 64 * 16;
    GSAT_REQUIRE(P.ncb * P.S * 2 * 64 <= 4 * DT, GSAT_ERR_UNSUPPORTED, "dual_gemm: weight chunk too large");
    P.ntiles = (int)ceil_div(R, D_RM);
    char* w = static_cast<char*>(ws);
    uint4* Wq = reinterpret_cast<uint4*>(w); w += align_up((size_t)P.nch * P.ncb * P.S * 2 * 64 * 16, 256);
    P.Wq = Wq; P.part = reinterpret_cast<float*>(w);
    // the kernel indexes a partial slab with KAp rows in MODE 1 and nch*64 rows in MODE 2: pass the slab's row count through KAp there
    const int nwg = std::min(dual_nwg(), P.ntiles);
    {
        const int64_t n = (int64_t)P.nch * P.ncb * P.S * 64;
        k_dual_pack<<<(unsigned)std::min<int64_t>(ceil_div(n, 256), 256), 256, 0, stream>>>(W, ldw, KA, NO, mode, P.nch, P.ncb, P.S, Wq);
        GSAT_LAUNCH_CHECK();
    }
    static size_t allowed1 = 64 * 1024, allowed2 = 64 * 1024;
    if (lds > allowed1) {
        GSAT_CHECK_HIP(hipFuncSetAttribute((const void*)k_dual_gemm<1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        GSAT_CHECK_HIP(hipFuncSetAttribute((const void*)k_dual_gemm<2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        allowed1 = lds;
    }
    (void)allowed2;
    if (mode == 1) {
        k_dual_gemm<1><<<nwg, DT, lds, stream>>>(P);
    } else {
        DualArgs Q = P;
        Q.KAp = prow;                                             // slab rows (see above); the padded A extent is not used by MODE 2
        k_dual_gemm<2><<<nwg, DT, lds, stream>>>(Q);
    }
    GSAT_LAUNCH_CHECK();
    const int64_t outs = (int64_t)KA * KY;
    k_dual_reduce<<<(unsigned)ceil_div(outs, 256), 256, 0, stream>>>(P.part, nwg, (size_t)prow * P.ldpart, KA, KY, P.ldpart, DW, lddw);
    GSAT_LAUNCH_CHECK();
    return GSAT_OK;
}

}  // namespace gsat

#ifdef GSAT_FUSED_STAMPS
extern "C" const unsigned long long* gsat_debug_dual_stamps(void) {
    static unsigned long long host[16];
    if (hipDeviceSynchronize() != hipSuccess) return nullptr;
    if (hipMemcpyFromSymbol(host, HIP_SYMBOL(gsat::g_dual_stamps), sizeof(host)) != hipSuccess) return nullptr;
    return host;
}
#endif
