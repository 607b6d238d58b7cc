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

template <int MODE>
__global__ __launch_bounds__(DT, 2) void k_dual_gemm(const DualArgs P) {
    extern __shared__ __attribute__((aligned(16))) unsigned char dsm[];
    unsigned char* const Rhi = dsm;                                  // resident planes: [128][SBR] hi | lo   (MODE 1: A, MODE 2: Y)
    unsigned char* const Rlo = dsm + D_RM * P.SBR;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int SBR = P.SBR;
    // the chunked operand: MODE 1: Y (and OUT columns), MODE 2: A (the k extent of OUT)
    const float* const Cg = MODE == 1 ? P.Y : P.A;
    const int ldc = MODE == 1 ? P.ldy : P.lda, Cext = MODE == 1 ? P.KY : P.KA;
    const float* const Rg = MODE == 1 ? P.A : P.Y;
    const int ldr = MODE == 1 ? P.lda : P.ldy, Rext = MODE == 1 ? P.KA : P.KY, Rextp = MODE == 1 ? P.KAp : P.KYp;
    f32x16 accW[D_MAXCH];
    for (int k = 0; k < D_MAXCH; ++k)
#pragma unroll
        for (int r = 0; r < 16; ++r) accW[k][r] = 0.f;
    // chunk rows staged in registers: 128 rows x 64 floats = 2048 float4 / 512 threads = 4 per thread
    float4 stg[4];
    auto chunk_load = [&](int64_t row0, int c) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int i = tid + u * DT, r = i >> 4, q = i & 15;
            const int col = c * D_CH + 4 * q;
            const int64_t row = row0 + r;
            stg[u] = (row < P.R && col < Cext) ? ld4(Cg + (size_t)row * ldc + col) : f4zero();
        }
    };
    auto chunk_store = [&](unsigned char* buf) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int i = tid + u * DT, r = i >> 4, q = i & 15;
            d_store_split(buf, buf + D_RM * D_SBC, r * D_SBC + q * 8, stg[u]);
        }
    };

    for (int t = blockIdx.x; t < P.ntiles; t += gridDim.x) {
        const int64_t row0 = (int64_t)t * D_RM;
        d_barrier();                                        // the previous tile's planes are consumed
        // resident operand -> planes (zero beyond the matrix: padded columns and rows contribute nothing)
        {
            const int W4 = Rextp >> 2;
            for (int i = tid; i < D_RM * W4; i += DT) {
                const int r = i / W4, q = i - r * W4;
                const int64_t row = row0 + r;
                const float4 v = (row < P.R && 4 * q < Rext) ? ld4(Rg + (size_t)row * ldr + 4 * q) : f4zero();
                d_store_split(Rhi, Rlo, r * SBR + q * 8, v);
            }
        }
        chunk_load(row0, 0);
        chunk_store(dsm + P.offC0);
        f32x16 accO[2];
#pragma unroll
        for (int r = 0; r < 16; ++r) { accO[0][r] = 0.f; accO[1][r] = 0.f; }
        d_barrier();
        for (int c = 0; c < P.nch; ++c) {
            unsigned char* const Chi = dsm + ((c & 1) ? P.offC1 : P.offC0);
            unsigned char* const Clo = Chi + D_RM * D_SBC;
            if (c + 1 < P.nch) chunk_load(row0, c + 1);                 // in flight under this chunk's MFMAs
            if (MODE == 1) {
                // OUT[:, chunk] = A W[:, chunk]: 4 row blocks x 2 column blocks, one per wave; k = KA
                {
                    const int rb = wave >> 1, cb = wave & 1;
                    f32x16 acc;
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
                    const unsigned char* ah = Rhi + (rb * 32 + (lane & 31)) * SBR + (lane >> 5) * 16;
                    const unsigned char* al = ah + D_RM * SBR;
                    const uint4* bp = P.Wq + ((size_t)((c * 2 + cb) * P.S) * 2) * 64 + lane;
                    uint4 wh = bp[0], wl = bp[64];
                    for (int s = 0; s < P.S; ++s) {
                        const int sn = min(s + 1, P.S - 1);
                        const uint4 nh = bp[(size_t)sn * 128], nl = bp[(size_t)sn * 128 + 64];
                        const bf16x8 xh = *reinterpret_cast<const bf16x8*>(ah + s * 32);
                        const bf16x8 xl = *reinterpret_cast<const bf16x8*>(al + s * 32);
                        const bf16x8 bh = __builtin_bit_cast(bf16x8, wh), bl = __builtin_bit_cast(bf16x8, wl);
                        acc = D_MFMA(xl, bh, acc);
                        acc = D_MFMA(xh, bl, acc);
                        acc = D_MFMA(xh, bh, acc);
                        wh = nh; wl = nl;
                    }
                    const int col = c * D_CH + cb * 32 + (lane & 31);
                    if (col < P.NO) {
#pragma unroll
                        for (int r = 0; r < 16; ++r) {
                            const int64_t row = row0 + rb * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                            if (row < P.R) {
                                float* o = P.OUT + (size_t)row * P.ldo + col;
                                *o = P.accumulate ? *o + acc[r] : acc[r];
                            }
                        }
                    }
                }
                // DW[:, chunk] += A^T Y[:, chunk]: (KAp / 32) x 2 tiles, one per wave; k = the 128 tile rows
                {
                    const int ib = wave >> 1, jb = wave & 1;
                    if (ib * 32 < P.KAp) {
                        const int gq = lane >> 4, hh = gq >> 1, c16 = (gq & 1) * 16;
                        f32x16 acc = accW[c];
                        for (int s = 0; s < D_RM / 16; ++s) {
                            const int r0 = 16 * s + 8 * hh;
                            const bf16x8 ah = d_tr_frag(Rhi, SBR, r0, ib * 32 + c16, lane);
                            const bf16x8 al = d_tr_frag(Rlo, SBR, r0, ib * 32 + c16, lane);
                            const bf16x8 bh = d_tr_frag(Chi, D_SBC, r0, jb * 32 + c16, lane);
                            const bf16x8 bl = d_tr_frag(Clo, D_SBC, r0, jb * 32 + c16, lane);
                            acc = D_MFMA(al, bh, acc);
                            acc = D_MFMA(ah, bl, acc);
                            acc = D_MFMA(ah, bh, acc);
                        }
                        accW[c] = acc;
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
                            const uint4* bp = P.Wq + ((size_t)((c * P.ncb + cb) * 4) * 2) * 64 + lane;
                            f32x16 acc = accO[u];
#pragma unroll
                            for (int s = 0; s < 4; ++s) {
                                const uint4 wh = bp[(size_t)s * 128], wl = bp[(size_t)s * 128 + 64];
                                const bf16x8 xh = *reinterpret_cast<const bf16x8*>(ah + s * 32);
                                const bf16x8 xl = *reinterpret_cast<const bf16x8*>(al + s * 32);
                                const bf16x8 bh = __builtin_bit_cast(bf16x8, wh), bl = __builtin_bit_cast(bf16x8, wl);
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
                        f32x16 acc = accW[c];
                        for (int s = 0; s < D_RM / 16; ++s) {
                            const int r0 = 16 * s + 8 * hh;
                            const bf16x8 ah = d_tr_frag(Chi, D_SBC, r0, ib * 32 + c16, lane);
                            const bf16x8 al = d_tr_frag(Clo, D_SBC, r0, ib * 32 + c16, lane);
                            const bf16x8 bh = d_tr_frag(Rhi, SBR, r0, jb * 32 + c16, lane);
                            const bf16x8 bl = d_tr_frag(Rlo, SBR, r0, jb * 32 + c16, lane);
                            acc = D_MFMA(al, bh, acc);
                            acc = D_MFMA(ah, bl, acc);
                            acc = D_MFMA(ah, bh, acc);
                        }
                        accW[c] = acc;
                    }
                }
            }
            if (c + 1 < P.nch) chunk_store(dsm + ((c & 1) ? P.offC0 : P.offC1));      // nobody reads the other buffer during this chunk
            d_barrier();
        }
        if (MODE == 2) {
            const int rb = wave >> 1;
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int cb = 2 * (wave & 1) + u;
                const int col = cb * 32 + (lane & 31);
                if (cb < P.ncb && col < P.NO) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int64_t row = row0 + rb * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                        if (row < P.R) {
                            float* o = P.OUT + (size_t)row * P.ldo + col;
                            *o = P.accumulate ? *o + accO[u][r] : accO[u][r];
                        }
                    }
                }
            }
        }
    }
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
}

// DW[k, j] = sum over the workgroup slabs (fixed order); four threads share an output element (slabs q, q + 4, ...) and combine in a fixed
// tree, eight loads in flight each
__global__ void k_dual_reduce(const float* __restrict__ part, int nslab, size_t slab_stride, int rows, int cols, int ldpart, float* __restrict__ out,
                              int ldout) {
    const int64_t tg = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int q = (int)(tg & 3);
    const int64_t i = tg >> 2;
    const bool live = i < (int64_t)rows * cols;
    float a[4] = {0.f, 0.f, 0.f, 0.f};
    if (live) {
        const int r = (int)(i / cols), c = (int)(i % cols);
        const float* p0 = part + (size_t)r * ldpart + c;
        int s = q, j = 0;
        for (; s < nslab; s += 4, j = (j + 1) & 3) a[j] += p0[(size_t)s * slab_stride];
    }
    float v = (a[0] + a[1]) + (a[2] + a[3]);
    v += __shfl_xor(v, 1, 64);
    v += __shfl_xor(v, 2, 64);
    if (live && q == 0) { const int r = (int)(i / cols), c = (int)(i % cols); out[(size_t)r * ldout + c] = v; }
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
    const char* env = getenv("GSAT_DUAL_GEMM");
    if (env && atoi(env) == 0) return false;
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
    P.offC0 = 2 * D_RM * P.SBR;
    P.offC1 = P.offC0 + 2 * D_RM * D_SBC;
    const size_t lds = (size_t)P.offC1 + 2 * D_RM * D_SBC;
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
    if (mode == 1) {
        if (lds > allowed1) { GSAT_CHECK_HIP(hipFuncSetAttribute((const void*)k_dual_gemm<1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); allowed1 = lds; }
        k_dual_gemm<1><<<nwg, DT, lds, stream>>>(P);
    } else {
        DualArgs Q = P;
        Q.KAp = prow;                                             // slab rows (see above); the padded A extent is not used by MODE 2
        if (lds > allowed2) { GSAT_CHECK_HIP(hipFuncSetAttribute((const void*)k_dual_gemm<2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); allowed2 = lds; }
        k_dual_gemm<2><<<nwg, DT, lds, stream>>>(Q);
    }
    GSAT_LAUNCH_CHECK();
    const int64_t outs = (int64_t)KA * KY;
    k_dual_reduce<<<(unsigned)ceil_div(outs * 4, 256), 256, 0, stream>>>(P.part, nwg, (size_t)prow * P.ldpart, KA, KY, P.ldpart, DW, lddw);
    GSAT_LAUNCH_CHECK();
    return GSAT_OK;
}

}  // namespace gsat
