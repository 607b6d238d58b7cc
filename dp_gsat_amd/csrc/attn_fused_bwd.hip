// Fused backward of the attention extractor through the head, the second InstanceNorm, layer 2 and the first InstanceNorm
// (example/gsat.py:131-139 / src/utils/get_model.py:47-68 under autograd), ONE launch per batch + one reduction launch.
// It replaces k_dz, the head's statistics kernel, the column sums, the da1 = dh2 W2 and dW2 = dh2^T a1 GEMMs and the layer-1 statistics
// kernel of the staged backward (attn.hip); what it hands on is dh1 [M, C1] (gradient w.r.t. the layer-1 pre-activation), from which the
// staged code forms demb / dW1 (node mode) or dP / dQ (edge mode) as before.
//
// A 512-thread workgroup owns a tile of whole graphs (<= 128 MLP rows, the planner of attn_fused.hip).  Per tile:
//   * a THREAD owns one channel of one graph ("column thread"): it pulls that column of h2 (or P | P[src]+Q[dst]) straight from global
//     memory -- consecutive lanes = consecutive channels, 32 rows in flight, kept in registers -- and produces, without any cross-thread
//     reduction, the InstanceNorm-backward sums S1 = mean(dy), S2 = mean(dy y) and then dh = rstd (dy - S1 - y S2);
//   * dropout keep bits come from one Philox draw per 4 channels (a float4-mapped pass that leaves a bit per element in LDS);
//   * dh2 and the recomputed a1 chunk are written to LDS as split-bf16 planes (hi | lo); da1 = dh2 W2[:, chunk] and
//     dW2[:, chunk] += dh2^T a1 run on v_mfma_f32_32x32x16_bf16 as hi*hi + hi*lo + lo*hi (the precision policy of the staged backward);
//     the k-major operands of the weight gradient come out of the row-major planes through ds_read_b64_tr_b16;
//   * dW2 accumulates in registers across the tiles of a workgroup (static round-robin tiles: fixed summation order), and leaves as one
//     partial per workgroup; k_attn_bwd_reduce sums the partials of dW2 / dW3 / db3 in workgroup order.
// Nothing of a1 / da1 [M, C1] reaches HBM.  A graph larger than a tile is walked in 128-row slabs with two sweeps per channel chunk
// (statistics, then gradients): correct for any size, meant for the occasional large molecule.
#include "common.h"
#include "attn_fused.h"
#include <algorithm>
#include <cstdlib>

namespace gsat {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(8))) short s16x8;

constexpr int BT = 512;
constexpr int B_RM = 128;                 // MLP rows per tile
constexpr int B_CH = 64;                  // layer-1 channels per chunk
constexpr int B_SBA = B_CH * 2 + 16;      // bytes per row of an a1 plane (128 B of bf16 + 16 B pad)
constexpr int B_LDT = B_CH + 4;           // floats per row of the da1 tile
constexpr int B_CB = 16;                  // rows of a column in flight per batch

struct BwdGeom {
    int C1, C2, C2p, NCH, S2b, SBH;       // S2b = C2p / 16 k-steps of the da1 product; SBH = bytes per row of a dh2 plane
    int C1pad;                            // NCHT * 64: row stride of a dW2 partial
    int offA1, offT, offF2, offF1, offMeta, lds_bytes;      // byte offsets
};

struct BwdArgs {
    const float *h2, *b2, *w3, *P, *Q, *b1, *stats, *dlogits, *datt, *att, *mask1, *mask2;
    const int32_t *src, *dst, *seg_ptr, *order;
    const uint4* Wq2;
    float* dh1;
    float *partW2, *partW3, *partB3;
    const FTile* tiles;
    const int* counters;
    int64_t M, G;
    SeedRef seed;
    float p;
    int training;
    BwdGeom g;
};

__device__ __forceinline__ unsigned short bf16_bits(float x) {
    const __bf16 h = (__bf16)x;
    return __builtin_bit_cast(unsigned short, h);
}
__device__ __forceinline__ float bf16_val(unsigned short b) { return __builtin_bit_cast(float, (unsigned)b << 16); }
__device__ __forceinline__ void bwd_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// 8 consecutive k (tile rows r0 .. r0 + 7) of one column as a k-major MFMA operand, from a row-major bf16 plane: two transposed reads.
// Lane 4q + p of a 16-lane group addresses row (r0 + q), columns c0 + 4p .. + 3; lane i of the group receives column c0 + i.
__device__ __forceinline__ bf16x8 tr_frag(const unsigned char* plane, int SB, int r0, int c0, int lane) {
    const int q = (lane & 15) >> 2, p = lane & 3;
    const unsigned char* a = plane + (r0 + q) * SB + (c0 + 4 * p) * 2;
    const s16x4 v0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(a));
    const s16x4 v1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(a + 4 * SB));
    const s16x8 v = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
    return __builtin_bit_cast(bf16x8, v);
}

// keep decision of ONE element (the big-graph path evaluates it per element instead of staging bits in LDS)
__device__ __forceinline__ bool keep_one(const float* mask, SeedRef seed, int layer, int m, int c, int C, float p, bool drop) {
    if (!drop) return true;
    if (mask) return mask[(size_t)m * C + c] != 0.f;
    const float4 k = philox_keep4(seed.get(), layer, m, c & ~3, p);
    const int j = c & 3;
    return (j == 0 ? k.x : j == 1 ? k.y : j == 2 ? k.z : k.w) != 0.f;
}

#define GSAT_MFMA_BF16(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_bf16((a), (b), (c), 0, 0, 0)

// W2 [C2, C1] -> split-bf16 B-operand stream of the da1 product (k = c2, column = c1):
//   Wq2[(((kc*2 + cb)*S2b + s)*2 + plane)*64 + lane] = 8 bf16: W2[16 s + 8 (lane>>5) + j][kc*64 + cb*32 + (lane&31)], j = 0..7
__global__ void k_attn_bwd_pack(const float* __restrict__ W2, int C1, int C2, int NCH, int S2b, uint4* __restrict__ Wq2) {
    const int64_t n = (int64_t)NCH * 2 * S2b * 64;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const int lane = (int)(i & 63), c = lane & 31, h = lane >> 5;
        const int s = (int)((i >> 6) % S2b);
        const int st = (int)((i >> 6) / S2b), cb = st & 1, kc = st >> 1;
        const int col = kc * 64 + cb * 32 + c;
        unsigned short hi[8], lo[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int k = 16 * s + 8 * h + j;
            const float v = (k < C2 && col < C1) ? W2[(size_t)k * C1 + col] : 0.f;
            hi[j] = bf16_bits(v);
            lo[j] = bf16_bits(v - bf16_val(hi[j]));
        }
        const size_t o = ((size_t)(st * S2b + s) * 2) * 64 + lane;
        Wq2[o] = make_uint4(hi[0] | (unsigned)hi[1] << 16, hi[2] | (unsigned)hi[3] << 16, hi[4] | (unsigned)hi[5] << 16, hi[6] | (unsigned)hi[7] << 16);
        Wq2[o + 64] = make_uint4(lo[0] | (unsigned)lo[1] << 16, lo[2] | (unsigned)lo[3] << 16, lo[4] | (unsigned)lo[5] << 16, lo[6] | (unsigned)lo[7] << 16);
    }
}

template <bool EDGE, int NCHT, bool BIG>
__global__ __launch_bounds__(BT, 2) void k_attn_fused_bwd(const BwdArgs A) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const BwdGeom& g = A.g;
    const int C1 = g.C1, C2 = g.C2, C2p = g.C2p, SBH = g.SBH;
    unsigned char* const D2hi = smem;                                 // [128][SBH] dh2, bf16 hi
    unsigned char* const D2lo = smem + B_RM * SBH;
    unsigned char* const A1hi = smem + g.offA1;                       // [128][B_SBA] a1 chunk, bf16 hi
    unsigned char* const A1lo = A1hi + B_RM * B_SBA;
    float* const T = reinterpret_cast<float*>(smem + g.offT);         // [128][B_LDT] da1 chunk
    unsigned char* const F2 = smem + g.offF2;                         // [128][C2p / 4] keep bits of layer 2 (4 channels per byte)
    unsigned char* const F1 = smem + g.offF1;                         // [128][16] keep bits of the layer-1 chunk
    int* const meta = reinterpret_cast<int*>(smem + g.offMeta);
    int* const sGptr = meta;                                          // [F_GT + 1]
    int* const sRowId = meta + 32;                                    // [128] global MLP row
    int* const sSrc = sRowId + B_RM;                                  // [128] edge mode: global source / destination node
    int* const sDst = sSrc + B_RM;
    float* const sDz = reinterpret_cast<float*>(sDst + B_RM);         // [128]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int64_t G = A.G;
    const float* const mean1 = A.stats;
    const float* const rstd1 = mean1 + (size_t)G * C1;
    const float* const mean2 = rstd1 + (size_t)G * C1;
    const float* const rstd2 = mean2 + (size_t)G * C2;
    const bool train = A.training != 0;
    const float sc = (train && A.p > 0.f) ? 1.f / (1.f - A.p) : 1.f;
    const bool drop = train && A.p > 0.f;
    const int ntiles = A.counters[0];
    const int FB2 = C2p >> 2;                                         // bytes per row of F2
    // layer-2 column threads: channel c2 = tid % C2p of the graphs slot2, slot2 + nslot2, ...; layer 1: channel tid % 64, slot tid / 64
    const int cc2 = tid % C2p, slot2 = tid / C2p, nslot2 = BT / C2p;
    const int cc1 = tid & 63, slot1 = tid >> 6;
    f32x16 accW[NCHT];
#pragma unroll
    for (int k = 0; k < NCHT; ++k)
#pragma unroll
        for (int r = 0; r < 16; ++r) accW[k][r] = 0.f;
    float dw3acc = 0.f, db3acc = 0.f;
    // upstream gradient of the logit: dz = dlogits + datt * att * (1 - att)
    auto dz_of = [&](int m) {
        float v = A.dlogits ? A.dlogits[m] : 0.f;
        if (A.datt) { const float a = A.att[m]; v = fmaf(A.datt[m], a * (1.f - a), v); }
        return v;
    };

    // da1 chunk = dh2 (planes, rows [0, nrows)) x W2[:, chunk kc] -> T      (wave: row block wave >> 1, column block wave & 1)
    auto mfma_da1 = [&](int kc, int nrows) {
        const int rb = wave >> 1, cb = wave & 1;
        if (rb * 32 < nrows) {
            f32x16 acc;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = 0.f;
            const unsigned char* ah = D2hi + (rb * 32 + (lane & 31)) * SBH + (lane >> 5) * 16;
            const unsigned char* al = ah + B_RM * SBH;
            const uint4* bp = A.Wq2 + ((size_t)((kc * 2 + cb) * g.S2b) * 2) * 64 + lane;
            uint4 wh = bp[0], wl = bp[64];
            for (int s = 0; s < g.S2b; ++s) {
                const int sn = min(s + 1, g.S2b - 1);                          // next step's weights fly under this step's MFMAs
                const uint4 nh = bp[(size_t)sn * 128], nl = bp[(size_t)sn * 128 + 64];
                const bf16x8 xh = *reinterpret_cast<const bf16x8*>(ah + s * 32);
                const bf16x8 xl = *reinterpret_cast<const bf16x8*>(al + s * 32);
                const bf16x8 bh = __builtin_bit_cast(bf16x8, wh), bl = __builtin_bit_cast(bf16x8, wl);
                acc = GSAT_MFMA_BF16(xl, bh, acc);
                acc = GSAT_MFMA_BF16(xh, bl, acc);
                acc = GSAT_MFMA_BF16(xh, bh, acc);
                wh = nh; wl = nl;
            }
            float* const o = T + (rb * 32 + 4 * (lane >> 5)) * B_LDT + cb * 32 + (lane & 31);
#pragma unroll
            for (int r = 0; r < 16; ++r) o[((r & 3) + 8 * (r >> 2)) * B_LDT] = acc[r];
        }
    };
    // acc += dh2^T a1 over the tile rows [0, nrows16) (planes zero-padded to a multiple of 16 rows)
    auto mfma_dw2 = [&](f32x16 acc, int nrows16) {
        const int ib = wave >> 1, jb = wave & 1;
        if (ib * 32 < C2p) {
            const int gq = lane >> 4, hh = gq >> 1, c16 = (gq & 1) * 16;
            for (int s = 0; s < nrows16 / 16; ++s) {
                const int r0 = 16 * s + 8 * hh;
                const bf16x8 ah = tr_frag(D2hi, SBH, r0, ib * 32 + c16, lane);
                const bf16x8 al = tr_frag(D2lo, SBH, r0, ib * 32 + c16, lane);
                const bf16x8 bh = tr_frag(A1hi, B_SBA, r0, jb * 32 + c16, lane);
                const bf16x8 bl = tr_frag(A1lo, B_SBA, r0, jb * 32 + c16, lane);
                acc = GSAT_MFMA_BF16(al, bh, acc);
                acc = GSAT_MFMA_BF16(ah, bl, acc);
                acc = GSAT_MFMA_BF16(ah, bh, acc);
            }
        }
        return acc;
    };

#ifdef GSAT_FUSED_STAMPS
    long long stamp[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    long long tlast = __builtin_amdgcn_s_memtime();
#define BSTAMP(i) do { if (tid == 0) { const long long n_ = __builtin_amdgcn_s_memtime(); stamp[i] += n_ - tlast; tlast = n_; } } while (0)
#else
#define BSTAMP(i)
#endif
    const int nbig = A.counters[2];
    if ((int)blockIdx.x >= (BIG ? nbig : ntiles - nbig)) return;      // no tile for this workgroup: no partial either (the reduction counts slabs the same way)
    const int slab = (BIG ? (int)gridDim.x : 0) + (int)blockIdx.x;
    for (int t = (BIG ? 0 : nbig) + blockIdx.x; t < (BIG ? nbig : ntiles); t += gridDim.x) {
        const FTile tl = A.tiles[t];
        if (BIG) {
            // ======================= one graph larger than a tile: 128-row slabs, statistics over ALL its rows first =======================
            const int nR = tl.nrows, gi = tl.g0;
            const float inv_n = 1.f / (float)max(nR, 1);
            float* const red = T;                                          // cross-slot sums (T is free between the MFMA phases)
            auto rowid = [&](int r) { const int gr = tl.row0 + r; return EDGE ? (A.order ? A.order[gr] : gr) : gr; };
            // layer 2: S1, S2 of every channel over all rows, rows dealt to the slots, slot partials summed in slot order
            float S1 = 0.f, S2 = 0.f, mu2 = 0.f, rs2 = 0.f, bias2 = 0.f, w3c = 0.f;
            bwd_barrier();
            {
                float s1 = 0.f, s2 = 0.f;
                if (slot2 < nslot2 && cc2 < C2) {
                    mu2 = mean2[(size_t)gi * C2 + cc2]; rs2 = rstd2[(size_t)gi * C2 + cc2]; bias2 = A.b2[cc2]; w3c = A.w3[cc2];
                    for (int r = slot2; r < nR; r += nslot2) {
                        const int m = rowid(r);
                        const float yy = ((A.h2[(size_t)m * C2 + cc2] + bias2) - mu2) * rs2;
                        const bool on = yy > 0.f && keep_one(A.mask2, A.seed, 2, m, cc2, C2, A.p, drop);
                        const float d = dz_of(m);
                        const float dy = on ? d * w3c * sc : 0.f;
                        s1 += dy; s2 = fmaf(dy, yy, s2);
                        dw3acc += on ? d * (yy * sc) : 0.f;
                        if (cc2 == 0) db3acc += d;
                    }
                }
                if (slot2 < nslot2) { red[slot2 * C2p + cc2] = s1; red[(nslot2 + slot2) * C2p + cc2] = s2; }
                bwd_barrier();
                if (slot2 < nslot2) {
                    for (int k = 0; k < nslot2; ++k) { S1 += red[k * C2p + cc2]; S2 += red[(nslot2 + k) * C2p + cc2]; }
                    S1 *= inv_n; S2 *= inv_n;
                }
                bwd_barrier();
            }
            // dh2 of the slab rows [s0, s0 + ns) -> planes (rows 0 .. ns - 1, zero-padded to a multiple of 16)
            auto slab_dh2 = [&](int s0, int ns) {
                if (slot2 < nslot2) {
                    unsigned short* const ph = reinterpret_cast<unsigned short*>(D2hi) + cc2;
                    unsigned short* const pl = reinterpret_cast<unsigned short*>(D2lo) + cc2;
                    const int SH = SBH >> 1, ns16 = (ns + 15) & ~15;
                    for (int r = slot2; r < ns16; r += nslot2) {
                        float dh = 0.f;
                        if (r < ns && cc2 < C2) {
                            const int m = rowid(s0 + r);
                            const float yy = ((A.h2[(size_t)m * C2 + cc2] + bias2) - mu2) * rs2;
                            const bool on = yy > 0.f && keep_one(A.mask2, A.seed, 2, m, cc2, C2, A.p, drop);
                            const float dy = on ? dz_of(m) * w3c * sc : 0.f;
                            dh = rs2 * (dy - S1 - yy * S2);
                        }
                        const unsigned short hb = bf16_bits(dh);
                        ph[r * SH] = hb;
                        pl[r * SH] = bf16_bits(dh - bf16_val(hb));
                    }
                }
            };
#pragma unroll
            for (int kc = 0; kc < NCHT; ++kc) {
                if (kc * B_CH < C1) {
                const int col = kc * B_CH + cc1;
                const bool live = col < C1;
                const float mu1 = live ? mean1[(size_t)gi * C1 + col] : 0.f, rs1 = live ? rstd1[(size_t)gi * C1 + col] : 0.f;
                const float bias1 = live ? A.b1[col] : 0.f;
                auto y1_of = [&](int m) {
                    float h;
                    if (EDGE) h = A.P[(size_t)A.src[m] * C1 + col] + A.Q[(size_t)A.dst[m] * C1 + col];
                    else h = A.P[(size_t)m * C1 + col];
                    return ((h + bias1) - mu1) * rs1;
                };
                // sweep A: S1', S2' of the chunk over all rows (slot partials summed in slot order)
                float q1 = 0.f, q2 = 0.f;
                for (int s0 = 0; s0 < nR; s0 += B_RM) {
                    const int ns = min(B_RM, nR - s0);
                    slab_dh2(s0, ns);
                    bwd_barrier();
                    mfma_da1(kc, ns);
                    bwd_barrier();
                    if (live)
                        for (int r = slot1; r < ns; r += BT / 64) {
                            const int m = rowid(s0 + r);
                            const float yy = y1_of(m);
                            const bool on = yy > 0.f && keep_one(A.mask1, A.seed, 1, m, col, C1, A.p, drop);
                            const float dy = on ? T[r * B_LDT + cc1] * sc : 0.f;
                            q1 += dy; q2 = fmaf(dy, yy, q2);
                        }
                    bwd_barrier();
                }
                red[slot1 * 64 + cc1] = q1; red[(8 + slot1) * 64 + cc1] = q2;
                bwd_barrier();
                float Q1 = 0.f, Q2 = 0.f;
                for (int k = 0; k < 8; ++k) { Q1 += red[k * 64 + cc1]; Q2 += red[(8 + k) * 64 + cc1]; }
                Q1 *= inv_n; Q2 *= inv_n;
                bwd_barrier();
                // sweep B: dh1 -> global, a1 -> planes, dW2 chunk
                f32x16 acc = accW[kc];
                for (int s0 = 0; s0 < nR; s0 += B_RM) {
                    const int ns = min(B_RM, nR - s0), ns16 = (ns + 15) & ~15;
                    slab_dh2(s0, ns);
                    bwd_barrier();
                    mfma_da1(kc, ns);
                    bwd_barrier();
                    {
                        unsigned short* const ph = reinterpret_cast<unsigned short*>(A1hi) + cc1;
                        unsigned short* const pl = reinterpret_cast<unsigned short*>(A1lo) + cc1;
                        constexpr int SH = B_SBA >> 1;
                        for (int r = slot1; r < ns16; r += BT / 64) {
                            float a1 = 0.f;
                            if (r < ns && live) {
                                const int m = rowid(s0 + r);
                                const float yy = y1_of(m);
                                const bool on = yy > 0.f && keep_one(A.mask1, A.seed, 1, m, col, C1, A.p, drop);
                                const float dy = on ? T[r * B_LDT + cc1] * sc : 0.f;
                                a1 = on ? yy * sc : 0.f;
                                A.dh1[(size_t)m * C1 + col] = rs1 * (dy - Q1 - yy * Q2);
                            }
                            const unsigned short hb = bf16_bits(a1);
                            ph[r * SH] = hb;
                            pl[r * SH] = bf16_bits(a1 - bf16_val(hb));
                        }
                    }
                    bwd_barrier();
                    acc = mfma_dw2(acc, ns16);
                    bwd_barrier();
                }
                accW[kc] = acc;
                }
            }
            continue;
        }

        // ======================================= whole graphs in one tile =======================================
        const int nr = tl.nrows, nr16 = (nr + 15) & ~15;
        bwd_barrier();
        // ---- P0: tile meta, dz, keep bits of layer 2 -------------------------------------------------------------------------
        if (tid <= tl.ng) sGptr[tid] = A.seg_ptr[tl.g0 + tid] - tl.row0;
        for (int r = tid; r < nr; r += BT) {
            const int gr = tl.row0 + r;
            const int m = EDGE ? (A.order ? A.order[gr] : gr) : gr;
            sRowId[r] = m;
            if (EDGE) { sSrc[r] = A.src[m]; sDst[r] = A.dst[m]; }
            sDz[r] = dz_of(m);
        }
        for (int i = tid; i < nr * FB2; i += BT) {
            const int r = i / FB2, q = i - r * FB2;
            unsigned bits = 15u;
            if (drop && 4 * q < C2) {
                const int gr = tl.row0 + r;
                const int m = EDGE ? (A.order ? A.order[gr] : gr) : gr;
                const float4 k = keep4f(A.mask2, A.seed, 2, m, 4 * q, C2, A.p, true);
                bits = (k.x != 0.f ? 1u : 0u) | (k.y != 0.f ? 2u : 0u) | (k.z != 0.f ? 4u : 0u) | (k.w != 0.f ? 8u : 0u);
            }
            F2[i] = (unsigned char)bits;
        }
        bwd_barrier();
        BSTAMP(0);
        // ---- P1: layer-2 column pass: S1, S2, dW3 / db3 partials, dh2 -> bf16 planes ---------------------------------------------------
        // A column is walked in batches of CB rows, twice (sums, then gradients): every load of a batch is unconditional (row clamped
        // into the graph) and in flight together, nothing but the two sums lives across batches, and the clamped duplicates of the last
        // row rewrite identical values.
        if (slot2 < nslot2) {
            unsigned short* const ph = reinterpret_cast<unsigned short*>(D2hi) + cc2;
            unsigned short* const pl = reinterpret_cast<unsigned short*>(D2lo) + cc2;
            const int SH = SBH >> 1;                                   // plane row stride in bf16 units
            if (cc2 < C2) {
                const float bias = A.b2[cc2], w = A.w3[cc2];
                const int sh2 = cc2 & 3;
                for (int gl = slot2; gl < tl.ng; gl += nslot2) {
                    const int b = sGptr[gl], e = sGptr[gl + 1], n = e - b;
                    if (n <= 0) continue;
                    const float inv_n = 1.f / (float)n;
                    const float mu = mean2[(size_t)(tl.g0 + gl) * C2 + cc2], rs = rstd2[(size_t)(tl.g0 + gl) * C2 + cc2];
                    float yv[B_CB], dy[B_CB], dzv[B_CB];
                    auto batch = [&](int r0) {
                        unsigned kb[B_CB];
#pragma unroll
                        for (int j = 0; j < B_CB; ++j) {
                            const int r = min(r0 + j, e - 1);
                            const int m = EDGE ? sRowId[r] : tl.row0 + r;
                            yv[j] = A.h2[(size_t)m * C2 + cc2];
                            kb[j] = F2[r * FB2 + (cc2 >> 2)];
                            dzv[j] = sDz[r];
                        }
#pragma unroll
                        for (int j = 0; j < B_CB; ++j) {
                            yv[j] = ((yv[j] + bias) - mu) * rs;
                            const bool on = (((kb[j] >> sh2) & 1u) != 0u) & (yv[j] > 0.f);
                            dy[j] = on ? dzv[j] * w * sc : 0.f;
                        }
                    };
                    float s1 = 0.f, s2 = 0.f;
                    for (int r0 = b; r0 < e; r0 += B_CB) {
                        batch(r0);
#pragma unroll
                        for (int j = 0; j < B_CB; ++j) {
                            const bool live = r0 + j < e;
                            s1 += live ? dy[j] : 0.f;
                            s2 += live ? dy[j] * yv[j] : 0.f;
                            dw3acc += (live & (dy[j] != 0.f)) ? dzv[j] * (yv[j] * sc) : 0.f;
                            db3acc += (live & (cc2 == 0)) ? dzv[j] : 0.f;
                        }
                    }
                    const float S1 = s1 * inv_n, S2 = s2 * inv_n;
                    for (int r0 = b; r0 < e; r0 += B_CB) {
                        batch(r0);
#pragma unroll
                        for (int j = 0; j < B_CB; ++j) {
                            const int r = min(r0 + j, e - 1);
                            const float dh = rs * (dy[j] - S1 - yv[j] * S2);
                            const unsigned short hb = bf16_bits(dh);
                            ph[r * SH] = hb;
                            pl[r * SH] = bf16_bits(dh - bf16_val(hb));
                        }
                    }
                }
                for (int r = nr + slot2; r < nr16; r += nslot2) { ph[r * SH] = 0; pl[r * SH] = 0; }     // k-padding rows of the dW2 product
            } else {
                for (int r = slot2; r < nr16; r += nslot2) { ph[r * SH] = 0; pl[r * SH] = 0; }          // padded channels
            }
        }
        bwd_barrier();
        BSTAMP(1);
        // ---- chunks of 64 layer-1 channels ---------------------------------------------------------------------------------------
#pragma unroll
        for (int kc = 0; kc < NCHT; ++kc) {
            if (kc * B_CH < C1) {          // (no `break`: accW[kc] keeps a compile-time index)
            // P2: da1 chunk = dh2 x W2[:, chunk] ; keep bits of the chunk
            mfma_da1(kc, nr);
            for (int i = tid; i < nr * 16; i += BT) {
                const int r = i >> 4, q = i & 15;
                unsigned bits = 15u;
                const int col = kc * B_CH + 4 * q;
                if (drop && col < C1) {
                    const float4 k = keep4f(A.mask1, A.seed, 1, sRowId[r], col, C1, A.p, true);
                    bits = (k.x != 0.f ? 1u : 0u) | (k.y != 0.f ? 2u : 0u) | (k.z != 0.f ? 4u : 0u) | (k.w != 0.f ? 8u : 0u);
                }
                F1[i] = (unsigned char)bits;
            }
            bwd_barrier();
            BSTAMP(2);
            // P3: layer-1 column pass: y1 from P (| P[src] + Q[dst]), a1 -> planes, S1', S2', dh1 -> global
            {
                const int col = kc * B_CH + cc1;
                unsigned short* const ph = reinterpret_cast<unsigned short*>(A1hi) + cc1;
                unsigned short* const pl = reinterpret_cast<unsigned short*>(A1lo) + cc1;
                constexpr int SH = B_SBA >> 1;
                if (col < C1) {
                    const float bias = A.b1[col];
                    const int sh1 = cc1 & 3;
                    for (int gl = slot1; gl < tl.ng; gl += BT / 64) {
                        const int b = sGptr[gl], e = sGptr[gl + 1], n = e - b;
                        if (n <= 0) continue;
                        const float inv_n = 1.f / (float)n;
                        const float mu = mean1[(size_t)(tl.g0 + gl) * C1 + col], rs = rstd1[(size_t)(tl.g0 + gl) * C1 + col];
                        float yv[B_CB], dy[B_CB];
                        float s1 = 0.f, s2 = 0.f;
                        for (int r0 = b; r0 < e; r0 += B_CB) {
                            unsigned kb[B_CB];
                            float q[B_CB];
#pragma unroll
                            for (int j = 0; j < B_CB; ++j) {
                                const int r = min(r0 + j, e - 1);
                                if (EDGE) { yv[j] = A.P[(size_t)sSrc[r] * C1 + col]; q[j] = A.Q[(size_t)sDst[r] * C1 + col]; }
                                else { yv[j] = A.P[(size_t)(tl.row0 + r) * C1 + col]; q[j] = 0.f; }
                                kb[j] = F1[r * 16 + (cc1 >> 2)];
                                dy[j] = T[r * B_LDT + cc1];
                            }
#pragma unroll
                            for (int j = 0; j < B_CB; ++j) {
                                if (EDGE) yv[j] += q[j];
                                yv[j] = ((yv[j] + bias) - mu) * rs;
                                const bool on = (((kb[j] >> sh1) & 1u) != 0u) & (yv[j] > 0.f);
                                const float d = on ? dy[j] * sc : 0.f;
                                const bool live = r0 + j < e;
                                s1 += live ? d : 0.f;
                                s2 += live ? d * yv[j] : 0.f;
                            }
                        }
                        const float S1 = s1 * inv_n, S2 = s2 * inv_n;
                        for (int r0 = b; r0 < e; r0 += B_CB) {
                            unsigned kb[B_CB];
                            float q[B_CB];
#pragma unroll
                            for (int j = 0; j < B_CB; ++j) {
                                const int r = min(r0 + j, e - 1);
                                if (EDGE) { yv[j] = A.P[(size_t)sSrc[r] * C1 + col]; q[j] = A.Q[(size_t)sDst[r] * C1 + col]; }
                                else { yv[j] = A.P[(size_t)(tl.row0 + r) * C1 + col]; q[j] = 0.f; }
                                kb[j] = F1[r * 16 + (cc1 >> 2)];
                                dy[j] = T[r * B_LDT + cc1];
                            }
#pragma unroll
                            for (int j = 0; j < B_CB; ++j) {
                                const int r = min(r0 + j, e - 1);
                                if (EDGE) yv[j] += q[j];
                                yv[j] = ((yv[j] + bias) - mu) * rs;
                                const bool on = (((kb[j] >> sh1) & 1u) != 0u) & (yv[j] > 0.f);
                                const float d = on ? dy[j] * sc : 0.f;
                                const float a1 = on ? yv[j] * sc : 0.f;
                                const unsigned short hb = bf16_bits(a1);
                                ph[r * SH] = hb;
                                pl[r * SH] = bf16_bits(a1 - bf16_val(hb));
                                const int m = EDGE ? sRowId[r] : tl.row0 + r;
                                A.dh1[(size_t)m * C1 + col] = rs * (d - S1 - yv[j] * S2);
                            }
                        }
                    }
                    for (int r = nr + slot1; r < nr16; r += BT / 64) { ph[r * SH] = 0; pl[r * SH] = 0; }
                } else {
                    for (int r = slot1; r < nr16; r += BT / 64) { ph[r * SH] = 0; pl[r * SH] = 0; }
                }
            }
            bwd_barrier();
            BSTAMP(3);
            // P4: dW2[:, chunk] += dh2^T a1 (wave: c2 block wave >> 1, chunk column block wave & 1); k = tile rows, zero-padded to 16
            accW[kc] = mfma_dw2(accW[kc], nr16);
            bwd_barrier();
            BSTAMP(4);
            }
        }
    }
    // ---- this workgroup's partials: dW2 [C2p, C1pad] from the accumulators, dW3 [C2p] and db3 over the column threads' slots ----------
    {
        const int ib = wave >> 1, jb = wave & 1;
        float* const pw = A.partW2 + (size_t)slab * C2p * g.C1pad;
        if (ib * 32 < C2p) {
#pragma unroll
            for (int kc = 0; kc < NCHT; ++kc) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int c2 = ib * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                    pw[(size_t)c2 * g.C1pad + kc * B_CH + jb * 32 + (lane & 31)] = accW[kc][r];
                }
            }
        }
        float* const red = reinterpret_cast<float*>(smem);                 // [nslot2][C2p] | [nslot2] (the planes are dead)
        __syncthreads();
        if (slot2 < nslot2) {
            red[slot2 * C2p + cc2] = dw3acc;
            if (cc2 == 0) red[nslot2 * C2p + slot2] = db3acc;
        }
        __syncthreads();
        if (tid < C2p) {
            float s = 0.f;
            for (int k = 0; k < nslot2; ++k) s += red[k * C2p + tid];
            A.partW3[(size_t)slab * C2p + tid] = s;
        }
        if (tid == 0) {
            float s = 0.f;
            for (int k = 0; k < nslot2; ++k) s += red[nslot2 * C2p + k];
            A.partB3[slab] = s;
        }
    }
#ifdef GSAT_FUSED_STAMPS
    BSTAMP(5);
    if (tid == 0) {
        unsigned long long* o = reinterpret_cast<unsigned long long*>(const_cast<int*>(A.counters) + 16);
        for (int i = 0; i < 8; ++i) atomicAdd(o + i, (unsigned long long)stamp[i]);
        atomicAdd(o + 14, 1ull);
    }
#endif
}

// dW2 / dW3 / db3 = sums of the per-workgroup partials in workgroup order (eight partial sums in flight: fixed order, bitwise
// reproducible); the two bias gradients that are identically zero (b1, b2 sit in front of an InstanceNorm) are cleared here too.
__global__ void k_attn_bwd_reduce(const float* __restrict__ partW2, const float* __restrict__ partW3, const float* __restrict__ partB3, int nwg,
                                  const int* __restrict__ counters, int C1, int C2, int C2p, int C1pad, float* __restrict__ dW2,
                                  float* __restrict__ dW3, float* __restrict__ db3, float* __restrict__ db1, float* __restrict__ db2) {
    // slabs that were written: workgroups [0, min(nwg, small tiles)) of the main launch, then [nwg, nwg + min(nwg, big tiles)) of the
    // big-graph launch
    const int n_small = min(nwg, counters[0] - counters[2]), n_big = min(nwg, counters[2]);
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t nW2 = (int64_t)C2 * C1;
    const float* p0;
    size_t stride;
    float* out;
    if (i < nW2) {
        const int c2 = (int)(i / C1), c1 = (int)(i % C1);
        p0 = partW2 + (size_t)c2 * C1pad + c1; stride = (size_t)C2p * C1pad; out = dW2 + i;
    } else if (i < nW2 + C2) {
        p0 = partW3 + (i - nW2); stride = (size_t)C2p; out = dW3 + (i - nW2);
    } else if (i == nW2 + C2) {
        p0 = partB3; stride = 1; out = db3;
    } else {
        const int64_t k = i - nW2 - C2 - 1;
        if (k < C1) db1[k] = 0.f; else if (k < C1 + C2) db2[k - C1] = 0.f;
        return;
    }
    float a[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    int s = 0;
    for (; s + 8 <= n_small; s += 8) {
#pragma unroll
        for (int j = 0; j < 8; ++j) a[j] += p0[(size_t)(s + j) * stride];
    }
    for (int j = 0; s < n_small; ++s, ++j) a[j] += p0[(size_t)s * stride];
    float bsum = 0.f;
    for (int k = 0; k < n_big; ++k) bsum += p0[(size_t)(nwg + k) * stride];
    *out = (((a[0] + a[1]) + (a[2] + a[3])) + ((a[4] + a[5]) + (a[6] + a[7]))) + bsum;
}

static int bwd_nwg() {
    static int cus = 0;
    if (!cus) {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0) cus = prop.multiProcessorCount;
        else cus = 256;
    }
    return cus;
}

static bool bwd_geometry(int C1, int C2, BwdGeom* out) {
    if (C1 < 4 || C1 % 4 || C1 > 512 || C2 < 4 || C2 % 4 || C2 > 128) return false;
    BwdGeom g{};
    g.C1 = C1; g.C2 = C2; g.C2p = (C2 + 31) / 32 * 32;
    g.NCH = (C1 + B_CH - 1) / B_CH;
    g.S2b = g.C2p / 16;
    g.SBH = g.C2p * 2 + 16;
    const int ncht = g.NCH <= 2 ? 2 : g.NCH <= 4 ? 4 : 8;
    g.C1pad = ncht * B_CH;
    g.offA1 = 2 * B_RM * g.SBH;
    g.offT = g.offA1 + 2 * B_RM * B_SBA;
    g.offF2 = g.offT + B_RM * B_LDT * 4;
    g.offF1 = g.offF2 + B_RM * (g.C2p / 4);
    g.offMeta = g.offF1 + B_RM * 16;
    g.lds_bytes = g.offMeta + (32 + 3 * B_RM) * 4 + B_RM * 4;
    if (g.lds_bytes > 160 * 1024) return false;
    *out = g;
    return true;
}

bool attn_fused_bwd_eligible(const gsat_attn_args* a) {
    const char* env = getenv("GSAT_ATTN_BWD_FUSED");
    if (!(env && atoi(env) != 0)) return false;          // opt-in while the path for graphs larger than a tile is being built
    BwdGeom g;
    if (a->M <= 0 || !attn_plan_ok(a->G)) return false;
    if (!a->edge_mode && a->seg_order) return false;
    if (!bwd_geometry(a->C1, a->C2, &g)) return false;
    if (a->M > a->G * (int64_t)(2 * B_RM)) return false;          // batches of huge graphs: the streaming pipeline
    return true;
}

size_t attn_fused_bwd_ws_bytes(const gsat_attn_args* a) {
    BwdGeom g;
    if (!bwd_geometry(a->C1, a->C2, &g)) return 0;
    const size_t nwg = (size_t)bwd_nwg();
    size_t b = 256 + attn_plan_bytes(a->G);
    b += align_up((size_t)g.NCH * 2 * g.S2b * 2 * 64 * 16, 256);
    b += align_up(2 * nwg * g.C2p * g.C1pad * 4, 256) + align_up(2 * nwg * g.C2p * 4, 256) + align_up(2 * nwg * 4, 256);      // small + big launch
    return b;
}

template <bool EDGE, int NCHT, bool BIG>
static int launch_bwd1(hipStream_t stream, const BwdArgs& ba, int grid) {
    static size_t allowed = 64 * 1024;
    const size_t lds = (size_t)ba.g.lds_bytes;
    if (lds > allowed) {
        GSAT_CHECK_HIP(hipFuncSetAttribute((const void*)k_attn_fused_bwd<EDGE, NCHT, BIG>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        allowed = lds;
    }
    k_attn_fused_bwd<EDGE, NCHT, BIG><<<grid, BT, lds, stream>>>(ba);
    GSAT_LAUNCH_CHECK();
    return GSAT_OK;
}
// whole-graph tiles, then (same grid; workgroups without a big tile exit at once) the graphs larger than a tile
template <bool EDGE, int NCHT>
static int launch_bwd(hipStream_t stream, const BwdArgs& ba, int grid) {
    const int rc = launch_bwd1<EDGE, NCHT, false>(stream, ba, grid);
    return rc ? rc : launch_bwd1<EDGE, NCHT, true>(stream, ba, grid);
}

// dh1 [M, C1] <- everything from (dlogits, datt) down to the layer-1 pre-activation; dW2, dW3, db3 written, db1 / db2 cleared
int attn_fused_bwd(hipStream_t stream, const gsat_attn_args* a, const gsat_attn_grads* gr, float* dh1, void* ws, size_t ws_bytes) {
    BwdGeom g;
    GSAT_REQUIRE(bwd_geometry(a->C1, a->C2, &g), GSAT_ERR_UNSUPPORTED, "attn_fused_bwd: unsupported widths");
    const size_t need = attn_fused_bwd_ws_bytes(a);
    GSAT_REQUIRE(ws && ws_bytes >= need, GSAT_ERR_WORKSPACE, "gsat_attn_bwd: workspace %zu < %zu (fused path)", ws_bytes, need);
    const int nwg = (int)std::min<int64_t>(bwd_nwg(), a->G);
    char* w = static_cast<char*>(ws);
    int* counters = reinterpret_cast<int*>(w); w += 256;
    FTile* tiles = reinterpret_cast<FTile*>(w); w += attn_plan_bytes(a->G);
    uint4* Wq2 = reinterpret_cast<uint4*>(w); w += align_up((size_t)g.NCH * 2 * g.S2b * 2 * 64 * 16, 256);
    float* partW2 = reinterpret_cast<float*>(w); w += align_up((size_t)2 * bwd_nwg() * g.C2p * g.C1pad * 4, 256);
    float* partW3 = reinterpret_cast<float*>(w); w += align_up((size_t)2 * bwd_nwg() * g.C2p * 4, 256);
    float* partB3 = reinterpret_cast<float*>(w);
#ifdef GSAT_FUSED_STAMPS
    extern int* g_last_bwd_counters_ref(int*);
    g_last_bwd_counters_ref(counters);
#endif
    int rc = attn_plan_launch(stream, a->seg_ptr, a->seg_ptr, a->G, B_RM, B_RM, tiles, counters);
    if (rc) return rc;
    {
        const int64_t n = (int64_t)g.NCH * 2 * g.S2b * 64;
        k_attn_bwd_pack<<<(unsigned)std::min<int64_t>(ceil_div(n, 256), 256), 256, 0, stream>>>(a->W2, a->C1, a->C2, g.NCH, g.S2b, Wq2);
        GSAT_LAUNCH_CHECK();
    }
    BwdArgs ba{};
    ba.h2 = a->h2; ba.b2 = a->b2; ba.w3 = a->W3; ba.P = a->P; ba.Q = a->Q; ba.b1 = a->b1; ba.stats = a->stats;
    ba.dlogits = gr->dlogits; ba.datt = gr->datt; ba.att = a->att; ba.mask1 = a->mask1; ba.mask2 = a->mask2;
    ba.src = a->src; ba.dst = a->dst; ba.seg_ptr = a->seg_ptr; ba.order = a->seg_order;
    ba.Wq2 = Wq2; ba.dh1 = dh1; ba.partW2 = partW2; ba.partW3 = partW3; ba.partB3 = partB3;
    ba.tiles = tiles; ba.counters = counters; ba.M = a->M; ba.G = a->G;
    ba.seed = SeedRef{a->seed, a->seed_dev}; ba.p = a->p_drop; ba.training = a->training; ba.g = g;
    const int ncht = g.C1pad / B_CH;
#define GO(E, N) rc = launch_bwd<E, N>(stream, ba, nwg)
    if (a->edge_mode) { if (ncht == 2) GO(true, 2); else if (ncht == 4) GO(true, 4); else GO(true, 8); }
    else { if (ncht == 2) GO(false, 2); else if (ncht == 4) GO(false, 4); else GO(false, 8); }
#undef GO
    if (rc) return rc;
    const int64_t total = (int64_t)a->C2 * a->C1 + a->C2 + 1 + a->C1 + a->C2;
    k_attn_bwd_reduce<<<(unsigned)ceil_div(total, 256), 256, 0, stream>>>(partW2, partW3, partB3, nwg, counters, a->C1, a->C2, g.C2p, g.C1pad, gr->dW2,
                                                                           gr->dW3, gr->db3, gr->db1, gr->db2);
    GSAT_LAUNCH_CHECK();
    return GSAT_OK;
}

}  // namespace gsat

#ifdef GSAT_FUSED_STAMPS
// diagnostic builds only: host copy of the counters block of the last fused backward (tiles, -, big tiles, ..., stamp words from [16])
static int* g_last_bwd_counters = nullptr;
namespace gsat { int* g_last_bwd_counters_ref(int* p) { g_last_bwd_counters = p; return p; } }
extern "C" const int* gsat_debug_bwd_counters(void) {
    static int host[64];
    if (!g_last_bwd_counters) return nullptr;
    if (hipDeviceSynchronize() != hipSuccess) return nullptr;
    if (hipMemcpy(host, g_last_bwd_counters, sizeof(host), hipMemcpyDeviceToHost) != hipSuccess) return nullptr;
    return host;
}
#endif
