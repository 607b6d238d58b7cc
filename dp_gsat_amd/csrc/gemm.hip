// fp32-input MFMA GEMM for the attention extractor (gfx950: v_mfma_f32_32x32x2_f32, exact fp32 FMA chain,
// 64 FLOP/clk/SIMD).  Shapes here are tall and skinny (rows = nodes/edges, 16..1024 channels), where the
// vendor sgemm heuristics pick poor kernels (1.4 ms for a 51k x 256 x 128 product, profiles/r01_c3_*).
//
// Tile: 128 x 128 x 32 per 256-thread workgroup, 4 wavefronts as 2 x 2, each 64 x 64 = 2 x 2 MFMA tiles
// (64 accumulator registers).  Operands are staged global -> registers -> LDS with the next K-slab's loads
// in flight under the current slab's MFMAs (two LDS buffers, one barrier per slab).  LDS images are
// k-major ([k][m] and [k][n], row stride 132 floats) so an MFMA operand fetch is one conflict-free
// ds_read_b32 per lane.
//
// Three operand layouts cover forward, backward-data and backward-weight:
//   A_T = false: A is [M,K] row-major (k contiguous)      A_T = true: A is [K,M] row-major (reduction-major)
//   B_T = true : B is [N,K] row-major (nn.Linear weight)  B_T = false: B is [K,N] row-major
// Weight gradients reduce over the row dimension (K = #rows, huge; M x N small): blockIdx.z splits K and
// writes partial slabs that a second kernel sums in slab order (deterministic, no atomics).
#include "common.h"
#include "pna_math.h"
#include <cstdlib>

namespace gsat {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int GK = 32, GT = 256;      // block tile (64*TM) x (64*TN) x GK, TM, TN = 1 | 2; 4 waves as 2 x 2
constexpr int LD_KC = 129;   // k-contiguous source: transposing scalar stores, odd stride -> conflict-free
constexpr int LD_KM = 132;   // k-major source: 16-byte vector stores need a multiple of 4

// stage one 128(rows) x 32(k) slab of a k-contiguous operand:  4 float4 per thread
struct StageKC { float4 v[4]; };   // 128 (or 64: two entries unused) rows/columns x 32 k
// X[r][k], r in [r0, r0+128), k in [k0, k0+32): thread t -> row (t>>3) + 32p, k-quad t&7
template <int ROWS>
__device__ __forceinline__ void load_kcontig(StageKC& s, const float* __restrict__ X, int64_t ld, int r0, int R, int k0, int K, int t) {
    const int kq = (t & 7) * 4;
#pragma unroll
    for (int p = 0; p < ROWS / 32; ++p) {
        const int r = r0 + (t >> 3) + 32 * p;
        s.v[p] = (r < R && k0 + kq < K) ? ld4(X + (size_t)r * ld + k0 + kq) : f4zero();
    }
}
template <int ROWS>
__device__ __forceinline__ void store_kcontig(const StageKC& s, float* __restrict__ L, int t) {   // L[k][LD_KC]
    const int kq = (t & 7) * 4;
#pragma unroll
    for (int p = 0; p < ROWS / 32; ++p) {
        const int r = (t >> 3) + 32 * p;
        L[(kq + 0) * LD_KC + r] = s.v[p].x;
        L[(kq + 1) * LD_KC + r] = s.v[p].y;
        L[(kq + 2) * LD_KC + r] = s.v[p].z;
        L[(kq + 3) * LD_KC + r] = s.v[p].w;
    }
}
// X[k][c], k in [k0,k0+32), c in [c0,c0+COLS): thread t -> k (t / (COLS/4)) + (1024/COLS) p, column quad t % (COLS/4)
template <int COLS>
__device__ __forceinline__ void load_kmajor(StageKC& s, const float* __restrict__ X, int64_t ld, int c0, int Ccols, int k0, int K, int t) {
    constexpr int QPR = COLS / 4, KPP = GT / QPR;          // quads per k-row, k-rows per pass
    const int cq = (t % QPR) * 4;
#pragma unroll
    for (int p = 0; p < GK / KPP; ++p) {
        const int k = k0 + t / QPR + KPP * p;
        s.v[p] = (k < K && c0 + cq < Ccols) ? ld4(X + (size_t)k * ld + c0 + cq) : f4zero();
    }
}
template <int COLS>
__device__ __forceinline__ void store_kmajor(const StageKC& s, float* __restrict__ L, int t) {
    constexpr int QPR = COLS / 4, KPP = GT / QPR;
    const int cq = (t % QPR) * 4;
#pragma unroll
    for (int p = 0; p < GK / KPP; ++p) st4(L + (t / QPR + KPP * p) * LD_KM + cq, s.v[p]);
}

// Logical tile of this workgroup.  Workgroups are dealt to the 8 XCDs round-robin in linear launch order, so the column
// tiles of ONE row tile (consecutive blockIdx.x) would land on different XCDs and each L2 would fetch the same A rows again.
// The remap hands every XCD a contiguous range of the linear tile list: tiles that share an operand slab run on the same
// L2, back to back (post_nn dW 86 -> 79 us, c5s layer 2 978 -> 906 us; GSAT_GEMM_XCD=0 switches it off for comparison).
__device__ int GEMM_XCD_REMAP_ON = 1;
__device__ __forceinline__ void gemm_tile_coords(int& bx, int& by, int& bz) {
    const unsigned gx = gridDim.x, gy = gridDim.y, total = gx * gy * gridDim.z;
    const unsigned L = (blockIdx.z * gy + blockIdx.y) * gx + blockIdx.x;
    if (!GEMM_XCD_REMAP_ON) { bx = blockIdx.x; by = blockIdx.y; bz = blockIdx.z; return; }
    const unsigned q = total >> 3, r = total & 7, xcd = L & 7, i = L >> 3;
    const unsigned t = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + i;
    bx = (int)(t % gx); by = (int)((t / gx) % gy); bz = (int)(t / (gx * gy));
}

template <bool A_T, bool B_T, int TM, int TN>
__global__ __launch_bounds__(GT) void k_gemm_f32(const float* __restrict__ A, int64_t lda, const float* __restrict__ B, int64_t ldb,
                                                 float* __restrict__ C, int64_t ldc, int M, int N, int K, int k_per_split,
                                                 const float* __restrict__ bias, int accumulate, size_t slab_stride) {
    constexpr int GM = 64 * TM, GN = 64 * TN;
    constexpr int LDA = A_T ? LD_KM : LD_KC, LDB = B_T ? LD_KC : LD_KM;
    constexpr int EP_LD = 36;                                // epilogue staging: 32 x 32 tile per wave, padded rows (16-B aligned)
    static_assert(GK * LDA + GK * LDB >= 4 * 32 * EP_LD, "epilogue staging must fit in the operand images");
    __shared__ __attribute__((aligned(16))) float lds[GK * LDA + GK * LDB];
    float* const As = lds;
    float* const Bs = lds + GK * LDA;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    int bx, by, bz;
    gemm_tile_coords(bx, by, bz);
    const int m0 = by * GM, n0 = bx * GN;
    const int kbeg = bz * k_per_split, kend = min(K, kbeg + k_per_split);
    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    StageKC sa, sb;
    auto gload = [&](int k0) {
        if (A_T) load_kmajor<GM>(sa, A, lda, m0, M, k0, kend, t); else load_kcontig<GM>(sa, A, lda, m0, M, k0, kend, t);
        if (B_T) load_kcontig<GN>(sb, B, ldb, n0, N, k0, kend, t); else load_kmajor<GN>(sb, B, ldb, n0, N, k0, kend, t);
    };
    auto lstore = [&]() {
        if (A_T) store_kmajor<GM>(sa, As, t); else store_kcontig<GM>(sa, As, t);
        if (B_T) store_kcontig<GN>(sb, Bs, t); else store_kmajor<GN>(sb, Bs, t);
    };
    if (kbeg < kend) {
        gload(kbeg);
        lstore();
    }
    __syncthreads();
    const int arow = wm * 32 * TM + (lane & 31), bcol = wn * 32 * TN + (lane & 31), kh = lane >> 5;
    for (int k0 = kbeg; k0 < kend; k0 += GK) {
        const bool more = k0 + GK < kend;
        if (more) gload(k0 + GK);                       // next slab's global loads fly under this slab's MFMAs
#pragma unroll
        for (int kk = 0; kk < GK / 2; ++kk) {
            const int ka = (2 * kk + kh) * LDA, kb = (2 * kk + kh) * LDB;
            float a[TM], b[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) a[i] = As[ka + arow + 32 * i];
#pragma unroll
            for (int j = 0; j < TN; ++j) b[j] = Bs[kb + bcol + 32 * j];
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int i = 0; i < TM; ++i) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[j], acc[i][j], 0, 0, 0);
        }
        __syncthreads();                                // every wave is done reading this slab
        if (more) lstore();
        __syncthreads();
    }
    // ---- epilogue -----------------------------------------------------------------------------------------------
    // C/D map of the 32x32 MFMA: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5): a lane holds ONE column, so storing
    // straight from registers is 4 bytes per lane per instruction.  Each 32x32 tile goes through a per-wave LDS patch
    // instead and leaves as 16-byte row segments (4 store instructions per tile instead of 16).
    float* Cb = C + (size_t)bz * slab_stride;
    float* patch = lds + wave * (32 * EP_LD);               // all waves passed the loop's final barrier: As/Bs are free
    const int prow = lane >> 3, pcol = (lane & 7) * 4;      // read-back: 8 lanes per row, 8 rows per pass
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
#pragma unroll
            for (int r = 0; r < 16; ++r) patch[((r & 3) + 8 * (r >> 2) + 4 * kh) * EP_LD + (lane & 31)] = acc[i][j][r];
            __builtin_amdgcn_wave_barrier();
            const int col = n0 + wn * 32 * TN + j * 32 + pcol;
            float4 bv = f4zero();
            if (bias && col + 3 < N) bv = ld4(bias + col);
            else if (bias && col < N) { bv.x = bias[col]; if (col + 1 < N) bv.y = bias[col + 1]; if (col + 2 < N) bv.z = bias[col + 2]; }
#pragma unroll
            for (int p = 0; p < 4; ++p) {
                const int rr = prow + 8 * p;
                const int row = m0 + wm * 32 * TM + i * 32 + rr;
                float4 v = ld4(patch + rr * EP_LD + pcol);
                if (row < M && col < N) {
                    float* dst = Cb + (size_t)row * ldc + col;
                    v = make_float4(v.x + bv.x, v.y + bv.y, v.z + bv.z, v.w + bv.w);
                    if (col + 3 < N) {
                        if (accumulate & 1) { float4 o = ld4(dst); v = make_float4(v.x + o.x, v.y + o.y, v.z + o.z, v.w + o.w); }
                        if (accumulate & 2) st4_nt(dst, v); else st4(dst, v);      // bit 1: streaming output (not re-read soon)
                    } else {                                 // ragged right edge (N % 4 != 0 only for k-contiguous B)
                        const float e[4] = {v.x, v.y, v.z, v.w};
                        for (int q = 0; q < 4 && col + q < N; ++q) dst[q] = (accumulate & 1) ? dst[q] + e[q] : e[q];
                    }
                }
            }
            __builtin_amdgcn_wave_barrier();
        }
}


// ================================================================================================================
// Split-bf16 ("bf16x3") variant for large products: x = hi + lo with hi = bf16(x), lo = bf16(x - hi); the product is
// accumulated in fp32 as hi*hi + hi*lo + lo*hi on v_mfma_f32_32x32x16_bf16 (16x the fp32 MFMA rate per instruction, so
// ~5x per product at three instructions).  The dropped lo*lo term and the rounding of lo are ~2^-17 relative to |a||b|,
// i.e. the result stays inside the 1e-4 parity band with two orders of magnitude to spare (tests/test_gpu_gemm.py).
// LDS holds four bf16 planes (A_hi, A_lo, B_hi, B_lo) in [row][k] order with 80-byte rows (64 B of k + 16 B pad: the
// ds_read_b128 operand fetches of a 16-lane group then hit 16 distinct 16-byte bank slots).
// ================================================================================================================
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
constexpr int BRS = 80;                              // bytes per LDS row of one bf16 plane (32 k)

__device__ __forceinline__ unsigned pack_bf16(float a, float b) {       // two RNE conversions packed as (a | b << 16)
    typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
    bf16x2 v = {(__bf16)a, (__bf16)b};
    return *reinterpret_cast<unsigned*>(&v);
}
__device__ __forceinline__ float bf16_hi(float x) { return (float)(__bf16)x; }

// k-contiguous source X[r][k]: thread t -> rows (t>>3) + 32p, k-quad t&7 ; writes 4 bf16 (8 B) per plane
template <int ROWS>
__device__ __forceinline__ void store_split_kcontig(const StageKC& s, unsigned char* __restrict__ hi, unsigned char* __restrict__ lo, int t) {
    const int kq = (t & 7) * 4;
#pragma unroll
    for (int p = 0; p < ROWS / 32; ++p) {
        const int r = (t >> 3) + 32 * p;
        const float4 v = s.v[p];
        const float hx = bf16_hi(v.x), hy = bf16_hi(v.y), hz = bf16_hi(v.z), hw = bf16_hi(v.w);
        uint2 H = make_uint2(pack_bf16(v.x, v.y), pack_bf16(v.z, v.w));
        uint2 L = make_uint2(pack_bf16(v.x - hx, v.y - hy), pack_bf16(v.z - hz, v.w - hw));
        *reinterpret_cast<uint2*>(hi + r * BRS + kq * 2) = H;
        *reinterpret_cast<uint2*>(lo + r * BRS + kq * 2) = L;
    }
}
// k-major source X[k][c] (weight-gradient operands, W of the dx product): the LDS image is [k/8][col][8 bf16], so a lane
// that owns 8 consecutive k of one column writes one 16-byte chunk and consecutive lanes write consecutive chunks
// (conflict-free ds_write_b128), while the MFMA operand fetch of lane (r, h) at step ks is chunk (2 ks + h) of column r:
// again consecutive lanes, consecutive 16 bytes.  Work item = 8 k x 2 columns: 8 coalesced float2 loads.
struct StageKM { float2 v[8]; };
template <int COLS>
__device__ __forceinline__ void load_split_kmajor(StageKM& s, const float* __restrict__ X, int64_t ld, int c0, int Ccols, int k0, int K, int t) {
    constexpr int PAIRS = COLS / 2;                       // 64 (COLS = 128) or 32 (COLS = 64) column pairs x 4 k-octets
    if (t >= PAIRS * 4) return;
    const int cp = (t % PAIRS) * 2, k8 = t / PAIRS;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int k = k0 + k8 * 8 + j;
        s.v[j] = (k < K && c0 + cp < Ccols) ? *reinterpret_cast<const float2*>(X + (size_t)k * ld + c0 + cp) : make_float2(0.f, 0.f);
    }
}
// the same two stagings for the PNA aggregate as a virtual operand (pna_math.h: PnaVirt): x_j columns from the compact aggregate, x_i
// columns recomputed from x and the row's coefficient pair.  A 32-wide k slab / a COLS-wide column tile lies inside ONE segment (the
// dispatch requires H % 32 == 0 resp. H % COLS == 0), so the choice is uniform over the workgroup.  Loads are issued unconditionally
// (row clamped) into the stage registers like the plain loaders'; the arithmetic runs when the slab is stored to LDS, a whole MFMA slab
// later, so nothing waits on a load.
template <int ROWS>
__device__ __forceinline__ void load_kcontig_virt(StageKC& s, const PnaVirt& v, int r0, int R, int k0, int t) {
    const int seg = k0 / v.H, c = k0 - seg * v.H + (t & 7) * 4, a = seg >> 1;
#pragma unroll
    for (int p = 0; p < ROWS / 32; ++p) {
        const int r = min(r0 + (t >> 3) + 32 * p, R - 1);
        s.v[p] = (seg & 1) ? ld4(v.aggj + (size_t)r * (v.NAGG * v.H) + a * v.H + c) : ld4(v.x + (size_t)r * v.H + c);
    }
}
// coef: LDS image of scal rows [r0, r0 + ROWS) (8 floats each), staged once per workgroup
template <int ROWS>
__device__ __forceinline__ void synth_kcontig_virt(StageKC& s, const float* coef, const PnaVirt& v, int r0, int R, int k0, int t) {
    const int seg = k0 / v.H, a = seg >> 1, co = pna_coef_offset(a);
    if (seg & 1) return;                                         // x_j columns: as loaded (rows beyond R are never stored to C)
#pragma unroll
    for (int p = 0; p < ROWS / 32; ++p) {
        const float2 cf = *reinterpret_cast<const float2*>(coef + ((t >> 3) + 32 * p) * 8 + co);
        const float4 x = s.v[p];
        s.v[p] = make_float4(pna_self(a, x.x, cf), pna_self(a, x.y, cf), pna_self(a, x.z, cf), pna_self(a, x.w, cf));
    }
}
template <int COLS>
__device__ __forceinline__ void load_split_kmajor_virt(StageKM& s, const PnaVirt& v, int c0, int k0, int K, int t) {
    constexpr int PAIRS = COLS / 2;
    if (t >= PAIRS * 4) return;
    const int cp = (t % PAIRS) * 2, k8 = t / PAIRS;
    const int seg = c0 / v.H, c = c0 - seg * v.H + cp, a = seg >> 1;
    const float* base = (seg & 1) ? v.aggj + a * v.H + c : v.x + c;
    const size_t ld = (seg & 1) ? (size_t)v.NAGG * v.H : (size_t)v.H;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int k = min(k0 + k8 * 8 + j, K - 1);
        s.v[j] = *reinterpret_cast<const float2*>(base + (size_t)k * ld);
    }
}
// coef: LDS image of scal rows [k0, k0 + 32) of the slab
template <int COLS>
__device__ __forceinline__ void synth_kmajor_virt(StageKM& s, const float* coef, const PnaVirt& v, int c0, int k0, int K, int t) {
    constexpr int PAIRS = COLS / 2;
    if (t >= PAIRS * 4) return;
    const int k8 = t / PAIRS;
    const int seg = c0 / v.H, a = seg >> 1, co = pna_coef_offset(a);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        float2 x = s.v[j];
        if (!(seg & 1)) {
            const float2 cf = *reinterpret_cast<const float2*>(coef + (k8 * 8 + j) * 8 + co);
            x = make_float2(pna_self(a, x.x, cf), pna_self(a, x.y, cf));
        }
        s.v[j] = k0 + k8 * 8 + j < K ? x : make_float2(0.f, 0.f);          // rows beyond K must not enter the reduction
    }
}
template <int COLS>
__device__ __forceinline__ void store_split_kmajor(const StageKM& s, unsigned char* __restrict__ hi, unsigned char* __restrict__ lo, int t) {
    constexpr int PAIRS = COLS / 2;
    if (t >= PAIRS * 4) return;
    const int cp = (t % PAIRS) * 2, k8 = t / PAIRS;
#pragma unroll
    for (int c = 0; c < 2; ++c) {
        float x[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) x[j] = c == 0 ? s.v[j].x : s.v[j].y;
        uint4 H = make_uint4(pack_bf16(x[0], x[1]), pack_bf16(x[2], x[3]), pack_bf16(x[4], x[5]), pack_bf16(x[6], x[7]));
        uint4 L = make_uint4(pack_bf16(x[0] - bf16_hi(x[0]), x[1] - bf16_hi(x[1])), pack_bf16(x[2] - bf16_hi(x[2]), x[3] - bf16_hi(x[3])),
                             pack_bf16(x[4] - bf16_hi(x[4]), x[5] - bf16_hi(x[5])), pack_bf16(x[6] - bf16_hi(x[6]), x[7] - bf16_hi(x[7])));
        const int off = (k8 * COLS + cp + c) * 16;
        *reinterpret_cast<uint4*>(hi + off) = H;
        *reinterpret_cast<uint4*>(lo + off) = L;
    }
}

// VIRT = 1: A (k-contiguous, !A_T) is the virtual PNA aggregate `pv` (post_nn forward); VIRT = 2: B (k-major, !B_T) is (post_nn dW)
template <bool A_T, bool B_T, int TM, int TN, int VIRT = 0>
__global__ __launch_bounds__(GT, (VIRT == 1 && TM == 2 && TN == 2) ? 3 : 1) void k_gemm_bf16x3(const float* __restrict__ A, int64_t lda, const float* __restrict__ B, int64_t ldb,
                                                    float* __restrict__ C, int64_t ldc, int M, int N, int K, int k_per_split,
                                                    const float* __restrict__ bias, int accumulate, size_t slab_stride, const PnaVirt pv = PnaVirt{}) {
    constexpr int GM = 64 * TM, GN = 64 * TN;
    constexpr int EP_LD = 36;
    constexpr int PLANE_A = GM * BRS, PLANE_B = GN * BRS;
    static_assert(2 * PLANE_A + 2 * PLANE_B >= 4 * 32 * EP_LD * 4, "epilogue staging must fit in the operand planes");
    __shared__ __attribute__((aligned(16))) unsigned char lds[2 * PLANE_A + 2 * PLANE_B];
    unsigned char* const Ahi = lds;
    unsigned char* const Alo = lds + PLANE_A;
    unsigned char* const Bhi = lds + 2 * PLANE_A;
    unsigned char* const Blo = lds + 2 * PLANE_A + PLANE_B;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    int bx, by, bz;
    gemm_tile_coords(bx, by, bz);
    const int m0 = by * GM, n0 = bx * GN;
    const int kbeg = bz * k_per_split, kend = min(K, kbeg + k_per_split);
    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    StageKC sa, sb;
    StageKM ma, mb;
    // virtual operand: the rows' coefficient records (8 floats) live in LDS -- VIRT = 1: the tile's GM rows, staged once; VIRT = 2: the 32 rows
    // of a slab, double-buffered and fetched one slab ahead of the rows themselves (the record of slab i + 2 is requested during slab i, written
    // to the buffer slab i used while slab i + 1 is converted: every hand-over has a barrier in between)
    __shared__ __attribute__((aligned(16))) float coef[VIRT == 1 ? GM * 8 : (VIRT == 2 ? 2 * GK * 8 : 4)];
    int kv = kbeg;                                             // first k of the slab in the stage registers
    float4 creg = f4zero();
    auto cload = [&](int k0) {                                 // VIRT = 2: 64 threads fetch the 32 records of slab k0 (clamped)
        if (VIRT == 2 && t < GK * 2) creg = ld4(pv.scal + (size_t)min(k0 + (t >> 1), max(kend - 1, 0)) * 8 + (t & 1) * 4);
    };
    auto cstore = [&](int k0) {
        if (VIRT == 2 && t < GK * 2) st4(coef + (((k0 - kbeg) / GK) & 1) * (GK * 8) + t * 4, creg);
    };
    if (VIRT == 1) {
        for (int i = t; i < GM * 2; i += GT) st4(coef + i * 4, ld4(pv.scal + (size_t)min(m0 + (i >> 1), M - 1) * 8 + (i & 1) * 4));
        __syncthreads();
    }
    if (VIRT == 2 && kbeg < kend) {
        cload(kbeg); cstore(kbeg);
        cload(kbeg + GK); cstore(kbeg + GK);
        __syncthreads();
    }
    auto gload = [&](int k0) {
        if (A_T) load_split_kmajor<GM>(ma, A, lda, m0, M, k0, kend, t);
        else if (VIRT == 1) { load_kcontig_virt<GM>(sa, pv, m0, M, k0, t); kv = k0; }
        else load_kcontig<GM>(sa, A, lda, m0, M, k0, kend, t);
        if (B_T) load_kcontig<GN>(sb, B, ldb, n0, N, k0, kend, t);
        else if (VIRT == 2) { load_split_kmajor_virt<GN>(mb, pv, n0, k0, kend, t); kv = k0; cload(k0 + GK); }
        else load_split_kmajor<GN>(mb, B, ldb, n0, N, k0, kend, t);
    };
    auto lstore = [&]() {
        if (VIRT == 1) synth_kcontig_virt<GM>(sa, coef, pv, m0, M, kv, t);
        if (VIRT == 2) synth_kmajor_virt<GN>(mb, coef + (((kv - kbeg) / GK) & 1) * (GK * 8), pv, n0, kv, kend, t);
        if (A_T) store_split_kmajor<GM>(ma, Ahi, Alo, t); else store_split_kcontig<GM>(sa, Ahi, Alo, t);
        if (B_T) store_split_kcontig<GN>(sb, Bhi, Blo, t); else store_split_kmajor<GN>(mb, Bhi, Blo, t);
    };
    if (kbeg < kend) {
        gload(kbeg);
        lstore();
    }
    __syncthreads();
    const int r32 = lane & 31, kh = lane >> 5;
    // byte offset of the 16-byte operand chunk of (tile row/col `rc`, MFMA step ks): [row][k] image vs [k/8][col][8] image
    auto a_chunk = [&](int i, int ks) { const int rc = wm * 32 * TM + 32 * i + r32; return A_T ? ((2 * ks + kh) * GM + rc) * 16 : rc * BRS + ks * 32 + kh * 16; };
    auto b_chunk = [&](int j, int ks) { const int rc = wn * 32 * TN + 32 * j + r32; return B_T ? rc * BRS + ks * 32 + kh * 16 : ((2 * ks + kh) * GN + rc) * 16; };
    for (int k0 = kbeg; k0 < kend; k0 += GK) {
        const bool more = k0 + GK < kend;
        if (more) gload(k0 + GK);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {              // two 16-deep MFMA steps per 32-deep slab
            bf16x8 ah[TM], al[TM], bh[TN], bl[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                ah[i] = *reinterpret_cast<const bf16x8*>(Ahi + a_chunk(i, ks));
                al[i] = *reinterpret_cast<const bf16x8*>(Alo + a_chunk(i, ks));
            }
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                bh[j] = *reinterpret_cast<const bf16x8*>(Bhi + b_chunk(j, ks));
                bl[j] = *reinterpret_cast<const bf16x8*>(Blo + b_chunk(j, ks));
            }
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int i = 0; i < TM; ++i) {
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[i], bh[j], acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bl[j], acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bh[j], acc[i][j], 0, 0, 0);
                }
        }
        __syncthreads();
        if (more) {
            lstore();
            if (VIRT == 2) cstore(k0 + 2 * GK);                // record of slab k0 + 2 GK (requested by gload(k0 + GK)) -> the buffer slab k0 used
        }
        __syncthreads();
    }
    // ---- epilogue: identical to the fp32 kernel (32x32 patch per wave through LDS, 16-byte row stores) --------------
    float* Cb = C + (size_t)bz * slab_stride;
    float* patch = reinterpret_cast<float*>(lds) + wave * (32 * EP_LD);
    const int prow = lane >> 3, pcol = (lane & 7) * 4;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
#pragma unroll
            for (int r = 0; r < 16; ++r) patch[((r & 3) + 8 * (r >> 2) + 4 * kh) * EP_LD + r32] = acc[i][j][r];
            __builtin_amdgcn_wave_barrier();
            const int col = n0 + wn * 32 * TN + j * 32 + pcol;
            float4 bv = f4zero();
            if (bias && col + 3 < N) bv = ld4(bias + col);
            else if (bias && col < N) { bv.x = bias[col]; if (col + 1 < N) bv.y = bias[col + 1]; if (col + 2 < N) bv.z = bias[col + 2]; }
#pragma unroll
            for (int p = 0; p < 4; ++p) {
                const int rr = prow + 8 * p;
                const int row = m0 + wm * 32 * TM + i * 32 + rr;
                float4 v = ld4(patch + rr * EP_LD + pcol);
                if (row < M && col < N) {
                    float* dst = Cb + (size_t)row * ldc + col;
                    v = make_float4(v.x + bv.x, v.y + bv.y, v.z + bv.z, v.w + bv.w);
                    if (col + 3 < N) {
                        if (accumulate & 1) { float4 o = ld4(dst); v = make_float4(v.x + o.x, v.y + o.y, v.z + o.z, v.w + o.w); }
                        if (accumulate & 2) st4_nt(dst, v); else st4(dst, v);      // bit 1: streaming output (not re-read soon)
                    } else {
                        const float e[4] = {v.x, v.y, v.z, v.w};
                        for (int q = 0; q < 4 && col + q < N; ++q) dst[q] = (accumulate & 1) ? dst[q] + e[q] : e[q];
                    }
                }
            }
            __builtin_amdgcn_wave_barrier();
        }
}

// ================================================================================================================
// Weight-stationary kernel for the tall-skinny products of the extractor: C[M,N] = A[M,K] op(B) with M ~ 1e4..1e7 rows and
// a weight B of at most a few hundred rows and columns.  The 128 x 128 tile kernel above reaches ~50 % of the fp32 MFMA peak at
// K = 128 / 256: four K-slabs per tile cannot hide the tile prologue / epilogue, and 1.6 rounds of tiles over the chip leave the
// last round half empty.  Here one persistent 8-wave workgroup per CU keeps ITS SHARE OF B IN REGISTERS for the whole launch
// (wave w owns the 32 columns of column block w % NB and the k-range of split w / NB: (K/KS)/2 floats per lane) and streams
// 32-row tiles of A through a double-buffered LDS image; per tile a wave issues (K/KS)/2 back-to-back MFMAs on one accumulator,
// fed by one conflict-free ds_read_b128 per four MFMAs.  Tiles are dealt round-robin (all cost the same), so the chip is
// balanced to 1/6 of a CU's share at C3.  With KS > 1 the k-splits of a column block are summed through the idle LDS buffer in
// fixed order.  The two k-slots of the 32x32x2 MFMA are mapped to the two HALVES of a wave's k-range (a permutation of the
// summation order only), so every lane reads contiguous k from both operands.
// ================================================================================================================
constexpr int WS_THREADS = 512, WS_ROWS = 32;

template <bool B_T, int KR, int NQ>
__global__ __launch_bounds__(WS_THREADS, (KR <= 64 && NQ <= 4) ? 4 : 2) void k_gemm_ws(
    const float* __restrict__ A, int64_t lda, const float* __restrict__ B, int64_t ldb, float* __restrict__ C, int64_t ldc, int M, int N,
    int K, int NB, const float* __restrict__ bias, int tiles) {
    // KR = (K/KS)/2 floats of B per lane, NQ = K/64 float4 of an A tile per thread.  Small shares (KR <= 64) fit 128 registers:
    // two workgroups per CU run out of phase, so one's staging / barrier / epilogue hides under the other's MFMAs.
    extern __shared__ __attribute__((aligned(16))) float ws_lds[];
    const int LDA = K + 4;                                  // 16-byte slots of consecutive rows fall on distinct bank groups
    float* const buf0 = ws_lds;
    float* const buf1 = ws_lds + WS_ROWS * LDA;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int cb = wave % NB, ks = wave / NB, KS = 8 / NB;
    const int Kw = K / KS;                                  // k-range of this wave, = 2 * KR
    const int c = lane & 31, h = lane >> 5;
    const int n0 = blockIdx.y * (NB * 32) + cb * 32;        // first output column of this wave
    const int kbase = ks * Kw + h * KR;                     // this lane's contiguous k-range [kbase, kbase + KR)
    // ---- this wave's share of B -> registers (once) -----------------------------------------------------
    float b[KR];
    if (B_T) {                                              // B[n][k]: KR contiguous floats per lane
        const float* src = B + (size_t)min(n0 + c, N - 1) * ldb + kbase;
#pragma unroll
        for (int q = 0; q < KR / 4; ++q) {
            const float4 v = ld4(src + 4 * q);
            b[4 * q] = v.x; b[4 * q + 1] = v.y; b[4 * q + 2] = v.z; b[4 * q + 3] = v.w;
        }
    } else {                                                // B[k][n]: a row per k, 32 consecutive columns per half-wave
#pragma unroll
        for (int q = 0; q < KR; ++q) b[q] = B[(size_t)(kbase + q) * ldb + min(n0 + c, N - 1)];
    }
    const float bv = (bias && ks == 0 && n0 + c < N) ? bias[n0 + c] : 0.f;
    // ---- A tile staging: 32 x K floats = NQ float4 per thread; thread-constant offsets, one running row pointer ----
    const int K4 = K >> 2;
    int srow[NQ], soff[NQ];
#pragma unroll
    for (int p = 0; p < NQ; ++p) {
        const int idx = t + p * WS_THREADS;
        srow[p] = idx / K4;
        soff[p] = 4 * (idx % K4);
    }
    float4 st[NQ];
    auto gload = [&](int tile) {
        const int m0 = tile * WS_ROWS;
        // rows past M are clamped to the last row instead of zero-filled (their outputs are never stored): a conditional
        // zero-fill makes the compiler drain every outstanding store of the previous tile before it may overwrite st[]
#pragma unroll
        for (int p = 0; p < NQ; ++p) st[p] = ld4(A + (size_t)min(m0 + srow[p], M - 1) * lda + soff[p]);
    };
    auto lstore = [&](float* dst) {
#pragma unroll
        for (int p = 0; p < NQ; ++p) st4(dst + srow[p] * LDA + soff[p], st[p]);
    };
    int tile = blockIdx.x;
    if (tile < tiles) { gload(tile); lstore(buf0); }
    // B and the bias are in registers NOW: without this the compiler's wait-count pass, which cannot tell the pre-loop loads from
    // the loop's own prefetch across the back edge, puts vmcnt(0) in front of the first MFMA of every tile and exposes the
    // whole latency of the next tile's rows
    __builtin_amdgcn_s_waitcnt(0x0F70);                     // vmcnt(0)
    __syncthreads();
    float* cur = buf0;
    float* nxt = buf1;
    const bool col_ok = n0 + c < N;
    for (; tile < tiles; tile += gridDim.x) {
        const int ntile = tile + gridDim.x;
        if (ntile < tiles) gload(ntile);                    // next tile's rows fly under this tile's MFMAs
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
        const float* arow = cur + c * LDA + kbase;
#pragma unroll
        for (int q = 0; q < KR / 4; ++q) {
            const float4 a4 = ld4(arow + 4 * q);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.x, b[4 * q], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.y, b[4 * q + 1], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.z, b[4 * q + 2], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.w, b[4 * q + 3], acc, 0, 0, 0);
        }
        if (KS > 1) {
            __syncthreads();                                // every wave is done reading `cur`: it becomes the exchange buffer
            if (ks > 0) {
                float* part = cur + ((ks - 1) * NB + cb) * 1024;
#pragma unroll
                for (int r = 0; r < 16; ++r) part[r * 64 + lane] = acc[r];
            }
        }
        if (ntile < tiles) lstore(nxt);
        __syncthreads();
        if (ks == 0) {
            if (KS > 1) {
                for (int s2 = 1; s2 < KS; ++s2) {           // fixed order: bitwise reproducible
                    const float* part = cur + ((s2 - 1) * NB + cb) * 1024;
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[r] += part[r * 64 + lane];
                }
            }
            const int m0 = tile * WS_ROWS;
            float* crow = C + (size_t)(m0 + 4 * h) * ldc + n0 + c;
            if (col_ok) {
                if (m0 + WS_ROWS <= M) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) crow[(size_t)((r & 3) + 8 * (r >> 2)) * ldc] = acc[r] + bv;
                } else {
#pragma unroll
                    for (int r = 0; r < 16; ++r)
                        if (m0 + 4 * h + (r & 3) + 8 * (r >> 2) < M) crow[(size_t)((r & 3) + 8 * (r >> 2)) * ldc] = acc[r] + bv;
                }
            }
        }
        if (KS > 1) __syncthreads();                        // the exchange buffer is read: the next round may overwrite it
        float* tmp = cur; cur = nxt; nxt = tmp;
    }
}

// ================================================================================================================
// The same weight-stationary scheme on the split-bf16 path (hi*hi + hi*lo + lo*hi, 32x32x16 bf16 MFMA) for the BACKWARD-data products of
// the extractor (da1 = dh2 W2, demb = dh1 W1: 51 639 x 256 x 128 and 51 639 x 128 x 256 at C3).  On the 128 x 128 tile kernel these run
// 28-30 us against a ~16 us floor for their 79 MB of operands: four K-slabs per tile, a prologue and an epilogue per tile.  Here a
// persistent workgroup keeps its share of B as bf16 hi / lo fragments in registers (wave w: column block w % NB, k-split w / NB; KSTEPS
// 16-deep steps = 8 KSTEPS registers) and streams 32-row tiles of A: the staging threads split each fp32 value ONCE into the two bf16
// planes of a double-buffered LDS image, every wave then reads its operand fragments with one ds_read_b128 per plane and step.  The MFMA
// time is a quarter of the fp32 kernel's, so the launch is bound by its HBM traffic.  k-splits meet in LDS in fixed order.
// ================================================================================================================
template <bool B_T, int KSTEPS, int NQ>
__global__ __launch_bounds__(WS_THREADS, 2) void k_gemm_ws_x3(const float* __restrict__ A, int64_t lda, const float* __restrict__ B, int64_t ldb,
                                                             float* __restrict__ C, int64_t ldc, int M, int N, int K, int NB, int accumulate, int tiles) {
    extern __shared__ __attribute__((aligned(16))) unsigned char wx_lds[];
    const int RS = 2 * K + 16;                               // bytes per row of one plane
    const int PLANE = WS_ROWS * RS;
    unsigned char* const buf0 = wx_lds;                      // [hi plane | lo plane]
    unsigned char* const buf1 = wx_lds + 2 * PLANE;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int cb = wave % NB, ks = wave / NB, KS = 8 / NB;
    const int c = lane & 31, h = lane >> 5;
    const int n0 = blockIdx.y * (NB * 32) + cb * 32;
    const int kw0 = ks * (KSTEPS * 16);                      // first k of this wave's range
    // ---- this wave's share of B -> hi / lo fragments in registers (once): lane (c, h) holds k = kw0 + 16 s + 8 h + j of column n0 + c ----
    bf16x8 bh[KSTEPS], bl[KSTEPS];
    {
        const int col = min(n0 + c, N - 1);
#pragma unroll
        for (int s_ = 0; s_ < KSTEPS; ++s_) {
            float v[8];
            const int k0 = kw0 + 16 * s_ + 8 * h;
            if (B_T) {
                const float4 v0 = ld4(B + (size_t)col * ldb + k0), v1 = ld4(B + (size_t)col * ldb + k0 + 4);
                v[0] = v0.x; v[1] = v0.y; v[2] = v0.z; v[3] = v0.w; v[4] = v1.x; v[5] = v1.y; v[6] = v1.z; v[7] = v1.w;
            } else {
#pragma unroll
                for (int j = 0; j < 8; ++j) v[j] = B[(size_t)(k0 + j) * ldb + col];
            }
            uint4 H = make_uint4(pack_bf16(v[0], v[1]), pack_bf16(v[2], v[3]), pack_bf16(v[4], v[5]), pack_bf16(v[6], v[7]));
            uint4 L = make_uint4(pack_bf16(v[0] - bf16_hi(v[0]), v[1] - bf16_hi(v[1])), pack_bf16(v[2] - bf16_hi(v[2]), v[3] - bf16_hi(v[3])),
                                 pack_bf16(v[4] - bf16_hi(v[4]), v[5] - bf16_hi(v[5])), pack_bf16(v[6] - bf16_hi(v[6]), v[7] - bf16_hi(v[7])));
            bh[s_] = *reinterpret_cast<bf16x8*>(&H);
            bl[s_] = *reinterpret_cast<bf16x8*>(&L);
        }
    }
    // ---- A tile staging: 32 x K floats = NQ float4 per thread, split into the two planes at the LDS store ----
    const int K4 = K >> 2;
    int srow[NQ], soff[NQ];
#pragma unroll
    for (int p = 0; p < NQ; ++p) {
        const int idx = t + p * WS_THREADS;
        srow[p] = idx / K4;
        soff[p] = 4 * (idx % K4);
    }
    float4 st[NQ];
    auto gload = [&](int tile) {
        const int m0 = tile * WS_ROWS;
#pragma unroll
        for (int p = 0; p < NQ; ++p) st[p] = ld4(A + (size_t)min(m0 + srow[p], M - 1) * lda + soff[p]);       // rows past M: clamped, never stored
    };
    auto lstore = [&](unsigned char* dst) {
#pragma unroll
        for (int p = 0; p < NQ; ++p) {
            const float4 v = st[p];
            const uint2 H = make_uint2(pack_bf16(v.x, v.y), pack_bf16(v.z, v.w));
            const uint2 L = make_uint2(pack_bf16(v.x - bf16_hi(v.x), v.y - bf16_hi(v.y)), pack_bf16(v.z - bf16_hi(v.z), v.w - bf16_hi(v.w)));
            *reinterpret_cast<uint2*>(dst + srow[p] * RS + soff[p] * 2) = H;
            *reinterpret_cast<uint2*>(dst + PLANE + srow[p] * RS + soff[p] * 2) = L;
        }
    };
    int tile = blockIdx.x;
    if (tile < tiles) { gload(tile); lstore(buf0); }
    __builtin_amdgcn_s_waitcnt(0x0F70);                      // vmcnt(0): B is in registers now (see k_gemm_ws)
    __syncthreads();
    unsigned char* cur = buf0;
    unsigned char* nxt = buf1;
    const bool col_ok = n0 + c < N;
    for (; tile < tiles; tile += gridDim.x) {
        const int ntile = tile + gridDim.x;
        if (ntile < tiles) gload(ntile);
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
        const unsigned char* arow = cur + c * RS + (kw0 + 8 * h) * 2;
#pragma unroll
        for (int s_ = 0; s_ < KSTEPS; ++s_) {
            const bf16x8 ah = *reinterpret_cast<const bf16x8*>(arow + 32 * s_);
            const bf16x8 al = *reinterpret_cast<const bf16x8*>(arow + PLANE + 32 * s_);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh[s_], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl[s_], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh[s_], acc, 0, 0, 0);
        }
        float* const xch = reinterpret_cast<float*>(cur);
        if (KS > 1) {
            __syncthreads();                                 // every wave is done reading `cur`: it becomes the exchange buffer
            if (ks > 0) {
                float* part = xch + ((ks - 1) * NB + cb) * 1024;
#pragma unroll
                for (int r = 0; r < 16; ++r) part[r * 64 + lane] = acc[r];
            }
        }
        if (ntile < tiles) lstore(nxt);
        __syncthreads();
        if (ks == 0) {
            if (KS > 1) {
                for (int s2 = 1; s2 < KS; ++s2) {            // fixed order: bitwise reproducible
                    const float* part = xch + ((s2 - 1) * NB + cb) * 1024;
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[r] += part[r * 64 + lane];
                }
            }
            const int m0 = tile * WS_ROWS;
            float* crow = C + (size_t)(m0 + 4 * h) * ldc + n0 + c;
            if (col_ok) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int rr = (r & 3) + 8 * (r >> 2);
                    if (m0 + 4 * h + rr < M) {
                        if (accumulate & 2) __builtin_nontemporal_store(acc[r], crow + (size_t)rr * ldc);
                        else crow[(size_t)rr * ldc] = (accumulate & 1) ? crow[(size_t)rr * ldc] + acc[r] : acc[r];
                    }
                }
            }
        }
        if (KS > 1) __syncthreads();
        unsigned char* tmp = cur; cur = nxt; nxt = tmp;
    }
}

// out[i] = (accumulate ? out[i] : 0) + sum_s slabs[s][i].  Eight independent partial sums (slab s goes to
// partial s % 8) keep eight loads in flight; the order is fixed, so the result is bitwise reproducible.
__global__ void k_slab_reduce(const float* __restrict__ slabs, int nslab, size_t slab_stride, int M, int N, int64_t ldc,
                              int accumulate, float* __restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (int64_t)M * N) return;
    const float* p0 = slabs + i;
    float a[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    int s = 0;
    for (; s + 8 <= nslab; s += 8) {
#pragma unroll
        for (int j = 0; j < 8; ++j) a[j] += p0[(size_t)(s + j) * slab_stride];
    }
    for (int j = 0; s < nslab; ++s, ++j) a[j] += p0[(size_t)s * slab_stride];
    const float acc = ((a[0] + a[1]) + (a[2] + a[3])) + ((a[4] + a[5]) + (a[6] + a[7]));
    const int r = (int)(i / N), c = (int)(i % N);
    float* p = out + (size_t)r * ldc + c;
    *p = accumulate ? *p + acc : acc;
}

// The same sum with four lanes per float4 of the output: lane q of a quad adds slabs q, q+4, q+8, ... (two independent partial sums, so
// the loads pipeline), then the quad combines in a fixed tree.  4x the loads in flight and 16-byte accesses: 10.8 -> ~4 us for the
// 128 x 32768 slabs of the extractor's weight gradients at C3.  Bitwise reproducible (the order never depends on timing).
// P lanes share one float4 of the output (slab s goes to lane s % P): 4 for short sums, 8 from 64 slabs, 16 from 128 -- more threads in
// flight for a kernel that is pure load latency (C3: the 202-slab sums of dW2 and dW1 in one launch, 18.5 us at P = 4, 12.8 at P = 8)
template <int P>
__device__ __forceinline__ void slab_reduce4_body(int64_t t, const float* __restrict__ slabs, int nslab, size_t slab_stride, int M, int N, int64_t ldc,
                                                  int accumulate, float* __restrict__ out);
template <int P>
__global__ void k_slab_reduce4(const float* __restrict__ slabs, int nslab, size_t slab_stride, int M, int N, int64_t ldc,
                               int accumulate, float* __restrict__ out) {
    slab_reduce4_body<P>((int64_t)blockIdx.x * blockDim.x + threadIdx.x, slabs, nslab, slab_stride, M, N, ldc, accumulate, out);
}
// up to four pending sums in one launch: job j owns blocks [first[j], first[j + 1])
struct SlabJobs4 { SlabJob job[4]; int first[5]; };
template <int P>
__global__ void k_slab_reduce_jobs(const SlabJobs4 J) {
    int j = 0;
#pragma unroll
    for (int q = 1; q < 4; ++q) j += (int)blockIdx.x >= J.first[q] ? 1 : 0;
    const SlabJob& b = J.job[j];
    slab_reduce4_body<P>((int64_t)(blockIdx.x - J.first[j]) * blockDim.x + threadIdx.x, b.ws, b.nslab, b.slab, b.M, b.N, b.ldc, b.accumulate, b.out);
}
template <int P>
__device__ __forceinline__ void slab_reduce4_body(int64_t t, const float* __restrict__ slabs, int nslab, size_t slab_stride, int M, int N, int64_t ldc,
                                                  int accumulate, float* __restrict__ out) {
    const int q = (int)(t & (P - 1));
    const int64_t i4 = t / P;                                     // float4 index into the [M, N] output
    const bool live = i4 < (int64_t)M * N / 4;
    float4 a = f4zero(), b = f4zero();
    if (live) {
        const float* p0 = slabs + i4 * 4;
        int s = q;
        for (; s + P < nslab; s += 2 * P) {
            const float4 u = ld4(p0 + (size_t)s * slab_stride), v = ld4(p0 + (size_t)(s + P) * slab_stride);
            a.x += u.x; a.y += u.y; a.z += u.z; a.w += u.w;
            b.x += v.x; b.y += v.y; b.z += v.z; b.w += v.w;
        }
        if (s < nslab) {
            const float4 u = ld4(p0 + (size_t)s * slab_stride);
            a.x += u.x; a.y += u.y; a.z += u.z; a.w += u.w;
        }
    }
    float4 r = make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w);
#pragma unroll
    for (int o = 1; o < P; o <<= 1) {                             // ((q0 + q1) + (q2 + q3)) + ..., the same in every lane of the group
        const float4 w = make_float4(__shfl_xor(r.x, o, 64), __shfl_xor(r.y, o, 64), __shfl_xor(r.z, o, 64), __shfl_xor(r.w, o, 64));
        const bool low = (q & o) == 0;                            // keep (lower, upper) operand order identical in both partners
        r = low ? make_float4(r.x + w.x, r.y + w.y, r.z + w.z, r.w + w.w) : make_float4(w.x + r.x, w.y + r.y, w.z + r.z, w.w + r.w);
    }
    if (live && q == 0) {
        const int64_t e = i4 * 4;
        const int row = (int)(e / N), c = (int)(e % N);
        float* p = out + (size_t)row * ldc + c;
        if (accumulate) { const float4 o = ld4(p); r = make_float4(o.x + r.x, o.y + r.y, o.z + r.z, o.w + r.w); }
        st4(p, r);
    }
}

// Tile choice: (64*TM) x (64*TN).  128x128 for big outputs, 128x64 when that quantises better on 256 CUs, 64x64 for the
// short-K tall-skinny products of the node-mode extractor (more, lighter blocks hide the per-block load latency).
static void gemm_tile(int64_t M, int64_t N, int64_t K, int* tm, int* tn) {
    if (const char* e = getenv("GSAT_GEMM_TILE")) { int v = atoi(e); *tm = v / 10 == 2 ? 2 : 1; *tn = v % 10 == 2 ? 2 : 1; return; }   // tuning override "TMTN"
    *tm = 2;
    if (N <= 64) { *tn = 1; return; }
    const int64_t b2 = ceil_div(M, 128) * ceil_div(N, 128), b1 = ceil_div(M, 128) * ceil_div(N, 64);
    const int64_t cost2 = ceil_div(b2, 256 * 3) * 2, cost1 = ceil_div(b1, 256 * 4) * 1;
    *tn = cost1 < cost2 ? 1 : 2;
    (void)K;
}

// number of K splits used for an M x N output reduced over K rows (shared by the workspace query)
int gemm_splits(int64_t M, int64_t N, int64_t K, bool reduce_rows) {
    if (!reduce_rows) return 1;                       // only weight gradients (K = #rows) are split
    const int64_t tiles = ceil_div(M, 128) * ceil_div(N, 128);
    if (tiles >= 256 || K <= 4 * GK) return 1;
    static const int div_env = getenv("GSAT_GEMM_SPLITK_SLABS") ? atoi(getenv("GSAT_GEMM_SPLITK_SLABS")) : 0;     // min 32-k slabs per split
    static const int target = getenv("GSAT_GEMM_SPLITK_BLOCKS") ? atoi(getenv("GSAT_GEMM_SPLITK_BLOCKS")) : 768;
    // a 64 x 64 weight gradient (GIN at H = 64) is ONE tile: with 8 slabs per split its 12 800 rows made 50 workgroups walking 8 slabs
    // each (26 us for 0.1 GFLOP); 2 slabs per split give 200 short workgroups on 64 x 64 tiles
    const int div = div_env > 0 ? div_env : (M <= 64 && N <= 64 ? 4 : 8);      // (4, not 2: the slab sum of 200 partials cost more than the shorter GEMM saved: C2 whole step 0.995 -> 0.97 ms)
    int64_t s = std::min<int64_t>(ceil_div(target, tiles), ceil_div(K, (int64_t)div * GK));
    return (int)std::max<int64_t>(1, std::min<int64_t>(s, 512));
}

size_t gemm_workspace_floats(int64_t M, int64_t N, int64_t K, bool reduce_rows) {
    const int s = gemm_splits(M, N, K, reduce_rows);
    return s > 1 ? (size_t)s * M * N : 0;
}

// Precision policy.  The attention extractor always asks for exact fp32 (`allow_split` = false): its GEMM outputs feed
// per-graph InstanceNorms whose 1/sigma amplifies a 1e-5 perturbation by up to ~3e2 per layer in the backward (the C4
// baseline-size parity test fails with split-bf16 there).  Callers without such an amplifier (the backbone's Linear layers)
// may allow the split-bf16 kernel for large products.  GSAT_GEMM_PRECISION=fp32|bf16x3 overrides everything (tuning).
static void xcd_switch_once() {
    static const bool done = [] {
        if (const char* e = getenv("GSAT_GEMM_XCD")) { int v = atoi(e); (void)hipMemcpyToSymbol(HIP_SYMBOL(GEMM_XCD_REMAP_ON), &v, sizeof(int)); }
        return true;
    }();
    (void)done;
}

static bool use_bf16x3(int64_t M, int64_t N, int64_t K, bool allow_split) {
    if (const char* e = getenv("GSAT_GEMM_PRECISION")) return e[0] == 'b';
    return allow_split && 2.0 * (double)M * (double)N * (double)K >= 2e9 && K >= 64;
}

// ---- weight-stationary dispatch ------------------------------------------------------------------------------------
static bool ws_geometry(int64_t N, int64_t K, int* nb, int* kr) {
    const int64_t ncols = N > 256 ? 256 : N;              // wider outputs run as 256-column chunks (gridDim.y)
    if (N % 32 != 0 || (N > 256 && N % 256 != 0) || (K != 64 && K != 128 && K != 256 && K != 512)) return false;
    const int64_t NB = ncols / 32;
    if (NB != 1 && NB != 2 && NB != 4 && NB != 8) return false;
    const int64_t KS = 8 / NB, Kw = K / KS;
    if (K % KS != 0 || Kw % 8 != 0) return false;
    const int64_t KR = Kw / 2;
    if (KR != 16 && KR != 32 && KR != 64 && KR != 128) return false;
    if ((KS - 1) * NB * 1024 > WS_ROWS * (K + 4)) return false;      // the k-split exchange must fit one A buffer
    *nb = (int)NB; *kr = (int)KR;
    return true;
}
static bool ws_applicable(int64_t M, int64_t N, int64_t K) {
    static const int on = getenv("GSAT_GEMM_WS") ? atoi(getenv("GSAT_GEMM_WS")) : 1;
    int nb, kr;
    return on && M >= 8192 && M < (1ll << 31) - 64 && ws_geometry(N, K, &nb, &kr);
}
template <bool B_T, int KR, int NQ>
static int ws_launch(hipStream_t stream, dim3 grid, size_t lds, const float* A, int64_t lda, const float* B, int64_t ldb, float* C, int64_t ldc,
                     int M, int N, int K, int nb, const float* bias, int tiles) {
    static size_t allowed = 64 * 1024;
    if (lds > allowed) {
        GSAT_CHECK_HIP(hipFuncSetAttribute((const void*)k_gemm_ws<B_T, KR, NQ>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        allowed = lds;
    }
    k_gemm_ws<B_T, KR, NQ><<<grid, WS_THREADS, lds, stream>>>(A, lda, B, ldb, C, ldc, M, N, K, nb, bias, tiles);
    GSAT_LAUNCH_CHECK();
    return GSAT_OK;
}
static int gemm_ws(hipStream_t stream, bool b_t, int64_t M, int64_t N, int64_t K, const float* A, int64_t lda, const float* B, int64_t ldb,
                   float* C, int64_t ldc, const float* bias) {
    int nb = 0, kr = 0;
    GSAT_REQUIRE(ws_geometry(N, K, &nb, &kr), GSAT_ERR_UNSUPPORTED, "gemm_ws: unsupported shape");
    GSAT_REQUIRE(lda % 4 == 0 && ((uintptr_t)A % 16 == 0) && ((uintptr_t)B % 16 == 0) && (!b_t || ldb % 4 == 0), GSAT_ERR_ARG, "gemm_ws: alignment");
    const int tiles = (int)ceil_div(M, WS_ROWS);
    const int chunks = (int)ceil_div(N, 256);
    const size_t lds = (size_t)2 * WS_ROWS * (K + 4) * sizeof(float);
    const int nq = (int)(K / 64);
    // persistent workgroups: two per CU where registers and LDS allow it (they run out of phase), tiles dealt round-robin
    const int per_cu = (kr <= 64 && nq <= 4 && 2 * lds <= 150 * 1024) ? 2 : 1;
    const dim3 grid((unsigned)std::min<int64_t>(tiles, (int64_t)256 * per_cu / std::min(chunks, per_cu)), (unsigned)chunks);
#define WSGO(BT, R, Q) return ws_launch<BT, R, Q>(stream, grid, lds, A, lda, B, ldb, C, ldc, (int)M, (int)N, (int)K, nb, bias, tiles)
#define WSQ(BT, R) do { switch (nq) { case 1: WSGO(BT, R, 1); case 2: WSGO(BT, R, 2); case 4: WSGO(BT, R, 4); case 8: WSGO(BT, R, 8); default: break; } } while (0)
#define WSKR(BT) do { switch (kr) { case 16: WSQ(BT, 16); break; case 32: WSQ(BT, 32); break; case 64: WSQ(BT, 64); break; default: WSQ(BT, 128); break; } } while (0)
    if (b_t) WSKR(true); else WSKR(false);
#undef WSKR
#undef WSQ
#undef WSGO
    GSAT_REQUIRE(false, GSAT_ERR_UNSUPPORTED, "gemm_ws: K = %lld is not 64, 128, 256 or 512", (long long)K);
}

// ---- weight-stationary split-bf16 dispatch ------------------------------------------------------------------------
static bool wsx3_geometry(int64_t N, int64_t K, int* nb, int* ksteps) {
    // wider outputs run as 256-column chunks (gridDim.y): every chunk streams all rows of A (the re-reads stay in the Infinity Cache for the
    // backbone's dx = dy W: 51 639 x 1024 x 128 at C3, 85 us on the tile kernel for a 211 MB output, 62 us here)
    if (N % 32 != 0 || (N > 256 && N % 256 != 0) || (K != 64 && K != 128 && K != 256 && K != 512)) return false;
    const int64_t NB = (N > 256 ? 256 : N) / 32;
    if (NB != 1 && NB != 2 && NB != 4 && NB != 8) return false;
    const int64_t KS = 8 / NB, Kw = K / KS;
    if (Kw % 16 != 0) return false;
    const int64_t st = Kw / 16;
    if (st != 2 && st != 4 && st != 8) return false;                          // 16 / 32 / 64 fragment registers
    if ((KS - 1) * NB * 4096 > 2 * WS_ROWS * (2 * K + 16)) return false;      // the k-split exchange must fit one A buffer
    *nb = (int)NB; *ksteps = (int)st;
    return true;
}
static bool wsx3_applicable(int64_t M, int64_t N, int64_t K) {
    static const int on = getenv("GSAT_GEMM_WSX3") ? atoi(getenv("GSAT_GEMM_WSX3")) : 1;
    int nb, st;
    return on && M >= 8192 && M < (1ll << 31) - 64 && wsx3_geometry(N, K, &nb, &st);
}
template <bool B_T, int KSTEPS, int NQ>
static int wsx3_launch(hipStream_t stream, dim3 grid, size_t lds, const float* A, int64_t lda, const float* B, int64_t ldb, float* C, int64_t ldc,
                       int M, int N, int K, int nb, int accumulate, int tiles) {
    static size_t allowed = 64 * 1024;
    if (lds > allowed) {
        GSAT_CHECK_HIP(hipFuncSetAttribute((const void*)k_gemm_ws_x3<B_T, KSTEPS, NQ>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        allowed = lds;
    }
    k_gemm_ws_x3<B_T, KSTEPS, NQ><<<grid, WS_THREADS, lds, stream>>>(A, lda, B, ldb, C, ldc, M, N, K, nb, accumulate, tiles);
    GSAT_LAUNCH_CHECK();
    return GSAT_OK;
}
static int gemm_wsx3(hipStream_t stream, bool b_t, int64_t M, int64_t N, int64_t K, const float* A, int64_t lda, const float* B, int64_t ldb,
                     float* C, int64_t ldc, bool accumulate) {
    int nb = 0, st = 0;
    GSAT_REQUIRE(wsx3_geometry(N, K, &nb, &st), GSAT_ERR_UNSUPPORTED, "gemm_wsx3: unsupported shape");
    const int tiles = (int)ceil_div(M, WS_ROWS);
    const size_t lds = (size_t)4 * WS_ROWS * (2 * K + 16);
    const int nq = (int)(K / 64);
    const int per_cu = (2 * lds <= 150 * 1024 && st * 8 + nq * 9 <= 84) ? 2 : 1;      // two workgroups per CU need <= 128 registers (fragments + staging)
    const int chunks = (int)ceil_div(N, 256);
    const dim3 grid((unsigned)std::min<int64_t>(tiles, std::max<int64_t>(1, (int64_t)256 * per_cu / chunks)), (unsigned)chunks);
    // bit 1: streaming output (too large for the caches to keep until its consumer runs)
    static const int nt_mode = getenv("GSAT_GEMM_NT") ? atoi(getenv("GSAT_GEMM_NT")) : 1;
    const int acc_flag = (accumulate ? 1 : 0) | ((nt_mode && !accumulate && (size_t)M * N * 4 >= ((size_t)64 << 20)) ? 2 : 0);
#define XGO(BT, S, Q) return wsx3_launch<BT, S, Q>(stream, grid, lds, A, lda, B, ldb, C, ldc, (int)M, (int)N, (int)K, nb, acc_flag, tiles)
#define XQ(BT, S) do { switch (nq) { case 1: XGO(BT, S, 1); case 2: XGO(BT, S, 2); case 4: XGO(BT, S, 4); case 8: XGO(BT, S, 8); default: break; } } while (0)
#define XS(BT) do { switch (st) { case 2: XQ(BT, 2); break; case 4: XQ(BT, 4); break; default: XQ(BT, 8); break; } } while (0)
    if (b_t) XS(true); else XS(false);
#undef XS
#undef XQ
#undef XGO
    GSAT_REQUIRE(false, GSAT_ERR_UNSUPPORTED, "gemm_wsx3: K = %lld is not 64, 128, 256 or 512", (long long)K);
}

// C[M,N] (+)= op(A) op(B) (+ bias).  a_t: A given as [K,M]; b_t: B given as [N,K].  K % 4 == 0 and the
// contiguous extents must be multiples of 4 (float4 staging).  `ws` is needed when gemm_splits() > 1.
int gemm_f32(hipStream_t stream, bool a_t, bool b_t, int64_t M, int64_t N, int64_t K, const float* A, int64_t lda, const float* B,
             int64_t ldb, float* C, int64_t ldc, const float* bias, bool accumulate, float* ws, size_t ws_floats, bool allow_split, SlabJob* defer) {
    if (defer) defer->nslab = 0;
    if (M <= 0 || N <= 0) return GSAT_OK;
    xcd_switch_once();
    GSAT_REQUIRE(K > 0 && A && B && C, GSAT_ERR_ARG, "gemm_f32: bad argument");
    GSAT_REQUIRE(lda % 4 == 0 && ldb % 4 == 0 && (a_t ? M % 4 == 0 : K % 4 == 0) && (b_t ? K % 4 == 0 : N % 4 == 0), GSAT_ERR_UNSUPPORTED,
                 "gemm_f32: contiguous extents and leading dimensions must be multiples of 4 (M=%lld N=%lld K=%lld)", (long long)M, (long long)N, (long long)K);
    GSAT_REQUIRE(((uintptr_t)A % 16 == 0) && ((uintptr_t)B % 16 == 0) && ((uintptr_t)C % 16 == 0) && ldc % 4 == 0 && (!bias || (uintptr_t)bias % 16 == 0),
                 GSAT_ERR_ARG, "gemm_f32: operands, output and bias must be 16-byte aligned with ldc % 4 == 0");
    const bool split = use_bf16x3(M, N, K, allow_split);
    if (!a_t && !split && !accumulate && ws_applicable(M, N, K)) return gemm_ws(stream, b_t, M, N, K, A, lda, B, ldb, C, ldc, bias);
    if (!a_t && split && !bias && wsx3_applicable(M, N, K) && (b_t ? ldb % 4 == 0 : true)) return gemm_wsx3(stream, b_t, M, N, K, A, lda, B, ldb, C, ldc, accumulate);
    const int splits = gemm_splits(M, N, K, a_t);
    int tm = 2, tn = 2;
    if (splits == 1) gemm_tile(M, N, K, &tm, &tn);
    else if (!getenv("GSAT_GEMM_TILE")) { tm = M <= 64 ? 1 : 2; tn = N <= 64 ? 1 : 2; }       // split-K: do not pad a small output to 128 x 128
    {   // split-K on the exact-fp32 kernel with fewer workgroups than 1.5 per CU (a 128 x 128 weight gradient over ~3e4 rows: ONE tile x 123
        // splits): 64 x 64 tiles give four times the workgroups (C4: 26.9 -> ~15 us per product).  GSAT_GEMM_WGRAD_SMALL_TILES=0 switches it off
        static const int small_tiles = getenv("GSAT_GEMM_WGRAD_SMALL_TILES") ? atoi(getenv("GSAT_GEMM_WGRAD_SMALL_TILES")) : 1;
        if (small_tiles && splits > 1 && !split && ceil_div(M, 64 * tm) * ceil_div(N, 64 * tn) * splits < 384) tm = tn = 1;
    }
    // split-bf16: the per-tile staging (fp32 -> hi/lo, LDS planes) is what costs, so the largest tile wins even when it leaves
    // fewer workgroups than CUs x occupancy (51 639 x 128 x 1024: 57 us at 128x128 against 69 us at 128x64)
    if (split && splits == 1 && !getenv("GSAT_GEMM_TILE")) { tm = 2; tn = N > 64 ? 2 : 1; }
    dim3 grid((unsigned)ceil_div(N, 64 * tn), (unsigned)ceil_div(M, 64 * tm), (unsigned)splits);
    int kps = (int)(ceil_div(ceil_div(K, splits), GK) * GK);
    float* out = C;
    int64_t ldo = ldc;
    size_t slab = 0;
    // outputs too large for the L2s / Infinity Cache to keep until their consumer runs are written with non-temporal stores
    static const int nt_mode = getenv("GSAT_GEMM_NT") ? atoi(getenv("GSAT_GEMM_NT")) : 1;
    int acc_flag = (accumulate ? 1 : 0) | ((nt_mode && !accumulate && (size_t)M * N * 4 >= ((size_t)64 << 20)) ? 2 : 0);
    const float* bptr = bias;
    if (splits > 1) {
        GSAT_REQUIRE(ws && ws_floats >= (size_t)splits * M * N, GSAT_ERR_WORKSPACE, "gemm_f32: split-K workspace too small");
        GSAT_REQUIRE(!bias, GSAT_ERR_UNSUPPORTED, "gemm_f32: bias with split-K");
        out = ws; ldo = N; slab = (size_t)M * N; acc_flag = 0;
    }
#define LAUNCH(AT, BT)                                                                                                              \
    do {                                                                                                                            \
        if (split) {                                                                                                                \
            if (tm == 2 && tn == 2) k_gemm_bf16x3<AT, BT, 2, 2><<<grid, GT, 0, stream>>>(A, lda, B, ldb, out, ldo, (int)M, (int)N, (int)K, kps, bptr, acc_flag, slab); \
            else if (tm == 2) k_gemm_bf16x3<AT, BT, 2, 1><<<grid, GT, 0, stream>>>(A, lda, B, ldb, out, ldo, (int)M, (int)N, (int)K, kps, bptr, acc_flag, slab);       \
            else if (tn == 2) k_gemm_bf16x3<AT, BT, 1, 2><<<grid, GT, 0, stream>>>(A, lda, B, ldb, out, ldo, (int)M, (int)N, (int)K, kps, bptr, acc_flag, slab);       \
            else k_gemm_bf16x3<AT, BT, 1, 1><<<grid, GT, 0, stream>>>(A, lda, B, ldb, out, ldo, (int)M, (int)N, (int)K, kps, bptr, acc_flag, slab);                    \
        } else if (tm == 2 && tn == 2) k_gemm_f32<AT, BT, 2, 2><<<grid, GT, 0, stream>>>(A, lda, B, ldb, out, ldo, (int)M, (int)N, (int)K, kps, bptr, acc_flag, slab); \
        else if (tm == 2) k_gemm_f32<AT, BT, 2, 1><<<grid, GT, 0, stream>>>(A, lda, B, ldb, out, ldo, (int)M, (int)N, (int)K, kps, bptr, acc_flag, slab);       \
        else if (tn == 2) k_gemm_f32<AT, BT, 1, 2><<<grid, GT, 0, stream>>>(A, lda, B, ldb, out, ldo, (int)M, (int)N, (int)K, kps, bptr, acc_flag, slab);       \
        else k_gemm_f32<AT, BT, 1, 1><<<grid, GT, 0, stream>>>(A, lda, B, ldb, out, ldo, (int)M, (int)N, (int)K, kps, bptr, acc_flag, slab);                    \
    } while (0)
    if (a_t) { if (b_t) LAUNCH(true, true); else LAUNCH(true, false); }
    else { if (b_t) LAUNCH(false, true); else LAUNCH(false, false); }
#undef LAUNCH
    GSAT_LAUNCH_CHECK();
    if (splits > 1) {
        const bool vec = N % 4 == 0 && ldc % 4 == 0 && slab % 4 == 0 && ((uintptr_t)C & 15) == 0 && ((uintptr_t)ws & 15) == 0;
        if (defer && vec) { *defer = SlabJob{ws, splits, slab, (int)M, (int)N, ldc, accumulate ? 1 : 0, C}; return GSAT_OK; }
        if (vec && splits >= 128)
            k_slab_reduce4<16><<<(unsigned)ceil_div(M * N * 4, 256), 256, 0, stream>>>(ws, splits, slab, (int)M, (int)N, ldc, accumulate ? 1 : 0, C);
        else if (vec && splits >= 64)
            k_slab_reduce4<8><<<(unsigned)ceil_div(M * N * 2, 256), 256, 0, stream>>>(ws, splits, slab, (int)M, (int)N, ldc, accumulate ? 1 : 0, C);
        else if (vec)
            k_slab_reduce4<4><<<(unsigned)ceil_div(M * N, 256), 256, 0, stream>>>(ws, splits, slab, (int)M, (int)N, ldc, accumulate ? 1 : 0, C);
        else
            k_slab_reduce<<<(unsigned)ceil_div(M * N, 256), 256, 0, stream>>>(ws, splits, slab, (int)M, (int)N, ldc, accumulate ? 1 : 0, C);
        GSAT_LAUNCH_CHECK();
    }
    return GSAT_OK;
}

int slab_reduce_jobs(hipStream_t stream, const SlabJob* jobs, int n) {
    SlabJobs4 J{};
    int m = 0, blocks = 0, wide = 0;
    for (int i = 0; i < n; ++i) wide = std::max(wide, jobs[i].nslab >= 128 ? 4 : (jobs[i].nslab >= 64 ? 2 : 0));
    for (int i = 0; i < n; ++i) {
        if (jobs[i].nslab <= 0) continue;
        GSAT_REQUIRE(m < 4, GSAT_ERR_ARG, "slab_reduce_jobs: more than four pending sums");
        J.job[m] = jobs[i];
        J.first[m] = blocks;
        blocks += (int)ceil_div((int64_t)jobs[i].M * jobs[i].N * (wide ? wide : 1), 256);
        ++m;
    }
    for (int q = m; q <= 4; ++q) J.first[q] = blocks;
    if (!blocks) return GSAT_OK;
    if (wide == 4) k_slab_reduce_jobs<16><<<blocks, 256, 0, stream>>>(J);
    else if (wide == 2) k_slab_reduce_jobs<8><<<blocks, 256, 0, stream>>>(J);
    else k_slab_reduce_jobs<4><<<blocks, 256, 0, stream>>>(J);
    GSAT_LAUNCH_CHECK();
    return GSAT_OK;
}

// post_nn of PNAConvSimple on the virtual aggregate: out [N, Ho] = Agg W^T + b (forward) and dW [Ho, NAGG*2*H] = dOut^T Agg (split over
// the rows, slabs summed in order), both on the split-bf16 tile kernel with the operand loader that recomputes the x_i columns.
int pna_post_fwd(hipStream_t stream, const PnaVirt& pv, int64_t Nrows, const float* W, int64_t ldw, const float* bias, int64_t Ho, float* out, int64_t ldo) {
    const int64_t K = (int64_t)pv.NAGG * 2 * pv.H;
    GSAT_REQUIRE(pv.H % 32 == 0 && Ho % 4 == 0 && ldw % 4 == 0 && ldo % 4 == 0 && Nrows < (1ll << 31), GSAT_ERR_UNSUPPORTED, "gsat_pna_post_fwd: H must be a multiple of 32, Ho of 4");
    GSAT_REQUIRE(((uintptr_t)W % 16 == 0) && ((uintptr_t)out % 16 == 0) && (!bias || (uintptr_t)bias % 16 == 0) && ((uintptr_t)pv.x % 16 == 0) && ((uintptr_t)pv.aggj % 16 == 0),
                 GSAT_ERR_ARG, "gsat_pna_post_fwd: operands must be 16-byte aligned");
    if (Nrows <= 0 || Ho <= 0) return GSAT_OK;
    xcd_switch_once();
    const int tn = Ho > 64 ? 2 : 1;
    dim3 grid((unsigned)ceil_div(Ho, 64 * tn), (unsigned)ceil_div(Nrows, 128), 1);
    if (tn == 2) k_gemm_bf16x3<false, true, 2, 2, 1><<<grid, GT, 0, stream>>>(nullptr, 0, W, ldw, out, ldo, (int)Nrows, (int)Ho, (int)K, (int)K, bias, 0, 0, pv);
    else k_gemm_bf16x3<false, true, 2, 1, 1><<<grid, GT, 0, stream>>>(nullptr, 0, W, ldw, out, ldo, (int)Nrows, (int)Ho, (int)K, (int)K, bias, 0, 0, pv);
    GSAT_LAUNCH_CHECK();
    return GSAT_OK;
}

int pna_post_dw(hipStream_t stream, const PnaVirt& pv, int64_t Nrows, const float* dout, int64_t ldd, int64_t Ho, float* dW, int64_t lddw, float* ws,
                size_t ws_floats) {
    const int64_t Kc = (int64_t)pv.NAGG * 2 * pv.H;                    // columns of dW
    GSAT_REQUIRE(pv.H % 64 == 0 && Ho % 4 == 0 && ldd % 4 == 0 && lddw % 4 == 0 && Nrows < (1ll << 31), GSAT_ERR_UNSUPPORTED, "gsat_pna_post_dw: H must be a multiple of 64, Ho of 4");
    GSAT_REQUIRE(((uintptr_t)dout % 16 == 0) && ((uintptr_t)dW % 16 == 0) && ((uintptr_t)pv.x % 16 == 0) && ((uintptr_t)pv.aggj % 16 == 0), GSAT_ERR_ARG,
                 "gsat_pna_post_dw: operands must be 16-byte aligned");
    if (Ho <= 0) return GSAT_OK;
    xcd_switch_once();
    const int splits = gemm_splits(Ho, Kc, Nrows, true);
    const int tm = Ho <= 64 ? 1 : 2, tn = pv.H % 128 == 0 ? 2 : 1;     // a column tile must stay inside one segment of the aggregate
    dim3 grid((unsigned)ceil_div(Kc, 64 * tn), (unsigned)ceil_div(Ho, 64 * tm), (unsigned)splits);
    const int kps = (int)(ceil_div(ceil_div(Nrows, splits), GK) * GK);
    float* o = dW;
    int64_t ldo = lddw;
    size_t slab = 0;
    if (splits > 1) {
        GSAT_REQUIRE(ws && ws_floats >= (size_t)splits * Ho * Kc, GSAT_ERR_WORKSPACE, "gsat_pna_post_dw: split-K workspace too small");
        o = ws; ldo = Kc; slab = (size_t)Ho * Kc;
    }
#define GO(TM_, TN_) k_gemm_bf16x3<true, false, TM_, TN_, 2><<<grid, GT, 0, stream>>>(dout, ldd, nullptr, 0, o, ldo, (int)Ho, (int)Kc, (int)Nrows, kps, nullptr, 0, slab, pv)
    if (tm == 2 && tn == 2) GO(2, 2); else if (tm == 2) GO(2, 1); else if (tn == 2) GO(1, 2); else GO(1, 1);
#undef GO
    GSAT_LAUNCH_CHECK();
    if (splits > 1) {
        k_slab_reduce4<4><<<(unsigned)ceil_div(Ho * Kc, 256), 256, 0, stream>>>(ws, splits, slab, (int)Ho, (int)Kc, lddw, 0, dW);
        GSAT_LAUNCH_CHECK();
    }
    return GSAT_OK;
}

}  // namespace gsat

using namespace gsat;

extern "C" {

static int pna_virt_make(const char* who, const float* x, const float* aggj, const float* scal, int64_t H, int nagg, PnaVirt* pv) {
    GSAT_REQUIRE(x && aggj && scal && (nagg == 4 || nagg == 5) && H > 0 && ((uintptr_t)scal % 16 == 0), GSAT_ERR_ARG, "%s: bad aggregate operand", who);
    pv->x = x; pv->aggj = aggj; pv->scal = scal; pv->H = (int)H; pv->NAGG = nagg;
    return GSAT_OK;
}

int gsat_pna_post_fwd(const float* x, const float* aggj, const float* scal, int64_t N, int64_t H, int num_aggregators,
                      const float* W, int64_t ldw, const float* bias, int64_t Ho, float* out, void* stream) {
    PnaVirt pv;
    int rc = pna_virt_make("gsat_pna_post_fwd", x, aggj, scal, H, num_aggregators, &pv);
    if (rc) return rc;
    GSAT_REQUIRE(W && out, GSAT_ERR_ARG, "gsat_pna_post_fwd: null pointer");
    return pna_post_fwd((hipStream_t)stream, pv, N, W, ldw, bias, Ho, out, Ho);
}

size_t gsat_pna_post_dw_workspace_floats(int64_t N, int64_t H, int num_aggregators, int64_t Ho) {
    return gemm_workspace_floats(Ho, (int64_t)num_aggregators * 2 * H, N, true);
}

int gsat_pna_post_dw(const float* x, const float* aggj, const float* scal, int64_t N, int64_t H, int num_aggregators,
                     const float* dout, int64_t Ho, float* dW, float* workspace, size_t workspace_floats, void* stream) {
    PnaVirt pv;
    int rc = pna_virt_make("gsat_pna_post_dw", x, aggj, scal, H, num_aggregators, &pv);
    if (rc) return rc;
    GSAT_REQUIRE(dout && dW, GSAT_ERR_ARG, "gsat_pna_post_dw: null pointer");
    return pna_post_dw((hipStream_t)stream, pv, N, dout, Ho, Ho, dW, (int64_t)num_aggregators * 2 * H, workspace, workspace_floats);
}

/* C[M,N] = (accumulate ? C : 0) + op(A) op(B) + bias ; see include/gsat_hip.h */
int gsat_gemm_f32(int a_t, int b_t, int64_t M, int64_t N, int64_t K, const float* A, int64_t lda, const float* B, int64_t ldb,
                  float* C, int64_t ldc, const float* bias, int accumulate, float* workspace, size_t workspace_floats, void* stream) {
    return gemm_f32((hipStream_t)stream, a_t != 0, b_t != 0, M, N, K, A, lda, B, ldb, C, ldc, bias, accumulate != 0, workspace, workspace_floats, false);
}

/* same product on the split-bf16 (hi*hi + hi*lo + lo*hi) MFMA path when it is large enough to pay; see include/gsat_hip.h */
int gsat_gemm_bf16x3(int a_t, int b_t, int64_t M, int64_t N, int64_t K, const float* A, int64_t lda, const float* B, int64_t ldb,
                     float* C, int64_t ldc, const float* bias, int accumulate, float* workspace, size_t workspace_floats, void* stream) {
    return gemm_f32((hipStream_t)stream, a_t != 0, b_t != 0, M, N, K, A, lda, B, ldb, C, ldc, bias, accumulate != 0, workspace, workspace_floats, true);
}

size_t gsat_gemm_workspace_floats(int a_t, int64_t M, int64_t N, int64_t K) { return gemm_workspace_floats(M, N, K, a_t != 0); }

}  // extern "C"
