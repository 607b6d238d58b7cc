// Categorical encoders (ogb AtomEncoder / BondEncoder, src/models/gin.py:22-25, src/models/pna.py:19-22):
//   out[n,:] = sum_col  W[offset[col] + x[n,col], :]
// Forward is a gather of `ncol` tiny-table rows per node (tables are L2-resident).  Backward is NOT a scatter-add:
// the tables have only a few hundred rows, so every row collects thousands of contributions (atomics would
// serialise, torch's embedding backward sorts the indices on every call -- 35 % of the reference-shaped training
// step in profiles/r01_c3_fullstep_*).  Instead dW = O^T dout with the one-hot matrix O [N, R] built once per batch,
// i.e. one split-K MFMA GEMM (gemm.hip), deterministic.
#include "common.h"

namespace gsat {

constexpr int MAXCOL = 16;
struct ColOffsets { int ncol; int off[MAXCOL]; int dim[MAXCOL]; };

template <int LPR>
__global__ __launch_bounds__(256) void k_embsum_fwd(const int64_t* __restrict__ x, ColOffsets co, const float* __restrict__ W,
                                                    int64_t N, int H, float* __restrict__ out) {
    const int lane = threadIdx.x % LPR;
    for (int64_t n = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) / LPR; n < N; n += (int64_t)gridDim.x * blockDim.x / LPR) {
        int rows[MAXCOL];
#pragma unroll
        for (int i = 0; i < MAXCOL; ++i) {
            if (i < co.ncol) {
                int64_t v = x[n * co.ncol + i];
                v = v < 0 ? 0 : (v >= co.dim[i] ? co.dim[i] - 1 : v);      // clamp: never read outside the table
                rows[i] = co.off[i] + (int)v;
            }
        }
        for (int c = lane * 4; c < H; c += LPR * 4) {
            float4 acc = f4zero();
#pragma unroll
            for (int i = 0; i < MAXCOL; ++i) {
                if (i < co.ncol) {
                    float4 w = ld4(W + (size_t)rows[i] * H + c);
                    acc.x += w.x; acc.y += w.y; acc.z += w.z; acc.w += w.w;
                }
            }
            st4(out + n * H + c, acc);
        }
    }
}

__global__ void k_onehot(const int64_t* __restrict__ x, ColOffsets co, int64_t N, int R, float* __restrict__ O) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N * co.ncol) return;
    const int64_t n = i / co.ncol;
    const int col = (int)(i % co.ncol);
    int64_t v = x[i];
    v = v < 0 ? 0 : (v >= co.dim[col] ? co.dim[col] - 1 : v);
    O[n * R + co.off[col] + (int)v] = 1.f;
}

static int make_offsets(const int32_t* dims, int ncol, ColOffsets* co, int* total) {
    GSAT_REQUIRE(dims && ncol >= 1 && ncol <= MAXCOL, GSAT_ERR_ARG, "categorical encoder: 1..%d columns supported", MAXCOL);
    co->ncol = ncol;
    int off = 0;
    for (int i = 0; i < MAXCOL; ++i) { co->off[i] = 0; co->dim[i] = 1; }
    for (int i = 0; i < ncol; ++i) {
        GSAT_REQUIRE(dims[i] >= 1, GSAT_ERR_ARG, "categorical encoder: empty table");
        co->off[i] = off; co->dim[i] = dims[i]; off += dims[i];
    }
    *total = off;
    return GSAT_OK;
}

}  // namespace gsat

using namespace gsat;

extern "C" {

int gsat_embsum_fwd(const int64_t* x, const int32_t* dims, int ncol, const float* W, int64_t N, int64_t H, float* out, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    ColOffsets co; int R;
    int rc = make_offsets(dims, ncol, &co, &R);
    if (rc) return rc;
    GSAT_REQUIRE(N >= 0 && H > 0 && H % 4 == 0, GSAT_ERR_UNSUPPORTED, "gsat_embsum_fwd: H must be a positive multiple of 4");
    if (N == 0) return GSAT_OK;
    GSAT_REQUIRE(x && W && out, GSAT_ERR_ARG, "gsat_embsum_fwd: null pointer");
    const int q = (int)(H / 4);
    const int lpr = q <= 4 ? 4 : q <= 8 ? 8 : q <= 16 ? 16 : q <= 32 ? 32 : 64;
    const int nb = (int)std::min<int64_t>(ceil_div(N, 256 / lpr), 256 * 16);
#define CALL(L) k_embsum_fwd<L><<<nb, 256, 0, stream>>>(x, co, W, N, (int)H, out)
    switch (lpr) { case 4: CALL(4); break; case 8: CALL(8); break; case 16: CALL(16); break; case 32: CALL(32); break; default: CALL(64); break; }
#undef CALL
    GSAT_LAUNCH_CHECK();
    return GSAT_OK;
}

int gsat_onehot_rows(const int64_t* x, const int32_t* dims, int ncol, int64_t N, int64_t R_padded, float* O, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    ColOffsets co; int R;
    int rc = make_offsets(dims, ncol, &co, &R);
    if (rc) return rc;
    GSAT_REQUIRE(N >= 0 && R_padded >= R && R_padded % 4 == 0, GSAT_ERR_ARG, "gsat_onehot_rows: R_padded must be a multiple of 4 and >= %d", R);
    if (N == 0) return GSAT_OK;
    GSAT_REQUIRE(x && O, GSAT_ERR_ARG, "gsat_onehot_rows: null pointer");
    GSAT_CHECK_HIP(gsat::zero_async(O, sizeof(float) * (size_t)N * R_padded, stream));
    k_onehot<<<(unsigned)ceil_div(N * ncol, 256), 256, 0, stream>>>(x, co, N, (int)R_padded, O);
    GSAT_LAUNCH_CHECK();
    return GSAT_OK;
}

}  // extern "C"
