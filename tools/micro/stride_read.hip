// Micro-benchmark: how fast can 256-thread blocks stream a [R, K] fp32 matrix when each block owns a 128-row tile and walks K
// in slabs of C floats per row (the access pattern of a GEMM's k-contiguous operand), one slab prefetched ahead?
// build: hipcc --offload-arch=gfx950 -O3 tools/micro/stride_read.hip -o gpurun_out/stride_read
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int C>      // floats per row per slab: 32 (128 B), 64, 128, 256
__global__ __launch_bounds__(256) void k_walk(const float* __restrict__ X, int R, int K, float* __restrict__ out) {
    constexpr int QPR = C / 4;              // float4 per row per slab
    constexpr int RPP = 256 / QPR;          // rows per pass
    constexpr int P = 128 / RPP;            // passes per slab
    const int t = threadIdx.x, q = t % QPR, r0 = blockIdx.x * 128 + t / QPR;
    float4 cur[P], nxt[P];
    float acc = 0.f;
    auto load = [&](float4* v, int k0) {
#pragma unroll
        for (int p = 0; p < P; ++p) {
            const int r = r0 + p * RPP;
            v[p] = r < R ? *reinterpret_cast<const float4*>(X + (size_t)r * K + k0 + q * 4) : make_float4(0, 0, 0, 0);
        }
    };
    load(cur, 0);
    for (int k0 = 0; k0 < K; k0 += C) {
        if (k0 + C < K) load(nxt, k0 + C);
#pragma unroll
        for (int p = 0; p < P; ++p) acc += cur[p].x + cur[p].y + cur[p].z + cur[p].w;
        __syncthreads();
#pragma unroll
        for (int p = 0; p < P; ++p) cur[p] = nxt[p];
    }
    if (acc == 123.456f) out[0] = acc;
}

template <int C>
void run(const float* X, int R, int K, float* out) {
    const int nb = (R + 127) / 128;
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    for (int i = 0; i < 3; ++i) k_walk<C><<<nb, 256>>>(X, R, K, out);
    hipEventRecord(a);
    for (int i = 0; i < 20; ++i) k_walk<C><<<nb, 256>>>(X, R, K, out);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    printf("R=%d K=%d slab %4d B/row: %7.1f us  %7.1f GB/s\n", R, K, C * 4, ms / 20 * 1e3, (double)R * K * 4 / (ms / 20 * 1e-3) / 1e9);
}

int main() {
    for (int R : {51639, 413000}) {
        const int K = 1024;
        float *X, *out;
        hipMalloc(&X, (size_t)R * K * 4); hipMalloc(&out, 4);
        hipMemset(X, 0, (size_t)R * K * 4);
        run<32>(X, R, K, out); run<64>(X, R, K, out); run<128>(X, R, K, out); run<256>(X, R, K, out);
        hipFree(X); hipFree(out);
    }
    return 0;
}
