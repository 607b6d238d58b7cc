// fp32 MFMA (v_mfma_f32_32x32x2_f32) chip-wide rate: NACC independent accumulator chains per wave, WPS waves per SIMD.
// hipcc --offload-arch=gfx950 -O3 tools/micro/mfma_f32_peak.hip -o tools/micro/mfma_f32_peak && ./mfma_f32_peak
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int NACC>
__global__ __launch_bounds__(512) void k(float* out, int iters, float a0, float b0) {
    f32x16 acc[NACC];
    for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    float a = a0 + threadIdx.x, b = b0;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 16; ++u)
#pragma unroll
            for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
    }
    float s = 0.f;
    for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int NACC>
void run(int threads, int blocks, const char* label) {
    float* out; hipMalloc(&out, sizeof(float) * threads * blocks);
    const int iters = 2000 / NACC;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<NACC><<<blocks, threads>>>(out, iters, 1.f, 2.f);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int r = 0; r < 5; ++r) k<NACC><<<blocks, threads>>>(out, iters, 1.f, 2.f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 5;
    const double mfma = (double)blocks * (threads / 64) * iters * 16 * NACC;
    printf("%-34s %8.1f us  %7.1f TFLOP/s  (%.1f cycles per MFMA per SIMD at 2.4 GHz)\n", label, ms * 1e3, mfma * 4096 / ms / 1e9,
           ms * 1e-3 * 2.4e9 / (mfma / 1024));
    hipFree(out);
}
int main() {
    run<1>(256, 256, "1 wave/SIMD, 1 accumulator");
    run<2>(256, 256, "1 wave/SIMD, 2 accumulators");
    run<4>(256, 256, "1 wave/SIMD, 4 accumulators");
    run<1>(512, 256, "2 waves/SIMD, 1 accumulator");
    run<2>(512, 256, "2 waves/SIMD, 2 accumulators");
    run<1>(256, 1024, "4 waves/SIMD, 1 accumulator");
    run<4>(512, 256, "2 waves/SIMD, 4 accumulators");
    return 0;
}
