// What the exact-fp32 matrix instruction sustains on this part: v_mfma_f32_16x16x4_f32 issued back to back on
// 16 / 32 independent accumulators per wave, 1..4 waves per SIMD, nothing else in the loop.  Prints TFLOP/s.
// Build: hipcc --offload-arch=gfx950 -O3 -o mfma_f32_peak mfma_f32_peak.hip
#include <hip/hip_runtime.h>
#include <cstdio>

typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int NACC>
__global__ __launch_bounds__(256) void mfma_loop(float *out, int iters, float a0, float b0) {
    f32x4 acc[NACC];
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    float a = a0 + threadIdx.x, b = b0 + threadIdx.x;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
    }
    float t = 0.f;
#pragma unroll
    for (int i = 0; i < NACC; ++i) t += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = t;
}

template <int NACC>
static void run(int blocks_per_cu, int threads) {
    const int cus = 256, iters = 4096;
    float *out;
    hipMalloc(&out, sizeof(float) * cus * blocks_per_cu * threads);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(mfma_loop<NACC>, dim3(cus * blocks_per_cu), dim3(threads), 0, 0, out, iters, 1.0f, 2.0f);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
    }
    float ms = 0.f;
    hipEventElapsedTime(&ms, e0, e1);
    const double waves = (double)cus * blocks_per_cu * threads / 64;
    const double flop = waves * iters * NACC * 2.0 * 16 * 16 * 4;
    printf("{\"acc_per_wave\": %d, \"waves_per_simd\": %.2f, \"ms\": %.3f, \"tflops\": %.1f}\n", NACC,
           blocks_per_cu * threads / 64 / 4.0, ms, flop / ms * 1e-9);
    hipFree(out);
}

int main() {
    run<16>(1, 256);
    run<32>(1, 256);
    run<32>(2, 256);
    run<32>(3, 256);
    run<32>(4, 256);
    return 0;
}
