// Issue rate of the integer multiplies a counter-based dropout hash can be built from, on gfx950:
// v_mul_lo_u32 (32 x 32 -> low 32) against v_mul_u32_u24 / v_mad_u32_u24 (24 x 24 -> 32) and a plain xor-shift step.
// Build: hipcc -O3 --offload-arch=gfx950 int_mul_rate.hip -o int_mul_rate     Run: ./int_mul_rate
#include <hip/hip_runtime.h>
#include <cstdio>

constexpr int kIters = 4096;
constexpr int kChains = 4;   // independent chains per lane: latency is hidden, issue rate shows

template <int MODE>
__global__ __launch_bounds__(256) void chain_kernel(uint32_t *out, uint32_t c) {
    uint32_t x[kChains];
    for (int k = 0; k < kChains; ++k) x[k] = threadIdx.x * 2654435761u + k + blockIdx.x;
    for (int it = 0; it < kIters; ++it) {
#pragma unroll
        for (int k = 0; k < kChains; ++k) {
            if (MODE == 0) x[k] = x[k] * c;                                   // v_mul_lo_u32
            if (MODE == 1) x[k] = __umul24(x[k], c);                          // v_mul_u32_u24
            if (MODE == 2) x[k] = x[k] ^ (x[k] >> 15);                        // shift + xor
            if (MODE == 3) x[k] = __umul24(x[k], c) + (x[k] >> 24);           // v_mad_u32_u24 with the top byte folded in
            if (MODE == 4) x[k] = __builtin_amdgcn_alignbit(x[k], x[k], 13) + c;   // rotate + add
            if (MODE == 5) x[k] = x[k] * c + (x[k] >> 24);                    // v_mul_lo_u32 (or v_mad_u64_u32) in a chain the compiler cannot fold
            if (MODE == 6) x[k] = (x[k] ^ (x[k] >> 16)) * c;                  // one round of the two-round mixer: shift, xor, 32-bit multiply
            if (MODE == 7) x[k] = __umul24(x[k] ^ (x[k] >> 16), c) ^ (x[k] >> 24);   // the same with the 24-bit multiply
        }
    }
    uint32_t r = 0;
    for (int k = 0; k < kChains; ++k) r ^= x[k];
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}

template <int MODE>
static float run(uint32_t *out, const char *name, int ops_per_step) {
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const int blocks = 256 * 8;       // 8 blocks of 4 waves per CU: 8 waves per SIMD
    hipLaunchKernelGGL(chain_kernel<MODE>, dim3(blocks), dim3(256), 0, 0, out, 0x7FEB35u);
    hipEventRecord(e0);
    hipLaunchKernelGGL(chain_kernel<MODE>, dim3(blocks), dim3(256), 0, 0, out, 0x7FEB35u);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    // wave-instructions per SIMD: blocks * 4 waves * iters * chains * ops / 1024 SIMDs
    const double winst = (double)blocks * 4 * kIters * kChains * ops_per_step / 1024.0;
    printf("%-34s %8.3f ms   %.2f ns per wave-instruction per SIMD (%d per step)\n", name, ms, ms * 1e6 / winst, ops_per_step);
    return ms;
}

int main() {
    uint32_t *out;
    hipMalloc(&out, 256 * 8 * 256 * sizeof(uint32_t));
    run<2>(out, "x ^= x >> 15", 2);
    run<0>(out, "x *= c (v_mul_lo_u32)", 1);
    run<1>(out, "x = mul24(x, c)", 1);
    run<3>(out, "x = mul24(x, c) + (x >> 24)", 2);
    run<4>(out, "x = rotl(x) + c", 2);
    run<5>(out, "x = x * c + (x >> 24)", 3);
    run<6>(out, "x = (x ^ (x >> 16)) * c", 3);
    run<7>(out, "x = mul24(x ^ x >> 16, c) ^ x >> 24", 5);
    hipFree(out);
    return 0;
}
