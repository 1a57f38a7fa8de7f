// Device-side helpers shared by the gfx950 kernels of libgatmi.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/gatmi.h"

namespace gatmi {

constexpr float kLog2e = 1.4426950408889634f;

// exp(x) for x <= 0 (softmax numerators): one v_exp_f32 after scaling the already-subtracted
// argument, so the rounding error of the scaling is relative to |x| and not to the logits.
__device__ __forceinline__ float exp_neg(float x) { return __builtin_amdgcn_exp2f(x * kLog2e); }

__device__ __forceinline__ float leaky(float z, float slope) { return fmaxf(z, z * slope); }

// max of values read straight from memory (mask entries): fmaxf would first quiet possible signalling NaNs with
// one extra v_max per operand; a mask holds none, and the comparisons that follow treat any NaN as "not visited"
__device__ __forceinline__ float raw_max(float a, float b) {
    float r;
    asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ float raw_max3(float a, float b, float c) {
    float r;
    asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}

// Cross-lane steps without LDS traffic (DPP modifiers; a __shfl is a ds_bpermute with LDS latency).
// CTRL: 0xB1 / 0x4E swap lanes inside quads (xor 1 / xor 2), 0x141 / 0x140 mirror half rows / rows of
// 16 lanes, 0x142 / 0x143 broadcast lane 15 / 31 of a row to the following rows, 0x100 + n takes the
// lane n places up within its row of 16.
template <int CTRL, int ROW_MASK = 0xF>
__device__ __forceinline__ int dpp_i(int old, int v) { return __builtin_amdgcn_update_dpp(old, v, CTRL, ROW_MASK, 0xF, false); }
template <int CTRL, int ROW_MASK = 0xF>
__device__ __forceinline__ float dpp_f(float old, float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, old), __builtin_bit_cast(int, v), CTRL, ROW_MASK, 0xF, false));
}

// Sum over each aligned group of `N` adjacent lanes (N a power of two <= 64), returned in every lane
// of the group and bit-identical in all of them: every step adds two numbers that are each the same
// across the lanes that read them.  Up to 16 lanes the steps are DPP modifiers (quad swaps, then the
// mirror of a half row / row: after the quad steps a quad holds one value, so the mirrored lane
// carries the other quad's sum); wider groups go on with ds_bpermute.
template <int N>
__device__ __forceinline__ float group_sum(float v) {
    if constexpr (N >= 2) v += dpp_f<0xB1>(v, v);
    if constexpr (N >= 4) v += dpp_f<0x4E>(v, v);
    if constexpr (N >= 8) v += dpp_f<0x141>(v, v);
    if constexpr (N >= 16) v += dpp_f<0x140>(v, v);
#pragma unroll
    for (int o = 16; o < N; o <<= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// maximum over the 64 lanes of a wave, returned in every lane
__device__ __forceinline__ float wave_max(float v) {
    v = fmaxf(v, dpp_f<0xB1>(v, v));
    v = fmaxf(v, dpp_f<0x4E>(v, v));
    v = fmaxf(v, dpp_f<0x141>(v, v));
    v = fmaxf(v, dpp_f<0x140>(v, v));            // every lane of a row of 16 holds the row's maximum
    v = fmaxf(v, dpp_f<0x142, 0xA>(v, v));       // rows 1 and 3 take in rows 0 and 2
    v = fmaxf(v, dpp_f<0x143, 0xC>(v, v));       // rows 2 and 3 take in lane 31: lane 63 holds the total
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}

// inclusive prefix sum over the 64 lanes of a wave (lane l gets v[0] + ... + v[l])
__device__ __forceinline__ int wave_scan_add(int v) {
    v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xF, 0xF, true);     // row_shr:1, lanes without a source add 0
    v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xF, 0xF, true);
    v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xF, 0xF, true);
    v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xF, 0xF, true);     // scan inside each row of 16 lanes
    v += dpp_i<0x142, 0xA>(0, v);                                      // rows 1 and 3 add the totals of rows 0 and 2
    v += dpp_i<0x143, 0xC>(0, v);                                      // rows 2 and 3 add the total of rows 0 and 1
    return v;
}

// OR over each aligned group of 8 lanes, valid in the group's first lane
__device__ __forceinline__ uint32_t or8_to_first(uint32_t w) {
    w |= (uint32_t)dpp_i<0xB1>(0, (int)w);
    w |= (uint32_t)dpp_i<0x4E>(0, (int)w);
    w |= (uint32_t)dpp_i<0x104>(0, (int)w);      // lane i takes lane i + 4 of its row
    return w;
}

// Counter-based dropout mask: a 32-bit integer mixer (two multiply / xor-shift rounds) of the 64-bit
// (seed, index) pair folded to one word; keep when the top 24 bits, read as a fraction in [0,1), fall
// below keep_prob.  About a dozen 32-bit operations: the mask is evaluated per entry and head inside the
// gather kernels, where a 64-bit mixer (two 64-bit multiplies) was a visible share of the time at the
// reference's dropout rates.  The same function is restated on the host by the tests to replay masks.
__device__ __forceinline__ bool keep_mask(uint64_t seed, uint64_t idx, float keep_prob) {
    uint32_t x = ((uint32_t)idx ^ (uint32_t)seed) + ((uint32_t)(idx >> 32) ^ (uint32_t)(seed >> 32)) * 0x9E3779B9u;
    x ^= x >> 16;
    x *= 0x7FEB352Du;
    x ^= x >> 15;
    x *= 0x846CA68Bu;
    x ^= x >> 16;
    return (float)(x >> 8) * (1.0f / 16777216.0f) < keep_prob;
}

constexpr uint64_t kFeatSeedSalt = 0x5DEECE66Dull;

// Dropout under HIP-graph replay: kernel arguments are frozen at capture, so a captured step would
// replay one set of masks for ever.  A launch may therefore carry, next to its seed, the address of a
// 64-bit step word in device memory (gatmi_dropout_step_source); the kernel reads it when it runs and
// uses seed + word * kSeedStepMul.  The odd multiplier keeps (seed + small head-group offset, step)
// pairs apart.  nullptr: the seed as passed.  A uniform read-only load: the compiler issues it as a
// scalar load once per use site.
//
// The (stepped) seed then goes through the splitmix64 finaliser before any mask is drawn from it.
// keep_mask / feat_group_bits fold the seed into the index by XOR, so two RAW seeds that differ in a
// few low bits (seed + head group, seed = epoch) would give masks that are XOR-permutations of each
// other's index space; after the finaliser neighbouring seeds share no such structure.  All of it is
// uniform arithmetic on kernel arguments: scalar instructions, once per kernel.
constexpr uint64_t kSeedStepMul = 0xD1B54A32D192ED03ull;
__host__ __device__ __forceinline__ uint64_t scramble_seed(uint64_t z) {
    z ^= z >> 30;
    z *= 0xBF58476D1CE4E5B9ull;
    z ^= z >> 27;
    z *= 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
template <class Args>
__device__ __forceinline__ uint64_t seed_now(const Args &a) {
    return scramble_seed(a.seed_step ? a.seed + *a.seed_step * kSeedStepMul : a.seed);
}

// Feature dropout (utils/layers.py:22): ONE hash serves four consecutive elements of the [rows, hd] table,
// 16 bits each — the keep probability is resolved to 2^-16 — because the row kernels apply it to every row they
// gather: hashed element by element it doubled the forward kernel's time at the reference's rate of 0.6.
// Element e = row * hd + column belongs to group e >> 2 and takes bits 16 (e & 3) .. 16 (e & 3) + 15.
__device__ __forceinline__ uint32_t mix32(uint32_t x) {
    x ^= x >> 16;
    x *= 0x7FEB352Du;
    x ^= x >> 15;
    x *= 0x846CA68Bu;
    return x ^ (x >> 16);
}
// 64 bits for one group: the mixer of keep_mask on the folded (salted seed, group) pair, and once more on its result
__device__ __forceinline__ uint64_t feat_group_bits(uint64_t seed, uint64_t group) {
    const uint64_t s = seed ^ kFeatSeedSalt;
    const uint32_t lo = mix32(((uint32_t)group ^ (uint32_t)s) + ((uint32_t)(group >> 32) ^ (uint32_t)(s >> 32)) * 0x9E3779B9u);
    const uint32_t hi = mix32(lo + 0x9E3779B9u);
    return ((uint64_t)hi << 32) | lo;
}
__device__ __forceinline__ uint32_t feat_keep_threshold(float keep) { return (uint32_t)(keep * 65536.0f); }

// the four 16-bit fields of z against keep: kept elements scaled by 1/keep, the others zero
__device__ __forceinline__ float4 drop_by_bits(float4 v, uint64_t z, float keep) {
    const uint32_t lo = (uint32_t)z, hi = (uint32_t)(z >> 32), thr = feat_keep_threshold(keep);
    const float inv = 1.0f / keep;
    v.x = (lo & 0xffffu) < thr ? v.x * inv : 0.f;
    v.y = (lo >> 16) < thr ? v.y * inv : 0.f;
    v.z = (hi & 0xffffu) < thr ? v.z * inv : 0.f;
    v.w = (hi >> 16) < thr ? v.w * inv : 0.f;
    return v;
}

// feature-dropout of one float4 of row j (columns c0..c0+3, c0 and hd multiples of 4: one group)
__device__ __forceinline__ float4 drop_feat(float4 v, uint64_t seed, int64_t j, int hd, int c0, float keep) {
    return drop_by_bits(v, feat_group_bits(seed, ((uint64_t)j * (uint64_t)hd + (uint64_t)c0) >> 2), keep);
}

// Dropout of the layer's INPUT (utils/layers.py:9) inside the projection kernels: the same 64 bits per group of
// four elements, drawn under another salt; element (row, k) of the [n, F] input (F a multiple of 4) belongs to
// group row * (F / 4) + k / 4 and takes field k & 3.  The forward projection masks the tile it loads, the weight
// gradient masks the same tile again, and the input gradient is masked as it is stored: no dropped copy of X.
constexpr uint64_t kInSeedSalt = 0x9FB21C651E98DF25ull;
__device__ __forceinline__ float4 drop_in4(float4 v, uint64_t seed, int64_t row, int f4, int k4, float keep) {
    return drop_by_bits(v, feat_group_bits(seed ^ kInSeedSalt, (uint64_t)row * (uint64_t)f4 + (uint64_t)k4), keep);
}

// one element of the same mask (row j, column c)
__device__ __forceinline__ float drop_feat1(float v, uint64_t seed, int64_t j, int hd, int c, float keep) {
    const uint64_t e = (uint64_t)j * (uint64_t)hd + (uint64_t)c;
    const uint32_t field = (uint32_t)(feat_group_bits(seed, e >> 2) >> (16 * (e & 3))) & 0xffffu;
    return field < feat_keep_threshold(keep) ? v / keep : 0.f;
}

// coefficient-dropout factor of entry (i, j) of graph b, head hg of a dense [B, N, N] batch:
// 0 or 1/keep (utils/layers.py:20)
__device__ __forceinline__ float coef_drop_scale(uint64_t seed, int64_t b, int N, int H, int i, int j, int hg, float keep) {
    const uint64_t idx = (((uint64_t)b * N + (uint64_t)i) * N + (uint64_t)j) * H + hg;
    return keep_mask(seed, idx, keep) ? 1.0f / keep : 0.f;
}

// float4 arithmetic of the row kernels (one float4 of a feature row per lane)
// Written on two-element vectors so that the compiler can use the packed fp32 instructions of
// CDNA3/4 (v_pk_fma_f32, v_pk_mul_f32, v_pk_add_f32: two lanes' worth of fp32 per issue slot).
typedef float f32x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ float4 f4_fma(float a, float4 x, float4 y) {
    const f32x2_t s = {a, a};
    const f32x2_t lo = __builtin_elementwise_fma(s, (f32x2_t){x.x, x.y}, (f32x2_t){y.x, y.y});
    const f32x2_t hi = __builtin_elementwise_fma(s, (f32x2_t){x.z, x.w}, (f32x2_t){y.z, y.w});
    return make_float4(lo.x, lo.y, hi.x, hi.y);
}
__device__ __forceinline__ float4 f4_scale(float a, float4 x) {
    const f32x2_t s = {a, a};
    const f32x2_t lo = s * (f32x2_t){x.x, x.y}, hi = s * (f32x2_t){x.z, x.w};
    return make_float4(lo.x, lo.y, hi.x, hi.y);
}
__device__ __forceinline__ float4 f4_add(float4 x, float4 y) {
    const f32x2_t lo = (f32x2_t){x.x, x.y} + (f32x2_t){y.x, y.y}, hi = (f32x2_t){x.z, x.w} + (f32x2_t){y.z, y.w};
    return make_float4(lo.x, lo.y, hi.x, hi.y);
}
// Store of a result row that this kernel does not read again: the non-temporal hint keeps the written
// lines from displacing the gathered table in L2.
__device__ __forceinline__ void store_stream4(float *p, float4 v) {
    typedef float f32x4_t __attribute__((ext_vector_type(4)));
    __builtin_nontemporal_store((f32x4_t){v.x, v.y, v.z, v.w}, reinterpret_cast<f32x4_t *>(p));
}
// Load of a row that a streaming kernel reads exactly once.  (Index streams are better left alone: with
// the hint the four ids of a batch are no longer fetched as one 16-byte load.)
__device__ __forceinline__ float4 load_stream4(const float *p) {
    typedef float f32x4_t __attribute__((ext_vector_type(4)));
    const f32x4_t v = __builtin_nontemporal_load(reinterpret_cast<const f32x4_t *>(p));
    return make_float4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ float dot4(float4 x, float4 y) { return fmaf(x.x, y.x, fmaf(x.y, y.y, fmaf(x.z, y.z, x.w * y.w))); }

__device__ __forceinline__ float elu_f(float x) { return x > 0.f ? x : expm1f(x); }

// derivative of ELU written in terms of its OUTPUT y: 1 for y > 0, y + 1 (= exp(x)) otherwise
__device__ __forceinline__ float elu_grad_from_out(float y) { return y > 0.f ? 1.f : y + 1.f; }

// the layer's output activation (utils/layers.py:34) and its derivative in terms of the output
__device__ __forceinline__ float act_f(float x, int activation) {
    return activation == GATMI_ACT_ELU ? elu_f(x) : (activation == GATMI_ACT_RELU ? fmaxf(x, 0.f) : x);
}
__device__ __forceinline__ float act_grad_from_out(float y, int activation) {
    return activation == GATMI_ACT_ELU ? elu_grad_from_out(y) : (activation == GATMI_ACT_RELU ? (y > 0.f ? 1.f : 0.f) : 1.f);
}
// the same derivative in terms of the activation's INPUT x (ELU: exp(x) for x <= 0)
__device__ __forceinline__ float act_grad_from_pre(float x, int activation) {
    return activation == GATMI_ACT_ELU ? (x > 0.f ? 1.f : expf(x)) : (activation == GATMI_ACT_RELU ? (x > 0.f ? 1.f : 0.f) : 1.f);
}
__host__ __device__ inline bool act_known(int activation) {
    return activation == GATMI_ACT_IDENTITY || activation == GATMI_ACT_ELU || activation == GATMI_ACT_RELU;
}

// bf16 storage of gathered tables: two values per dword, element 0 in the low half.
__device__ __forceinline__ uint32_t pack_bf16x2(float a, float b) {
    typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    const bf16x2 r = __builtin_convertvector((f32x2){a, b}, bf16x2);   // v_cvt_pk_bf16_f32, round to nearest even
    return __builtin_bit_cast(uint32_t, r);
}

// Four consecutive elements c0..c0+3 of row `row` of a table with row stride `hd` elements, stored as fp32 or bf16.
template <bool BF16>
__device__ __forceinline__ float4 load_row4(const void *base, int64_t row, int64_t hd, int c0) {
    if constexpr (BF16) {
        const uint2 u = *reinterpret_cast<const uint2 *>(reinterpret_cast<const uint16_t *>(base) + row * hd + c0);
        return make_float4(__uint_as_float(u.x << 16), __uint_as_float(u.x & 0xffff0000u),
                           __uint_as_float(u.y << 16), __uint_as_float(u.y & 0xffff0000u));
    } else {
        return *reinterpret_cast<const float4 *>(reinterpret_cast<const float *>(base) + row * hd + c0);
    }
}

// Fixed-order column sum of per-block partial results.  Launch with dim3(64, 16) threads: thread
// (cx, ry) adds rows ry, ry+16, ... of its column, the 16 partial sums meet in LDS and row 0 adds
// them in order, so the result does not depend on scheduling.  `load(r)` returns row r's value.
// Eight loads are in flight per thread (the adds keep their order): a chain of dependent loads
// made a 4096-row sum take 0.1 ms.
constexpr int kSumCols = 64, kSumRows = 16;
template <typename Load>
__device__ __forceinline__ float block_column_sum(int rows, Load load) {
    __shared__ float part[kSumRows][kSumCols];
    float t = 0.f;
    for (int r = threadIdx.y; r < rows; r += 8 * kSumRows) {
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = r + u * kSumRows < rows ? load(r + u * kSumRows) : 0.f;
#pragma unroll
        for (int u = 0; u < 8; ++u) t += v[u];
    }
    part[threadIdx.y][threadIdx.x] = t;
    __syncthreads();
    float total = 0.f;
    if (threadIdx.y == 0) {
#pragma unroll
        for (int k = 0; k < kSumRows; ++k) total += part[k][threadIdx.x];
    }
    return total;
}

}  // namespace gatmi

#define GATMI_LAUNCH_CHECK(name)                                                         \
    do {                                                                                 \
        hipError_t e__ = hipGetLastError();                                              \
        if (e__ != hipSuccess)                                                           \
            return gatmi::fail(GATMI_EHIP, "%s: %s", name, hipGetErrorString(e__));      \
    } while (0)
