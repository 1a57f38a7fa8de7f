// Inner-product decoder for gfx950: utils/layers.py:307-313.
//     out[b, i, j] = act( <Zd[b,i,:], Zd[b,j,:]> ),   Zd = dropout(Z),   act = sigmoid (default) or identity
// The [B, N, N] result is the only large array (the kernel is bound by writing it once); the
// contraction over the F features runs on v_mfma_f32_16x16x4_f32.  Block (b, 64 destination rows):
// wave w owns rows 16*w .. 16*w+15, keeps their features as the A operand in registers and walks
// the graph's rows in LDS-staged chunks as the B operand.
//
// Backward: with S = (dOut + dOut^T) . act'(x) (act' = out (1 - out) for the sigmoid, which is
// symmetric because x is), dZd = S . Zd.  The A operand S[i, j] is built per lane from one direct
// and one transposed read of dOut and one read of the saved output; Zd chunks are the B operand.
#include <hip/hip_runtime.h>

#include <algorithm>

#include "common.h"
#include "device_common.h"

namespace gatmi {

typedef float f32x4 __attribute__((ext_vector_type(4)));

struct IpdArgs {
    int N, F;
    const float *Z, *out_saved, *dOut;
    float *out, *dZ;
    float keep;
    uint64_t seed;
    const uint64_t *seed_step;   // device word added to the seed at run time (seed_now), or nullptr
    int activation;
};

template <int NTF>
struct IpdShape {
    static constexpr int FP = 16 * NTF;            // feature columns, zero-padded
    static constexpr int LD = FP + 4;              // LDS row stride: 4 mod 8 keeps the operand reads spread over the banks
    static constexpr int KS = FP / 4;              // MFMA k-steps of the forward
    static constexpr int JC = NTF <= 4 ? 128 : 64; // rows staged per chunk
};

__device__ __forceinline__ float ipd_load(const IpdArgs &a, int64_t row, int c) {
    float v = a.Z[row * a.F + c];
    if (a.keep < 1.f) v = drop_feat1(v, seed_now(a), row, a.F, c, a.keep);
    return v;
}

// rows jc .. jc+JC-1 of graph b into LDS (dropout applied, zero outside the graph / feature range)
template <int NTF>
__device__ __forceinline__ void ipd_stage(const IpdArgs &a, int64_t b, int jc, float *zs) {
    using S = IpdShape<NTF>;
    for (int e = threadIdx.x; e < S::JC * S::FP; e += 256) {
        const int r = e / S::FP, c = e % S::FP;
        float v = 0.f;
        if (jc + r < a.N && c < a.F) v = ipd_load(a, b * a.N + jc + r, c);
        zs[r * S::LD + c] = v;
    }
}

template <int NTF>
__global__ __launch_bounds__(256) void ipd_fwd_kernel(IpdArgs a) {
    using S = IpdShape<NTF>;
    __shared__ float zs[S::JC * S::LD];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int m = lane & 15, kq = lane >> 4;
    const int N = a.N;
    const int64_t b = blockIdx.x;
    const int i0 = blockIdx.y * 64 + wave * 16;
    const int64_t irow = b * N + min(i0 + m, N - 1);
    float av[S::KS];
#pragma unroll
    for (int s = 0; s < S::KS; ++s) av[s] = (4 * s + kq < a.F) ? ipd_load(a, irow, 4 * s + kq) : 0.f;

    for (int jc = 0; jc < N; jc += S::JC) {
        __syncthreads();
        ipd_stage<NTF>(a, b, jc, zs);
        __syncthreads();
        if (i0 >= N) continue;
        const int jend = min(S::JC, N - jc);
        for (int jt = 0; jt < jend; jt += 16) {
            f32x4 c = (f32x4){0.f, 0.f, 0.f, 0.f};
            const float *zr = zs + (jt + m) * S::LD + kq;
#pragma unroll
            for (int s = 0; s < S::KS; ++s) c = __builtin_amdgcn_mfma_f32_16x16x4f32(av[s], zr[4 * s], c, 0, 0, 0);
            const int j = jc + jt + m;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int i = i0 + 4 * kq + r;
                if (i < N && j < N) {
                    float v = c[r];
                    if (a.activation == GATMI_ACT_SIGMOID) v = __builtin_amdgcn_rcpf(1.f + __expf(-v));
                    a.out[(b * N + i) * (int64_t)N + j] = v;
                }
            }
        }
    }
}

template <int NTF>
__global__ __launch_bounds__(256) void ipd_bwd_kernel(IpdArgs a) {
    using S = IpdShape<NTF>;
    __shared__ float zs[S::JC * S::LD];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int m = lane & 15, kq = lane >> 4;
    const int N = a.N;
    const int64_t b = blockIdx.x;
    const int i0 = blockIdx.y * 64 + wave * 16;
    const bool iok = i0 + m < N;
    const int irow = min(i0 + m, N - 1);
    const float *grow = a.dOut + (b * N + irow) * (int64_t)N;
    const float *orow = a.out_saved ? a.out_saved + (b * N + irow) * (int64_t)N : nullptr;
    const float *gcol = a.dOut + b * N * (int64_t)N + irow;
    f32x4 acc[NTF];
#pragma unroll
    for (int nt = 0; nt < NTF; ++nt) acc[nt] = (f32x4){0.f, 0.f, 0.f, 0.f};

    for (int jc = 0; jc < N; jc += S::JC) {
        __syncthreads();
        ipd_stage<NTF>(a, b, jc, zs);
        __syncthreads();
        if (i0 >= N) continue;
        const int jend = min(S::JC, N - jc);
        for (int jt = 0; jt < jend; jt += 16) {
            float sv[4];
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                const int j = jc + jt + 4 * kq + s;
                float v = 0.f;
                if (iok && j < N) {
                    v = grow[j] + gcol[(int64_t)j * N];
                    if (orow) { const float o = orow[j]; v *= o * (1.f - o); }
                }
                sv[s] = v;
            }
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                const float *zr = zs + (jt + 4 * kq + s) * S::LD + m;
#pragma unroll
                for (int nt = 0; nt < NTF; ++nt) acc[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(sv[s], zr[16 * nt], acc[nt], 0, 0, 0);
            }
        }
    }
    if (i0 >= N) return;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int i = i0 + 4 * kq + r;
        if (i >= N) continue;
#pragma unroll
        for (int nt = 0; nt < NTF; ++nt) {
            const int c = 16 * nt + m;
            if (c < a.F) {
                float v = acc[nt][r];
                if (a.keep < 1.f) v = drop_feat1(v, seed_now(a), b * N + i, a.F, c, a.keep);
                a.dZ[(b * N + i) * (int64_t)a.F + c] = v;
            }
        }
    }
}

// ------------------------------------------------------------------------------------------
// Small graphs (the whole Zd of a graph fits LDS next to a second block): block (graph b, split y)
// stages Zd once, then its eight waves need no barrier.  Forward: the (row tile, column tile) pairs
// are dealt to the waves in contiguous runs, so a wave keeps its A operand over a run and writes
// neighbouring 64-byte segments of the same output rows.  Backward: a wave owns row tiles and takes
// two column tiles per step, all 24 loads of a step in flight before its MFMAs.
// ------------------------------------------------------------------------------------------
constexpr int kGraphWaves = 8;
constexpr size_t kGraphLdsMax = 80 * 1024;

template <int NTF>
__device__ __forceinline__ void ipd_stage_graph(const IpdArgs &a, int64_t b, int Np, float *zs) {
    using S = IpdShape<NTF>;
    if ((a.F & 3) == 0 && (reinterpret_cast<uintptr_t>(a.Z) & 15) == 0) {      // rows are 16-byte aligned: four columns per load
        for (int e = threadIdx.x; e < Np * (S::FP / 4); e += 64 * kGraphWaves) {
            const int r = e / (S::FP / 4), c = 4 * (e % (S::FP / 4));
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (r < a.N && c < a.F) {
                v = *reinterpret_cast<const float4 *>(a.Z + (b * a.N + r) * (int64_t)a.F + c);
                if (a.keep < 1.f) v = drop_feat(v, seed_now(a), b * a.N + r, a.F, c, a.keep);
            }
            *reinterpret_cast<float4 *>(zs + r * S::LD + c) = v;
        }
        return;
    }
    for (int e = threadIdx.x; e < Np * S::FP; e += 64 * kGraphWaves) {
        const int r = e / S::FP, c = e % S::FP;
        float v = 0.f;
        if (r < a.N && c < a.F) v = ipd_load(a, b * a.N + r, c);
        zs[r * S::LD + c] = v;
    }
}

template <int NTF>
__global__ __launch_bounds__(64 * kGraphWaves) void ipd_fwd_graph_kernel(IpdArgs a) {
    using S = IpdShape<NTF>;
    extern __shared__ __attribute__((aligned(16))) float zs_graph[];
    float *zs = zs_graph;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int m = lane & 15, kq = lane >> 4;
    const int N = a.N, Np = (N + 15) & ~15, nt = Np / 16;
    const int64_t b = blockIdx.x;
    ipd_stage_graph<NTF>(a, b, Np, zs);
    __syncthreads();
    const int n_pairs = nt * nt;
    const int n_waves = gridDim.y * kGraphWaves;
    const int per = (n_pairs + n_waves - 1) / n_waves;
    const int w = blockIdx.y * kGraphWaves + wave;
    const int p_end = min(n_pairs, (w + 1) * per);
    int cur = -1;
    float av[S::KS];
    for (int p = w * per; p < p_end; ++p) {
        const int it = p / nt, jt = p - it * nt;
        if (it != cur) {
            cur = it;
            const float *ar = zs + (16 * it + m) * S::LD + kq;
#pragma unroll
            for (int s = 0; s < S::KS; ++s) av[s] = ar[4 * s];
        }
        f32x4 c = (f32x4){0.f, 0.f, 0.f, 0.f};
        const float *zr = zs + (16 * jt + m) * S::LD + kq;
#pragma unroll
        for (int s = 0; s < S::KS; ++s) c = __builtin_amdgcn_mfma_f32_16x16x4f32(av[s], zr[4 * s], c, 0, 0, 0);
        const int j = 16 * jt + m;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int i = 16 * it + 4 * kq + r;
            if (i < N && j < N) {
                float v = c[r];
                if (a.activation == GATMI_ACT_SIGMOID) v = __builtin_amdgcn_rcpf(1.f + __expf(-v));
                a.out[(b * N + i) * (int64_t)N + j] = v;
            }
        }
    }
}

template <int NTF>
__global__ __launch_bounds__(64 * kGraphWaves) void ipd_bwd_graph_kernel(IpdArgs a) {
    using S = IpdShape<NTF>;
    extern __shared__ __attribute__((aligned(16))) float zs_graph[];
    float *zs = zs_graph;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int m = lane & 15, kq = lane >> 4;
    const int N = a.N, Np = (N + 15) & ~15, nt = Np / 16;
    const int64_t b = blockIdx.x;
    ipd_stage_graph<NTF>(a, b, Np, zs);
    __syncthreads();
    const float *gmat = a.dOut + b * N * (int64_t)N;
    const float *omat = a.out_saved ? a.out_saved + b * N * (int64_t)N : nullptr;
    const bool vec = (N & 3) == 0 && (reinterpret_cast<uintptr_t>(a.dOut) & 15) == 0 &&
                     (!a.out_saved || (reinterpret_cast<uintptr_t>(a.out_saved) & 15) == 0);
    const bool vec2 = (N & 1) == 0 && (reinterpret_cast<uintptr_t>(a.dOut) & 7) == 0 &&
                      (!a.out_saved || (reinterpret_cast<uintptr_t>(a.out_saved) & 7) == 0);
    for (int it = blockIdx.y * kGraphWaves + wave; it < nt; it += gridDim.y * kGraphWaves) {
        const int i0 = 16 * it;
        const bool iok = i0 + m < N;
        const uint32_t irow = (uint32_t)min(i0 + m, N - 1);
        f32x4 acc[NTF];
#pragma unroll
        for (int nt_ = 0; nt_ < NTF; ++nt_) acc[nt_] = (f32x4){0.f, 0.f, 0.f, 0.f};
        for (int j0 = 0; j0 < Np; j0 += 32) {
            float gd[2][4], gt[2][4], ov[2][4];
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int jb = j0 + 16 * h + 4 * kq;
                if (vec && jb + 3 < N) {                  // rows are 16-byte aligned: the direct reads are one load each
                    const float4 g4 = *reinterpret_cast<const float4 *>(gmat + irow * (uint32_t)N + jb);
                    gd[h][0] = g4.x; gd[h][1] = g4.y; gd[h][2] = g4.z; gd[h][3] = g4.w;
                    if (omat) {
                        const float4 o4 = *reinterpret_cast<const float4 *>(omat + irow * (uint32_t)N + jb);
                        ov[h][0] = o4.x; ov[h][1] = o4.y; ov[h][2] = o4.z; ov[h][3] = o4.w;
                    }
#pragma unroll
                    for (int s = 0; s < 4; ++s) gt[h][s] = gmat[(uint32_t)(jb + s) * (uint32_t)N + irow];
                } else if (vec2 && jb + 3 < N) {          // an even number of nodes (the 246-node atlas): rows are 8-byte aligned
                    const float2 g0 = *reinterpret_cast<const float2 *>(gmat + irow * (uint32_t)N + jb);
                    const float2 g1 = *reinterpret_cast<const float2 *>(gmat + irow * (uint32_t)N + jb + 2);
                    gd[h][0] = g0.x; gd[h][1] = g0.y; gd[h][2] = g1.x; gd[h][3] = g1.y;
                    if (omat) {
                        const float2 o0 = *reinterpret_cast<const float2 *>(omat + irow * (uint32_t)N + jb);
                        const float2 o1 = *reinterpret_cast<const float2 *>(omat + irow * (uint32_t)N + jb + 2);
                        ov[h][0] = o0.x; ov[h][1] = o0.y; ov[h][2] = o1.x; ov[h][3] = o1.y;
                    }
#pragma unroll
                    for (int s = 0; s < 4; ++s) gt[h][s] = gmat[(uint32_t)(jb + s) * (uint32_t)N + irow];
                } else {
#pragma unroll
                    for (int s = 0; s < 4; ++s) {
                        const uint32_t j = (uint32_t)min(jb + s, N - 1);
                        gd[h][s] = gmat[irow * (uint32_t)N + j];
                        gt[h][s] = gmat[j * (uint32_t)N + irow];
                        ov[h][s] = omat ? omat[irow * (uint32_t)N + j] : 0.5f;
                    }
                }
            }
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                if (j0 + 16 * h >= Np) break;
#pragma unroll
                for (int s = 0; s < 4; ++s) {
                    const int j = j0 + 16 * h + 4 * kq + s;
                    float v = gd[h][s] + gt[h][s];
                    if (omat) v *= ov[h][s] * (1.f - ov[h][s]);
                    if (!(iok && j < N)) v = 0.f;
                    const float *zr = zs + j * S::LD + m;
#pragma unroll
                    for (int nt_ = 0; nt_ < NTF; ++nt_) acc[nt_] = __builtin_amdgcn_mfma_f32_16x16x4f32(v, zr[16 * nt_], acc[nt_], 0, 0, 0);
                }
            }
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int i = i0 + 4 * kq + r;
            if (i >= N) continue;
#pragma unroll
            for (int nt_ = 0; nt_ < NTF; ++nt_) {
                const int c = 16 * nt_ + m;
                if (c < a.F) {
                    float v = acc[nt_][r];
                    if (a.keep < 1.f) v = drop_feat1(v, seed_now(a), b * N + i, a.F, c, a.keep);
                    a.dZ[(b * N + i) * (int64_t)a.F + c] = v;
                }
            }
        }
    }
}

}  // namespace gatmi

using namespace gatmi;

static int ipd_check(int64_t B, int32_t N, int32_t F, int32_t activation, float keep, const char *name) {
    GATMI_REQUIRE(B >= 0 && N >= 0 && F >= 1, GATMI_EINVAL, "%s: bad size B=%lld N=%d F=%d", name, (long long)B, N, F);
    GATMI_REQUIRE(F <= 128, GATMI_EUNSUPPORTED, "%s: feature width %d > 128 not built", name, F);
    GATMI_REQUIRE(activation == GATMI_ACT_SIGMOID || activation == GATMI_ACT_IDENTITY, GATMI_EUNSUPPORTED,
                  "%s: activation %d (sigmoid or identity)", name, activation);
    GATMI_REQUIRE(keep > 0.f && keep <= 1.f, GATMI_EINVAL, "%s: keep probability %g outside (0, 1]", name, keep);
    GATMI_REQUIRE(B <= 0x7fffffff && (N + 63) / 64 <= 65535, GATMI_ERANGE, "%s: too large for one launch", name);
    return GATMI_OK;
}

// kernel: the chunked kernel (any N); gkernel: the whole-graph kernel, taken when the graph's Zd fits
// kGraphLdsMax bytes of LDS (N * 64 <= 2^31 there, so 32-bit offsets inside a graph are safe)
#define GATMI_IPD_LAUNCH(kernel, gkernel, name)                                             \
    do {                                                                                    \
        hipStream_t st = reinterpret_cast<hipStream_t>(stream);                             \
        const int ntf = (F + 15) / 16;                                                      \
        const int ntf_pad = ntf <= 1 ? 1 : (ntf <= 2 ? 2 : (ntf <= 4 ? 4 : 8));             \
        const int np = (N + 15) & ~15;                                                      \
        const size_t lds = (size_t)np * (size_t)(16 * ntf_pad + 4) * sizeof(float);         \
        if (lds <= kGraphLdsMax) {                                                          \
            const int tiles = np / 16;                                                      \
            const int splits = (int)std::max<int64_t>(1, std::min<int64_t>((tiles + kGraphWaves - 1) / kGraphWaves, 2048 / std::max<int64_t>(B, 1))); \
            const dim3 grid((unsigned)B, (unsigned)splits);                                 \
            hipError_t e = hipSuccess;                                                      \
            if (ntf_pad == 1) { if (lds > 48 * 1024) e = hipFuncSetAttribute(reinterpret_cast<const void *>(&gkernel<1>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
                                if (e == hipSuccess) hipLaunchKernelGGL((gkernel<1>), grid, dim3(64 * kGraphWaves), lds, st, a); }  \
            else if (ntf_pad == 2) { if (lds > 48 * 1024) e = hipFuncSetAttribute(reinterpret_cast<const void *>(&gkernel<2>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
                                if (e == hipSuccess) hipLaunchKernelGGL((gkernel<2>), grid, dim3(64 * kGraphWaves), lds, st, a); }  \
            else if (ntf_pad == 4) { if (lds > 48 * 1024) e = hipFuncSetAttribute(reinterpret_cast<const void *>(&gkernel<4>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
                                if (e == hipSuccess) hipLaunchKernelGGL((gkernel<4>), grid, dim3(64 * kGraphWaves), lds, st, a); }  \
            else { if (lds > 48 * 1024) e = hipFuncSetAttribute(reinterpret_cast<const void *>(&gkernel<8>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
                                if (e == hipSuccess) hipLaunchKernelGGL((gkernel<8>), grid, dim3(64 * kGraphWaves), lds, st, a); }  \
            if (e != hipSuccess) return fail(GATMI_EHIP, "%s: %s", name, hipGetErrorString(e)); \
        } else {                                                                            \
            const dim3 grid((unsigned)B, (unsigned)((N + 63) / 64));                        \
            if (ntf_pad == 1) hipLaunchKernelGGL((kernel<1>), grid, dim3(256), 0, st, a);   \
            else if (ntf_pad == 2) hipLaunchKernelGGL((kernel<2>), grid, dim3(256), 0, st, a); \
            else if (ntf_pad == 4) hipLaunchKernelGGL((kernel<4>), grid, dim3(256), 0, st, a); \
            else hipLaunchKernelGGL((kernel<8>), grid, dim3(256), 0, st, a);                \
        }                                                                                   \
        GATMI_LAUNCH_CHECK(name);                                                           \
    } while (0)

extern "C" int gatmi_ipd_fwd(int64_t B, int32_t N, int32_t F, const float *Z, int32_t activation,
                             float keep, uint64_t seed, float *out, void *stream) {
    int rc = ipd_check(B, N, F, activation, keep, "ipd_fwd");
    if (rc) return rc;
    if (B == 0 || N == 0) return GATMI_OK;
    GATMI_REQUIRE(Z && out, GATMI_EINVAL, "ipd_fwd: null pointer");
    IpdArgs a = {};
    a.N = N; a.F = F; a.Z = Z; a.out = out; a.keep = keep; a.seed = seed; a.seed_step = keep < 1.f ? dropout_step_source() : nullptr; a.activation = activation;
    GATMI_IPD_LAUNCH(ipd_fwd_kernel, ipd_fwd_graph_kernel, "ipd_fwd");
    return GATMI_OK;
}

extern "C" int gatmi_ipd_bwd(int64_t B, int32_t N, int32_t F, const float *dOut, const float *out,
                             const float *Z, int32_t activation, float keep, uint64_t seed, float *dZ, void *stream) {
    int rc = ipd_check(B, N, F, activation, keep, "ipd_bwd");
    if (rc) return rc;
    if (B == 0 || N == 0) return GATMI_OK;
    GATMI_REQUIRE(dOut && Z && dZ, GATMI_EINVAL, "ipd_bwd: null pointer");
    GATMI_REQUIRE(activation != GATMI_ACT_SIGMOID || out, GATMI_EINVAL, "ipd_bwd: the sigmoid needs the saved output");
    IpdArgs a = {};
    a.N = N; a.F = F; a.Z = Z; a.dOut = dOut; a.out_saved = activation == GATMI_ACT_SIGMOID ? out : nullptr;
    a.dZ = dZ; a.keep = keep; a.seed = seed; a.seed_step = keep < 1.f ? dropout_step_source() : nullptr; a.activation = activation;
    GATMI_IPD_LAUNCH(ipd_bwd_kernel, ipd_bwd_graph_kernel, "ipd_bwd");
    return GATMI_OK;
}
