// Brain-network variant of the attention step for gfx950: utils/layers.py:171-179 with the
// coefficients of attn_v2e_BNF_V2 (:229-237).  Per graph b, head h, destination i:
//     C_ij   = softmax_j( LeakyReLU(f1[i,h] + f2[j,h]) )              (no mask)
//     n_ij   = m2[h] * [A_ij > 0] + A_ij                               (A = weighted adjacency)
//     agg_i  = sum_j n_ij (m1[h] + C_ij) X_j
//            = m1 (m2 U_i + V_i) + T2_i / s_i,   U = [A>0].X,  V = A.X,  T2 = sum_j n_ij p_ij X_j,
// with p_ij the un-normalised softmax numerator and s_i its row sum.  All contractions over j run
// on v_mfma_f32_16x16x4_f32 in the layout of dense_attn.hip: lane (i = l&15, kq = l>>4) computes
// its own A-operand values for sources j0 + 4*kq + t, X tiles come from LDS as the B operand.
//
// Backward (per source tile, all heads in one wave): D_ij = <G_i, X_j> by MFMA (K = feature
// width) lands in the C layout; there C, n, w = n (m1 + C) are recomputed, and w^T . G (C layout
// reused as the A layout of the transpose) accumulates dX_j.  dA_ij = (m1 + C_ij) D_ij is written
// per tile, df2 / dm1 / dm2 are register sums, and df1_i = sum_j dlogit_ij is summed over the 16 sources of a wave
// with DPP steps, over the waves in LDS and over the source blocks by a small second kernel (c_i = <G_i, T2_i / s_i>
// comes from a row kernel that runs first).
#include <hip/hip_runtime.h>

#include <algorithm>

#include "common.h"
#include "device_common.h"

namespace gatmi {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int kBnfJC = 64;      // sources (forward) / destinations (backward) staged in LDS per step
constexpr int kBnfBlocksMax = 1 << 20;
constexpr int kBnfUVBlocks = 512;   // blocks per group from which the head-independent sums get a pass of their own (below: a launch more costs more than it saves)

struct BnfFwdArgs {
    int N;
    const float *X, *A, *f1, *f2, *m1, *m2;
    float *out, *t2n, *lse, *q;
    float *uv;                  // [2][nodes][FP]: U = [A>0].X and V = A.X, written by the UV pass and read by the heads pass
    float slope;
    int64_t nodes;              // B * N: group g (blockIdx.z) reads f1, f2 at + g * nodes * H, saves at + g * nodes * H (* FP)
};

// Block (b, 64 destination rows): wave w owns rows 16*w .. 16*w+15 and all H heads.
// MODE kBnfAll: everything in one pass.  U and V do not depend on the head, so a launch that covers several groups of
// heads would form them once per group: kBnfUV forms them alone (no softmax work) and kBnfHeads reads them back.
constexpr int kBnfAll = 0, kBnfUV = 1, kBnfHeads = 2;
template <int FP, int H, bool SAVE, int MODE = kBnfAll>
__global__ __launch_bounds__(256) void bnf_attn_fwd_kernel(BnfFwdArgs a) {
    constexpr int LD = FP + 4;
    constexpr int NT = (FP + 15) / 16;
    constexpr bool NARROW = FP < 16;
    __shared__ __attribute__((aligned(16))) float xs[kBnfJC * LD];
    __shared__ float f2s[kBnfJC * H];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int m = lane & 15, kq = lane >> 4;
    const int N = a.N;
    const int64_t b = blockIdx.x;
    const int i0 = blockIdx.y * 64 + wave * 16;
    const int irow = min(i0 + m, N - 1);
    const float *arow = a.A + (b * N + irow) * (int64_t)N;
    // a group of H heads per blockIdx.z: its own logit halves, scalars and saved tensors, its share of the output rows
    const int64_t goff = (int64_t)blockIdx.z * a.nodes * H;
    a.f1 += goff; a.f2 += goff;
    a.m1 += blockIdx.z * H; a.m2 += blockIdx.z * H;
    a.out += (int64_t)blockIdx.z * H * a.nodes * FP;       // out is head-major: [groups * H][nodes][FP]
    if constexpr (SAVE) { a.t2n += goff * FP; a.lse += goff; a.q += 2 * goff; }
    const float *f2b = a.f2 + b * N * (int64_t)H;
    const bool vec = (N & 3) == 0, vec2 = (N & 1) == 0 && (reinterpret_cast<uintptr_t>(a.A) & 7) == 0;
    const float maskf = (!NARROW || m < FP) ? 1.f : 0.f;
    const int mcl = (!NARROW || m < FP) ? m : 0;

    float f1v[H], mx[H], m1v[H], m2v[H];
#pragma unroll
    for (int h = 0; h < H; ++h) {
        f1v[h] = a.f1[(b * N + irow) * H + h];
        mx[h] = -INFINITY;
        m1v[h] = a.m1[h];
        m2v[h] = a.m2[h];
    }
    // ---- sweep 1: row maxima of LeakyReLU(f1_i + f2_j) --------------------------------------
    for (int jc = 0; MODE != kBnfUV && jc < N; jc += kBnfJC) {
        __syncthreads();
        for (int e = threadIdx.x; e < kBnfJC * H; e += 256) f2s[e] = (jc + e / H < N) ? f2b[(int64_t)jc * H + e] : -INFINITY;
        __syncthreads();
        const int jend = min(kBnfJC, N - jc);
        for (int jl = 0; jl < jend; jl += 16)
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int h = 0; h < H; ++h) mx[h] = fmaxf(mx[h], leaky(f1v[h] + f2s[(jl + 4 * kq + t) * H + h], a.slope));
    }
#pragma unroll
    for (int h = 0; h < H; ++h) {
        mx[h] = fmaxf(mx[h], __shfl_xor(mx[h], 16, 64));
        mx[h] = fmaxf(mx[h], __shfl_xor(mx[h], 32, 64));
    }
    // ---- sweep 2 ----------------------------------------------------------------------------------
    f32x4 accU[NT], accV[NT], acc2[H][NT];
    float ssum[H], qsum[H];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) accU[nt] = accV[nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int h = 0; h < H; ++h) {
        ssum[h] = qsum[h] = 0.f;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) acc2[h][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
    }
    for (int jc = 0; jc < N; jc += kBnfJC) {
        __syncthreads();
        for (int e = threadIdx.x; e < kBnfJC * (FP / 4); e += 256) {
            const int r = e / (FP / 4), c = e % (FP / 4);
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (jc + r < N) v = *reinterpret_cast<const float4 *>(a.X + (b * N + jc + r) * (int64_t)FP + 4 * c);
            *reinterpret_cast<float4 *>(xs + r * LD + 4 * c) = v;
        }
        if constexpr (MODE != kBnfUV)
            for (int e = threadIdx.x; e < kBnfJC * H; e += 256) f2s[e] = (jc + e / H < N) ? f2b[(int64_t)jc * H + e] : -INFINITY;
        __syncthreads();
        const int jend = min(kBnfJC, N - jc);
        for (int jl = 0; jl < jend; jl += 16) {
            const int j = jc + jl + 4 * kq;
            float av[4];
            if (vec && j < N) {
                const float4 t4 = *reinterpret_cast<const float4 *>(arow + j);
                av[0] = t4.x; av[1] = t4.y; av[2] = t4.z; av[3] = t4.w;
            } else if (vec2 && j + 3 < N) {               // an even number of nodes (the 246-node atlas): 8-byte aligned rows
                const float2 t0 = *reinterpret_cast<const float2 *>(arow + j), t1 = *reinterpret_cast<const float2 *>(arow + j + 2);
                av[0] = t0.x; av[1] = t0.y; av[2] = t1.x; av[3] = t1.y;
            } else {
#pragma unroll
                for (int t = 0; t < 4; ++t) av[t] = (j + t < N) ? arow[j + t] : 0.f;
            }
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const float *xrow = xs + (jl + 4 * kq + t) * LD;
                float xr[NT];
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) xr[nt] = NARROW ? xrow[mcl] * maskf : xrow[16 * nt + m];
                const float th = av[t] > 0.f ? 1.f : 0.f;
                if constexpr (MODE != kBnfHeads) {
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) {
                        accU[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(th, xr[nt], accU[nt], 0, 0, 0);
                        accV[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[t], xr[nt], accV[nt], 0, 0, 0);
                    }
                }
#pragma unroll
                for (int h = 0; MODE != kBnfUV && h < H; ++h) {
                    const float z = f1v[h] + f2s[(jl + 4 * kq + t) * H + h];     // -inf for j >= N => p = 0
                    const float p = exp_neg(leaky(z, a.slope) - mx[h]);
                    ssum[h] += z > 0.f ? 0.f : p;            // the mass on negative logits; qsum: on positive ones
                    qsum[h] += z > 0.f ? p : 0.f;
                    const float np_ = (m2v[h] * th + av[t]) * p;
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) acc2[h][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(np_, xr[nt], acc2[h][nt], 0, 0, 0);
                }
            }
        }
    }
    if (i0 >= N) return;
    if constexpr (MODE == kBnfUV) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int i = i0 + 4 * kq + r;
            if (i >= N) continue;
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                const int c = 16 * nt + m;
                if (c < FP) {
                    a.uv[(b * N + i) * FP + c] = accU[nt][r];
                    a.uv[(a.nodes + b * N + i) * FP + c] = accV[nt][r];
                }
            }
        }
        return;
    }
#pragma unroll
    for (int h = 0; h < H; ++h) {
        ssum[h] += __shfl_xor(ssum[h], 16, 64); ssum[h] += __shfl_xor(ssum[h], 32, 64);
        qsum[h] += __shfl_xor(qsum[h], 16, 64); qsum[h] += __shfl_xor(qsum[h], 32, 64);
    }
    // C layout: lane (kq, m) holds rows 4*kq + r, column 16*nt + m
#pragma unroll
    for (int h = 0; h < H; ++h) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int i = i0 + 4 * kq + r;
            const float sneg = __shfl(ssum[h], 4 * kq + r, 64), qv = __shfl(qsum[h], 4 * kq + r, 64);
            const float s = sneg + qv;
            const float mrow = __shfl(mx[h], 4 * kq + r, 64);
            const float inv = s > 0.f ? 1.0f / s : 0.f;
            if (i >= N) continue;
            const int64_t node = b * N + i;
            if constexpr (SAVE) {
                if (m == 0) {
                    a.lse[node * H + h] = mrow + logf(s);
                    // the shares of the row's softmax mass on positive and on negative logits, each from its own sum: the
                    // small one of the two keeps its relative precision (1 - the other would not)
                    *reinterpret_cast<float2 *>(a.q + (node * H + h) * 2) = make_float2(qv * inv, sneg * inv);
                }
            }
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                const int c = 16 * nt + m;
                if (c < FP) {
                    const int64_t o = (node * H + h) * FP + c;
                    const float t2 = acc2[h][nt][r] * inv;
                    const float u = MODE == kBnfHeads ? a.uv[node * FP + c] : accU[nt][r];
                    const float v = MODE == kBnfHeads ? a.uv[(a.nodes + node) * FP + c] : accV[nt][r];
                    a.out[(h * a.nodes + node) * FP + c] = m1v[h] * (m2v[h] * u + v) + t2;
                    if constexpr (SAVE) a.t2n[o] = t2;
                }
            }
        }
    }
}

struct BnfBwdArgs {
    int N;
    const float *G, *stats, *q, *X, *A, *f2, *m1, *m2; // stats [nodes, H] x {f1, lse, c, positive share}; q [nodes, H] x {positive, negative share}
    float *dX, *dA, *df1p, *df2, *ws;                  // df1p [source blocks][nodes, H]: every block's share of df1
    float slope;
    int accumulate;             // dX, dA: add to what is there (a later group of heads of the same step)
    int64_t nodes, pairs;       // B * N, B * N * N: group g (blockIdx.z) reads G, stats, f2 and writes dX, dA, df1p, df2, ws at + g * its size
};

// Block (b, 64 sources): wave w owns sources 16*w .. 16*w+15, all heads, and walks all destinations.
template <int FP, int H>
__global__ __launch_bounds__(256, 3) void bnf_attn_bwd_kernel(BnfBwdArgs a) {      // three waves per SIMD: at most 168 registers
    constexpr int W = H * FP;
    constexpr int LDG = W + 4;                 // rows 4 banks apart: the A reads (row m, column kq) and the B reads (row 4 kq + r, column m) of a wave hit 64 banks
    constexpr int NT = (FP + 15) / 16;
    constexpr int KS = FP / 4;
    constexpr bool NARROW = FP < 16;
    __shared__ __attribute__((aligned(16))) float gs[kBnfJC * LDG];       // G rows (all heads) of the destination chunk
    __shared__ __attribute__((aligned(16))) float ss[kBnfJC * H * 4];     // {f1, lse, c, -}
    __shared__ float red[4][2 * H];
    __shared__ float sn[kBnfJC * H];                                       // negative-logit share of the chunk's rows
    __shared__ float p1[4][kBnfJC * H];                                    // every wave's share of df1 for the chunk's rows
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int m = lane & 15, kq = lane >> 4;
    const int N = a.N;
    const int64_t b = blockIdx.x;
    const int j0 = blockIdx.y * 64 + wave * 16;
    const int jcol = min(j0 + m, N - 1);
    const bool jok = j0 + m < N;
    {
        const int64_t g = blockIdx.z;
        a.G += g * a.nodes * W;                      // G is head-major: [groups * H][nodes][FP]
        a.q += g * a.nodes * (2 * H);
        a.stats += g * a.nodes * (4 * H); a.f2 += g * a.nodes * H;
        a.df1p += g * (int64_t)gridDim.y * a.nodes * H;
        a.m1 += g * H; a.m2 += g * H;
        a.dX += g * a.nodes * FP; a.dA += g * a.pairs; a.df2 += g * a.nodes * H;
        a.ws += g * (int64_t)gridDim.x * gridDim.y * (2 * H);
    }
    const float maskf = (!NARROW || m < FP) ? 1.f : 0.f;
    const int mcl = (!NARROW || m < FP) ? m : 0;

    float f2v[H], df2[H], dm1[H], dm2[H], m1v[H], m2v[H], xb[KS];
    f32x4 acc[NT];
#pragma unroll
    for (int h = 0; h < H; ++h) {
        f2v[h] = a.f2[(b * N + jcol) * H + h];
        df2[h] = dm1[h] = dm2[h] = 0.f;
        m1v[h] = a.m1[h];
        m2v[h] = a.m2[h];
    }
#pragma unroll
    for (int s = 0; s < KS; ++s) xb[s] = a.X[(b * N + jcol) * (int64_t)FP + 4 * s + kq];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) acc[nt] = (f32x4){0.f, 0.f, 0.f, 0.f};

    for (int ic = 0; ic < N; ic += kBnfJC) {
        __syncthreads();
        for (int e = threadIdx.x; e < kBnfJC * (W / 2); e += 256) {
            const int r = e / (W / 2), c = e % (W / 2);
            float2 v = make_float2(0.f, 0.f);
            if (ic + r < N) v = *reinterpret_cast<const float2 *>(a.G + (((2 * c) / FP) * a.nodes + b * N + ic + r) * (int64_t)FP + (2 * c) % FP);
            *reinterpret_cast<float2 *>(gs + r * LDG + 2 * c) = v;
        }
        for (int e = threadIdx.x; e < kBnfJC * H; e += 256) {
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            float qn = 0.f;
            if (ic + e / H < N) {
                v = *reinterpret_cast<const float4 *>(a.stats + ((b * N + ic) * (int64_t)H + e) * 4);
                qn = a.q[((b * N + ic) * (int64_t)H + e) * 2 + 1];
            }
            *reinterpret_cast<float4 *>(ss + e * 4) = v;
            sn[e] = qn;
        }
        for (int e = threadIdx.x; e < 4 * kBnfJC * H; e += 256) (&p1[0][0])[e] = 0.f;
        __syncthreads();
        if (j0 < N) {
        const int iend = min(kBnfJC, N - ic);
        for (int il = 0; il < iend; il += 16) {
            float av[4], th[4], dA[4];
            bool ok[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int i = ic + il + 4 * kq + r;
                ok[r] = i < N && jok;
                av[r] = ok[r] ? a.A[(b * N + i) * (int64_t)N + jcol] : 0.f;
                th[r] = av[r] > 0.f ? 1.f : 0.f;
                dA[r] = 0.f;
            }
#pragma unroll
            for (int h = 0; h < H; ++h) {
                f32x4 D = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int s = 0; s < KS; ++s)
                    D = __builtin_amdgcn_mfma_f32_16x16x4f32(gs[(il + m) * LDG + h * FP + 4 * s + kq], xb[s], D, 0, 0, 0);
                float wv[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float4 st = *reinterpret_cast<const float4 *>(ss + ((il + 4 * kq + r) * H + h) * 4);
                    const float z = st.x + f2v[h];
                    const float C = ok[r] ? exp_neg(leaky(z, a.slope) - st.y) : 0.f;
                    const float n = m2v[h] * th[r] + av[r];
                    const float dn = (m1v[h] + C) * D[r];
                    wv[r] = ok[r] ? n * (m1v[h] + C) : 0.f;
                    if (ok[r]) {
                        dA[r] += dn;
                        dm1[h] += n * D[r];
                        dm2[h] += th[r] * dn;
                    }
                    const float cnd = C * n * D[r];                                            // 0 where !ok: C = 0
                    const float sp = z > 0.f ? 1.f : a.slope;
                    df2[h] += (cnd - C * st.z) * sp;
                    // df1_i = sum_j C (n D - c_i) sp with c_i = sum_j C n D  =  (1 - slope) (qneg_i sum_{z > 0} C n D - qpos_i
                    // sum_{z <= 0} C n D): term by term 0 in a row whose logits all lie on one side of the kink, no c_i from
                    // another summation order, and the small share of a nearly one-sided row at full relative precision.
                    // Summed over the wave's 16 sources here.
                    const float rowsum = (1.f - a.slope) * group_sum<16>(cnd * (z > 0.f ? sn[(il + 4 * kq + r) * H + h] : -st.w));
                    if (m == 0) p1[wave][(il + 4 * kq + r) * H + h] = rowsum;
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float *grow = gs + (il + 4 * kq + r) * LDG + h * FP;
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) {
                        const float bval = NARROW ? grow[mcl] * maskf : grow[16 * nt + m];
                        acc[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(wv[r], bval, acc[nt], 0, 0, 0);
                    }
                }
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int i = ic + il + 4 * kq + r;
                if (ok[r]) {
                    float *p = a.dA + (b * N + i) * (int64_t)N + jcol;
                    *p = a.accumulate ? *p + dA[r] : dA[r];
                }
            }
        }
        }
        __syncthreads();
        for (int e = threadIdx.x; e < kBnfJC * H; e += 256)
            if (ic + e / H < N)
                a.df1p[((int64_t)blockIdx.y * a.nodes + b * N + ic) * H + e] = (p1[0][e] + p1[1][e]) + (p1[2][e] + p1[3][e]);
    }
    // per-wave totals of dm1 / dm2 -> block partial (fixed order) -> ws[block][2H]
#pragma unroll
    for (int h = 0; h < H; ++h) {
        float u = dm1[h], v = dm2[h];
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            u += __shfl_xor(u, o, 64);
            v += __shfl_xor(v, o, 64);
        }
        if (lane == 0) {
            red[wave][h] = u;
            red[wave][H + h] = v;
        }
    }
    __syncthreads();
    if (threadIdx.x < 2 * H) {
        const int64_t blk = (int64_t)blockIdx.x * gridDim.y + blockIdx.y;
        a.ws[blk * 2 * H + threadIdx.x] = red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
    }
    if (j0 >= N) return;
#pragma unroll
    for (int h = 0; h < H; ++h) {
        float t = df2[h];
        t += __shfl_xor(t, 16, 64);
        t += __shfl_xor(t, 32, 64);
        if (kq == 0 && jok) a.df2[(b * N + j0 + m) * H + h] = t;
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int j = j0 + 4 * kq + r;
        if (j >= N) continue;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            const int c = 16 * nt + m;
            if (c < FP) {
                float *p = a.dX + (b * N + j) * (int64_t)FP + c;
                *p = a.accumulate ? *p + acc[nt][r] : acc[nt][r];
            }
        }
    }
}

__global__ void bnf_dm_reduce_kernel(const float *ws, int nblocks, int width, float *out) {
    ws += (int64_t)blockIdx.y * nblocks * width;              // a group of heads per blockIdx.y
    out += blockIdx.y * width;
    const int c = blockIdx.x * kSumCols + threadIdx.x;
    const bool ok = c < width;
    const float t = block_column_sum(nblocks, [&](int r) { return ok ? ws[(int64_t)r * width + c] : 0.f; });
    if (ok && threadIdx.y == 0) out[c] = t;
}

// stats[row] = {f1, lse, c = <G_row, t2n_row>, the share of the row's softmax mass on positive logits} for every
// (group, node, head) row of FP floats; 64 / LP rows per wave
template <int FP>
__global__ __launch_bounds__(256) void bnf_row_stats_kernel(const float *G, const float *t2n, const float *f1, const float *lse,
                                                            const float *q, float slope, int64_t rows, int64_t nodes, int H, float *stats) {
    constexpr int LP = FP / 4;                                    // lanes per row, a float4 each
    const int lane = threadIdx.x & 63;
    const int64_t row = ((int64_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * (64 / LP) + lane / LP;
    const bool ok = row < rows;
    const int64_t rw = ok ? row : 0;
    const int64_t o = rw * FP + 4 * (lane % LP);                             // t2n, f1, lse, q, stats: [group][node][head]
    const int64_t grp = rw / (nodes * H), node = rw / H % nodes;
    const int64_t og = ((grp * H + rw % H) * nodes + node) * FP + 4 * (lane % LP);     // G: [group][head][node]
    const float4 g = *reinterpret_cast<const float4 *>(G + og), t = *reinterpret_cast<const float4 *>(t2n + o);
    const float c = group_sum<LP>((g.x * t.x + g.y * t.y) + (g.z * t.z + g.w * t.w));
    if (ok && lane % LP == 0) *reinterpret_cast<float4 *>(stats + row * 4) = make_float4(f1[row], lse[row], c, q[2 * row]);
}

// dst[e] (+)= sum over groups of part[g][e], in group order; blockIdx.y: one such sum per slab of part and dst
__global__ __launch_bounds__(256) void bnf_group_sum_kernel(const float *part, int groups, int64_t count, int accumulate, float *dst) {
    part += (int64_t)blockIdx.y * groups * count;
    dst += (int64_t)blockIdx.y * count;
    const int64_t stride = (int64_t)gridDim.x * 256;
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < count; e += stride) {
        float t = accumulate ? dst[e] : 0.f;
        for (int g = 0; g < groups; ++g) t += part[g * count + e];
        dst[e] = t;
    }
}

#define GATMI_BNF_SHAPES(X) X(4, 1) X(4, 2) X(8, 1) X(8, 2) X(16, 1) X(16, 2) X(32, 1) X(32, 2) X(64, 1) X(64, 2)

}  // namespace gatmi

using namespace gatmi;

static int bnf_check(int64_t B, int32_t N, int32_t H, int32_t fp, const char *name) {
    GATMI_REQUIRE(B >= 0 && N >= 0, GATMI_EINVAL, "%s: negative size", name);
    GATMI_REQUIRE((fp == 4 || fp == 8 || fp == 16 || fp == 32 || fp == 64) && (H == 1 || H == 2), GATMI_EUNSUPPORTED,
                  "%s: feature width %d (4..64, power of two) with %d heads (1 or 2) not built", name, fp, H);
    GATMI_REQUIRE(B <= 0x7fffffff && (int64_t)B * ((N + 63) / 64) < kBnfBlocksMax * 64LL, GATMI_ERANGE, "%s: batch too large for one launch", name);
    return GATMI_OK;
}

extern "C" size_t gatmi_bnf_attn_fwd_ws_floats(int64_t B, int32_t N, int32_t fp, int32_t groups) {
    return groups > 1 ? 2 * (size_t)B * N * fp : 1;
}

extern "C" int gatmi_bnf_attn_fwd(int64_t B, int32_t N, int32_t H, int32_t fp,
                                  const float *X, const float *A, const float *f1, const float *f2,
                                  const float *m1, const float *m2, float slope,
                                  float *out, float *t2n, float *lse, float *q, float *ws, int32_t groups, void *stream) {
    int rc = bnf_check(B, N, H, fp, "bnf_attn_fwd");
    if (rc) return rc;
    GATMI_REQUIRE(groups >= 1 && groups <= 1024, GATMI_EINVAL, "bnf_attn_fwd: groups must be 1..1024");
    if (B == 0 || N == 0) return GATMI_OK;
    GATMI_REQUIRE(X && A && f1 && f2 && m1 && m2 && out && (groups == 1 || ws), GATMI_EINVAL, "bnf_attn_fwd: null pointer");
    const bool save = t2n || lse || q;
    GATMI_REQUIRE(!save || (t2n && lse && q), GATMI_EINVAL, "bnf_attn_fwd: t2n, lse and q must be given together");
    GATMI_REQUIRE_ALIGNED(X); GATMI_REQUIRE_ALIGNED(A);
    BnfFwdArgs a;
    a.N = N; a.X = X; a.A = A; a.f1 = f1; a.f2 = f2; a.m1 = m1; a.m2 = m2;
    a.out = out; a.t2n = t2n; a.lse = lse; a.q = q; a.uv = ws; a.slope = slope; a.nodes = B * N;
    const dim3 grid((unsigned)B, (unsigned)((N + 63) / 64), (unsigned)groups);
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
#define X_(ff, hh)                                                                                  \
    if (fp == ff && H == hh) {                                                                      \
        if (groups > 1 && (int64_t)grid.x * grid.y >= kBnfUVBlocks) {                               \
            hipLaunchKernelGGL((bnf_attn_fwd_kernel<ff, hh, false, kBnfUV>), dim3(grid.x, grid.y), dim3(256), 0, st, a); \
            if (save) hipLaunchKernelGGL((bnf_attn_fwd_kernel<ff, hh, true, kBnfHeads>), grid, dim3(256), 0, st, a); \
            else hipLaunchKernelGGL((bnf_attn_fwd_kernel<ff, hh, false, kBnfHeads>), grid, dim3(256), 0, st, a);   \
        } else if (save) hipLaunchKernelGGL((bnf_attn_fwd_kernel<ff, hh, true>), grid, dim3(256), 0, st, a); \
        else hipLaunchKernelGGL((bnf_attn_fwd_kernel<ff, hh, false>), grid, dim3(256), 0, st, a);     \
        GATMI_LAUNCH_CHECK("bnf_attn_fwd");                                                         \
        return GATMI_OK;                                                                            \
    }
    GATMI_BNF_SHAPES(X_)
#undef X_
    return fail(GATMI_EUNSUPPORTED, "bnf_attn_fwd: no kernel for fp=%d H=%d", fp, H);
}

static size_t bnf_bwd_blocks(int64_t B, int32_t N) { return (size_t)B * (size_t)((N + 63) / 64); }

extern "C" size_t gatmi_bnf_attn_bwd_ws_floats(int64_t B, int32_t N, int32_t H, int32_t fp, int32_t groups) {
    const size_t nodes = (size_t)B * N, jb = (size_t)((N + 63) / 64);
    const size_t stats = (size_t)groups * nodes * H * 4;
    const size_t df1p = (size_t)groups * jb * nodes * H;
    const size_t dm = bnf_bwd_blocks(B, N) * (size_t)(2 * H) * (size_t)groups;
    // several groups in one launch: every group's share of dX and dA, summed afterwards
    return stats + df1p + dm + (groups > 1 ? (size_t)groups * (nodes * fp + nodes * N) : 0);
}

extern "C" int gatmi_bnf_attn_bwd(int64_t B, int32_t N, int32_t H, int32_t fp,
                                  const float *G, const float *t2n, const float *f1, const float *lse, const float *q,
                                  const float *X, const float *A,
                                  const float *f2, const float *m1, const float *m2, float slope,
                                  float *dX, float *dA, int accumulate, float *df1, float *df2, float *dm, float *ws, int32_t groups, void *stream) {
    int rc = bnf_check(B, N, H, fp, "bnf_attn_bwd");
    if (rc) return rc;
    GATMI_REQUIRE(groups >= 1 && groups <= 1024, GATMI_EINVAL, "bnf_attn_bwd: groups must be 1..1024");
    if (B == 0 || N == 0) return GATMI_OK;
    GATMI_REQUIRE(G && t2n && f1 && lse && q && X && A && f2 && m1 && m2 && dX && dA && df1 && df2 && dm && ws, GATMI_EINVAL, "bnf_attn_bwd: null pointer");
    GATMI_REQUIRE_ALIGNED(G); GATMI_REQUIRE_ALIGNED(t2n); GATMI_REQUIRE_ALIGNED(ws);
    const int64_t nodes = B * N, pairs = nodes * N;
    const int jb = (N + 63) / 64;
    const int nblocks = (int)bnf_bwd_blocks(B, N);
    float *stats = ws;
    float *df1p = stats + (size_t)groups * nodes * H * 4;
    float *dmws = df1p + (size_t)groups * jb * nodes * H;
    float *part = dmws + (size_t)nblocks * (2 * H) * groups;               // [groups][dX] then [groups][dA]
    BnfBwdArgs a;
    a.N = N; a.G = G; a.stats = stats; a.q = q; a.X = X; a.A = A; a.f2 = f2; a.m1 = m1; a.m2 = m2;
    a.dX = dX; a.dA = dA; a.df1p = df1p; a.df2 = df2; a.ws = dmws; a.slope = slope; a.accumulate = accumulate;
    a.nodes = nodes; a.pairs = pairs;
    if (groups > 1) { a.dX = part; a.dA = part + (size_t)groups * nodes * fp; a.accumulate = 0; }
    const dim3 grid((unsigned)B, (unsigned)jb, (unsigned)groups);
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const int64_t rows = (int64_t)groups * nodes * H;
#define X_(ff, hh)                                                                         \
    if (fp == ff && H == hh) {                                                             \
        const int64_t per_block = 4 * (64 / (ff / 4));                                     \
        hipLaunchKernelGGL(bnf_row_stats_kernel<ff>, dim3((unsigned)((rows + per_block - 1) / per_block)), dim3(256), 0, st, G, t2n, f1, lse, q, slope, rows, nodes, hh, stats); \
        GATMI_LAUNCH_CHECK("bnf_row_stats");                                               \
        hipLaunchKernelGGL((bnf_attn_bwd_kernel<ff, hh>), grid, dim3(256), 0, st, a);       \
        GATMI_LAUNCH_CHECK("bnf_attn_bwd");                                                \
        hipLaunchKernelGGL(bnf_dm_reduce_kernel, dim3(1, groups), dim3(kSumCols, kSumRows), 0, st, dmws, nblocks, 2 * hh, dm); \
        GATMI_LAUNCH_CHECK("bnf_dm_reduce");                                               \
        hipLaunchKernelGGL(bnf_group_sum_kernel, dim3((unsigned)std::min<int64_t>(1024, (nodes * hh + 255) / 256), groups), dim3(256), 0, st, df1p, jb, nodes * hh, 0, df1); \
        GATMI_LAUNCH_CHECK("bnf_df1_sum");                                                 \
        if (groups > 1) {                                                                  \
            hipLaunchKernelGGL(bnf_group_sum_kernel, dim3((unsigned)std::min<int64_t>(2048, (nodes * fp + 255) / 256)), dim3(256), 0, st, a.dX, groups, nodes * fp, accumulate, dX); \
            hipLaunchKernelGGL(bnf_group_sum_kernel, dim3((unsigned)std::min<int64_t>(2048, (pairs + 255) / 256)), dim3(256), 0, st, a.dA, groups, pairs, accumulate, dA); \
            GATMI_LAUNCH_CHECK("bnf_group_sum");                                           \
        }                                                                                  \
        return GATMI_OK;                                                                   \
    }
    GATMI_BNF_SHAPES(X_)
#undef X_
    return fail(GATMI_EUNSUPPORTED, "bnf_attn_bwd: no kernel for fp=%d H=%d", fp, H);
}
