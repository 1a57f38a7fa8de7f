// Batched dense masked graph attention for gfx950 (small graphs, [B, N, N] additive mask).
//
// Forward, per graph b, head h, destination row i:
//     alpha_ij = softmax_j( LeakyReLU(f1[i,h] + f2[j,h]) + bias_mat[b,i,j] ),  out_i = sum_j alpha_ij Wh_j
// Both contractions over j run on the exact-fp32 matrix instruction v_mfma_f32_16x16x4_f32.
// Lane l of a wave supplies A[row l&15][k l>>4] and B[k l>>4][col l&15]; a wave owns 16
// destination rows, so lane (i = l&15, kq = l>>4) computes the softmax numerator of row i for
// the sources j = j0 + 4*kq + t itself and feeds it straight in as the A operand — the N x N
// coefficient matrix never leaves registers.  The B operand is the source tile of Wh[:, head]
// with one extra column of ones, which makes the row sum (softmax denominator) fall out of the
// same product.  The row maximum is found in a first sweep over the mask (no exp, no MFMA).
//
// Backward, per graph b, head h, source tile j: dalpha = g . Wh^T (MFMA, K = dp) lands in the
// C layout (rows i = 4*kq+r, column j = l&15); alpha is recomputed in that layout, and the C
// layout of a 16x16 tile is exactly the A layout of its transpose taken four rows at a time, so
// dWh_j += alpha^T . g needs no data movement either.  df1 comes from the positive-part sums
// saved by the forward pass (see gat_sparse.hip), df2 from column sums held in registers.
#include <hip/hip_runtime.h>

#include <algorithm>

#include "common.h"
#include "device_common.h"

namespace gatmi {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int kJC = 64;  // sources staged in LDS per step

struct DenseFwdArgs {
    int N;
    const float *Wh, *f1, *f2, *bias_mat, *bias, *resid;
    float *out, *agg, *pos, *lse, *q;
    int activation;
    float slope;
};

template <int H, int DP, bool SAVE>
__global__ __launch_bounds__(256) void dense_attn_fwd_kernel(DenseFwdArgs a) {
    constexpr int HD = H * DP;
    constexpr int LD = HD + 4;           // 4*LD mod 32 == 16: the two k-rows of a 32-lane half hit disjoint banks
    constexpr int NTH = DP / 16 + 1;     // column tiles per head: dp value columns + the ones column
    __shared__ float vs[kJC * LD];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int m = lane & 15, kq = lane >> 4;
    const int N = a.N;
    const int64_t b = blockIdx.x;
    const int i0 = blockIdx.y * 64 + wave * 16;
    const int irow = min(i0 + m, N - 1);             // clamped: rows >= N compute garbage that is never stored
    const float *bm = a.bias_mat + (b * N + irow) * (int64_t)N;
    const float *f2b = a.f2 + b * N * (int64_t)H;
    const bool vec = (N & 3) == 0;

    float f1v[H], mx[H];
#pragma unroll
    for (int h = 0; h < H; ++h) {
        f1v[h] = a.f1[(b * N + irow) * H + h];
        mx[h] = -INFINITY;
    }
    // ---- sweep 1: row maxima --------------------------------------------------------------
    for (int j0 = 0; j0 < N; j0 += 16) {
        const int j = j0 + 4 * kq;
        float bv[4];
        if (vec && j < N) {
            const float4 t = *reinterpret_cast<const float4 *>(bm + j);
            bv[0] = t.x; bv[1] = t.y; bv[2] = t.z; bv[3] = t.w;
        } else {
#pragma unroll
            for (int t = 0; t < 4; ++t) bv[t] = (j + t < N) ? bm[j + t] : -INFINITY;
        }
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int jj = min(j + t, N - 1);
#pragma unroll
            for (int h = 0; h < H; ++h) mx[h] = fmaxf(mx[h], leaky(f1v[h] + f2b[jj * H + h], a.slope) + bv[t]);
        }
    }
#pragma unroll
    for (int h = 0; h < H; ++h) {
        mx[h] = fmaxf(mx[h], __shfl_xor(mx[h], 16, 64));
        mx[h] = fmaxf(mx[h], __shfl_xor(mx[h], 32, 64));
    }
    // ---- sweep 2: numerators into the matrix pipe -------------------------------------------
    f32x4 acc[H][NTH], accp[SAVE ? H : 1][NTH];
#pragma unroll
    for (int h = 0; h < H; ++h)
#pragma unroll
        for (int nt = 0; nt < NTH; ++nt) {
            acc[h][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
            if constexpr (SAVE) accp[h][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
        }
    for (int jc = 0; jc < N; jc += kJC) {
        __syncthreads();
        for (int e = threadIdx.x; e < kJC * (HD / 4); e += 256) {
            const int r = e / (HD / 4), c = e % (HD / 4);
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (jc + r < N) v = *reinterpret_cast<const float4 *>(a.Wh + (b * N + jc + r) * (int64_t)HD + 4 * c);
            *reinterpret_cast<float4 *>(vs + r * LD + 4 * c) = v;
        }
        __syncthreads();
        const int jend = min(kJC, N - jc);
        for (int jl = 0; jl < jend; jl += 16) {
            const int j = jc + jl + 4 * kq;
            float bv[4];
            if (vec && j < N) {
                const float4 t = *reinterpret_cast<const float4 *>(bm + j);
                bv[0] = t.x; bv[1] = t.y; bv[2] = t.z; bv[3] = t.w;
            } else {
#pragma unroll
                for (int t = 0; t < 4; ++t) bv[t] = (j + t < N) ? bm[j + t] : -INFINITY;
            }
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const int jj = min(j + t, N - 1);
                const float *vrow = vs + (jl + 4 * kq + t) * LD;
#pragma unroll
                for (int h = 0; h < H; ++h) {
                    const float z = f1v[h] + f2b[jj * H + h];
                    const float p = exp_neg(leaky(z, a.slope) + bv[t] - mx[h]);   // 0 for j >= N (bias -inf)
                    const float pp = z > 0.f ? p : 0.f;
#pragma unroll
                    for (int nt = 0; nt < NTH; ++nt) {
                        const int c = 16 * nt + m;
                        const float bval = c < DP ? vrow[h * DP + c] : (c == DP ? 1.f : 0.f);
                        acc[h][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(p, bval, acc[h][nt], 0, 0, 0);
                        if constexpr (SAVE) accp[h][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(pp, bval, accp[h][nt], 0, 0, 0);
                    }
                }
            }
        }
    }
    // ---- epilogue: C layout, lane (kq, c = m) holds rows 4*kq + r, column c of each tile --------
    constexpr int NT_ONES = DP / 16, C_ONES = DP % 16;   // tile / column of the ones column
#pragma unroll
    for (int h = 0; h < H; ++h) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int i = i0 + 4 * kq + r;
            const float s = __shfl(acc[h][NT_ONES][r], (lane & 48) | C_ONES, 64);
            const float mrow = __shfl(mx[h], 4 * kq + r, 64);
            const float inv = s > 0.f ? 1.0f / s : 0.f;
            float qv = 0.f;
            if constexpr (SAVE) qv = __shfl(accp[h][NT_ONES][r], (lane & 48) | C_ONES, 64) * inv;
            if (i < N) {
                const int64_t node = b * N + i;
                if constexpr (SAVE) {
                    if (m == 0) {
                        a.lse[node * H + h] = s > 0.f ? mrow + logf(s) : 0.f;
                        a.q[node * H + h] = qv;
                    }
                }
#pragma unroll
                for (int nt = 0; nt < NTH; ++nt) {
                    const int c = 16 * nt + m;
                    if (c < DP) {
                        const int64_t o = node * HD + h * DP + c;
                        const float ag = acc[h][nt][r] * inv;
                        if constexpr (SAVE) {
                            a.agg[o] = ag;
                            a.pos[o] = accp[h][nt][r] * inv;
                        }
                        float v = ag;
                        if (a.bias) v += a.bias[h * DP + c];
                        if (a.resid) v += a.resid[o];
                        if (a.activation == GATMI_ACT_ELU) v = elu_f(v);
                        a.out[o] = v;
                    }
                }
            }
        }
    }
}

struct DenseBwdArgs {
    int N;
    const float *g, *stats, *Wh, *f2, *bias_mat, *df1, *a1, *a2;
    float *dWh, *df2;
    float slope;
};

// Block (b, 64-source tile): wave w owns sources j0 .. j0+15 and walks all destination tiles.
template <int H, int DP>
__global__ __launch_bounds__(256) void dense_attn_bwd_kernel(DenseBwdArgs a) {
    constexpr int HD = H * DP;
    constexpr int LD = HD + 4;
    constexpr int NTH = (DP + 15) / 16;  // column tiles of dWh per head
    constexpr int KS = DP / 4;           // k-steps of the g . Wh^T product
    __shared__ float gs[kJC * LD];            // g rows of the current destination chunk
    __shared__ float ss[kJC * 3 * H];         // their {f1, lse, c}
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int m = lane & 15, kq = lane >> 4;
    const int N = a.N;
    const int64_t b = blockIdx.x;
    const int j0 = blockIdx.y * 64 + wave * 16;
    const int jcol = min(j0 + m, N - 1);

    float f2v[H], df2[H], whb[H][KS];
    f32x4 acc[H][NTH];
#pragma unroll
    for (int h = 0; h < H; ++h) {
        f2v[h] = a.f2[(b * N + jcol) * H + h];
        df2[h] = 0.f;
#pragma unroll
        for (int s = 0; s < KS; ++s) whb[h][s] = a.Wh[(b * N + jcol) * (int64_t)HD + h * DP + 4 * s + kq];
#pragma unroll
        for (int nt = 0; nt < NTH; ++nt) acc[h][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
    }
    const bool jok = j0 + m < N;
    for (int ic = 0; ic < N; ic += kJC) {
        __syncthreads();
        for (int e = threadIdx.x; e < kJC * (HD / 4); e += 256) {
            const int r = e / (HD / 4), c = e % (HD / 4);
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (ic + r < N) v = *reinterpret_cast<const float4 *>(a.g + (b * N + ic + r) * (int64_t)HD + 4 * c);
            *reinterpret_cast<float4 *>(gs + r * LD + 4 * c) = v;
        }
        for (int e = threadIdx.x; e < kJC * 3 * H; e += 256) {
            const int r = e / (3 * H);
            ss[e] = (ic + r < N) ? a.stats[(b * N + ic) * (int64_t)(3 * H) + e] : 0.f;
        }
        __syncthreads();
        const int iend = min(kJC, N - ic);
        for (int il = 0; il < iend; il += 16) {
            // mask values of this 16x16 tile in the C layout: rows il+4kq+r, column jcol
            float bv[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int i = ic + il + 4 * kq + r;
                bv[r] = (i < N && jok) ? a.bias_mat[(b * N + i) * (int64_t)N + jcol] : -INFINITY;
            }
#pragma unroll
            for (int h = 0; h < H; ++h) {
                f32x4 da = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int s = 0; s < KS; ++s)
                    da = __builtin_amdgcn_mfma_f32_16x16x4f32(gs[(il + m) * LD + h * DP + 4 * s + kq], whb[h][s], da, 0, 0, 0);
                float al[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float *st = ss + (il + 4 * kq + r) * 3 * H;
                    const float z = st[h] + f2v[h];
                    const float alpha = exp_neg(leaky(z, a.slope) + bv[r] - st[H + h]);   // 0 outside the graph
                    al[r] = alpha;
                    df2[h] += alpha * (da[r] - st[2 * H + h]) * (z > 0.f ? 1.f : a.slope);
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float *grow = gs + (il + 4 * kq + r) * LD + h * DP;
#pragma unroll
                    for (int nt = 0; nt < NTH; ++nt) {
                        const int c = 16 * nt + m;
                        const float bval = c < DP ? grow[c] : 0.f;
                        acc[h][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(al[r], bval, acc[h][nt], 0, 0, 0);
                    }
                }
            }
        }
    }
#pragma unroll
    for (int h = 0; h < H; ++h) {
        float t = df2[h];
        t += __shfl_xor(t, 16, 64);
        t += __shfl_xor(t, 32, 64);
        if (kq == 0 && jok) a.df2[(b * N + j0 + m) * H + h] = t;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int j = j0 + 4 * kq + r;
            const float d2 = __shfl(t, 4 * kq + r, 64);   // column total of source j
            if (j >= N) continue;
            const float d1 = a.a1 ? a.df1[(b * N + j) * H + h] : 0.f;
#pragma unroll
            for (int nt = 0; nt < NTH; ++nt) {
                const int c = 16 * nt + m;
                if (c < DP) {
                    float v = acc[h][nt][r];
                    if (a.a1) v += d1 * a.a1[h * DP + c] + d2 * a.a2[h * DP + c];
                    a.dWh[(b * N + j) * (int64_t)HD + h * DP + c] = v;
                }
            }
        }
    }
}

#define GATMI_DENSE_SHAPES(X) X(8, 8) X(4, 8) X(2, 8) X(1, 8) X(1, 4) X(2, 4) X(4, 4) X(8, 4) X(1, 16) X(2, 16) X(4, 16) X(1, 32) X(2, 32) X(1, 64)

}  // namespace gatmi

using namespace gatmi;

extern "C" int gatmi_dense_attn_fwd(int64_t B, int32_t N, int32_t H, int32_t dp,
                                    const float *Wh, const float *f1, const float *f2, const float *bias_mat,
                                    const float *bias, const float *resid, int32_t activation, float slope,
                                    float *out, float *agg, float *pos, float *lse, float *q, void *stream) {
    int G, LPH;
    int rc = check_shape(H, dp, &G, &LPH);
    if (rc) return rc;
    GATMI_REQUIRE(B >= 0 && N >= 0, GATMI_EINVAL, "dense_attn_fwd: negative size");
    if (B == 0 || N == 0) return GATMI_OK;
    GATMI_REQUIRE(B <= 0x7fffffff, GATMI_ERANGE, "dense_attn_fwd: batch %lld too large for one launch", (long long)B);
    GATMI_REQUIRE(Wh && f1 && f2 && bias_mat && out, GATMI_EINVAL, "dense_attn_fwd: null pointer");
    GATMI_REQUIRE(activation == GATMI_ACT_IDENTITY || activation == GATMI_ACT_ELU, GATMI_EUNSUPPORTED, "dense_attn_fwd: activation %d", activation);
    const bool save = agg || pos || lse || q;
    GATMI_REQUIRE(!save || (agg && pos && lse && q), GATMI_EINVAL, "dense_attn_fwd: agg, pos, lse, q must be given together");
    GATMI_REQUIRE_ALIGNED(Wh); GATMI_REQUIRE_ALIGNED(bias_mat);
    DenseFwdArgs a;
    a.N = N; a.Wh = Wh; a.f1 = f1; a.f2 = f2; a.bias_mat = bias_mat; a.bias = bias; a.resid = resid;
    a.out = out; a.agg = agg; a.pos = pos; a.lse = lse; a.q = q; a.activation = activation; a.slope = slope;
    const dim3 grid((unsigned)B, (unsigned)((N + 63) / 64));
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
#define X(hh, dd)                                                                                             \
    if (H == hh && dp == dd) {                                                                                \
        if (save) hipLaunchKernelGGL((dense_attn_fwd_kernel<hh, dd, true>), grid, dim3(256), 0, st, a);        \
        else hipLaunchKernelGGL((dense_attn_fwd_kernel<hh, dd, false>), grid, dim3(256), 0, st, a);            \
        GATMI_LAUNCH_CHECK("dense_attn_fwd");                                                                 \
        return GATMI_OK;                                                                                      \
    }
    GATMI_DENSE_SHAPES(X)
#undef X
    return fail(GATMI_EUNSUPPORTED, "dense_attn_fwd: no kernel for H=%d dp=%d", H, dp);
}

extern "C" int gatmi_dense_attn_bwd(int64_t B, int32_t N, int32_t H, int32_t dp,
                                    const float *g, const float *stats, const float *Wh, const float *f2,
                                    const float *bias_mat, const float *df1, const float *a1, const float *a2,
                                    float slope, float *dWh, float *df2, void *stream) {
    int G, LPH;
    int rc = check_shape(H, dp, &G, &LPH);
    if (rc) return rc;
    GATMI_REQUIRE(B >= 0 && N >= 0, GATMI_EINVAL, "dense_attn_bwd: negative size");
    if (B == 0 || N == 0) return GATMI_OK;
    GATMI_REQUIRE(B <= 0x7fffffff, GATMI_ERANGE, "dense_attn_bwd: batch %lld too large for one launch", (long long)B);
    GATMI_REQUIRE(g && stats && Wh && f2 && bias_mat && dWh && df2, GATMI_EINVAL, "dense_attn_bwd: null pointer");
    GATMI_REQUIRE((a1 == nullptr) == (a2 == nullptr) && (!a1 || df1), GATMI_EINVAL, "dense_attn_bwd: a1, a2 and df1 go together");
    GATMI_REQUIRE_ALIGNED(g);
    DenseBwdArgs a;
    a.df1 = df1; a.a1 = a1; a.a2 = a2;
    a.N = N; a.g = g; a.stats = stats; a.Wh = Wh; a.f2 = f2; a.bias_mat = bias_mat; a.dWh = dWh; a.df2 = df2; a.slope = slope;
    const dim3 grid((unsigned)B, (unsigned)((N + 63) / 64));
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
#define X(hh, dd)                                                                                 \
    if (H == hh && dp == dd) {                                                                    \
        hipLaunchKernelGGL((dense_attn_bwd_kernel<hh, dd>), grid, dim3(256), 0, st, a);            \
        GATMI_LAUNCH_CHECK("dense_attn_bwd");                                                     \
        return GATMI_OK;                                                                          \
    }
    GATMI_DENSE_SHAPES(X)
#undef X
    return fail(GATMI_EUNSUPPORTED, "dense_attn_bwd: no kernel for H=%d dp=%d", H, dp);
}
