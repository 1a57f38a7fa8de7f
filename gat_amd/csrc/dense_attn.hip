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
    int64_t B;
    int N;
    const float *Wh, *f1, *f2, *bias_mat, *bias, *resid;
    float *out, *agg, *pos, *lse, *q;
    int activation;
    float slope, coef_keep, feat_keep;
    uint64_t seed;
    const uint64_t *seed_step;   // device word added to the seed at run time (seed_now), or nullptr
};

// How the four waves of a block split the work: heads first (each wave keeps only HPW heads'
// accumulators, so registers stay low and many blocks fit a CU), 16-row tiles with what is left.
template <int H>
struct WaveSplit {
    static constexpr int WH = H >= 4 ? 4 : H;   // waves along heads
    static constexpr int WT = 4 / WH;           // 16-row (or 16-source) tiles per block
    static constexpr int HPW = H / WH;          // heads per wave
};

__device__ __forceinline__ void load_mask4(const float *bm, int j, int N, bool vec, float (&bv)[4]) {
    if (vec && j < N) {
        const float4 t = *reinterpret_cast<const float4 *>(bm + j);
        bv[0] = t.x; bv[1] = t.y; bv[2] = t.z; bv[3] = t.w;
    } else {
#pragma unroll
        for (int t = 0; t < 4; ++t) bv[t] = (j + t < N) ? bm[j + t] : -INFINITY;
    }
}

template <int H, int DP, bool SAVE, bool DROP>
__global__ __launch_bounds__(256) void dense_attn_fwd_kernel(DenseFwdArgs a) {
    using S = WaveSplit<H>;
    constexpr int HPW = S::HPW;
    constexpr int HD = H * DP;
    constexpr int LD = HD + 4;           // 4*LD mod 32 == 16: the two k-rows of a 32-lane half hit disjoint banks
    constexpr int NTH = DP / 16 + 1;     // column tiles per head: dp value columns + the ones column
    __shared__ __attribute__((aligned(16))) float vs[kJC * LD];   // Wh rows of the current source chunk
    __shared__ __attribute__((aligned(16))) float f2s[kJC * H];   // their f2
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int m = lane & 15, kq = lane >> 4;
    const int h0 = (wave % S::WH) * HPW;
    const int N = a.N;
    const int64_t b = blockIdx.x;
    const int i0 = (blockIdx.y * S::WT + wave / S::WH) * 16;
    const int irow = min(i0 + m, N - 1);             // clamped: rows >= N compute values that are never stored
    const float *bm = a.bias_mat + (b * N + irow) * (int64_t)N;
    const float *f2b = a.f2 + b * N * (int64_t)H;
    const bool vec = (N & 3) == 0;

    float f1v[HPW], mx[HPW];
#pragma unroll
    for (int h = 0; h < HPW; ++h) {
        f1v[h] = a.f1[(b * N + irow) * H + h0 + h];
        mx[h] = -INFINITY;
    }
    // ---- sweep 1: row maxima (no exp, no MFMA) ---------------------------------------------
    for (int jc = 0; jc < N; jc += kJC) {
        __syncthreads();
        for (int e = threadIdx.x; e < kJC * H; e += 256) f2s[e] = (jc + e / H < N) ? f2b[(int64_t)jc * H + e] : 0.f;
        __syncthreads();
        if (i0 < N) {
            const int jend = min(kJC, N - jc);
            for (int jl = 0; jl < jend; jl += 16) {
                float bv[4];
                load_mask4(bm, jc + jl + 4 * kq, N, vec, bv);
#pragma unroll
                for (int t = 0; t < 4; ++t)
#pragma unroll
                    for (int h = 0; h < HPW; ++h)
                        mx[h] = fmaxf(mx[h], leaky(f1v[h] + f2s[(jl + 4 * kq + t) * H + h0 + h], a.slope) + bv[t]);
            }
        }
    }
#pragma unroll
    for (int h = 0; h < HPW; ++h) {
        mx[h] = fmaxf(mx[h], __shfl_xor(mx[h], 16, 64));
        mx[h] = fmaxf(mx[h], __shfl_xor(mx[h], 32, 64));
    }
    // ---- sweep 2: numerators into the matrix pipe -------------------------------------------
    f32x4 acc[HPW][NTH], accp[SAVE ? HPW : 1][NTH];
#pragma unroll
    for (int h = 0; h < HPW; ++h)
#pragma unroll
        for (int nt = 0; nt < NTH; ++nt) {
            acc[h][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
            if constexpr (SAVE) accp[h][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
        }
    float ssum[HPW], qsum[HPW];            // dropout only: denominators from the UNdropped numerators
#pragma unroll
    for (int h = 0; h < HPW; ++h) ssum[h] = qsum[h] = 0.f;
    for (int jc = 0; jc < N; jc += kJC) {
        __syncthreads();
        for (int e = threadIdx.x; e < kJC * (HD / 4); e += 256) {
            const int r = e / (HD / 4), c = e % (HD / 4);
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (jc + r < N) v = *reinterpret_cast<const float4 *>(a.Wh + (b * N + jc + r) * (int64_t)HD + 4 * c);
            if constexpr (DROP) {
                if (a.feat_keep < 1.f) v = drop_feat(v, seed_now(a), b * N + jc + r, HD, 4 * c, a.feat_keep);
            }
            *reinterpret_cast<float4 *>(vs + r * LD + 4 * c) = v;
        }
        for (int e = threadIdx.x; e < kJC * H; e += 256) f2s[e] = (jc + e / H < N) ? f2b[(int64_t)jc * H + e] : 0.f;
        __syncthreads();
        if (i0 >= N) continue;
        const int jend = min(kJC, N - jc);
        for (int jl = 0; jl < jend; jl += 16) {
            float bv[4];
            load_mask4(bm, jc + jl + 4 * kq, N, vec, bv);
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const float *vrow = vs + (jl + 4 * kq + t) * LD;
#pragma unroll
                for (int h = 0; h < HPW; ++h) {
                    const float z = f1v[h] + f2s[(jl + 4 * kq + t) * H + h0 + h];
                    const float p0 = exp_neg(leaky(z, a.slope) + bv[t] - mx[h]);   // 0 for j >= N (mask -inf)
                    float p = p0;
                    if constexpr (DROP) {
                        ssum[h] += p0;
                        qsum[h] += z > 0.f ? p0 : 0.f;
                        if (a.coef_keep < 1.f) p = p0 * coef_drop_scale(seed_now(a), b, N, H, irow, min(jc + jl + 4 * kq + t, N - 1), h0 + h, a.coef_keep);
                    }
                    const float pp = z > 0.f ? p : 0.f;
#pragma unroll
                    for (int nt = 0; nt < NTH; ++nt) {
                        const int c = 16 * nt + m;
                        const float bval = c < DP ? vrow[(h0 + h) * DP + c] : (c == DP ? 1.f : 0.f);
                        acc[h][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(p, bval, acc[h][nt], 0, 0, 0);
                        if constexpr (SAVE) accp[h][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(pp, bval, accp[h][nt], 0, 0, 0);
                    }
                }
            }
        }
    }
    if (i0 >= N) return;
    if constexpr (DROP) {
#pragma unroll
        for (int h = 0; h < HPW; ++h) {
            ssum[h] += __shfl_xor(ssum[h], 16, 64); ssum[h] += __shfl_xor(ssum[h], 32, 64);
            qsum[h] += __shfl_xor(qsum[h], 16, 64); qsum[h] += __shfl_xor(qsum[h], 32, 64);
        }
    }
    // ---- epilogue: C layout, lane (kq, c = m) holds rows 4*kq + r, column c of each tile --------
    constexpr int NT_ONES = DP / 16, C_ONES = DP % 16;   // tile / column of the ones column
#pragma unroll
    for (int h = 0; h < HPW; ++h) {
        const int hg = h0 + h;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int i = i0 + 4 * kq + r;
            const float s = DROP ? __shfl(ssum[h], 4 * kq + r, 64) : __shfl(acc[h][NT_ONES][r], (lane & 48) | C_ONES, 64);
            const float mrow = __shfl(mx[h], 4 * kq + r, 64);
            const float inv = s > 0.f ? 1.0f / s : 0.f;
            float qv = 0.f;
            if constexpr (SAVE) qv = (DROP ? __shfl(qsum[h], 4 * kq + r, 64) : __shfl(accp[h][NT_ONES][r], (lane & 48) | C_ONES, 64)) * inv;
            if (i < N) {
                const int64_t node = b * N + i;
                if constexpr (SAVE) {
                    if (m == 0) {
                        a.lse[node * H + hg] = s > 0.f ? mrow + logf(s) : 0.f;
                        a.q[node * H + hg] = qv;
                    }
                }
#pragma unroll
                for (int nt = 0; nt < NTH; ++nt) {
                    const int c = 16 * nt + m;
                    if (c < DP) {
                        const int64_t o = node * HD + hg * DP + c;
                        const float ag = acc[h][nt][r] * inv;
                        if constexpr (SAVE) {
                            a.agg[o] = ag;
                            a.pos[o] = accp[h][nt][r] * inv;
                        }
                        float v = ag;
                        if (a.bias) v += a.bias[hg * DP + c];
                        if (a.resid) v += a.resid[o];
                        v = act_f(v, a.activation);
                        a.out[o] = v;
                    }
                }
            }
        }
    }
}

struct DenseBwdArgs {
    int64_t B;
    int N;
    const float *g, *stats, *Wh, *f2, *bias_mat, *df1, *a1, *a2;
    float *dWh, *df2;
    float slope, coef_keep, feat_keep;
    uint64_t seed;
    const uint64_t *seed_step;   // device word added to the seed at run time (seed_now), or nullptr
};

// Block (b, source tiles): a wave owns 16 sources x HPW heads and walks all destination tiles.
template <int H, int DP, bool DROP>
__global__ __launch_bounds__(256) void dense_attn_bwd_kernel(DenseBwdArgs a) {
    using S = WaveSplit<H>;
    constexpr int HPW = S::HPW;
    constexpr int HD = H * DP;
    constexpr int LD = HD + 4;
    constexpr int NTH = (DP + 15) / 16;  // column tiles of dWh per head
    constexpr int KS = DP / 4;           // k-steps of the g . Wh^T product
    __shared__ __attribute__((aligned(16))) float gs[kJC * LD];       // g rows of the current destination chunk
    __shared__ __attribute__((aligned(16))) float ss[kJC * 3 * H];    // their {f1, lse, c}
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int m = lane & 15, kq = lane >> 4;
    const int h0 = (wave % S::WH) * HPW;
    const int N = a.N;
    const int64_t b = blockIdx.x;
    const int j0 = (blockIdx.y * S::WT + wave / S::WH) * 16;
    const int jcol = min(j0 + m, N - 1);
    const bool jok = j0 + m < N;

    float f2v[HPW], df2[HPW], whb[HPW][KS];
    f32x4 acc[HPW][NTH];
#pragma unroll
    for (int h = 0; h < HPW; ++h) {
        f2v[h] = a.f2[(b * N + jcol) * H + h0 + h];
        df2[h] = 0.f;
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            whb[h][s] = a.Wh[(b * N + jcol) * (int64_t)HD + (h0 + h) * DP + 4 * s + kq];
            if constexpr (DROP) {
                if (a.feat_keep < 1.f) whb[h][s] = drop_feat1(whb[h][s], seed_now(a), b * N + jcol, HD, (h0 + h) * DP + 4 * s + kq, a.feat_keep);
            }
        }
#pragma unroll
        for (int nt = 0; nt < NTH; ++nt) acc[h][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
    }
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
        if (j0 >= N) continue;
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
            for (int h = 0; h < HPW; ++h) {
                const int hg = h0 + h;
                f32x4 da = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int s = 0; s < KS; ++s)
                    da = __builtin_amdgcn_mfma_f32_16x16x4f32(gs[(il + m) * LD + hg * DP + 4 * s + kq], whb[h][s], da, 0, 0, 0);
                float al[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float *st = ss + (il + 4 * kq + r) * 3 * H;
                    const float z = st[hg] + f2v[h];
                    const float alpha = exp_neg(leaky(z, a.slope) + bv[r] - st[H + hg]);   // 0 outside the graph
                    float mk = 1.f;
                    if constexpr (DROP) {
                        if (a.coef_keep < 1.f) mk = coef_drop_scale(seed_now(a), b, N, H, min(ic + il + 4 * kq + r, N - 1), jcol, hg, a.coef_keep);
                    }
                    al[r] = alpha * mk;
                    df2[h] += alpha * (da[r] * mk - st[2 * H + hg]) * (z > 0.f ? 1.f : a.slope);
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float *grow = gs + (il + 4 * kq + r) * LD + hg * DP;
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
    if (j0 >= N) return;
#pragma unroll
    for (int h = 0; h < HPW; ++h) {
        const int hg = h0 + h;
        float t = df2[h];
        t += __shfl_xor(t, 16, 64);
        t += __shfl_xor(t, 32, 64);
        if (kq == 0 && jok) a.df2[(b * N + j0 + m) * H + hg] = t;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int j = j0 + 4 * kq + r;
            const float d2 = __shfl(t, 4 * kq + r, 64);   // column total of source j
            if (j >= N) continue;
            const float d1 = a.a1 ? a.df1[(b * N + j) * H + hg] : 0.f;
#pragma unroll
            for (int nt = 0; nt < NTH; ++nt) {
                const int c = 16 * nt + m;
                if (c < DP) {
                    float v = acc[h][nt][r];
                    if constexpr (DROP) {
                        if (a.feat_keep < 1.f) v = drop_feat1(v, seed_now(a), b * N + j, HD, hg * DP + c, a.feat_keep);
                    }
                    if (a.a1) v += d1 * a.a1[hg * DP + c] + d2 * a.a2[hg * DP + c];
                    a.dWh[(b * N + j) * (int64_t)HD + hg * DP + c] = v;
                }
            }
        }
    }
}


// ------------------------------------------------------------------------------------------
// Small-graph variants: one block per (graph, head group).  The block keeps the head group's
// slice of Wh (or g) for ALL nodes of the graph in LDS, staged once, so there is no barrier in
// the sweeps, little LDS per block (many blocks per CU) and few registers per wave.  Each of the
// four waves walks 16-row tiles i0 = 16*wave, +64, ...  Used when the slice fits 64 KiB of LDS;
// larger graphs take the chunked kernels above.
// ------------------------------------------------------------------------------------------
// Block id -> (graph, head group) for the small-graph kernels.  Workgroups are dealt round-robin
// over the 8 XCDs, so ids that are equal mod 8 share an L2: the head groups of one graph get ids
// 8 apart and re-read the graph's N x N mask from that L2 instead of from HBM.  (Placement is a
// speed matter only; nothing depends on it.)  Launch ceil(B/8)*8*WH blocks; ids past B exit.
__device__ __forceinline__ void graph_and_head_group(int WH, int64_t &graph, int &hg) {
    const int64_t L = blockIdx.x;
    hg = (int)((L >> 3) % WH);
    graph = (L / (8 * WH)) * 8 + (L & 7);
}

constexpr int kMaskBatch = 6;      // 16-source steps whose mask loads are issued together (forward)
constexpr int kMaskBatchBwd = 1;   // 16-destination steps whose mask loads are issued together (backward)

template <int H, int DP, bool SAVE, bool DROP>
__global__ __launch_bounds__(256, 5) void dense_attn_fwd_hg_kernel(DenseFwdArgs a) {      // five waves per SIMD (at most 96 registers): 2.45 -> 2.24 ms at 4096 x 264
    using S = WaveSplit<H>;
    constexpr int HPW = S::HPW;
    constexpr int HD = H * DP;
    constexpr int W = HPW * DP;          // columns of this head group
    constexpr int LDV = W + 4;           // 4*LDV mod 32 == 16 for W = 16, 32, 64: the k-rows of a half-wave sit on disjoint banks
    constexpr int NTV = DP / 16;         // column tiles made of value columns only
    constexpr int TAIL = DP % 16;        // value columns in the last tile (dp = 4 or 8), 0 otherwise
    constexpr int NTH = NTV + 1;         // + the tile that carries the ones column
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int N = a.N, Np = (N + 15) & ~15;
    float *vs = smem;                    // [Np][LDV]
    float *f2s = smem + Np * LDV;        // [Np][HPW]
    int64_t b;
    int hgi;
    graph_and_head_group(S::WH, b, hgi);
    if (b >= a.B) return;
    const int h0 = hgi * HPW;
    for (int e = threadIdx.x; e < Np * (W / 4); e += 256) {
        const int r = e / (W / 4), c = e % (W / 4);
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (r < N) v = *reinterpret_cast<const float4 *>(a.Wh + (b * N + r) * (int64_t)HD + h0 * DP + 4 * c);
        if constexpr (DROP) {
            if (a.feat_keep < 1.f) v = drop_feat(v, seed_now(a), b * N + r, HD, h0 * DP + 4 * c, a.feat_keep);
        }
        *reinterpret_cast<float4 *>(vs + r * LDV + 4 * c) = v;
    }
    for (int e = threadIdx.x; e < Np * HPW; e += 256) {
        const int r = e / HPW, h = e % HPW;
        f2s[e] = r < N ? a.f2[(b * N + r) * H + h0 + h] : 0.f;
    }
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int m = lane & 15, kq = lane >> 4;
    const bool vec = (N & 3) == 0;
    const float maskf = (TAIL && m < TAIL) ? 1.f : 0.f, onesf = (m == TAIL) ? 1.f : 0.f;
    const int mcl = (TAIL && m < TAIL) ? m : 0;
    constexpr int NT_ONES = NTV, C_ONES = TAIL;

    for (int i0 = wave * 16; i0 < N; i0 += 64) {
        const int irow = min(i0 + m, N - 1);
        const float *bm = a.bias_mat + (b * N + irow) * (int64_t)N;
        float f1v[HPW], mx[HPW];
#pragma unroll
        for (int h = 0; h < HPW; ++h) {
            f1v[h] = a.f1[(b * N + irow) * H + h0 + h];
            mx[h] = -INFINITY;
        }
        for (int jg = 0; jg < Np; jg += 16 * kMaskBatch) {         // sweep 1: row maxima
            float bvv[kMaskBatch][4];
#pragma unroll
            for (int u = 0; u < kMaskBatch; ++u) load_mask4(bm, jg + 16 * u + 4 * kq, N, vec, bvv[u]);   // all loads first
#pragma unroll
            for (int u = 0; u < kMaskBatch; ++u) {
                const int j0 = jg + 16 * u;
                if (j0 < Np) {
                    float f2r[4][HPW];
#pragma unroll
                    for (int t = 0; t < 4; ++t)
#pragma unroll
                        for (int h = 0; h < HPW; ++h) f2r[t][h] = f2s[(j0 + 4 * kq + t) * HPW + h];
#pragma unroll
                    for (int t = 0; t < 4; ++t)
#pragma unroll
                        for (int h = 0; h < HPW; ++h) mx[h] = fmaxf(mx[h], leaky(f1v[h] + f2r[t][h], a.slope) + bvv[u][t]);
                }
            }
        }
#pragma unroll
        for (int h = 0; h < HPW; ++h) {
            mx[h] = fmaxf(mx[h], __shfl_xor(mx[h], 16, 64));
            mx[h] = fmaxf(mx[h], __shfl_xor(mx[h], 32, 64));
        }
        f32x4 acc[HPW][NTH], accp[SAVE ? HPW : 1][NTH];
#pragma unroll
        for (int h = 0; h < HPW; ++h)
#pragma unroll
            for (int nt = 0; nt < NTH; ++nt) {
                acc[h][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
                if constexpr (SAVE) accp[h][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
            }
        float ssum[HPW], qsum[HPW];        // dropout only: denominators from the UNdropped numerators
#pragma unroll
        for (int h = 0; h < HPW; ++h) ssum[h] = qsum[h] = 0.f;
        for (int jg = 0; jg < Np; jg += 16 * kMaskBatch) {         // sweep 2: numerators -> matrix pipe
          float bvv[kMaskBatch][4];
#pragma unroll
          for (int u = 0; u < kMaskBatch; ++u) load_mask4(bm, jg + 16 * u + 4 * kq, N, vec, bvv[u]);
#pragma unroll
          for (int u = 0; u < kMaskBatch; ++u) {
            const int j0 = jg + 16 * u;
            if (j0 >= Np) break;
            const float (&bv)[4] = bvv[u];
            float f2r[4][HPW], vr[4][HPW][NTH];
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int h = 0; h < HPW; ++h) {
                    f2r[t][h] = f2s[(j0 + 4 * kq + t) * HPW + h];
                    const float *row = vs + (j0 + 4 * kq + t) * LDV + h * DP;
#pragma unroll
                    for (int nt = 0; nt < NTV; ++nt) vr[t][h][nt] = row[16 * nt + m];
                    vr[t][h][NTV] = TAIL ? fmaf(row[NTV * 16 + mcl], maskf, onesf) : onesf;
                }
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int h = 0; h < HPW; ++h) {
                    const float z = f1v[h] + f2r[t][h];
                    const float p0 = exp_neg(leaky(z, a.slope) + bv[t] - mx[h]);   // 0 for j >= N (mask -inf)
                    float p = p0;
                    if constexpr (DROP) {
                        ssum[h] += p0;
                        qsum[h] += z > 0.f ? p0 : 0.f;
                        if (a.coef_keep < 1.f) p = p0 * coef_drop_scale(seed_now(a), b, N, H, irow, min(j0 + 4 * kq + t, N - 1), h0 + h, a.coef_keep);
                    }
                    const float pp = z > 0.f ? p : 0.f;
#pragma unroll
                    for (int nt = 0; nt < NTH; ++nt) {
                        acc[h][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(p, vr[t][h][nt], acc[h][nt], 0, 0, 0);
                        if constexpr (SAVE) accp[h][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(pp, vr[t][h][nt], accp[h][nt], 0, 0, 0);
                    }
                }
          }
        }
        if constexpr (DROP) {
#pragma unroll
            for (int h = 0; h < HPW; ++h) {
                ssum[h] += __shfl_xor(ssum[h], 16, 64); ssum[h] += __shfl_xor(ssum[h], 32, 64);
                qsum[h] += __shfl_xor(qsum[h], 16, 64); qsum[h] += __shfl_xor(qsum[h], 32, 64);
            }
        }
#pragma unroll
        for (int h = 0; h < HPW; ++h) {
            const int hg = h0 + h;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int i = i0 + 4 * kq + r;
                const float s = DROP ? __shfl(ssum[h], 4 * kq + r, 64) : __shfl(acc[h][NT_ONES][r], (lane & 48) | C_ONES, 64);
                const float mrow = __shfl(mx[h], 4 * kq + r, 64);
                const float inv = s > 0.f ? 1.0f / s : 0.f;
                float qv = 0.f;
                if constexpr (SAVE) qv = (DROP ? __shfl(qsum[h], 4 * kq + r, 64) : __shfl(accp[h][NT_ONES][r], (lane & 48) | C_ONES, 64)) * inv;
                if (i < N) {
                    const int64_t node = b * N + i;
                    if constexpr (SAVE) {
                        if (m == 0) {
                            a.lse[node * H + hg] = s > 0.f ? mrow + logf(s) : 0.f;
                            a.q[node * H + hg] = qv;
                        }
                    }
#pragma unroll
                    for (int nt = 0; nt < NTH; ++nt) {
                        const int c = 16 * nt + m;
                        if (c < DP) {
                            const int64_t o = node * HD + hg * DP + c;
                            const float ag = acc[h][nt][r] * inv;
                            if constexpr (SAVE) {
                                a.agg[o] = ag;
                                a.pos[o] = accp[h][nt][r] * inv;
                            }
                            float v = ag;
                            if (a.bias) v += a.bias[hg * DP + c];
                            if (a.resid) v += a.resid[o];
                            v = act_f(v, a.activation);
                            a.out[o] = v;
                        }
                    }
                }
            }
        }
    }
}

template <int H, int DP, bool DROP>
__global__ __launch_bounds__(256) void dense_attn_bwd_hg_kernel(DenseBwdArgs a) {
    using S = WaveSplit<H>;
    constexpr int HPW = S::HPW;
    constexpr int HD = H * DP;
    constexpr int W = HPW * DP;
    constexpr int LDG = W + 2;           // even, and (LDG*m + kq) mod 32 distinct over a half-wave: conflict-free A and B reads
    constexpr int NTH = (DP + 15) / 16;
    constexpr int KS = DP / 4;
    constexpr bool NARROW = DP < 16;     // one column tile with dp < 16 value columns
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int N = a.N, Np = (N + 15) & ~15;
    float *ss = smem;                    // [Np][HPW][4] = {f1, lse, c, -}
    float *gs = smem + Np * HPW * 4;     // [Np][LDG]
    int64_t b;
    int hgi;
    graph_and_head_group(S::WH, b, hgi);
    if (b >= a.B) return;
    const int h0 = hgi * HPW;
    for (int e = threadIdx.x; e < Np * (W / 2); e += 256) {
        const int r = e / (W / 2), c = e % (W / 2);
        float2 v = make_float2(0.f, 0.f);
        if (r < N) v = *reinterpret_cast<const float2 *>(a.g + (b * N + r) * (int64_t)HD + h0 * DP + 2 * c);
        *reinterpret_cast<float2 *>(gs + r * LDG + 2 * c) = v;
    }
    for (int e = threadIdx.x; e < Np * HPW; e += 256) {
        const int r = e / HPW, h = e % HPW;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (r < N) {
            const float *st = a.stats + (b * N + r) * (int64_t)(3 * H) + h0 + h;
            v = make_float4(st[0], st[H], st[2 * H], 0.f);
        }
        *reinterpret_cast<float4 *>(ss + e * 4) = v;
    }
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int m = lane & 15, kq = lane >> 4;
    const float maskf = (!NARROW || m < DP) ? 1.f : 0.f;
    const int mcl = (!NARROW || m < DP) ? m : 0;

    for (int j0 = wave * 16; j0 < N; j0 += 64) {
        const int jcol = min(j0 + m, N - 1);
        const bool jok = j0 + m < N;
        float f2v[HPW], df2[HPW], whb[HPW][KS];
        f32x4 acc[HPW][NTH];
#pragma unroll
        for (int h = 0; h < HPW; ++h) {
            f2v[h] = a.f2[(b * N + jcol) * H + h0 + h];
            df2[h] = 0.f;
#pragma unroll
            for (int s = 0; s < KS; ++s) {
            whb[h][s] = a.Wh[(b * N + jcol) * (int64_t)HD + (h0 + h) * DP + 4 * s + kq];
            if constexpr (DROP) {
                if (a.feat_keep < 1.f) whb[h][s] = drop_feat1(whb[h][s], seed_now(a), b * N + jcol, HD, (h0 + h) * DP + 4 * s + kq, a.feat_keep);
            }
        }
#pragma unroll
            for (int nt = 0; nt < NTH; ++nt) acc[h][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
        }
        for (int ig = 0; ig < Np; ig += 16 * kMaskBatchBwd) {
          float bvv[kMaskBatchBwd][4];
#pragma unroll
          for (int u = 0; u < kMaskBatchBwd; ++u)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int i = ig + 16 * u + 4 * kq + r;
                bvv[u][r] = (i < N && jok) ? a.bias_mat[(b * N + i) * (int64_t)N + jcol] : -INFINITY;
            }
#pragma unroll
          for (int u = 0; u < kMaskBatchBwd; ++u) {
            const int il = ig + 16 * u;
            if (il >= Np) break;
            const float (&bv)[4] = bvv[u];
#pragma unroll
            for (int h = 0; h < HPW; ++h) {
                float ga[KS], gb[4][NTH];
                float4 st[4];
#pragma unroll
                for (int s = 0; s < KS; ++s) ga[s] = gs[(il + m) * LDG + h * DP + 4 * s + kq];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    st[r] = *reinterpret_cast<const float4 *>(ss + ((il + 4 * kq + r) * HPW + h) * 4);
                    const float *grow = gs + (il + 4 * kq + r) * LDG + h * DP;
#pragma unroll
                    for (int nt = 0; nt < NTH; ++nt) gb[r][nt] = NARROW ? grow[mcl] * maskf : grow[16 * nt + m];
                }
                f32x4 da = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int s = 0; s < KS; ++s) da = __builtin_amdgcn_mfma_f32_16x16x4f32(ga[s], whb[h][s], da, 0, 0, 0);
                float al[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float z = st[r].x + f2v[h];
                    const float alpha = exp_neg(leaky(z, a.slope) + bv[r] - st[r].y);   // 0 outside the graph
                    float mk = 1.f;
                    if constexpr (DROP) {
                        if (a.coef_keep < 1.f) mk = coef_drop_scale(seed_now(a), b, N, H, min(il + 4 * kq + r, N - 1), jcol, h0 + h, a.coef_keep);
                    }
                    al[r] = alpha * mk;
                    df2[h] += alpha * (da[r] * mk - st[r].z) * (z > 0.f ? 1.f : a.slope);
                }
#pragma unroll
                for (int r = 0; r < 4; ++r)
#pragma unroll
                    for (int nt = 0; nt < NTH; ++nt)
                        acc[h][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(al[r], gb[r][nt], acc[h][nt], 0, 0, 0);
            }
          }
        }
#pragma unroll
        for (int h = 0; h < HPW; ++h) {
            const int hg = h0 + h;
            float t = df2[h];
            t += __shfl_xor(t, 16, 64);
            t += __shfl_xor(t, 32, 64);
            if (kq == 0 && jok) a.df2[(b * N + j0 + m) * H + hg] = t;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int j = j0 + 4 * kq + r;
                const float d2 = __shfl(t, 4 * kq + r, 64);   // column total of source j
                if (j >= N) continue;
                const float d1 = a.a1 ? a.df1[(b * N + j) * H + hg] : 0.f;
#pragma unroll
                for (int nt = 0; nt < NTH; ++nt) {
                    const int c = 16 * nt + m;
                    if (c < DP) {
                        float v = acc[h][nt][r];
                    if constexpr (DROP) {
                        if (a.feat_keep < 1.f) v = drop_feat1(v, seed_now(a), b * N + j, HD, hg * DP + c, a.feat_keep);
                    }
                        if (a.a1) v += d1 * a.a1[hg * DP + c] + d2 * a.a2[hg * DP + c];
                        a.dWh[(b * N + j) * (int64_t)HD + hg * DP + c] = v;
                    }
                }
            }
        }
    }
}

constexpr size_t kSmallGraphLds = 64 * 1024;

template <int H, int DP>
static inline size_t fwd_hg_lds(int N) {
    const size_t Np = (size_t)((N + 15) & ~15);
    return Np * (size_t)(WaveSplit<H>::HPW * DP + 4 + WaveSplit<H>::HPW) * sizeof(float);
}
template <int H, int DP>
static inline size_t bwd_hg_lds(int N) {
    const size_t Np = (size_t)((N + 15) & ~15);
    return Np * (size_t)(WaveSplit<H>::HPW * DP + 2 + WaveSplit<H>::HPW * 4) * sizeof(float);
}

template <int H>
static inline dim3 dense_grid(int64_t B, int N) {
    const int rows_per_block = 16 * WaveSplit<H>::WT;
    return dim3((unsigned)B, (unsigned)((N + rows_per_block - 1) / rows_per_block));
}

#define GATMI_DENSE_SHAPES(X) X(8, 8) X(4, 8) X(2, 8) X(1, 8) X(1, 4) X(2, 4) X(4, 4) X(8, 4) X(1, 16) X(2, 16) X(4, 16) X(1, 32) X(2, 32) X(1, 64)

}  // namespace gatmi

using namespace gatmi;

extern "C" int gatmi_dense_attn_fwd(int64_t B, int32_t N, int32_t H, int32_t dp,
                                    const float *Wh, const float *f1, const float *f2, const float *bias_mat,
                                    const float *bias, const float *resid, int32_t activation, float slope,
                                    float coef_keep, float feat_keep, uint64_t seed,
                                    float *out, float *agg, float *pos, float *lse, float *q, void *stream) {
    int G, LPH;
    int rc = check_shape(H, dp, &G, &LPH);
    if (rc) return rc;
    GATMI_REQUIRE(B >= 0 && N >= 0, GATMI_EINVAL, "dense_attn_fwd: negative size");
    if (B == 0 || N == 0) return GATMI_OK;
    GATMI_REQUIRE(B <= 0x7fffffff, GATMI_ERANGE, "dense_attn_fwd: batch %lld too large for one launch", (long long)B);
    GATMI_REQUIRE(Wh && f1 && f2 && bias_mat && out, GATMI_EINVAL, "dense_attn_fwd: null pointer");
    GATMI_REQUIRE(act_known(activation), GATMI_EUNSUPPORTED, "dense_attn_fwd: activation %d", activation);
    const bool save = agg || pos || lse || q;
    GATMI_REQUIRE(!save || (agg && pos && lse && q), GATMI_EINVAL, "dense_attn_fwd: agg, pos, lse, q must be given together");
    GATMI_REQUIRE(coef_keep > 0.f && coef_keep <= 1.f && feat_keep > 0.f && feat_keep <= 1.f, GATMI_EINVAL, "dense_attn_fwd: keep probabilities must be in (0,1]");
    const bool drop = coef_keep < 1.f || feat_keep < 1.f;
    GATMI_REQUIRE(!drop || save, GATMI_EINVAL, "dense_attn_fwd: dropout is a training-time option and needs agg, pos, lse, q");
    GATMI_REQUIRE_ALIGNED(Wh); GATMI_REQUIRE_ALIGNED(bias_mat);
    DenseFwdArgs a;
    a.B = B; a.N = N; a.Wh = Wh; a.f1 = f1; a.f2 = f2; a.bias_mat = bias_mat; a.bias = bias; a.resid = resid;
    a.out = out; a.agg = agg; a.pos = pos; a.lse = lse; a.q = q; a.activation = activation; a.slope = slope; a.coef_keep = coef_keep; a.feat_keep = feat_keep; a.seed = seed; a.seed_step = (coef_keep < 1.f || feat_keep < 1.f) ? dropout_step_source() : nullptr;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
#define X(hh, dd)                                                                                             \
    if (H == hh && dp == dd) {                                                                                \
        const dim3 grid = dense_grid<hh>(B, N);                                                               \
        const size_t lds = fwd_hg_lds<hh, dd>(N);                                                             \
        const dim3 hgrid((unsigned)(((B + 7) / 8) * 8 * WaveSplit<hh>::WH));                                           \
        if (lds <= kSmallGraphLds && drop) hipLaunchKernelGGL((dense_attn_fwd_hg_kernel<hh, dd, true, true>), hgrid, dim3(256), lds, st, a);  \
        else if (lds <= kSmallGraphLds && save) hipLaunchKernelGGL((dense_attn_fwd_hg_kernel<hh, dd, true, false>), hgrid, dim3(256), lds, st, a);  \
        else if (lds <= kSmallGraphLds) hipLaunchKernelGGL((dense_attn_fwd_hg_kernel<hh, dd, false, false>), hgrid, dim3(256), lds, st, a);    \
        else if (drop) hipLaunchKernelGGL((dense_attn_fwd_kernel<hh, dd, true, true>), grid, dim3(256), 0, st, a);   \
        else if (save) hipLaunchKernelGGL((dense_attn_fwd_kernel<hh, dd, true, false>), grid, dim3(256), 0, st, a);  \
        else hipLaunchKernelGGL((dense_attn_fwd_kernel<hh, dd, false, false>), grid, dim3(256), 0, st, a);            \
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
                                    float slope, float coef_keep, float feat_keep, uint64_t seed,
                                    float *dWh, float *df2, void *stream) {
    int G, LPH;
    int rc = check_shape(H, dp, &G, &LPH);
    if (rc) return rc;
    GATMI_REQUIRE(B >= 0 && N >= 0, GATMI_EINVAL, "dense_attn_bwd: negative size");
    if (B == 0 || N == 0) return GATMI_OK;
    GATMI_REQUIRE(B <= 0x7fffffff, GATMI_ERANGE, "dense_attn_bwd: batch %lld too large for one launch", (long long)B);
    GATMI_REQUIRE(g && stats && Wh && f2 && bias_mat && dWh && df2, GATMI_EINVAL, "dense_attn_bwd: null pointer");
    GATMI_REQUIRE((a1 == nullptr) == (a2 == nullptr) && (!a1 || df1), GATMI_EINVAL, "dense_attn_bwd: a1, a2 and df1 go together");
    GATMI_REQUIRE_ALIGNED(g);
    GATMI_REQUIRE(coef_keep > 0.f && coef_keep <= 1.f && feat_keep > 0.f && feat_keep <= 1.f, GATMI_EINVAL, "dense_attn_bwd: keep probabilities must be in (0,1]");
    const bool drop = coef_keep < 1.f || feat_keep < 1.f;
    DenseBwdArgs a;
    a.coef_keep = coef_keep; a.feat_keep = feat_keep; a.seed = seed; a.seed_step = (coef_keep < 1.f || feat_keep < 1.f) ? dropout_step_source() : nullptr;
    a.df1 = df1; a.a1 = a1; a.a2 = a2;
    a.B = B; a.N = N; a.g = g; a.stats = stats; a.Wh = Wh; a.f2 = f2; a.bias_mat = bias_mat; a.dWh = dWh; a.df2 = df2; a.slope = slope;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
#define X(hh, dd)                                                                                 \
    if (H == hh && dp == dd) {                                                                    \
        const dim3 grid = dense_grid<hh>(B, N);                                                   \
        const size_t lds = bwd_hg_lds<hh, dd>(N);                                                 \
        const dim3 hgrid((unsigned)(((B + 7) / 8) * 8 * WaveSplit<hh>::WH));                               \
        if (lds <= kSmallGraphLds && drop) hipLaunchKernelGGL((dense_attn_bwd_hg_kernel<hh, dd, true>), hgrid, dim3(256), lds, st, a);  \
        else if (lds <= kSmallGraphLds) hipLaunchKernelGGL((dense_attn_bwd_hg_kernel<hh, dd, false>), hgrid, dim3(256), lds, st, a);  \
        else if (drop) hipLaunchKernelGGL((dense_attn_bwd_kernel<hh, dd, true>), grid, dim3(256), 0, st, a);  \
        else hipLaunchKernelGGL((dense_attn_bwd_kernel<hh, dd, false>), grid, dim3(256), 0, st, a);            \
        GATMI_LAUNCH_CHECK("dense_attn_bwd");                                                     \
        return GATMI_OK;                                                                          \
    }
    GATMI_DENSE_SHAPES(X)
#undef X
    return fail(GATMI_EUNSUPPORTED, "dense_attn_bwd: no kernel for H=%d dp=%d", H, dp);
}
