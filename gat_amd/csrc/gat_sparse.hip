// CSR / CSC graph-attention kernels for gfx950 (MI355X): fused edge-softmax + aggregation forward,
// per-destination backward, per-source backward.
//
// Work decomposition (all three row kernels).  A feature row has hd = H*dp floats; it is covered
// by G = hd/4 adjacent lanes holding one float4 each, so one 64-lane wave walks 64/G rows at once
// and every row access is a contiguous 16-byte-per-lane load or store.  Lane `sub` of a group
// belongs to head sub / LPH (LPH = dp/4 lanes per head).  Rows come as work items sorted by
// length (longest first), so the row groups of one wave loop about equally long, and rows longer
// than the chunk size are cut into pieces whose partial results meet in a small merge kernel.
//
// Softmax is the online form: running maximum m, denominator s and un-normalised weighted sums,
// rescaled when the maximum grows; neighbours are taken four at a time so that one rescale
// factor and four numerators cost five v_exp_f32.
//
// The backward pass never revisits the CSR side.  With z_ij = f1_i + f2_j, e_ij = LeakyReLU(z_ij),
// alpha = softmax_j(e), the gradient of f1 is
//     df1_i = sum_j alpha_ij (dalpha_ij - c_i) * LeakyReLU'(z_ij),   c_i = sum_j alpha_ij dalpha_ij.
// Since sum_j alpha_ij (dalpha_ij - c_i) = 0 and LeakyReLU' = slope + (1-slope)*[z>0],
//     df1_i = (1-slope) * ( <g_i, pos_i> - c_i * q_i ),
// where pos_i / q_i are the aggregate / coefficient mass over the entries with z_ij > 0, which the
// forward kernel accumulates next to the full aggregate.  c_i = <g_i, agg_i>.  So the only
// gather in the backward pass is the per-source one (transposed SpMM + SDDMM over CSC).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <mutex>

#include "common.h"
#include "device_common.h"

namespace gatmi {

// Partial-result store of gatmi_gat_fwd_noalloc (the entry point for callers that may not allocate):
// a fixed array inside the code object instead of a caller workspace.  Every GPU the library runs
// on holds its own copy at its own address until the process ends.  In host code the NAME below is
// only a handle for hipGetSymbolAddress; kernels always receive the per-device address it returns.
constexpr size_t kScratchFloats = GATMI_NOALLOC_SCRATCH_BYTES / sizeof(float);
static __device__ __align__(256) float scratch[kScratchFloats];

struct FwdArgs {
    const int4 *items;
    int64_t n_items;
    const int32_t *col;
    const void *feat;          // gathered source rows, fp32 or bf16 (template parameter BF16)
    int64_t ldf;               // their row stride, in elements of their type
    const float *f1, *f2, *bias, *resid;
    const float *la2, *lb2;    // with f2 == nullptr: f2_j = <feat_j, la2> + lb2, formed from the gathered row (kernels built with F2ROW)
    float *out, *agg, *pos, *lse, *q, *ws;
    int64_t ws_slots;          // partial-result records `ws` has room for (pieces beyond it are dropped, never written)
    int activation, reduce;
    float slope, coef_keep, feat_keep;
    uint64_t seed;
    const uint64_t *seed_step;   // device word added to the seed at run time (seed_now), or nullptr
};

struct RowState {
    float m, s, q;
    float4 acc, pos;
};

// One batch of U neighbours of one row group.  F2ROW: the source half of the logit is not gathered
// from a table but comes with the gathered row: formed from it, f2_j = <feat_j, a2> + b2 per head
// (utils/layers.py:15; fp32 rows: four multiply-adds and a cross-lane step in the shadow of the row load),
// or read from the tail of the row's record (bf16 rows, whose rounding must stay out of the logits).
template <int G, int LPH, int U, bool SAVE, bool DROP, bool BF16, bool F2ROW>
__device__ __forceinline__ void fwd_batch(RowState &st, const FwdArgs &a, const int32_t *cp, int64_t epos,
                                          float f1v, float4 a2v, float b2h, int sub, int head) {
    constexpr int H = G / LPH;
    constexpr int HD = 4 * G;
    int32_t j[U];
    float z[U];
    float4 v[U];
#pragma unroll
    for (int t = 0; t < U; ++t) j[t] = cp[t];
#pragma unroll
    for (int t = 0; t < U; ++t) {
        if (!F2ROW) z[t] = f1v + a.f2[(int64_t)j[t] * H + head];
        v[t] = load_row4<BF16>(a.feat, j[t], a.ldf, sub * 4);
        // bf16 records: the exact fp32 f2 of the row sits behind its HD bf16 values, in the lines this gather fetches
        if (F2ROW && BF16) z[t] = f1v + *reinterpret_cast<const float *>(reinterpret_cast<const uint16_t *>(a.feat) + (int64_t)j[t] * a.ldf + HD + 2 * head);
    }
    if (F2ROW && !BF16) {
#pragma unroll
        for (int t = 0; t < U; ++t) z[t] = f1v + (group_sum<LPH>(dot4(v[t], a2v)) + b2h);
    }
    float e[U];
    float mn = st.m;
#pragma unroll
    for (int t = 0; t < U; ++t) {
        e[t] = leaky(z[t], a.slope);
        mn = fmaxf(mn, e[t]);
    }
    const float sc = exp_neg(st.m - mn);  // 0 on the first batch (m = -inf)
    // The batch is summed on its own and then added to the running sums: the long accumulators
    // take one rounding per batch instead of one per neighbour (rows of 10^4..10^5 entries).
    float bs = 0.f, bq = 0.f;
    float4 bacc = make_float4(0.f, 0.f, 0.f, 0.f), bpos = bacc;
#pragma unroll
    for (int t = 0; t < U; ++t) {
        const float p = exp_neg(e[t] - mn);
        bs += p;
        float pa = p;
        float4 vv = v[t];
        if (DROP) {
            if (a.coef_keep < 1.f)
                pa = keep_mask(seed_now(a), (uint64_t)(epos + t) * H + head, a.coef_keep) ? p / a.coef_keep : 0.f;
            if (a.feat_keep < 1.f) vv = drop_feat(vv, seed_now(a), j[t], HD, sub * 4, a.feat_keep);
        }
        bacc = f4_fma(pa, vv, bacc);
        if (SAVE) {
            const bool ps = z[t] > 0.f;
            bq += ps ? p : 0.f;
            bpos = f4_fma(ps ? pa : 0.f, vv, bpos);
        }
    }
    st.s = fmaf(st.s, sc, bs);
    st.acc = f4_fma(sc, st.acc, bacc);
    if (SAVE) {
        st.q = fmaf(st.q, sc, bq);
        st.pos = f4_fma(sc, st.pos, bpos);
    }
    st.m = mn;
}

// Normalise one finished row and write every output of the forward pass.
template <int G, int LPH, bool SAVE>
__device__ __forceinline__ void fwd_finish(const RowState &st, const FwdArgs &a, int64_t row, int sub, int head) {
    constexpr int H = G / LPH;
    constexpr int HD = 4 * G;
    const bool empty = !(st.s > 0.f);
    const float inv = empty ? 0.f : 1.0f / st.s;
    const float4 ag = f4_scale(inv, st.acc);
    if (SAVE) {
        store_stream4(a.agg + row * HD + sub * 4, ag);
        store_stream4(a.pos + row * HD + sub * 4, f4_scale(inv, st.pos));
        if (sub % LPH == 0) {
            a.lse[row * H + head] = empty ? 0.f : st.m + logf(st.s);
            a.q[row * H + head] = st.q * inv;
        }
    }
    float4 r = ag;
    if (a.bias) r = f4_add(r, *reinterpret_cast<const float4 *>(a.bias + sub * 4));
    if (a.reduce == GATMI_REDUCE_MEAN) {
        // identity activation only (checked on the host): mean over heads of (agg + bias + resid);
        // the residual projection is per head ([n, hd], utils/layers.py:30), so it is averaged too
        if (a.resid) r = f4_add(r, *reinterpret_cast<const float4 *>(a.resid + row * HD + sub * 4));
#pragma unroll
        for (int o = LPH; o < G; o <<= 1) {
            r.x += __shfl_xor(r.x, o, 64);
            r.y += __shfl_xor(r.y, o, 64);
            r.z += __shfl_xor(r.z, o, 64);
            r.w += __shfl_xor(r.w, o, 64);
        }
        if (sub < LPH) {
            r = f4_scale(1.0f / H, r);
            *reinterpret_cast<float4 *>(a.out + row * (4 * LPH) + sub * 4) = r;
        }
        return;
    }
    if (a.resid) r = f4_add(r, *reinterpret_cast<const float4 *>(a.resid + row * HD + sub * 4));
    if (a.activation != GATMI_ACT_IDENTITY) r = make_float4(act_f(r.x, a.activation), act_f(r.y, a.activation), act_f(r.z, a.activation), act_f(r.w, a.activation));
    store_stream4(a.out + row * HD + sub * 4, r);
}

template <int G, int LPH, bool SAVE, bool DROP, bool BF16, bool F2ROW>
__global__ __launch_bounds__(256) void gat_fwd_kernel(FwdArgs a) {
    constexpr int RPW = 64 / G;
    constexpr int H = G / LPH;
    constexpr int HD = 4 * G;
    const int lane = threadIdx.x & 63;
    const int sub = lane % G, grp = lane / G;
    const int head = sub / LPH;
    const int64_t it = ((int64_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * RPW + grp;
    if (it >= a.n_items) return;
    const int4 item = a.items[it];
    const int64_t row = item.x;
    const int len = item.z;
    const int32_t *cp = a.col + item.y;
    const float f1v = a.f1[row * H + head];
    float4 a2v = make_float4(0.f, 0.f, 0.f, 0.f);
    float b2h = 0.f;
    if (F2ROW && !BF16) {
        a2v = *reinterpret_cast<const float4 *>(a.la2 + sub * 4);
        if (a.lb2) b2h = a.lb2[head];
    }

    RowState st;
    st.m = -INFINITY;
    st.s = 0.f;
    st.q = 0.f;
    st.acc = make_float4(0.f, 0.f, 0.f, 0.f);
    st.pos = make_float4(0.f, 0.f, 0.f, 0.f);
    if (item.w <= -2 && (int64_t)(-item.w - 2) < a.ws_slots) {
        // second phase of a row whose first entries were swept by an earlier launch (two-phase work
        // items of a row-partitioned shard): take up the state that launch parked, then finish the row
        const float *rec = a.ws + (int64_t)(-item.w - 2) * (3 * H + 2 * HD);
        st.m = rec[head];
        st.s = rec[H + head];
        st.q = rec[2 * H + head];
        st.acc = *reinterpret_cast<const float4 *>(rec + 3 * H + sub * 4);
        st.pos = *reinterpret_cast<const float4 *>(rec + 3 * H + HD + sub * 4);
    }
    int k = 0;
    for (; k + 4 <= len; k += 4) fwd_batch<G, LPH, 4, SAVE, DROP, BF16, F2ROW>(st, a, cp + k, (int64_t)item.y + k, f1v, a2v, b2h, sub, head);
    if (k + 2 <= len) {
        fwd_batch<G, LPH, 2, SAVE, DROP, BF16, F2ROW>(st, a, cp + k, (int64_t)item.y + k, f1v, a2v, b2h, sub, head);
        k += 2;
    }
    if (k < len) fwd_batch<G, LPH, 1, SAVE, DROP, BF16, F2ROW>(st, a, cp + k, (int64_t)item.y + k, f1v, a2v, b2h, sub, head);

    if (item.w < 0) {
        fwd_finish<G, LPH, SAVE>(st, a, row, sub, head);
    } else if (item.w < a.ws_slots) {
        // piece of a long row: park the un-normalised state; layout [m H | s H | q H | acc HD | pos HD]
        float *rec = a.ws + (int64_t)item.w * (3 * H + 2 * HD);
        if (sub % LPH == 0) {
            rec[head] = st.m;
            rec[H + head] = st.s;
            rec[2 * H + head] = st.q;
        }
        *reinterpret_cast<float4 *>(rec + 3 * H + sub * 4) = st.acc;
        *reinterpret_cast<float4 *>(rec + 3 * H + HD + sub * 4) = st.pos;
    }
}

// Combine the pieces of each long row (one row group per split record).
template <int G, int LPH, bool SAVE>
__global__ __launch_bounds__(256) void gat_fwd_merge_kernel(FwdArgs a, const int4 *split, int64_t n_split) {
    constexpr int RPW = 64 / G;
    constexpr int H = G / LPH;
    constexpr int HD = 4 * G;
    const int lane = threadIdx.x & 63;
    const int sub = lane % G, grp = lane / G;
    const int head = sub / LPH;
    const int64_t r = ((int64_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * RPW + grp;
    if (r >= n_split) return;
    const int4 rec = split[r];
    RowState st;
    st.m = -INFINITY;
    st.s = 0.f;
    st.q = 0.f;
    st.acc = make_float4(0.f, 0.f, 0.f, 0.f);
    st.pos = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int c = 0; c < rec.z && rec.y + c < a.ws_slots; ++c) {
        const float *p = a.ws + (int64_t)(rec.y + c) * (3 * H + 2 * HD);
        const float pm = p[head], ps = p[H + head], pq = p[2 * H + head];
        const float4 pa = *reinterpret_cast<const float4 *>(p + 3 * H + sub * 4);
        const float4 pp = *reinterpret_cast<const float4 *>(p + 3 * H + HD + sub * 4);
        const float mn = fmaxf(st.m, pm);
        const float s0 = exp_neg(st.m - mn), s1 = exp_neg(pm - mn);
        st.s = st.s * s0 + ps * s1;
        st.q = st.q * s0 + pq * s1;
        st.acc = f4_fma(s1, pa, f4_scale(s0, st.acc));
        st.pos = f4_fma(s1, pp, f4_scale(s0, st.pos));
        st.m = mn;
    }
    fwd_finish<G, LPH, SAVE>(st, a, rec.x, sub, head);
}

// ------------------------------------------------------------------------------------------
// backward, per-destination part
// ------------------------------------------------------------------------------------------
struct BwdNodeArgs {
    int64_t n;
    const float *dout, *out, *agg, *pos, *q, *f1, *lse;
    const float *bias;         // with out == nullptr: the pre-activation is agg + bias, read from registers instead of `out`
    float *g, *stats, *df1, *ws;
    int64_t ldg, lds;          // row strides of g and stats in floats
    int64_t ldg16;             // row stride of g16 in bf16 elements
    uint32_t *g16;             // optional bf16 copy of g (two values per dword), the table the per-source kernel gathers
    int activation, reduce;
    float slope;
};

constexpr int kNodeBlocks = 1024;  // persistent grid of the per-destination kernel

template <int G, int LPH>
__global__ __launch_bounds__(256) void gat_bwd_node_kernel(BwdNodeArgs a) {
    constexpr int RPW = 64 / G;
    constexpr int H = G / LPH;
    constexpr int HD = 4 * G;
    __shared__ float4 red[256];
    const int lane = threadIdx.x & 63;
    const int sub = lane % G, grp = lane / G;
    const int head = sub / LPH;
    const int64_t stride = (int64_t)gridDim.x * 4 * RPW;
    float4 gsum = make_float4(0.f, 0.f, 0.f, 0.f);
    const float4 biasv = a.bias ? *reinterpret_cast<const float4 *>(a.bias + sub * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
    // g and the statistics as one record per row (stats = g + hd, common stride, a whole number of 16-byte pieces)
    // (or, bf16: stats right behind the HD bf16 values of g16, strides equal in bytes)
    const bool rec32 = a.g && a.stats == a.g + HD && a.lds == a.ldg;
    const bool rec16 = a.g16 && a.stats == reinterpret_cast<float *>(a.g16 + HD / 2) && 2 * a.lds == a.ldg16;
    const int tail = (int)(rec32 ? a.ldg - HD : a.lds - HD / 2);
    const bool record = H % 4 == 0 && (rec32 || rec16) && tail >= 3 * H && tail <= 4 * G &&
                        ((reinterpret_cast<uintptr_t>(a.f1) | reinterpret_cast<uintptr_t>(a.lse) | reinterpret_cast<uintptr_t>(a.stats)) & 15) == 0;
    for (int64_t row = ((int64_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * RPW + grp; row < a.n; row += stride) {
        float4 gv;
        if (a.reduce == GATMI_REDUCE_MEAN) {
            gv = f4_scale(1.0f / H, *reinterpret_cast<const float4 *>(a.dout + row * (4 * LPH) + (sub % LPH) * 4));
        } else {
            gv = load_stream4(a.dout + row * HD + sub * 4);
            if (a.activation != GATMI_ACT_IDENTITY && a.out) {
                const float4 o = *reinterpret_cast<const float4 *>(a.out + row * HD + sub * 4);
                gv.x *= act_grad_from_out(o.x, a.activation);
                gv.y *= act_grad_from_out(o.y, a.activation);
                gv.z *= act_grad_from_out(o.z, a.activation);
                gv.w *= act_grad_from_out(o.w, a.activation);
            }
        }
        const float4 ag = load_stream4(a.agg + row * HD + sub * 4);
        const float4 ps = load_stream4(a.pos + row * HD + sub * 4);
        if (a.reduce != GATMI_REDUCE_MEAN && a.activation != GATMI_ACT_IDENTITY && !a.out) {
            // no residual: the layer's pre-activation is agg + bias, both at hand (saves the read of `out`)
            const float4 pre = f4_add(ag, biasv);
            gv.x *= act_grad_from_pre(pre.x, a.activation);
            gv.y *= act_grad_from_pre(pre.y, a.activation);
            gv.z *= act_grad_from_pre(pre.z, a.activation);
            gv.w *= act_grad_from_pre(pre.w, a.activation);
        }
        const float c = group_sum<LPH>(dot4(gv, ag));
        const float gp = group_sum<LPH>(dot4(gv, ps));
        if (a.g) store_stream4(a.g + row * a.ldg + sub * 4, gv);
        if (a.g16) *reinterpret_cast<uint2 *>(a.g16 + (row * a.ldg16 + sub * 4) / 2) = make_uint2(pack_bf16x2(gv.x, gv.y), pack_bf16x2(gv.z, gv.w));
        if (record) {
            // [g | f1 lse c | zeros] is one record of whole 128-byte lines: its tail goes out as 16-byte stores too
            // (lane k of the row writes floats 4k..4k+3 of the tail), so the wave writes the lines of its rows whole
            constexpr int NQ = H >= 4 ? H / 4 : 1;          // 16-byte pieces per statistic
            const int piece = sub % NQ;
            float4 cq;
            cq.x = __shfl(c, (lane - sub) + (4 * piece + 0) * LPH, 64);
            cq.y = __shfl(c, (lane - sub) + (4 * piece + 1) * LPH, 64);
            cq.z = __shfl(c, (lane - sub) + (4 * piece + 2) * LPH, 64);
            cq.w = __shfl(c, (lane - sub) + (4 * piece + 3) * LPH, 64);
            if (4 * sub < tail) {
                const int stat = sub / NQ;
                float4 t = make_float4(0.f, 0.f, 0.f, 0.f);
                if (stat == 0) t = *reinterpret_cast<const float4 *>(a.f1 + row * H + 4 * piece);
                else if (stat == 1) t = *reinterpret_cast<const float4 *>(a.lse + row * H + 4 * piece);
                else if (stat == 2) t = cq;
                store_stream4(a.stats + row * a.lds + 4 * sub, t);
            }
        }
        if (sub % LPH == 0) {
            const float qv = a.q[row * H + head];
            if (!record) {
                a.stats[row * a.lds + head] = a.f1[row * H + head];
                a.stats[row * a.lds + H + head] = a.lse[row * H + head];
                a.stats[row * a.lds + 2 * H + head] = c;
            }
            a.df1[row * H + head] = (1.0f - a.slope) * (gp - c * qv);
        }
        gsum = f4_add(gsum, gv);
    }
    // column sums of g for dbias: lanes with equal `sub` across the block, fixed order
    red[threadIdx.x] = gsum;
    __syncthreads();
    if (threadIdx.x < G) {
        float4 t = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int i = threadIdx.x; i < 256; i += G) t = f4_add(t, red[i]);
        *reinterpret_cast<float4 *>(a.ws + (int64_t)blockIdx.x * HD + threadIdx.x * 4) = t;
    }
}

// out[c] = sum over `rows` partial rows of ws[r][c] (fixed order => reproducible)
__global__ void column_sum_kernel(const float *ws, int rows, int width, float *out) {
    const int c = blockIdx.x * kSumCols + threadIdx.x;
    const bool ok = c < width;
    const float t = block_column_sum(rows, [&](int r) { return ok ? ws[(int64_t)r * width + c] : 0.f; });
    if (ok && threadIdx.y == 0) out[c] = t;
}

// ------------------------------------------------------------------------------------------
// backward, per-source part (CSC)
// ------------------------------------------------------------------------------------------
struct BwdCscArgs {
    const int4 *items;
    int64_t n_items;
    const int32_t *rowidx, *perm;
    const void *g;             // gathered destination rows of g, fp32 or bf16 (template parameter BF16)
    const float *stats, *feat, *f2, *df1, *a1, *a2;
    int64_t ldg, lds;          // row strides of g (in elements of its type) and stats (floats)
    const float *la2, *lb2;    // with f2 == nullptr: f2_j = <feat_j, la2> + lb2, as the forward kernel formed it
    float *dfeat, *df2, *ws;
    float slope, coef_keep, feat_keep;
    uint64_t seed;
    const uint64_t *seed_step;   // device word added to the seed at run time (seed_now), or nullptr
};

struct SrcState {
    float4 acc;
    float df2;
};

template <int G, int LPH, int U, bool DROP, bool BF16>
__device__ __forceinline__ void bwd_batch(SrcState &st, const BwdCscArgs &a, const int32_t *rp, int64_t cpos,
                                          float f2v, float4 whj, int sub, int head) {
    constexpr int H = G / LPH;
    int32_t i[U];
    float4 gv[U];
    float f1v[U], lsev[U], cv[U];
#pragma unroll
    for (int t = 0; t < U; ++t) i[t] = rp[t];
#pragma unroll
    for (int t = 0; t < U; ++t) {
        gv[t] = load_row4<BF16>(a.g, i[t], a.ldg, sub * 4);
        const float *s = a.stats + (int64_t)i[t] * a.lds + head;
        f1v[t] = s[0];
        lsev[t] = s[H];
        cv[t] = s[2 * H];
    }
    float bdf2 = 0.f;
    float4 bacc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int t = 0; t < U; ++t) {
        const float z = f1v[t] + f2v;
        const float alpha = exp_neg(leaky(z, a.slope) - lsev[t]);
        float dalpha = group_sum<LPH>(dot4(gv[t], whj));
        float aa = alpha;
        if (DROP) {
            if (a.coef_keep < 1.f) {
                const bool kp = keep_mask(seed_now(a), (uint64_t)a.perm[cpos + t] * H + head, a.coef_keep);
                const float mk = kp ? 1.0f / a.coef_keep : 0.f;
                aa = alpha * mk;
                dalpha *= mk;
            }
        }
        const float dz = alpha * (dalpha - cv[t]) * (z > 0.f ? 1.f : a.slope);
        bacc = f4_fma(aa, gv[t], bacc);
        bdf2 += dz;
    }
    st.acc = f4_add(st.acc, bacc);      // one rounding of the long sums per batch, not per entry
    st.df2 += bdf2;
}

template <int G, int LPH, bool DROP>
__device__ __forceinline__ void bwd_finish(const SrcState &st, const BwdCscArgs &a, int64_t j, int sub, int head) {
    constexpr int H = G / LPH;
    constexpr int HD = 4 * G;
    float4 r = st.acc;
    if (DROP) {
        if (a.feat_keep < 1.f) r = drop_feat(r, seed_now(a), j, HD, sub * 4, a.feat_keep);
    }
    if (a.a1) {
        const float d1 = a.df1[j * H + head];
        r = f4_fma(d1, *reinterpret_cast<const float4 *>(a.a1 + sub * 4), r);
        r = f4_fma(st.df2, *reinterpret_cast<const float4 *>(a.a2 + sub * 4), r);
    }
    store_stream4(a.dfeat + j * HD + sub * 4, r);
    if (sub % LPH == 0) a.df2[j * H + head] = st.df2;
}

template <int G, int LPH, bool DROP, bool BF16>
__global__ __launch_bounds__(256) void gat_bwd_csc_kernel(BwdCscArgs a) {
    constexpr int RPW = 64 / G;
    constexpr int H = G / LPH;
    constexpr int HD = 4 * G;
    const int lane = threadIdx.x & 63;
    const int sub = lane % G, grp = lane / G;
    const int head = sub / LPH;
    const int64_t it = ((int64_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * RPW + grp;
    if (it >= a.n_items) return;
    const int4 item = a.items[it];
    const int64_t j = item.x;
    const int len = item.z;
    const int32_t *rp = a.rowidx + item.y;
    float4 whj = *reinterpret_cast<const float4 *>(a.feat + j * HD + sub * 4);
    float f2v;
    if (a.f2) {
        f2v = a.f2[j * H + head];
    } else {
        // the same expression, on the same numbers, as gat_fwd_kernel<..., F2ROW> on fp32 rows
        const float4 w = whj;
        f2v = group_sum<LPH>(dot4(w, *reinterpret_cast<const float4 *>(a.la2 + sub * 4))) + (a.lb2 ? a.lb2[head] : 0.f);
    }
    if (DROP) {
        if (a.feat_keep < 1.f) whj = drop_feat(whj, seed_now(a), j, HD, sub * 4, a.feat_keep);
    }
    SrcState st;
    st.acc = make_float4(0.f, 0.f, 0.f, 0.f);
    st.df2 = 0.f;
    int k = 0;
    for (; k + 4 <= len; k += 4) bwd_batch<G, LPH, 4, DROP, BF16>(st, a, rp + k, (int64_t)item.y + k, f2v, whj, sub, head);
    if (k + 2 <= len) {
        bwd_batch<G, LPH, 2, DROP, BF16>(st, a, rp + k, (int64_t)item.y + k, f2v, whj, sub, head);
        k += 2;
    }
    if (k < len) bwd_batch<G, LPH, 1, DROP, BF16>(st, a, rp + k, (int64_t)item.y + k, f2v, whj, sub, head);
    if (item.w < 0) {
        bwd_finish<G, LPH, DROP>(st, a, j, sub, head);
    } else {
        float *rec = a.ws + (int64_t)item.w * (H + HD);  // [df2 H | acc HD]
        if (sub % LPH == 0) rec[head] = st.df2;
        *reinterpret_cast<float4 *>(rec + H + sub * 4) = st.acc;
    }
}

template <int G, int LPH, bool DROP>
__global__ __launch_bounds__(256) void gat_bwd_merge_kernel(BwdCscArgs a, const int4 *split, int64_t n_split) {
    constexpr int RPW = 64 / G;
    constexpr int H = G / LPH;
    constexpr int HD = 4 * G;
    const int lane = threadIdx.x & 63;
    const int sub = lane % G, grp = lane / G;
    const int head = sub / LPH;
    const int64_t r = ((int64_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * RPW + grp;
    if (r >= n_split) return;
    const int4 rec = split[r];
    SrcState st;
    st.acc = make_float4(0.f, 0.f, 0.f, 0.f);
    st.df2 = 0.f;
    for (int c = 0; c < rec.z; ++c) {
        const float *p = a.ws + (int64_t)(rec.y + c) * (H + HD);
        st.df2 += p[head];
        st.acc = f4_add(st.acc, *reinterpret_cast<const float4 *>(p + H + sub * 4));
    }
    bwd_finish<G, LPH, DROP>(st, a, rec.x, sub, head);
}

// dfeat_j += df1_j * a1 + df2_j * a2 for owned rows, after halo contributions have been added
// (row-partitioned runs keep this step out of the per-source kernel).
template <int G, int LPH>
__global__ __launch_bounds__(256) void gat_bwd_finalize_kernel(int64_t n, const float *df1, const float *df2,
                                                                const float *a1, const float *a2, float *dfeat) {
    constexpr int RPW = 64 / G;
    constexpr int H = G / LPH;
    constexpr int HD = 4 * G;
    const int lane = threadIdx.x & 63;
    const int sub = lane % G, grp = lane / G;
    const int head = sub / LPH;
    const float4 a1v = *reinterpret_cast<const float4 *>(a1 + sub * 4);
    const float4 a2v = *reinterpret_cast<const float4 *>(a2 + sub * 4);
    const int64_t stride = (int64_t)gridDim.x * 4 * RPW;
    for (int64_t row = ((int64_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * RPW + grp; row < n; row += stride) {
        float4 r = *reinterpret_cast<const float4 *>(dfeat + row * HD + sub * 4);
        r = f4_fma(df1[row * H + head], a1v, r);
        r = f4_fma(df2[row * H + head], a2v, r);
        *reinterpret_cast<float4 *>(dfeat + row * HD + sub * 4) = r;
    }
}

// The same step with the peers' contributions taken in on the way (row-partitioned runs): row r first adds
// the rows recv[pos[k]], k in [rptr[r], rptr[r+1]), that the reverse halo exchange returned for it -- in
// list order, so the sum is the one gatmi_halo_unpack_sum forms -- to dfeat and df2, then the logit terms.
// One pass over the owned rows instead of an unpack pass plus a finalize pass.
template <int G, int LPH>
__global__ __launch_bounds__(256) void gat_bwd_finalize_halo_kernel(int64_t n, const float *df1, float *df2, const float *a1, const float *a2,
                                                                     float *dfeat, const int64_t *rptr, const int64_t *pos,
                                                                     const float *recv_feat, const float *recv_df2) {
    constexpr int RPW = 64 / G;
    constexpr int H = G / LPH;
    constexpr int HD = 4 * G;
    const int lane = threadIdx.x & 63;
    const int sub = lane % G, grp = lane / G;
    const int head = sub / LPH;
    const float4 a1v = *reinterpret_cast<const float4 *>(a1 + sub * 4);
    const float4 a2v = *reinterpret_cast<const float4 *>(a2 + sub * 4);
    const int64_t stride = (int64_t)gridDim.x * 4 * RPW;
    for (int64_t row = ((int64_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * RPW + grp; row < n; row += stride) {
        const int64_t k0 = rptr[row], k1 = rptr[row + 1];
        float4 r = load_stream4(dfeat + row * HD + sub * 4);
        float d2 = df2[row * H + head];
        const float d1 = df1[row * H + head];
        // Up to four returned rows at a time: their positions, then all their rows, are requested together (the loads
        // are unconditional, clamped inside the list; a row-by-row walk is a chain of dependent loads), and they are
        // added in list order, so the sum is the one gatmi_halo_unpack_sum forms.
        for (int64_t k = k0; k < k1; k += 4) {
            int64_t p[4];
            float4 v[4];
            float e[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) p[u] = pos[k + u < k1 ? k + u : k1 - 1];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                v[u] = load_stream4(recv_feat + p[u] * HD + sub * 4);
                e[u] = recv_df2[p[u] * H + head];
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                if (k + u < k1) {
                    r = f4_add(r, v[u]);
                    d2 += e[u];
                }
            }
        }
        r = f4_fma(d1, a1v, r);
        r = f4_fma(d2, a2v, r);
        store_stream4(dfeat + row * HD + sub * 4, r);
        if (sub % LPH == 0) df2[row * H + head] = d2;
    }
}

// ------------------------------------------------------------------------------------------
// host-side dispatch over (G, LPH)
// ------------------------------------------------------------------------------------------
static inline unsigned blocks_for(int64_t n_groups, int G) {
    const int64_t per_block = 4 * (64 / G);
    return (unsigned)((n_groups + per_block - 1) / per_block);
}

#define GATMI_FOR_SHAPES(X) \
    X(1, 1) X(2, 1) X(2, 2) X(4, 1) X(4, 2) X(4, 4) X(8, 1) X(8, 2) X(8, 4) X(8, 8) \
    X(16, 1) X(16, 2) X(16, 4) X(16, 8) X(16, 16) X(32, 2) X(32, 4) X(32, 8) X(32, 16) X(32, 32) \
    X(64, 2) X(64, 4) X(64, 8) X(64, 16) X(64, 32) X(64, 64)

template <int G, int LPH, bool SAVE, bool DROP, bool BF16>
static void launch_fwd_rows(const FwdArgs &a, unsigned nb, hipStream_t st) {
    if (!a.f2) hipLaunchKernelGGL((gat_fwd_kernel<G, LPH, SAVE, DROP, BF16, true>), dim3(nb), dim3(256), 0, st, a);
    else hipLaunchKernelGGL((gat_fwd_kernel<G, LPH, SAVE, DROP, BF16, false>), dim3(nb), dim3(256), 0, st, a);
}

template <int G, int LPH>
static int launch_fwd(const FwdArgs &a, const int4 *split, int64_t n_split, bool save, bool drop, bool bf16, hipStream_t st) {
    const unsigned nb = blocks_for(a.n_items, G);
    if (nb) {
        const int variant = (save ? 4 : 0) | (drop ? 2 : 0) | (bf16 ? 1 : 0);
        switch (variant) {
        case 0: launch_fwd_rows<G, LPH, false, false, false>(a, nb, st); break;
        case 1: launch_fwd_rows<G, LPH, false, false, true>(a, nb, st); break;
        case 2: launch_fwd_rows<G, LPH, false, true, false>(a, nb, st); break;
        case 3: launch_fwd_rows<G, LPH, false, true, true>(a, nb, st); break;
        case 4: launch_fwd_rows<G, LPH, true, false, false>(a, nb, st); break;
        case 5: launch_fwd_rows<G, LPH, true, false, true>(a, nb, st); break;
        case 6: launch_fwd_rows<G, LPH, true, true, false>(a, nb, st); break;
        default: launch_fwd_rows<G, LPH, true, true, true>(a, nb, st); break;
        }
        GATMI_LAUNCH_CHECK("gat_fwd");
    }
    if (n_split > 0) {
        const unsigned mb = blocks_for(n_split, G);
        if (save) hipLaunchKernelGGL((gat_fwd_merge_kernel<G, LPH, true>), dim3(mb), dim3(256), 0, st, a, split, n_split);
        else hipLaunchKernelGGL((gat_fwd_merge_kernel<G, LPH, false>), dim3(mb), dim3(256), 0, st, a, split, n_split);
        GATMI_LAUNCH_CHECK("gat_fwd_merge");
    }
    return GATMI_OK;
}

template <int G, int LPH>
static int launch_bwd_node(const BwdNodeArgs &a, float *dbias, hipStream_t st) {
    constexpr int HD = 4 * G;
    const int64_t per_block = 4 * (64 / G);
    const unsigned nb = (unsigned)std::min<int64_t>(kNodeBlocks, std::max<int64_t>(1, (a.n + per_block - 1) / per_block));
    hipLaunchKernelGGL((gat_bwd_node_kernel<G, LPH>), dim3(nb), dim3(256), 0, st, a);
    GATMI_LAUNCH_CHECK("gat_bwd_node");
    if (dbias) {
        hipLaunchKernelGGL(column_sum_kernel, dim3((HD + kSumCols - 1) / kSumCols), dim3(kSumCols, kSumRows), 0, st, a.ws, (int)nb, HD, dbias);
        GATMI_LAUNCH_CHECK("gat_bwd_node dbias");
    }
    return GATMI_OK;
}

template <int G, int LPH>
static int launch_bwd_csc(const BwdCscArgs &a, const int4 *split, int64_t n_split, bool drop, bool bf16, hipStream_t st) {
    const unsigned nb = blocks_for(a.n_items, G);
    if (nb) {
        if (bf16 && drop) hipLaunchKernelGGL((gat_bwd_csc_kernel<G, LPH, true, true>), dim3(nb), dim3(256), 0, st, a);
        else if (bf16) hipLaunchKernelGGL((gat_bwd_csc_kernel<G, LPH, false, true>), dim3(nb), dim3(256), 0, st, a);
        else if (drop) hipLaunchKernelGGL((gat_bwd_csc_kernel<G, LPH, true, false>), dim3(nb), dim3(256), 0, st, a);
        else hipLaunchKernelGGL((gat_bwd_csc_kernel<G, LPH, false, false>), dim3(nb), dim3(256), 0, st, a);
        GATMI_LAUNCH_CHECK("gat_bwd_csc");
    }
    if (n_split > 0) {
        const unsigned mb = blocks_for(n_split, G);
        if (drop) hipLaunchKernelGGL((gat_bwd_merge_kernel<G, LPH, true>), dim3(mb), dim3(256), 0, st, a, split, n_split);
        else hipLaunchKernelGGL((gat_bwd_merge_kernel<G, LPH, false>), dim3(mb), dim3(256), 0, st, a, split, n_split);
        GATMI_LAUNCH_CHECK("gat_bwd_merge");
    }
    return GATMI_OK;
}

template <int G, int LPH>
static int launch_bwd_finalize(int64_t n, const float *df1, const float *df2, const float *a1, const float *a2,
                               float *dfeat, hipStream_t st) {
    const int64_t per_block = 4 * (64 / G);
    const unsigned nb = (unsigned)std::min<int64_t>(4096, std::max<int64_t>(1, (n + per_block - 1) / per_block));
    hipLaunchKernelGGL((gat_bwd_finalize_kernel<G, LPH>), dim3(nb), dim3(256), 0, st, n, df1, df2, a1, a2, dfeat);
    GATMI_LAUNCH_CHECK("gat_bwd_finalize");
    return GATMI_OK;
}

template <int G, int LPH>
static int launch_bwd_finalize_halo(int64_t n, const float *df1, float *df2, const float *a1, const float *a2, float *dfeat,
                                    const int64_t *rptr, const int64_t *pos, const float *recv_feat, const float *recv_df2, hipStream_t st) {
    const int64_t per_block = 4 * (64 / G);
    const unsigned nb = (unsigned)std::min<int64_t>(8192, std::max<int64_t>(1, (n + per_block - 1) / per_block));
    hipLaunchKernelGGL((gat_bwd_finalize_halo_kernel<G, LPH>), dim3(nb), dim3(256), 0, st, n, df1, df2, a1, a2, dfeat, rptr, pos, recv_feat, recv_df2);
    GATMI_LAUNCH_CHECK("gat_bwd_finalize_halo");
    return GATMI_OK;
}

}  // namespace gatmi

using namespace gatmi;

extern "C" int gatmi_gat_bwd_finalize_halo(int64_t n, int32_t H, int32_t dp, const float *df1, float *df2,
                                           const float *a1, const float *a2, float *dfeat,
                                           const int64_t *rptr, const int64_t *pos, const float *recv_dfeat, const float *recv_df2,
                                           void *stream) {
    int G, LPH;
    int rc = check_shape(H, dp, &G, &LPH);
    if (rc) return rc;
    GATMI_REQUIRE(n >= 0, GATMI_EINVAL, "gat_bwd_finalize_halo: negative size");
    if (n == 0) return GATMI_OK;
    GATMI_REQUIRE(df1 && df2 && a1 && a2 && dfeat && rptr && pos && recv_dfeat && recv_df2, GATMI_EINVAL, "gat_bwd_finalize_halo: null pointer");
    GATMI_REQUIRE_ALIGNED(a1); GATMI_REQUIRE_ALIGNED(a2); GATMI_REQUIRE_ALIGNED(dfeat); GATMI_REQUIRE_ALIGNED(recv_dfeat);
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
#define X(gg, l) if (G == gg && LPH == l) return launch_bwd_finalize_halo<gg, l>(n, df1, df2, a1, a2, dfeat, rptr, pos, recv_dfeat, recv_df2, st);
    GATMI_FOR_SHAPES(X)
#undef X
    return fail(GATMI_EUNSUPPORTED, "gat_bwd_finalize_halo: no kernel for H=%d dp=%d", H, dp);
}

extern "C" int gatmi_gat_bwd_finalize(int64_t n, int32_t H, int32_t dp, const float *df1, const float *df2,
                                      const float *a1, const float *a2, float *dfeat, void *stream) {
    int G, LPH;
    int rc = check_shape(H, dp, &G, &LPH);
    if (rc) return rc;
    GATMI_REQUIRE(n >= 0, GATMI_EINVAL, "gat_bwd_finalize: negative size");
    if (n == 0) return GATMI_OK;
    GATMI_REQUIRE(df1 && df2 && a1 && a2 && dfeat, GATMI_EINVAL, "gat_bwd_finalize: null pointer");
    GATMI_REQUIRE_ALIGNED(a1); GATMI_REQUIRE_ALIGNED(a2); GATMI_REQUIRE_ALIGNED(dfeat);
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
#define X(gg, l) if (G == gg && LPH == l) return launch_bwd_finalize<gg, l>(n, df1, df2, a1, a2, dfeat, st);
    GATMI_FOR_SHAPES(X)
#undef X
    return fail(GATMI_EUNSUPPORTED, "gat_bwd_finalize: no kernel for H=%d dp=%d", H, dp);
}

namespace gatmi {
__global__ __launch_bounds__(256) void to_bf16_kernel(const float4 *src, int64_t n4, uint2 *dst) {
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < n4; e += (int64_t)gridDim.x * 256) {
        const float4 v = src[e];
        dst[e] = make_uint2(pack_bf16x2(v.x, v.y), pack_bf16x2(v.z, v.w));
    }
}
}  // namespace gatmi

namespace gatmi {
// Row records [hd bf16 | f2: H fp32 | pad] of the bf16 gather table (ld16 bf16 elements apart); pad is left alone.
__global__ __launch_bounds__(256) void to_bf16_records_kernel(const float *src, const float *f2, int64_t n, int hd, int H, int64_t ld16, uint16_t *dst) {
    const int per_row = hd / 4 + H;                    // 8-byte pieces of Wh, then 4-byte f2 values
    const int64_t total = n * per_row;
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
        const int64_t row = e / per_row;
        const int k = (int)(e - row * per_row);
        if (k < hd / 4) {
            const float4 v = *reinterpret_cast<const float4 *>(src + row * hd + 4 * k);
            *reinterpret_cast<uint2 *>(dst + row * ld16 + 4 * k) = make_uint2(pack_bf16x2(v.x, v.y), pack_bf16x2(v.z, v.w));
        } else {
            *reinterpret_cast<float *>(dst + row * ld16 + hd + 2 * (k - hd / 4)) = f2[row * H + (k - hd / 4)];
        }
    }
}
}  // namespace gatmi

extern "C" int gatmi_to_bf16_records(const float *src, const float *f2, int64_t n, int32_t H, int32_t dp, int64_t ld16,
                                     uint16_t *dst, void *stream) {
    int G, LPH;
    int rc = check_shape(H, dp, &G, &LPH);
    if (rc) return rc;
    const int hd = H * dp;
    GATMI_REQUIRE(n >= 0 && ld16 >= (int64_t)hd + 2 * H && ld16 % 4 == 0, GATMI_EINVAL, "to_bf16_records: n=%lld ld16=%lld (>= hd + 2H, multiple of 4)", (long long)n, (long long)ld16);
    if (n == 0) return GATMI_OK;
    GATMI_REQUIRE(src && f2 && dst, GATMI_EINVAL, "to_bf16_records: null pointer");
    GATMI_REQUIRE_ALIGNED(src); GATMI_REQUIRE_ALIGNED(dst);
    const int64_t total = n * (hd / 4 + H);
    const unsigned nb = (unsigned)std::min<int64_t>(8192, (total + 255) / 256);
    hipLaunchKernelGGL(to_bf16_records_kernel, dim3(nb), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), src, f2, n, hd, H, ld16, dst);
    GATMI_LAUNCH_CHECK("to_bf16_records");
    return GATMI_OK;
}

extern "C" int gatmi_to_bf16(const float *src, int64_t n_elems, uint16_t *dst, void *stream) {
    GATMI_REQUIRE(n_elems >= 0 && n_elems % 4 == 0, GATMI_EINVAL, "to_bf16: element count %lld must be a non-negative multiple of 4", (long long)n_elems);
    if (n_elems == 0) return GATMI_OK;
    GATMI_REQUIRE(src && dst, GATMI_EINVAL, "to_bf16: null pointer");
    GATMI_REQUIRE_ALIGNED(src); GATMI_REQUIRE_ALIGNED(dst);
    const int64_t n4 = n_elems / 4;
    const unsigned nb = (unsigned)std::min<int64_t>(8192, (n4 + 255) / 256);
    hipLaunchKernelGGL(to_bf16_kernel, dim3(nb), dim3(256), 0, reinterpret_cast<hipStream_t>(stream),
                       reinterpret_cast<const float4 *>(src), n4, reinterpret_cast<uint2 *>(dst));
    GATMI_LAUNCH_CHECK("to_bf16");
    return GATMI_OK;
}

extern "C" size_t gatmi_gat_fwd_ws_floats(int64_t n_slots, int32_t H, int32_t dp) {
    if (n_slots <= 0) return 0;
    return (size_t)n_slots * (size_t)(3 * H + 2 * H * dp);
}

static int gat_fwd_impl(const char *name, int64_t n_dst, int32_t H, int32_t dp,
                        const int32_t *items, int64_t n_items, const int32_t *split, int64_t n_split,
                        const int32_t *col, const void *feat, int32_t feat_dtype, int64_t ldfeat, const float *f1, const float *f2,
                        const float *logit_a2, const float *logit_b2,
                        const float *bias, const float *resid, int32_t activation, int32_t reduce,
                        float slope, float coef_keep, float feat_keep, uint64_t seed,
                        float *out, float *agg, float *pos, float *lse, float *q,
                        float *ws, int64_t ws_slots, void *stream) {
    int G, LPH;
    int rc = check_shape(H, dp, &G, &LPH);
    if (rc) return rc;
    GATMI_REQUIRE(n_dst >= 0 && n_items >= 0 && n_split >= 0, GATMI_EINVAL, "%s: negative size", name);
    if (n_dst == 0 || n_items == 0) return GATMI_OK;
    GATMI_REQUIRE(items && feat && f1 && out, GATMI_EINVAL, "%s: null pointer", name);
    GATMI_REQUIRE(feat_dtype == GATMI_DTYPE_F32 || feat_dtype == GATMI_DTYPE_BF16, GATMI_EUNSUPPORTED, "%s: feat_dtype %d", name, feat_dtype);
    if (feat_dtype == GATMI_DTYPE_F32) {
        GATMI_REQUIRE((f2 != nullptr) != (logit_a2 != nullptr), GATMI_EINVAL, "%s: give either the table f2 or logit_a2 (f2 formed from the gathered rows), not both or neither", name);
        GATMI_REQUIRE(ldfeat >= (int64_t)H * dp && ldfeat % 4 == 0, GATMI_EINVAL, "%s: row stride ldfeat=%lld (>= hd, a multiple of 4)", name, (long long)ldfeat);
    } else {
        GATMI_REQUIRE(!logit_a2, GATMI_EINVAL, "%s: bf16 rows take f2 from the table or from their records (f2 = NULL), not from logit_a2", name);
        GATMI_REQUIRE(ldfeat >= (int64_t)H * dp + (f2 ? 0 : 2 * H) && ldfeat % 4 == 0, GATMI_EINVAL,
                      "%s: row stride ldfeat=%lld (bf16 elements; >= hd, or >= hd + 2H for records that carry f2; a multiple of 4)", name, (long long)ldfeat);
    }
    GATMI_REQUIRE(logit_a2 || !logit_b2, GATMI_EINVAL, "%s: logit_b2 goes with logit_a2", name);
    GATMI_REQUIRE_ALIGNED(logit_a2);
    GATMI_REQUIRE(n_split == 0 || (split && ws), GATMI_EINVAL, "%s: split rows need `split` and a partial-result store", name);
    GATMI_REQUIRE(act_known(activation), GATMI_EUNSUPPORTED, "%s: activation %d", name, activation);
    GATMI_REQUIRE(reduce == GATMI_REDUCE_CONCAT || (reduce == GATMI_REDUCE_MEAN && activation == GATMI_ACT_IDENTITY), GATMI_EUNSUPPORTED,
                  "%s: reduce %d with activation %d (mean over heads needs the identity activation)", name, reduce, activation);
    GATMI_REQUIRE(coef_keep > 0.f && coef_keep <= 1.f && feat_keep > 0.f && feat_keep <= 1.f, GATMI_EINVAL, "%s: keep probabilities must be in (0,1]", name);
    const bool save = agg || pos || lse || q;
    GATMI_REQUIRE(!save || (agg && pos && lse && q), GATMI_EINVAL, "%s: agg, pos, lse, q must be given together", name);
    GATMI_REQUIRE_ALIGNED(items); GATMI_REQUIRE_ALIGNED(split); GATMI_REQUIRE_ALIGNED(feat); GATMI_REQUIRE_ALIGNED(bias);
    GATMI_REQUIRE_ALIGNED(resid); GATMI_REQUIRE_ALIGNED(out); GATMI_REQUIRE_ALIGNED(agg); GATMI_REQUIRE_ALIGNED(pos); GATMI_REQUIRE_ALIGNED(ws);
    FwdArgs a;
    a.items = reinterpret_cast<const int4 *>(items);
    a.n_items = n_items;
    a.col = col; a.feat = feat; a.ldf = ldfeat; a.f1 = f1; a.f2 = f2; a.la2 = logit_a2; a.lb2 = logit_b2; a.bias = bias; a.resid = resid;
    a.out = out; a.agg = agg; a.pos = pos; a.lse = lse; a.q = q; a.ws = ws; a.ws_slots = ws_slots;
    a.activation = activation; a.reduce = reduce;
    a.slope = slope; a.coef_keep = coef_keep; a.feat_keep = feat_keep; a.seed = seed; a.seed_step = (coef_keep < 1.f || feat_keep < 1.f) ? dropout_step_source() : nullptr;
    const bool drop = coef_keep < 1.f || feat_keep < 1.f;
    const bool bf16 = feat_dtype == GATMI_DTYPE_BF16;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const int4 *sp = reinterpret_cast<const int4 *>(split);
#define X(g, l) if (G == g && LPH == l) return launch_fwd<g, l>(a, sp, n_split, save, drop, bf16, st);
    GATMI_FOR_SHAPES(X)
#undef X
    return fail(GATMI_EUNSUPPORTED, "%s: no kernel for H=%d dp=%d", name, H, dp);
}

extern "C" int gatmi_gat_fwd(int64_t n_dst, int32_t H, int32_t dp,
                             const int32_t *items, int64_t n_items, const int32_t *split, int64_t n_split,
                             const int32_t *col, const void *feat, int32_t feat_dtype, int64_t ldfeat, const float *f1, const float *f2,
                             const float *logit_a2, const float *logit_b2,
                             const float *bias, const float *resid, int32_t activation, int32_t reduce,
                             float slope, float coef_keep, float feat_keep, uint64_t seed,
                             float *out, float *agg, float *pos, float *lse, float *q,
                             float *ws, void *stream) {
    // the caller sized `ws` with gatmi_gat_fwd_ws_floats for its own slot count: no limit to enforce here
    return gat_fwd_impl("gat_fwd", n_dst, H, dp, items, n_items, split, n_split, col, feat, feat_dtype, ldfeat, f1, f2, logit_a2, logit_b2, bias, resid,
                        activation, reduce, slope, coef_keep, feat_keep, seed, out, agg, pos, lse, q, ws, INT64_MAX, stream);
}

// Address of this device's copy of `scratch`, looked up once per device with that device current.
namespace {
constexpr int kMaxDevices = 64;
std::mutex scratch_mutex;
float *scratch_of_device[kMaxDevices] = {};

int scratch_for_current_device(float **out) {
    int dev = -1;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return fail(GATMI_EHIP, "gat_fwd_noalloc: hipGetDevice: %s", hipGetErrorString(e));
    GATMI_REQUIRE(dev >= 0 && dev < kMaxDevices, GATMI_EUNSUPPORTED, "gat_fwd_noalloc: device ordinal %d (at most %d devices)", dev, kMaxDevices);
    std::lock_guard<std::mutex> lock(scratch_mutex);
    if (!scratch_of_device[dev]) {
        float *p = nullptr;
        e = hipGetSymbolAddress(reinterpret_cast<void **>(&p), HIP_SYMBOL(scratch));
        if (e != hipSuccess || !p)
            return fail(GATMI_EHIP, "gat_fwd_noalloc: hipGetSymbolAddress on device %d: %s", dev, hipGetErrorString(e));
        scratch_of_device[dev] = p;
    }
    *out = scratch_of_device[dev];
    return GATMI_OK;
}
}  // namespace

extern "C" int gatmi_gat_fwd_noalloc(int64_t n_dst, int32_t H, int32_t dp,
                                     const int32_t *items, int64_t n_items, const int32_t *split, int64_t n_split,
                                     int64_t n_slots,
                                     const int32_t *col, const void *feat, int32_t feat_dtype, int64_t ldfeat, const float *f1, const float *f2,
                                     const float *logit_a2, const float *logit_b2,
                                     const float *bias, const float *resid, int32_t activation, int32_t reduce,
                                     float slope, float coef_keep, float feat_keep, uint64_t seed,
                                     float *out, float *agg, float *pos, float *lse, float *q, void *stream) {
    // two-phase items (gatmi_work_items_phased) park rows in slots without a split record, so slots
    // without split rows are fine; split rows without any slot are not
    GATMI_REQUIRE(n_slots >= 0 && (n_split == 0 || n_slots > 0), GATMI_EINVAL,
                  "gat_fwd_noalloc: n_slots=%lld with n_split=%lld (both come from gatmi_work_items_count or gatmi_work_items_phased_count)", (long long)n_slots, (long long)n_split);
    const size_t need = gatmi_gat_fwd_ws_floats(n_slots, H, dp);
    GATMI_REQUIRE(need <= kScratchFloats, GATMI_ERANGE,
                  "gat_fwd_noalloc: %lld partial-result slots need %zu bytes, the built-in store holds %zu; use gatmi_gat_fwd "
                  "with a workspace, or a larger chunk so that fewer rows are split", (long long)n_slots, need * sizeof(float), (size_t)GATMI_NOALLOC_SCRATCH_BYTES);
    float *ws = nullptr;
    if (n_slots > 0) {
        int rc = scratch_for_current_device(&ws);
        if (rc) return rc;
    }
    // Nothing in the store has to start at zero: each piece's record is written whole by the row kernel
    // before the merge kernel, ordered behind it on `stream`, reads it.
    return gat_fwd_impl("gat_fwd_noalloc", n_dst, H, dp, items, n_items, split, n_split, col, feat, feat_dtype, ldfeat, f1, f2, logit_a2, logit_b2, bias, resid,
                        activation, reduce, slope, coef_keep, feat_keep, seed, out, agg, pos, lse, q, ws, n_slots, stream);
}

// Step words of gatmi_dropout_step_source, one per device.  The counter lets the launch path skip the
// device query while no word is registered anywhere (the usual case).
namespace {
std::atomic<const uint64_t *> step_word_of_device[kMaxDevices] = {};
std::atomic<int> step_words_registered{0};
}  // namespace

const uint64_t *gatmi::dropout_step_source() {
    if (step_words_registered.load(std::memory_order_acquire) == 0) return nullptr;
    int dev = -1;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kMaxDevices) return nullptr;
    return step_word_of_device[dev].load(std::memory_order_acquire);
}

extern "C" int gatmi_dropout_step_source(const uint64_t *device_word) {
    int dev = -1;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return fail(GATMI_EHIP, "dropout_step_source: hipGetDevice: %s", hipGetErrorString(e));
    GATMI_REQUIRE(dev >= 0 && dev < kMaxDevices, GATMI_EUNSUPPORTED, "dropout_step_source: device ordinal %d (at most %d devices)", dev, kMaxDevices);
    GATMI_REQUIRE((reinterpret_cast<uintptr_t>(device_word) & 7u) == 0, GATMI_EALIGN, "dropout_step_source: the word is not 8-byte aligned");
    const uint64_t *old = step_word_of_device[dev].exchange(device_word, std::memory_order_acq_rel);
    step_words_registered.fetch_add((device_word ? 1 : 0) - (old ? 1 : 0), std::memory_order_acq_rel);
    return GATMI_OK;
}

extern "C" const uint64_t *gatmi_dropout_step_source_current(void) {
    int dev = -1;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kMaxDevices) return nullptr;
    return step_word_of_device[dev].load(std::memory_order_acquire);
}

extern "C" uint64_t gatmi_dropout_step_multiplier(void) { return kSeedStepMul; }

extern "C" size_t gatmi_gat_bwd_node_ws_floats(int32_t H, int32_t dp) { return (size_t)kNodeBlocks * (size_t)H * dp; }

extern "C" int gatmi_gat_bwd_node(int64_t n_dst, int32_t H, int32_t dp,
                                  const float *dout, const float *out, const float *agg, const float *pos,
                                  const float *q, const float *f1, const float *lse, const float *bias,
                                  int32_t activation, int32_t reduce, float slope,
                                  float *g, int64_t ldg, uint16_t *g16, int64_t ldg16, float *stats, int64_t lds, float *df1, float *dbias,
                                  float *ws, void *stream) {
    int G, LPH;
    int rc = check_shape(H, dp, &G, &LPH);
    if (rc) return rc;
    GATMI_REQUIRE(n_dst >= 0, GATMI_EINVAL, "gat_bwd_node: negative size");
    GATMI_REQUIRE(dout && agg && pos && q && f1 && lse && (g || g16) && stats && df1 && ws, GATMI_EINVAL, "gat_bwd_node: null pointer");
    GATMI_REQUIRE_ALIGNED(g16);
    GATMI_REQUIRE(act_known(activation), GATMI_EUNSUPPORTED, "gat_bwd_node: activation %d", activation);
    GATMI_REQUIRE_ALIGNED(bias);
    GATMI_REQUIRE(reduce == GATMI_REDUCE_CONCAT || (reduce == GATMI_REDUCE_MEAN && activation == GATMI_ACT_IDENTITY), GATMI_EUNSUPPORTED, "gat_bwd_node: reduce %d", reduce);
    GATMI_REQUIRE_ALIGNED(dout); GATMI_REQUIRE_ALIGNED(out); GATMI_REQUIRE_ALIGNED(agg); GATMI_REQUIRE_ALIGNED(pos);
    GATMI_REQUIRE_ALIGNED(g); GATMI_REQUIRE_ALIGNED(ws);
    BwdNodeArgs a;
    a.n = n_dst; a.dout = dout; a.out = out; a.bias = out ? nullptr : bias; a.agg = agg; a.pos = pos; a.q = q; a.f1 = f1; a.lse = lse;
    GATMI_REQUIRE((!g || (ldg >= (int64_t)H * dp && ldg % 4 == 0)) && lds >= 3 * (int64_t)H, GATMI_EINVAL,
                  "gat_bwd_node: row strides ldg=%lld (>= hd, multiple of 4) lds=%lld (>= 3H)", (long long)ldg, (long long)lds);
    GATMI_REQUIRE(!g16 || (ldg16 >= (int64_t)H * dp && ldg16 % 4 == 0), GATMI_EINVAL, "gat_bwd_node: row stride ldg16=%lld (>= hd, multiple of 4)", (long long)ldg16);
    a.g = g; a.ldg = ldg; a.g16 = reinterpret_cast<uint32_t *>(g16); a.ldg16 = ldg16; a.stats = stats; a.lds = lds; a.df1 = df1; a.ws = ws;
    a.activation = activation; a.reduce = reduce; a.slope = slope;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
#define X(gg, l) if (G == gg && LPH == l) return launch_bwd_node<gg, l>(a, dbias, st);
    GATMI_FOR_SHAPES(X)
#undef X
    return fail(GATMI_EUNSUPPORTED, "gat_bwd_node: no kernel for H=%d dp=%d", H, dp);
}

extern "C" size_t gatmi_gat_bwd_ws_floats(int64_t n_slots, int32_t H, int32_t dp) {
    if (n_slots <= 0) return 0;
    return (size_t)n_slots * (size_t)(H + H * dp);
}

extern "C" int gatmi_gat_bwd_csc(int64_t n_src, int32_t H, int32_t dp,
                                 const int32_t *items, int64_t n_items, const int32_t *split, int64_t n_split,
                                 const int32_t *rowidx, const int32_t *perm,
                                 const void *g, int32_t g_dtype, int64_t ldg, const float *stats, int64_t lds, const float *feat, const float *f2,
                                 const float *logit_a2, const float *logit_b2,
                                 const float *df1, const float *a1, const float *a2,
                                 float slope, float coef_keep, float feat_keep, uint64_t seed,
                                 float *dfeat, float *df2, float *ws, void *stream) {
    int G, LPH;
    int rc = check_shape(H, dp, &G, &LPH);
    if (rc) return rc;
    GATMI_REQUIRE(n_src >= 0 && n_items >= 0 && n_split >= 0, GATMI_EINVAL, "gat_bwd_csc: negative size");
    if (n_src == 0 || n_items == 0) return GATMI_OK;
    GATMI_REQUIRE(items && g && stats && feat && dfeat && df2, GATMI_EINVAL, "gat_bwd_csc: null pointer");
    GATMI_REQUIRE((f2 != nullptr) != (logit_a2 != nullptr), GATMI_EINVAL, "gat_bwd_csc: give either the table f2 or logit_a2 (f2 formed from the rows of feat), as the forward call did");
    GATMI_REQUIRE(logit_a2 || !logit_b2, GATMI_EINVAL, "gat_bwd_csc: logit_b2 goes with logit_a2");
    GATMI_REQUIRE(g_dtype != GATMI_DTYPE_BF16 || f2, GATMI_EINVAL, "gat_bwd_csc: with bf16 rows the source logit half comes from the table f2 (the forward pass read the same fp32 values)");
    GATMI_REQUIRE_ALIGNED(logit_a2);
    GATMI_REQUIRE((a1 == nullptr) == (a2 == nullptr) && (!a1 || df1), GATMI_EINVAL, "gat_bwd_csc: a1, a2 and df1 go together");
    GATMI_REQUIRE(n_split == 0 || (split && ws), GATMI_EINVAL, "gat_bwd_csc: split rows need `split` and `ws`");
    GATMI_REQUIRE(coef_keep > 0.f && coef_keep <= 1.f && feat_keep > 0.f && feat_keep <= 1.f, GATMI_EINVAL, "gat_bwd_csc: keep probabilities must be in (0,1]");
    GATMI_REQUIRE(coef_keep >= 1.f || perm, GATMI_EINVAL, "gat_bwd_csc: coefficient dropout needs `perm`");
    GATMI_REQUIRE_ALIGNED(items); GATMI_REQUIRE_ALIGNED(split); GATMI_REQUIRE_ALIGNED(g); GATMI_REQUIRE_ALIGNED(feat);
    GATMI_REQUIRE_ALIGNED(a1); GATMI_REQUIRE_ALIGNED(a2); GATMI_REQUIRE_ALIGNED(dfeat); GATMI_REQUIRE_ALIGNED(ws);
    BwdCscArgs a;
    a.items = reinterpret_cast<const int4 *>(items);
    a.n_items = n_items; a.rowidx = rowidx; a.perm = perm;
    GATMI_REQUIRE(ldg >= (int64_t)H * dp && ldg % 4 == 0 && lds >= 3 * (int64_t)H, GATMI_EINVAL,
                  "gat_bwd_csc: row strides ldg=%lld (>= hd, multiple of 4) lds=%lld (>= 3H)", (long long)ldg, (long long)lds);
    a.g = g; a.ldg = ldg; a.stats = stats; a.lds = lds; a.feat = feat; a.f2 = f2; a.la2 = logit_a2; a.lb2 = logit_b2; a.df1 = df1; a.a1 = a1; a.a2 = a2;
    a.dfeat = dfeat; a.df2 = df2; a.ws = ws;
    a.slope = slope; a.coef_keep = coef_keep; a.feat_keep = feat_keep; a.seed = seed; a.seed_step = (coef_keep < 1.f || feat_keep < 1.f) ? dropout_step_source() : nullptr;
    const bool drop = coef_keep < 1.f || feat_keep < 1.f;
    GATMI_REQUIRE(g_dtype == GATMI_DTYPE_F32 || g_dtype == GATMI_DTYPE_BF16, GATMI_EUNSUPPORTED, "gat_bwd_csc: g_dtype %d", g_dtype);
    const bool bf16 = g_dtype == GATMI_DTYPE_BF16;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const int4 *sp = reinterpret_cast<const int4 *>(split);
#define X(gg, l) if (G == gg && LPH == l) return launch_bwd_csc<gg, l>(a, sp, n_split, drop, bf16, st);
    GATMI_FOR_SHAPES(X)
#undef X
    return fail(GATMI_EUNSUPPORTED, "gat_bwd_csc: no kernel for H=%d dp=%d", H, dp);
}
