// Batched dense masked graph attention for SPARSE masks (gfx950): the layer of dense_attn.hip,
//     alpha_ij = softmax_j( LeakyReLU(f1[i,h] + f2[j,h]) + bias_mat[b,i,j] ),  out_i = sum_j alpha_ij Wh_j,
// computed by visiting only the entries of bias_mat that can contribute.  adj_to_bias masks
// (utils/process.py:14-25) are 0 on a graph's edges and -1e9 elsewhere; at a density of 10 % nine
// tenths of the (i, j, head) work of the matrix-core kernels multiplies zeros.  Here a wave reads
// the mask rows (coalesced, once from HBM), compacts the contributing column indices into LDS with
// wave ballots, and then runs the row kernels of gat_sparse.hip's design on the compacted lists:
// G = hd/4 lanes per row, online softmax four neighbours at a time, Wh rows gathered from L2 (a
// graph's Wh is a few tens of KB).
//
// Which entries contribute: entry (i, j) is visited iff bias_mat[i,j] > max_j' bias_mat[i,j'] - 1e6.
// An entry below that line has a softmax numerator exp(logit_ij - 1e6 - (row maximum)) = 0 in fp32
// (and in fp64) for any logits below 4e5, so skipping it changes nothing; the bias values of the
// visited entries are added to the logits as in the dense kernels, so soft masks are exact too,
// and a row whose entries are all -1e9 keeps all of them (uniform attention, as the dense kernels
// give).
//
// In training the forward kernel also writes which entries it visited: one bit per entry, W =
// ceil(N/32) + 1 words per row, the last word holding bit 0 = "some visited entry of this row has a
// non-zero bias".  The backward pass never reads the fp32 mask of an adj_to_bias batch again: block
// (graph b, 32 source columns) reads one word per destination row, turns the 32 bit columns into
// row-index lists in LDS with 32 wave ballots per 64 rows (lists of a column are filled in row
// order, so every sum runs in a fixed order), and G lanes per column walk the lists like
// gat_bwd_csc_kernel walks CSC rows, fetching a bias value only for rows flagged as above.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>

#include "common.h"
#include "device_common.h"

namespace gatmi {

constexpr float kMaskCutoffFwd = 1.0e6f;
constexpr int kListBytesPerWave = 4096;   // forward: LDS for the compacted lists of one wave's rows
constexpr int kBwdCols = 32;              // backward: source columns per block = bits of one word
constexpr int kBwdChunks = 8;             // backward: 64-row chunks listed per pass
constexpr int kMaskedMaxN = 32767;        // list entries are 15-bit node ids + 1 flag bit

struct MaskedFwdArgs {
    int64_t B;
    int N;
    const float *Wh, *f1, *f2, *bias_mat, *bias, *resid;
    float *out, *agg, *pos, *lse, *q;
    uint32_t *present;         // [B*N, W] visited-entry bits + flag word, or nullptr
    int activation;
    float slope, coef_keep, feat_keep;
    uint64_t seed;
    const uint64_t *seed_step;   // device word added to the seed at run time (seed_now), or nullptr
    // prebuilt lists of a mask that does not change from step to step (dense_graph_fwd8_kernel<.., LISTS>):
    // row r of graph b visits columns lcol[lptr[b*N + r] .. lptr[b*N + r + 1]), padded to whole batches with the
    // dummy id N; lord[b*N + k] = the graph's row with the k-th longest list
    const int32_t *lptr;
    const uint16_t *lcol, *lord;
    int lcap;                  // list entries of one graph the kernel's LDS area holds
};

struct MRowState {
    float m, s, q;
    float4 acc, pos;
};

#ifndef GATMI_MASK_BATCH
#define GATMI_MASK_BATCH 8
#endif
#ifndef GATMI_MASK_MINBLOCKS
#define GATMI_MASK_MINBLOCKS 4
#endif
constexpr int kBatch = GATMI_MASK_BATCH;  // list entries per step of a row / column walk (independent gathers in flight)

// ids[t] = t-th of NB consecutive uint16 list entries at lp (16-byte aligned for 8, 8-byte for 4)
template <int NB>
__device__ __forceinline__ void load_ids(const uint16_t *lp, int (&ids)[NB]) {
    static_assert(NB == 4 || NB == 8, "four or eight 16-bit ids per step");
    uint32_t w[4];
    if constexpr (NB == 8) {
        const uint4 packed = *reinterpret_cast<const uint4 *>(lp);
        w[0] = packed.x; w[1] = packed.y; w[2] = packed.z; w[3] = packed.w;
    } else {
        const uint2 packed = *reinterpret_cast<const uint2 *>(lp);
        w[0] = packed.x; w[1] = packed.y; w[2] = w[3] = 0;
    }
#pragma unroll
    for (int t = 0; t < NB; ++t) ids[t] = (w[t >> 1] >> (16 * (t & 1))) & 0xffff;
}

// One batch of up to kBatch listed sources of one destination row.
// whb / f2b: this lane's float4 column of the graph's Wh rows / this lane's head of its f2 rows, so the
// gathers need one 32-bit multiply-add per address (row ids of one graph are below 2^15).
template <int G, int LPH, bool SAVE, bool DROP>
__device__ __forceinline__ void masked_fwd_batch(MRowState &st, const MaskedFwdArgs &a, const uint16_t *lp, int nvalid,
                                                 int64_t b, int i, const float *bm, bool soft, const float *whb, const float *f2b,
                                                 float f1v, int sub, int head) {
    constexpr int H = G / LPH;
    constexpr int HD = 4 * G;
    constexpr int U = kBatch;
    const int N = a.N;
    int j[U];
    float z[U], e[U];
    float4 v[U];
    load_ids<U>(lp, j);
#pragma unroll
    for (int t = 0; t < U; ++t) {
        if (t >= nvalid) j[t] = j[0];
        z[t] = f1v + f2b[(uint32_t)j[t] * H];
        v[t] = *reinterpret_cast<const float4 *>(whb + (uint32_t)j[t] * HD);
        e[t] = soft ? bm[(uint32_t)j[t]] : 0.f;          // rows whose visited entries are all 0 need no bias values
    }
    float mn = st.m;
#pragma unroll
    for (int t = 0; t < U; ++t) {
        e[t] = t < nvalid ? leaky(z[t], a.slope) + e[t] : -INFINITY;
        mn = fmaxf(mn, e[t]);
    }
    const float sc = exp_neg(st.m - mn);
    float bs = 0.f, bq = 0.f;
    float4 bacc = make_float4(0.f, 0.f, 0.f, 0.f), bpos = bacc;
#pragma unroll
    for (int t = 0; t < U; ++t) {
        const float p = exp_neg(e[t] - mn);
        bs += p;
        float pa = p;
        float4 vv = v[t];
        if (DROP) {
            if (a.coef_keep < 1.f) pa = p * coef_drop_scale(seed_now(a), b, N, H, i, j[t], head, a.coef_keep);
            if (a.feat_keep < 1.f) vv = drop_feat(vv, seed_now(a), b * N + j[t], HD, sub * 4, a.feat_keep);
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

// Block (graph b, 4 * 64/G destination rows): wave w owns rows base + w*RPW .. + RPW-1.
template <int G, int LPH, bool SAVE, bool DROP>
__global__ __launch_bounds__(256, GATMI_MASK_MINBLOCKS) void dense_masked_fwd_kernel(MaskedFwdArgs a) {
    constexpr int RPW = 64 / G;
    constexpr int H = G / LPH;
    constexpr int HD = 4 * G;
    constexpr int LC = kListBytesPerWave / 2 / RPW;      // list capacity per row = mask entries compacted per pass
    __shared__ __attribute__((aligned(16))) uint16_t lists[4][RPW][LC];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int sub = lane % G, grp = lane / G;
    const int head = sub / LPH;
    const int N = a.N;
    const int64_t b = blockIdx.x;
    const int row0 = (blockIdx.y * 4 + wave) * RPW;
    if (row0 >= N) return;                               // whole wave past the graph (no block-wide barrier below)
    const int i = min(row0 + grp, N - 1);                // this lane group's destination row
    const bool row_ok = row0 + grp < N;
    const float *bmat = a.bias_mat + b * N * (int64_t)N;
    const float f1v = a.f1[(b * N + i) * H + head];
    const int W = (N + 31) / 32 + 1;                     // words per row of a.present

    MRowState st;
    st.m = -INFINITY;
    st.s = st.q = 0.f;
    st.acc = st.pos = make_float4(0.f, 0.f, 0.f, 0.f);
    const float *bm_own = bmat + (int64_t)i * N;
    const float *whb = a.Wh + b * N * (int64_t)HD + sub * 4;
    const float *f2b = a.f2 + b * N * (int64_t)H + head;
    const uint64_t lt_mask = lane == 0 ? 0ull : (~0ull >> (64 - lane));
    constexpr int U = 2;                                 // 16-byte loads per lane and row in the register-resident path
    if ((N & 3) == 0 && N <= 256 * U && N <= LC) {
        // Small graphs: every mask row of the wave is loaded once, 16 bytes per lane and all loads in
        // flight together; row maxima and compaction work from registers.
        float4 mv[RPW][U];
#pragma unroll
        for (int rr = 0; rr < RPW; ++rr) {
            const float *bm = bmat + (int64_t)min(row0 + rr, N - 1) * N;
#pragma unroll
            for (int u = 0; u < U; ++u) {
                mv[rr][u] = make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
                if (N - 256 * u > 64) {                    // a 256-column block: four columns per lane
                    const int j = 4 * lane + 256 * u;
                    if (j < N) mv[rr][u] = *reinterpret_cast<const float4 *>(bm + j);
                } else if (256 * u + lane < N) {           // a tail of at most 64 columns: one per lane, in .x
                    mv[rr][u].x = bm[256 * u + lane];
                }
            }
        }
        int mycnt = 0;
        bool mysoft = false;
#pragma unroll
        for (int rr = 0; rr < RPW; ++rr) {
            float mx = -INFINITY;
#pragma unroll
            for (int u = 0; u < U; ++u) mx = fmaxf(fmaxf(mx, fmaxf(mv[rr][u].x, mv[rr][u].y)), fmaxf(mv[rr][u].z, mv[rr][u].w));
#pragma unroll
            for (int o = 32; o; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
            const float cut = mx - kMaskCutoffFwd;
            int cnt = 0;
            bool nonzero = false;
            uint32_t *prow = a.present ? a.present + (b * N + min(row0 + rr, N - 1)) * (int64_t)W : nullptr;
#pragma unroll
            for (int u = 0; u < U; ++u) {
                if (256 * u >= N) break;
                if (N - 256 * u <= 64) {                   // tail: one ballot covers it, and is its two bit words
                    const bool present = mv[rr][u].x > cut;
                    const uint64_t bal = __ballot(present);
                    if (present) lists[wave][rr][cnt + __popcll(bal & lt_mask)] = (uint16_t)(256 * u + lane);
                    cnt += __popcll(bal);
                    nonzero |= present && mv[rr][u].x != 0.f;
                    if (prow && lane < 2 && 8 * u + lane < W - 1) prow[8 * u + lane] = (uint32_t)(bal >> (32 * lane));
                    break;
                }
                const float comp[4] = {mv[rr][u].x, mv[rr][u].y, mv[rr][u].z, mv[rr][u].w};
                uint32_t nib = 0;
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    const bool present = comp[t] > cut;
                    const uint64_t bal = __ballot(present);
                    if (present) lists[wave][rr][cnt + __popcll(bal & lt_mask)] = (uint16_t)(4 * lane + 256 * u + t);
                    cnt += __popcll(bal);
                    nib |= present ? (1u << t) : 0u;
                    nonzero |= present && comp[t] != 0.f;
                }
                if (prow) {                               // word 8u + lane/8 holds columns 4*lane+256u .. of 8 adjacent lanes
                    uint32_t w = nib << (4 * (lane & 7));
                    w |= __shfl_xor(w, 1, 64);
                    w |= __shfl_xor(w, 2, 64);
                    w |= __shfl_xor(w, 4, 64);
                    const int wi = 8 * u + (lane >> 3);
                    if ((lane & 7) == 0 && wi < W - 1) prow[wi] = w;
                }
            }
            const uint64_t nzb = __ballot(nonzero);
            if (prow && lane == 0) prow[W - 1] = nzb != 0 ? 1u : 0u;
            if (rr == grp) {
                mycnt = cnt;
                mysoft = nzb != 0;
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        if (row_ok) {
            const uint16_t *lp = lists[wave][grp];
            for (int k = 0; k < mycnt; k += kBatch)
                masked_fwd_batch<G, LPH, SAVE, DROP>(st, a, lp + k, min(kBatch, mycnt - k), b, i, bm_own, mysoft, whb, f2b, f1v, sub, head);
        }
    } else {
        // General sizes: row maxima first (one row of the wave at a time with all 64 lanes), then the
        // rows are compacted and consumed in passes of LC mask entries; the second read hits L2.
        float bmax = -INFINITY;
        for (int rr = 0; rr < RPW; ++rr) {
            const float *bm = bmat + (int64_t)min(row0 + rr, N - 1) * N;
            float mx0 = -INFINITY, mx1 = -INFINITY, mx2 = -INFINITY, mx3 = -INFINITY;
            int j = lane;
            for (; j + 192 < N; j += 256) {                  // four independent loads in flight
                mx0 = fmaxf(mx0, bm[j]); mx1 = fmaxf(mx1, bm[j + 64]); mx2 = fmaxf(mx2, bm[j + 128]); mx3 = fmaxf(mx3, bm[j + 192]);
            }
            for (; j < N; j += 64) mx0 = fmaxf(mx0, bm[j]);
            float mx = fmaxf(fmaxf(mx0, mx1), fmaxf(mx2, mx3));
#pragma unroll
            for (int o = 32; o; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
            if (rr == grp) bmax = mx;
        }
        for (int seg0 = 0; seg0 < N; seg0 += LC) {
            const int seg_end = min(N, seg0 + LC);
            int mycnt = 0;
            bool mysoft = false;
            for (int rr = 0; rr < RPW; ++rr) {               // compact the contributing columns of row row0 + rr
                const float *bm = bmat + (int64_t)min(row0 + rr, N - 1) * N;
                const float cut = __shfl(bmax, rr * G, 64) - kMaskCutoffFwd;
                uint32_t *prow = a.present ? a.present + (b * N + min(row0 + rr, N - 1)) * (int64_t)W : nullptr;
                int cnt = 0;
                uint64_t nzb = 0;
                for (int j0 = seg0; j0 < seg_end; j0 += 64) {   // seg0 and LC are multiples of 64: a ballot is two words
                    const int j = j0 + lane;
                    const float bvj = j < seg_end ? bm[j] : -INFINITY;
                    const bool present = bvj > cut;
                    const uint64_t bal = __ballot(present);
                    if (present) lists[wave][rr][cnt + __popcll(bal & lt_mask)] = (uint16_t)j;
                    cnt += __popcll(bal);
                    nzb |= __ballot(present && bvj != 0.f);
                    if (prow && lane < 2 && (j0 >> 5) + lane < W - 1) prow[(j0 >> 5) + lane] = (uint32_t)(bal >> (32 * lane));
                }
                if (prow && lane == 0) {                        // flag word: OR over the passes
                    const uint32_t f = nzb != 0 ? 1u : 0u;
                    prow[W - 1] = seg0 == 0 ? f : (prow[W - 1] | f);
                }
                if (rr == grp) {
                    mycnt = cnt;
                    mysoft = nzb != 0;
                }
            }
            // the lists are written and read by this wave only; its lanes run in lockstep
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            if (row_ok) {
                const uint16_t *lp = lists[wave][grp];
                for (int k = 0; k < mycnt; k += kBatch)
                    masked_fwd_batch<G, LPH, SAVE, DROP>(st, a, lp + k, min(kBatch, mycnt - k), b, i, bm_own, mysoft, whb, f2b, f1v, sub, head);
            }
            __builtin_amdgcn_wave_barrier();                 // all reads done before the next pass overwrites the lists
        }
    }
    if (!row_ok) return;
    const int64_t row = b * N + i;
    const bool empty = !(st.s > 0.f);
    const float inv = empty ? 0.f : 1.0f / st.s;
    const float4 ag = f4_scale(inv, st.acc);
    if (SAVE) {
        *reinterpret_cast<float4 *>(a.agg + row * HD + sub * 4) = ag;
        *reinterpret_cast<float4 *>(a.pos + row * HD + sub * 4) = f4_scale(inv, st.pos);
        if (sub % LPH == 0) {
            a.lse[row * H + head] = empty ? 0.f : st.m + logf(st.s);
            a.q[row * H + head] = st.q * inv;
        }
    }
    float4 r = ag;
    if (a.bias) r = f4_add(r, *reinterpret_cast<const float4 *>(a.bias + sub * 4));
    if (a.resid) r = f4_add(r, *reinterpret_cast<const float4 *>(a.resid + row * HD + sub * 4));
    if (a.activation != GATMI_ACT_IDENTITY) r = make_float4(act_f(r.x, a.activation), act_f(r.y, a.activation), act_f(r.z, a.activation), act_f(r.w, a.activation));
    *reinterpret_cast<float4 *>(a.out + row * HD + sub * 4) = r;
}


// ------------------------------------------------------------------------------------------
// forward, graph-resident form: one block per graph, the graph's Wh rows and f2 values staged in
// LDS once, so the per-entry gathers of the list walk are LDS reads (256 B per clock and CU) instead
// of L2 reads.  For batches of many small graphs (the block count is the batch size) whose rows are
// a multiple of 16 bytes; the mask handling and the lists are those of the register-resident path of
// dense_masked_fwd_kernel, and every wave walks row quads wave, wave + nwaves, ... on its own (no
// block-wide barrier after the staging).
// ------------------------------------------------------------------------------------------
constexpr int kGraphBatch = 4;             // list entries per step of the LDS walks (short latency: four in flight suffice)
constexpr int kGraphPadRows = 4;           // LDS rows after the graph's own: row N is the dummy row the list padding names

// One batch of kGraphBatch listed sources of one destination row, gathered from LDS.  The lists are padded
// to a whole batch with the id of a dummy row (Wh = 0, f2 = -inf: coefficient exactly 0), so no entry
// needs a validity test.  SOFT: the visited entries of some row of the wave carry non-zero biases.
template <int G, int LPH, bool SAVE, bool SOFT>
__device__ __forceinline__ void graph_fwd_batch(MRowState &st, float slope, const uint16_t *lp, const float *bm, int N,
                                                const float *whs, const float *f2s, float f1v) {
    constexpr int H = G / LPH;
    constexpr int HD = 4 * G;
    constexpr int U = kGraphBatch;
    int j[U];
    float z[U], e[U];
    float4 v[U];
    load_ids<U>(lp, j);
#pragma unroll
    for (int t = 0; t < U; ++t) {
        z[t] = f1v + f2s[j[t] * H];
        v[t] = *reinterpret_cast<const float4 *>(whs + j[t] * HD);
        if (SOFT) e[t] = bm[min(j[t], N - 1)];
    }
    float mn = st.m;
#pragma unroll
    for (int t = 0; t < U; ++t) {
        e[t] = SOFT ? leaky(z[t], slope) + e[t] : leaky(z[t], slope);
        mn = fmaxf(mn, e[t]);
    }
    const float sc = exp_neg(st.m - mn);
    float bs = 0.f, bq = 0.f;
    float4 bacc = make_float4(0.f, 0.f, 0.f, 0.f), bpos = bacc;
#pragma unroll
    for (int t = 0; t < U; ++t) {
        const float p = exp_neg(e[t] - mn);
        bs += p;
        bacc = f4_fma(p, v[t], bacc);
        if (SAVE) {
            const float pp = z[t] > 0.f ? p : 0.f;
            bq += pp;
            bpos = f4_fma(pp, v[t], bpos);
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

constexpr int kGraphFwdThreads = 1024;
// LDS one workgroup of the graph-resident kernels may ask for: what the current device grants a single
// block (all 160 KB of a CU on gfx950), looked up once per device.  The gates below compare against it, so
// that on a part with less LDS the shapes that do not fit go to the row-block kernels instead of failing
// in hipFuncSetAttribute.
static size_t graph_lds_limit() {
    constexpr int kDevs = 64;
    static std::atomic<size_t> cached[kDevs] = {};
    int dev = -1;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kDevs) return 64 * 1024;
    size_t v = cached[dev].load(std::memory_order_acquire);
    if (v == 0) {
        int bytes = 0;
        if (hipDeviceGetAttribute(&bytes, hipDeviceAttributeMaxSharedMemoryPerBlock, dev) != hipSuccess || bytes <= 0) bytes = 64 * 1024;
        v = (size_t)bytes;
        cached[dev].store(v, std::memory_order_release);
    }
    return v;
}

// list capacity per row: every column, plus one batch of padding, as whole batches
static inline int graph_list_capacity(int N) { return (N + 2 * kGraphBatch - 1) / kGraphBatch * kGraphBatch; }

// LDS of the graph-resident forward kernel: Wh [N + pad][HD] | f2 [N + pad][H] | lists [waves][RPW][lc] (uint16)
static inline size_t graph_fwd_lds(int N, int G, int H, int lc) {
    return (size_t)(N + kGraphPadRows) * (4 * G + H) * sizeof(float) + (size_t)(kGraphFwdThreads / 64) * (64 / G) * lc * sizeof(uint16_t);
}

// Mask row of one lane group in registers: lane `sub` of the group holds columns 64k + 4 sub .. + 3 of
// the group's own row for k < kGraphK (a group's 16 lanes cover 256 contiguous bytes per k); columns
// beyond the row are -inf.  A group of 16 lanes is one DPP row, so the row maximum, the prefix sum of
// the lanes' counts and the bit words below are DPP steps for all rows of the wave at once.
constexpr int kGraphK = 5;                 // 5 x 64 columns: graphs of at most 320 nodes
struct MaskRow {
    float4 v[kGraphK];
};

__device__ __forceinline__ void load_mask_row(MaskRow &m, const float *bm, int N, int sub) {
#pragma unroll
    // Every load is issued whatever N is (columns past the row read its last four, which the caches hold, and are
    // replaced by -inf): with a load under a branch the compiler can no longer count what is in flight, and the
    // next wait for anything becomes a wait for these rows, which are meant to travel during the current rows' work.
    // The replacement happens where the row is used (mask_row_trim), not here: a select right behind the load
    // lets the compiler put the load back under a branch.
    for (int k = 0; k < kGraphK; ++k)
        m.v[k] = *reinterpret_cast<const float4 *>(bm + min(64 * k + 4 * sub, N - 4));      // N % 4 == 0: in bounds
}

// the columns past the row's end become -inf (coefficient exactly 0, never listed)
__device__ __forceinline__ MaskRow mask_row_trim(const MaskRow &m, int N, int sub) {
    MaskRow r;
#pragma unroll
    for (int k = 0; k < kGraphK; ++k) r.v[k] = 64 * k + 4 * sub < N ? m.v[k] : make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
    return r;
}

// maximum / sum over the 16 lanes of a DPP row, in every lane of the row
__device__ __forceinline__ float row16_max(float v) {
    v = fmaxf(v, dpp_f<0xB1>(v, v));
    v = fmaxf(v, dpp_f<0x4E>(v, v));
    v = fmaxf(v, dpp_f<0x141>(v, v));
    return fmaxf(v, dpp_f<0x140>(v, v));
}
__device__ __forceinline__ int row16_sum(int v) {
    v += dpp_i<0xB1>(0, v);
    v += dpp_i<0x4E>(0, v);
    v += dpp_i<0x141>(0, v);
    return v + dpp_i<0x140>(0, v);
}
// inclusive prefix sum inside each DPP row
__device__ __forceinline__ int row16_scan(int v) {
    v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xF, 0xF, true);
    v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xF, 0xF, true);
    v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xF, 0xF, true);
    return v + __builtin_amdgcn_update_dpp(0, v, 0x118, 0xF, 0xF, true);
}

// Rows of 64 floats only (16 lanes per row = one DPP row): G is fixed, LPH selects the head layout.
template <int LPH, bool SAVE>
__global__ __launch_bounds__(kGraphFwdThreads) void dense_graph_fwd_kernel(MaskedFwdArgs a, int lc) {
    constexpr int G = 16;
    constexpr int RPW = 64 / G;
    constexpr int H = G / LPH;
    constexpr int HD = 4 * G;
    extern __shared__ __attribute__((aligned(16))) unsigned char graph_smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int nwaves = blockDim.x >> 6;
    const int sub = lane % G, grp = lane / G;
    const int head = sub / LPH;
    const int N = a.N;
    const int NP = N + kGraphPadRows;
    const int64_t b = blockIdx.x;
    float *whs = reinterpret_cast<float *>(graph_smem);
    float *f2s = whs + NP * HD;
    uint16_t *lrow = reinterpret_cast<uint16_t *>(f2s + NP * H) + (wave * RPW + grp) * lc;     // this group's list; NP % 4 == 0: 16-byte aligned
    const float *bmat = a.bias_mat + b * N * (int64_t)N;
    MaskRow next;
    int row0 = wave * RPW;
    if (row0 < N) load_mask_row(next, bmat + min(row0 + grp, N - 1) * N, N, sub);          // in flight during the staging
    {
        const float4 *src = reinterpret_cast<const float4 *>(a.Wh + b * N * (int64_t)HD);
        for (int e = threadIdx.x; e < NP * (HD / 4); e += blockDim.x)
            reinterpret_cast<float4 *>(whs)[e] = e < N * (HD / 4) ? src[e] : make_float4(0.f, 0.f, 0.f, 0.f);
        const float *src2 = a.f2 + b * N * (int64_t)H;
        for (int e = threadIdx.x; e < NP * H; e += blockDim.x) f2s[e] = e < N * H ? src2[e] : -INFINITY;
    }
    __syncthreads();
    const int W = (N + 31) / 32 + 1;
    uint32_t *present_b = a.present ? a.present + b * N * (int64_t)W : nullptr;
    const float *whl = whs + sub * 4;
    const float *f2l = f2s + head;
    for (; row0 < N; row0 += nwaves * RPW) {
        const MaskRow cur = mask_row_trim(next, N, sub);
        if (row0 + nwaves * RPW < N)                     // the next rows travel during this quad's work
            load_mask_row(next, bmat + min(row0 + nwaves * RPW + grp, N - 1) * N, N, sub);
        const int i = min(row0 + grp, N - 1);
        const bool row_ok = row0 + grp < N;
        const float f1v = a.f1[(b * N + i) * H + head];
        float mx = -INFINITY;
#pragma unroll
        for (int k = 0; k < kGraphK; ++k) mx = raw_max3(mx, raw_max(cur.v[k].x, cur.v[k].y), raw_max(cur.v[k].z, cur.v[k].w));
        mx = row16_max(mx);
        const float cut = mx - kMaskCutoffFwd;
        // bit 4k + c of `bits`: column 64k + 4 sub + c is visited.  A visited entry is non-zero iff the row's
        // maximum is non-zero (it is visited itself) or the entry lies strictly between the cut and zero.
        uint32_t bits = 0;
        bool nonzero = mx != 0.f;
#pragma unroll
        for (int k = 0; k < kGraphK; ++k) {
            const float comp[4] = {cur.v[k].x, cur.v[k].y, cur.v[k].z, cur.v[k].w};
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const bool present = comp[c] > cut;
                bits |= present ? (1u << (4 * k + c)) : 0u;
                nonzero |= present && comp[c] < 0.f;
            }
        }
        const int cnt = __popc(bits);
        const int total = row16_sum(cnt);
        int pos = row16_scan(cnt) - cnt;
        const uint64_t nzb = __ballot(nonzero && bits != 0);
        const bool anysoft = nzb != 0;
        if (present_b && row_ok) {
            uint32_t *prow = present_b + i * W;
#pragma unroll
            for (int k = 0; k < kGraphK; ++k) {
                if (64 * k < N) {                         // words 2k and 2k + 1: columns of 8 adjacent lanes each
                    const uint32_t w = or8_to_first(((bits >> (4 * k)) & 15u) << (4 * (sub & 7)));
                    const int wi = 2 * k + (sub >> 3);
                    if ((sub & 7) == 0 && wi < W - 1) prow[wi] = w;
                }
            }
            if (sub == 0) prow[W - 1] = ((nzb >> (16 * grp)) & 0xffffull) != 0 ? 1u : 0u;
        }
        for (uint32_t rest = bits; __ballot(rest != 0) != 0;) {          // each lane appends its columns at its offset
            if (rest != 0) {
                const int t = __ffs(rest) - 1;
                lrow[pos++] = (uint16_t)(64 * (t >> 2) + 4 * sub + (t & 3));
                rest &= rest - 1;
            }
        }
        if (sub < kGraphBatch) lrow[total + sub] = (uint16_t)N;          // padding: the dummy row
        // the lists are written and read by this wave only; its lanes run in lockstep
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        MRowState st;
        st.m = -INFINITY;
        st.s = st.q = 0.f;
        st.acc = st.pos = make_float4(0.f, 0.f, 0.f, 0.f);
        if (row_ok) {
            const float *bm_own = bmat + i * N;
            if (anysoft) {                                // wave-uniform: biases are fetched for all of the wave's rows
                for (int k = 0; k < total; k += kGraphBatch)
                    graph_fwd_batch<G, LPH, SAVE, true>(st, a.slope, lrow + k, bm_own, N, whl, f2l, f1v);
            } else {
                for (int k = 0; k < total; k += kGraphBatch)
                    graph_fwd_batch<G, LPH, SAVE, false>(st, a.slope, lrow + k, bm_own, N, whl, f2l, f1v);
            }
        }
        __builtin_amdgcn_wave_barrier();                 // every lane is done with the lists before the next rows overwrite them
        if (row_ok) {
            const int64_t row = b * N + i;
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
            if (a.resid) r = f4_add(r, *reinterpret_cast<const float4 *>(a.resid + row * HD + sub * 4));
            if (a.activation != GATMI_ACT_IDENTITY) r = make_float4(act_f(r.x, a.activation), act_f(r.y, a.activation), act_f(r.z, a.activation), act_f(r.w, a.activation));
            store_stream4(a.out + row * HD + sub * 4, r);
        }
    }
}

// ------------------------------------------------------------------------------------------
// The same kernel with the walk on eight lanes per row.  Compaction stays as above (16 lanes per row,
// one DPP row each, four rows per wave), but a wave compacts TWO row quads before it walks, and in the
// walk a row belongs to eight lanes of eight floats each.  For heads of 8 floats or more a lane then
// holds whole heads' worth of one head only, so the per-head part of an entry's work (logit, LeakyReLU,
// running maximum, exponentials: half of the walk's instructions) is no longer done twice by the two
// float4 lanes of a head, and eight rows share every instruction instead of four.
// LPH8 = lanes per head in the walk = dp / 8.
// ------------------------------------------------------------------------------------------
struct MRowState8 {
    float m, s, q;
    float4 acc0, acc1, pos0, pos1;
};

// DROP: coefficient dropout (utils/layers.py:20) per (entry, head) from the counter hash; `cbase` is the hash index of
// column 0 of this lane's (row, head), an entry's index is cbase + j * H.  Feature dropout (:22) needs nothing here:
// the rows in LDS were dropped when they were staged.
// wha / whb: this lane's two float4 pieces of row 0 (acc0 / pos0 belong to the first, acc1 / pos1 to the second).
// Which piece is read first can alternate between neighbouring rows of the octet (walk_pieces): a ds_read_b128
// serves sixteen lanes at a time, among them the same four lane slots of two different rows, and with one order for
// all their 16-byte pieces fall on the same banks (every row starts at bank 0): two LDS cycles instead of one.
template <int LPH8, bool SAVE, bool SOFT, bool DROP>
__device__ __forceinline__ void graph_fwd_batch8_ids(MRowState8 &st, float slope, const int (&j)[kGraphBatch], const float *bm, int N,
                                                     const float *wha, const float *whb, const float *f2s, float f1v,
                                                     uint64_t seed, uint64_t cbase, float coef_keep) {
    constexpr int H = 8 / LPH8;
    constexpr int HD = 64;
    constexpr int U = kGraphBatch;
    float z[U], e[U];
    float4 v0[U], v1[U];
#pragma unroll
    for (int t = 0; t < U; ++t) {
        z[t] = f1v + f2s[j[t] * H];
        v0[t] = *reinterpret_cast<const float4 *>(wha + j[t] * HD);
        v1[t] = *reinterpret_cast<const float4 *>(whb + j[t] * HD);
        if (SOFT) e[t] = bm[min(j[t], N - 1)];
    }
    float mn = st.m;
#pragma unroll
    for (int t = 0; t < U; ++t) {
        e[t] = SOFT ? leaky(z[t], slope) + e[t] : leaky(z[t], slope);
        mn = fmaxf(mn, e[t]);
    }
    const float sc = exp_neg(st.m - mn);
    float bs = 0.f, bq = 0.f;
    const float4 zero = make_float4(0.f, 0.f, 0.f, 0.f);
    float4 bacc0 = zero, bacc1 = zero, bpos0 = zero, bpos1 = zero;
    const float inv_keep = DROP ? 1.0f / coef_keep : 1.0f;
#pragma unroll
    for (int t = 0; t < U; ++t) {
        const float p = exp_neg(e[t] - mn);
        bs += p;
        float pa = p;
        if (DROP) {
            if (coef_keep < 1.f) pa = keep_mask(seed, cbase + (uint64_t)(uint32_t)j[t] * H, coef_keep) ? p * inv_keep : 0.f;
        }
        bacc0 = f4_fma(pa, v0[t], bacc0);
        bacc1 = f4_fma(pa, v1[t], bacc1);
        if (SAVE) {
            const bool ps = z[t] > 0.f;
            bq += ps ? p : 0.f;
            const float pp = ps ? pa : 0.f;
            bpos0 = f4_fma(pp, v0[t], bpos0);
            bpos1 = f4_fma(pp, v1[t], bpos1);
        }
    }
    st.s = fmaf(st.s, sc, bs);
    st.acc0 = f4_fma(sc, st.acc0, bacc0);
    st.acc1 = f4_fma(sc, st.acc1, bacc1);
    if (SAVE) {
        st.q = fmaf(st.q, sc, bq);
        st.pos0 = f4_fma(sc, st.pos0, bpos0);
        st.pos1 = f4_fma(sc, st.pos1, bpos1);
    }
    st.m = mn;
}

template <int LPH8, bool SAVE, bool SOFT, bool DROP>
__device__ __forceinline__ void graph_fwd_batch8(MRowState8 &st, float slope, const uint16_t *lp, const float *bm, int N,
                                                 const float *wha, const float *whb, const float *f2s, float f1v,
                                                 uint64_t seed, uint64_t cbase, float coef_keep) {
    int j[kGraphBatch];
    load_ids<kGraphBatch>(lp, j);
    graph_fwd_batch8_ids<LPH8, SAVE, SOFT, DROP>(st, slope, j, bm, N, wha, whb, f2s, f1v, seed, cbase, coef_keep);
}

__device__ __forceinline__ void row_state_reset(MRowState8 &st, float m) {
    st.m = m;
    st.s = st.q = 0.f;
    st.acc0 = st.acc1 = st.pos0 = st.pos1 = make_float4(0.f, 0.f, 0.f, 0.f);
}

// Offsets (in floats, within a lane's eight) of the piece a lane of the walk reads first and second: lanes whose
// bit `bit` is set take the upper piece first.  bit 3 for eight lanes per row (the rows of an octet alternate),
// bit 2 for four lanes per column.
// Measured at 4096 graphs x 264 nodes: bank conflicts fall by four fifths (forward) and a quarter (half-row
// backward), the second address costs 6 % more vector instructions, and the kernels take the same time or 3 % more:
// the LDS array is not what they wait for.  Off; the walks then read one address with an immediate offset.
constexpr bool kAlternatePieces = false;
__device__ __forceinline__ void walk_pieces(int lane, int bit, int &first, int &second) {
    first = kAlternatePieces ? ((lane >> bit) & 1) * 4 : 0;
    second = 4 - first;
}

constexpr int kGraphWalkRows = 8;          // rows per wave in the eight-lane walk
constexpr int kGraph8MaxWaves = 12;        // three waves per SIMD: 168 registers per lane (two mask rows + the walk state)

// waves per block: as few as finish the graph's row octets in the same number of rounds as the maximum would
static inline int graph8_waves(int N) {
    const int octs = (N + kGraphWalkRows - 1) / kGraphWalkRows;
    const int rounds = (octs + kGraph8MaxWaves - 1) / kGraph8MaxWaves;
    return (octs + rounds - 1) / rounds;
}
// LDS: Wh [N + pad][64] | f2 [N + pad][H] | lists [waves][8][lc] (uint16) | entry counts [waves][8] (int)
static inline size_t graph8_fwd_lds(int N, int H, int lc, int waves) {
    return (size_t)(N + kGraphPadRows) * (64 + H) * sizeof(float) + (size_t)waves * kGraphWalkRows * lc * sizeof(uint16_t) +
           (size_t)waves * kGraphWalkRows * sizeof(int);
}
// LDS with prebuilt lists: Wh | f2 | list offsets [N + 1 -> even] (int) | row order [N -> multiple of 4] (uint16) | `cap` list entries (uint16)
static inline size_t graph8_lists_head_lds(int N) { return (size_t)((N + 2) & ~1) * sizeof(int) + (size_t)((N + 3) & ~3) * sizeof(uint16_t); }
static inline size_t graph8_fwd_lists_lds(int N, int H, int cap) {
    return (size_t)(N + kGraphPadRows) * (64 + H) * sizeof(float) + graph8_lists_head_lds(N) + (size_t)cap * sizeof(uint16_t);
}

// Stage one graph's prebuilt lists: entries [ptr[b*N], ptr[(b+1)*N]) of `ids` (whole batches of four: 8-byte pieces),
// the rows' offsets relative to the graph's first entry, and the graph's row order.  Offsets are clamped to `cap`, the
// entries the LDS area holds, so a caller that understated its lists cannot make the block write or read past it.
__device__ __forceinline__ void stage_graph_lists(int64_t b, int N, const int32_t *ptr, const uint16_t *ids, const uint16_t *ord, int cap,
                                                  int *off_s, uint16_t *ord_s, uint16_t *ids_s) {
    const int32_t base = ptr[b * N];
    const int seg = min(ptr[(b + 1) * N] - base, cap & ~3);
    const uint2 *src = reinterpret_cast<const uint2 *>(ids + base);
    for (int e = threadIdx.x; e < seg / 4; e += blockDim.x) reinterpret_cast<uint2 *>(ids_s)[e] = src[e];
    for (int r = threadIdx.x; r <= N; r += blockDim.x) off_s[r] = min(ptr[b * N + r] - base, seg);
    for (int r = threadIdx.x; r < N; r += blockDim.x) ord_s[r] = ord[b * N + r];
}

// LISTS: the mask is not read at all; the lists come prebuilt (gatmi_dense_lists_fill) and are staged next to the
// graph's rows, a wave takes the rows eight at a time in the order of their list lengths (rows of one octet finish
// together), and what remains is the walk.  DROP: dropout of utils/layers.py:20,22 from the counter hashes, the
// feature mask applied once per row while the graph is staged.
template <int LPH8, bool SAVE, bool DROP>
__global__ __launch_bounds__(64 * kGraph8MaxWaves) void dense_graph_fwd8_kernel(MaskedFwdArgs a, int lc) {
    constexpr bool LISTS = false;          // (the walk-only form of this kernel is dense_graph_fwd8_lists_kernel)
    constexpr int G = 16;                  // compaction: lanes per row
    constexpr int H = 8 / LPH8;
    constexpr int HD = 64;
    extern __shared__ __attribute__((aligned(16))) unsigned char graph_smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int nwaves = blockDim.x >> 6;
    const int sub = lane % G, grp = lane / G;
    const int N = a.N;
    const int NP = N + kGraphPadRows;
    const int64_t b = blockIdx.x;
    float *whs = reinterpret_cast<float *>(graph_smem);
    float *f2s = whs + NP * HD;
    // compaction mode: this wave's eight lists and their entry counts; lists mode: offsets, row order, all lists of the graph
    uint16_t *lists = reinterpret_cast<uint16_t *>(f2s + NP * H) + (LISTS ? 0 : wave * kGraphWalkRows * lc);
    int *cnt = reinterpret_cast<int *>(reinterpret_cast<uint16_t *>(f2s + NP * H) + nwaves * kGraphWalkRows * lc) + wave * kGraphWalkRows;
    int *off_s = reinterpret_cast<int *>(f2s + NP * H);
    uint16_t *ord_s = reinterpret_cast<uint16_t *>(off_s + ((N + 2) & ~1));
    if constexpr (LISTS) lists = ord_s + ((N + 3) & ~3);
    const float *bmat = LISTS ? nullptr : a.bias_mat + b * N * (int64_t)N;
    const uint64_t seed = DROP ? seed_now(a) : 0;
    // quads of this wave in order: (oct0, first half), (oct0, second half), (oct0 + nwaves * 8, first half), ...
    MaskRow next;
    int oct0 = wave * kGraphWalkRows;
    if constexpr (!LISTS) load_mask_row(next, bmat + min(oct0 + grp, N - 1) * N, N, sub);
    {
        const float4 *src = reinterpret_cast<const float4 *>(a.Wh + b * N * (int64_t)HD);
        for (int e = threadIdx.x; e < NP * (HD / 4); e += blockDim.x) {
            float4 v = e < N * (HD / 4) ? src[e] : make_float4(0.f, 0.f, 0.f, 0.f);
            if (DROP) {
                if (a.feat_keep < 1.f) v = drop_feat(v, seed, b * N + (e >> 4), HD, (e & 15) * 4, a.feat_keep);
            }
            reinterpret_cast<float4 *>(whs)[e] = v;
        }
        const float *src2 = a.f2 + b * N * (int64_t)H;
        for (int e = threadIdx.x; e < NP * H; e += blockDim.x) f2s[e] = e < N * H ? src2[e] : -INFINITY;
        if constexpr (LISTS) stage_graph_lists(b, N, a.lptr, a.lcol, a.lord, a.lcap, off_s, ord_s, lists);
    }
    __syncthreads();
    const int W = (N + 31) / 32 + 1;
    uint32_t *present_b = (!LISTS && a.present) ? a.present + b * N * (int64_t)W : nullptr;
    // walk geometry: row r8 of the octet, lane s8 of the row
    const int r8 = lane >> 3, s8 = lane & 7;
    const int head8 = s8 / LPH8;
    int pa, pb;
    walk_pieces(lane, 3, pa, pb);
    const int oa = s8 * 8 + pa, ob = s8 * 8 + pb;        // columns of this lane's first and second piece
    const float *wha = whs + oa, *whb = whs + ob;
    const float *f2l = f2s + head8;
    for (; oct0 < N; oct0 += nwaves * kGraphWalkRows) {
        bool anysoft = false;
        if constexpr (!LISTS) {
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            const int row0 = oct0 + 4 * half;
            const MaskRow cur = mask_row_trim(next, N, sub);
            {                                                // the next quad's rows travel during this quad's work
                const int nrow0 = half == 0 ? row0 + 4 : oct0 + nwaves * kGraphWalkRows;
                load_mask_row(next, bmat + min(nrow0 + grp, N - 1) * N, N, sub);      // past the graph: its last row again, never used
            }
            uint16_t *lrow = lists + (4 * half + grp) * lc;
            if (row0 >= N) {                                 // wave-uniform: an octet's second quad beyond the graph
                if (sub == 0) cnt[4 * half + grp] = 0;
                continue;
            }
            const int i = min(row0 + grp, N - 1);
            const bool row_ok = row0 + grp < N;
            float mx = -INFINITY;
#pragma unroll
            for (int k = 0; k < kGraphK; ++k) mx = raw_max3(mx, raw_max(cur.v[k].x, cur.v[k].y), raw_max(cur.v[k].z, cur.v[k].w));
            mx = row16_max(mx);
            const float cut = mx - kMaskCutoffFwd;
            uint32_t bits = 0;
            bool nonzero = mx != 0.f;
#pragma unroll
            for (int k = 0; k < kGraphK; ++k) {
                const float comp[4] = {cur.v[k].x, cur.v[k].y, cur.v[k].z, cur.v[k].w};
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const bool present = comp[c] > cut;
                    bits |= present ? (1u << (4 * k + c)) : 0u;
                    nonzero |= present && comp[c] < 0.f;
                }
            }
            const int cntl = __popc(bits);
            const int total = row16_sum(cntl);
            int pos = row16_scan(cntl) - cntl;
            const uint64_t nzb = __ballot(nonzero && bits != 0);
            anysoft |= nzb != 0;
            if (present_b && row_ok) {
                uint32_t *prow = present_b + i * W;
#pragma unroll
                for (int k = 0; k < kGraphK; ++k) {
                    if (64 * k < N) {
                        const uint32_t w = or8_to_first(((bits >> (4 * k)) & 15u) << (4 * (sub & 7)));
                        const int wi = 2 * k + (sub >> 3);
                        if ((sub & 7) == 0 && wi < W - 1) prow[wi] = w;
                    }
                }
                if (sub == 0) prow[W - 1] = ((nzb >> (16 * grp)) & 0xffffull) != 0 ? 1u : 0u;
            }
            for (uint32_t rest = bits; __ballot(rest != 0) != 0;) {
                if (rest != 0) {
                    const int t = __ffs(rest) - 1;
                    lrow[pos++] = (uint16_t)(64 * (t >> 2) + 4 * sub + (t & 3));
                    rest &= rest - 1;
                }
            }
            if (sub < kGraphBatch) lrow[total + sub] = (uint16_t)N;          // padding: the dummy row
            if (sub == 0) cnt[4 * half + grp] = row_ok ? total : 0;
        }
        // the lists and counts are written and read by this wave only; its lanes run in lockstep
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        }
        const bool row_ok = oct0 + r8 < N;
        int i, total;
        const uint16_t *lrow;
        if constexpr (LISTS) {
            i = ord_s[min(oct0 + r8, N - 1)];
            const int o0 = off_s[i];
            total = row_ok ? off_s[i + 1] - o0 : 0;
            lrow = lists + o0;
        } else {
            i = min(oct0 + r8, N - 1);
            total = cnt[r8];
            lrow = lists + r8 * lc;
        }
        const float f1v = a.f1[(b * N + i) * H + head8];
        const uint64_t cbase = DROP ? ((uint64_t)(b * N + i) * (uint64_t)N) * H + head8 : 0;
        MRowState8 st;
        row_state_reset(st, -INFINITY);
        {
            const float *bm_own = LISTS ? nullptr : bmat + i * N;
            if (!LISTS && anysoft) {                      // wave-uniform: biases are fetched for all of the wave's rows
                for (int k = 0; k < total; k += kGraphBatch)
                    graph_fwd_batch8<LPH8, SAVE, !LISTS, DROP>(st, a.slope, lrow + k, bm_own, N, wha, whb, f2l, f1v, seed, cbase, a.coef_keep);
            } else {
                for (int k = 0; k < total; k += kGraphBatch)
                    graph_fwd_batch8<LPH8, SAVE, false, DROP>(st, a.slope, lrow + k, bm_own, N, wha, whb, f2l, f1v, seed, cbase, a.coef_keep);
            }
        }
        if constexpr (!LISTS) __builtin_amdgcn_wave_barrier();   // every lane is done with the lists before the next rows overwrite them
        if (row_ok) {
            const int64_t row = b * N + i;
            const bool empty = !(st.s > 0.f);
            const float inv = empty ? 0.f : 1.0f / st.s;
            const float4 ag0 = f4_scale(inv, st.acc0), ag1 = f4_scale(inv, st.acc1);
            float *o = nullptr;
            if (SAVE) {
                o = a.agg + row * HD;
                store_stream4(o + oa, ag0);
                store_stream4(o + ob, ag1);
                o = a.pos + row * HD;
                store_stream4(o + oa, f4_scale(inv, st.pos0));
                store_stream4(o + ob, f4_scale(inv, st.pos1));
                if (s8 % LPH8 == 0) {
                    a.lse[row * H + head8] = empty ? 0.f : st.m + logf(st.s);
                    a.q[row * H + head8] = st.q * inv;
                }
            }
            float4 r0 = ag0, r1 = ag1;
            if (a.bias) {
                r0 = f4_add(r0, *reinterpret_cast<const float4 *>(a.bias + oa));
                r1 = f4_add(r1, *reinterpret_cast<const float4 *>(a.bias + ob));
            }
            if (a.resid) {
                r0 = f4_add(r0, *reinterpret_cast<const float4 *>(a.resid + row * HD + oa));
                r1 = f4_add(r1, *reinterpret_cast<const float4 *>(a.resid + row * HD + ob));
            }
            if (a.activation != GATMI_ACT_IDENTITY) {
                r0 = make_float4(act_f(r0.x, a.activation), act_f(r0.y, a.activation), act_f(r0.z, a.activation), act_f(r0.w, a.activation));
                r1 = make_float4(act_f(r1.x, a.activation), act_f(r1.y, a.activation), act_f(r1.z, a.activation), act_f(r1.w, a.activation));
            }
            o = a.out + row * HD;
            store_stream4(o + oa, r0);
            store_stream4(o + ob, r1);
        }
    }
}

// ------------------------------------------------------------------------------------------
// The walk alone, on prebuilt lists (gatmi_dense_lists_fill): no mask read, no compaction.  The lists are not
// staged either: a row's eight lanes fetch 32 of its entries with one 8-byte load each, straight from memory,
// and hand them round with ds_bpermute; the next octet's entries are requested before the current octet is
// walked.  LDS then holds only the graph's rows (Wh, f2), the list offsets and the row order — under 80 KB
// at 264 nodes, so that TWO blocks share a CU and one block's staging and stores overlap the other's walk
// (a block alone alternates between a memory phase and an arithmetic phase).  Rows are taken in the order of
// their list lengths: the eight rows of an octet finish together.
// ------------------------------------------------------------------------------------------
constexpr int kGraph8ListsWaves = 8;       // two blocks per CU: four waves per SIMD, 128 registers per lane
static inline int graph8_lists_waves(int N) {
    const int octs = (N + kGraphWalkRows - 1) / kGraphWalkRows;
    const int rounds = (octs + kGraph8ListsWaves - 1) / kGraph8ListsWaves;
    return (octs + rounds - 1) / rounds;
}
// LDS: Wh [N + pad][64] | f2 [N + pad][H] | list offsets [N + 1 -> even] (int) | row order [N -> multiple of 4] (uint16)
static inline size_t graph8_fwd_walk_lds(int N, int H) {
    return (size_t)(N + kGraphPadRows) * (64 + H) * sizeof(float) + graph8_lists_head_lds(N);
}

template <int LPH8, bool SAVE, bool DROP>
__global__ __launch_bounds__(64 * kGraph8ListsWaves, 2) void dense_graph_fwd8_lists_kernel(MaskedFwdArgs a) {
    constexpr int H = 8 / LPH8;
    constexpr int HD = 64;
    extern __shared__ __attribute__((aligned(16))) unsigned char graph_smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int nwaves = blockDim.x >> 6;
    const int N = a.N;
    const int NP = N + kGraphPadRows;
    const int64_t b = blockIdx.x;
    float *whs = reinterpret_cast<float *>(graph_smem);
    float *f2s = whs + NP * HD;
    int *off_s = reinterpret_cast<int *>(f2s + NP * H);
    uint16_t *ord_s = reinterpret_cast<uint16_t *>(off_s + ((N + 2) & ~1));
    const uint64_t seed = DROP ? seed_now(a) : 0;
    const int32_t base = a.lptr[b * N];
    {
        const float4 *src = reinterpret_cast<const float4 *>(a.Wh + b * N * (int64_t)HD);
        for (int e = threadIdx.x; e < NP * (HD / 4); e += blockDim.x) {
            float4 v = e < N * (HD / 4) ? src[e] : make_float4(0.f, 0.f, 0.f, 0.f);
            if (DROP) {
                if (a.feat_keep < 1.f) v = drop_feat(v, seed, b * N + (e >> 4), HD, (e & 15) * 4, a.feat_keep);
            }
            reinterpret_cast<float4 *>(whs)[e] = v;
        }
        const float *src2 = a.f2 + b * N * (int64_t)H;
        for (int e = threadIdx.x; e < NP * H; e += blockDim.x) f2s[e] = e < N * H ? src2[e] : -INFINITY;
        for (int r = threadIdx.x; r <= N; r += blockDim.x) off_s[r] = a.lptr[b * N + r] - base;
        for (int r = threadIdx.x; r < N; r += blockDim.x) ord_s[r] = a.lord[b * N + r];
    }
    __syncthreads();
    const int r8 = lane >> 3, s8 = lane & 7;
    const int head8 = s8 / LPH8;
    int pa, pb;
    walk_pieces(lane, 3, pa, pb);
    const int oa = s8 * 8 + pa, ob = s8 * 8 + pb;        // columns of this lane's first and second piece
    const float *wha = whs + oa, *whb = whs + ob;
    const float *f2l = f2s + head8;
    const uint16_t *gcol = a.lcol + base;
    const uint32_t dummy = (uint32_t)N | ((uint32_t)N << 16);      // the zero row (Wh = 0, f2 = -inf: coefficient exactly 0)
    const int lane0 = lane & 56;                                    // first lane of this row's group
    // entries 4 s8 .. 4 s8 + 3 of chunk c0 of a row's list (a whole batch each); past the list: the dummy id.  The
    // load is unconditional (clamped inside the list) so that the request for the NEXT octet can stay in flight
    // across the walk of the current one.
    auto fetch = [&](int o0, int total, int c0) -> uint2 {
        const int at = c0 + 4 * s8;
        const uint2 v = *reinterpret_cast<const uint2 *>(gcol + o0 + min(at, max(total - 4, 0)));
        return at < total ? v : make_uint2(dummy, dummy);
    };
    auto octet = [&](int oct, int &i, int &o0, int &total) {
        i = ord_s[min(oct + r8, N - 1)];
        o0 = off_s[i];
        total = oct + r8 < N ? off_s[i + 1] - o0 : 0;
    };
    int oct0 = wave * kGraphWalkRows;
    int i = 0, o0 = 0, total = 0;
    uint2 ch = make_uint2(dummy, dummy);
    if (oct0 < N) {
        octet(oct0, i, o0, total);
        ch = fetch(o0, total, 0);
    }
    for (; oct0 < N; oct0 += nwaves * kGraphWalkRows) {
        const int octn = oct0 + nwaves * kGraphWalkRows;
        int in, on, tn;
        octet(min(octn, N - 1), in, on, tn);                        // past the graph: the clamped row, never walked
        if (octn >= N) tn = 0;
        const uint2 chn = fetch(on, tn, 0);
        __builtin_amdgcn_sched_barrier(0);
        const bool row_ok = oct0 + r8 < N;
        const float f1v = a.f1[(b * N + i) * H + head8];
        const uint64_t cbase = DROP ? ((uint64_t)(b * N + i) * (uint64_t)N) * H + head8 : 0;
        MRowState8 st;
        row_state_reset(st, -INFINITY);
        for (int c0 = 0; __any(c0 < total); c0 += 32) {
            if (c0 > 0) ch = fetch(o0, total, c0);                  // lists of more than 32 entries: further chunks as they come
#pragma unroll
            for (int t = 0; t < 8; ++t) {
                if (!__any(c0 + 4 * t < total)) break;              // wave-uniform
                const uint32_t w0 = (uint32_t)__shfl((int)ch.x, lane0 + t, 64), w1 = (uint32_t)__shfl((int)ch.y, lane0 + t, 64);
                const int j[kGraphBatch] = {(int)(w0 & 0xffff), (int)(w0 >> 16), (int)(w1 & 0xffff), (int)(w1 >> 16)};
                graph_fwd_batch8_ids<LPH8, SAVE, false, DROP>(st, a.slope, j, nullptr, N, wha, whb, f2l, f1v, seed, cbase, a.coef_keep);
            }
        }
        if (row_ok) {
            const int64_t row = b * N + i;
            const bool empty = !(st.s > 0.f);
            const float inv = empty ? 0.f : 1.0f / st.s;
            const float4 ag0 = f4_scale(inv, st.acc0), ag1 = f4_scale(inv, st.acc1);
            float *o = nullptr;
            if (SAVE) {
                o = a.agg + row * HD;
                store_stream4(o + oa, ag0);
                store_stream4(o + ob, ag1);
                o = a.pos + row * HD;
                store_stream4(o + oa, f4_scale(inv, st.pos0));
                store_stream4(o + ob, f4_scale(inv, st.pos1));
                if (s8 % LPH8 == 0) {
                    a.lse[row * H + head8] = empty ? 0.f : st.m + logf(st.s);
                    a.q[row * H + head8] = st.q * inv;
                }
            }
            float4 r0 = ag0, r1 = ag1;
            if (a.bias) {
                r0 = f4_add(r0, *reinterpret_cast<const float4 *>(a.bias + oa));
                r1 = f4_add(r1, *reinterpret_cast<const float4 *>(a.bias + ob));
            }
            if (a.resid) {
                r0 = f4_add(r0, *reinterpret_cast<const float4 *>(a.resid + row * HD + oa));
                r1 = f4_add(r1, *reinterpret_cast<const float4 *>(a.resid + row * HD + ob));
            }
            if (a.activation != GATMI_ACT_IDENTITY) {
                r0 = make_float4(act_f(r0.x, a.activation), act_f(r0.y, a.activation), act_f(r0.z, a.activation), act_f(r0.w, a.activation));
                r1 = make_float4(act_f(r1.x, a.activation), act_f(r1.y, a.activation), act_f(r1.z, a.activation), act_f(r1.w, a.activation));
            }
            o = a.out + row * HD;
            store_stream4(o + oa, r0);
            store_stream4(o + ob, r1);
        }
        i = in; o0 = on; total = tn; ch = chn;
    }
}

// head layouts of the graph-resident kernels (lanes per head; rows are always 64 floats: 16 lanes)
#define GATMI_GRAPH_SHAPES(X) X(1) X(2) X(4) X(8) X(16)
constexpr int kGraphMaxN = 64 * kGraphK;    // mask columns a lane group holds in registers
constexpr int64_t kGraphMinBatch = 256;     // at least one block per CU, else the row-block kernels fill the chip better

// ------------------------------------------------------------------------------------------
// backward
// ------------------------------------------------------------------------------------------
struct MaskedBwdArgs {
    int64_t B;
    int N;
    const float *g, *stats, *Wh, *f2, *bias_mat, *df1, *a1, *a2;
    const uint32_t *present;   // [B*N, W] bits written by the forward kernel
    float *dWh, *df2;
    float slope, coef_keep, feat_keep;
    uint64_t seed;
    const uint64_t *seed_step;   // device word added to the seed at run time (seed_now), or nullptr
    // fused per-destination step (dense_graph_bwd8_kernel<.., true>): g, stats and df1 are formed from these while
    // the graph is staged, instead of being read from arrays an earlier launch wrote
    const float *dout, *out, *agg, *pos, *q, *f1, *lse, *bias;
    float *g_out, *df1_out, *dbias_ws;
    int activation;
    // dbias_ws holds one partial row of ws_stride floats per graph: [0, 64) the bias gradient and, from
    // dense_graph_bwd8_half_kernel when param_grads is set, [64, 128) da1, [128, 192) da2, then db1 [H], db2 [H]
    int ws_stride, param_grads;
    // prebuilt lists (dense_graph_bwd8_kernel<.., LISTS>): column c of graph b is visited by rows
    // lcol[lptr[b*N + c] .. lptr[b*N + c + 1]) (padded to whole batches with the dummy id N); lord = columns by list length
    const int32_t *lptr;
    const uint16_t *lcol, *lord;
    int lcap;
};

struct MSrcState {
    float4 acc;
    float df2;
};

// gb / statb / bmcol: this lane's float4 column of the graph's g rows, its head's first statistic and
// column jcol of the graph's mask, addressed by 32-bit row offsets.
template <int G, int LPH, bool DROP>
__device__ __forceinline__ void masked_bwd_batch(MSrcState &st, const MaskedBwdArgs &a, const uint16_t *lp, int nvalid,
                                                 int64_t b, int jcol, const float *gb, const float *statb, const float *bmcol,
                                                 float f2v, float4 whj, int sub, int head) {
    constexpr int H = G / LPH;
    constexpr int HD = 4 * G;
    constexpr int U = kBatch;
    const int N = a.N;
    int i[U];
    float4 gv[U];
    float f1v[U], lsev[U], cv[U], bv[U];
    load_ids<U>(lp, i);                                  // entries: row id | (needs its bias value) << 15
#pragma unroll
    for (int t = 0; t < U; ++t) {
        if (t >= nvalid) i[t] = i[0];
        const bool soft = i[t] & 0x8000;
        i[t] &= 0x7fff;
        gv[t] = *reinterpret_cast<const float4 *>(gb + (uint32_t)i[t] * HD);
        const float *s = statb + (uint32_t)i[t] * (3 * H);
        f1v[t] = s[0];
        lsev[t] = s[H];
        cv[t] = s[2 * H];
        bv[t] = soft ? bmcol[(uint32_t)i[t] * (uint32_t)N] : 0.f;
    }
    float bdf2 = 0.f;
    float4 bacc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int t = 0; t < U; ++t) {
        const float z = f1v[t] + f2v;
        float alpha = exp_neg(leaky(z, a.slope) + bv[t] - lsev[t]);
        if (t >= nvalid) alpha = 0.f;
        float dalpha = group_sum<LPH>(dot4(gv[t], whj));
        float aa = alpha;
        if (DROP) {
            if (a.coef_keep < 1.f) {
                const float mk = coef_drop_scale(seed_now(a), b, N, H, i[t], jcol, head, a.coef_keep);
                aa = alpha * mk;
                dalpha *= mk;
            }
        }
        bdf2 += alpha * (dalpha - cv[t]) * (z > 0.f ? 1.f : a.slope);
        bacc = f4_fma(aa, gv[t], bacc);
    }
    st.acc = f4_add(st.acc, bacc);
    st.df2 += bdf2;
}

template <int G, int LPH, bool DROP>
__global__ __launch_bounds__(256, GATMI_MASK_MINBLOCKS) void dense_masked_bwd_kernel(MaskedBwdArgs a, int n_chunks) {
    constexpr int RPW = 64 / G;
    constexpr int H = G / LPH;
    constexpr int HD = 4 * G;
    constexpr int GB = 4 * RPW;                       // column groups per block
    constexpr int PASSES = kBwdCols / GB;             // columns per group
    static_assert(kBwdCols % GB == 0 && PASSES >= 1 && kBwdCols == 32, "32 columns per block: the bits of one word");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    uint16_t *lists = reinterpret_cast<uint16_t *>(smem_raw);                                  // [col][n_chunks * 64]
    int *counts = reinterpret_cast<int *>(smem_raw + (size_t)kBwdCols * n_chunks * 64 * 2);     // [col][kBwdChunks + 1]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int sub = lane % G, grp = lane / G;
    const int head = sub / LPH;
    const int N = a.N;
    const int W = (N + 31) / 32 + 1;
    const int64_t b = blockIdx.x;
    const int c0 = blockIdx.y * kBwdCols;
    const int gidx = wave * RPW + grp;                // this lane group's index in the block
    const uint64_t lt_mask = lane == 0 ? 0ull : (~0ull >> (64 - lane));
    const float *gb = a.g + b * N * (int64_t)HD + sub * 4;
    const float *statb = a.stats + b * N * (int64_t)(3 * H) + head;

    MSrcState st[PASSES];
    float f2v[PASSES];
    float4 whj[PASSES];
#pragma unroll
    for (int p = 0; p < PASSES; ++p) {
        const int jcol = min(c0 + p * GB + gidx, N - 1);
        st[p].acc = make_float4(0.f, 0.f, 0.f, 0.f);
        st[p].df2 = 0.f;
        f2v[p] = a.f2[(b * N + jcol) * H + head];
        whj[p] = *reinterpret_cast<const float4 *>(a.Wh + (b * N + jcol) * (int64_t)HD + sub * 4);
        if (DROP) {
            if (a.feat_keep < 1.f) whj[p] = drop_feat(whj[p], seed_now(a), b * N + jcol, HD, sub * 4, a.feat_keep);
        }
    }

    const int seg_rows = n_chunks * 64;               // capacity of one column's list
    for (int seg0 = 0; seg0 < N; seg0 += seg_rows) {
        __syncthreads();                              // the previous pass has finished reading the lists
        // This wave's chunks: c = wave and c = wave + 4 (n_chunks <= 8); lane = destination row.
        uint32_t word[2] = {0u, 0u};
        uint16_t entry[2] = {0, 0};
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int c = wave + 4 * q;
            const int i = seg0 + 64 * c + lane;
            if (c < n_chunks && i < N) {
                const uint32_t *prow = a.present + (b * N + i) * (int64_t)W;
                word[q] = prow[blockIdx.y];
                entry[q] = (uint16_t)(i | ((prow[W - 1] & 1u) << 15));
            }
        }
#pragma unroll
        for (int q = 0; q < 2; ++q) {                 // entries per (column, chunk)
            const int c = wave + 4 * q;
            if (c >= n_chunks) continue;
            int mycount = 0;
#pragma unroll
            for (int col = 0; col < kBwdCols; ++col) {
                const int n = __popcll(__ballot((word[q] >> col) & 1u));
                if (lane == col) mycount = n;
            }
            if (lane < kBwdCols) counts[lane * (kBwdChunks + 1) + c] = mycount;
        }
        __syncthreads();
        if (threadIdx.x < kBwdCols) {                 // exclusive prefix over the chunks of a column; slot kBwdChunks = total
            int run = 0;
            for (int c = 0; c < n_chunks; ++c) {
                const int n = counts[threadIdx.x * (kBwdChunks + 1) + c];
                counts[threadIdx.x * (kBwdChunks + 1) + c] = run;
                run += n;
            }
            counts[threadIdx.x * (kBwdChunks + 1) + kBwdChunks] = run;
        }
        __syncthreads();
#pragma unroll
        for (int q = 0; q < 2; ++q) {                 // fill: a column's list is contiguous and sorted by row
            const int c = wave + 4 * q;
            if (c >= n_chunks) continue;
#pragma unroll
            for (int col = 0; col < kBwdCols; ++col) {
                const bool bit = (word[q] >> col) & 1u;
                const uint64_t bal = __ballot(bit);
                if (bit) lists[(size_t)col * seg_rows + counts[col * (kBwdChunks + 1) + c] + __popcll(bal & lt_mask)] = entry[q];
            }
        }
        __syncthreads();
#pragma unroll
        for (int p = 0; p < PASSES; ++p) {
            const int cl = p * GB + gidx;             // column within the block
            const int jcol = c0 + cl;
            if (jcol < N) {
                const int cnt = counts[cl * (kBwdChunks + 1) + kBwdChunks];
                const uint16_t *lp = lists + (size_t)cl * seg_rows;
                const float *bmcol = a.bias_mat + b * N * (int64_t)N + jcol;
                for (int k = 0; k < cnt; k += kBatch)
                    masked_bwd_batch<G, LPH, DROP>(st[p], a, lp + k, min(kBatch, cnt - k), b, jcol, gb, statb, bmcol, f2v[p], whj[p], sub, head);
            }
        }
    }
#pragma unroll
    for (int p = 0; p < PASSES; ++p) {
        const int jcol = c0 + p * GB + gidx;
        if (jcol >= N) continue;
        const int64_t j = b * N + jcol;
        float4 r = st[p].acc;
        if (DROP) {
            if (a.feat_keep < 1.f) r = drop_feat(r, seed_now(a), j, HD, sub * 4, a.feat_keep);
        }
        if (a.a1) {
            const float d1 = a.df1[j * H + head];
            r = f4_fma(d1, *reinterpret_cast<const float4 *>(a.a1 + sub * 4), r);
            r = f4_fma(st[p].df2, *reinterpret_cast<const float4 *>(a.a2 + sub * 4), r);
        }
        *reinterpret_cast<float4 *>(a.dWh + j * HD + sub * 4) = r;
        if (sub % LPH == 0) a.df2[j * H + head] = st[p].df2;
    }
}

// ------------------------------------------------------------------------------------------
// backward, graph-resident form: one block per graph; the graph's g rows, per-node statistics and
// visited-entry bits are staged in LDS once, and every wave takes source columns four at a time
// (one per lane group): the 16 lanes of a group collect the destination rows whose bit for the
// column is set (lane sub looks at rows sub, sub + 16, ...), a DPP prefix sum orders them into the
// group's list, and the walk gathers from LDS.  Lists are padded with a dummy row (g = 0,
// lse = +inf: coefficient exactly 0).  Same `present` format as dense_masked_bwd_kernel.
// ------------------------------------------------------------------------------------------
template <int LPH, bool SOFT>
__device__ __forceinline__ void graph_bwd_batch(MSrcState &st, float slope, const uint16_t *lp, const float *bmcol, int N,
                                                const float *gs, const float *ss, float f2v, float4 whj) {
    constexpr int G = 16;
    constexpr int H = G / LPH;
    constexpr int HD = 4 * G;
    constexpr int U = kGraphBatch;
    int i[U];
    float4 gv[U];
    float f1v[U], lsev[U], cv[U], bv[U];
    load_ids<U>(lp, i);                                  // entries: row id | (needs its bias value) << 15
#pragma unroll
    for (int t = 0; t < U; ++t) {
        const bool soft = i[t] & 0x8000;
        i[t] &= 0x7fff;
        gv[t] = *reinterpret_cast<const float4 *>(gs + i[t] * HD);
        const float *sp = ss + i[t] * (3 * H);
        f1v[t] = sp[0];
        lsev[t] = sp[H];
        cv[t] = sp[2 * H];
        bv[t] = 0.f;
        if (SOFT) {
            if (soft) bv[t] = bmcol[(uint32_t)min(i[t], N - 1) * (uint32_t)N];
        }
    }
    float bdf2 = 0.f;
    float4 bacc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int t = 0; t < U; ++t) {
        const float z = f1v[t] + f2v;
        const float alpha = exp_neg((SOFT ? leaky(z, slope) + bv[t] : leaky(z, slope)) - lsev[t]);
        const float dalpha = group_sum<LPH>(dot4(gv[t], whj));
        bdf2 += alpha * (dalpha - cv[t]) * (z > 0.f ? 1.f : slope);
        bacc = f4_fma(alpha, gv[t], bacc);
    }
    st.acc = f4_add(st.acc, bacc);
    st.df2 += bdf2;
}

constexpr int kGraphBwdRows = (kGraphMaxN + 15) / 16;    // destination rows a lane looks at per column

// LDS of the graph-resident backward kernel: g [N + pad][64] | stats [N + pad][3H] | bits [N][W] | lists
static inline size_t graph_bwd_lds(int N, int H, int lc) {
    const int W = (N + 31) / 32 + 1;
    return (size_t)(N + kGraphPadRows) * (64 + 3 * H) * sizeof(float) + (size_t)N * W * sizeof(uint32_t) +
           (size_t)(kGraphFwdThreads / 64) * 4 * lc * sizeof(uint16_t);
}

template <int LPH>
__global__ __launch_bounds__(kGraphFwdThreads) void dense_graph_bwd_kernel(MaskedBwdArgs a, int lc) {
    constexpr int G = 16;
    constexpr int RPW = 4;
    constexpr int H = G / LPH;
    constexpr int HD = 4 * G;
    extern __shared__ __attribute__((aligned(16))) unsigned char graph_smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int nwaves = blockDim.x >> 6;
    const int sub = lane % G, grp = lane / G;
    const int head = sub / LPH;
    const int N = a.N;
    const int NP = N + kGraphPadRows;
    const int W = (N + 31) / 32 + 1;
    const int64_t b = blockIdx.x;
    float *gs = reinterpret_cast<float *>(graph_smem);                        // [NP][HD]
    float *ss = gs + NP * HD;                                                 // [NP][3H]
    uint32_t *ps = reinterpret_cast<uint32_t *>(ss + NP * 3 * H);             // [N][W]
    uint16_t *lrow = reinterpret_cast<uint16_t *>(ps + N * W) + (wave * RPW + grp) * lc;      // N * W even (N % 4 == 0): 8-byte aligned
    {
        const float4 *src = reinterpret_cast<const float4 *>(a.g + b * N * (int64_t)HD);
        for (int e = threadIdx.x; e < NP * (HD / 4); e += blockDim.x)
            reinterpret_cast<float4 *>(gs)[e] = e < N * (HD / 4) ? src[e] : make_float4(0.f, 0.f, 0.f, 0.f);
        const float *src2 = a.stats + b * N * (int64_t)(3 * H);
        for (int e = threadIdx.x; e < NP * 3 * H; e += blockDim.x) {
            // dummy rows: f1 = 0, lse = +inf (coefficient 0), c = 0
            const int k = (e % (3 * H)) / H;
            ss[e] = e < N * 3 * H ? src2[e] : (k == 1 ? INFINITY : 0.f);
        }
        const uint32_t *src3 = a.present + b * N * (int64_t)W;
        for (int e = threadIdx.x; e < N * W; e += blockDim.x) ps[e] = src3[e];
    }
    __syncthreads();
    const float *gl = gs + sub * 4;
    const float *sl = ss + head;
    for (int j0 = wave * RPW; j0 < N; j0 += nwaves * RPW) {
        const int jcol = min(j0 + grp, N - 1);
        const bool col_ok = j0 + grp < N;
        const int64_t j = b * N + jcol;
        const float f2v = a.f2[j * H + head];
        const float4 whj = *reinterpret_cast<const float4 *>(a.Wh + j * HD + sub * 4);
        // bit r of `bits`: destination row 16 r + sub visited column jcol; `softs`: and needs its bias value
        uint32_t bits = 0, softs = 0;
        const int wj = jcol >> 5, bj = jcol & 31;
#pragma unroll
        for (int r = 0; r < kGraphBwdRows; ++r) {
            const int i = 16 * r + sub;
            if (16 * r < N) {                             // uniform
                const int ic = min(i, N - 1);
                const uint32_t word = ps[ic * W + wj], flag = ps[ic * W + W - 1];
                const uint32_t bit = i < N ? (word >> bj) & 1u : 0u;
                bits |= bit << r;
                softs |= (bit & flag) << r;
            }
        }
        const int cnt = __popc(bits);
        const int total = row16_sum(cnt);
        int pos = row16_scan(cnt) - cnt;
        const bool anysoft = __ballot(softs != 0) != 0;
        for (uint32_t rest = bits; __ballot(rest != 0) != 0;) {          // each lane appends its rows at its offset
            if (rest != 0) {
                const int t = __ffs(rest) - 1;
                lrow[pos++] = (uint16_t)((16 * t + sub) | (((softs >> t) & 1u) << 15));
                rest &= rest - 1;
            }
        }
        if (sub < kGraphBatch) lrow[total + sub] = (uint16_t)N;          // padding: the dummy row
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        MSrcState st;
        st.acc = make_float4(0.f, 0.f, 0.f, 0.f);
        st.df2 = 0.f;
        if (col_ok) {
            const float *bmcol = a.bias_mat + b * N * (int64_t)N + jcol;
            if (anysoft) {
                for (int k = 0; k < total; k += kGraphBatch) graph_bwd_batch<LPH, true>(st, a.slope, lrow + k, bmcol, N, gl, sl, f2v, whj);
            } else {
                for (int k = 0; k < total; k += kGraphBatch) graph_bwd_batch<LPH, false>(st, a.slope, lrow + k, bmcol, N, gl, sl, f2v, whj);
            }
        }
        __builtin_amdgcn_wave_barrier();                 // every lane is done with the lists before the next columns overwrite them
        if (col_ok) {
            float4 r = st.acc;
            if (a.a1) {
                const float d1 = a.df1[j * H + head];
                r = f4_fma(d1, *reinterpret_cast<const float4 *>(a.a1 + sub * 4), r);
                r = f4_fma(st.df2, *reinterpret_cast<const float4 *>(a.a2 + sub * 4), r);
            }
            store_stream4(a.dWh + j * HD + sub * 4, r);
            if (sub % LPH == 0) a.df2[j * H + head] = st.df2;
        }
    }
}

// ------------------------------------------------------------------------------------------
// The backward kernel with the walk on eight lanes per column (see dense_graph_fwd8_kernel): a wave
// builds the lists of two column quads with 16 lanes per column, then walks eight columns at once,
// a lane holding eight floats of its column's Wh row and of every gathered g row.  With heads of
// eight floats the dot product <g_i, Wh_j> of a head stays inside one lane.
// ------------------------------------------------------------------------------------------
struct MSrcState8 {
    float4 acc0, acc1;
    float df2;
};

// <a, b> over eight floats as four packed two-float multiply-adds and one add.  Written on two-float vectors so
// that the products of ONE entry share the packed instructions: left to itself the compiler pairs the products of
// two entries instead and spends two register moves per packed instruction on bringing their operands together.
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ float dot8_pairs(float4 a0, float4 a1, float4 b0, float4 b1) {
    f32x2 s = (f32x2){a0.x, a0.y} * (f32x2){b0.x, b0.y};
    s = __builtin_elementwise_fma((f32x2){a0.z, a0.w}, (f32x2){b0.z, b0.w}, s);
    s = __builtin_elementwise_fma((f32x2){a1.x, a1.y}, (f32x2){b1.x, b1.y}, s);
    s = __builtin_elementwise_fma((f32x2){a1.z, a1.w}, (f32x2){b1.z, b1.w}, s);
    return s.x + s.y;
}

// DROP: coefficient dropout per (entry, head); `cbase` is the hash index of destination row 0 for this lane's
// (column, head), an entry's index is cbase + i * (N * H).  wh0 / wh1 already carry the feature mask.
// ga / gb: this lane's two float4 pieces of g row 0, in the order walk_pieces gives them (wh0 / acc0 belong to the
// first, wh1 / acc1 to the second).  GS / HS: floats of a g row and heads of a statistics row as the block keeps them in LDS (the whole row, or the
// half of it that dense_graph_bwd8_half_kernel holds); the dropout hash always counts the layer's H heads.
template <int LPH8, bool SOFT, bool DROP, int GS = 64, int HS = 8 / LPH8>
__device__ __forceinline__ void graph_bwd_batch8_ids(MSrcState8 &st, float slope, int (&i)[kGraphBatch], const float *bmcol, int N,
                                                     const float *ga, const float *gb, const float *ss, float f2v, float4 wh0, float4 wh1,
                                                     uint64_t seed, uint64_t cbase, float coef_keep) {
    constexpr int H = 8 / LPH8;
    constexpr int U = kGraphBatch;
    float4 g0[U], g1[U];
    float f1v[U], lsev[U], cv[U], bv[U];
#pragma unroll
    for (int t = 0; t < U; ++t) {                        // entries: row id | (needs its bias value) << 15
        const bool soft = i[t] & 0x8000;
        i[t] &= 0x7fff;
        g0[t] = *reinterpret_cast<const float4 *>(ga + i[t] * GS);
        g1[t] = *reinterpret_cast<const float4 *>(gb + i[t] * GS);
        const float *sp = ss + i[t] * (3 * HS);
        f1v[t] = sp[0];
        lsev[t] = sp[HS];
        cv[t] = sp[2 * HS];
        bv[t] = 0.f;
        if (SOFT) {
            if (soft) bv[t] = bmcol[(uint32_t)min(i[t], N - 1) * (uint32_t)N];
        }
    }
    float bdf2 = 0.f;
    float4 bacc0 = make_float4(0.f, 0.f, 0.f, 0.f), bacc1 = bacc0;
    const float inv_keep = DROP ? 1.0f / coef_keep : 1.0f;
#pragma unroll
    for (int t = 0; t < U; ++t) {
        const float z = f1v[t] + f2v;
        const float alpha = exp_neg((SOFT ? leaky(z, slope) + bv[t] : leaky(z, slope)) - lsev[t]);
        float dalpha = group_sum<LPH8>(dot8_pairs(g0[t], g1[t], wh0, wh1));
        float aa = alpha;
        if (DROP) {
            if (coef_keep < 1.f) {
                const float mk = keep_mask(seed, cbase + (uint64_t)(uint32_t)i[t] * (uint64_t)(N * H), coef_keep) ? inv_keep : 0.f;
                aa = alpha * mk;
                dalpha *= mk;
            }
        }
        bdf2 += alpha * (dalpha - cv[t]) * (z > 0.f ? 1.f : slope);
        bacc0 = f4_fma(aa, g0[t], bacc0);
        bacc1 = f4_fma(aa, g1[t], bacc1);
    }
    st.acc0 = f4_add(st.acc0, bacc0);
    st.acc1 = f4_add(st.acc1, bacc1);
    st.df2 += bdf2;
}

template <int LPH8, bool SOFT, bool DROP>
__device__ __forceinline__ void graph_bwd_batch8(MSrcState8 &st, float slope, const uint16_t *lp, const float *bmcol, int N,
                                                 const float *ga, const float *gb, const float *ss, float f2v, float4 wh0, float4 wh1,
                                                 uint64_t seed, uint64_t cbase, float coef_keep) {
    int i[kGraphBatch];
    load_ids<kGraphBatch>(lp, i);
    graph_bwd_batch8_ids<LPH8, SOFT, DROP>(st, slope, i, bmcol, N, ga, gb, ss, f2v, wh0, wh1, seed, cbase, coef_keep);
}

// LDS: g [N + pad][64] | stats [N + pad][3H] | bits [N][W] | lists [waves][8][lc] (uint16) | entry counts [waves][8] (int)
// The fused kernel first sums the bias gradient over the block through the list area (one float4 per thread): small
// graphs have lists shorter than that, so the area is at least that large.
static inline size_t graph8_bwd_lds(int N, int H, int lc, int waves) {
    const int W = (N + 31) / 32 + 1;
    const size_t lists = (size_t)waves * kGraphWalkRows * lc * sizeof(uint16_t) + (size_t)waves * kGraphWalkRows * sizeof(int);
    const size_t block_sum = (size_t)waves * 64 * sizeof(float4);
    return (size_t)(N + kGraphPadRows) * (64 + 3 * H) * sizeof(float) + (size_t)N * W * sizeof(uint32_t) + std::max(lists, block_sum);
}
// with prebuilt lists: g | stats | max(block sum area, list offsets + column order + `cap` list entries)
static inline size_t graph8_bwd_lists_lds(int N, int H, int cap, int waves) {
    const size_t lists = graph8_lists_head_lds(N) + (size_t)cap * sizeof(uint16_t);
    const size_t block_sum = (size_t)waves * 64 * sizeof(float4);
    return (size_t)(N + kGraphPadRows) * (64 + 3 * H) * sizeof(float) + std::max(lists, block_sum);
}

template <int LPH8, bool FUSED, bool LISTS, bool DROP>
__global__ __launch_bounds__(64 * kGraph8MaxWaves) void dense_graph_bwd8_kernel(MaskedBwdArgs a, int lc) {
    constexpr int G = 16;
    constexpr int H = 8 / LPH8;
    constexpr int HD = 64;
    extern __shared__ __attribute__((aligned(16))) unsigned char graph_smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int nwaves = blockDim.x >> 6;
    const int sub = lane % G, grp = lane / G;
    const int N = a.N;
    const int NP = N + kGraphPadRows;
    const int W = (N + 31) / 32 + 1;
    const int64_t b = blockIdx.x;
    float *gs = reinterpret_cast<float *>(graph_smem);                        // [NP][HD]
    float *ss = gs + NP * HD;                                                 // [NP][3H]
    uint32_t *ps = reinterpret_cast<uint32_t *>(ss + NP * 3 * H);             // [N][W] (compaction mode)
    unsigned char *area = reinterpret_cast<unsigned char *>(LISTS ? ps : ps + N * W);      // block-sum area, then the lists
    uint16_t *lists = reinterpret_cast<uint16_t *>(area) + (LISTS ? 0 : wave * kGraphWalkRows * lc);      // N * W even (N % 4 == 0): 8-byte aligned
    int *cnt = reinterpret_cast<int *>(reinterpret_cast<uint16_t *>(area) + nwaves * kGraphWalkRows * lc) + wave * kGraphWalkRows;
    int *off_s = reinterpret_cast<int *>(area);
    uint16_t *ord_s = reinterpret_cast<uint16_t *>(off_s + ((N + 2) & ~1));
    if constexpr (LISTS) lists = ord_s + ((N + 3) & ~3);
    const uint64_t seed = DROP ? seed_now(a) : 0;
    if (!FUSED) {
        const float4 *src = reinterpret_cast<const float4 *>(a.g + b * N * (int64_t)HD);
        for (int e = threadIdx.x; e < NP * (HD / 4); e += blockDim.x)
            reinterpret_cast<float4 *>(gs)[e] = e < N * (HD / 4) ? src[e] : make_float4(0.f, 0.f, 0.f, 0.f);
        const float *src2 = a.stats + b * N * (int64_t)(3 * H);
        for (int e = threadIdx.x; e < NP * 3 * H; e += blockDim.x) {
            // dummy rows: f1 = 0, lse = +inf (coefficient 0), c = 0
            const int k = (e % (3 * H)) / H;
            ss[e] = e < N * 3 * H ? src2[e] : (k == 1 ? INFINITY : 0.f);
        }
    } else {
        // The per-destination step of the backward pass (gat_bwd_node_kernel) done while the graph is staged:
        // g = dOut * activation'(pre), c = <g, agg>, df1 = (1 - slope)(<g, pos> - c q) per head, four rows per wave
        // with 16 lanes per row; g and {f1, lse, c} go straight into LDS, df1 (and g, if asked for) to memory,
        // the column sums of g (the bias gradient) to one partial row per graph.
        constexpr int LPH = 2 * LPH8;                   // float4 lanes per head
        const int head = sub / LPH;
        const float4 biasv = a.bias ? *reinterpret_cast<const float4 *>(a.bias + sub * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
        float4 gsum = make_float4(0.f, 0.f, 0.f, 0.f);
        // four rows of the wave's share at a time: their twelve 16-byte loads are issued together (one block per CU:
        // nothing else would cover the latency of a row-by-row walk)
        constexpr int SR = 4;
        for (int r0 = wave * 4 + grp; r0 < NP; r0 += SR * nwaves * 4) {
            float4 dv[SR], av[SR], pv[SR], ov[SR];
#pragma unroll
            for (int u = 0; u < SR; ++u) {
                const int r = r0 + u * nwaves * 4;
                const int64_t row = b * N + min(r, N - 1);
                dv[u] = load_stream4(a.dout + row * HD + sub * 4);
                av[u] = load_stream4(a.agg + row * HD + sub * 4);
                pv[u] = load_stream4(a.pos + row * HD + sub * 4);
                ov[u] = a.out ? *reinterpret_cast<const float4 *>(a.out + row * HD + sub * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
            }
#pragma unroll
            for (int u = 0; u < SR; ++u) {
                const int r = r0 + u * nwaves * 4;
                if (r >= NP) break;
                float4 gv = make_float4(0.f, 0.f, 0.f, 0.f);
                float f1v = 0.f, lsev = INFINITY, c = 0.f;      // dummy rows: coefficient exactly 0
                if (r < N) {
                    const int64_t row = b * N + r;
                    gv = dv[u];
                    const float4 ag = av[u], po = pv[u];
                    if (a.activation != GATMI_ACT_IDENTITY) {
                        if (a.out) {                          // a residual was added: the derivative comes from the saved output
                            gv.x *= act_grad_from_out(ov[u].x, a.activation);
                            gv.y *= act_grad_from_out(ov[u].y, a.activation);
                            gv.z *= act_grad_from_out(ov[u].z, a.activation);
                            gv.w *= act_grad_from_out(ov[u].w, a.activation);
                        } else {
                            const float4 pre = f4_add(ag, biasv);
                            gv.x *= act_grad_from_pre(pre.x, a.activation);
                            gv.y *= act_grad_from_pre(pre.y, a.activation);
                            gv.z *= act_grad_from_pre(pre.z, a.activation);
                            gv.w *= act_grad_from_pre(pre.w, a.activation);
                        }
                    }
                    c = group_sum<LPH>(dot4(gv, ag));
                    const float gp = group_sum<LPH>(dot4(gv, po));
                    f1v = a.f1[row * H + head];
                    lsev = a.lse[row * H + head];
                    if (sub % LPH == 0) a.df1_out[row * H + head] = (1.0f - a.slope) * (gp - c * a.q[row * H + head]);
                    if (a.g_out) store_stream4(a.g_out + row * HD + sub * 4, gv);
                    gsum = f4_add(gsum, gv);
                }
                *reinterpret_cast<float4 *>(gs + r * HD + sub * 4) = gv;
                if (sub % LPH == 0) {
                    ss[r * 3 * H + head] = f1v;
                    ss[r * 3 * H + H + head] = lsev;
                    ss[r * 3 * H + 2 * H + head] = c;
                }
            }
        }
        if (a.dbias_ws) {
            // lanes with equal `sub` across the block, summed in a fixed order through the (not yet used) list area
            float4 *red = reinterpret_cast<float4 *>(area);
            red[threadIdx.x] = gsum;
            __syncthreads();
            if (threadIdx.x < G) {
                float4 t = make_float4(0.f, 0.f, 0.f, 0.f);
                for (int i = threadIdx.x; i < (int)blockDim.x; i += G) t = f4_add(t, red[i]);
                *reinterpret_cast<float4 *>(a.dbias_ws + b * a.ws_stride + threadIdx.x * 4) = t;
            }
            __syncthreads();                              // the list area is free again
        }
    }
    // `graph_soft`: some row of this graph carries biases that must be added to its logits (its flag word is
    // set).  Hard masks, the usual case, have none: the list building below then skips the flag words altogether.
    bool graph_soft = false;
    if constexpr (LISTS) {
        stage_graph_lists(b, N, a.lptr, a.lcol, a.lord, a.lcap, off_s, ord_s, lists);
        __syncthreads();
    } else {
        int soft_seen = 0;
        const uint32_t *src3 = a.present + b * N * (int64_t)W;
        for (int e = threadIdx.x; e < N * W; e += blockDim.x) {
            const uint32_t w = src3[e];
            ps[e] = w;
            soft_seen |= (e % W == W - 1) ? (int)w : 0;
        }
        graph_soft = __syncthreads_or(soft_seen) != 0;
    }
    const int c8 = lane >> 3, s8 = lane & 7;
    const int head8 = s8 / LPH8;
    int pa, pb;
    walk_pieces(lane, 3, pa, pb);
    const int oa = s8 * 8 + pa, ob = s8 * 8 + pb;        // columns of this lane's first and second piece
    const float *ga = gs + oa, *gb = gs + ob;
    const float *sl = ss + head8;
    for (int j0 = wave * kGraphWalkRows; j0 < N; j0 += nwaves * kGraphWalkRows) {
        bool anysoft = false;
        if constexpr (!LISTS) {
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            const int q0 = j0 + 4 * half;
            uint16_t *lrow = lists + (4 * half + grp) * lc;
            if (q0 >= N) {                                // wave-uniform: the octet's second quad lies beyond the graph
                if (sub == 0) cnt[4 * half + grp] = 0;
                continue;
            }
            const int jcol = min(q0 + grp, N - 1);
            const bool col_ok = q0 + grp < N;
            // bit r of `bits`: destination row 16 r + sub visited column jcol; `softs`: and needs its bias value
            uint32_t bits = 0, softs = 0;
            const int wj = jcol >> 5, bj = jcol & 31;
            if (graph_soft) {
#pragma unroll
                for (int r = 0; r < kGraphBwdRows; ++r) {
                    const int i = 16 * r + sub;
                    if (16 * r < N) {                         // uniform
                        const int ic = min(i, N - 1);
                        const uint32_t word = ps[ic * W + wj], flag = ps[ic * W + W - 1];
                        const uint32_t bit = i < N ? (word >> bj) & 1u : 0u;
                        bits |= bit << r;
                        softs |= (bit & flag) << r;
                    }
                }
            } else {
#pragma unroll
                for (int r = 0; r < kGraphBwdRows; ++r) {
                    const int i = 16 * r + sub;
                    if (16 * r < N) {                         // uniform
                        const uint32_t word = ps[min(i, N - 1) * W + wj];
                        bits |= (i < N ? (word >> bj) & 1u : 0u) << r;
                    }
                }
            }
            const int cntl = __popc(bits);
            const int total = row16_sum(cntl);
            int pos = row16_scan(cntl) - cntl;
            anysoft |= __ballot(softs != 0) != 0;
            for (uint32_t rest = bits; __ballot(rest != 0) != 0;) {          // each lane appends its rows at its offset
                if (rest != 0) {
                    const int t = __ffs(rest) - 1;
                    lrow[pos++] = (uint16_t)((16 * t + sub) | (((softs >> t) & 1u) << 15));
                    rest &= rest - 1;
                }
            }
            if (sub < kGraphBatch) lrow[total + sub] = (uint16_t)N;          // padding: the dummy row
            if (sub == 0) cnt[4 * half + grp] = col_ok ? total : 0;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        }
        const bool col_ok = j0 + c8 < N;
        int jcol, total;
        const uint16_t *lrow;
        if constexpr (LISTS) {
            jcol = ord_s[min(j0 + c8, N - 1)];
            const int o0 = off_s[jcol];
            total = col_ok ? off_s[jcol + 1] - o0 : 0;
            lrow = lists + o0;
        } else {
            jcol = min(j0 + c8, N - 1);
            total = cnt[c8];
            lrow = lists + c8 * lc;
        }
        const int64_t j = b * N + jcol;
        const float f2v = a.f2[j * H + head8];
        float4 wh0 = *reinterpret_cast<const float4 *>(a.Wh + j * HD + oa);
        float4 wh1 = *reinterpret_cast<const float4 *>(a.Wh + j * HD + ob);
        if (DROP) {
            if (a.feat_keep < 1.f) {
                wh0 = drop_feat(wh0, seed, j, HD, oa, a.feat_keep);
                wh1 = drop_feat(wh1, seed, j, HD, ob, a.feat_keep);
            }
        }
        const uint64_t cbase = DROP ? ((uint64_t)(b * N) * (uint64_t)N + (uint64_t)jcol) * H + head8 : 0;
        MSrcState8 st;
        st.acc0 = st.acc1 = make_float4(0.f, 0.f, 0.f, 0.f);
        st.df2 = 0.f;
        {
            const float *bmcol = LISTS ? nullptr : a.bias_mat + b * N * (int64_t)N + jcol;
            if (!LISTS && anysoft) {
                for (int k = 0; k < total; k += kGraphBatch)
                    graph_bwd_batch8<LPH8, !LISTS, DROP>(st, a.slope, lrow + k, bmcol, N, ga, gb, sl, f2v, wh0, wh1, seed, cbase, a.coef_keep);
            } else {
                for (int k = 0; k < total; k += kGraphBatch)
                    graph_bwd_batch8<LPH8, false, DROP>(st, a.slope, lrow + k, bmcol, N, ga, gb, sl, f2v, wh0, wh1, seed, cbase, a.coef_keep);
            }
        }
        if constexpr (!LISTS) __builtin_amdgcn_wave_barrier();   // every lane is done with the lists before the next columns overwrite them
        if (col_ok) {
            float4 r0 = st.acc0, r1 = st.acc1;
            if (DROP) {
                if (a.feat_keep < 1.f) {
                    r0 = drop_feat(r0, seed, j, HD, oa, a.feat_keep);
                    r1 = drop_feat(r1, seed, j, HD, ob, a.feat_keep);
                }
            }
            if (a.a1) {
                const float d1 = (FUSED ? a.df1_out : a.df1)[j * H + head8];
                r0 = f4_fma(d1, *reinterpret_cast<const float4 *>(a.a1 + oa), r0);
                r1 = f4_fma(d1, *reinterpret_cast<const float4 *>(a.a1 + ob), r1);
                r0 = f4_fma(st.df2, *reinterpret_cast<const float4 *>(a.a2 + oa), r0);
                r1 = f4_fma(st.df2, *reinterpret_cast<const float4 *>(a.a2 + ob), r1);
            }
            store_stream4(a.dWh + j * HD + oa, r0);
            store_stream4(a.dWh + j * HD + ob, r1);
            if (s8 % LPH8 == 0) a.df2[j * H + head8] = st.df2;
        }
    }
}

// ------------------------------------------------------------------------------------------
// The fused backward kernel over prebuilt lists, split by halves of the 64-float row.  Block (graph, half)
// keeps 32 floats of every g row and the statistics of that half's heads: under 49 KB at 264 nodes, so
// THREE blocks of four waves share a CU (the same twelve waves as one whole-row block), and one block's
// memory phase (the per-destination step reads dOut, agg and pos of its graph) runs under the other
// blocks' walks, where a single block per CU alternates between the two.  A column belongs to four lanes
// of eight floats, sixteen columns to a wave; the lists are read from memory as in
// dense_graph_fwd8_lists_kernel (a lane fetches two batches of four ids, the group passes them round),
// columns in the order of their list lengths.  Heads must not straddle the halves: H >= 2.
// Every (column, head) adds the same terms in the same order as dense_graph_bwd8_kernel; only the partial
// sums of the bias gradient are grouped differently.  param_grads: the block also sums df1_j * Wh_j, df2_j * Wh_j
// (da1, da2: utils/layers.py:14-15) and df1, df2 (db1, db2) over its columns while it holds them, so that no later
// pass has to read Wh again for them.
// ------------------------------------------------------------------------------------------
constexpr int kHalfFloats = 32;            // floats of a row per block
constexpr int kHalfCols = 16;              // columns per wave in the four-lane walk
// LDS: g [N + pad][32] | stats [N + pad][3 H/2] | max(list offsets + column order, the block sums: one float4 per
// row-slice lane and wave for the bias gradient, eighteen floats per lane slot and wave for da1, da2, db1, db2)
constexpr int kHalfParamSums = 18;         // per lane: da1 and da2 of its eight columns, db1, db2 of its head
// bytes of the rows' part, rounded up to the 16 bytes the float4 sums behind it need (any N: prebuilt lists do not ask for N % 4 == 0)
__host__ __device__ inline size_t graph8_bwd_half_rows(int N, int heads_of_half) {
    return ((size_t)(N + kGraphPadRows) * (kHalfFloats + 3 * heads_of_half) * sizeof(float) + 15) & ~(size_t)15;
}
static inline size_t graph8_bwd_half_lds(int N, int H, int waves) {
    const size_t sums = std::max((size_t)waves * 8 * sizeof(float4), (size_t)waves * 4 * kHalfParamSums * sizeof(float));
    return graph8_bwd_half_rows(N, H / 2) + std::max(graph8_lists_head_lds(N), sums);
}

// Block id -> (graph, half): workgroups are dealt round-robin over the 8 XCDs, so the two halves of a graph get
// ids 8 apart and share one L2 for the graph's lists.  Launch ceil(B / 8) * 16 blocks; graphs past B exit.
__device__ __forceinline__ void graph_and_half(int64_t &graph, int &half) {
    const int64_t L = blockIdx.x;
    half = (int)((L >> 3) & 1);
    graph = (L >> 4) * 8 + (L & 7);
}

template <int LPH8, int WAVES, bool DROP>
__global__ __launch_bounds__(64 * WAVES, 12 / WAVES) void dense_graph_bwd8_half_kernel(MaskedBwdArgs a) {
    static_assert(LPH8 == 1 || LPH8 == 2 || LPH8 == 4, "a half row holds whole heads");
    constexpr int H = 8 / LPH8;            // heads of the layer
    constexpr int HB = H / 2;              // heads of this block
    constexpr int HD = 64, HW = kHalfFloats;
    constexpr int SS = 3 * HB;             // statistics per row: {f1, lse, c} of each head
    extern __shared__ __attribute__((aligned(16))) unsigned char graph_smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int N = a.N;
    const int NP = N + kGraphPadRows;
    int64_t b;
    int hf;
    graph_and_half(b, hf);
    if (b >= a.B) return;
    float *gs = reinterpret_cast<float *>(graph_smem);                        // [NP][HW]
    float *ss = gs + NP * HW;                                                 // [NP][SS]
    unsigned char *area = graph_smem + graph8_bwd_half_rows(N, HB);
    int *off_s = reinterpret_cast<int *>(area);
    uint16_t *ord_s = reinterpret_cast<uint16_t *>(off_s + ((N + 2) & ~1));
    const uint64_t seed = DROP ? seed_now(a) : 0;
    {
        // the per-destination step (see dense_graph_bwd8_kernel), eight lanes of one float4 per row slice
        constexpr int LPH = 2 * LPH8;                   // float4 lanes per head
        const int sub = lane & 7, rg = lane >> 3;
        const int headl = sub / LPH, head = hf * HB + headl;
        const int col = hf * HW + sub * 4;
        const float4 biasv = a.bias ? *reinterpret_cast<const float4 *>(a.bias + col) : make_float4(0.f, 0.f, 0.f, 0.f);
        float4 gsum = make_float4(0.f, 0.f, 0.f, 0.f);
        constexpr int SR = 4;                           // row slices whose loads are issued together
        for (int r0 = wave * 8 + rg; r0 < NP; r0 += SR * WAVES * 8) {
            float4 dv[SR], av[SR], pv[SR], ov[SR];
#pragma unroll
            for (int u = 0; u < SR; ++u) {
                const int64_t row = b * N + min(r0 + u * WAVES * 8, N - 1);
                dv[u] = load_stream4(a.dout + row * HD + col);
                av[u] = load_stream4(a.agg + row * HD + col);
                pv[u] = load_stream4(a.pos + row * HD + col);
                ov[u] = a.out ? *reinterpret_cast<const float4 *>(a.out + row * HD + col) : make_float4(0.f, 0.f, 0.f, 0.f);
            }
#pragma unroll
            for (int u = 0; u < SR; ++u) {
                const int r = r0 + u * WAVES * 8;
                if (r >= NP) break;
                float4 gv = make_float4(0.f, 0.f, 0.f, 0.f);
                float f1v = 0.f, lsev = INFINITY, c = 0.f;      // dummy rows: coefficient exactly 0
                if (r < N) {
                    const int64_t row = b * N + r;
                    gv = dv[u];
                    const float4 ag = av[u], po = pv[u];
                    if (a.activation != GATMI_ACT_IDENTITY) {
                        if (a.out) {
                            gv.x *= act_grad_from_out(ov[u].x, a.activation);
                            gv.y *= act_grad_from_out(ov[u].y, a.activation);
                            gv.z *= act_grad_from_out(ov[u].z, a.activation);
                            gv.w *= act_grad_from_out(ov[u].w, a.activation);
                        } else {
                            const float4 pre = f4_add(ag, biasv);
                            gv.x *= act_grad_from_pre(pre.x, a.activation);
                            gv.y *= act_grad_from_pre(pre.y, a.activation);
                            gv.z *= act_grad_from_pre(pre.z, a.activation);
                            gv.w *= act_grad_from_pre(pre.w, a.activation);
                        }
                    }
                    c = group_sum<LPH>(dot4(gv, ag));
                    const float gp = group_sum<LPH>(dot4(gv, po));
                    f1v = a.f1[row * H + head];
                    lsev = a.lse[row * H + head];
                    if (sub % LPH == 0) a.df1_out[row * H + head] = (1.0f - a.slope) * (gp - c * a.q[row * H + head]);
                    if (a.g_out) store_stream4(a.g_out + row * HD + col, gv);
                    gsum = f4_add(gsum, gv);
                }
                *reinterpret_cast<float4 *>(gs + r * HW + sub * 4) = gv;
                if (sub % LPH == 0) {
                    ss[r * SS + headl] = f1v;
                    ss[r * SS + HB + headl] = lsev;
                    ss[r * SS + 2 * HB + headl] = c;
                }
            }
        }
        if (a.dbias_ws) {
            // column sums of g over the block in a fixed order: the eight row groups of a wave, then the waves
#pragma unroll
            for (int o = 8; o < 64; o <<= 1) {
                gsum.x += __shfl_xor(gsum.x, o, 64);
                gsum.y += __shfl_xor(gsum.y, o, 64);
                gsum.z += __shfl_xor(gsum.z, o, 64);
                gsum.w += __shfl_xor(gsum.w, o, 64);
            }
            float4 *red = reinterpret_cast<float4 *>(area);
            if (lane < 8) red[wave * 8 + lane] = gsum;
            __syncthreads();
            if (threadIdx.x < 8) {
                float4 t = red[threadIdx.x];
                for (int w = 1; w < WAVES; ++w) t = f4_add(t, red[w * 8 + threadIdx.x]);
                *reinterpret_cast<float4 *>(a.dbias_ws + b * a.ws_stride + hf * HW + threadIdx.x * 4) = t;
            }
            __syncthreads();                              // the area is free for the list offsets
        }
    }
    const int32_t base = a.lptr[b * N];
    for (int r = threadIdx.x; r <= N; r += 64 * WAVES) off_s[r] = a.lptr[b * N + r] - base;
    for (int r = threadIdx.x; r < N; r += 64 * WAVES) ord_s[r] = a.lord[b * N + r];
    __syncthreads();
    const int c4 = lane >> 2, s4 = lane & 3;
    const int headl = s4 / LPH8, head = hf * HB + headl;
    // (walk_pieces: neighbouring columns may take their two pieces in opposite order.  The four columns that one LDS
    // cycle serves read rows that start on one of two bank halves at random: with one order for all, four columns
    // share two sixteen-bank sets, 2.75 cycles on average; alternating, 1.75.)
    int pa, pb;
    walk_pieces(lane, 2, pa, pb);
    const int la = s4 * 8 + pa, lb = s4 * 8 + pb;          // within the half row
    const int oa = hf * HW + la, ob = hf * HW + lb;        // within the whole row
    const float *ga = gs + la, *gb = gs + lb;
    const float *sl = ss + headl;
    const uint16_t *gcol = a.lcol + base;
    const uint32_t dummy = (uint32_t)N | ((uint32_t)N << 16);      // the dummy row: lse = +inf, coefficient exactly 0
    const int lane0 = lane & 60;                                    // first lane of this column's group
    // batches s4 and 4 + s4 of the eight that start at entry c0 of a column's list; past the list: the dummy id.
    // The loads are unconditional (clamped inside the list) so that the request for the NEXT columns stays in
    // flight across the walk of the current ones.
    auto fetch = [&](int o0, int total, int c0) -> uint4 {
        const int at0 = c0 + 4 * s4, at1 = at0 + 16;
        const int last = max(total - 4, 0);
        const uint2 v0 = *reinterpret_cast<const uint2 *>(gcol + o0 + min(at0, last));
        const uint2 v1 = *reinterpret_cast<const uint2 *>(gcol + o0 + min(at1, last));
        return make_uint4(at0 < total ? v0.x : dummy, at0 < total ? v0.y : dummy, at1 < total ? v1.x : dummy, at1 < total ? v1.y : dummy);
    };
    auto columns = [&](int j0, int &jcol, int &o0, int &total) {
        jcol = ord_s[min(j0 + c4, N - 1)];
        o0 = off_s[jcol];
        total = j0 + c4 < N ? off_s[jcol + 1] - o0 : 0;
    };
    float4 da1a = make_float4(0.f, 0.f, 0.f, 0.f), da1b = da1a, da2a = da1a, da2b = da1a;
    float db1s = 0.f, db2s = 0.f;
    int j0 = wave * kHalfCols;
    int jcol = 0, o0 = 0, total = 0;
    uint4 ch = make_uint4(dummy, dummy, dummy, dummy);
    if (j0 < N) {
        columns(j0, jcol, o0, total);
        ch = fetch(o0, total, 0);
    }
    for (; j0 < N; j0 += WAVES * kHalfCols) {
        const int jn0 = j0 + WAVES * kHalfCols;
        int jn, on, tn;
        columns(min(jn0, N - 1), jn, on, tn);                       // past the graph: the clamped column, never walked
        if (jn0 >= N) tn = 0;
        const uint4 chn = fetch(on, tn, 0);
        __builtin_amdgcn_sched_barrier(0);
        const bool col_ok = j0 + c4 < N;
        const int64_t j = b * N + jcol;
        const float f2v = a.f2[j * H + head];
        float4 wh0 = *reinterpret_cast<const float4 *>(a.Wh + j * HD + oa);
        float4 wh1 = *reinterpret_cast<const float4 *>(a.Wh + j * HD + ob);
        const float4 whu0 = wh0, whu1 = wh1;                         // the logits were formed from the undropped rows
        if (DROP) {
            if (a.feat_keep < 1.f) {
                wh0 = drop_feat(wh0, seed, j, HD, oa, a.feat_keep);
                wh1 = drop_feat(wh1, seed, j, HD, ob, a.feat_keep);
            }
        }
        const uint64_t cbase = DROP ? ((uint64_t)(b * N) * (uint64_t)N + (uint64_t)jcol) * H + head : 0;
        MSrcState8 st;
        st.acc0 = st.acc1 = make_float4(0.f, 0.f, 0.f, 0.f);
        st.df2 = 0.f;
        for (int c0 = 0; __any(c0 < total); c0 += 32) {
            if (c0 > 0) ch = fetch(o0, total, c0);                  // lists of more than 32 entries: further chunks as they come
#pragma unroll
            for (int t = 0; t < 8; ++t) {
                if (!__any(c0 + 4 * t < total)) break;              // wave-uniform
                const uint32_t w0 = (uint32_t)__shfl((int)(t < 4 ? ch.x : ch.z), lane0 + (t & 3), 64);
                const uint32_t w1 = (uint32_t)__shfl((int)(t < 4 ? ch.y : ch.w), lane0 + (t & 3), 64);
                int i[kGraphBatch] = {(int)(w0 & 0xffff), (int)(w0 >> 16), (int)(w1 & 0xffff), (int)(w1 >> 16)};
                graph_bwd_batch8_ids<LPH8, false, DROP, HW, HB>(st, a.slope, i, nullptr, N, ga, gb, sl, f2v, wh0, wh1, seed, cbase, a.coef_keep);
            }
        }
        if (col_ok) {
            float4 r0 = st.acc0, r1 = st.acc1;
            if (DROP) {
                if (a.feat_keep < 1.f) {
                    r0 = drop_feat(r0, seed, j, HD, oa, a.feat_keep);
                    r1 = drop_feat(r1, seed, j, HD, ob, a.feat_keep);
                }
            }
            if (a.a1) {
                const float d1 = a.df1_out[j * H + head];
                r0 = f4_fma(d1, *reinterpret_cast<const float4 *>(a.a1 + oa), r0);
                r1 = f4_fma(d1, *reinterpret_cast<const float4 *>(a.a1 + ob), r1);
                r0 = f4_fma(st.df2, *reinterpret_cast<const float4 *>(a.a2 + oa), r0);
                r1 = f4_fma(st.df2, *reinterpret_cast<const float4 *>(a.a2 + ob), r1);
                if (a.param_grads) {
                    da1a = f4_fma(d1, whu0, da1a);
                    da1b = f4_fma(d1, whu1, da1b);
                    da2a = f4_fma(st.df2, whu0, da2a);
                    da2b = f4_fma(st.df2, whu1, da2b);
                    db1s += d1;
                    db2s += st.df2;
                }
            }
            store_stream4(a.dWh + j * HD + oa, r0);
            store_stream4(a.dWh + j * HD + ob, r1);
            if (s4 % LPH8 == 0) a.df2[j * H + head] = st.df2;
        }
        jcol = jn; o0 = on; total = tn; ch = chn;
    }
    if (a.param_grads) {
        // block sums in a fixed order: the sixteen columns of a wave (lanes with equal s4: two rotations inside the
        // rows of sixteen lanes, then across the rows), then the waves through the area the list offsets leave free
        float v[kHalfParamSums] = {da1a.x, da1a.y, da1a.z, da1a.w, da1b.x, da1b.y, da1b.z, da1b.w,
                                   da2a.x, da2a.y, da2a.z, da2a.w, da2b.x, da2b.y, da2b.z, da2b.w, db1s, db2s};
#pragma unroll
        for (int k = 0; k < kHalfParamSums; ++k) {
            float t = v[k];
            t += dpp_f<0x124>(t, t);                      // row_ror:4
            t += dpp_f<0x128>(t, t);                      // row_ror:8: lanes i, i+4, i+8, i+12 of the row
            t += __shfl_xor(t, 16, 64);
            t += __shfl_xor(t, 32, 64);
            v[k] = t;
        }
        __syncthreads();                                  // every wave has finished its walk: offsets and order are dead
        float *red = reinterpret_cast<float *>(area);     // [WAVES][4][18]
        if (lane < 4) {
#pragma unroll
            for (int k = 0; k < kHalfParamSums; ++k) red[(wave * 4 + lane) * kHalfParamSums + k] = v[k];
        }
        __syncthreads();
        if (threadIdx.x < 4) {                            // lane slot s4 = threadIdx.x: pa, pb, head as computed above
#pragma unroll
            for (int k = 0; k < kHalfParamSums; ++k) {
                float t = red[threadIdx.x * kHalfParamSums + k];
                for (int w = 1; w < WAVES; ++w) t += red[(w * 4 + threadIdx.x) * kHalfParamSums + k];
                v[k] = t;
            }
            float *row = a.dbias_ws + b * a.ws_stride;
            *reinterpret_cast<float4 *>(row + HD + oa) = make_float4(v[0], v[1], v[2], v[3]);
            *reinterpret_cast<float4 *>(row + HD + ob) = make_float4(v[4], v[5], v[6], v[7]);
            *reinterpret_cast<float4 *>(row + 2 * HD + oa) = make_float4(v[8], v[9], v[10], v[11]);
            *reinterpret_cast<float4 *>(row + 2 * HD + ob) = make_float4(v[12], v[13], v[14], v[15]);
            if (s4 % LPH8 == 0) {
                row[3 * HD + head] = v[16];
                row[3 * HD + H + head] = v[17];
            }
        }
    }
}

// out[c] = sum over the per-graph partial rows ws[b][c] (fixed order: reproducible)
__global__ void graph_dbias_sum_kernel(const float *ws, int rows, int width, float *out) {
    const int c = blockIdx.x * kSumCols + threadIdx.x;
    const bool ok = c < width;
    const float t = block_column_sum(rows, [&](int r) { return ok ? ws[(int64_t)r * width + c] : 0.f; });
    if (ok && threadIdx.y == 0) out[c] = t;
}

// the same over rows of [dbias 64 | da1 64 | da2 64 | db1 H | db2 H]: each column's total goes to its own array
__global__ void graph_param_sum_kernel(const float *ws, int rows, int width, int H, float *dbias, float *da1, float *da2, float *db1, float *db2) {
    const int c = blockIdx.x * kSumCols + threadIdx.x;
    const bool ok = c < width;
    const float t = block_column_sum(rows, [&](int r) { return ok ? ws[(int64_t)r * width + c] : 0.f; });
    if (!ok || threadIdx.y != 0) return;
    if (c < 64) { if (dbias) dbias[c] = t; }
    else if (c < 128) da1[c - 64] = t;
    else if (c < 192) da2[c - 128] = t;
    else if (c < 192 + H) db1[c - 192] = t;
    else db2[c - 192 - H] = t;
}

// first stage for many graphs: block (bx, by) sums rows [by * per, (by + 1) * per) into out[by][c]
constexpr int kDbiasStageRows = 64;
__global__ void graph_dbias_stage_kernel(const float *ws, int rows, int width, float *out) {
    const int c = blockIdx.x * kSumCols + threadIdx.x;
    const bool ok = c < width;
    const int r0 = blockIdx.y * kDbiasStageRows;
    const float t = block_column_sum(min(kDbiasStageRows, rows - r0), [&](int r) { return ok ? ws[(int64_t)(r0 + r) * width + c] : 0.f; });
    if (ok && threadIdx.y == 0) out[(int64_t)blockIdx.y * width + c] = t;
}

// ------------------------------------------------------------------------------------------
// Prebuilt lists for a mask that stays the same step after step (the reference feeds one bias_mat
// every epoch, execute_cora.py:100-107): bits of the visited entries, list lengths, lists.
// ------------------------------------------------------------------------------------------
// One wave per mask row: the `present` words of that row (bit j of the row = entry j is within 1e6 of the row
// maximum) and, in the flag word, whether the bias values matter to the row's softmax, so that lists alone cannot
// stand for the mask: some visited entry differs from the row maximum, or several entries share a maximum so
// large that fp32 rounds the logits added to it (a row masked everywhere attends uniformly, utils/layers.py:17).
__global__ __launch_bounds__(256) void mask_bits_kernel(const float *bias_mat, int N, int W, float shared_limit, uint32_t *bits) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t b = blockIdx.x;
    const int i = blockIdx.y * 4 + wave;
    if (i >= N) return;
    const float *bm = bias_mat + (b * N + i) * (int64_t)N;
    float mx = -INFINITY;
    for (int j = lane; j < N; j += 64) mx = fmaxf(mx, bm[j]);
    mx = wave_max(mx);
    const float cut = mx - kMaskCutoffFwd;
    uint32_t *prow = bits + (b * N + i) * (int64_t)W;
    uint64_t differs = 0;
    int visited = 0;
    for (int j0 = 0; j0 < N; j0 += 64) {
        const int j = j0 + lane;
        const float v = j < N ? bm[j] : -INFINITY;
        const bool present = v > cut;
        const uint64_t bal = __ballot(present);
        visited += __popcll(bal);
        differs |= __ballot(present && v != mx);
        if (lane < 2 && (j0 >> 5) + lane < W - 1) prow[(j0 >> 5) + lane] = (uint32_t)(bal >> (32 * lane));
    }
    if (lane == 0) prow[W - 1] = (differs != 0 || (fabsf(mx) > shared_limit && visited > 1) || !(mx == mx)) ? 1u : 0u;
}

constexpr int kListsThreads = 256;

// block per graph: padded list lengths of its rows (visited columns) and of its columns (visiting rows)
__global__ __launch_bounds__(kListsThreads) void lists_count_kernel(const uint32_t *bits, int N, int W, int32_t *row_cnt, int32_t *col_cnt) {
    extern __shared__ uint32_t lists_bits[];
    const int64_t b = blockIdx.x;
    for (int e = threadIdx.x; e < N * W; e += blockDim.x) lists_bits[e] = bits[b * N * (int64_t)W + e];
    __syncthreads();
    for (int r = threadIdx.x; r < N; r += blockDim.x) {
        int n = 0;
        for (int w = 0; w < W - 1; ++w) n += __popc(lists_bits[r * W + w]);
        row_cnt[b * N + r] = (n + kGraphBatch - 1) / kGraphBatch * kGraphBatch;
    }
    for (int c = threadIdx.x; c < N; c += blockDim.x) {
        int n = 0;
        for (int i = 0; i < N; ++i) n += (lists_bits[i * W + (c >> 5)] >> (c & 31)) & 1u;
        col_cnt[b * N + c] = (n + kGraphBatch - 1) / kGraphBatch * kGraphBatch;
    }
}

// block per graph: the lists themselves, each in the order in which the compacting kernels build it (a lane group of
// 16 appends lane by lane: lane `sub` holds columns 64k + 4 sub + c of a row, rows 16 r + sub of a column), so that
// the kernels that walk prebuilt lists add the same terms in the same order as the kernels that read the mask.
__global__ __launch_bounds__(kListsThreads) void lists_fill_kernel(const uint32_t *bits, int N, int W, const int32_t *lptr, const int32_t *cptr,
                                                                   uint16_t *lcol, uint16_t *crow) {
    extern __shared__ uint32_t lists_bits[];
    const int64_t b = blockIdx.x;
    for (int e = threadIdx.x; e < N * W; e += blockDim.x) lists_bits[e] = bits[b * N * (int64_t)W + e];
    __syncthreads();
    for (int r = threadIdx.x; r < N; r += blockDim.x) {
        int at = lptr[b * N + r];
        const int end = lptr[b * N + r + 1];
        for (int sub = 0; sub < 16; ++sub)
            for (int k = 0; 64 * k + 4 * sub < N; ++k)
                for (int c = 0; c < 4; ++c) {
                    const int col = 64 * k + 4 * sub + c;
                    if (col < N && ((lists_bits[r * W + (col >> 5)] >> (col & 31)) & 1u) && at < end) lcol[at++] = (uint16_t)col;
                }
        while (at < end) lcol[at++] = (uint16_t)N;
    }
    for (int c = threadIdx.x; c < N; c += blockDim.x) {
        int at = cptr[b * N + c];
        const int end = cptr[b * N + c + 1];
        for (int sub = 0; sub < 16; ++sub)
            for (int i = sub; i < N; i += 16)
                if (((lists_bits[i * W + (c >> 5)] >> (c & 31)) & 1u) && at < end) crow[at++] = (uint16_t)i;
        while (at < end) crow[at++] = (uint16_t)N;
    }
}

// (G, LPH) pairs built: rows of 32..256 floats (narrower rows leave most of a wave idle in the
// compaction; those shapes stay with the matrix-core kernels)
#define GATMI_MASKED_SHAPES(X) X(8, 1) X(8, 2) X(8, 4) X(8, 8) X(16, 1) X(16, 2) X(16, 4) X(16, 8) X(16, 16) \
    X(32, 1) X(32, 2) X(32, 4) X(32, 8) X(32, 16) X(32, 32) X(64, 1) X(64, 2) X(64, 4) X(64, 8) X(64, 16) X(64, 32) X(64, 64)

}  // namespace gatmi

using namespace gatmi;

// shapes of the graph-resident kernels with the eight-lane walk: rows of 64 floats in heads of >= 8, graphs of
// 4 .. 320 nodes (a multiple of 4), one block per graph and at least one block per CU
// (N a multiple of 4 is what the mask-reading kernels need for their 16-byte mask loads; kernels that walk prebuilt lists
// from memory take any N: lists_only)
static bool graph8_shape(int64_t B, int32_t N, int G, int LPH, bool lists_only = false) {
    return G == 16 && LPH >= 2 && (lists_only || (N & 3) == 0) && N >= 4 && N <= kGraphMaxN && B >= kGraphMinBatch && B <= 0x7fffffff;
}

static int launch_graph_fwd8(const MaskedFwdArgs &a, int LPH, bool save, bool drop, size_t lds, int lc, hipStream_t st, const char *name) {
    const int waves = graph8_waves(a.N);
    hipError_t e = hipSuccess;
#define GATMI_L(ll, sv, dr)                                                                                          \
    {                                                                                                                \
        if (lds > 48 * 1024) e = hipFuncSetAttribute(reinterpret_cast<const void *>(&dense_graph_fwd8_kernel<ll, sv, dr>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
        if (e == hipSuccess) hipLaunchKernelGGL((dense_graph_fwd8_kernel<ll, sv, dr>), dim3((unsigned)a.B), dim3(64 * waves), lds, st, a, lc); \
    }
#define X(ll)                                                                                                        \
    if (LPH == 2 * ll) {                                                                                            \
        if (drop) GATMI_L(ll, true, true) else if (save) GATMI_L(ll, true, false) else GATMI_L(ll, false, false)     \
        if (e != hipSuccess) return fail(GATMI_EHIP, "%s: %s", name, hipGetErrorString(e));                         \
        GATMI_LAUNCH_CHECK(name);                                                                                   \
        return GATMI_OK;                                                                                            \
    }
    X(1) X(2) X(4) X(8)
#undef X
#undef GATMI_L
    return fail(GATMI_EUNSUPPORTED, "%s: no kernel for %d lanes per head", name, LPH);
}

static int launch_graph_fwd8_lists(const MaskedFwdArgs &a, int LPH, bool save, bool drop, size_t lds, hipStream_t st, const char *name) {
    const int waves = graph8_lists_waves(a.N);
    hipError_t e = hipSuccess;
#define GATMI_L(ll, sv, dr)                                                                                          \
    {                                                                                                                \
        if (lds > 48 * 1024) e = hipFuncSetAttribute(reinterpret_cast<const void *>(&dense_graph_fwd8_lists_kernel<ll, sv, dr>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
        if (e == hipSuccess) hipLaunchKernelGGL((dense_graph_fwd8_lists_kernel<ll, sv, dr>), dim3((unsigned)a.B), dim3(64 * waves), lds, st, a); \
    }
#define X(ll)                                                                                                        \
    if (LPH == 2 * ll) {                                                                                            \
        if (drop) GATMI_L(ll, true, true) else if (save) GATMI_L(ll, true, false) else GATMI_L(ll, false, false)     \
        if (e != hipSuccess) return fail(GATMI_EHIP, "%s: %s", name, hipGetErrorString(e));                         \
        GATMI_LAUNCH_CHECK(name);                                                                                   \
        return GATMI_OK;                                                                                            \
    }
    X(1) X(2) X(4) X(8)
#undef X
#undef GATMI_L
    return fail(GATMI_EUNSUPPORTED, "%s: no kernel for %d lanes per head", name, LPH);
}

template <bool LISTS>
static int launch_graph_bwd8_fused(const MaskedBwdArgs &a, int LPH, bool drop, size_t lds, int lc, hipStream_t st, const char *name) {
    const int waves = graph8_waves(a.N);
    hipError_t e = hipSuccess;
#define GATMI_L(ll, dr)                                                                                              \
    {                                                                                                                \
        if (lds > 48 * 1024) e = hipFuncSetAttribute(reinterpret_cast<const void *>(&dense_graph_bwd8_kernel<ll, true, LISTS, dr>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
        if (e == hipSuccess) hipLaunchKernelGGL((dense_graph_bwd8_kernel<ll, true, LISTS, dr>), dim3((unsigned)a.B), dim3(64 * waves), lds, st, a, lc); \
    }
#define X(ll)                                                                                                        \
    if (LPH == 2 * ll) {                                                                                            \
        if (drop) GATMI_L(ll, true) else GATMI_L(ll, false)                                                          \
        if (e != hipSuccess) return fail(GATMI_EHIP, "%s: %s", name, hipGetErrorString(e));                         \
        GATMI_LAUNCH_CHECK(name);                                                                                   \
        return GATMI_OK;                                                                                            \
    }
    X(1) X(2) X(4) X(8)
#undef X
#undef GATMI_L
    return fail(GATMI_EUNSUPPORTED, "%s: no kernel for %d lanes per head", name, LPH);
}

constexpr int kHalfWaves = 4;              // three blocks per CU: twelve waves, 168 registers per lane (six waves x two blocks measured slower)
static int launch_graph_bwd8_half(const MaskedBwdArgs &a, int LPH, bool drop, hipStream_t st, const char *name) {
    const int H = 16 / LPH;
    const size_t lds = graph8_bwd_half_lds(a.N, H, kHalfWaves);
    const unsigned blocks = (unsigned)((a.B + 7) / 8 * 16);
    hipError_t e = hipSuccess;
#define GATMI_L(ll, dr)                                                                                              \
    {                                                                                                                \
        if (lds > 48 * 1024) e = hipFuncSetAttribute(reinterpret_cast<const void *>(&dense_graph_bwd8_half_kernel<ll, kHalfWaves, dr>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
        if (e == hipSuccess) hipLaunchKernelGGL((dense_graph_bwd8_half_kernel<ll, kHalfWaves, dr>), dim3(blocks), dim3(64 * kHalfWaves), lds, st, a); \
    }
#define X(ll)                                                                                                        \
    if (LPH == 2 * ll) {                                                                                            \
        if (drop) GATMI_L(ll, true) else GATMI_L(ll, false)                                                          \
        if (e != hipSuccess) return fail(GATMI_EHIP, "%s: %s", name, hipGetErrorString(e));                         \
        GATMI_LAUNCH_CHECK(name);                                                                                   \
        return GATMI_OK;                                                                                            \
    }
    X(1) X(2) X(4)
#undef X
#undef GATMI_L
    return fail(GATMI_EUNSUPPORTED, "%s: no kernel for %d lanes per head", name, LPH);
}

static int masked_check(int64_t B, int32_t N, int32_t H, int32_t dp, int *G, int *LPH, const char *name) {
    int rc = check_shape(H, dp, G, LPH);
    if (rc) return rc;
    GATMI_REQUIRE(B >= 0 && N >= 0, GATMI_EINVAL, "%s: negative size", name);
    GATMI_REQUIRE(*G >= 8, GATMI_EUNSUPPORTED, "%s: rows of H*dp = %d floats are too narrow for the list kernels (>= 32)", name, 4 * *G);
    GATMI_REQUIRE(N <= kMaskedMaxN, GATMI_EUNSUPPORTED, "%s: N = %d does not fit the 15-bit node ids of the lists", name, N);
    GATMI_REQUIRE(B <= 0x7fffffff && (int64_t)(N + 3) / 4 <= 65535, GATMI_ERANGE, "%s: too large for one launch", name);
    return GATMI_OK;
}

extern "C" int64_t gatmi_dense_lists_capacity(int32_t N, int32_t H, int32_t dp, int32_t backward);

extern "C" size_t gatmi_dense_mask_words(int32_t N) { return N < 0 ? 0 : (size_t)(N + 31) / 32 + 1; }

extern "C" int gatmi_dense_attn_fwd_masked(int64_t B, int32_t N, int32_t H, int32_t dp,
                                           const float *Wh, const float *f1, const float *f2, const float *bias_mat,
                                           const float *bias, const float *resid, int32_t activation, float slope,
                                           float coef_keep, float feat_keep, uint64_t seed,
                                           float *out, float *agg, float *pos, float *lse, float *q,
                                           uint32_t *present, void *stream) {
    int G, LPH;
    int rc = masked_check(B, N, H, dp, &G, &LPH, "dense_attn_fwd_masked");
    if (rc) return rc;
    if (B == 0 || N == 0) return GATMI_OK;
    GATMI_REQUIRE(Wh && f1 && f2 && bias_mat && out, GATMI_EINVAL, "dense_attn_fwd_masked: null pointer");
    GATMI_REQUIRE(act_known(activation), GATMI_EUNSUPPORTED, "dense_attn_fwd_masked: activation %d", activation);
    const bool save = agg || pos || lse || q;
    GATMI_REQUIRE(!save || (agg && pos && lse && q), GATMI_EINVAL, "dense_attn_fwd_masked: agg, pos, lse, q must be given together");
    GATMI_REQUIRE(coef_keep > 0.f && coef_keep <= 1.f && feat_keep > 0.f && feat_keep <= 1.f, GATMI_EINVAL, "dense_attn_fwd_masked: keep probabilities must be in (0,1]");
    const bool drop = coef_keep < 1.f || feat_keep < 1.f;
    GATMI_REQUIRE(!drop || save, GATMI_EINVAL, "dense_attn_fwd_masked: dropout is a training-time option and needs agg, pos, lse, q");
    GATMI_REQUIRE_ALIGNED(Wh); GATMI_REQUIRE_ALIGNED(out); GATMI_REQUIRE_ALIGNED(bias); GATMI_REQUIRE_ALIGNED(resid);
    GATMI_REQUIRE_ALIGNED(bias_mat);
    GATMI_REQUIRE_ALIGNED(agg); GATMI_REQUIRE_ALIGNED(pos);
    MaskedFwdArgs a;
    a.B = B; a.N = N; a.Wh = Wh; a.f1 = f1; a.f2 = f2; a.bias_mat = bias_mat; a.bias = bias; a.resid = resid;
    a.out = out; a.agg = agg; a.pos = pos; a.lse = lse; a.q = q; a.activation = activation; a.slope = slope;
    a.coef_keep = coef_keep; a.feat_keep = feat_keep; a.seed = seed; a.seed_step = (coef_keep < 1.f || feat_keep < 1.f) ? dropout_step_source() : nullptr; a.present = present;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    // many small graphs: one block per graph with the graph's rows resident in LDS
    const int lc = graph_list_capacity(N);
    if (graph8_shape(B, N, G, LPH) && graph8_fwd_lds(N, H, lc, graph8_waves(N)) <= graph_lds_limit()) {
        a.lptr = nullptr; a.lcol = a.lord = nullptr; a.lcap = 0;
        return launch_graph_fwd8(a, LPH, save, drop, graph8_fwd_lds(N, H, lc, graph8_waves(N)), lc, st, "dense_attn_fwd_masked (graph-resident, eight-lane walk)");
    }
    if (!drop && G == 16 && (N & 3) == 0 && N >= 4 && N <= kGraphMaxN && B >= kGraphMinBatch && graph_fwd_lds(N, G, H, lc) <= graph_lds_limit()) {
        const size_t lds = graph_fwd_lds(N, G, H, lc);
        hipError_t e = hipSuccess;
#define X(ll)                                                                                                      \
        if (LPH == ll) {                                                                                          \
            if (save) {                                                                                           \
                if (lds > 48 * 1024) e = hipFuncSetAttribute(reinterpret_cast<const void *>(&dense_graph_fwd_kernel<ll, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
                if (e == hipSuccess) hipLaunchKernelGGL((dense_graph_fwd_kernel<ll, true>), dim3((unsigned)B), dim3(kGraphFwdThreads), lds, st, a, lc); \
            } else {                                                                                              \
                if (lds > 48 * 1024) e = hipFuncSetAttribute(reinterpret_cast<const void *>(&dense_graph_fwd_kernel<ll, false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
                if (e == hipSuccess) hipLaunchKernelGGL((dense_graph_fwd_kernel<ll, false>), dim3((unsigned)B), dim3(kGraphFwdThreads), lds, st, a, lc); \
            }                                                                                                     \
            if (e != hipSuccess) return fail(GATMI_EHIP, "dense_attn_fwd_masked: %s", hipGetErrorString(e));       \
            GATMI_LAUNCH_CHECK("dense_attn_fwd_masked (graph-resident)");                                         \
            return GATMI_OK;                                                                                      \
        }
        GATMI_GRAPH_SHAPES(X)
#undef X
    }
    const int rows_per_block = 4 * (64 / G);
    const dim3 grid((unsigned)B, (unsigned)((N + rows_per_block - 1) / rows_per_block));
#define X(gg, ll)                                                                                                 \
    if (G == gg && LPH == ll) {                                                                                   \
        if (drop) hipLaunchKernelGGL((dense_masked_fwd_kernel<gg, ll, true, true>), grid, dim3(256), 0, st, a);    \
        else if (save) hipLaunchKernelGGL((dense_masked_fwd_kernel<gg, ll, true, false>), grid, dim3(256), 0, st, a); \
        else hipLaunchKernelGGL((dense_masked_fwd_kernel<gg, ll, false, false>), grid, dim3(256), 0, st, a);       \
        GATMI_LAUNCH_CHECK("dense_attn_fwd_masked");                                                              \
        return GATMI_OK;                                                                                          \
    }
    GATMI_MASKED_SHAPES(X)
#undef X
    return fail(GATMI_EUNSUPPORTED, "dense_attn_fwd_masked: no kernel for H=%d dp=%d", H, dp);
}

extern "C" int gatmi_dense_attn_bwd_masked(int64_t B, int32_t N, int32_t H, int32_t dp,
                                           const float *g, const float *stats, const float *Wh, const float *f2,
                                           const float *bias_mat, const float *df1, const float *a1, const float *a2,
                                           const uint32_t *present,
                                           float slope, float coef_keep, float feat_keep, uint64_t seed,
                                           float *dWh, float *df2, void *stream) {
    int G, LPH;
    int rc = masked_check(B, N, H, dp, &G, &LPH, "dense_attn_bwd_masked");
    if (rc) return rc;
    if (B == 0 || N == 0) return GATMI_OK;
    GATMI_REQUIRE(g && stats && Wh && f2 && bias_mat && present && dWh && df2, GATMI_EINVAL, "dense_attn_bwd_masked: null pointer");
    GATMI_REQUIRE((a1 == nullptr) == (a2 == nullptr) && (!a1 || df1), GATMI_EINVAL, "dense_attn_bwd_masked: a1, a2 and df1 go together");
    GATMI_REQUIRE_ALIGNED(g); GATMI_REQUIRE_ALIGNED(Wh); GATMI_REQUIRE_ALIGNED(dWh); GATMI_REQUIRE_ALIGNED(a1); GATMI_REQUIRE_ALIGNED(a2);
    GATMI_REQUIRE(coef_keep > 0.f && coef_keep <= 1.f && feat_keep > 0.f && feat_keep <= 1.f, GATMI_EINVAL, "dense_attn_bwd_masked: keep probabilities must be in (0,1]");
    const bool drop = coef_keep < 1.f || feat_keep < 1.f;
    MaskedBwdArgs a;
    a.B = B; a.N = N; a.g = g; a.stats = stats; a.Wh = Wh; a.f2 = f2; a.bias_mat = bias_mat; a.df1 = df1; a.a1 = a1; a.a2 = a2;
    a.dWh = dWh; a.df2 = df2; a.slope = slope; a.coef_keep = coef_keep; a.feat_keep = feat_keep; a.seed = seed; a.seed_step = (coef_keep < 1.f || feat_keep < 1.f) ? dropout_step_source() : nullptr; a.present = present;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    // many small graphs: one block per graph with the graph's rows resident in LDS
    const int lcg = graph_list_capacity(N);
    if (!drop && G == 16 && LPH >= 2 && (N & 3) == 0 && N >= 4 && N <= kGraphMaxN && B >= kGraphMinBatch &&
        graph8_bwd_lds(N, H, lcg, graph8_waves(N)) <= graph_lds_limit()) {
        const int waves = graph8_waves(N);
        const size_t glds = graph8_bwd_lds(N, H, lcg, waves);
        hipError_t e = hipSuccess;
#define X(ll)                                                                                                      \
        if (LPH == 2 * ll) {                                                                                      \
            if (glds > 48 * 1024) e = hipFuncSetAttribute(reinterpret_cast<const void *>(&dense_graph_bwd8_kernel<ll, false, false, false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)glds); \
            if (e == hipSuccess) hipLaunchKernelGGL((dense_graph_bwd8_kernel<ll, false, false, false>), dim3((unsigned)B), dim3(64 * waves), glds, st, a, lcg); \
            if (e != hipSuccess) return fail(GATMI_EHIP, "dense_attn_bwd_masked: %s", hipGetErrorString(e));       \
            GATMI_LAUNCH_CHECK("dense_attn_bwd_masked (graph-resident, eight-lane walk)");                        \
            return GATMI_OK;                                                                                      \
        }
        X(1) X(2) X(4) X(8)
#undef X
    }
    if (!drop && G == 16 && (N & 3) == 0 && N >= 4 && N <= kGraphMaxN && B >= kGraphMinBatch && graph_bwd_lds(N, H, lcg) <= graph_lds_limit()) {
        const size_t glds = graph_bwd_lds(N, H, lcg);
        hipError_t e = hipSuccess;
#define X(ll)                                                                                                      \
        if (LPH == ll) {                                                                                          \
            if (glds > 48 * 1024) e = hipFuncSetAttribute(reinterpret_cast<const void *>(&dense_graph_bwd_kernel<ll>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)glds); \
            if (e == hipSuccess) hipLaunchKernelGGL((dense_graph_bwd_kernel<ll>), dim3((unsigned)B), dim3(kGraphFwdThreads), glds, st, a, lcg); \
            if (e != hipSuccess) return fail(GATMI_EHIP, "dense_attn_bwd_masked: %s", hipGetErrorString(e));       \
            GATMI_LAUNCH_CHECK("dense_attn_bwd_masked (graph-resident)");                                         \
            return GATMI_OK;                                                                                      \
        }
        GATMI_GRAPH_SHAPES(X)
#undef X
    }
    const dim3 grid((unsigned)B, (unsigned)((N + kBwdCols - 1) / kBwdCols));
    const int n_chunks = std::min(kBwdChunks, (N + 63) / 64);
    const size_t lds = (size_t)kBwdCols * (n_chunks * 64 * sizeof(uint16_t) + (kBwdChunks + 1) * sizeof(int));
#define X(gg, ll)                                                                                          \
    if (G == gg && LPH == ll) {                                                                            \
        if (drop) hipLaunchKernelGGL((dense_masked_bwd_kernel<gg, ll, true>), grid, dim3(256), lds, st, a, n_chunks);   \
        else hipLaunchKernelGGL((dense_masked_bwd_kernel<gg, ll, false>), grid, dim3(256), lds, st, a, n_chunks);       \
        GATMI_LAUNCH_CHECK("dense_attn_bwd_masked");                                                       \
        return GATMI_OK;                                                                                   \
    }
    GATMI_MASKED_SHAPES(X)
#undef X
    return fail(GATMI_EUNSUPPORTED, "dense_attn_bwd_masked: no kernel for H=%d dp=%d", H, dp);
}

// ------------------------------------------------------------------------------------------
// backward for batches of many small graphs with the per-destination step fused in
// ------------------------------------------------------------------------------------------
static bool graph_fused_shape(int64_t B, int32_t N, int G, int LPH, int H) {
    return graph8_shape(B, N, G, LPH) && graph8_bwd_lds(N, H, graph_list_capacity(N), graph8_waves(N)) <= graph_lds_limit();
}

extern "C" int gatmi_dense_attn_bwd_fused_ok(int64_t B, int32_t N, int32_t H, int32_t dp, float coef_keep, float feat_keep) {
    if (H <= 0 || dp < 4 || !is_pow2(dp) || !is_pow2(H) || (int64_t)H * dp > 256) return 0;
    if (!(coef_keep > 0.f && coef_keep <= 1.f && feat_keep > 0.f && feat_keep <= 1.f)) return 0;
    return graph_fused_shape(B, N, H * dp / 4, dp / 4, H) ? 1 : 0;
}

extern "C" size_t gatmi_dense_attn_bwd_fused_ws_floats(int64_t B, int32_t H, int32_t dp) {
    // one partial row per graph, then one per kDbiasStageRows graphs; a row holds dbias and, when the kernel also
    // sums the attention parameters' gradients, da1, da2, db1, db2
    return B <= 0 ? 0 : (size_t)(B + (B + kDbiasStageRows - 1) / kDbiasStageRows) * (size_t)(3 * H * dp + 2 * H);
}

extern "C" int gatmi_dense_attn_bwd_fused_param_grads(int32_t N, int32_t H, int32_t dp, int32_t lists) {
    // the half-row kernel over prebuilt lists: rows of 64 floats in at least two heads
    return lists && H >= 2 && gatmi_dense_lists_capacity(N, H, dp, 1) > 0 ? 1 : 0;
}

extern "C" int64_t gatmi_dense_lists_capacity(int32_t N, int32_t H, int32_t dp, int32_t backward) {
    if (H <= 0 || dp < 8 || !is_pow2(dp) || !is_pow2(H) || H * dp != 64 || N < 4 || N > kGraphMaxN) return 0;
    const size_t limit = graph_lds_limit();
    if (!backward)                                       // the forward kernel reads its lists from memory: any length, if the rows fit
        return graph8_fwd_walk_lds(N, H) <= limit ? (int64_t)(N + kGraphBatch) * N : 0;
    if (H >= 2)                                          // half-row blocks read their lists from memory as well
        return graph8_bwd_half_lds(N, H, kHalfWaves) <= limit ? (int64_t)(N + kGraphBatch) * N : 0;
    if (N & 3) return 0;                                 // one head of 64 floats: the whole-row kernel stages the lists (8-byte aligned areas)
    const size_t fixed = graph8_bwd_lists_lds(N, H, 0, 0);
    if (fixed + 64 >= limit) return 0;
    return (int64_t)((limit - fixed) / sizeof(uint16_t)) & ~(int64_t)3;
}

extern "C" int gatmi_dense_mask_bits(int64_t B, int32_t N, const float *bias_mat, float shared_limit, uint32_t *bits, void *stream) {
    GATMI_REQUIRE(B >= 0 && N >= 0, GATMI_EINVAL, "dense_mask_bits: negative size");
    if (B == 0 || N == 0) return GATMI_OK;
    GATMI_REQUIRE(bias_mat && bits, GATMI_EINVAL, "dense_mask_bits: null pointer");
    GATMI_REQUIRE(B <= 0x7fffffff && (N + 3) / 4 <= 65535, GATMI_ERANGE, "dense_mask_bits: too large for one launch");
    const int W = (N + 31) / 32 + 1;
    hipLaunchKernelGGL(mask_bits_kernel, dim3((unsigned)B, (unsigned)((N + 3) / 4)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), bias_mat, N, W, shared_limit, bits);
    GATMI_LAUNCH_CHECK("dense_mask_bits");
    return GATMI_OK;
}

static int lists_check(int64_t B, int32_t N, const char *name) {
    GATMI_REQUIRE(B >= 0 && N >= 0, GATMI_EINVAL, "%s: negative size", name);
    GATMI_REQUIRE(N >= 4 && N <= kGraphMaxN, GATMI_EUNSUPPORTED,
                  "%s: prebuilt lists are for the graph-resident kernels (graphs of 4 .. %d nodes), N = %d", name, kGraphMaxN, N);
    GATMI_REQUIRE(B <= 0x7fffffff && B * (int64_t)N < 0x7fffffff, GATMI_ERANGE, "%s: too large for one launch", name);
    return GATMI_OK;
}

extern "C" int gatmi_dense_lists_count(int64_t B, int32_t N, const uint32_t *bits, int32_t *row_cnt, int32_t *col_cnt, void *stream) {
    if (B == 0) return GATMI_OK;
    int rc = lists_check(B, N, "dense_lists_count");
    if (rc) return rc;
    GATMI_REQUIRE(bits && row_cnt && col_cnt, GATMI_EINVAL, "dense_lists_count: null pointer");
    const int W = (N + 31) / 32 + 1;
    hipLaunchKernelGGL(lists_count_kernel, dim3((unsigned)B), dim3(kListsThreads), (size_t)N * W * sizeof(uint32_t), reinterpret_cast<hipStream_t>(stream),
                       bits, N, W, row_cnt, col_cnt);
    GATMI_LAUNCH_CHECK("dense_lists_count");
    return GATMI_OK;
}

extern "C" int gatmi_dense_lists_fill(int64_t B, int32_t N, const uint32_t *bits, const int32_t *lptr, const int32_t *cptr,
                                      uint16_t *lcol, uint16_t *crow, void *stream) {
    if (B == 0) return GATMI_OK;
    int rc = lists_check(B, N, "dense_lists_fill");
    if (rc) return rc;
    GATMI_REQUIRE(bits && lptr && cptr && lcol && crow, GATMI_EINVAL, "dense_lists_fill: null pointer");
    const int W = (N + 31) / 32 + 1;
    hipLaunchKernelGGL(lists_fill_kernel, dim3((unsigned)B), dim3(kListsThreads), (size_t)N * W * sizeof(uint32_t), reinterpret_cast<hipStream_t>(stream),
                       bits, N, W, lptr, cptr, lcol, crow);
    GATMI_LAUNCH_CHECK("dense_lists_fill");
    return GATMI_OK;
}

extern "C" int gatmi_dense_attn_fwd_lists(int64_t B, int32_t N, int32_t H, int32_t dp,
                                          const float *Wh, const float *f1, const float *f2,
                                          const int32_t *lptr, const uint16_t *lcol, const uint16_t *lord, int64_t max_entries,
                                          const float *bias, const float *resid, int32_t activation, float slope,
                                          float coef_keep, float feat_keep, uint64_t seed,
                                          float *out, float *agg, float *pos, float *lse, float *q, void *stream) {
    int G, LPH;
    int rc = masked_check(B, N, H, dp, &G, &LPH, "dense_attn_fwd_lists");
    if (rc) return rc;
    if (B == 0 || N == 0) return GATMI_OK;
    GATMI_REQUIRE(Wh && f1 && f2 && lptr && lcol && lord && out, GATMI_EINVAL, "dense_attn_fwd_lists: null pointer");
    GATMI_REQUIRE(graph8_shape(B, N, G, LPH, true), GATMI_EUNSUPPORTED,
                  "dense_attn_fwd_lists: B=%lld N=%d H=%d dp=%d is not a shape of the graph-resident kernels (rows of 64 floats in heads of >= 8, "
                  "graphs of 4 .. %d nodes, at least %lld graphs)", (long long)B, N, H, dp, kGraphMaxN, (long long)kGraphMinBatch);
    const int64_t cap_max = gatmi_dense_lists_capacity(N, H, dp, 0);
    GATMI_REQUIRE(max_entries >= 0 && max_entries <= cap_max, GATMI_ERANGE,
                  "dense_attn_fwd_lists: a graph with %lld list entries does not fit beside its rows in LDS (at most %lld here); use gatmi_dense_attn_fwd_masked",
                  (long long)max_entries, (long long)cap_max);
    GATMI_REQUIRE(act_known(activation), GATMI_EUNSUPPORTED, "dense_attn_fwd_lists: activation %d", activation);
    const bool save = agg || pos || lse || q;
    GATMI_REQUIRE(!save || (agg && pos && lse && q), GATMI_EINVAL, "dense_attn_fwd_lists: agg, pos, lse, q must be given together");
    GATMI_REQUIRE(coef_keep > 0.f && coef_keep <= 1.f && feat_keep > 0.f && feat_keep <= 1.f, GATMI_EINVAL, "dense_attn_fwd_lists: keep probabilities must be in (0,1]");
    const bool drop = coef_keep < 1.f || feat_keep < 1.f;
    GATMI_REQUIRE(!drop || save, GATMI_EINVAL, "dense_attn_fwd_lists: dropout is a training-time option and needs agg, pos, lse, q");
    GATMI_REQUIRE_ALIGNED(Wh); GATMI_REQUIRE_ALIGNED(out); GATMI_REQUIRE_ALIGNED(bias); GATMI_REQUIRE_ALIGNED(resid);
    GATMI_REQUIRE_ALIGNED(agg); GATMI_REQUIRE_ALIGNED(pos);
    GATMI_REQUIRE((reinterpret_cast<uintptr_t>(lcol) & 7u) == 0, GATMI_EALIGN, "dense_attn_fwd_lists: lcol is not 8-byte aligned");
    MaskedFwdArgs a;
    a.B = B; a.N = N; a.Wh = Wh; a.f1 = f1; a.f2 = f2; a.bias_mat = nullptr; a.bias = bias; a.resid = resid;
    a.out = out; a.agg = agg; a.pos = pos; a.lse = lse; a.q = q; a.activation = activation; a.slope = slope;
    a.coef_keep = coef_keep; a.feat_keep = feat_keep; a.seed = seed; a.seed_step = drop ? dropout_step_source() : nullptr; a.present = nullptr;
    const int cap = (int)((max_entries + 3) & ~(int64_t)3);
    a.lptr = lptr; a.lcol = lcol; a.lord = lord; a.lcap = cap;
    return launch_graph_fwd8_lists(a, LPH, save, drop, graph8_fwd_walk_lds(N, H), reinterpret_cast<hipStream_t>(stream), "dense_attn_fwd_lists");
}

extern "C" int gatmi_dense_attn_bwd_fused(int64_t B, int32_t N, int32_t H, int32_t dp,
                                          const float *dout, const float *out, const float *agg, const float *pos,
                                          const float *q, const float *f1, const float *lse, const float *bias, int32_t activation,
                                          const float *Wh, const float *f2, const float *bias_mat, const uint32_t *present,
                                          const int32_t *cptr, const uint16_t *crow, const uint16_t *cord, int64_t max_entries,
                                          const float *a1, const float *a2, float slope,
                                          float coef_keep, float feat_keep, uint64_t seed,
                                          float *g, float *df1, float *dbias,
                                          float *da1, float *da2, float *db1, float *db2, float *ws,
                                          float *dWh, float *df2, void *stream) {
    int G, LPH;
    int rc = masked_check(B, N, H, dp, &G, &LPH, "dense_attn_bwd_fused");
    if (rc) return rc;
    if (B == 0 || N == 0) return GATMI_OK;
    const bool lists = cptr != nullptr;
    // the mask route keeps the graph's `present` bits and the waves' lists beside its rows; prebuilt lists need less
    // (gatmi_dense_lists_capacity says whether a graph's rows fit at all, and how many entries beside them)
    GATMI_REQUIRE(lists ? (graph8_shape(B, N, G, LPH, true) && gatmi_dense_lists_capacity(N, H, dp, 1) > 0) : graph_fused_shape(B, N, G, LPH, H), GATMI_EUNSUPPORTED,
                  "dense_attn_bwd_fused: B=%lld N=%d H=%d dp=%d is not a shape of the graph-resident kernels (ask gatmi_dense_attn_bwd_fused_ok%s)",
                  (long long)B, N, H, dp, lists ? " / gatmi_dense_lists_capacity" : "");
    GATMI_REQUIRE(dout && agg && pos && q && f1 && lse && Wh && f2 && df1 && dWh && df2, GATMI_EINVAL, "dense_attn_bwd_fused: null pointer");
    GATMI_REQUIRE(lists ? (crow && cord) : (bias_mat && present), GATMI_EINVAL,
                  "dense_attn_bwd_fused: give either the mask with the forward pass's `present` bits, or prebuilt lists (cptr, crow, cord)");
    GATMI_REQUIRE((a1 == nullptr) == (a2 == nullptr), GATMI_EINVAL, "dense_attn_bwd_fused: a1 and a2 go together");
    const bool params = da1 || da2 || db1 || db2;
    GATMI_REQUIRE(!params || (da1 && da2 && db1 && db2 && a1), GATMI_EINVAL, "dense_attn_bwd_fused: da1, da2, db1, db2 go together and need a1, a2");
    GATMI_REQUIRE(!params || gatmi_dense_attn_bwd_fused_param_grads(N, H, dp, lists ? 1 : 0), GATMI_EUNSUPPORTED,
                  "dense_attn_bwd_fused: this route does not sum da1, da2, db1, db2 (ask gatmi_dense_attn_bwd_fused_param_grads); pass NULL and use gatmi_proj_bwd_params");
    GATMI_REQUIRE(!(dbias || params) || ws, GATMI_EINVAL, "dense_attn_bwd_fused: dbias, da1.. need ws (gatmi_dense_attn_bwd_fused_ws_floats)");
    GATMI_REQUIRE(act_known(activation), GATMI_EUNSUPPORTED, "dense_attn_bwd_fused: activation %d", activation);
    GATMI_REQUIRE(coef_keep > 0.f && coef_keep <= 1.f && feat_keep > 0.f && feat_keep <= 1.f, GATMI_EINVAL, "dense_attn_bwd_fused: keep probabilities must be in (0,1]");
    const bool drop = coef_keep < 1.f || feat_keep < 1.f;
    GATMI_REQUIRE_ALIGNED(dout); GATMI_REQUIRE_ALIGNED(out); GATMI_REQUIRE_ALIGNED(agg); GATMI_REQUIRE_ALIGNED(pos); GATMI_REQUIRE_ALIGNED(bias);
    GATMI_REQUIRE_ALIGNED(Wh); GATMI_REQUIRE_ALIGNED(g); GATMI_REQUIRE_ALIGNED(dWh); GATMI_REQUIRE_ALIGNED(a1); GATMI_REQUIRE_ALIGNED(a2); GATMI_REQUIRE_ALIGNED(ws);
    MaskedBwdArgs a{};
    a.B = B; a.N = N; a.Wh = Wh; a.f2 = f2; a.bias_mat = bias_mat; a.a1 = a1; a.a2 = a2; a.present = present;
    a.dWh = dWh; a.df2 = df2; a.slope = slope; a.coef_keep = coef_keep; a.feat_keep = feat_keep; a.seed = seed;
    a.seed_step = drop ? dropout_step_source() : nullptr;
    a.dout = dout; a.out = out; a.agg = agg; a.pos = pos; a.q = q; a.f1 = f1; a.lse = lse; a.bias = out ? nullptr : bias;
    a.g_out = g; a.df1_out = df1; a.dbias_ws = (dbias || params) ? ws : nullptr; a.activation = activation;
    const int hd = H * dp;
    a.param_grads = params ? 1 : 0;
    a.ws_stride = params ? 3 * hd + 2 * H : hd;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const int waves = graph8_waves(N);
    if (lists && LPH <= 8) {
        // heads of at most 32 floats: one block per half row, three blocks per CU, lists read from memory (any length)
        GATMI_REQUIRE(max_entries >= 0, GATMI_ERANGE, "dense_attn_bwd_fused: max_entries = %lld", (long long)max_entries);
        GATMI_REQUIRE((reinterpret_cast<uintptr_t>(crow) & 7u) == 0, GATMI_EALIGN, "dense_attn_bwd_fused: crow is not 8-byte aligned");
        a.lptr = cptr; a.lcol = crow; a.lord = cord; a.lcap = 0;
        rc = launch_graph_bwd8_half(a, LPH, drop, st, "dense_attn_bwd_fused (prebuilt lists, half rows)");
    } else if (lists) {
        const int64_t cap_max = gatmi_dense_lists_capacity(N, H, dp, 1);
        GATMI_REQUIRE(max_entries >= 0 && max_entries <= cap_max, GATMI_ERANGE,
                      "dense_attn_bwd_fused: a graph with %lld list entries does not fit beside its rows in LDS (at most %lld here); pass the mask and `present` instead",
                      (long long)max_entries, (long long)cap_max);
        GATMI_REQUIRE((reinterpret_cast<uintptr_t>(crow) & 7u) == 0, GATMI_EALIGN, "dense_attn_bwd_fused: crow is not 8-byte aligned");
        const int cap = (int)((max_entries + 3) & ~(int64_t)3);
        a.lptr = cptr; a.lcol = crow; a.lord = cord; a.lcap = cap;
        rc = launch_graph_bwd8_fused<true>(a, LPH, drop, graph8_bwd_lists_lds(N, H, cap, waves), 0, st, "dense_attn_bwd_fused (prebuilt lists)");
    } else {
        const int lc = graph_list_capacity(N);
        rc = launch_graph_bwd8_fused<false>(a, LPH, drop, graph8_bwd_lds(N, H, lc, waves), lc, st, "dense_attn_bwd_fused");
    }
    if (rc) return rc;
    if (dbias || params) {
        const int width = a.ws_stride;
        const float *rows = ws;
        int n_rows = (int)B;
        if (B > 4 * kDbiasStageRows) {
            n_rows = (int)((B + kDbiasStageRows - 1) / kDbiasStageRows);
            float *stage = ws + (size_t)B * width;
            hipLaunchKernelGGL(graph_dbias_stage_kernel, dim3((width + kSumCols - 1) / kSumCols, n_rows), dim3(kSumCols, kSumRows), 0, st, ws, (int)B, width, stage);
            GATMI_LAUNCH_CHECK("dense_attn_bwd_fused dbias stage");
            rows = stage;
        }
        if (params)
            hipLaunchKernelGGL(graph_param_sum_kernel, dim3((width + kSumCols - 1) / kSumCols), dim3(kSumCols, kSumRows), 0, st, rows, n_rows, width, H, dbias, da1, da2, db1, db2);
        else
            hipLaunchKernelGGL(graph_dbias_sum_kernel, dim3((width + kSumCols - 1) / kSumCols), dim3(kSumCols, kSumRows), 0, st, rows, n_rows, width, dbias);
        GATMI_LAUNCH_CHECK("dense_attn_bwd_fused dbias");
    }
    return GATMI_OK;
}
