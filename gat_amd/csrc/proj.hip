// Feature projection kernels for gfx950: Wh = X.W with the attention-logit halves fused into the
// epilogue, and the gradients dW = X^T.dWh, dX = dWh.W^T, da1/da2/db1/db2.
//
// All three contractions are "tall and skinny": one dimension is the node count (millions), the
// other two are small (f_in ~ 10^2..10^3, hd <= 256).  They are HBM-bound streams of X / Wh / dWh
// rows with a small operand kept on chip, and run on the exact-fp32 matrix instruction
// v_mfma_f32_16x16x4_f32 (lane l supplies A[l&15][l>>4] and B[l>>4][l&15]; the 16x16 result has
// column l&15 and rows 4*(l>>4)+r in register r).
#include <hip/hip_runtime.h>

#include <algorithm>

#include "common.h"
#include "device_common.h"

namespace gatmi {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int kTileRows = 128;  // rows of C per block (4 waves x 2 M-tiles of 16)
constexpr int kTileCols = 64;   // columns of C per block (4 N-tiles of 16)
constexpr int kChunkK = 64;     // K extent of the B slab staged in LDS
constexpr int kLdB = 68;        // LDS row stride of B and C tiles: 4*68 mod 32 = 16 => the two
                                // k-rows read by one 32-lane half land on disjoint banks

struct GemmArgs {
    const float *A;   // [n, K], row stride lda
    int64_t n;
    int K;
    int64_t lda;
    const float *B;   // element (k, c) at B[k*sbk + c*sbc]
    int64_t sbk, sbc;
    int Nc;
    float *C;         // [n, Nc], row stride ldc
    int64_t ldc;
    uint32_t *C16;    // optional second copy of C in bf16 (Nc % 4 == 0), two values per dword, rows ldc16 dwords apart;
    int64_t ldc16;    //   with the fused logit halves, f2 of the row follows its Nc bf16 values (fp32, dword Nc/2 + head)
    int accumulate;
    const float *cbias;  // [Nc] added to C, or null
    // fused logit halves (only when the whole row is inside one column tile)
    const float *a1, *a2, *b1, *b2;
    float *f1, *f2;
    int H, LPH;
    // dropout of the layer's input (utils/layers.py:9; device_common.h: drop_in4), 0 = off: a_keep masks A as it is
    // loaded (A = X), c_keep masks C as it is stored (C = dX); rows of the masked matrix hold drop_f4 groups of four
    float a_keep, c_keep;
    int drop_f4;
    uint64_t seed;
    const uint64_t *seed_step;
};

// C[n, Nc] = A[n, K] . B[K, Nc]; B small.  VEC: A rows are 16-byte aligned and K % 4 == 0.
// SPLITK = false: a block owns 128 rows, each wave 32 of them, and walks K in slabs of 64.
// SPLITK = true (few rows, long K: Cora's 2708 x 1433 features): a block owns 32 rows, its four
// waves take a quarter of every 128-deep slab each and their partial tiles are added in wave
// order through LDS, so four times as many blocks run and each walks a quarter of K.
template <bool VEC, bool SPLITK>
__global__ __launch_bounds__(256) void tall_gemm_kernel(GemmArgs g) {
    constexpr int ROWS = SPLITK ? 32 : kTileRows;      // rows of C per block
    constexpr int SLAB = SPLITK ? 128 : kChunkK;       // K extent staged per step
    constexpr int S = SPLITK ? 2 : 4;                  // 16-deep k-steps a wave takes from a slab
    __shared__ __attribute__((aligned(16))) float lds[kTileRows * kLdB];  // B slab, later the C tile
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int m = lane & 15, kq = lane >> 4;
    const int64_t row0 = (int64_t)blockIdx.x * ROWS;
    const int col0 = blockIdx.y * kTileCols;
    const int wrow = SPLITK ? 0 : wave * 32;           // first row of this wave inside the block
    const int wk = SPLITK ? wave * 32 : 0;             // first slab row of this wave
    const uint64_t dseed = (g.a_keep != 0.f || g.c_keep != 0.f) ? seed_now(g) : 0;

    f32x4 acc[2][4];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) acc[mt][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};

    int64_t arow[2];
    bool rok[2];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
        arow[mt] = row0 + wrow + mt * 16 + m;
        rok[mt] = arow[mt] < g.n;
    }

    if constexpr (SPLITK) {
        // No staging and no barrier: B is small and sits in L2, every wave fetches its own fragments
        // (lane (m, kq) holds B[k = .. + 4*kq + t][col0 + 16*nt + m]); the 32 B loads and 4 A loads of
        // a step are independent, so one round trip covers them all.
        for (int kc = wk; kc < g.K; kc += SLAB) {
            float a[S][2][4], b[S][4][4];
#pragma unroll
            for (int s = 0; s < S; ++s) {
                const int k = kc + 16 * s + 4 * kq;
#pragma unroll
                for (int mt = 0; mt < 2; ++mt) {
                    if (VEC) {
                        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                        if (rok[mt] && k < g.K) v = *reinterpret_cast<const float4 *>(g.A + arow[mt] * g.lda + k);
                        if (g.a_keep != 0.f) v = drop_in4(v, dseed, arow[mt], g.drop_f4, k >> 2, g.a_keep);
                        a[s][mt][0] = v.x; a[s][mt][1] = v.y; a[s][mt][2] = v.z; a[s][mt][3] = v.w;
                    } else {
#pragma unroll
                        for (int t = 0; t < 4; ++t) a[s][mt][t] = (rok[mt] && k + t < g.K) ? g.A[arow[mt] * g.lda + k + t] : 0.f;
                    }
                }
#pragma unroll
                for (int t = 0; t < 4; ++t)
#pragma unroll
                    for (int nt = 0; nt < 4; ++nt) {
                        const int c = col0 + 16 * nt + m;
                        b[s][t][nt] = (k + t < g.K && c < g.Nc) ? g.B[(int64_t)(k + t) * g.sbk + (int64_t)c * g.sbc] : 0.f;
                    }
            }
#pragma unroll
            for (int s = 0; s < S; ++s)
#pragma unroll
                for (int t = 0; t < 4; ++t)
#pragma unroll
                    for (int nt = 0; nt < 4; ++nt)
#pragma unroll
                        for (int mt = 0; mt < 2; ++mt)
                            acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[s][mt][t], b[s][t][nt], acc[mt][nt], 0, 0, 0);
        }
    } else {
        for (int kc = 0; kc < g.K; kc += SLAB) {
            __syncthreads();
            // stage B[kc .. kc+SLAB) x [col0 .. col0+64), zero-filled outside
            if (g.sbc == 1) {
                for (int e = threadIdx.x; e < SLAB * kTileCols; e += 256) {
                    const int kk = e >> 6, c = e & 63;
                    const bool ok = (kc + kk < g.K) && (col0 + c < g.Nc);
                    lds[kk * kLdB + c] = ok ? g.B[(int64_t)(kc + kk) * g.sbk + (col0 + c)] : 0.f;
                }
            } else {
                for (int e = threadIdx.x; e < SLAB * kTileCols; e += 256) {
                    const int c = e / SLAB, kk = e % SLAB;
                    const bool ok = (kc + kk < g.K) && (col0 + c < g.Nc);
                    lds[kk * kLdB + c] = ok ? g.B[(int64_t)(kc + kk) * g.sbk + (int64_t)(col0 + c) * g.sbc] : 0.f;
                }
            }
            __syncthreads();
            // A fragments of this slab: lane (m, kq) takes k = kc + wk + 16*s + 4*kq + t, t = 0..3
            float a[S][2][4];
    #pragma unroll
            for (int s = 0; s < S; ++s) {
    #pragma unroll
                for (int mt = 0; mt < 2; ++mt) {
                    const int k = kc + wk + 16 * s + 4 * kq;
                    if (VEC) {
                        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                        if (rok[mt] && k < g.K) v = *reinterpret_cast<const float4 *>(g.A + arow[mt] * g.lda + k);
                        if (g.a_keep != 0.f) v = drop_in4(v, dseed, arow[mt], g.drop_f4, k >> 2, g.a_keep);
                        a[s][mt][0] = v.x; a[s][mt][1] = v.y; a[s][mt][2] = v.z; a[s][mt][3] = v.w;
                    } else {
    #pragma unroll
                        for (int t = 0; t < 4; ++t) a[s][mt][t] = (rok[mt] && k + t < g.K) ? g.A[arow[mt] * g.lda + k + t] : 0.f;
                    }
                }
            }
    #pragma unroll
            for (int s = 0; s < S; ++s) {
    #pragma unroll
                for (int t = 0; t < 4; ++t) {
                    const float *brow = lds + (wk + 16 * s + 4 * kq + t) * kLdB + m;
    #pragma unroll
                    for (int nt = 0; nt < 4; ++nt) {
                        const float b = brow[16 * nt];
    #pragma unroll
                        for (int mt = 0; mt < 2; ++mt)
                            acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[s][mt][t], b, acc[mt][nt], 0, 0, 0);
                    }
                }
            }
        }
    }

    // C tile through LDS so that global stores are whole-row float4 runs
    __syncthreads();
    if constexpr (SPLITK) {
        for (int w = 0; w < 4; ++w) {                  // the four K-quarters, added in wave order
            if (wave == w) {
#pragma unroll
                for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                    for (int nt = 0; nt < 4; ++nt)
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            float *p = lds + (mt * 16 + 4 * kq + r) * kLdB + 16 * nt + m;
                            *p = (w == 0) ? acc[mt][nt][r] : *p + acc[mt][nt][r];
                        }
            }
            __syncthreads();
        }
    } else {
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int nt = 0; nt < 4; ++nt)
#pragma unroll
                for (int r = 0; r < 4; ++r) lds[(wave * 32 + mt * 16 + 4 * kq + r) * kLdB + 16 * nt + m] = acc[mt][nt][r];
        __syncthreads();
    }

    const int c4 = threadIdx.x & 15;
    const int col = col0 + 4 * c4;
    const bool vec_store = ((g.ldc & 3) == 0) && ((reinterpret_cast<uintptr_t>(g.C) & 15) == 0);
    float4 bv = make_float4(0.f, 0.f, 0.f, 0.f);
    if (g.cbias) {
        if (col + 0 < g.Nc) bv.x = g.cbias[col + 0];
        if (col + 1 < g.Nc) bv.y = g.cbias[col + 1];
        if (col + 2 < g.Nc) bv.z = g.cbias[col + 2];
        if (col + 3 < g.Nc) bv.w = g.cbias[col + 3];
    }
    float4 a1v = make_float4(0.f, 0.f, 0.f, 0.f), a2v = a1v;
    int head = 0;
    if (g.f1) {  // whole row inside this tile: Nc <= 64, columns padded to a multiple of 4
        if (col < g.Nc) {
            a1v = *reinterpret_cast<const float4 *>(g.a1 + col);
            a2v = *reinterpret_cast<const float4 *>(g.a2 + col);
        }
        head = c4 / g.LPH;
    }
#pragma unroll
    for (int i = 0; i < ROWS / 16; ++i) {
        const int rl = (threadIdx.x >> 4) + 16 * i;
        const int64_t row = row0 + rl;
        float4 v = *reinterpret_cast<const float4 *>(lds + rl * kLdB + 4 * c4);
        if (g.f1) {
            float p1 = v.x * a1v.x + v.y * a1v.y + v.z * a1v.z + v.w * a1v.w;
            float p2 = v.x * a2v.x + v.y * a2v.y + v.z * a2v.z + v.w * a2v.w;
            for (int o = 1; o < g.LPH; o <<= 1) {
                p1 += __shfl_xor(p1, o, 64);
                p2 += __shfl_xor(p2, o, 64);
            }
            if (row < g.n && col < g.Nc && (c4 % g.LPH) == 0) {
                g.f1[row * g.H + head] = p1 + g.b1[head];
                g.f2[row * g.H + head] = p2 + g.b2[head];
                if (g.C16) reinterpret_cast<float *>(g.C16)[row * g.ldc16 + g.Nc / 2 + head] = p2 + g.b2[head];
            }
        }
        if (row >= g.n || col >= g.Nc) continue;
        v.x += bv.x; v.y += bv.y; v.z += bv.z; v.w += bv.w;
        float *cp = g.C + row * g.ldc + col;
        if (vec_store && col + 3 < g.Nc) {
            if (g.c_keep != 0.f) v = drop_in4(v, dseed, row, g.drop_f4, col >> 2, g.c_keep);
            if (g.accumulate) {
                const float4 o = *reinterpret_cast<const float4 *>(cp);
                v.x += o.x; v.y += o.y; v.z += o.z; v.w += o.w;
            }
            *reinterpret_cast<float4 *>(cp) = v;
            if (g.C16) *reinterpret_cast<uint2 *>(g.C16 + row * g.ldc16 + col / 2) = make_uint2(pack_bf16x2(v.x, v.y), pack_bf16x2(v.z, v.w));
        } else {
            const float vv[4] = {v.x, v.y, v.z, v.w};
            for (int t = 0; t < 4 && col + t < g.Nc; ++t) cp[t] = g.accumulate ? cp[t] + vv[t] : vv[t];
        }
    }
}


// ------------------------------------------------------------------------------------------
// Resident-weight forward projection: for K <= 128 and Nc <= 64 the whole B = W stays in LDS for
// the life of the block and every wave walks 16-row tiles on its own (no block barrier in the
// loop): A fragments of the NEXT tile are loaded to registers before the current tile's MFMAs,
// the C tile goes through a wave-private LDS patch so that rows leave as whole 256-byte runs, and
// the logit halves f1/f2 are taken from that patch.  NS = ceil(K/16) k-steps of 16, NT column tiles.
// ------------------------------------------------------------------------------------------
constexpr int kResWaves = 8;

template <int NS, int NT, bool DROPIN>
__global__ __launch_bounds__(64 * kResWaves) void proj_resident_kernel(GemmArgs g, int64_t n_tiles) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float *Ws = smem;                                   // [NS*16][kLdB]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float *Cs = smem + NS * 16 * kLdB + wave * 16 * kLdB;   // [16][kLdB], private to this wave
    const int m = lane & 15, kq = lane >> 4;

    for (int e = threadIdx.x; e < NS * 16 * 64; e += 64 * kResWaves) {
        const int k = e >> 6, c = e & 63;
        Ws[k * kLdB + c] = (k < g.K && c < g.Nc) ? g.B[(int64_t)k * g.sbk + c] : 0.f;
    }
    __syncthreads();

    const int c4 = lane & 15;
    const int col = 4 * c4;
    float4 a1v = make_float4(0.f, 0.f, 0.f, 0.f), a2v = a1v, bv = a1v;
    int head = 0;
    float b1h = 0.f, b2h = 0.f;          // read once: a load inside the tile loop would make it wait for the prefetched tile
    if (col < g.Nc) {
        if (g.f1) {
            a1v = *reinterpret_cast<const float4 *>(g.a1 + col);
            a2v = *reinterpret_cast<const float4 *>(g.a2 + col);
            head = c4 / g.LPH;
            b1h = g.b1[head];
            b2h = g.b2[head];
        }
        if (g.cbias) bv = *reinterpret_cast<const float4 *>(g.cbias + col);
    }
    const int64_t tw = (int64_t)gridDim.x * kResWaves;
    int64_t tile = (int64_t)blockIdx.x * kResWaves + wave;
    const uint64_t dseed = DROPIN ? seed_now(g) : 0;

    // The loads of the next tile are unconditional (rows past the end read the last row, k past K reads k = 0, and
    // the values are replaced by zeros when used): a load under a branch makes the compiler wait for ALL outstanding
    // loads before the matrix instructions, the prefetched ones included.
    float4 cur[NS], nxt[NS];
    auto load_tile = [&](int64_t t, float4 (&dst)[NS]) {
        const int64_t row = t * 16 + m < g.n ? t * 16 + m : g.n - 1;
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            const int k = 16 * s + 4 * kq;
            dst[s] = *reinterpret_cast<const float4 *>(g.A + row * g.lda + (k < g.K ? k : 0));
        }
    };
    load_tile(tile, cur);
    for (; tile < n_tiles; tile += tw) {
        load_tile(tile + tw, nxt);
        __builtin_amdgcn_sched_barrier(0);          // the requests go out before the matrix instructions, not after
        const bool rok = tile * 16 + m < g.n;
        f32x4 acc[NT];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) acc[nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            const bool ok = rok && 16 * s + 4 * kq < g.K;
            float4 cs = cur[s];
            if constexpr (DROPIN) cs = drop_in4(cs, dseed, tile * 16 + m, g.drop_f4, 4 * s + kq, g.a_keep);      // the tile is masked where it is used
            const float av[4] = {ok ? cs.x : 0.f, ok ? cs.y : 0.f, ok ? cs.z : 0.f, ok ? cs.w : 0.f};
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const float *brow = Ws + (16 * s + 4 * kq + t) * kLdB + m;
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) acc[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[t], brow[16 * nt], acc[nt], 0, 0, 0);
            }
        }
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int r = 0; r < 4; ++r) Cs[(4 * kq + r) * kLdB + 16 * nt + m] = acc[nt][r];
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int rl = (lane >> 4) + 4 * i;
            const int64_t row = tile * 16 + rl;
            float4 v = *reinterpret_cast<const float4 *>(Cs + rl * kLdB + col);
            if (g.f1) {
                float p1 = v.x * a1v.x + v.y * a1v.y + v.z * a1v.z + v.w * a1v.w;
                float p2 = v.x * a2v.x + v.y * a2v.y + v.z * a2v.z + v.w * a2v.w;
                for (int o = 1; o < g.LPH; o <<= 1) {
                    p1 += __shfl_xor(p1, o, 64);
                    p2 += __shfl_xor(p2, o, 64);
                }
                if (row < g.n && col < g.Nc && (c4 % g.LPH) == 0) {
                    g.f1[row * g.H + head] = p1 + b1h;
                    g.f2[row * g.H + head] = p2 + b2h;
                    if (g.C16) reinterpret_cast<float *>(g.C16)[row * g.ldc16 + g.Nc / 2 + head] = p2 + b2h;
                }
            }
            if (row < g.n && col < g.Nc) {
                v.x += bv.x; v.y += bv.y; v.z += bv.z; v.w += bv.w;
                store_stream4(g.C + row * g.ldc + col, v);
                if (g.C16) *reinterpret_cast<uint2 *>(g.C16 + row * g.ldc16 + col / 2) = make_uint2(pack_bf16x2(v.x, v.y), pack_bf16x2(v.z, v.w));
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
        for (int s = 0; s < NS; ++s) cur[s] = nxt[s];
    }
}

// f1/f2 from finished Wh rows when hd > 64 (a row spans several column tiles).
template <int G>
__global__ __launch_bounds__(256) void logit_halves_kernel(const float *Wh, int64_t n, int H, int LPH,
                                                            const float *a1, const float *a2, const float *b1,
                                                            const float *b2, float *f1, float *f2, float *rec, int64_t ldrec) {
    constexpr int RPW = 64 / G;
    constexpr int HD = 4 * G;
    const int lane = threadIdx.x & 63;
    const int sub = lane % G, grp = lane / G;
    const int head = sub / LPH;
    const float4 a1v = *reinterpret_cast<const float4 *>(a1 + sub * 4);
    const float4 a2v = *reinterpret_cast<const float4 *>(a2 + sub * 4);
    const int64_t stride = (int64_t)gridDim.x * 4 * RPW;
    for (int64_t row = ((int64_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * RPW + grp; row < n; row += stride) {
        const float4 v = *reinterpret_cast<const float4 *>(Wh + row * HD + sub * 4);
        float p1 = v.x * a1v.x + v.y * a1v.y + v.z * a1v.z + v.w * a1v.w;
        float p2 = v.x * a2v.x + v.y * a2v.y + v.z * a2v.z + v.w * a2v.w;
        for (int o = 1; o < LPH; o <<= 1) {
            p1 += __shfl_xor(p1, o, 64);
            p2 += __shfl_xor(p2, o, 64);
        }
        if (sub % LPH == 0) {
            f1[row * H + head] = p1 + b1[head];
            f2[row * H + head] = p2 + b2[head];
            if (rec) rec[row * ldrec + HD / 2 + head] = p2 + b2[head];     // bf16 record of the row: f2 behind its HD bf16 values
        }
    }
}

// ------------------------------------------------------------------------------------------
// out[r, :] = sum_k val[k] * M[idx[k], :] over the stored entries k of row r of a CSR matrix:
// the projection Wh = X . W for sparse node features (X in CSR, M = W) and, with the CSR of X^T
// and M = dWh, the weight gradient dW = X^T . dWh.  G = width/4 lanes per row, entries in stored
// order (fixed summation order).  With a1 != nullptr the logit halves f1 / f2 of the finished row
// are written as well.
// ------------------------------------------------------------------------------------------
template <int G>
__global__ __launch_bounds__(256) void spmm_rows_kernel(const int64_t *rowptr, const int32_t *idx, const float *val,
                                                         int64_t n_rows, const float *M, float *out, int H, int LPH,
                                                         const float *a1, const float *a2, const float *b1,
                                                         const float *b2, float *f1, float *f2) {
    constexpr int RPW = 64 / G;
    constexpr int HD = 4 * G;
    const int lane = threadIdx.x & 63;
    const int sub = lane % G, grp = lane / G;
    const int64_t stride = (int64_t)gridDim.x * 4 * RPW;
    for (int64_t row = ((int64_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * RPW + grp; row < n_rows; row += stride) {
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        const int64_t end = rowptr[row + 1];
        for (int64_t k = rowptr[row]; k < end; ++k) {
            const float v = val[k];
            const float4 m = *reinterpret_cast<const float4 *>(M + (int64_t)idx[k] * HD + sub * 4);
            acc.x = fmaf(v, m.x, acc.x); acc.y = fmaf(v, m.y, acc.y);
            acc.z = fmaf(v, m.z, acc.z); acc.w = fmaf(v, m.w, acc.w);
        }
        *reinterpret_cast<float4 *>(out + row * HD + sub * 4) = acc;
        if (a1) {
            const float4 a1v = *reinterpret_cast<const float4 *>(a1 + sub * 4);
            const float4 a2v = *reinterpret_cast<const float4 *>(a2 + sub * 4);
            float p1 = acc.x * a1v.x + acc.y * a1v.y + acc.z * a1v.z + acc.w * a1v.w;
            float p2 = acc.x * a2v.x + acc.y * a2v.y + acc.z * a2v.z + acc.w * a2v.w;
            for (int o = 1; o < LPH; o <<= 1) {
                p1 += __shfl_xor(p1, o, 64);
                p2 += __shfl_xor(p2, o, 64);
            }
            if (sub % LPH == 0) {
                const int head = sub / LPH;
                f1[row * H + head] = p1 + b1[head];
                f2[row * H + head] = p2 + b2[head];
            }
        }
    }
}

// ------------------------------------------------------------------------------------------
// dW = X^T . dWh, reduced over nodes.  Block (bx, by): node chunks bx, bx+gridDim.x, ...;
// feature rows [by*16*MT, +16*MT).  Each wave keeps an [16*MT x 16*NT] partial in registers.
// ------------------------------------------------------------------------------------------
constexpr int kRedBlocks = 768;
constexpr int kNodeChunk = 64;

template <int MT, int NT>
__global__ __launch_bounds__(256) void xt_gemm_kernel(const float *X, int64_t n, int F, int64_t ldx,
                                                       const float *dY, int ldy, int hd, float *ws) {
    __shared__ float red[MT * 16 * (NT * 16 + 1)];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int m = lane & 15, kq = lane >> 4;
    const int f0 = blockIdx.y * MT * 16;
    f32x4 acc[MT][NT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};

    int fcol[MT];
    bool fok[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        fcol[mt] = f0 + 16 * mt + m;
        fok[mt] = fcol[mt] < F;
        if (!fok[mt]) fcol[mt] = 0;
    }
    int ycol[NT];
    bool yok[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        ycol[nt] = 16 * nt + m;
        yok[nt] = ycol[nt] < hd;
        if (!yok[nt]) ycol[nt] = 0;
    }
    const int64_t wave_id = (int64_t)blockIdx.x * 4 + wave;
    const int64_t wave_stride = (int64_t)gridDim.x * 4 * kNodeChunk;
    for (int64_t base = wave_id * kNodeChunk; base < n; base += wave_stride) {
#pragma unroll 4
        for (int s = 0; s < kNodeChunk / 4; ++s) {
            const int64_t node = base + 4 * s + kq;
            const bool nok = node < n;
            float a[MT], b[NT];
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) a[mt] = (nok && fok[mt]) ? X[node * ldx + fcol[mt]] : 0.f;
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) b[nt] = (nok && yok[nt]) ? dY[node * ldy + ycol[nt]] : 0.f;
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
                    acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[mt], b[nt], acc[mt][nt], 0, 0, 0);
        }
    }
    // sum the four waves in a fixed order, then park the block's partial
    constexpr int LD = NT * 16 + 1;
    for (int w = 0; w < 4; ++w) {
        if (wave == w) {
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        float *p = red + (16 * mt + 4 * kq + r) * LD + 16 * nt + m;
                        *p = (w == 0) ? acc[mt][nt][r] : *p + acc[mt][nt][r];
                    }
        }
        __syncthreads();
    }
    // ws layout: [gridDim.x][F_pad = gridDim.y*16*MT][NT*16]
    const int Fp = gridDim.y * MT * 16;
    float *out = ws + ((int64_t)blockIdx.x * Fp + f0) * (NT * 16);
    for (int e = threadIdx.x; e < MT * 16 * NT * 16; e += 256) {
        const int r = e / (NT * 16), c = e % (NT * 16);
        out[(int64_t)r * (NT * 16) + c] = red[r * LD + c];
    }
}


// ------------------------------------------------------------------------------------------
// dW = X^T . dY with whole-line loads (X rows and dY rows 16-byte aligned).  Lane (m, kq) of a wave
// loads float4 X[node+kq][64*fs + 4m ..] and float4 dY[node+kq][y0 + 4m ..]: 16 lanes cover 256
// contiguous bytes of one node's row.  Component j of the X vector is the A operand of tile
// (fs, j) (feature 64*fs + 4*row + j), component c of the dY vector the B operand of tile c
// (column y0 + 4*col + c): a fixed permutation of the output, undone when the partial is parked.
// Block (bx, by, bz): node chunks bx, bx+gridDim.x, ...; features [128*by, +64*FS); columns [64*bz, +64).
// ------------------------------------------------------------------------------------------
struct InDrop {
    float keep;                  // 0 = off
    int f4;
    uint64_t seed;
    const uint64_t *seed_step;
};

template <int FS, bool DROPIN>
__global__ __launch_bounds__(256, FS == 1 ? 3 : 2) void xt_gemm_lines_kernel(const float *X, int64_t n, int F, int64_t ldx,
                                                             const float *dY, int hd, float *ws, InDrop dr) {
    __shared__ float red[64 * FS * 65];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int m = lane & 15, kq = lane >> 4;
    const int f0 = blockIdx.y * 64 * FS;
    const int y0 = blockIdx.z * 64;
    f32x4 acc[FS][4][4];
#pragma unroll
    for (int fs = 0; fs < FS; ++fs)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int c = 0; c < 4; ++c) acc[fs][j][c] = (f32x4){0.f, 0.f, 0.f, 0.f};
    bool fok[FS];
#pragma unroll
    for (int fs = 0; fs < FS; ++fs) fok[fs] = f0 + 64 * fs + 4 * m < F;      // F % 4 == 0: a float4 is all in or all out
    const bool yok = y0 + 4 * m < hd;
    const f32x4 zero = (f32x4){0.f, 0.f, 0.f, 0.f};
    const uint64_t dseed = DROPIN ? seed_now(dr) : 0;
    // Every wave owns one contiguous run of rows (a multiple of four long) and walks it four rows a step.
    // Software pipeline: the rows of step t + 1 are requested before the matrix instructions of step t issue, so a
    // wave covers its own load latency.  The loads are unconditional (lanes past the run read its last row, lanes
    // past the width read column 0, and their values are replaced by zeros when used): a load under a branch
    // makes the compiler wait for ALL outstanding loads before the matrix instructions, the prefetched ones included.
    const int64_t n_waves = (int64_t)gridDim.x * 4;
    const int64_t per_wave = ((n + n_waves - 1) / n_waves + 3) & ~(int64_t)3;
    const int64_t start = ((int64_t)blockIdx.x * 4 + wave) * per_wave;
    const int rem = (int)(start < n ? (n - start < per_wave ? n - start : per_wave) : 0);      // rows of this wave
    const unsigned ldx_bytes = (unsigned)(ldx * sizeof(float)), ldy_bytes = (unsigned)(hd * sizeof(float));
    const char *xb[FS];
#pragma unroll
    for (int fs = 0; fs < FS; ++fs) xb[fs] = reinterpret_cast<const char *>(X + start * ldx + (fok[fs] ? f0 + 64 * fs + 4 * m : 0));
    const char *yb0 = reinterpret_cast<const char *>(dY + start * hd + (yok ? y0 + 4 * m : 0));
    auto load_step = [&](int r, f32x4 (&xa)[FS], f32x4 &yb) {
        const unsigned rr = (unsigned)(r < rem ? r : rem - 1);
#pragma unroll
        for (int fs = 0; fs < FS; ++fs)
            xa[fs] = __builtin_nontemporal_load(reinterpret_cast<const f32x4 *>(xb[fs] + (uint64_t)rr * ldx_bytes));
        yb = __builtin_nontemporal_load(reinterpret_cast<const f32x4 *>(yb0 + (uint64_t)rr * ldy_bytes));
    };
    constexpr int D = 2;                   // steps in flight behind the one being multiplied
    f32x4 xq[D][FS], yq[D];
    if (rem > 0) {
#pragma unroll
        for (int d = 0; d < D; ++d) load_step(kq + 4 * d, xq[d], yq[d]);
    }
    for (int r = kq; r - kq < rem; r += 4 * D) {
#pragma unroll
        for (int d = 0; d < D; ++d) {
            const bool nok = r + 4 * d < rem;      // a group's later steps may lie past the run: they multiply zeros
            f32x4 xa[FS], yb;
#pragma unroll
            for (int fs = 0; fs < FS; ++fs) {
                xa[fs] = (nok && fok[fs]) ? xq[d][fs] : zero;
                if constexpr (DROPIN) {                  // the forward projection's mask of this piece of X (utils/layers.py:9)
                    const float4 t = drop_in4(make_float4(xa[fs][0], xa[fs][1], xa[fs][2], xa[fs][3]), dseed, start + r + 4 * d, dr.f4,
                                              (f0 + 64 * fs + 4 * m) >> 2, dr.keep);
                    xa[fs] = (f32x4){t.x, t.y, t.z, t.w};
                }
            }
            yb = (nok && yok) ? yq[d] : zero;
            load_step(r + 4 * (d + D), xq[d], yq[d]);          // past the end: the last row again, never used
#pragma unroll
            for (int fs = 0; fs < FS; ++fs)
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int c = 0; c < 4; ++c)
                        acc[fs][j][c] = __builtin_amdgcn_mfma_f32_16x16x4f32(xa[fs][j], yb[c], acc[fs][j][c], 0, 0, 0);
        }
    }
    // fixed-order sum of the four waves; red[feature_local][column_local]
    for (int w = 0; w < 4; ++w) {
        if (wave == w) {
#pragma unroll
            for (int fs = 0; fs < FS; ++fs)
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int c = 0; c < 4; ++c)
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            float *p = red + (64 * fs + 4 * (4 * kq + r) + j) * 65 + 4 * m + c;
                            *p = (w == 0) ? acc[fs][j][c][r] : *p + acc[fs][j][c][r];
                        }
        }
        __syncthreads();
    }
    // ws layout: [gridDim.x][Fp = gridDim.y*64*FS][hdp = gridDim.z*64]
    const int Fp = gridDim.y * 64 * FS, hdp = gridDim.z * 64;
    float *out = ws + ((int64_t)blockIdx.x * Fp + f0) * hdp + y0;
    for (int e = threadIdx.x; e < 64 * FS * 64; e += 256) {
        const int r = e >> 6, c = e & 63;
        out[(int64_t)r * hdp + c] = red[r * 65 + c];
    }
}

// dW[f][c] = sum_b ws[b][f][c]   (f < F, c < hd; ws rows are NT16 wide)
// dW[f, c] (row stride ldw) = sum over blocks of ws[b][f][c] for f < F, c < hd
__global__ void reduce_dw_kernel(const float *ws, int nblocks, int Fp, int NT16, int F, int hd, int ldw, float *dW) {
    const int e = blockIdx.x * kSumCols + threadIdx.x;
    const bool ok = e < F * hd;
    const int f = ok ? e / hd : 0, c = ok ? e % hd : 0;
    const float t = block_column_sum(nblocks, [&](int b) { return ok ? ws[((int64_t)b * Fp + f) * NT16 + c] : 0.f; });
    if (ok && threadIdx.y == 0) dW[(int64_t)f * ldw + c] = t;
}

// da1 = sum_i df1[i,h] Wh[i,h,:], da2 likewise with df2, db1 = sum_i df1, db2 = sum_i df2.
template <int G>
__global__ __launch_bounds__(256) void attn_param_grad_kernel(const float *Wh, const float *df1, const float *df2,
                                                               int64_t n, int H, int LPH, float *ws) {
    constexpr int RPW = 64 / G;
    constexpr int HD = 4 * G;
    __shared__ float4 red[256];
    const int lane = threadIdx.x & 63;
    const int sub = lane % G, grp = lane / G;
    const int head = sub / LPH;
    float4 s1 = make_float4(0.f, 0.f, 0.f, 0.f), s2 = s1;
    float t1 = 0.f, t2 = 0.f;
    const int64_t stride = (int64_t)gridDim.x * 4 * RPW;
    // four rows per lane group in flight (the loop is a plain stream: its speed is the bytes it keeps outstanding);
    // rows beyond n contribute exact zeros, the order of the sums does not depend on the grid
    for (int64_t row0 = ((int64_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * RPW + grp; row0 < n; row0 += 4 * stride) {
        float4 v[4];
        float d1[4], d2[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int64_t row = row0 + u * stride;
            const bool ok = row < n;
            v[u] = ok ? load_stream4(Wh + row * HD + sub * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
            d1[u] = ok ? df1[row * H + head] : 0.f;
            d2[u] = ok ? df2[row * H + head] : 0.f;
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            s1 = f4_fma(d1[u], v[u], s1);
            s2 = f4_fma(d2[u], v[u], s2);
            t1 += d1[u];
            t2 += d2[u];
        }
    }
    // block partial layout in ws: [da1 HD | da2 HD | db1 H | db2 H]
    float *out = ws + (int64_t)blockIdx.x * (2 * HD + 2 * H);
    for (int pass = 0; pass < 3; ++pass) {
        red[threadIdx.x] = pass == 0 ? s1 : (pass == 1 ? s2 : make_float4(t1, t2, 0.f, 0.f));
        __syncthreads();
        if (threadIdx.x < G) {
            float4 t = make_float4(0.f, 0.f, 0.f, 0.f);
            for (int i = threadIdx.x; i < 256; i += G) t = make_float4(t.x + red[i].x, t.y + red[i].y, t.z + red[i].z, t.w + red[i].w);
            if (pass < 2) *reinterpret_cast<float4 *>(out + pass * HD + threadIdx.x * 4) = t;
            else if (threadIdx.x % LPH == 0) {
                out[2 * HD + threadIdx.x / LPH] = t.x;
                out[2 * HD + H + threadIdx.x / LPH] = t.y;
            }
        }
        __syncthreads();
    }
}

__global__ void reduce_attn_param_kernel(const float *ws, int nblocks, int hd, int H, float *da1, float *da2,
                                         float *db1, float *db2) {
    const int width = 2 * hd + 2 * H;
    const int c = blockIdx.x * kSumCols + threadIdx.x;
    const bool ok = c < width;
    const float t = block_column_sum(nblocks, [&](int b) { return ok ? ws[(int64_t)b * width + c] : 0.f; });
    if (!ok || threadIdx.y != 0) return;
    if (c < hd) da1[c] = t;
    else if (c < 2 * hd) da2[c - hd] = t;
    else if (c < 2 * hd + H) db1[c - 2 * hd] = t;
    else db2[c - 2 * hd - H] = t;
}

template <int NS>
static int launch_resident(const GemmArgs &g, hipStream_t st, const char *name) {
    const int64_t n_tiles = (g.n + 15) / 16;
    const unsigned nb = (unsigned)std::min<int64_t>(512, (n_tiles + kResWaves - 1) / kResWaves);
    const size_t lds = (size_t)(NS * 16 + kResWaves * 16) * kLdB * sizeof(float);
    if (g.a_keep != 0.f) {
        if (g.Nc <= 16) hipLaunchKernelGGL((proj_resident_kernel<NS, 1, true>), dim3(nb), dim3(64 * kResWaves), lds, st, g, n_tiles);
        else hipLaunchKernelGGL((proj_resident_kernel<NS, 4, true>), dim3(nb), dim3(64 * kResWaves), lds, st, g, n_tiles);
    } else if (g.Nc <= 16) hipLaunchKernelGGL((proj_resident_kernel<NS, 1, false>), dim3(nb), dim3(64 * kResWaves), lds, st, g, n_tiles);
    else hipLaunchKernelGGL((proj_resident_kernel<NS, 4, false>), dim3(nb), dim3(64 * kResWaves), lds, st, g, n_tiles);
    GATMI_LAUNCH_CHECK(name);
    return GATMI_OK;
}

static int launch_tall_gemm(const GemmArgs &g, hipStream_t st, const char *name) {
    if (g.n == 0 || g.Nc == 0) return GATMI_OK;
    const bool lines = (g.lda % 4 == 0) && (g.K % 4 == 0) && aligned16(g.A) && (g.ldc % 4 == 0) && aligned16(g.C);
    if (lines && g.sbc == 1 && g.K <= 128 && g.Nc <= 64 && g.Nc % 4 == 0 && !g.accumulate && g.n >= 4096 && g.c_keep == 0.f &&
        (!g.cbias || aligned16(g.cbias))) {
        switch ((g.K + 15) / 16) {
            case 1: return launch_resident<1>(g, st, name);
            case 2: return launch_resident<2>(g, st, name);
            case 3: return launch_resident<3>(g, st, name);
            case 4: return launch_resident<4>(g, st, name);
            case 5: return launch_resident<5>(g, st, name);
            case 6: return launch_resident<6>(g, st, name);
            case 7: return launch_resident<7>(g, st, name);
            default: return launch_resident<8>(g, st, name);
        }
    }
    const unsigned gy = (unsigned)((g.Nc + kTileCols - 1) / kTileCols);
    const bool vec = (g.lda % 4 == 0) && (g.K % 4 == 0) && aligned16(g.A);
    GATMI_REQUIRE(g.a_keep == 0.f || vec, GATMI_EUNSUPPORTED, "%s: input dropout inside the projection needs 16-byte aligned rows of a width that is a multiple of 4", name);
    GATMI_REQUIRE(g.c_keep == 0.f || ((g.ldc & 3) == 0 && aligned16(g.C) && g.Nc % 4 == 0), GATMI_EUNSUPPORTED,
                  "%s: input dropout on the input gradient needs 16-byte aligned rows of a width that is a multiple of 4", name);
    // few row blocks and a long K: split K over the waves of smaller blocks
    const bool splitk = g.K >= 256 && (g.n + kTileRows - 1) / kTileRows * gy < 512;
    if (splitk) {
        const dim3 grid((unsigned)((g.n + 31) / 32), gy);
        if (vec) hipLaunchKernelGGL((tall_gemm_kernel<true, true>), grid, dim3(256), 0, st, g);
        else hipLaunchKernelGGL((tall_gemm_kernel<false, true>), grid, dim3(256), 0, st, g);
    } else {
        const dim3 grid((unsigned)((g.n + kTileRows - 1) / kTileRows), gy);
        if (vec) hipLaunchKernelGGL((tall_gemm_kernel<true, false>), grid, dim3(256), 0, st, g);
        else hipLaunchKernelGGL((tall_gemm_kernel<false, false>), grid, dim3(256), 0, st, g);
    }
    GATMI_LAUNCH_CHECK(name);
    return GATMI_OK;
}

template <int G>
static int launch_logit_halves(const float *Wh, int64_t n, int H, int LPH, const float *a1, const float *a2,
                               const float *b1, const float *b2, float *f1, float *f2, float *rec, int64_t ldrec, hipStream_t st) {
    const int64_t per_block = 4 * (64 / G);
    const unsigned nb = (unsigned)std::min<int64_t>(2048, std::max<int64_t>(1, (n + per_block - 1) / per_block));
    hipLaunchKernelGGL(logit_halves_kernel<G>, dim3(nb), dim3(256), 0, st, Wh, n, H, LPH, a1, a2, b1, b2, f1, f2, rec, ldrec);
    GATMI_LAUNCH_CHECK("logit_halves");
    return GATMI_OK;
}

template <int G>
static int launch_attn_param_grad(const float *Wh, const float *df1, const float *df2, int64_t n, int H, int LPH,
                                  float *ws, float *da1, float *da2, float *db1, float *db2, hipStream_t st) {
    const int64_t per_block = 4 * (64 / G);
    const unsigned nb = (unsigned)std::min<int64_t>(kRedBlocks, std::max<int64_t>(1, (n + per_block - 1) / per_block));
    hipLaunchKernelGGL(attn_param_grad_kernel<G>, dim3(nb), dim3(256), 0, st, Wh, df1, df2, n, H, LPH, ws);
    GATMI_LAUNCH_CHECK("attn_param_grad");
    const int width = 2 * 4 * G + 2 * H;
    hipLaunchKernelGGL(reduce_attn_param_kernel, dim3((width + kSumCols - 1) / kSumCols), dim3(kSumCols, kSumRows), 0, st, ws, (int)nb, 4 * G, H, da1, da2, db1, db2);
    GATMI_LAUNCH_CHECK("reduce_attn_param");
    return GATMI_OK;
}

static int launch_xt_lines(const float *X, int64_t n, int F, int64_t ldx, const float *dY, int hd, float *ws,
                           float *dW, const InDrop &dr, hipStream_t st) {
    const int FS = F <= 64 ? 1 : 2;
    const unsigned gy = (unsigned)((F + 64 * FS - 1) / (64 * FS)), gz = (unsigned)((hd + 63) / 64);
    const unsigned gx = (unsigned)std::min<int64_t>(kRedBlocks, std::max<int64_t>(1, (n + 4 * kNodeChunk - 1) / (4 * kNodeChunk)));
    const dim3 grid(gx, gy, gz);
    if (dr.keep != 0.f) {
        if (FS == 1) hipLaunchKernelGGL((xt_gemm_lines_kernel<1, true>), grid, dim3(256), 0, st, X, n, F, ldx, dY, hd, ws, dr);
        else hipLaunchKernelGGL((xt_gemm_lines_kernel<2, true>), grid, dim3(256), 0, st, X, n, F, ldx, dY, hd, ws, dr);
    } else if (FS == 1) hipLaunchKernelGGL((xt_gemm_lines_kernel<1, false>), grid, dim3(256), 0, st, X, n, F, ldx, dY, hd, ws, dr);
    else hipLaunchKernelGGL((xt_gemm_lines_kernel<2, false>), grid, dim3(256), 0, st, X, n, F, ldx, dY, hd, ws, dr);
    GATMI_LAUNCH_CHECK("xt_gemm_lines");
    hipLaunchKernelGGL(reduce_dw_kernel, dim3((F * hd + kSumCols - 1) / kSumCols), dim3(kSumCols, kSumRows), 0, st, ws, (int)gx,
                       (int)(gy * 64 * FS), (int)(gz * 64), F, hd, hd, dW);
    GATMI_LAUNCH_CHECK("reduce_dw");
    return GATMI_OK;
}

template <int MT>
static int launch_xt_gemm(const float *X, int64_t n, int F, int64_t ldx, const float *dY, int ldy, int hd, float *ws,
                          float *dW, hipStream_t st) {
    const int NT = (hd + 15) / 16;
    const unsigned gy = (unsigned)((F + MT * 16 - 1) / (MT * 16));
    const unsigned gx = (unsigned)std::min<int64_t>(kRedBlocks, std::max<int64_t>(1, (n + 4 * kNodeChunk - 1) / (4 * kNodeChunk)));
    const dim3 grid(gx, gy);
    int nt_pad;
    if (NT <= 1) { nt_pad = 1; hipLaunchKernelGGL((xt_gemm_kernel<MT, 1>), grid, dim3(256), 0, st, X, n, F, ldx, dY, ldy, hd, ws); }
    else if (NT <= 2) { nt_pad = 2; hipLaunchKernelGGL((xt_gemm_kernel<MT, 2>), grid, dim3(256), 0, st, X, n, F, ldx, dY, ldy, hd, ws); }
    else { nt_pad = 4; hipLaunchKernelGGL((xt_gemm_kernel<MT, 4>), grid, dim3(256), 0, st, X, n, F, ldx, dY, ldy, hd, ws); }
    GATMI_LAUNCH_CHECK("xt_gemm");
    const int Fp = gy * MT * 16;
    hipLaunchKernelGGL(reduce_dw_kernel, dim3((F * hd + kSumCols - 1) / kSumCols), dim3(kSumCols, kSumRows), 0, st, ws, (int)gx, Fp, nt_pad * 16, F, hd, ldy, dW);
    GATMI_LAUNCH_CHECK("reduce_dw");
    return GATMI_OK;
}

// Column sums of a tall matrix Y [n, width] (row stride ld), first stage: block (b, slab) sums its run of rows for 64
// columns; four rows in flight per thread, fixed order.  ws[b][width].
__global__ __launch_bounds__(256) void column_partial_kernel(const float *Y, int64_t n, int width, int64_t ld, float *ws) {
    __shared__ float part[4][64];
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    const int c = blockIdx.y * 64 + tx;
    const bool ok = c < width;
    const int64_t per = (n + gridDim.x - 1) / gridDim.x;
    const int64_t r0 = (int64_t)blockIdx.x * per, r1 = r0 + per < n ? r0 + per : n;
    float s[4] = {0.f, 0.f, 0.f, 0.f};
    const float *col = Y + (ok ? c : 0);
    for (int64_t r = r0 + ty; r < r1; r += 16) {
#pragma unroll
        for (int u = 0; u < 4; ++u) s[u] += (ok && r + 4 * u < r1) ? col[(r + 4 * u) * ld] : 0.f;
    }
    part[ty][tx] = (s[0] + s[1]) + (s[2] + s[3]);
    __syncthreads();
    if (ty == 0 && ok) ws[(int64_t)blockIdx.x * width + c] = (part[0][tx] + part[1][tx]) + (part[2][tx] + part[3][tx]);
}

__global__ void column_total_kernel(const float *ws, int nblocks, int width, float *out) {
    const int c = blockIdx.x * kSumCols + threadIdx.x;
    const bool ok = c < width;
    const float t = block_column_sum(nblocks, [&](int b) { return ok ? ws[(int64_t)b * width + c] : 0.f; });
    if (ok && threadIdx.y == 0) out[c] = t;
}

// Y = act(X + b) for every row of X [n, w] (rows ld floats apart in both): the bias and activation of a width-1
// convolution (tf.layers.conv1d(., w, 1, activation), utils/layers.py:180,183,223-224) in one pass.
template <int ACT>
__device__ __forceinline__ float bias_act_value(float v) {
    if constexpr (ACT == GATMI_ACT_RELU) return fmaxf(v, 0.f);
    if constexpr (ACT == GATMI_ACT_ELU) return v > 0.f ? v : __expf(v) - 1.f;
    return v;
}
// the activation's derivative from its OUTPUT y (ReLU: [y > 0]; ELU: 1 for y > 0, y + 1 otherwise)
template <int ACT>
__device__ __forceinline__ float bias_act_slope(float y) {
    if constexpr (ACT == GATMI_ACT_RELU) return y > 0.f ? 1.f : 0.f;
    if constexpr (ACT == GATMI_ACT_ELU) return y > 0.f ? 1.f : y + 1.f;
    return 1.f;
}

// four columns per thread: w, ld multiples of 4 and 16-byte aligned pointers
template <int ACT>
__global__ __launch_bounds__(256) void bias_act_fwd4_kernel(const float *X, int64_t n, int w4, int64_t ld, const float *b, float *Y) {
    const int64_t total = n * w4;
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
        const int64_t r = e / w4;
        const int c = 4 * (int)(e - r * w4);
        const float4 v = *reinterpret_cast<const float4 *>(X + r * ld + c), bb = *reinterpret_cast<const float4 *>(b + c);
        *reinterpret_cast<float4 *>(Y + r * ld + c) = make_float4(bias_act_value<ACT>(v.x + bb.x), bias_act_value<ACT>(v.y + bb.y),
                                                                 bias_act_value<ACT>(v.z + bb.z), bias_act_value<ACT>(v.w + bb.w));
    }
}

template <int ACT>
__global__ __launch_bounds__(256) void bias_act_fwd_kernel(const float *X, int64_t n, int w, int64_t ld, const float *b, float *Y) {
    const int64_t total = n * w;
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
        const int64_t r = e / w;
        const int c = (int)(e - r * w);
        Y[r * ld + c] = bias_act_value<ACT>(X[r * ld + c] + b[c]);
    }
}

// dPre = dY * act'(Y) and, in the same pass, every row block's column sums of dPre (the bias gradient): the shape of
// column_partial_kernel with the product formed and stored on the way.  ws[b][width].
template <int ACT>
__global__ __launch_bounds__(256) void bias_act_bwd_kernel(const float *dY, const float *Y, int64_t n, int width, int64_t ld,
                                                           float *dPre, float *ws) {
    __shared__ float part[4][64];
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    const int c = blockIdx.y * 64 + tx;
    const bool ok = c < width;
    const int64_t per = (n + gridDim.x - 1) / gridDim.x;
    const int64_t r0 = (int64_t)blockIdx.x * per, r1 = r0 + per < n ? r0 + per : n;
    float s[4] = {0.f, 0.f, 0.f, 0.f};
    const int64_t col = ok ? c : 0;
    for (int64_t r = r0 + ty; r < r1; r += 16) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int64_t o = (r + 4 * u) * ld + col;
            if (ok && r + 4 * u < r1) {
                const float v = dY[o] * bias_act_slope<ACT>(Y[o]);
                dPre[o] = v;
                s[u] += v;
            }
        }
    }
    part[ty][tx] = (s[0] + s[1]) + (s[2] + s[3]);
    __syncthreads();
    if (ty == 0 && ok) ws[(int64_t)blockIdx.x * width + c] = (part[0][tx] + part[1][tx]) + (part[2][tx] + part[3][tx]);
}

static unsigned column_sum_blocks(int64_t n) { return (unsigned)std::min<int64_t>(kRedBlocks, std::max<int64_t>(1, (n + 63) / 64)); }

// preprocess_features (utils/process.py:155-162) for a CSR matrix: out = val / rowsum, rows that
// sum to zero stay zero (r_inv = inf is replaced by 0 there).  One thread per row, stored order.
__global__ void csr_row_normalize_kernel(int64_t n, const int64_t *rowptr, const float *val, float *out) {
    const int64_t row = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (row >= n) return;
    const int64_t b = rowptr[row], e = rowptr[row + 1];
    float sum = 0.f;
    for (int64_t k = b; k < e; ++k) sum += val[k];
    float r_inv = 1.0f / sum;
    if (isinf(r_inv)) r_inv = 0.f;
    for (int64_t k = b; k < e; ++k) out[k] = val[k] * r_inv;
}

template <int G>
static int launch_spmm_rows(const int64_t *rowptr, const int32_t *idx, const float *val, int64_t n_rows, const float *M,
                            float *out, int H, int LPH, const float *a1, const float *a2, const float *b1,
                            const float *b2, float *f1, float *f2, hipStream_t st, const char *name) {
    const int64_t rows_per_block = 4 * (64 / G);
    const int64_t nb = std::min<int64_t>((n_rows + rows_per_block - 1) / rows_per_block, 1 << 20);
    hipLaunchKernelGGL(spmm_rows_kernel<G>, dim3((unsigned)nb), dim3(256), 0, st, rowptr, idx, val, n_rows, M, out, H, LPH,
                       a1, a2, b1, b2, f1, f2);
    GATMI_LAUNCH_CHECK(name);
    return GATMI_OK;
}

}  // namespace gatmi

using namespace gatmi;

extern "C" int gatmi_proj_fwd_csr(const int64_t *xrowptr, const int32_t *xcol, const float *xval, int64_t n, int32_t f_in,
                                  const float *W, const float *a1, const float *a2, const float *b1, const float *b2,
                                  int32_t H, int32_t dp, float *Wh, float *f1, float *f2, void *stream) {
    int G, LPH;
    int rc = check_shape(H, dp, &G, &LPH);
    if (rc) return rc;
    GATMI_REQUIRE(n >= 0 && f_in > 0, GATMI_EINVAL, "proj_fwd_csr: bad dims n=%lld f_in=%d", (long long)n, f_in);
    if (n == 0) return GATMI_OK;
    GATMI_REQUIRE(xrowptr && xcol && xval && W && Wh, GATMI_EINVAL, "proj_fwd_csr: null pointer");
    const bool logits = f1 || f2;
    GATMI_REQUIRE(!logits || (f1 && f2 && a1 && a2 && b1 && b2), GATMI_EINVAL, "proj_fwd_csr: f1, f2, a1, a2, b1, b2 go together");
    GATMI_REQUIRE_ALIGNED(W); GATMI_REQUIRE_ALIGNED(Wh); GATMI_REQUIRE_ALIGNED(a1); GATMI_REQUIRE_ALIGNED(a2);
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    if (!logits) a1 = a2 = nullptr;
#define X_(gg) if (G == gg) return launch_spmm_rows<gg>(xrowptr, xcol, xval, n, W, Wh, H, LPH, a1, a2, b1, b2, f1, f2, st, "proj_fwd_csr");
    X_(1) X_(2) X_(4) X_(8) X_(16) X_(32) X_(64)
#undef X_
    return fail(GATMI_EUNSUPPORTED, "proj_fwd_csr: no kernel for %d lanes per row", G);
}

extern "C" int gatmi_proj_bwd_params_csr(const int64_t *xt_rowptr, const int32_t *xt_row, const float *xt_val,
                                         int64_t n, int32_t f_in, const float *dWh, const float *Wh,
                                         const float *df1, const float *df2, int32_t H, int32_t dp,
                                         float *dW, float *da1, float *da2, float *db1, float *db2,
                                         float *ws, void *stream) {
    int G, LPH;
    int rc = check_shape(H, dp, &G, &LPH);
    if (rc) return rc;
    GATMI_REQUIRE(n >= 0 && f_in > 0, GATMI_EINVAL, "proj_bwd_params_csr: bad dims");
    GATMI_REQUIRE(xt_rowptr && dWh && dW, GATMI_EINVAL, "proj_bwd_params_csr: null pointer");
    GATMI_REQUIRE(!Wh || (df1 && df2 && da1 && da2 && db1 && db2 && ws), GATMI_EINVAL, "proj_bwd_params_csr: Wh needs df1, df2, da1, da2, db1, db2, ws");
    GATMI_REQUIRE_ALIGNED(Wh); GATMI_REQUIRE_ALIGNED(ws); GATMI_REQUIRE_ALIGNED(dWh); GATMI_REQUIRE_ALIGNED(dW);
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    if (Wh) {
#define X_(gg) if (G == gg) rc = launch_attn_param_grad<gg>(Wh, df1, df2, n, H, LPH, ws, da1, da2, db1, db2, st);
        X_(1) X_(2) X_(4) X_(8) X_(16) X_(32) X_(64)
#undef X_
        if (rc) return rc;
    }
#define X_(gg) if (G == gg) return launch_spmm_rows<gg>(xt_rowptr, xt_row, xt_val, f_in, dWh, dW, H, LPH, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, st, "proj_bwd_params_csr");
    X_(1) X_(2) X_(4) X_(8) X_(16) X_(32) X_(64)
#undef X_
    return fail(GATMI_EUNSUPPORTED, "proj_bwd_params_csr: no kernel for %d lanes per row", G);
}

extern "C" int gatmi_proj_fwd(const float *X, int64_t n, int32_t f_in, int64_t ldx,
                              const float *W, const float *a1, const float *a2,
                              const float *b1, const float *b2, int32_t H, int32_t dp,
                              float *Wh, uint16_t *Wh16, int64_t ld16, float *f1, float *f2,
                              const float *W_res, const float *b_res, float *res_out,
                              float in_keep, uint64_t seed, void *stream) {
    int G, LPH;
    int rc = check_shape(H, dp, &G, &LPH);
    if (rc) return rc;
    GATMI_REQUIRE(n >= 0 && f_in > 0 && ldx >= f_in, GATMI_EINVAL, "proj_fwd: bad dims n=%lld f_in=%d ldx=%lld", (long long)n, f_in, (long long)ldx);
    GATMI_REQUIRE(in_keep > 0.f && in_keep <= 1.f, GATMI_EINVAL, "proj_fwd: in_keep must be in (0,1]");
    GATMI_REQUIRE(in_keep >= 1.f || f_in % 4 == 0, GATMI_EUNSUPPORTED, "proj_fwd: input dropout inside the projection needs f_in %% 4 == 0 (f_in = %d); mask the input outside", f_in);
    if (n == 0) return GATMI_OK;
    GATMI_REQUIRE(X && W && Wh, GATMI_EINVAL, "proj_fwd: null pointer");
    const bool logits = f1 || f2;
    GATMI_REQUIRE(!logits || (f1 && f2 && a1 && a2 && b1 && b2), GATMI_EINVAL, "proj_fwd: f1, f2, a1, a2, b1, b2 go together");
    GATMI_REQUIRE((W_res == nullptr) == (res_out == nullptr), GATMI_EINVAL, "proj_fwd: W_res and res_out go together");
    GATMI_REQUIRE_ALIGNED(Wh); GATMI_REQUIRE_ALIGNED(a1); GATMI_REQUIRE_ALIGNED(a2); GATMI_REQUIRE_ALIGNED(res_out); GATMI_REQUIRE_ALIGNED(Wh16);
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const int hd = H * dp;
    GemmArgs g{};
    GATMI_REQUIRE(!Wh16 || (logits && ld16 >= (int64_t)hd + 2 * H && ld16 % 8 == 0), GATMI_EINVAL,
                  "proj_fwd: Wh16 records need the logit halves and a row stride ld16=%lld >= hd + 2H, a multiple of 8", (long long)ld16);
    g.C16 = reinterpret_cast<uint32_t *>(Wh16);
    g.ldc16 = ld16 / 2;
    g.A = X; g.n = n; g.K = f_in; g.lda = ldx;
    g.B = W; g.sbk = hd; g.sbc = 1; g.Nc = hd;
    g.C = Wh; g.ldc = hd; g.accumulate = 0; g.cbias = nullptr;
    if (in_keep < 1.f) { g.a_keep = in_keep; g.drop_f4 = f_in / 4; g.seed = seed; g.seed_step = dropout_step_source(); }
    const bool fused = logits && hd <= kTileCols;
    if (fused) { g.a1 = a1; g.a2 = a2; g.b1 = b1; g.b2 = b2; g.f1 = f1; g.f2 = f2; g.H = H; g.LPH = LPH; }
    rc = launch_tall_gemm(g, st, "proj_fwd");
    if (rc) return rc;
    if (logits && !fused) {
        float *rec = reinterpret_cast<float *>(Wh16);
        if (G == 32) rc = launch_logit_halves<32>(Wh, n, H, LPH, a1, a2, b1, b2, f1, f2, rec, ld16 / 2, st);
        else rc = launch_logit_halves<64>(Wh, n, H, LPH, a1, a2, b1, b2, f1, f2, rec, ld16 / 2, st);
        if (rc) return rc;
    }
    if (W_res) {
        GemmArgs r{};
        r.A = X; r.n = n; r.K = f_in; r.lda = ldx;
        r.B = W_res; r.sbk = hd; r.sbc = 1; r.Nc = hd;
        r.C = res_out; r.ldc = hd; r.accumulate = 0; r.cbias = b_res;
        r.a_keep = g.a_keep; r.drop_f4 = g.drop_f4; r.seed = g.seed; r.seed_step = g.seed_step;      // the residual branch reads the same dropped input (utils/layers.py:30)
        rc = launch_tall_gemm(r, st, "proj_fwd residual");
        if (rc) return rc;
    }
    return GATMI_OK;
}

static inline int dw_mt(int f_in) { return f_in <= 16 ? 1 : (f_in <= 32 ? 2 : (f_in <= 64 ? 4 : 7)); }

extern "C" size_t gatmi_proj_bwd_ws_floats(int32_t f_in, int32_t H, int32_t dp) {
    const int hd = H * dp;
    const int mt = dw_mt(f_in);
    const size_t fp = (size_t)((f_in + mt * 16 - 1) / (mt * 16)) * mt * 16;
    const size_t fs = f_in <= 64 ? 64 : 128;
    const size_t dw_lines = (size_t)kRedBlocks * (((size_t)f_in + fs - 1) / fs * fs) * (((size_t)hd + 63) / 64 * 64);
    const size_t dw = std::max((size_t)kRedBlocks * fp * 64, dw_lines);
    const size_t da = (size_t)kRedBlocks * (size_t)(2 * hd + 2 * H);
    return std::max(dw, da);
}

extern "C" int gatmi_proj_bwd_params(const float *X, int64_t n, int32_t f_in, int64_t ldx,
                                     const float *dWh, const float *Wh, const float *df1, const float *df2,
                                     int32_t H, int32_t dp,
                                     float *dW, float *da1, float *da2, float *db1, float *db2,
                                     float *ws, float in_keep, uint64_t seed, void *stream) {
    int G, LPH;
    int rc = check_shape(H, dp, &G, &LPH);
    if (rc) return rc;
    GATMI_REQUIRE(n >= 0 && f_in > 0 && ldx >= f_in, GATMI_EINVAL, "proj_bwd_params: bad dims");
    GATMI_REQUIRE(in_keep > 0.f && in_keep <= 1.f, GATMI_EINVAL, "proj_bwd_params: in_keep must be in (0,1]");
    GATMI_REQUIRE(X && dWh && dW && ws, GATMI_EINVAL, "proj_bwd_params: null pointer");
    GATMI_REQUIRE(!Wh || (df1 && df2 && da1 && da2 && db1 && db2), GATMI_EINVAL, "proj_bwd_params: Wh needs df1, df2, da1, da2, db1, db2");
    GATMI_REQUIRE_ALIGNED(Wh); GATMI_REQUIRE_ALIGNED(ws);
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const int hd = H * dp;
    if (Wh) {
#define X_(gg) if (G == gg) rc = launch_attn_param_grad<gg>(Wh, df1, df2, n, H, LPH, ws, da1, da2, db1, db2, st);
        X_(1) X_(2) X_(4) X_(8) X_(16) X_(32) X_(64)
#undef X_
        if (rc) return rc;
    }
    // (the kernel forms row offsets as 32-bit row x 32-bit row stride in bytes)
    InDrop dr{};
    if (in_keep < 1.f) { dr.keep = in_keep; dr.f4 = f_in / 4; dr.seed = seed; dr.seed_step = dropout_step_source(); }
    if (ldx % 4 == 0 && f_in % 4 == 0 && aligned16(X) && aligned16(dWh) && f_in <= 1024 && ldx < ((int64_t)1 << 29))
        return launch_xt_lines(X, n, f_in, ldx, dWh, hd, ws, dW, dr, st);
    GATMI_REQUIRE(in_keep >= 1.f, GATMI_EUNSUPPORTED, "proj_bwd_params: input dropout inside the projection needs 16-byte aligned rows of X and dWh and f_in %% 4 == 0, f_in <= 1024 (f_in = %d); mask the input outside", f_in);
    // rows of X that are not 16-byte aligned: dW in column slabs of 64 (one slab for hd <= 64); the
    // slabs share the workspace, which the stream order makes safe
    for (int c0 = 0; c0 < hd; c0 += 64) {
        const int w = std::min(64, hd - c0);
        switch (dw_mt(f_in)) {
            case 1: rc = launch_xt_gemm<1>(X, n, f_in, ldx, dWh + c0, hd, w, ws, dW + c0, st); break;
            case 2: rc = launch_xt_gemm<2>(X, n, f_in, ldx, dWh + c0, hd, w, ws, dW + c0, st); break;
            case 4: rc = launch_xt_gemm<4>(X, n, f_in, ldx, dWh + c0, hd, w, ws, dW + c0, st); break;
            default: rc = launch_xt_gemm<7>(X, n, f_in, ldx, dWh + c0, hd, w, ws, dW + c0, st); break;
        }
        if (rc) return rc;
    }
    return GATMI_OK;
}

static bool xt_lines_ok(const float *X, int64_t ldx, int f_in, const float *dY, int w) {
    return ldx % 4 == 0 && f_in % 4 == 0 && w % 4 == 0 && aligned16(X) && aligned16(dY) && f_in <= 1024 && ldx < ((int64_t)1 << 29);
}

extern "C" size_t gatmi_linear_bwd_ws_floats(int64_t n, int32_t f_in, int32_t w) {
    if (n <= 0 || f_in <= 0 || w <= 0) return 1;
    const size_t gx = (size_t)std::min<int64_t>(kRedBlocks, std::max<int64_t>(1, (n + 4 * kNodeChunk - 1) / (4 * kNodeChunk)));
    const size_t fs = f_in <= 64 ? 64 : 128;
    const size_t lines = gx * (((size_t)f_in + fs - 1) / fs * fs) * (((size_t)w + 63) / 64 * 64);
    const int mt = dw_mt(f_in);
    const size_t slabs = gx * ((size_t)((f_in + mt * 16 - 1) / (mt * 16)) * mt * 16) * 64;
    return std::max(std::max(lines, slabs), (size_t)column_sum_blocks(n) * (size_t)w);
}

extern "C" int gatmi_linear_bwd_params(const float *X, int64_t n, int32_t f_in, int64_t ldx, const float *dY, int32_t w,
                                       float *dW, float *db, float *ws, void *stream) {
    GATMI_REQUIRE(n >= 0 && f_in > 0 && w > 0 && ldx >= f_in, GATMI_EINVAL, "linear_bwd_params: bad dims");
    GATMI_REQUIRE(n < ((int64_t)1 << 31), GATMI_ERANGE, "linear_bwd_params: more than 2^31 rows");
    GATMI_REQUIRE(X && dY && dW && ws, GATMI_EINVAL, "linear_bwd_params: null pointer");
    GATMI_REQUIRE_ALIGNED(ws);
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    int rc = GATMI_OK;
    if (xt_lines_ok(X, ldx, f_in, dY, w)) {
        rc = launch_xt_lines(X, n, f_in, ldx, dY, w, ws, dW, InDrop{}, st);
    } else {
        for (int c0 = 0; c0 < w && !rc; c0 += 64) {          // column slabs share the workspace: the stream orders them
            const int cw = std::min(64, w - c0);
            switch (dw_mt(f_in)) {
                case 1: rc = launch_xt_gemm<1>(X, n, f_in, ldx, dY + c0, w, cw, ws, dW + c0, st); break;
                case 2: rc = launch_xt_gemm<2>(X, n, f_in, ldx, dY + c0, w, cw, ws, dW + c0, st); break;
                case 4: rc = launch_xt_gemm<4>(X, n, f_in, ldx, dY + c0, w, cw, ws, dW + c0, st); break;
                default: rc = launch_xt_gemm<7>(X, n, f_in, ldx, dY + c0, w, cw, ws, dW + c0, st); break;
            }
        }
    }
    if (rc || !db) return rc;
    const unsigned nb = column_sum_blocks(n);
    hipLaunchKernelGGL(column_partial_kernel, dim3(nb, (unsigned)((w + 63) / 64)), dim3(256), 0, st, dY, n, w, (int64_t)w, ws);
    GATMI_LAUNCH_CHECK("column_partial");
    hipLaunchKernelGGL(column_total_kernel, dim3((unsigned)((w + kSumCols - 1) / kSumCols)), dim3(kSumCols, kSumRows), 0, st, ws, (int)nb, w, db);
    GATMI_LAUNCH_CHECK("column_total");
    return GATMI_OK;
}

extern "C" int gatmi_bias_act_fwd(const float *X, int64_t n, int32_t w, int64_t ld, const float *b, int32_t activation,
                                  float *Y, void *stream) {
    GATMI_REQUIRE(n >= 0 && w > 0 && ld >= w, GATMI_EINVAL, "bias_act_fwd: bad dims");
    GATMI_REQUIRE(X && b && Y, GATMI_EINVAL, "bias_act_fwd: null pointer");
    if (n == 0) return GATMI_OK;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    if (w % 4 == 0 && ld % 4 == 0 && aligned16(X) && aligned16(Y) && aligned16(b)) {
        const unsigned nb4 = (unsigned)std::min<int64_t>(4096, (n * (w / 4) + 255) / 256);
        switch (activation) {
            case GATMI_ACT_RELU: hipLaunchKernelGGL(bias_act_fwd4_kernel<GATMI_ACT_RELU>, dim3(nb4), dim3(256), 0, st, X, n, w / 4, ld, b, Y); break;
            case GATMI_ACT_ELU: hipLaunchKernelGGL(bias_act_fwd4_kernel<GATMI_ACT_ELU>, dim3(nb4), dim3(256), 0, st, X, n, w / 4, ld, b, Y); break;
            case GATMI_ACT_IDENTITY: hipLaunchKernelGGL(bias_act_fwd4_kernel<GATMI_ACT_IDENTITY>, dim3(nb4), dim3(256), 0, st, X, n, w / 4, ld, b, Y); break;
            default: return fail(GATMI_EUNSUPPORTED, "bias_act_fwd: activation %d (identity, elu or relu)", activation);
        }
        GATMI_LAUNCH_CHECK("bias_act_fwd");
        return GATMI_OK;
    }
    const unsigned nb = (unsigned)std::min<int64_t>(4096, (n * w + 255) / 256);
    switch (activation) {
        case GATMI_ACT_RELU: hipLaunchKernelGGL(bias_act_fwd_kernel<GATMI_ACT_RELU>, dim3(nb), dim3(256), 0, st, X, n, w, ld, b, Y); break;
        case GATMI_ACT_ELU: hipLaunchKernelGGL(bias_act_fwd_kernel<GATMI_ACT_ELU>, dim3(nb), dim3(256), 0, st, X, n, w, ld, b, Y); break;
        case GATMI_ACT_IDENTITY: hipLaunchKernelGGL(bias_act_fwd_kernel<GATMI_ACT_IDENTITY>, dim3(nb), dim3(256), 0, st, X, n, w, ld, b, Y); break;
        default: return fail(GATMI_EUNSUPPORTED, "bias_act_fwd: activation %d (identity, elu or relu)", activation);
    }
    GATMI_LAUNCH_CHECK("bias_act_fwd");
    return GATMI_OK;
}

extern "C" int gatmi_bias_act_bwd(const float *dY, const float *Y, int64_t n, int32_t w, int64_t ld, int32_t activation,
                                  float *dPre, float *db, float *ws, void *stream) {
    GATMI_REQUIRE(n >= 0 && w > 0 && ld >= w, GATMI_EINVAL, "bias_act_bwd: bad dims");
    GATMI_REQUIRE(dY && Y && dPre && db && ws, GATMI_EINVAL, "bias_act_bwd: null pointer");
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const unsigned nb = column_sum_blocks(n);
    const dim3 grid(nb, (unsigned)((w + 63) / 64));
    switch (activation) {
        case GATMI_ACT_RELU: hipLaunchKernelGGL(bias_act_bwd_kernel<GATMI_ACT_RELU>, grid, dim3(256), 0, st, dY, Y, n, w, ld, dPre, ws); break;
        case GATMI_ACT_ELU: hipLaunchKernelGGL(bias_act_bwd_kernel<GATMI_ACT_ELU>, grid, dim3(256), 0, st, dY, Y, n, w, ld, dPre, ws); break;
        case GATMI_ACT_IDENTITY: hipLaunchKernelGGL(bias_act_bwd_kernel<GATMI_ACT_IDENTITY>, grid, dim3(256), 0, st, dY, Y, n, w, ld, dPre, ws); break;
        default: return fail(GATMI_EUNSUPPORTED, "bias_act_bwd: activation %d (identity, elu or relu)", activation);
    }
    GATMI_LAUNCH_CHECK("bias_act_bwd");
    hipLaunchKernelGGL(column_total_kernel, dim3((unsigned)((w + kSumCols - 1) / kSumCols)), dim3(kSumCols, kSumRows), 0, st, ws, (int)nb, w, db);
    GATMI_LAUNCH_CHECK("column_total");
    return GATMI_OK;
}

extern "C" size_t gatmi_column_sums_ws_floats(int64_t n, int32_t w) {
    return n <= 0 || w <= 0 ? 1 : (size_t)column_sum_blocks(n) * (size_t)w;
}

extern "C" int gatmi_column_sums(const float *Y, int64_t n, int32_t w, int64_t ld, float *out, float *ws, void *stream) {
    GATMI_REQUIRE(n >= 0 && w > 0 && ld >= w, GATMI_EINVAL, "column_sums: bad dims");
    GATMI_REQUIRE(Y && out && ws, GATMI_EINVAL, "column_sums: null pointer");
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const unsigned nb = column_sum_blocks(n);
    hipLaunchKernelGGL(column_partial_kernel, dim3(nb, (unsigned)((w + 63) / 64)), dim3(256), 0, st, Y, n, w, ld, ws);
    GATMI_LAUNCH_CHECK("column_partial");
    hipLaunchKernelGGL(column_total_kernel, dim3((unsigned)((w + kSumCols - 1) / kSumCols)), dim3(kSumCols, kSumRows), 0, st, ws, (int)nb, w, out);
    GATMI_LAUNCH_CHECK("column_total");
    return GATMI_OK;
}

extern "C" int gatmi_proj_bwd_input(const float *dWh, int64_t n, int32_t f_in, const float *W,
                                    int32_t H, int32_t dp, float *dX, int64_t lddx, int accumulate,
                                    float in_keep, uint64_t seed, void *stream) {
    int G, LPH;
    int rc = check_shape(H, dp, &G, &LPH);
    if (rc) return rc;
    GATMI_REQUIRE(n >= 0 && f_in > 0 && lddx >= f_in, GATMI_EINVAL, "proj_bwd_input: bad dims");
    GATMI_REQUIRE(in_keep > 0.f && in_keep <= 1.f, GATMI_EINVAL, "proj_bwd_input: in_keep must be in (0,1]");
    GATMI_REQUIRE(in_keep >= 1.f || f_in % 4 == 0, GATMI_EUNSUPPORTED, "proj_bwd_input: input dropout inside the projection needs f_in %% 4 == 0 (f_in = %d)", f_in);
    if (n == 0) return GATMI_OK;
    GATMI_REQUIRE(dWh && W && dX, GATMI_EINVAL, "proj_bwd_input: null pointer");
    GATMI_REQUIRE_ALIGNED(dWh);
    const int hd = H * dp;
    GemmArgs g{};
    g.A = dWh; g.n = n; g.K = hd; g.lda = hd;
    g.B = W; g.sbk = 1; g.sbc = hd; g.Nc = f_in;   // B[k=c_hd][c=f] = W[f][c_hd]
    g.C = dX; g.ldc = lddx; g.accumulate = accumulate; g.cbias = nullptr;
    if (in_keep < 1.f) { g.c_keep = in_keep; g.drop_f4 = f_in / 4; g.seed = seed; g.seed_step = dropout_step_source(); }
    return launch_tall_gemm(g, reinterpret_cast<hipStream_t>(stream), "proj_bwd_input");
}

extern "C" int gatmi_csr_row_normalize(int64_t n, const int64_t *rowptr, const float *val, float *out, void *stream) {
    GATMI_REQUIRE(n >= 0, GATMI_EINVAL, "csr_row_normalize: negative size");
    if (n == 0) return GATMI_OK;
    GATMI_REQUIRE(rowptr && val && out, GATMI_EINVAL, "csr_row_normalize: null pointer");
    hipLaunchKernelGGL(csr_row_normalize_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0,
                       reinterpret_cast<hipStream_t>(stream), n, rowptr, val, out);
    GATMI_LAUNCH_CHECK("csr_row_normalize");
    return GATMI_OK;
}
