// Masked softmax cross-entropy with its gradient and the masked accuracy in one sweep over the logits
// (models/base_gattn.py:140-165 of the reference; the Cora drivers call them at execute_cora.py:74-75).
//
//   loss     = mean_i( CE_i * mask_i / mean(mask) ) = sum_i mask_i CE_i / sum_i mask_i,
//              CE_i = -sum_c y_ic log softmax(z_i)_c = lse_i * sum_c y_ic - sum_c y_ic z_ic
//   accuracy = sum_i mask_i [argmax_c z_ic == argmax_c y_ic] / sum_i mask_i
//   dloss/dz_ic = mask_i / sum(mask) * (softmax(z_i)_c * sum_c' y_ic' - y_ic)
//
// Written as torch operations over the [N, C] logits of a large graph this is a dozen elementwise passes
// (log-softmax, product with the labels, row sums, mask weighting, and their adjoints).  Here the mask is
// summed first (per-block partials, every block of the main kernel adds them in a fixed order), then ONE
// kernel reads each row of logits and labels once, G = ceil(C / 4) -> power of two lanes per row, and
// writes the row of the gradient and per-block partial sums of loss and hits; a last one-block kernel adds
// those.  All sums run in a fixed order: results are reproducible run to run.
#include <hip/hip_runtime.h>

#include <algorithm>

#include "common.h"
#include "device_common.h"

namespace gatmi {

constexpr int kCeThreads = 256;
constexpr int kCeMaskBlocks = 256;        // partial sums of the mask
constexpr int kCeMaxBlocks = 2048;        // blocks (and partial sums) of the main kernel

// fixed-order sum over the block: lanes by DPP / shuffles, the waves' totals in wave order
__device__ __forceinline__ float block_sum_fixed(float v, float *scratch) {
    v = group_sum<64>(v);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) scratch[wave] = v;
    __syncthreads();
    float t = 0.f;
    for (int w = 0; w < (int)(blockDim.x >> 6); ++w) t += scratch[w];
    return t;
}

__global__ __launch_bounds__(kCeThreads) void ce_mask_sum_kernel(const float *mask, int64_t n, float *partial) {
    __shared__ float scratch[kCeThreads / 64];
    float t = 0.f;
    for (int64_t i = blockIdx.x * (int64_t)kCeThreads + threadIdx.x; i < n; i += (int64_t)gridDim.x * kCeThreads) t += mask[i];
    t = block_sum_fixed(t, scratch);
    if (threadIdx.x == 0) partial[blockIdx.x] = t;
}

struct CeArgs {
    int64_t n;
    int C;
    bool vec;                      // rows of logits / dlogits are 16-byte aligned and hold all 4 G columns: float4 accesses
    const float *logits, *labels, *mask;
    int64_t ldz, ldy;
    float *dlogits;
    const float *mask_partial;     // kCeMaskBlocks partial sums, or nullptr when there is no mask (sum = n)
    float *partial;                // [gridDim.x][2]: loss, hits
};

template <int G>
__global__ __launch_bounds__(kCeThreads) void masked_ce_kernel(CeArgs a) {
    __shared__ float scratch[kCeThreads / 64];
    constexpr int RPB = kCeThreads / G;                // rows per block and pass
    const int sub = threadIdx.x % G, grp = threadIdx.x / G;
    float msum;
    if (a.mask_partial) {
        float t = 0.f;
        for (int k = threadIdx.x; k < kCeMaskBlocks; k += kCeThreads) t += a.mask_partial[k];
        msum = block_sum_fixed(t, scratch);
    } else {
        msum = (float)a.n;
    }
    const float inv_msum = msum > 0.f ? 1.0f / msum : 0.f;
    const int C = a.C;
    float loss_t = 0.f, hit_t = 0.f;
    for (int64_t row0 = blockIdx.x * (int64_t)RPB; row0 < a.n; row0 += (int64_t)gridDim.x * RPB) {
        const int64_t i = min(row0 + grp, a.n - 1);
        const bool row_ok = row0 + grp < a.n;
        const float *z = a.logits + i * a.ldz;
        const float *y = a.labels + i * a.ldy;
        float zv[4], yv[4];
        if (a.vec) {
            const float4 v = *reinterpret_cast<const float4 *>(z + 4 * sub);
            zv[0] = v.x; zv[1] = v.y; zv[2] = v.z; zv[3] = v.w;
        }
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const int col = 4 * sub + c;
            if (!a.vec) zv[c] = col < C ? z[col] : -INFINITY;
            else if (col >= C) zv[c] = -INFINITY;
            yv[c] = col < C ? y[col] : -INFINITY;      // -inf: never the arg-max, and excluded from the sums below
        }
        // row maximum and first arg-max of logits and of labels (ties: lowest column, as tf.argmax / torch.argmax)
        float zm = zv[0], ym = yv[0];
        int zi = 4 * sub, yi = 4 * sub;
#pragma unroll
        for (int c = 1; c < 4; ++c) {
            if (zv[c] > zm) { zm = zv[c]; zi = 4 * sub + c; }
            if (yv[c] > ym) { ym = yv[c]; yi = 4 * sub + c; }
        }
#pragma unroll
        for (int o = 1; o < G; o <<= 1) {
            const float zo = __shfl_xor(zm, o, 64), yo = __shfl_xor(ym, o, 64);
            const int zio = __shfl_xor(zi, o, 64), yio = __shfl_xor(yi, o, 64);
            if (zo > zm || (zo == zm && zio < zi)) { zm = zo; zi = zio; }
            if (yo > ym || (yo == ym && yio < yi)) { ym = yo; yi = yio; }
        }
        float se = 0.f, sy = 0.f, syz = 0.f;
        float pv[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const bool ok = 4 * sub + c < C;
            pv[c] = ok ? expf(zv[c] - zm) : 0.f;
            se += pv[c];
            sy += ok ? yv[c] : 0.f;
            syz += ok ? yv[c] * (zv[c] - zm) : 0.f;
        }
        se = group_sum<G>(se);
        sy = group_sum<G>(sy);
        syz = group_sum<G>(syz);
        const float m = a.mask ? a.mask[i] : 1.0f;
        const float w = row_ok ? m * inv_msum : 0.f;
        if (sub == 0) {
            loss_t += w * (logf(se) * sy - syz);       // CE_i with the row maximum taken out of both terms
            hit_t += w * (zi == yi ? 1.f : 0.f);
        }
        if (a.dlogits && row_ok) {
            float *d = a.dlogits + i * a.ldz;
            const float ps = w * sy / se;
            float dv[4];
#pragma unroll
            for (int c = 0; c < 4; ++c) dv[c] = 4 * sub + c < C ? pv[c] * ps - w * yv[c] : 0.f;      // padded columns: zero gradient
            if (a.vec) {
                store_stream4(d + 4 * sub, make_float4(dv[0], dv[1], dv[2], dv[3]));
            } else {
#pragma unroll
                for (int c = 0; c < 4; ++c)
                    if (4 * sub + c < a.ldz) d[4 * sub + c] = dv[c];
            }
        }
    }
    const float lb = block_sum_fixed(loss_t, scratch);
    const float hb = block_sum_fixed(hit_t, scratch);
    if (threadIdx.x == 0) {
        a.partial[2 * blockIdx.x] = lb;
        a.partial[2 * blockIdx.x + 1] = hb;
        if (blockIdx.x == 0) a.partial[2 * gridDim.x] = msum;
    }
}

// out = {loss, accuracy, sum(mask)}
__global__ __launch_bounds__(kCeThreads) void ce_final_kernel(const float *partial, int blocks, float *out) {
    __shared__ float scratch[kCeThreads / 64];
    float l = 0.f, h = 0.f;
    for (int k = threadIdx.x; k < blocks; k += kCeThreads) {
        l += partial[2 * k];
        h += partial[2 * k + 1];
    }
    l = block_sum_fixed(l, scratch);
    h = block_sum_fixed(h, scratch);
    if (threadIdx.x == 0) {
        out[0] = l;
        out[1] = h;
        out[2] = partial[2 * blocks];
    }
}

// x *= *scale unless *scale == 1 (the gradient of a loss that is the last node of the tape arrives as exactly 1:
// the sweep over the saved gradient is then skipped by every block after one scalar load)
__global__ __launch_bounds__(256) void scale_by_device_scalar_kernel(float4 *x, int64_t n4, int tail, const float *scale) {
    const float s = *scale;
    if (s == 1.0f) return;
    for (int64_t i = blockIdx.x * 256ll + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
        float4 v = x[i];
        v.x *= s; v.y *= s; v.z *= s; v.w *= s;
        x[i] = v;
    }
    if (blockIdx.x == 0 && (int)threadIdx.x < tail) reinterpret_cast<float *>(x + n4)[threadIdx.x] *= s;
}

}  // namespace gatmi

using namespace gatmi;

extern "C" size_t gatmi_masked_ce_ws_floats(void) { return (size_t)kCeMaskBlocks + 2 * (size_t)kCeMaxBlocks + 1; }

extern "C" int gatmi_masked_ce(int64_t n, int32_t C, const float *logits, int64_t ld_logits, const float *labels, int64_t ld_labels,
                               const float *mask, float *dlogits, float *out3, float *ws, void *stream) {
    GATMI_REQUIRE(n >= 0 && C >= 1, GATMI_EINVAL, "masked_ce: n = %lld, C = %d", (long long)n, C);
    GATMI_REQUIRE(C <= 256, GATMI_EUNSUPPORTED, "masked_ce: %d classes (at most 256: one wave per row)", C);
    GATMI_REQUIRE(out3 && ws, GATMI_EINVAL, "masked_ce: null pointer");
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    if (n == 0) {
        hipError_t e = hipMemsetAsync(out3, 0, 3 * sizeof(float), st);
        if (e != hipSuccess) return fail(GATMI_EHIP, "masked_ce: %s", hipGetErrorString(e));
        return GATMI_OK;
    }
    GATMI_REQUIRE(logits && labels, GATMI_EINVAL, "masked_ce: null pointer");
    GATMI_REQUIRE(ld_logits >= C && ld_labels >= C, GATMI_EINVAL, "masked_ce: row strides %lld / %lld below C = %d", (long long)ld_logits, (long long)ld_labels, C);
    float *mask_partial = ws, *partial = ws + kCeMaskBlocks;
    if (mask) {
        hipLaunchKernelGGL(ce_mask_sum_kernel, dim3(kCeMaskBlocks), dim3(kCeThreads), 0, st, mask, n, mask_partial);
        GATMI_LAUNCH_CHECK("masked_ce (mask sum)");
    }
    int G = 1;
    while (4 * G < C) G *= 2;
    GATMI_REQUIRE(!dlogits || ld_logits <= 4 * G, GATMI_EINVAL,
                  "masked_ce: the gradient rows have %lld columns, the kernel writes %d (C = %d classes and the zero padding up to the next multiple it covers)",
                  (long long)ld_logits, 4 * G, C);
    CeArgs a;
    a.n = n; a.C = C; a.logits = logits; a.labels = labels; a.mask = mask; a.ldz = ld_logits; a.ldy = ld_labels;
    a.dlogits = dlogits; a.mask_partial = mask ? mask_partial : nullptr; a.partial = partial;
    a.vec = ld_logits == 4 * G && aligned16(logits) && (!dlogits || aligned16(dlogits));
    const int rows_per_block = kCeThreads / G;
    const int blocks = (int)std::min<int64_t>(kCeMaxBlocks, (n + rows_per_block - 1) / rows_per_block);
#define X(g) if (G == g) hipLaunchKernelGGL((masked_ce_kernel<g>), dim3(blocks), dim3(kCeThreads), 0, st, a);
    X(1) X(2) X(4) X(8) X(16) X(32) X(64)
#undef X
    GATMI_LAUNCH_CHECK("masked_ce");
    hipLaunchKernelGGL(ce_final_kernel, dim3(1), dim3(kCeThreads), 0, st, partial, blocks, out3);
    GATMI_LAUNCH_CHECK("masked_ce (final sums)");
    return GATMI_OK;
}

extern "C" int gatmi_scale_by_device_scalar(float *x, int64_t n, const float *scale, void *stream) {
    GATMI_REQUIRE(n >= 0, GATMI_EINVAL, "scale_by_device_scalar: n = %lld", (long long)n);
    if (n == 0) return GATMI_OK;
    GATMI_REQUIRE(x && scale, GATMI_EINVAL, "scale_by_device_scalar: null pointer");
    GATMI_REQUIRE_ALIGNED(x);
    const int64_t n4 = n / 4;
    const unsigned blocks = (unsigned)std::max<int64_t>(1, std::min<int64_t>(4096, (n4 + 255) / 256));
    hipLaunchKernelGGL(scale_by_device_scalar_kernel, dim3(blocks), dim3(256), 0, reinterpret_cast<hipStream_t>(stream),
                       reinterpret_cast<float4 *>(x), n4, (int)(n - 4 * n4), scale);
    GATMI_LAUNCH_CHECK("scale_by_device_scalar");
    return GATMI_OK;
}
