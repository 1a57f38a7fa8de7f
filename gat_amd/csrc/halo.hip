// Halo exchange helpers for the row-partitioned multi-GPU layer: gather the feature rows a peer
// asked for into a contiguous send buffer, and add returned gradient rows into their owners.
// The exchange itself (all-to-all over RCCL / xGMI) is issued by the caller between the two.
#include <hip/hip_runtime.h>

#include <algorithm>

#include "common.h"
#include "device_common.h"

namespace gatmi {

// rows of `width` floats (multiple of 4): each float4 of a row is one lane's load + store
__global__ __launch_bounds__(256) void halo_pack_kernel(const float *src, const int64_t *idx, int64_t n_idx, int w4,
                                                         float *dst) {
    const int64_t total = n_idx * w4;
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
        const int64_t k = e / w4;
        const int c = (int)(e % w4);
        reinterpret_cast<float4 *>(dst)[e] = reinterpret_cast<const float4 *>(src)[idx[k] * w4 + c];
    }
}

__global__ __launch_bounds__(256) void halo_unpack_add_kernel(const float *src, const int64_t *idx, int64_t n_idx,
                                                               int w4, float *dst) {
    const int64_t total = n_idx * w4;
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
        const int64_t k = e / w4;
        const int c = (int)(e % w4);
        float4 *d = reinterpret_cast<float4 *>(dst) + idx[k] * w4 + c;
        const float4 s = reinterpret_cast<const float4 *>(src)[e];
        float4 v = *d;
        v.x += s.x; v.y += s.y; v.z += s.z; v.w += s.w;
        *d = v;
    }
}

// any width: one float per lane (narrow rows such as f2 [n, H] with H < 4)
__global__ __launch_bounds__(256) void halo_pack_scalar_kernel(const float *src, const int64_t *idx, int64_t n_idx,
                                                                int w, float *dst) {
    const int64_t total = n_idx * w;
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256)
        dst[e] = src[idx[e / w] * w + e % w];
}

__global__ __launch_bounds__(256) void halo_unpack_add_scalar_kernel(const float *src, const int64_t *idx,
                                                                      int64_t n_idx, int w, float *dst) {
    const int64_t total = n_idx * w;
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256)
        dst[idx[e / w] * w + e % w] += src[e];
}

// dst[rows[u], :] += sum over k in [ptr[u], ptr[u+1]) of src[pos[k], :], the terms added in list order:
// every target row belongs to exactly one thread group, so rows that several peers return meet
// without atomics and in a fixed order (bitwise reproducible).
__global__ __launch_bounds__(256) void halo_unpack_sum_kernel(const float *src, const int64_t *rows, const int64_t *ptr,
                                                               const int64_t *pos, int64_t n_rows, int w4, float *dst) {
    const int64_t total = n_rows * w4;
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
        const int64_t u = e / w4;
        const int c = (int)(e % w4);
        float4 *d = reinterpret_cast<float4 *>(dst) + rows[u] * w4 + c;
        float4 v = *d;
        for (int64_t k = ptr[u]; k < ptr[u + 1]; ++k) {
            const float4 t = reinterpret_cast<const float4 *>(src)[pos[k] * w4 + c];
            v.x += t.x; v.y += t.y; v.z += t.z; v.w += t.w;
        }
        *d = v;
    }
}

__global__ __launch_bounds__(256) void halo_unpack_sum_scalar_kernel(const float *src, const int64_t *rows, const int64_t *ptr,
                                                                      const int64_t *pos, int64_t n_rows, int w, float *dst) {
    const int64_t total = n_rows * w;
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
        const int64_t u = e / w;
        const int c = (int)(e % w);
        float v = dst[rows[u] * w + c];
        for (int64_t k = ptr[u]; k < ptr[u + 1]; ++k) v += src[pos[k] * w + c];
        dst[rows[u] * w + c] = v;
    }
}

}  // namespace gatmi

using namespace gatmi;

static int halo_common(const float *src, const int64_t *idx, int64_t n_idx, int32_t width, float *dst, const char *name) {
    GATMI_REQUIRE(n_idx >= 0 && width > 0, GATMI_EINVAL, "%s: width=%d must be positive", name, width);
    GATMI_REQUIRE(n_idx == 0 || (src && idx && dst), GATMI_EINVAL, "%s: null pointer", name);
    return GATMI_OK;
}

extern "C" int gatmi_halo_pack(const float *src, const int64_t *idx, int64_t n_idx, int32_t width, float *dst, void *stream) {
    int rc = halo_common(src, idx, n_idx, width, dst, "halo_pack");
    if (rc || n_idx == 0) return rc;
    const bool vec = width % 4 == 0 && aligned16(src) && aligned16(dst);
    const int64_t total = vec ? n_idx * (width / 4) : n_idx * width;
    const unsigned nb = (unsigned)std::min<int64_t>(4096, (total + 255) / 256);
    if (vec) hipLaunchKernelGGL(halo_pack_kernel, dim3(nb), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), src, idx, n_idx, width / 4, dst);
    else hipLaunchKernelGGL(halo_pack_scalar_kernel, dim3(nb), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), src, idx, n_idx, width, dst);
    GATMI_LAUNCH_CHECK("halo_pack");
    return GATMI_OK;
}

extern "C" int gatmi_halo_unpack_add(const float *src, const int64_t *idx, int64_t n_idx, int32_t width, float *dst, void *stream) {
    int rc = halo_common(src, idx, n_idx, width, dst, "halo_unpack_add");
    if (rc || n_idx == 0) return rc;
    const bool vec = width % 4 == 0 && aligned16(src) && aligned16(dst);
    const int64_t total = vec ? n_idx * (width / 4) : n_idx * width;
    const unsigned nb = (unsigned)std::min<int64_t>(4096, (total + 255) / 256);
    if (vec) hipLaunchKernelGGL(halo_unpack_add_kernel, dim3(nb), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), src, idx, n_idx, width / 4, dst);
    else hipLaunchKernelGGL(halo_unpack_add_scalar_kernel, dim3(nb), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), src, idx, n_idx, width, dst);
    GATMI_LAUNCH_CHECK("halo_unpack_add");
    return GATMI_OK;
}

extern "C" int gatmi_halo_unpack_sum(const float *src, const int64_t *rows, const int64_t *ptr, const int64_t *pos,
                                     int64_t n_rows, int32_t width, float *dst, void *stream) {
    GATMI_REQUIRE(n_rows >= 0 && width > 0, GATMI_EINVAL, "halo_unpack_sum: width=%d must be positive", width);
    if (n_rows == 0) return GATMI_OK;
    GATMI_REQUIRE(src && rows && ptr && pos && dst, GATMI_EINVAL, "halo_unpack_sum: null pointer");
    const bool vec = width % 4 == 0 && aligned16(src) && aligned16(dst);
    const int64_t total = vec ? n_rows * (width / 4) : n_rows * width;
    const unsigned nb = (unsigned)std::min<int64_t>(4096, (total + 255) / 256);
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    if (vec) hipLaunchKernelGGL(halo_unpack_sum_kernel, dim3(nb), dim3(256), 0, st, src, rows, ptr, pos, n_rows, width / 4, dst);
    else hipLaunchKernelGGL(halo_unpack_sum_scalar_kernel, dim3(nb), dim3(256), 0, st, src, rows, ptr, pos, n_rows, width, dst);
    GATMI_LAUNCH_CHECK("halo_unpack_sum");
    return GATMI_OK;
}
