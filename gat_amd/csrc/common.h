// Shared host-side helpers of libgatmi: error reporting and argument checks.
#pragma once
#include <cstdarg>
#include <cstdint>
#include <cstdio>

#include "../../include/gatmi.h"

namespace gatmi {

// Thread-local message of the last failure; returned by gatmi_last_error().
char *error_buffer();
int fail(int code, const char *fmt, ...) __attribute__((format(printf, 2, 3)));

// The device word registered for the current device with gatmi_dropout_step_source, or nullptr.
// Entry points with dropout switched on hand it to their kernels (device_common.h: seed_now).
const uint64_t *dropout_step_source();

inline bool is_pow2(int64_t v) { return v > 0 && (v & (v - 1)) == 0; }

// (H, dp) -> lanes per row G = H*dp/4 and lanes per head LPH = dp/4; both powers of two.
inline int check_shape(int32_t H, int32_t dp, int *G, int *LPH) {
    if (H <= 0 || dp < 4 || !is_pow2(dp)) return fail(GATMI_EUNSUPPORTED, "dp=%d must be a power of two >= 4 (H=%d)", dp, H);
    int64_t hd = (int64_t)H * dp;
    if (!is_pow2(H) || hd / 4 > 64) return fail(GATMI_EUNSUPPORTED, "H=%d dp=%d: H must be a power of two and H*dp <= 256", H, dp);
    *G = (int)(hd / 4);
    *LPH = dp / 4;
    return GATMI_OK;
}

inline bool aligned16(const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

}  // namespace gatmi

#define GATMI_REQUIRE(cond, code, ...)                  \
    do {                                                \
        if (!(cond)) return gatmi::fail(code, __VA_ARGS__); \
    } while (0)

#define GATMI_REQUIRE_ALIGNED(p)                                                            \
    do {                                                                                    \
        if ((p) != nullptr && !gatmi::aligned16(p))                                         \
            return gatmi::fail(GATMI_EALIGN, "%s is not 16-byte aligned", #p);              \
    } while (0)
