// l3c_common.h -- shared host-side plumbing of libl3c_hip.so (error reporting, launch checks).
#ifndef L3C_COMMON_H_
#define L3C_COMMON_H_

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/l3c_hip.h"

namespace l3c {

char *error_buffer();  // thread-local, defined in l3c_api.hip

inline int fail(int code, const char *fmt, const char *a = "", long long b = 0, long long c = 0) {
    snprintf(error_buffer(), 512, fmt, a, b, c);
    return code;
}

inline int check_hip(hipError_t e, const char *what) {
    if (e == hipSuccess) return L3C_OK;
    snprintf(error_buffer(), 512, "%s: %s", what, hipGetErrorString(e));
    return L3C_ERR_HIP;
}

inline int check_launch(const char *kernel) { return check_hip(hipGetLastError(), kernel); }

inline hipStream_t as_stream(l3c_stream_t s) { return reinterpret_cast<hipStream_t>(s); }

}  // namespace l3c

#define L3C_REQUIRE(cond, msg)                                                         \
    do {                                                                               \
        if (!(cond)) return l3c::fail(L3C_ERR_INVALID_ARG, "%s (" #cond ")", msg);     \
    } while (0)

#endif
