// common.h -- error plumbing and launch helpers shared by the HIP translation units.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdarg>
#include <cstdio>
#include "../../include/cine_hip.h"

namespace cine {

void set_error(const char* fmt, ...);   // api.cpp (thread-local buffer)

inline hipStream_t as_stream(void* s) { return reinterpret_cast<hipStream_t>(s); }

inline int check_launch(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        set_error("%s: %s", what, hipGetErrorString(e));
        return CINE_EHIP;
    }
    return CINE_OK;
}

#define CINE_REQUIRE(cond, code, ...)            \
    do {                                         \
        if (!(cond)) {                           \
            ::cine::set_error(__VA_ARGS__);      \
            return (code);                       \
        }                                        \
    } while (0)

template <typename T> inline T ceil_div(T a, T b) { return (a + b - 1) / b; }

// Optional per-kernel-family timing (cine_profile_begin / cine_profile_end): when enabled every
// launch is bracketed by a hipEvent pair on ITS stream.  Disabled (the default, and always during
// graph capture) it costs one relaxed load.
enum Family { F_FFT_COL = 0, F_FFT_ROW, F_CONV3, F_STATS, F_TCONV, F_CONV1, F_PACK, F_MISC, F_COUNT };
struct ProfScope {
    int fam; hipStream_t st; hipEvent_t e0, e1; bool on;
    ProfScope(int family, hipStream_t stream);
    ~ProfScope();
};

}  // namespace cine
