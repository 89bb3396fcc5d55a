// common.h -- error plumbing and launch helpers shared by the HIP translation units.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdarg>
#include <cstdio>
#include "../../include/cine_hip.h"

// Diagnostic builds only (tools/*_stamps.hip define CINE_STAMPS): per-phase s_memtime stamps
// of workgroup-lane 0 into a side buffer.  In the product build the macro is empty.
#ifdef CINE_STAMPS
__device__ unsigned long long g_cine_stamps[1 << 20];
#define CINE_STAMP(slot)                                                                          \
    do {                                                                                          \
        __builtin_amdgcn_sched_barrier(0);                                                        \
        if (threadIdx.x == 0) {                                                                   \
            unsigned long long t_;                                                                \
            asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");             \
            g_cine_stamps[(((blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x) % (1 << 16)) * 16 + (slot)] = t_; \
        }                                                                                         \
        __builtin_amdgcn_sched_barrier(0);                                                        \
    } while (0)
// wall-clock (100 MHz s_memrealtime) stamp + hardware id (XCC, SE, CU) of the workgroup
#define CINE_STAMP_RT(slot)                                                                       \
    do {                                                                                          \
        __builtin_amdgcn_sched_barrier(0);                                                        \
        if (threadIdx.x == 0) {                                                                   \
            unsigned long long t_ = __builtin_amdgcn_s_memrealtime();                             \
            unsigned hw_ = __builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11));            \
            unsigned xcc_ = __builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (3 << 11));           \
            unsigned long long* p_ = &g_cine_stamps[(((blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x) % (1 << 16)) * 16]; \
            p_[(slot)] = t_; p_[15] = ((unsigned long long)xcc_ << 32) | hw_;                      \
        }                                                                                         \
        __builtin_amdgcn_sched_barrier(0);                                                        \
    } while (0)
#else
#define CINE_STAMP(slot) do { } while (0)
#define CINE_STAMP_RT(slot) do { } while (0)
#endif

namespace cine {

void set_error(const char* fmt, ...);   // api.cpp (thread-local buffer)
// Process-wide launch counters of the diagnostic kernel choices (cine_diag_counter): which of two bit-identical kernels a launch took.
enum Diag { D_WGRAD_PLANE = 0, D_WGRAD_GENERAL, D_UNET_BRANCHED, D_CRNN_SWEEP_C, D_COUNT };
void diag_count(int which);

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
