// ew_kernels.hip -- the reference's small tensor helpers as device kernels:
//   utils/math.py:20-44    complex_mul (with broadcasting), complex_conj, complex_abs_sq
//   utils/coil_combine.py  rss, rss_complex (root-sum-of-squares over one dimension)
//   utils/fftc.py:141-213  roll (fftshift / ifftshift are rolls by n / 2, (n + 1) / 2)
//   utils/padding.py:22-47 zero padding of the last two dimensions (pad_for_mwcnn)
// The fused path never calls them (sens-multiply, conjugate, magnitude, shifts and pads live inside the FFT / conv / pack
// kernels); they exist so that user code written against the reference's utils keeps running on the GPU.  The arithmetic uses
// the unfused IEEE operations of the reference's tensor expressions (__fmul_rn / __fadd_rn: no FMA contraction).
#include "common.h"

namespace cine {
namespace {

struct Bcast { int nd; int shape[6]; long xs[6], ys[6]; };     // strides in complex elements, 0 on broadcast dimensions

__global__ __launch_bounds__(256) void complex_mul_kernel(const float2* __restrict__ x, const float2* __restrict__ y, float2* __restrict__ out, Bcast b, long n) {
    for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < n; e += (long)gridDim.x * 256) {
        long r = e, xo = 0, yo = 0;
#pragma unroll
        for (int d = 5; d >= 0; --d) {
            if (d >= b.nd) continue;
            const long q = r / b.shape[d], i = r - q * b.shape[d];
            xo += i * b.xs[d]; yo += i * b.ys[d]; r = q;
        }
        const float2 a = x[xo], c = y[yo];
        out[e] = make_float2(__fsub_rn(__fmul_rn(a.x, c.x), __fmul_rn(a.y, c.y)), __fadd_rn(__fmul_rn(a.x, c.y), __fmul_rn(a.y, c.x)));
    }
}
__global__ __launch_bounds__(256) void complex_conj_kernel(const float2* __restrict__ x, float2* __restrict__ out, long n) {
    for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < n; e += (long)gridDim.x * 256) { const float2 a = x[e]; out[e] = make_float2(a.x, -a.y); }
}
__global__ __launch_bounds__(256) void complex_abs_sq_kernel(const float2* __restrict__ x, float* __restrict__ out, long n) {
    for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < n; e += (long)gridDim.x * 256) {
        const float2 a = x[e];
        out[e] = __fadd_rn(__fmul_rn(a.x, a.x), __fmul_rn(a.y, a.y));
    }
}
// out[o][i] = sqrt(sum_k v(x[o][k][i])), v = square (real data) or re^2 + im^2 (complex pairs); k ascending
template <bool CPLX>
__global__ __launch_bounds__(256) void rss_kernel(const float* __restrict__ x, float* __restrict__ out, long outer, int k, long inner) {
    const long n = outer * inner;
    for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < n; e += (long)gridDim.x * 256) {
        const long o = e / inner, i = e - o * inner;
        float s = 0.f;
        for (int j = 0; j < k; ++j) {
            if (CPLX) { const float2 a = reinterpret_cast<const float2*>(x)[(o * k + j) * inner + i]; s = __fadd_rn(s, __fadd_rn(__fmul_rn(a.x, a.x), __fmul_rn(a.y, a.y))); }
            else { const float a = x[(o * k + j) * inner + i]; s = __fadd_rn(s, __fmul_rn(a, a)); }
        }
        out[e] = sqrtf(s);                       // (correctly rounded: hipcc -fhip-fp32-correctly-rounded-divide-sqrt is the default)
    }
}
// out[o][(j + shift) mod n][i] = x[o][j][i]
__global__ __launch_bounds__(256) void roll_kernel(const float* __restrict__ x, float* __restrict__ out, long outer, int n, long inner, int shift) {
    const long tot = outer * n * inner;
    for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < tot; e += (long)gridDim.x * 256) {
        const long i = e % inner, r = e / inner;
        const int j = (int)(r % n);
        const long o = r / n;
        int src = j - shift; src %= n; if (src < 0) src += n;
        out[e] = x[(o * n + src) * inner + i];
    }
}
__global__ __launch_bounds__(256) void pad2d_kernel(const float* __restrict__ x, float* __restrict__ out, long planes, int h, int w, int top, int left, int hp, int wp) {
    const long tot = planes * hp * wp;
    for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < tot; e += (long)gridDim.x * 256) {
        const int xx = (int)(e % wp); const long r = e / wp; const int yy = (int)(r % hp); const long p = r / hp;
        const int sy = yy - top, sx = xx - left;
        out[e] = (sy >= 0 && sy < h && sx >= 0 && sx < w) ? x[(p * h + sy) * w + sx] : 0.f;
    }
}
unsigned grid_for(long n) { long g = (n + 255) / 256; return (unsigned)(g < 1 ? 1 : (g > 16384 ? 16384 : g)); }

}  // namespace
}  // namespace cine

using namespace cine;

extern "C" int cine_complex_mul(const float* x, const float* y, float* out, int ndim, const int* shape, const long* xstride, const long* ystride, void* stream) {
    CINE_REQUIRE(x && y && out && shape && xstride && ystride && ndim >= 0 && ndim <= 6, CINE_EINVAL, "cine_complex_mul: bad arguments (at most 6 dimensions besides the complex pair)");
    Bcast b{}; b.nd = ndim; long n = 1;
    for (int d = 0; d < 6; ++d) { b.shape[d] = d < ndim ? shape[d] : 1; b.xs[d] = d < ndim ? xstride[d] : 0; b.ys[d] = d < ndim ? ystride[d] : 0; }
    for (int d = 0; d < ndim; ++d) { CINE_REQUIRE(shape[d] > 0, CINE_EINVAL, "cine_complex_mul: empty dimension"); n *= shape[d]; }
    ProfScope prof(F_MISC, as_stream(stream));
    hipLaunchKernelGGL(complex_mul_kernel, dim3(grid_for(n)), dim3(256), 0, as_stream(stream), reinterpret_cast<const float2*>(x), reinterpret_cast<const float2*>(y),
                       reinterpret_cast<float2*>(out), b, n);
    return check_launch("complex_mul_kernel");
}
extern "C" int cine_complex_conj(const float* x, float* out, long n, void* stream) {
    CINE_REQUIRE(x && out && n > 0, CINE_EINVAL, "cine_complex_conj: bad arguments");
    ProfScope prof(F_MISC, as_stream(stream));
    hipLaunchKernelGGL(complex_conj_kernel, dim3(grid_for(n)), dim3(256), 0, as_stream(stream), reinterpret_cast<const float2*>(x), reinterpret_cast<float2*>(out), n);
    return check_launch("complex_conj_kernel");
}
extern "C" int cine_complex_abs_sq(const float* x, float* out, long n, void* stream) {
    CINE_REQUIRE(x && out && n > 0, CINE_EINVAL, "cine_complex_abs_sq: bad arguments");
    ProfScope prof(F_MISC, as_stream(stream));
    hipLaunchKernelGGL(complex_abs_sq_kernel, dim3(grid_for(n)), dim3(256), 0, as_stream(stream), reinterpret_cast<const float2*>(x), out, n);
    return check_launch("complex_abs_sq_kernel");
}
extern "C" int cine_rss(const float* x, float* out, long outer, int k, long inner, int is_complex, void* stream) {
    CINE_REQUIRE(x && out && outer > 0 && k > 0 && inner > 0, CINE_EINVAL, "cine_rss: bad arguments");
    ProfScope prof(F_MISC, as_stream(stream));
    if (is_complex) hipLaunchKernelGGL(rss_kernel<true>, dim3(grid_for(outer * inner)), dim3(256), 0, as_stream(stream), x, out, outer, k, inner);
    else hipLaunchKernelGGL(rss_kernel<false>, dim3(grid_for(outer * inner)), dim3(256), 0, as_stream(stream), x, out, outer, k, inner);
    return check_launch("rss_kernel");
}
extern "C" int cine_roll(const float* x, float* out, long outer, int n, long inner, int shift, void* stream) {
    CINE_REQUIRE(x && out && x != out && outer > 0 && n > 0 && inner > 0, CINE_EINVAL, "cine_roll: bad arguments (out of place only)");
    ProfScope prof(F_MISC, as_stream(stream));
    hipLaunchKernelGGL(roll_kernel, dim3(grid_for(outer * n * inner)), dim3(256), 0, as_stream(stream), x, out, outer, n, inner, shift);
    return check_launch("roll_kernel");
}
extern "C" int cine_pad2d(const float* x, float* out, long planes, int h, int w, int top, int left, int hp, int wp, void* stream) {
    CINE_REQUIRE(x && out && planes > 0 && h > 0 && w > 0 && top >= 0 && left >= 0 && hp >= h + top && wp >= w + left, CINE_EINVAL, "cine_pad2d: bad arguments");
    ProfScope prof(F_MISC, as_stream(stream));
    hipLaunchKernelGGL(pad2d_kernel, dim3(grid_for(planes * hp * wp)), dim3(256), 0, as_stream(stream), x, out, planes, h, w, top, left, hp, wp);
    return check_launch("pad2d_kernel");
}


// ---------------------------------------------------------------- diagnostics: a kernel of known duration on one workgroup
// cine_spin(us): one wave reads the constant 100 MHz clock until `us` microseconds have passed (or an iteration cap is reached: it always
// terminates).  The binding uses it to find out whether two streams really run side by side: the runtime maps streams onto a limited number
// of hardware queues, and two streams on one queue execute one after the other whatever the events between them say.
namespace cine {
__global__ void spin_kernel(unsigned long long ticks, unsigned long long* out) {
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    unsigned long long t = t0;
    for (int i = 0; i < (1 << 22) && t - t0 < ticks; ++i) { __builtin_amdgcn_s_sleep(8); t = __builtin_amdgcn_s_memrealtime(); }
    if (out && threadIdx.x == 0) *out = t - t0;
}
}  // namespace cine
extern "C" int cine_spin(int microseconds, void* stream) {
    CINE_REQUIRE(microseconds > 0 && microseconds <= 100000, CINE_EINVAL, "cine_spin: 1 .. 100000 us");
    hipLaunchKernelGGL(cine::spin_kernel, dim3(1), dim3(64), 0, cine::as_stream(stream), 100ull * (unsigned long long)microseconds,
                       static_cast<unsigned long long*>(nullptr));
    return cine::check_launch("spin_kernel");
}
