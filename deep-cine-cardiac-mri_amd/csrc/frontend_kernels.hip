// frontend_kernels.hip -- the steps in FRONT of the hot path on the device.
//
//   data front-end   reference data/mri_data.py:283-303: raw k-space -> centered IFFT2 (cine_fft2c) -> crop + frame selection
//                    (crop_select_kernel) -> Gaussian filter, one pass per axis (gauss_axis_kernel: scipy.ndimage's 'reflect'
//                    boundary, truncate 4, float64 accumulation, float32 store -- data/transforms.py:186-220) -> centered FFT2;
//                    coil-combined magnitude target with center crop (combine_target_kernel, mri_data.py:302-303)
//   ESPIRiT          stands in for `bart ecalib` (mri_data.py:296, data/transforms.py:429): the c x c image-space operator
//                    M(r) comes from (2k-1)^2 lag kernels of the row-space projector (espirit_lag_kernel) through ONE batch of
//                    c^2 inverse FFTs; espirit_eig_kernel runs a power iteration per pixel (M(r) streamed from memory,
//                    coalesced over pixels, the vector in registers), normalises, references the phase to coil 0 and crops by
//                    the eigenvalue.  The small dense steps (Gram matrix of the calibration patches, its Hermitian
//                    eigen-decomposition) are library calls made by the host glue (cine_hip/frontend.py).
// All kernels are memory / latency bound pre-processing (once per slice, before the cascades).
#include "common.h"

namespace cine {

constexpr int kMaxTaps = 33;             // Gaussian radius <= 16  (sigma <= 4 at truncate 4)
struct GaussW { double w[kMaxTaps]; int r; };

// (t_in, c, hin, win) complex -> (t_out, c, hout, wout): first t_out frames, centered crop (transforms.py:209-214)
__global__ void crop_select_kernel(const float2* __restrict__ in, float2* __restrict__ out, long total,
                                   int c, int hin, int win, int hout, int wout, int y0, int x0) {
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
        const int x = (int)(e % wout);
        long r = e / wout;
        const int y = (int)(r % hout);
        const long tc = r / hout;                         // frame * c + coil (same in both arrays)
        out[e] = in[(tc * hin + y0 + y) * win + x0 + x];
    }
}

__device__ __forceinline__ int reflect_index(int i, int n) {     // scipy 'reflect': d c b a | a b c d | d c b a
    const int p = 2 * n;
    i %= p;
    if (i < 0) i += p;
    return i >= n ? p - 1 - i : i;
}

// one axis pass over a (outer, n, inner) complex array: out[o, i, j] = sum_k w[k] in[o, reflect(i + k), j]
__global__ void gauss_axis_kernel(const float2* __restrict__ in, float2* __restrict__ out, long total, int n, long inner, GaussW g) {
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
        const long j = e % inner;
        const long oi = e / inner;
        const int i = (int)(oi % n);
        const long base = (oi - i) * inner + j;
        double re = 0.0, im = 0.0;
        for (int k = -g.r; k <= g.r; ++k) {
            const float2 v = in[base + (long)reflect_index(i + k, n) * inner];
            re += g.w[k + g.r] * (double)v.x;
            im += g.w[k + g.r] * (double)v.y;
        }
        out[e] = make_float2((float)re, (float)im);
    }
}

// target[t, y, x] = | sum_c img[t, c, y0 + y, x0 + x] * conj(sens[c, y0 + y, x0 + x]) |
__global__ void combine_target_kernel(const float2* __restrict__ img, const float2* __restrict__ sens, float* __restrict__ out,
                                      long total, int c, int h, int w, int ch, int cw, int y0, int x0) {
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
        const int x = (int)(e % cw);
        long r = e / cw;
        const int y = (int)(r % ch);
        const long t = r / ch;
        const long pix = (long)(y0 + y) * w + x0 + x;
        float re = 0.f, im = 0.f;
        for (int k = 0; k < c; ++k) {
            const float2 a = img[(t * c + k) * h * w + pix], s = sens[(long)k * h * w + pix];
            re += a.x * s.x + a.y * s.y;
            im += a.y * s.x - a.x * s.y;
        }
        out[e] = sqrtf(re * re + im * im);
    }
}

// Lag kernels of the row-space projector W (kk*kk*c square, row / column index (py, px, coil)):
//   K[c][d](ly, lx) = scale * sum_{p - q = l} conj(W[(p, c), (q, d)]),   l in [-(kk-1), kk-1]^2
// written into the zero-padded centered array kpad (c*c, ny, nx) at (ny/2 + ly, nx/2 + lx), so that
// ifft2c(kpad)[c][d](r) = M_cd(r) when scale = sqrt(ny nx) / kk^2.  One thread per (c, d, lag); kpad is zeroed by the caller.
__global__ void espirit_lag_kernel(const float2* __restrict__ w, float2* __restrict__ kpad, int c, int kk, int ny, int nx, float scale) {
    const int nl = 2 * kk - 1;
    const long total = (long)c * c * nl * nl;
    const int dim = kk * kk * c;
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
        const int lx = (int)(e % nl) - (kk - 1);
        long r = e / nl;
        const int ly = (int)(r % nl) - (kk - 1);
        r /= nl;
        const int d = (int)(r % c), cc = (int)(r / c);
        float re = 0.f, im = 0.f;
        for (int qy = 0; qy < kk; ++qy) {
            const int py = qy + ly;
            if (py < 0 || py >= kk) continue;
            for (int qx = 0; qx < kk; ++qx) {
                const int px = qx + lx;
                if (px < 0 || px >= kk) continue;
                const float2 v = w[(long)((py * kk + px) * c + cc) * dim + (qy * kk + qx) * c + d];
                re += v.x; im -= v.y;
            }
        }
        const int y = ny / 2 + ly, x = nx / 2 + lx;
        if (y >= 0 && y < ny && x >= 0 && x < nx)
            kpad[((long)(cc * c + d) * ny + y) * nx + x] = make_float2(scale * re, scale * im);
    }
}

// Dominant eigenpair of the Hermitian c x c matrix M(r) of every pixel by power iteration; m is (c*c, npix) complex (plane
// (c, d) = row c, column d).  maps (c, npix): unit-norm eigenvector with coil 0 real and non-negative, zeroed where the
// eigenvalue is below `crop`; lam (npix) = Rayleigh quotient.
template <int CMAX>
__global__ __launch_bounds__(64) void espirit_eig_kernel(const float2* __restrict__ m, float2* __restrict__ maps, float* __restrict__ lam,
                                                         int c, long npix, int iters, float crop) {
    const long p = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= npix) return;
    float2 v[CMAX], u[CMAX];
    const float v0 = rsqrtf((float)c);
#pragma unroll
    for (int i = 0; i < CMAX; ++i) v[i] = make_float2(i < c ? v0 : 0.f, 0.f);
    float ev = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < CMAX; ++i) {
            float re = 0.f, im = 0.f;
            if (i < c) {
                const float2* row = m + (long)i * c * npix + p;
#pragma unroll
                for (int j = 0; j < CMAX; ++j) {
                    if (j < c) {
                        const float2 a = row[(long)j * npix];
                        re += a.x * v[j].x - a.y * v[j].y;
                        im += a.x * v[j].y + a.y * v[j].x;
                    }
                }
            }
            u[i] = make_float2(re, im);
        }
        float nn = 0.f, rq = 0.f;
#pragma unroll
        for (int i = 0; i < CMAX; ++i) { nn += u[i].x * u[i].x + u[i].y * u[i].y; rq += u[i].x * v[i].x + u[i].y * v[i].y; }
        ev = rq;                                           // v^H M v with |v| = 1
        const float inv = nn > 0.f ? rsqrtf(nn) : 0.f;
#pragma unroll
        for (int i = 0; i < CMAX; ++i) v[i] = make_float2(u[i].x * inv, u[i].y * inv);
    }
    // phase reference: coil 0 real, non-negative
    const float a0 = sqrtf(v[0].x * v[0].x + v[0].y * v[0].y);
    const float cr = a0 > 0.f ? v[0].x / a0 : 1.f, ci = a0 > 0.f ? -v[0].y / a0 : 0.f;
    const bool keep = ev >= crop;
#pragma unroll
    for (int i = 0; i < CMAX; ++i)
        if (i < c) maps[(long)i * npix + p] = keep ? make_float2(v[i].x * cr - v[i].y * ci, v[i].x * ci + v[i].y * cr) : make_float2(0.f, 0.f);
    lam[p] = ev;
}

static unsigned grid_for(long n, int threads) {
    long g = ceil_div(n, (long)threads);
    return (unsigned)(g > 16384 ? 16384 : (g < 1 ? 1 : g));
}

}  // namespace cine

using namespace cine;

extern "C" int cine_crop_select(const float* in, float* out, int t_in, int c, int hin, int win, int t_out, int hout, int wout, void* stream) {
    CINE_REQUIRE(in && out && in != out, CINE_EINVAL, "cine_crop_select: null or aliased pointers");
    CINE_REQUIRE(t_out > 0 && t_out <= t_in && c > 0 && hout > 0 && hout <= hin && wout > 0 && wout <= win, CINE_EINVAL,
                 "cine_crop_select: Invalid shapes.");
    const long total = (long)t_out * c * hout * wout;
    ProfScope prof(F_PACK, as_stream(stream));
    hipLaunchKernelGGL(crop_select_kernel, dim3(grid_for(total, 256)), dim3(256), 0, as_stream(stream),
                       reinterpret_cast<const float2*>(in), reinterpret_cast<float2*>(out), total, c, hin, win, hout, wout,
                       (hin - hout) / 2, (win - wout) / 2);
    return check_launch("crop_select_kernel");
}

extern "C" int cine_gauss_axis(const float* in, float* out, long outer, int n, long inner, double sigma, void* stream) {
    CINE_REQUIRE(in && out && in != out, CINE_EINVAL, "cine_gauss_axis: null or aliased pointers");
    CINE_REQUIRE(outer > 0 && n > 0 && inner > 0 && sigma > 1e-15, CINE_EINVAL, "cine_gauss_axis: bad sizes (sigma must be > 0)");
    GaussW g{};
    g.r = (int)(4.0 * sigma + 0.5);                         // scipy: int(truncate * sd + 0.5), truncate = 4
    CINE_REQUIRE(2 * g.r + 1 <= kMaxTaps, CINE_EUNSUPPORTED, "cine_gauss_axis: sigma %g needs %d taps (max %d)", sigma, 2 * g.r + 1, kMaxTaps);
    double sum = 0.0;
    for (int k = -g.r; k <= g.r; ++k) { g.w[k + g.r] = exp(-0.5 / (sigma * sigma) * (double)(k * k)); sum += g.w[k + g.r]; }
    for (int k = 0; k <= 2 * g.r; ++k) g.w[k] /= sum;
    const long total = outer * n * inner;
    ProfScope prof(F_PACK, as_stream(stream));
    hipLaunchKernelGGL(gauss_axis_kernel, dim3(grid_for(total, 256)), dim3(256), 0, as_stream(stream),
                       reinterpret_cast<const float2*>(in), reinterpret_cast<float2*>(out), total, n, inner, g);
    return check_launch("gauss_axis_kernel");
}

extern "C" int cine_combine_target(const float* img, const float* sens, float* out, int t, int c, int h, int w, int ch, int cw, void* stream) {
    CINE_REQUIRE(img && sens && out, CINE_EINVAL, "cine_combine_target: null pointer");
    CINE_REQUIRE(t > 0 && c > 0 && ch > 0 && ch <= h && cw > 0 && cw <= w, CINE_EINVAL, "cine_combine_target: Invalid shapes.");
    const long total = (long)t * ch * cw;
    ProfScope prof(F_MISC, as_stream(stream));
    hipLaunchKernelGGL(combine_target_kernel, dim3(grid_for(total, 256)), dim3(256), 0, as_stream(stream),
                       reinterpret_cast<const float2*>(img), reinterpret_cast<const float2*>(sens), out, total, c, h, w, ch, cw,
                       (h - ch) / 2, (w - cw) / 2);
    return check_launch("combine_target_kernel");
}

extern "C" int cine_espirit_lag_kernels(const float* proj, float* kpad, int c, int kk, int ny, int nx, void* stream) {
    CINE_REQUIRE(proj && kpad, CINE_EINVAL, "cine_espirit_lag_kernels: null pointer");
    CINE_REQUIRE(c > 0 && kk > 0 && ny >= 2 * kk - 1 && nx >= 2 * kk - 1, CINE_EINVAL, "cine_espirit_lag_kernels: bad sizes");
    hipStream_t st = as_stream(stream);
    ProfScope prof(F_MISC, st);
    hipError_t e = hipMemsetAsync(kpad, 0, (size_t)c * c * ny * nx * sizeof(float2), st);
    CINE_REQUIRE(e == hipSuccess, CINE_EHIP, "cine_espirit_lag_kernels: hipMemsetAsync: %s", hipGetErrorString(e));
    const long total = (long)c * c * (2 * kk - 1) * (2 * kk - 1);
    const float scale = sqrtf((float)ny * (float)nx) / (float)(kk * kk);
    hipLaunchKernelGGL(espirit_lag_kernel, dim3(grid_for(total, 256)), dim3(256), 0, st,
                       reinterpret_cast<const float2*>(proj), reinterpret_cast<float2*>(kpad), c, kk, ny, nx, scale);
    return check_launch("espirit_lag_kernel");
}

extern "C" int cine_espirit_eig(const float* m, float* maps, float* lam, int c, long npix, int iters, float crop, void* stream) {
    CINE_REQUIRE(m && maps && lam, CINE_EINVAL, "cine_espirit_eig: null pointer");
    CINE_REQUIRE(c > 0 && c <= 32 && npix > 0 && iters > 0, CINE_EINVAL, "cine_espirit_eig: 1..32 coils, iters > 0");
    hipStream_t st = as_stream(stream);
    ProfScope prof(F_MISC, st);
    const dim3 grid((unsigned)ceil_div(npix, 64L)), block(64);
    if (c <= 8) hipLaunchKernelGGL(espirit_eig_kernel<8>, grid, block, 0, st, reinterpret_cast<const float2*>(m), reinterpret_cast<float2*>(maps), lam, c, npix, iters, crop);
    else if (c <= 16) hipLaunchKernelGGL(espirit_eig_kernel<16>, grid, block, 0, st, reinterpret_cast<const float2*>(m), reinterpret_cast<float2*>(maps), lam, c, npix, iters, crop);
    else hipLaunchKernelGGL(espirit_eig_kernel<32>, grid, block, 0, st, reinterpret_cast<const float2*>(m), reinterpret_cast<float2*>(maps), lam, c, npix, iters, crop);
    return check_launch("espirit_eig_kernel");
}
