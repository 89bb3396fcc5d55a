// metrics_kernels.hip -- the steps either side of the reconstruction path:
//   cine_apply_mask      : data * mask + 0.0 for Cartesian row masks             (data/transforms.py:66-92)
//   cine_scale           : x *= s (fft2c / ifft2c with norm=None / "forward")    (utils/fftc.py:59-110, run_inference.py:66)
//   cine_image_metrics   : center crop -> SSIM (7x7 uniform window, sample covariance, K1 .01, K2 .03, mean over the
//                          window-valid region and over frames), NMSE, PSNR, MSE (utils/evaluate.py:6-50 = skimage's
//                          structural_similarity / peak_signal_noise_ratio defaults; utils/losses.py:25-58 for the per-frame
//                          data range of SSIMLoss; data/transforms.py:161-183 for the crop)
// The metric arithmetic is float64 like skimage's (the inputs are float32 images): the window moments u_xx - u_x^2 cancel.
#include "common.h"

namespace cine {

__global__ void apply_mask_kernel(const float2* k, const uint8_t* mask, float2* out, long hw, int h, int w, int c) {
    // grid.y = (frame, coil) image; mask row set = image / c
    const long img = blockIdx.y;
    const uint8_t* m = mask + (img / c) * h;
    const float2* src = k + img * hw;
    float2* dst = out + img * hw;
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < hw; e += (long)gridDim.x * blockDim.x) {
        const int row = (int)(e / w);
        const float2 v = src[e];
        const float mv = m[row] ? 1.f : 0.f;
        dst[e] = make_float2(v.x * mv + 0.0f, v.y * mv + 0.0f);      // "+ 0.0" turns -0 into +0 as the reference does (:91)
    }
}

__global__ void scale_kernel(float* x, long n, float s) {
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (long)gridDim.x * blockDim.x) x[e] *= s;
}

struct MetricArgs {
    const float* gt; const float* pred;
    int T, Hg, Wg, Hp, Wp;          // stored frame sizes
    int ch, cw;                     // common center crop (data/transforms.py:161-183)
    int og_y, og_x, op_y, op_x;     // crop offsets into gt / pred
    int win; double k1, k2;
    int range_mode; double maxval;  // 0: max of the cropped gt volume, 1: max of each gt frame (SSIMLoss), 2: maxval
    double* frame;                  // [T][4] = {max gt, sum (gt-pred)^2, sum gt^2, sum of the SSIM map}
    double* part; int nblk;         // [T][nblk] partial SSIM-map sums (fixed order: deterministic)
    double* out;                    // [4 + T] = {ssim, nmse, psnr, mse, ssim of frame 0..T-1}
};

__device__ __forceinline__ double block_sum(double v, double* red) {
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    const int wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[wave] = v;
    __syncthreads();
    double t = 0.0;
    for (int i = 0; i < nw; ++i) t += red[i];
    return t;
}
__device__ __forceinline__ double block_max(double v, double* red) {
    for (int o = 32; o > 0; o >>= 1) v = fmax(v, __shfl_xor(v, o, 64));
    const int wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[wave] = v;
    __syncthreads();
    double t = red[0];
    for (int i = 1; i < nw; ++i) t = fmax(t, red[i]);
    return t;
}

// one workgroup per frame: max of gt, squared error, energy of gt over the crop
__global__ __launch_bounds__(256) void metric_reduce_kernel(MetricArgs a) {
    __shared__ double red[4];
    const int f = blockIdx.x;
    const float* g = a.gt + (long)f * a.Hg * a.Wg;
    const float* p = a.pred + (long)f * a.Hp * a.Wp;
    double mx = -1e300, se = 0.0, en = 0.0;
    for (int e = threadIdx.x; e < a.ch * a.cw; e += blockDim.x) {
        const int y = e / a.cw, x = e - y * a.cw;
        const double gv = g[(long)(y + a.og_y) * a.Wg + x + a.og_x], pv = p[(long)(y + a.op_y) * a.Wp + x + a.op_x];
        mx = fmax(mx, gv); se += (gv - pv) * (gv - pv); en += gv * gv;
    }
    mx = block_max(mx, red); se = block_sum(se, red); en = block_sum(en, red);
    if (threadIdx.x == 0) { a.frame[4 * f] = mx; a.frame[4 * f + 1] = se; a.frame[4 * f + 2] = en; }
}

constexpr int kSsimTile = 32, kSsimMaxWin = 11;
// SSIM map of one 32 x 32 tile of window-valid positions of one frame; the tile's (32 + win - 1)^2 inputs sit in LDS
__global__ __launch_bounds__(256) void ssim_tile_kernel(MetricArgs a) {
    __shared__ double gx[(kSsimTile + kSsimMaxWin - 1) * (kSsimTile + kSsimMaxWin - 1)];
    __shared__ double px[(kSsimTile + kSsimMaxWin - 1) * (kSsimTile + kSsimMaxWin - 1)];
    __shared__ double red[4];
    const int f = blockIdx.z, win = a.win, span = kSsimTile + win - 1;
    const int vy = a.ch - win + 1, vx = a.cw - win + 1;            // window-valid positions
    const int y0 = blockIdx.y * kSsimTile, x0 = blockIdx.x * kSsimTile;
    const float* g = a.gt + (long)f * a.Hg * a.Wg;
    const float* p = a.pred + (long)f * a.Hp * a.Wp;
    for (int e = threadIdx.x; e < span * span; e += blockDim.x) {
        const int ty = e / span, tx = e - ty * span;
        const int y = min(y0 + ty, a.ch - 1), x = min(x0 + tx, a.cw - 1);
        gx[e] = g[(long)(y + a.og_y) * a.Wg + x + a.og_x];
        px[e] = p[(long)(y + a.op_y) * a.Wp + x + a.op_x];
    }
    double range = a.maxval;
    if (a.range_mode == 0) { range = a.frame[0]; for (int i = 1; i < a.T; ++i) range = fmax(range, a.frame[4 * i]); }
    else if (a.range_mode == 1) range = a.frame[4 * f];
    const double c1 = (a.k1 * range) * (a.k1 * range), c2 = (a.k2 * range) * (a.k2 * range);
    const double np = (double)(win * win), cov = np / (np - 1.0);
    __syncthreads();
    double acc = 0.0;
    for (int e = threadIdx.x; e < kSsimTile * kSsimTile; e += blockDim.x) {
        const int oy = e / kSsimTile, ox = e - oy * kSsimTile;
        if (y0 + oy >= vy || x0 + ox >= vx) continue;
        double sx = 0, sy = 0, sxx = 0, syy = 0, sxy = 0;
        for (int dy = 0; dy < win; ++dy)
            for (int dx = 0; dx < win; ++dx) {
                const double xv = gx[(oy + dy) * span + ox + dx], yv = px[(oy + dy) * span + ox + dx];
                sx += xv; sy += yv; sxx += xv * xv; syy += yv * yv; sxy += xv * yv;
            }
        const double ux = sx / np, uy = sy / np;
        const double vxv = cov * (sxx / np - ux * ux), vyv = cov * (syy / np - uy * uy), vxy = cov * (sxy / np - ux * uy);
        acc += ((2 * ux * uy + c1) * (2 * vxy + c2)) / ((ux * ux + uy * uy + c1) * (vxv + vyv + c2));
    }
    acc = block_sum(acc, red);
    if (threadIdx.x == 0) a.part[(long)f * a.nblk + blockIdx.y * gridDim.x + blockIdx.x] = acc;
}

__global__ void metric_final_kernel(MetricArgs a) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    const double nvalid = (double)(a.ch - a.win + 1) * (a.cw - a.win + 1);
    double ssim = 0.0, se = 0.0, en = 0.0, mx = -1e300;
    for (int f = 0; f < a.T; ++f) {
        double s = 0.0;
        for (int i = 0; i < a.nblk; ++i) s += a.part[(long)f * a.nblk + i];
        s /= nvalid;
        a.out[4 + f] = s; ssim += s;
        mx = fmax(mx, a.frame[4 * f]); se += a.frame[4 * f + 1]; en += a.frame[4 * f + 2];
    }
    const double n = (double)a.T * a.ch * a.cw, mse = se / n;
    const double peak = a.range_mode == 2 ? a.maxval : mx;
    a.out[0] = ssim / a.T;                       // evaluate.py:25-42
    a.out[1] = se / en;                          // evaluate.py:11-13  ||gt - pred||^2 / ||gt||^2
    a.out[2] = 10.0 * log10(peak * peak / mse);  // evaluate.py:16-22 (skimage peak_signal_noise_ratio)
    a.out[3] = mse;                              // evaluate.py:6-8
}

static unsigned grid1(long n, int threads, long cap = 4096) {
    long g = (n + threads - 1) / threads;
    return (unsigned)(g < 1 ? 1 : (g > cap ? cap : g));
}

}  // namespace cine

using namespace cine;

extern "C" int cine_apply_mask(const float* kspace, const uint8_t* mask, float* out, long bt, int c, int h, int w, void* stream) {
    CINE_REQUIRE(kspace && mask && out, CINE_EINVAL, "cine_apply_mask: null pointer");
    CINE_REQUIRE(bt > 0 && c > 0 && h > 0 && w > 0 && bt * c <= 65535, CINE_EINVAL, "cine_apply_mask: bad sizes");
    hipStream_t st = as_stream(stream);
    ProfScope prof(F_MISC, st);
    const long hw = (long)h * w;
    hipLaunchKernelGGL(apply_mask_kernel, dim3(grid1(hw, 256, 64), (unsigned)(bt * c)), dim3(256), 0, st,
                       reinterpret_cast<const float2*>(kspace), mask, reinterpret_cast<float2*>(out), hw, h, w, c);
    return check_launch("apply_mask_kernel");
}

extern "C" int cine_scale(float* x, long n, float s, void* stream) {
    CINE_REQUIRE(x && n >= 0, CINE_EINVAL, "cine_scale: bad arguments");
    if (n == 0) return CINE_OK;
    hipStream_t st = as_stream(stream);
    ProfScope prof(F_MISC, st);
    hipLaunchKernelGGL(scale_kernel, dim3(grid1(n, 256)), dim3(256), 0, st, x, n, s);
    return check_launch("scale_kernel");
}

extern "C" size_t cine_image_metrics_ws_bytes(int t, int hg, int wg, int hp, int wp, int win) {
    if (t <= 0 || hg <= 0 || wg <= 0 || hp <= 0 || wp <= 0 || win < 1) return 0;
    const int ch = hg < hp ? hg : hp, cw = wg < wp ? wg : wp;
    if (ch < win || cw < win) return 0;
    const long nblk = (long)ceil_div(ch - win + 1, kSsimTile) * ceil_div(cw - win + 1, kSsimTile);
    return (size_t)((long)t * 4 + (long)t * nblk) * sizeof(double);
}

extern "C" int cine_image_metrics(const float* gt, const float* pred, int t, int hg, int wg, int hp, int wp,
                                  int win, double k1, double k2, int range_mode, double maxval,
                                  double* out, void* ws, size_t ws_bytes, void* stream) {
    CINE_REQUIRE(gt && pred && out && ws, CINE_EINVAL, "cine_image_metrics: null pointer");
    CINE_REQUIRE(t > 0 && t <= 65535 && hg > 0 && wg > 0 && hp > 0 && wp > 0, CINE_EINVAL, "cine_image_metrics: bad sizes");
    CINE_REQUIRE(win >= 1 && win <= kSsimMaxWin && (win & 1), CINE_EINVAL, "cine_image_metrics: window %d (odd, <= %d)", win, kSsimMaxWin);
    CINE_REQUIRE(range_mode >= 0 && range_mode <= 2, CINE_EINVAL, "cine_image_metrics: range_mode %d", range_mode);
    MetricArgs a{};
    a.gt = gt; a.pred = pred; a.T = t; a.Hg = hg; a.Wg = wg; a.Hp = hp; a.Wp = wp;
    a.ch = hg < hp ? hg : hp; a.cw = wg < wp ? wg : wp;
    CINE_REQUIRE(a.ch >= win && a.cw >= win, CINE_EINVAL, "cine_image_metrics: %dx%d crop smaller than the %d window", a.ch, a.cw, win);
    // center_crop (data/transforms.py:150-158): from = (size - crop) // 2
    a.og_y = (hg - a.ch) / 2; a.og_x = (wg - a.cw) / 2; a.op_y = (hp - a.ch) / 2; a.op_x = (wp - a.cw) / 2;
    a.win = win; a.k1 = k1; a.k2 = k2; a.range_mode = range_mode; a.maxval = maxval;
    const size_t need = cine_image_metrics_ws_bytes(t, hg, wg, hp, wp, win);
    CINE_REQUIRE(ws_bytes >= need, CINE_EWORKSPACE, "cine_image_metrics: workspace %zu < %zu", ws_bytes, need);
    const int gx = ceil_div(a.cw - win + 1, kSsimTile), gy = ceil_div(a.ch - win + 1, kSsimTile);
    a.frame = reinterpret_cast<double*>(ws); a.part = a.frame + (long)t * 4; a.nblk = gx * gy; a.out = out;
    hipStream_t st = as_stream(stream);
    ProfScope prof(F_MISC, st);
    hipLaunchKernelGGL(metric_reduce_kernel, dim3(t), dim3(256), 0, st, a);
    hipLaunchKernelGGL(ssim_tile_kernel, dim3(gx, gy, t), dim3(256), 0, st, a);
    hipLaunchKernelGGL(metric_final_kernel, dim3(1), dim3(64), 0, st, a);
    return check_launch("cine_image_metrics");
}
