// xpd_kernels.hip -- XPDNet image-buffer plumbing around the MWCNN (reference models/xpdnet.py:406-509).
//
// The primal buffer is (b, t, 1, h, w, 2n) real with channel layout [re_0..re_{n-1}, im_0..im_{n-1}]
// (repeat_interleave on the pair axis, xpdnet.py:306-307; math.py:97-135).  The I step appends the
// backward-operator image as complex channel n (:424-429), subtracts the temporal mean, applies
// XPDNet's own temporal transform ifftshift(fft(fftshift(.))) (:466 -- NOT fft1c for odd t), rotates into
// x-f / y-f planes of 2(n+1) channels (:470-471) and pads to a multiple of 2^n_scales with the extra
// element on the LEFT for odd sizes (utils/padding.py:26-47).  The way back drops channel n (:504-509).
// All tensors here are a few MB per call; kernels are simple gathers with the fastest index on lanes.
#include "common.h"
#include "fft_core.h"

namespace cine {

constexpr int kXPix = 32;

__device__ __forceinline__ void xpd_table(cf* tw, int T) {
    const double s = 1.0 / sqrt((double)T);
    for (int j = threadIdx.x; j < T; j += blockDim.x) {
        double sn, cs;
        sincospi(2.0 * (double)j / (double)T, &sn, &cs);
        tw[j] = mk((float)(cs * s), (float)(-sn * s));
    }
}

// value of complex channel k (< nc) of pixel p, frame t: channels < n come from buf, channel n from extra
__device__ __forceinline__ cf xpd_load(const float* buf, const cf* extra, long bt_pix, int k, int n) {
    if (k < n) { const float* q = buf + bt_pix * 2 * n; return mk(q[k], q[n + k]); }
    return extra[bt_pix];
}

// X[b][pix][k][i] = temporal transform of (x - mean_t); mean[b][pix][k]
__global__ void xpd_temporal_fwd_kernel(const float* buf, const cf* extra, cf* X, cf* mean, int T, long HW, int n, int xf) {
    extern __shared__ __align__(16) unsigned char smem[];
    const int nc = n + 1;
    cf* tile = reinterpret_cast<cf*>(smem);          // [T][kXPix * nc]
    cf* tw = tile + (size_t)T * kXPix * nc;
    const int b = blockIdx.y;
    const long p0 = (long)blockIdx.x * kXPix;
    const int np = (int)min((long)kXPix, HW - p0);
    const int cols = kXPix * nc;
    if (xf) xpd_table(tw, T);
    for (int e = threadIdx.x; e < T * cols; e += blockDim.x) {
        const int t = e / cols, r = e - t * cols, p = r / nc, k = r - p * nc;
        tile[e] = p < np ? xpd_load(buf, extra, ((long)b * T + t) * HW + p0 + p, k, n) : mk(0.f, 0.f);
    }
    __syncthreads();
    for (int r = threadIdx.x; r < cols; r += blockDim.x) {
        float sx = 0.f, sy = 0.f;
        for (int t = 0; t < T; ++t) { sx += tile[t * cols + r].x; sy += tile[t * cols + r].y; }
        const cf m = mk(sx / T, sy / T);                                  // xpdnet.py:459-461
        for (int t = 0; t < T; ++t) tile[t * cols + r] = csub(tile[t * cols + r], m);
        const int p = r / nc, k = r - p * nc;
        if (p < np) mean[((long)b * HW + p0 + p) * nc + k] = m;
    }
    __syncthreads();
    const int s_in = T / 2, s_out = (T + 1) / 2;                          // fftshift first, ifftshift after (:466)
    for (int e = threadIdx.x; e < np * nc * T; e += blockDim.x) {
        const int r = e / T, i = e - r * T;
        cf v;
        if (xf) {
            int kf = i - s_out; if (kf < 0) kf += T;
            float ax = 0.f, ay = 0.f;
            int idx = (s_in * kf) % T;
            for (int g = 0; g < T; ++g) {
                const cf w = tw[idx], x = tile[g * cols + r];
                ax += x.x * w.x - x.y * w.y; ay += x.x * w.y + x.y * w.x;
                idx += kf; if (idx >= T) idx -= T;
            }
            v = mk(ax, ay);
        } else {
            v = tile[i * cols + r];
        }
        X[((long)b * HW + p0) * nc * T + (long)r * T + i] = v;
    }
}

// planes[nidx][ch][ip][jp]: ch = part * nc + k, rows = the in-plane spatial index, cols = t; zero padded
struct XpdPlaneArgs {
    const cf* X; float* planes;
    int nc, T, I, Ip, Jp, pad_i, pad_j;
    int ninner; long s_outer, s_inner, s_i;       // element strides in X (complex units of [k][t] blocks)
};
// one workgroup = one plane sample x kXRows padded rows; a thread reads a complex value once and writes both of its planes
constexpr int kXRows = 25;
__global__ __launch_bounds__(256) void xpd_plane_pack_kernel(XpdPlaneArgs a) {
    const int nidx = blockIdx.x, ip0 = blockIdx.y * kXRows;
    const cf* src = a.X + ((long)(nidx / a.ninner) * a.s_outer + (long)(nidx % a.ninner) * a.s_inner) * a.nc * a.T;
    float* dst = a.planes + (long)nidx * 2 * a.nc * a.Ip * a.Jp;
    const int per = a.Ip * a.Jp, rows = min(kXRows, a.Ip - ip0), blk = rows * a.Jp;
    for (int e = threadIdx.x; e < a.nc * blk; e += blockDim.x) {
        const int k = e / blk, r = e - k * blk, ip = ip0 + r / a.Jp, jp = r % a.Jp;
        const int i = ip - a.pad_i, j = jp - a.pad_j;
        cf z = mk(0.f, 0.f);
        if (i >= 0 && i < a.I && j >= 0 && j < a.T) z = src[((long)i * a.s_i) * a.nc * a.T + (long)k * a.T + j];
        dst[(long)k * per + ip * a.Jp + jp] = z.x;
        dst[(long)(a.nc + k) * per + ip * a.Jp + jp] = z.y;
    }
}

struct XpdUnpackArgs {
    const float* pxf; const float* pyf; const cf* mean; float* out;
    int n, T, H, W, Wpx, Tpx, pad_wx, pad_tx, Hpy, Tpy, pad_hy, pad_ty, xf;
};
// out (b, t, h, w, 2n) = inverse temporal transform of 0.5 (xf + yf) + mean channels 0..n-1
__global__ void xpd_unpack_kernel(XpdUnpackArgs a) {
    extern __shared__ __align__(16) unsigned char smem[];
    const int n = a.n, T = a.T, H = a.H, W = a.W;
    cf* tile = reinterpret_cast<cf*>(smem);          // [T][kXPix * n]
    cf* tw = tile + (size_t)T * kXPix * n;
    const int b = blockIdx.z, h = blockIdx.y, w0 = blockIdx.x * kXPix;
    const int np = min(kXPix, W - w0);
    const int cols = kXPix * n;
    if (a.xf) xpd_table(tw, T);
    const float* px = a.pxf + ((long)b * H + h) * 2 * n * a.Wpx * a.Tpx;
    for (int e = threadIdx.x; e < np * n * T; e += blockDim.x) {
        const int r = e / T, t = e - r * T, p = r / n, k = r - p * n;
        const int w = w0 + p;
        const long qx = (long)(w + a.pad_wx) * a.Tpx + t + a.pad_tx;
        const float* py = a.pyf + ((long)b * W + w) * 2 * n * a.Hpy * a.Tpy;
        const long qy = (long)(h + a.pad_hy) * a.Tpy + t + a.pad_ty;
        const long chx = (long)a.Wpx * a.Tpx, chy = (long)a.Hpy * a.Tpy;
        tile[t * cols + r] = mk(0.5f * (px[k * chx + qx] + py[k * chy + qy]),
                                0.5f * (px[(n + k) * chx + qx] + py[(n + k) * chy + qy]));     // xpdnet.py:493-496
    }
    __syncthreads();
    const int s_in = (T + 1) / 2, s_out = T / 2;      // fftshift(ifft(ifftshift(.))) (:500) = the fftc.py order
    const long HW = (long)H * W;
    for (int e = threadIdx.x; e < T * np * n; e += blockDim.x) {
        const int i = e / (np * n), r = e - i * (np * n), p = r / n, k = r - p * n;
        cf v;
        if (a.xf) {
            int kf = i - s_out; if (kf < 0) kf += T;
            float ax = 0.f, ay = 0.f;
            int idx = (s_in * kf) % T;
            for (int g = 0; g < T; ++g) {
                const cf w = tw[idx], x = tile[g * cols + r];
                ax += x.x * w.x + x.y * w.y; ay += x.y * w.x - x.x * w.y;
                idx += kf; if (idx >= T) idx -= T;
            }
            v = mk(ax, ay);
        } else {
            v = tile[i * cols + r];
        }
        const long pix = (long)h * W + w0 + p;
        const cf m = a.mean[((long)b * HW + pix) * (n + 1) + k];          // residual drops channel n (:504-509)
        float* o = a.out + (((long)b * T + i) * HW + pix) * 2 * n;
        o[k] = v.x + m.x; o[n + k] = v.y + m.y;
    }
}

// generic (N, HW, C) channel-last -> zero-padded planes (N, C, Hp, Wp) and back (2-D mode, xpdnet.py:442-444)
__global__ void chanlast_to_planes_kernel(const float* x, float* planes, int C, int H, int W, int Hp, int Wp, int ph, int pw) {
    const int n = blockIdx.y;
    const long total = (long)C * Hp * Wp;
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
        const int c = (int)(e / ((long)Hp * Wp));
        const long r = e - (long)c * Hp * Wp;
        const int hp = (int)(r / Wp), wp = (int)(r - (long)hp * Wp);
        const int h = hp - ph, w = wp - pw;
        planes[(long)n * total + e] = (h >= 0 && h < H && w >= 0 && w < W) ? x[(((long)n * H + h) * W + w) * C + c] : 0.f;
    }
}
__global__ void planes_to_chanlast_kernel(const float* planes, float* y, int C, int H, int W, int Hp, int Wp, int ph, int pw) {
    const int n = blockIdx.y;
    const long total = (long)H * W * C;
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
        const int c = (int)(e % C);
        const long pix = e / C;
        const int h = (int)(pix / W), w = (int)(pix - (long)h * W);
        y[(long)n * total + e] = planes[(((long)n * C + c) * Hp + h + ph) * Wp + w + pw];
    }
}

// complex image (re = channel c_re, im = channel c_im) out of a channel-last real buffer (xpdnet.py:128, 161)
__global__ void extract_complex_kernel(const float* buf, cf* out, long npix, int C, int c_re, int c_im) {
    for (long p = (long)blockIdx.x * blockDim.x + threadIdx.x; p < npix; p += (long)gridDim.x * blockDim.x)
        out[p] = mk(buf[p * C + c_re], buf[p * C + c_im]);
}
// repeat_interleave(image, n, dim=-1) of a complex image (xpdnet.py:307): [re x n, im x n]
__global__ void repeat_complex_kernel(const cf* img, float* buf, long npix, int n) {
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < npix * 2 * n; e += (long)gridDim.x * blockDim.x) {
        const long p = e / (2 * n); const int c = (int)(e - p * 2 * n);
        buf[e] = c < n ? img[p].x : img[p].y;
    }
}

static unsigned xgrid(long n, int threads, long cap = 4096) {
    long g = ceil_div(n, (long)threads);
    return (unsigned)(g > cap ? cap : (g < 1 ? 1 : g));
}
}  // namespace cine

using namespace cine;

// padding of utils/padding.py:26-47 for one dim: multiple of 2^n_scales, odd sizes get the extra on the left
extern "C" int cine_mwcnn_pad(int size, int n_scales, int* left, int* right) {
    const int m = 1 << n_scales;
    const int n_pad = size % m == 0 ? 0 : (size / m + 1) * m - size;
    const int l = (size % 2 == 0 || n_pad == 0) ? n_pad / 2 : 1 + n_pad / 2;
    if (left) *left = l;
    if (right) *right = n_pad / 2;
    return size + l + n_pad / 2;
}

extern "C" size_t cine_xpd_ws_bytes(int b, int t, int h, int w, int n_primal) {
    return (size_t)b * t * h * w * (n_primal + 1) * sizeof(cf);
}

extern "C" int cine_xpd_pack(const float* buf, const float* extra, float* planes_xf, float* planes_yf, float* mean,
                             int b, int t, int h, int w, int n_primal, int n_scales, int xf,
                             void* ws, size_t ws_bytes, void* stream) {
    CINE_REQUIRE(buf && extra && planes_xf && planes_yf && mean && ws, CINE_EINVAL, "cine_xpd_pack: null pointer");
    CINE_REQUIRE(b > 0 && t > 1 && t <= 64 && h > 0 && w > 0 && n_primal >= 1 && n_primal <= 15 && b <= 65535, CINE_EINVAL,
                 "cine_xpd_pack: bad sizes");
    CINE_REQUIRE(ws_bytes >= cine_xpd_ws_bytes(b, t, h, w, n_primal), CINE_EWORKSPACE, "cine_xpd_pack: workspace too small");
    hipStream_t st = as_stream(stream);
    const int nc = n_primal + 1;
    const long HW = (long)h * w;
    cf* X = reinterpret_cast<cf*>(ws);                   // [b][h][w][k][t]
    ProfScope prof(F_PACK, st);
    const size_t lds = ((size_t)t * kXPix * nc + t) * sizeof(cf);
    CINE_REQUIRE(lds <= 64 * 1024, CINE_EUNSUPPORTED, "cine_xpd_pack: tile does not fit LDS");
    hipLaunchKernelGGL(xpd_temporal_fwd_kernel, dim3((unsigned)ceil_div(HW, (long)kXPix), b), dim3(256), lds, st,
                       buf, reinterpret_cast<const cf*>(extra), X, reinterpret_cast<cf*>(mean), t, HW, n_primal, xf);
    if (int e = check_launch("xpd_temporal_fwd_kernel")) return e;
    int lt, rt, lw, rw, lh, rh;
    const int tp = cine_mwcnn_pad(t, n_scales, &lt, &rt), wp = cine_mwcnn_pad(w, n_scales, &lw, &rw),
              hp = cine_mwcnn_pad(h, n_scales, &lh, &rh);
    XpdPlaneArgs a{};
    a.X = X; a.nc = nc; a.T = t;
    // x-f: sample (b, h), rows = w                      (xpdnet.py:470)
    a.planes = planes_xf; a.I = w; a.Ip = wp; a.Jp = tp; a.pad_i = lw; a.pad_j = lt;
    a.ninner = h; a.s_outer = HW; a.s_inner = w; a.s_i = 1;
    hipLaunchKernelGGL(xpd_plane_pack_kernel, dim3(b * h, ceil_div(wp, kXRows)), dim3(256), 0, st, a);
    // y-f: sample (b, w), rows = h                      (xpdnet.py:471)
    a.planes = planes_yf; a.I = h; a.Ip = hp; a.Jp = tp; a.pad_i = lh; a.pad_j = lt;
    a.ninner = w; a.s_outer = HW; a.s_inner = 1; a.s_i = w;
    hipLaunchKernelGGL(xpd_plane_pack_kernel, dim3(b * w, ceil_div(hp, kXRows)), dim3(256), 0, st, a);
    return check_launch("xpd_plane_pack_kernel");
}

extern "C" int cine_xpd_unpack(const float* planes_xf, const float* planes_yf, const float* mean, float* out,
                               int b, int t, int h, int w, int n_primal, int n_scales, int xf, void* stream) {
    CINE_REQUIRE(planes_xf && planes_yf && mean && out, CINE_EINVAL, "cine_xpd_unpack: null pointer");
    CINE_REQUIRE(b > 0 && t > 1 && t <= 64 && h > 0 && w > 0 && n_primal >= 1 && h <= 65535 && b <= 65535, CINE_EINVAL,
                 "cine_xpd_unpack: bad sizes");
    XpdUnpackArgs a{};
    a.pxf = planes_xf; a.pyf = planes_yf; a.mean = reinterpret_cast<const cf*>(mean); a.out = out;
    a.n = n_primal; a.T = t; a.H = h; a.W = w; a.xf = xf;
    int r;
    a.Tpx = a.Tpy = cine_mwcnn_pad(t, n_scales, &a.pad_tx, &r); a.pad_ty = a.pad_tx;
    a.Wpx = cine_mwcnn_pad(w, n_scales, &a.pad_wx, &r);
    a.Hpy = cine_mwcnn_pad(h, n_scales, &a.pad_hy, &r);
    const size_t lds = ((size_t)t * kXPix * n_primal + t) * sizeof(cf);
    CINE_REQUIRE(lds <= 64 * 1024, CINE_EUNSUPPORTED, "cine_xpd_unpack: tile does not fit LDS");
    ProfScope prof(F_PACK, as_stream(stream));
    hipLaunchKernelGGL(xpd_unpack_kernel, dim3(ceil_div(w, kXPix), h, b), dim3(256), lds, as_stream(stream), a);
    return check_launch("xpd_unpack_kernel");
}

extern "C" int cine_chanlast_to_planes(const float* x, float* planes, int n, int c, int h, int w, int n_scales, void* stream) {
    CINE_REQUIRE(x && planes && n > 0 && n <= 65535 && c > 0 && h > 0 && w > 0, CINE_EINVAL, "cine_chanlast_to_planes: bad arguments");
    int lh, lw, r;
    const int hp = cine_mwcnn_pad(h, n_scales, &lh, &r), wp = cine_mwcnn_pad(w, n_scales, &lw, &r);
    ProfScope prof(F_PACK, as_stream(stream));
    hipLaunchKernelGGL(chanlast_to_planes_kernel, dim3(xgrid((long)c * hp * wp, 256, 1024), n), dim3(256), 0, as_stream(stream),
                       x, planes, c, h, w, hp, wp, lh, lw);
    return check_launch("chanlast_to_planes_kernel");
}

extern "C" int cine_planes_to_chanlast(const float* planes, float* y, int n, int c, int h, int w, int n_scales, void* stream) {
    CINE_REQUIRE(planes && y && n > 0 && n <= 65535 && c > 0 && h > 0 && w > 0, CINE_EINVAL, "cine_planes_to_chanlast: bad arguments");
    int lh, lw, r;
    const int hp = cine_mwcnn_pad(h, n_scales, &lh, &r), wp = cine_mwcnn_pad(w, n_scales, &lw, &r);
    ProfScope prof(F_PACK, as_stream(stream));
    hipLaunchKernelGGL(planes_to_chanlast_kernel, dim3(xgrid((long)c * h * w, 256, 1024), n), dim3(256), 0, as_stream(stream),
                       planes, y, c, h, w, hp, wp, lh, lw);
    return check_launch("planes_to_chanlast_kernel");
}

extern "C" int cine_extract_complex(const float* buf, float* out, long npix, int c, int c_re, int c_im, void* stream) {
    CINE_REQUIRE(buf && out && npix > 0 && c > 0 && c_re >= 0 && c_re < c && c_im >= 0 && c_im < c, CINE_EINVAL,
                 "cine_extract_complex: bad arguments");
    ProfScope prof(F_MISC, as_stream(stream));
    hipLaunchKernelGGL(extract_complex_kernel, dim3(xgrid(npix, 256)), dim3(256), 0, as_stream(stream), buf,
                       reinterpret_cast<cf*>(out), npix, c, c_re, c_im);
    return check_launch("extract_complex_kernel");
}

extern "C" int cine_repeat_complex(const float* img, float* buf, long npix, int n, void* stream) {
    CINE_REQUIRE(img && buf && npix > 0 && n > 0, CINE_EINVAL, "cine_repeat_complex: bad arguments");
    ProfScope prof(F_MISC, as_stream(stream));
    hipLaunchKernelGGL(repeat_complex_kernel, dim3(xgrid(npix * 2 * n, 256)), dim3(256), 0, as_stream(stream),
                       reinterpret_cast<const cf*>(img), buf, npix, n);
    return check_launch("repeat_complex_kernel");
}
