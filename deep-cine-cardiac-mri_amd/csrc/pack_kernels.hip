// pack_kernels.hip -- NormUnet front/back halves, XT/XF rotations, sens-map prologue.
//
// These are the small HBM-bound byte-moving steps around the regulariser
// (SURVEY.md K3/K4/K5/K13/K14): every tensor here is a few MB, so the kernels
// favour simplicity; each reads/writes with the fastest-varying index on lanes.
#include <algorithm>
#include "common.h"
#include "fft_core.h"

namespace cine {

// ---------------------------------------------------------------- block reductions
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
// sum over the workgroup; `red` holds >= 16 floats; all threads get the result
__device__ __forceinline__ float block_sum(float v, float* red) {
    v = wave_sum(v);
    const int wave = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[wave] = v;
    __syncthreads();
    float s = 0.f;
    for (int i = 0; i < nw; ++i) s += red[i];
    return s;
}

// ---------------------------------------------------------------- NormUnet pack / unpack
// Generic strided gather of sample n, channel ch (re|im), plane element (i, j):
//   x[(n / ninner) * s_outer + (n % ninner) * s_inner + i * si + j * sj + ch]
struct PackArgs {
    const float* x; float* planes; float* stats;
    int n, I, J, Ip, Jp, pad_i, pad_j;
    int ninner; long s_outer, s_inner, si, sj;
    int norm;                     // 0: plain re/im planes (CineNet / XPDNet feed a bare Unet), 1: NormUnet group norm
};

// Two argument sets per launch (blockIdx.y): the x-f and y-f plane sets of one cascade go out together.
struct PackArgs2 { PackArgs s[2]; };
constexpr int kPackRegs = 16;       // plane elements a thread keeps in registers on the single-read path

__global__ void normunet_pack_kernel(PackArgs2 two) {
    __shared__ float red[16];
    const PackArgs& a = two.s[blockIdx.y];
    const int n = blockIdx.x;
    if (n >= a.n) return;
    const float* src = a.x + (long)(n / a.ninner) * a.s_outer + (long)(n % a.ninner) * a.s_inner;
    const int cnt = a.I * a.J;
    const int nt = blockDim.x;
    float* pr = a.planes + (long)n * 2 * a.Ip * a.Jp;
    float* pi = pr + (long)a.Ip * a.Jp;
    if (!a.norm) {
        for (int e = threadIdx.x; e < a.Ip * a.Jp; e += nt) {
            const int ip = e / a.Jp, jp = e - ip * a.Jp;
            const int i = ip - a.pad_i, j = jp - a.pad_j;
            float2 v = make_float2(0.f, 0.f);
            if (i >= 0 && i < a.I && j >= 0 && j < a.J) v = *reinterpret_cast<const float2*>(src + i * a.si + j * a.sj);
            pr[e] = v.x; pi[e] = v.y;
        }
        return;
    }
    const bool inreg = cnt <= kPackRegs * nt;          // uniform: the whole plane fits the workgroup's registers
    float2 keep[kPackRegs];
    // pass 1: means (norm_unet.py:64)
    float sr = 0.f, si_ = 0.f;
    if (inreg) {
#pragma unroll
        for (int k = 0; k < kPackRegs; ++k) {
            const int e = threadIdx.x + k * nt;
            const int ec = min(e, cnt - 1);
            const int i = ec / a.J, j = ec - i * a.J;
            keep[k] = *reinterpret_cast<const float2*>(src + i * a.si + j * a.sj);
            if (e < cnt) { sr += keep[k].x; si_ += keep[k].y; }
        }
    } else {
        for (int e = threadIdx.x; e < cnt; e += nt) {
            const int i = e / a.J, j = e - i * a.J;
            const float2 v = *reinterpret_cast<const float2*>(src + i * a.si + j * a.sj);
            sr += v.x; si_ += v.y;
        }
    }
    const float mr = block_sum(sr, red) / cnt;
    const float mi = block_sum(si_, red) / cnt;
    // pass 2: unbiased std (norm_unet.py:65, torch.std default)
    float qr = 0.f, qi = 0.f;
    if (inreg) {
#pragma unroll
        for (int k = 0; k < kPackRegs; ++k)
            if (threadIdx.x + k * nt < cnt) { qr += (keep[k].x - mr) * (keep[k].x - mr); qi += (keep[k].y - mi) * (keep[k].y - mi); }
    } else {
        for (int e = threadIdx.x; e < cnt; e += nt) {
            const int i = e / a.J, j = e - i * a.J;
            const float2 v = *reinterpret_cast<const float2*>(src + i * a.si + j * a.sj);
            qr += (v.x - mr) * (v.x - mr); qi += (v.y - mi) * (v.y - mi);
        }
    }
    const float sdr = sqrtf(block_sum(qr, red) / (cnt - 1));
    const float sdi = sqrtf(block_sum(qi, red) / (cnt - 1));
    if (threadIdx.x == 0) {
        float* st = a.stats + (long)n * 4;
        st[0] = mr; st[1] = sdr; st[2] = mi; st[3] = sdi;
    }
    // pass 3: (x - mean) / std into the zero-padded planes (norm_unet.py:69, 76-86)
    if (inreg) {
        // interior from the registers, then the zero frame
#pragma unroll
        for (int k = 0; k < kPackRegs; ++k) {
            const int e = threadIdx.x + k * nt;
            if (e < cnt) {
                const int i = e / a.J, j = e - i * a.J;
                const int q = (i + a.pad_i) * a.Jp + j + a.pad_j;
                pr[q] = (keep[k].x - mr) / sdr; pi[q] = (keep[k].y - mi) / sdi;
            }
        }
        if (a.Ip != a.I || a.Jp != a.J) {
            for (int e = threadIdx.x; e < a.Ip * a.Jp; e += nt) {
                const int ip = e / a.Jp, jp = e - ip * a.Jp;
                const int i = ip - a.pad_i, j = jp - a.pad_j;
                if (i < 0 || i >= a.I || j < 0 || j >= a.J) { pr[e] = 0.f; pi[e] = 0.f; }
            }
        }
        return;
    }
    for (int e = threadIdx.x; e < a.Ip * a.Jp; e += nt) {
        const int ip = e / a.Jp, jp = e - ip * a.Jp;
        const int i = ip - a.pad_i, j = jp - a.pad_j;
        float vr = 0.f, vi = 0.f;
        if (i >= 0 && i < a.I && j >= 0 && j < a.J) {
            const float2 v = *reinterpret_cast<const float2*>(src + i * a.si + j * a.sj);
            vr = (v.x - mr) / sdr; vi = (v.y - mi) / sdi;
        }
        pr[e] = vr; pi[e] = vi;
    }
}

// threads per workgroup: the smallest of 256 / 512 / 1024 that keeps a plane in registers, else 1024
static int pack_threads(const PackArgs& a) {
    const long cnt = (long)a.I * a.J;
    for (int nt : {256, 512, 1024}) if (cnt <= (long)kPackRegs * nt) return nt;
    return 1024;
}
static int launch_pack(const PackArgs& a0, const PackArgs* a1, hipStream_t st) {
    PackArgs2 two{};
    two.s[0] = a0; two.s[1] = a1 ? *a1 : a0;
    const int nt = std::max(pack_threads(a0), a1 ? pack_threads(*a1) : 0);
    hipLaunchKernelGGL(normunet_pack_kernel, dim3(std::max(a0.n, a1 ? a1->n : 0), a1 ? 2 : 1), dim3(nt), 0, st, two);
    return check_launch("normunet_pack_kernel");
}

__global__ void normunet_unpack_kernel(const float* planes, const float* stats, float* y,
                                       int I, int J, int Ip, int Jp, int pad_i, int pad_j) {
    const int n = blockIdx.y;
    float mr = 0.f, sdr = 1.f, mi = 0.f, sdi = 1.f;
    if (stats) { const float* st = stats + (long)n * 4; mr = st[0]; sdr = st[1]; mi = st[2]; sdi = st[3]; }
    const float* pr = planes + (long)n * 2 * Ip * Jp;
    const float* pi = pr + (long)Ip * Jp;
    for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < I * J; e += gridDim.x * blockDim.x) {
        const int i = e / J, j = e - i * J;
        const int q = (i + pad_i) * Jp + (j + pad_j);
        float2 v;
        v.x = pr[q] * sdr + mr;          // norm_unet.py:71-74
        v.y = pi[q] * sdi + mi;
        reinterpret_cast<float2*>(y)[(long)n * I * J + e] = v;
    }
}

// ---------------------------------------------------------------- temporal front half
// One workgroup = PIX consecutive pixels of (b, h*w); LDS holds [T][PIX] complex.
// X[b][pix][k] = centered ortho DFT over t of (img - mean_t), or just img - mean_t (XT).
constexpr int kPix = 64;

__device__ __forceinline__ void temporal_table(cf* tw, int T) {
    const double s = 1.0 / sqrt((double)T);
    for (int j = threadIdx.x; j < T; j += blockDim.x) {
        double sn, cs;
        sincospi(2.0 * (double)j / (double)T, &sn, &cs);
        tw[j] = mk((float)(cs * s), (float)(-sn * s));
    }
}

__global__ void temporal_fwd_kernel(const cf* img, cf* X, cf* mean_img, int T, long HW, int xf) {
    extern __shared__ __align__(16) unsigned char smem[];
    cf* buf = reinterpret_cast<cf*>(smem);          // [T][kPix]
    cf* tw = buf + T * kPix;                        // [T]
    const int b = blockIdx.y;
    const long p0 = (long)blockIdx.x * kPix;
    const int np = (int)min((long)kPix, HW - p0);
    if (xf) temporal_table(tw, T);
    for (int e = threadIdx.x; e < T * kPix; e += blockDim.x) {
        const int t = e / kPix, p = e - t * kPix;
        buf[e] = p < np ? img[((long)b * T + t) * HW + p0 + p] : mk(0.f, 0.f);
    }
    __syncthreads();
    // temporal mean per pixel (varnet.py:205-206); kept in the extra row after use
    if (threadIdx.x < kPix) {
        const int p = threadIdx.x;
        float sx = 0.f, sy = 0.f;
        for (int t = 0; t < T; ++t) { sx += buf[t * kPix + p].x; sy += buf[t * kPix + p].y; }
        const cf m = mk(sx / T, sy / T);
        for (int t = 0; t < T; ++t) buf[t * kPix + p] = csub(buf[t * kPix + p], m);   // :207
        if (p < np) mean_img[(long)b * HW + p0 + p] = m;
    }
    __syncthreads();
    const int s_in = (T + 1) / 2, s_out = T / 2;
    // output element (p, i): X[(b*HW + p0 + p) * T + i]; lanes run over i fastest for coalescing
    for (int e = threadIdx.x; e < np * T; e += blockDim.x) {
        const int p = e / T, i = e - p * T;
        cf r;
        if (xf) {                                    // fft1c over t, varnet.py:209-213
            int k = i - s_out; if (k < 0) k += T;
            float ax = 0.f, ay = 0.f;
            int idx = (s_in * k) % T;                // n = (g + s_in) % T  ->  (n * k) % T
            for (int g = 0; g < T; ++g) {
                const cf w = tw[idx];
                const cf x = buf[g * kPix + p];
                ax += x.x * w.x - x.y * w.y; ay += x.x * w.y + x.y * w.x;
                idx += k; if (idx >= T) idx -= T;
            }
            r = mk(ax, ay);
        } else {
            r = buf[i * kPix + p];
        }
        X[((long)b * HW + p0 + p) * T + i] = r;
    }
}

// back half: avg of unnormalised xf / yf planes -> inverse temporal DFT -> + mean
struct UnpackArgs {
    const float* pxf; const float* pyf; const float* sxf; const float* syf;
    const cf* mean_img; cf* out;
    int T, H, W, Tp, Hp, Wp, pad_t, pad_h, pad_w, xf;
};

__global__ void xfyf_unpack_kernel(UnpackArgs a) {
    extern __shared__ __align__(16) unsigned char smem[];
    cf* buf = reinterpret_cast<cf*>(smem);          // [T][kPix]
    cf* tw = buf + a.T * kPix;
    const int T = a.T, H = a.H, W = a.W;
    const int b = blockIdx.z, h = blockIdx.y;
    const int w0 = blockIdx.x * kPix;
    const int np = min(kPix, W - w0);
    if (a.xf) temporal_table(tw, T);
    // xf plane sample n = b*H + h: (2, Wp, Tp), element [ch][w + pad_w][t + pad_t]
    // yf plane sample n = b*W + w: (2, Hp, Tp), element [ch][h + pad_h][t + pad_t]
    const long nxf = (long)b * H + h;
    float xmr = 0.f, xsr = 1.f, xmi = 0.f, xsi = 1.f;
    if (a.sxf) { const float* sx = a.sxf + nxf * 4; xmr = sx[0]; xsr = sx[1]; xmi = sx[2]; xsi = sx[3]; }
    const float* pxr = a.pxf + nxf * 2 * a.Wp * a.Tp;
    const float* pxi = pxr + (long)a.Wp * a.Tp;
    for (int e = threadIdx.x; e < np * T; e += blockDim.x) {
        const int p = e / T, t = e - p * T;
        const int w = w0 + p;
        const int qx = (w + a.pad_w) * a.Tp + t + a.pad_t;
        const float xr = pxr[qx] * xsr + xmr, xi = pxi[qx] * xsi + xmi;      // norm_unet.py:71-74
        const long nyf = (long)b * W + w;
        float ymr = 0.f, ysr = 1.f, ymi = 0.f, ysi = 1.f;
        if (a.syf) { const float* sy = a.syf + nyf * 4; ymr = sy[0]; ysr = sy[1]; ymi = sy[2]; ysi = sy[3]; }
        const float* pyr = a.pyf + nyf * 2 * a.Hp * a.Tp;
        const float* pyi = pyr + (long)a.Hp * a.Tp;
        const int qy = (h + a.pad_h) * a.Tp + t + a.pad_t;
        const float yr = pyr[qy] * ysr + ymr, yi = pyi[qy] * ysi + ymi;
        buf[t * kPix + p] = mk(0.5f * (xr + yr), 0.5f * (xi + yi));          // varnet.py:232
    }
    __syncthreads();
    const int s_in = (T + 1) / 2, s_out = T / 2;
    const long HW = (long)H * W;
    for (int e = threadIdx.x; e < T * np; e += blockDim.x) {
        const int i = e / np, p = e - i * np;       // lanes over pixels: coalesced stores
        cf r;
        if (a.xf) {                                  // ifft1c over t, varnet.py:234-238
            int k = i - s_out; if (k < 0) k += T;
            float ax = 0.f, ay = 0.f;
            int idx = (s_in * k) % T;
            for (int g = 0; g < T; ++g) {
                const cf w = tw[idx];
                const cf x = buf[g * kPix + p];
                ax += x.x * w.x + x.y * w.y; ay += x.y * w.x - x.x * w.y;     // x * conj(w)
                idx += k; if (idx >= T) idx -= T;
            }
            r = mk(ax, ay);
        } else {
            r = buf[i * kPix + p];
        }
        const long pix = (long)h * W + w0 + p;
        const cf m = a.mean_img[(long)b * HW + pix];
        a.out[((long)b * T + i) * HW + pix] = cadd(r, m);                    // varnet.py:241
    }
}

// ---------------------------------------------------------------- sens-map prologue
// mean over frames of the rows kept by mask_center (varnet.py:71, transforms.py:95-108)
// (win != NULL: the window {lo, hi} is read from device memory -- cine_acs_window -- so that a forward pass needs no host read-back of the mask)
__global__ void time_mean_center_kernel(const cf* k, cf* out, int T, int C, int H, int W, int lo, int hi, const int* __restrict__ win) {
    if (win) { lo = win[0]; hi = win[1]; }
    const long HW = (long)H * W;
    const long total = (long)C * HW;                // per batch element
    const int b = blockIdx.y;
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
        const int c = (int)(e / HW);
        const long pix = e - (long)c * HW;
        const int h = (int)(pix / W);
        cf r = mk(0.f, 0.f);
        if (h >= lo && h < hi) {
            float sx = 0.f, sy = 0.f;
            for (int t = 0; t < T; ++t) {
                const cf v = k[(((long)b * T + t) * C + c) * HW + pix];
                sx += v.x; sy += v.y;
            }
            r = mk(sx / T, sy / T);
        }
        out[((long)b * C + c) * HW + pix] = r;
    }
}

// x / rss_complex(x, coil dim) (varnet.py:58-59)
__global__ void rss_normalise_kernel(cf* x, int C, long HW) {
    const int b = blockIdx.y;
    for (long p = (long)blockIdx.x * blockDim.x + threadIdx.x; p < HW; p += (long)gridDim.x * blockDim.x) {
        float s = 0.f;
        for (int c = 0; c < C; ++c) {
            const cf v = x[((long)b * C + c) * HW + p];
            s += v.x * v.x + v.y * v.y;
        }
        const float r = sqrtf(s);
        for (int c = 0; c < C; ++c) {
            cf v = x[((long)b * C + c) * HW + p];
            x[((long)b * C + c) * HW + p] = mk(v.x / r, v.y / r);
        }
    }
}

__global__ void complex_abs_kernel(const cf* x, float* y, long n) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const cf v = x[i];
        y[i] = sqrtf(v.x * v.x + v.y * v.y);
    }
}

static unsigned grid_for(long n, int threads, long cap = 4096) {
    long g = ceil_div(n, (long)threads);
    if (g > cap) g = cap;
    if (g < 1) g = 1;
    return (unsigned)g;
}

}  // namespace cine

using namespace cine;

static void pad_split(int n, int& np, int& lo, bool norm = true) {
    np = norm ? cine_pad16(n) : n;    // plain planes are not padded (cinenet.py:194-195, 242)
    lo = (np - n) / 2;                // floor on the left/top, ceil on the right/bottom (norm_unet.py:82-83)
}

extern "C" int cine_normunet_pack(const float* x, float* planes, float* stats, int n, int h, int w, int norm, void* stream) {
    CINE_REQUIRE(x && planes && (stats || !norm), CINE_EINVAL, "cine_normunet_pack: null pointer");
    CINE_REQUIRE(n > 0 && h > 0 && w > 0 && (long)h * w > 1, CINE_EINVAL, "cine_normunet_pack: bad sizes");
    PackArgs a{};
    a.x = x; a.planes = planes; a.stats = stats; a.n = n; a.I = h; a.J = w; a.norm = norm != 0;
    pad_split(h, a.Ip, a.pad_i, a.norm); pad_split(w, a.Jp, a.pad_j, a.norm);
    a.ninner = 1; a.s_outer = (long)h * w * 2; a.s_inner = 0; a.si = (long)w * 2; a.sj = 2;
    ProfScope prof(F_PACK, as_stream(stream));
    return launch_pack(a, nullptr, as_stream(stream));
}

extern "C" int cine_normunet_unpack(const float* planes, const float* stats, float* y, int n, int h, int w, void* stream) {
    CINE_REQUIRE(planes && y, CINE_EINVAL, "cine_normunet_unpack: null pointer");
    CINE_REQUIRE(n > 0 && n <= 65535 && h > 0 && w > 0, CINE_EINVAL, "cine_normunet_unpack: bad sizes");
    int hp, ph, wp, pw;
    pad_split(h, hp, ph, stats != nullptr); pad_split(w, wp, pw, stats != nullptr);
    ProfScope prof(F_PACK, as_stream(stream));
    hipLaunchKernelGGL(normunet_unpack_kernel, dim3(grid_for((long)h * w, 256, 64), n), dim3(256), 0, as_stream(stream),
                       planes, stats, y, h, w, hp, wp, ph, pw);
    return check_launch("normunet_unpack_kernel");
}

extern "C" size_t cine_xfyf_ws_bytes(int b, int t, int h, int w) {
    return (size_t)b * t * h * w * 2 * sizeof(float);
}

extern "C" int cine_xfyf_pack(const float* img, float* planes_xf, float* planes_yf, float* stats_xf, float* stats_yf,
                              float* mean_img, int b, int t, int h, int w, int xf, int norm, void* ws, size_t ws_bytes, void* stream) {
    CINE_REQUIRE(img && planes_xf && planes_yf && mean_img && ws && (!norm || (stats_xf && stats_yf)), CINE_EINVAL,
                 "cine_xfyf_pack: null pointer");
    CINE_REQUIRE(b > 0 && t > 1 && h > 0 && w > 0, CINE_EINVAL, "cine_xfyf_pack: bad sizes");
    CINE_REQUIRE(t <= 64, CINE_EUNSUPPORTED, "cine_xfyf_pack: %d frames > 64", t);
    CINE_REQUIRE(ws_bytes >= cine_xfyf_ws_bytes(b, t, h, w), CINE_EWORKSPACE, "cine_xfyf_pack: workspace too small");
    hipStream_t st = as_stream(stream);
    cf* X = reinterpret_cast<cf*>(ws);          // [b][h][w][t]
    const long HW = (long)h * w;
    const size_t lds = ((size_t)t * kPix + t) * sizeof(cf);
    ProfScope prof(F_PACK, st);
    hipLaunchKernelGGL(temporal_fwd_kernel, dim3((unsigned)ceil_div(HW, (long)kPix), b), dim3(256), lds, st,
                       reinterpret_cast<const cf*>(img), X, reinterpret_cast<cf*>(mean_img), t, HW, xf);
    if (int e = check_launch("temporal_fwd_kernel")) return e;
    // xf planes: sample (b, h), rows = w, cols = t          (varnet.py:216)
    PackArgs a{};
    a.x = reinterpret_cast<const float*>(X); a.planes = planes_xf; a.stats = stats_xf;
    a.n = b * h; a.I = w; a.J = t; a.norm = norm != 0;
    pad_split(w, a.Ip, a.pad_i, a.norm); pad_split(t, a.Jp, a.pad_j, a.norm);
    a.ninner = h; a.s_outer = HW * t * 2; a.s_inner = (long)w * t * 2; a.si = (long)t * 2; a.sj = 2;
    const PackArgs axf = a;
    // yf planes: sample (b, w), rows = h, cols = t          (varnet.py:217)
    a.planes = planes_yf; a.stats = stats_yf;
    a.n = b * w; a.I = h; a.J = t;
    pad_split(h, a.Ip, a.pad_i, a.norm); pad_split(t, a.Jp, a.pad_j, a.norm);
    a.ninner = w; a.s_outer = HW * t * 2; a.s_inner = (long)t * 2; a.si = (long)w * t * 2; a.sj = 2;
    return launch_pack(axf, &a, st);
}

extern "C" int cine_xfyf_unpack(const float* planes_xf, const float* planes_yf, const float* stats_xf,
                                const float* stats_yf, const float* mean_img, float* out,
                                int b, int t, int h, int w, int xf, void* stream) {
    CINE_REQUIRE(planes_xf && planes_yf && mean_img && out && ((stats_xf != nullptr) == (stats_yf != nullptr)), CINE_EINVAL,
                 "cine_xfyf_unpack: null pointer");
    CINE_REQUIRE(b > 0 && t > 1 && t <= 64 && h > 0 && w > 0 && h <= 65535 && b <= 65535, CINE_EINVAL,
                 "cine_xfyf_unpack: bad sizes");
    UnpackArgs a{};
    a.pxf = planes_xf; a.pyf = planes_yf; a.sxf = stats_xf; a.syf = stats_yf;
    a.mean_img = reinterpret_cast<const cf*>(mean_img); a.out = reinterpret_cast<cf*>(out);
    a.T = t; a.H = h; a.W = w; a.xf = xf;
    const bool nrm = stats_xf != nullptr;     // plain planes (no stats) are unpadded
    pad_split(t, a.Tp, a.pad_t, nrm); pad_split(h, a.Hp, a.pad_h, nrm); pad_split(w, a.Wp, a.pad_w, nrm);
    const size_t lds = ((size_t)t * kPix + t) * sizeof(cf);
    ProfScope prof(F_PACK, as_stream(stream));
    hipLaunchKernelGGL(xfyf_unpack_kernel, dim3(ceil_div(w, kPix), h, b), dim3(256), lds, as_stream(stream), a);
    return check_launch("xfyf_unpack_kernel");
}

static int sens_prologue_impl(const float* k, float* out, int b, int t, int c, int h, int w, int row_lo, int row_hi, const int* win, void* stream) {
    CINE_REQUIRE(k && out, CINE_EINVAL, "cine_sens_prologue: null pointer");
    CINE_REQUIRE(b > 0 && b <= 65535 && t > 0 && c > 0 && h > 0 && w > 0, CINE_EINVAL, "cine_sens_prologue: bad sizes");
    { ProfScope prof(F_MISC, as_stream(stream));
    hipLaunchKernelGGL(time_mean_center_kernel, dim3(grid_for((long)c * h * w, 256), b), dim3(256), 0, as_stream(stream),
                       reinterpret_cast<const cf*>(k), reinterpret_cast<cf*>(out), t, c, h, w, row_lo, row_hi, win); }
    if (int e = check_launch("time_mean_center_kernel")) return e;
    return cine_fft2c(out, out, b * c, h, w, 1, stream);
}
extern "C" int cine_sens_prologue(const float* k, float* out, int b, int t, int c, int h, int w,
                                  int row_lo, int row_hi, void* stream) {
    return sens_prologue_impl(k, out, b, t, c, h, w, row_lo, row_hi, nullptr, stream);
}
// The same with the window {row_lo, row_hi} in DEVICE memory (cine_acs_window wrote it): no host read-back of the mask between the caller and the launch.
extern "C" int cine_sens_prologue_win(const float* k, float* out, int b, int t, int c, int h, int w, const int* window, void* stream) {
    CINE_REQUIRE(window, CINE_EINVAL, "cine_sens_prologue_win: null window");
    return sens_prologue_impl(k, out, b, t, c, h, w, 0, 0, window, stream);
}

// The fully sampled centre rows of a row mask, found on the device (reference varnet.py:64-68 reads the mask on the host): rows = the 1-D
// pattern of frame 0 (n >= h entries, 0 = not sampled), cent = h / 2, left = the last unsampled row below cent (-1: none), right = the first
// one at or above it (n: none), n_low = right - left, pad = (h - n_low + 1) / 2; window = {pad, pad + n_low}.
namespace cine {
__global__ __launch_bounds__(256) void acs_window_kernel(const float* __restrict__ rows, int n, int h, int* __restrict__ win) {
    __shared__ int sl[4], sr[4];
    const int cent = h / 2;
    n = min(n, h);                  // frame 0 of batch element 0 only (varnet.py:64-68 is written for batch 1; later rows must not supply `right`)
    int l = -1, r = n;
    for (int i = threadIdx.x; i < n; i += 256)
        if (rows[i] == 0.f) { if (i < cent) l = max(l, i); else r = min(r, i); }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { l = max(l, __shfl_xor(l, o, 64)); r = min(r, __shfl_xor(r, o, 64)); }
    if ((threadIdx.x & 63) == 0) { sl[threadIdx.x >> 6] = l; sr[threadIdx.x >> 6] = r; }
    __syncthreads();
    if (threadIdx.x == 0) {
        l = max(max(sl[0], sl[1]), max(sl[2], sl[3])); r = min(min(sr[0], sr[1]), min(sr[2], sr[3]));
        // no unsampled row below / above the centre (a fully sampled mask): the reference's nonzero(...)[-1] raises there; a kernel cannot,
        // so the window is every row -- mask_center then keeps the whole k-space, which is what "all of it is calibration data" means
        if (l < 0 || r >= h) { win[0] = 0; win[1] = h; return; }
        const int n_low = r - l, pad = (h - n_low + 1) / 2;
        win[0] = max(pad, 0); win[1] = min(pad + n_low, h);
    }
}
}  // namespace cine
extern "C" int cine_acs_window(const float* mask_rows, int n, int h, int* window, void* stream) {
    CINE_REQUIRE(mask_rows && window && h > 0 && n >= h, CINE_EINVAL, "cine_acs_window: bad arguments");
    ProfScope prof(F_MISC, as_stream(stream));
    hipLaunchKernelGGL(cine::acs_window_kernel, dim3(1), dim3(256), 0, as_stream(stream), mask_rows, n, h, window);
    return check_launch("acs_window_kernel");
}

extern "C" int cine_rss_normalise(float* x, int b, int c, int h, int w, void* stream) {
    CINE_REQUIRE(x, CINE_EINVAL, "cine_rss_normalise: null pointer");
    CINE_REQUIRE(b > 0 && b <= 65535 && c > 0 && h > 0 && w > 0, CINE_EINVAL, "cine_rss_normalise: bad sizes");
    ProfScope prof(F_MISC, as_stream(stream));
    hipLaunchKernelGGL(rss_normalise_kernel, dim3(grid_for((long)h * w, 256), b), dim3(256), 0, as_stream(stream),
                       reinterpret_cast<cf*>(x), c, (long)h * w);
    return check_launch("rss_normalise_kernel");
}

extern "C" int cine_complex_abs(const float* x, float* y, long n, void* stream) {
    CINE_REQUIRE(x && y && n >= 0, CINE_EINVAL, "cine_complex_abs: bad arguments");
    if (n == 0) return CINE_OK;
    ProfScope prof(F_MISC, as_stream(stream));
    hipLaunchKernelGGL(complex_abs_kernel, dim3(grid_for(n, 256)), dim3(256), 0, as_stream(stream),
                       reinterpret_cast<const cf*>(x), y, n);
    return check_launch("complex_abs_kernel");
}

// ---------------------------------------------------------------- CG vector ops (cinenet.py:136-171)
// Scalars stay in device memory (the reference pulls them to the host with .item(), :159-169).
namespace cine {
constexpr int kDotBlocks = 256;
__global__ void dot_partial_kernel(const float* a, const float* b, long n, float* part) {
    __shared__ float red[16];
    float s = 0.f;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) s += a[i] * b[i];
    s = block_sum(s, red);
    if (threadIdx.x == 0) part[blockIdx.x] = s;
}
__global__ void dot_final_kernel(const float* part, int np, float* out) {
    __shared__ float red[16];
    float s = 0.f;
    for (int i = threadIdx.x; i < np; i += blockDim.x) s += part[i];
    s = block_sum(s, red);
    if (threadIdx.x == 0) *out = s;
}
// out = a + sign * (num / den) * b      (den == nullptr -> 1; num == nullptr -> softplus(*lam))
__global__ void axpby_dev_kernel(float* out, const float* a, const float* b, long n, const float* num, const float* den,
                                 const float* lam, float sign) {
    float s;
    if (num) s = den ? *num / *den : *num;
    else { const float l = *lam; s = l > 20.f ? l : log1pf(expf(l)); }
    s *= sign;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) out[i] = a[i] + s * b[i];
}
// One conjugate-gradient iteration after d = H p (reference cinenet.py:155-169) in three launches instead of eight: the p.d partial
// sums (dot_partial_kernel), then cg_update_kernel -- every workgroup adds the 256 partials itself (same order as dot_final_kernel),
// alpha = rr_old / (p.d), x += alpha p, r -= alpha d, partial sums of r.r -- then cg_direction_kernel: rr_new from the partials,
// beta = rr_new / rr_old, p = r + beta p.  Same arithmetic and summation orders as cine_dot / cine_axpby_dev: bit-identical.
__global__ __launch_bounds__(256) void cg_update_kernel(float* x, float* r, const float* p, const float* d, long n,
                                                        const float* pd_part, const float* rr_old, float* rr_part, float* pd_out) {
    __shared__ float red[16];
    const float pd = block_sum(pd_part[threadIdx.x], red);
    if (pd_out && blockIdx.x == 0 && threadIdx.x == 0) *pd_out = pd;          // recorded for the adjoint recurrence (training)
    const float alpha = *rr_old / pd;
    const float nalpha = alpha * -1.0f;
    float s = 0.f;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        x[i] = x[i] + alpha * p[i];
        const float rn = r[i] + nalpha * d[i];
        r[i] = rn;
        s += rn * rn;
    }
    __syncthreads();
    s = block_sum(s, red);
    if (threadIdx.x == 0) rr_part[blockIdx.x] = s;
}
__global__ __launch_bounds__(256) void cg_direction_kernel(float* p, const float* r, long n, const float* rr_part, const float* rr_old,
                                                           float* rr_new) {
    __shared__ float red[16];
    const float rn = block_sum(rr_part[threadIdx.x], red);
    const float beta = rn / *rr_old;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) p[i] = r[i] + beta * p[i];
    if (blockIdx.x == 0 && threadIdx.x == 0) *rr_new = rn;
}
}  // namespace cine

// cg_update_kernel with the operator's last pass folded in (cine_normal_op_cg_fused, fft_kernels.hip): d = sum_z partial_z + softplus(lambda) p
// formed on the fly with imgdc_sum_kernel's arithmetic, p.d from the operator kernel's per-workgroup partial sums (each thread adds a strided
// share in index order, then the block sum: deterministic)
namespace cine {
typedef float2 cfp;
__global__ __launch_bounds__(256) void cg_update_fused_kernel(cfp* x, cfp* r, const cfp* p, const cfp* partial, int nz, long part_stride,
                                                              const float* lam, long ncf, const float* pd_wg, int npd,
                                                              const float* rr_old, float* rr_part, float* pd_out) {
    __shared__ float red[16];
    float acc = 0.f;
    for (int i = threadIdx.x; i < npd; i += 256) acc += pd_wg[i];
    const float pd = block_sum(acc, red);
    if (pd_out && blockIdx.x == 0 && threadIdx.x == 0) *pd_out = pd;
    const float alpha = *rr_old / pd;
    const float nalpha = alpha * -1.0f;
    const float l = *lam;
    const float beta = l > 20.f ? l : log1pf(expf(l));                 // softplus, as imgdc_weights reads it
    float s = 0.f;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < ncf; i += (long)gridDim.x * blockDim.x) {
        cfp d = partial[i];
        for (int z = 1; z < nz; ++z) { const cfp u = partial[z * part_stride + i]; d.x += u.x; d.y += u.y; }
        const cfp pv = p[i];
        d.x = fmaf(beta, pv.x, d.x); d.y = fmaf(beta, pv.y, d.y);
        cfp xv = x[i], rv = r[i];
        xv.x = xv.x + alpha * pv.x; xv.y = xv.y + alpha * pv.y;
        rv.x = rv.x + nalpha * d.x; rv.y = rv.y + nalpha * d.y;
        x[i] = xv; r[i] = rv;
        s += rv.x * rv.x; s += rv.y * rv.y;
    }
    __syncthreads();
    s = block_sum(s, red);
    if (threadIdx.x == 0) rr_part[blockIdx.x] = s;
}
// cine_conj_grad (fft_kernels.hip): the set-up and update kernels of the two-launch conjugate-gradient iteration.  The residual lives
// next to the direction it was computed with, as {p.x, p.y, r.x, r.y}: the next operator forms p_new = r + beta p on load from ONE element.
// rhs_ref != 0: `rhs` holds x_ref and the right-hand side is x_ref + softplus(lambda) x (cinenet.py:106-107 with x = the regulariser's
// output, which is also the start value: cine_axpby_dev's expression, one launch less per DC block)
__global__ __launch_bounds__(256) void cg_init_kernel(const cfp* x, const cfp* rhs, int rhs_ref, const cfp* partial, int nz, long part_stride, const float* lam,
                                                      long ncf, float4* pr, float* rr_part) {
    __shared__ float red[16];
    const float l = *lam;
    const float beta = l > 20.f ? l : log1pf(expf(l));                 // softplus, as imgdc_weights reads it
    float s = 0.f;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < ncf; i += (long)gridDim.x * blockDim.x) {
        cfp d = partial[i];
        for (int z = 1; z < nz; ++z) { const cfp u = partial[z * part_stride + i]; d.x += u.x; d.y += u.y; }
        const cfp xv = x[i];
        cfp bv = rhs[i];
        if (rhs_ref) { bv.x = bv.x + beta * xv.x; bv.y = bv.y + beta * xv.y; }
        d.x = fmaf(beta, xv.x, d.x); d.y = fmaf(beta, xv.y, d.y);        // H x0 (imgdc_sum_kernel's arithmetic)
        const float rx = bv.x + -1.0f * d.x, ry = bv.y + -1.0f * d.y;    // b - H x0 (cine_axpby_dev's expression)
        pr[i] = make_float4(0.f, 0.f, rx, ry);
        s += rx * rx; s += ry * ry;
    }
    s = block_sum(s, red);
    if (threadIdx.x == 0) rr_part[blockIdx.x] = s;
}
__global__ __launch_bounds__(256) void cg_update2_kernel(cfp* x, float4* pr, const cfp* p, const cfp* partial, int nz, long part_stride,
                                                         const float* lam, long ncf, const float* pd_wg, int npd,
                                                         const float* rr_prev, float* rr_cur, int last, float* rr_rec, float* pd_rec) {
    __shared__ float red[16];
    float acc = 0.f;
    for (int i = threadIdx.x; i < npd; i += 256) acc += pd_wg[i];
    const float pd = block_sum(acc, red);
    const float rr_old = block_sum(rr_prev[threadIdx.x], red);
    if (rr_rec && blockIdx.x == 0 && threadIdx.x == 0) { *rr_rec = rr_old; *pd_rec = pd; }      // training: the step sizes of this iteration for the adjoint recurrence
    const float alpha = rr_old / pd;
    const float nalpha = alpha * -1.0f;
    const float l = *lam;
    const float beta = l > 20.f ? l : log1pf(expf(l));
    float s = 0.f;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < ncf; i += (long)gridDim.x * blockDim.x) {
        cfp d = partial[i];
        for (int z = 1; z < nz; ++z) { const cfp u = partial[z * part_stride + i]; d.x += u.x; d.y += u.y; }
        const cfp pv = p[i];
        d.x = fmaf(beta, pv.x, d.x); d.y = fmaf(beta, pv.y, d.y);
        cfp xv = x[i];
        const float4 old = pr[i];
        xv.x = xv.x + alpha * pv.x; xv.y = xv.y + alpha * pv.y;
        const float rx = old.z + nalpha * d.x, ry = old.w + nalpha * d.y;
        x[i] = xv;
        if (!last) pr[i] = make_float4(pv.x, pv.y, rx, ry);
        s += rx * rx; s += ry * ry;
    }
    __syncthreads();
    s = block_sum(s, red);
    if (threadIdx.x == 0) rr_cur[blockIdx.x] = s;
}
int launch_cg_init(const float* x, const float* rhs, int rhs_ref, const float2* partial, int nz, long part_stride, const float* lam, long ncf,
                   float4* pr, float* rr_part, hipStream_t st) {
    ProfScope prof(F_MISC, st);
    hipLaunchKernelGGL(cg_init_kernel, dim3(kDotBlocks), dim3(256), 0, st, reinterpret_cast<const cfp*>(x), reinterpret_cast<const cfp*>(rhs), rhs_ref,
                       partial, nz, part_stride, lam, ncf, pr, rr_part);
    return check_launch("cg_init_kernel");
}
int launch_cg_update2(float* x, float4* pr, const float2* p, const float2* partial, int nz, long part_stride, const float* lam, long ncf,
                      const float* pd_wg, int npd, const float* rr_prev, float* rr_cur, int last, hipStream_t st, float* rr_rec, float* pd_rec, float* rr_final) {
    ProfScope prof(F_MISC, st);
    hipLaunchKernelGGL(cg_update2_kernel, dim3(kDotBlocks), dim3(256), 0, st, reinterpret_cast<cfp*>(x), pr, p, partial, nz, part_stride, lam, ncf,
                       pd_wg, npd, rr_prev, rr_cur, last, rr_rec, pd_rec);
    if (rr_final) hipLaunchKernelGGL(dot_final_kernel, dim3(1), dim3(256), 0, st, rr_cur, kDotBlocks, rr_final);      // r.r behind the last iteration
    return check_launch("cg_update2_kernel");
}
int launch_cg_update_fused(float* x, float* r, float* p, const float2* partial, int nz, long part_stride, const float* lam, long ncf,
                           const float* pd_wg, int npd, const float* rr_old, float* rr_new, float* rr_part, float* pd_out, hipStream_t st) {
    ProfScope prof(F_MISC, st);
    hipLaunchKernelGGL(cg_update_fused_kernel, dim3(kDotBlocks), dim3(256), 0, st, reinterpret_cast<cfp*>(x), reinterpret_cast<cfp*>(r),
                       reinterpret_cast<const cfp*>(p), partial, nz, part_stride, lam, ncf, pd_wg, npd, rr_old, rr_part, pd_out);
    hipLaunchKernelGGL(cg_direction_kernel, dim3(kDotBlocks), dim3(256), 0, st, p, r, 2 * ncf, rr_part, rr_old, rr_new);
    return check_launch("cine_normal_op_cg_fused");
}
}  // namespace cine

extern "C" size_t cine_dot_ws_bytes(void) { return kDotBlocks * sizeof(float); }
extern "C" size_t cine_cg_ws_bytes(void) { return 2 * kDotBlocks * sizeof(float); }

extern "C" int cine_cg_step(float* x, float* r, float* p, const float* d, long n, const float* rr_old_dev, float* rr_new_dev,
                            void* ws, void* stream) {
    CINE_REQUIRE(x && r && p && d && rr_old_dev && rr_new_dev && ws && n > 0, CINE_EINVAL, "cine_cg_step: bad arguments");
    CINE_REQUIRE(rr_old_dev != rr_new_dev, CINE_EINVAL, "cine_cg_step: rr_old and rr_new must be different scalars");
    static_assert(kDotBlocks == 256, "the fused kernels add one partial per thread");
    hipStream_t st = as_stream(stream);
    float* part = reinterpret_cast<float*>(ws);
    ProfScope prof(F_MISC, st);
    hipLaunchKernelGGL(dot_partial_kernel, dim3(kDotBlocks), dim3(256), 0, st, p, d, n, part);
    hipLaunchKernelGGL(cg_update_kernel, dim3(kDotBlocks), dim3(256), 0, st, x, r, p, d, n, part, rr_old_dev, part + kDotBlocks, static_cast<float*>(nullptr));
    hipLaunchKernelGGL(cg_direction_kernel, dim3(kDotBlocks), dim3(256), 0, st, p, r, n, part + kDotBlocks, rr_old_dev, rr_new_dev);
    return check_launch("cine_cg_step");
}

// cine_cg_step without its first launch: ws[0..256) already holds the partial sums of p.d (cine_normal_op_pd)
extern "C" int cine_cg_step_pd(float* x, float* r, float* p, const float* d, long n, const float* rr_old_dev, float* rr_new_dev,
                               void* ws, void* stream) {
    CINE_REQUIRE(x && r && p && d && rr_old_dev && rr_new_dev && ws && n > 0, CINE_EINVAL, "cine_cg_step_pd: bad arguments");
    CINE_REQUIRE(rr_old_dev != rr_new_dev, CINE_EINVAL, "cine_cg_step_pd: rr_old and rr_new must be different scalars");
    hipStream_t st = as_stream(stream);
    float* part = reinterpret_cast<float*>(ws);
    ProfScope prof(F_MISC, st);
    hipLaunchKernelGGL(cg_update_kernel, dim3(kDotBlocks), dim3(256), 0, st, x, r, p, d, n, part, rr_old_dev, part + kDotBlocks, static_cast<float*>(nullptr));
    hipLaunchKernelGGL(cg_direction_kernel, dim3(kDotBlocks), dim3(256), 0, st, p, r, n, part + kDotBlocks, rr_old_dev, rr_new_dev);
    return check_launch("cine_cg_step_pd");
}

// cine_cg_step_pd that also records p.d (training: the adjoint recurrence needs alpha_k = rr_k / pd_k)
extern "C" int cine_cg_step_pd2(float* x, float* r, float* p, const float* d, long n, const float* rr_old_dev, float* rr_new_dev,
                                float* pd_out_dev, void* ws, void* stream) {
    CINE_REQUIRE(x && r && p && d && rr_old_dev && rr_new_dev && pd_out_dev && ws && n > 0, CINE_EINVAL, "cine_cg_step_pd2: bad arguments");
    CINE_REQUIRE(rr_old_dev != rr_new_dev, CINE_EINVAL, "cine_cg_step_pd2: rr_old and rr_new must be different scalars");
    hipStream_t st = as_stream(stream);
    float* part = reinterpret_cast<float*>(ws);
    ProfScope prof(F_MISC, st);
    hipLaunchKernelGGL(cg_update_kernel, dim3(kDotBlocks), dim3(256), 0, st, x, r, p, d, n, part, rr_old_dev, part + kDotBlocks, pd_out_dev);
    hipLaunchKernelGGL(cg_direction_kernel, dim3(kDotBlocks), dim3(256), 0, st, p, r, n, part + kDotBlocks, rr_old_dev, rr_new_dev);
    return check_launch("cine_cg_step_pd2");
}

// ---------------------------------------------------------------- adjoint of the conjugate-gradient iteration (training, cinenet.py:136-171)
// The reference takes alpha_k and beta_k out of the graph (.item()), so the K iterations are linear in (x0, b) with recorded step sizes and
// their adjoint runs the same operator backwards (cine_hip/autograd.py ConjGradFn).  With q_k = gr_{k+1} + gp_{k+1} (the gradient reaching
// r_{k+1}) and hg = H(q_k), one reverse step is
//     gp_k = beta_k gp_{k+1} + alpha_k gx - alpha_k hg,    q_{k-1} = q_k + gp_k,    s_k = <q_k, p_k>
// in ONE launch (the axpby / dot / add sequence it replaces is 9 launches): the same fused-multiply-add chain as three cine_axpby_dev calls
// and the same partial-sum pattern as cine_dot, so the values are bit-identical to that sequence.
namespace cine {
__global__ __launch_bounds__(256) void cg_adjoint_step_kernel(float* gp, float* q, const float* gx, const float* hg, const float* pk, long n,
                                                              const float* rr, const float* pd, const float* rr_new, float* part) {
    __shared__ float red[16];
    const float beta = *rr_new / *rr, alpha = *rr / *pd;
    const float nalpha = alpha * -1.0f;
    float s = 0.f;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const float qi = q[i];
        s += qi * pk[i];
        const float t1 = 0.f + beta * gp[i];
        const float t2 = t1 + alpha * gx[i];
        const float g = t2 + nalpha * hg[i];
        gp[i] = g;
        q[i] = qi + g;
    }
    s = block_sum(s, red);
    if (threadIdx.x == 0) part[blockIdx.x] = s;
}
// gv = - sum_k alpha_k s_k, accumulated from the last iteration to the first like the host loop it replaces
__global__ __launch_bounds__(256) void cg_adjoint_finish_kernel(const float* part, const float* rr, const float* pd, int iters, float* gv) {
    __shared__ float red[16];
    float acc = 0.f;
    for (int k = iters - 1; k >= 0; --k) {
        const float sk = block_sum(part[(long)k * kDotBlocks + threadIdx.x], red);
        acc = acc - sk * (rr[k] / pd[k]);
    }
    if (threadIdx.x == 0) *gv = acc;
}
}  // namespace cine

extern "C" int cine_cg_adjoint_step(float* gp, float* q, const float* gx, const float* hg, const float* pk, long n, const float* rr_dev,
                                    const float* pd_dev, const float* rr_new_dev, float* part, void* stream) {
    CINE_REQUIRE(gp && q && gx && hg && pk && rr_dev && pd_dev && rr_new_dev && part && n > 0, CINE_EINVAL, "cine_cg_adjoint_step: bad arguments");
    hipStream_t st = as_stream(stream);
    ProfScope prof(F_MISC, st);
    hipLaunchKernelGGL(cg_adjoint_step_kernel, dim3(kDotBlocks), dim3(256), 0, st, gp, q, gx, hg, pk, n, rr_dev, pd_dev, rr_new_dev, part);
    return check_launch("cg_adjoint_step_kernel");
}
extern "C" size_t cine_cg_adjoint_part_floats(void) { return kDotBlocks; }
extern "C" int cine_cg_adjoint_finish(const float* part, const float* rr_dev, const float* pd_dev, int iters, float* gv_dev, void* stream) {
    CINE_REQUIRE(part && rr_dev && pd_dev && gv_dev && iters >= 0, CINE_EINVAL, "cine_cg_adjoint_finish: bad arguments");
    hipStream_t st = as_stream(stream);
    ProfScope prof(F_MISC, st);
    hipLaunchKernelGGL(cg_adjoint_finish_kernel, dim3(1), dim3(256), 0, st, part, rr_dev, pd_dev, iters, gv_dev);
    return check_launch("cg_adjoint_finish_kernel");
}

extern "C" int cine_dot(const float* a, const float* b, long n, float* out_dev, void* ws, void* stream) {
    CINE_REQUIRE(a && b && out_dev && ws && n > 0, CINE_EINVAL, "cine_dot: bad arguments");
    hipStream_t st = as_stream(stream);
    ProfScope prof(F_MISC, st);
    hipLaunchKernelGGL(dot_partial_kernel, dim3(kDotBlocks), dim3(256), 0, st, a, b, n, reinterpret_cast<float*>(ws));
    hipLaunchKernelGGL(dot_final_kernel, dim3(1), dim3(256), 0, st, reinterpret_cast<const float*>(ws), kDotBlocks, out_dev);
    return check_launch("cine_dot");
}

extern "C" int cine_axpby_dev(float* out, const float* a, const float* b, long n, const float* num_dev, const float* den_dev,
                              const float* lambda_dev, float sign, void* stream) {
    CINE_REQUIRE(out && a && b && n > 0 && (num_dev || lambda_dev), CINE_EINVAL, "cine_axpby_dev: bad arguments");
    ProfScope prof(F_MISC, as_stream(stream));
    hipLaunchKernelGGL(axpby_dev_kernel, dim3(grid_for(n, 256, 2048)), dim3(256), 0, as_stream(stream), out, a, b, n,
                       num_dev, den_dev, lambda_dev, sign);
    return check_launch("axpby_dev_kernel");
}

// ---------------------------------------------------------------- NormUnet3D front / back halves (norm_unet.py:149-219)
namespace cine {
// one workgroup per sample: x (n, T, H, W, 2) -> planes (n, 2, Tp, Hp, Wp), stats (n, 2, 2)
__global__ void normunet3d_pack_kernel(const float* x, float* planes, float* stats, int T, int H, int W,
                                       int Tp, int Hp, int Wp, int pt, int ph, int pw, int norm) {
    __shared__ float red[16];
    const int n = blockIdx.x;
    const long cnt = (long)T * H * W;
    const float2* src = reinterpret_cast<const float2*>(x) + (long)n * cnt;
    float mr = 0.f, mi = 0.f, sdr = 1.f, sdi = 1.f;
    if (norm) {
        float sr = 0.f, si_ = 0.f;
        for (long e = threadIdx.x; e < cnt; e += blockDim.x) { const float2 v = src[e]; sr += v.x; si_ += v.y; }
        mr = block_sum(sr, red) / cnt; mi = block_sum(si_, red) / cnt;
        float qr = 0.f, qi = 0.f;
        for (long e = threadIdx.x; e < cnt; e += blockDim.x) {
            const float2 v = src[e];
            qr += (v.x - mr) * (v.x - mr); qi += (v.y - mi) * (v.y - mi);
        }
        sdr = sqrtf(block_sum(qr, red) / (cnt - 1)); sdi = sqrtf(block_sum(qi, red) / (cnt - 1));   // unbiased (:166)
        if (threadIdx.x == 0) { float* st = stats + (long)n * 4; st[0] = mr; st[1] = sdr; st[2] = mi; st[3] = sdi; }
    }
    const long pcnt = (long)Tp * Hp * Wp;
    float* pr = planes + (long)n * 2 * pcnt;
    float* pi = pr + pcnt;
    for (long e = threadIdx.x; e < pcnt; e += blockDim.x) {
        const int wp = (int)(e % Wp); const long r = e / Wp;
        const int hp = (int)(r % Hp), tp = (int)(r / Hp);
        const int t = tp - pt, h = hp - ph, w = wp - pw;
        float vr = 0.f, vi = 0.f;
        if (t >= 0 && t < T && h >= 0 && h < H && w >= 0 && w < W) {
            const float2 v = src[((long)t * H + h) * W + w];
            vr = (v.x - mr) / sdr; vi = (v.y - mi) / sdi;
        }
        pr[e] = vr; pi[e] = vi;
    }
}
// norm == 0 (the bare 3-D U-Net of CineNet, cinenet.py:251-253): no statistics, so the repack is a plain grid-wide gather
// (one workgroup per sample took 460 us for a 15 x 200 x 200 volume)
__global__ __launch_bounds__(256) void normunet3d_repack_kernel(const float* x, float* planes, int T, int H, int W) {
    const int n = blockIdx.y;
    const long cnt = (long)T * H * W;
    const float2* src = reinterpret_cast<const float2*>(x) + (long)n * cnt;
    float* pr = planes + (long)n * 2 * cnt;
    float* pi = pr + cnt;
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < cnt; e += (long)gridDim.x * blockDim.x) {
        const float2 v = src[e];
        pr[e] = v.x; pi[e] = v.y;
    }
}
__global__ void normunet3d_unpack_kernel(const float* planes, const float* stats, float* y, int T, int H, int W,
                                         int Tp, int Hp, int Wp, int pt, int ph, int pw) {
    const int n = blockIdx.y;
    float mr = 0.f, sdr = 1.f, mi = 0.f, sdi = 1.f;
    if (stats) { const float* st = stats + (long)n * 4; mr = st[0]; sdr = st[1]; mi = st[2]; sdi = st[3]; }
    const long cnt = (long)T * H * W, pcnt = (long)Tp * Hp * Wp;
    const float* pr = planes + (long)n * 2 * pcnt;
    const float* pi = pr + pcnt;
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < cnt; e += (long)gridDim.x * blockDim.x) {
        const int w = (int)(e % W); const long r = e / W;
        const int h = (int)(r % H), t = (int)(r / H);
        const long q = ((long)(t + pt) * Hp + h + ph) * Wp + w + pw;
        reinterpret_cast<float2*>(y)[(long)n * cnt + e] = make_float2(pr[q] * sdr + mr, pi[q] * sdi + mi);
    }
}
}  // namespace cine

extern "C" int cine_normunet3d_pack(const float* x, float* planes, float* stats, int n, int t, int h, int w, int norm, void* stream) {
    CINE_REQUIRE(x && planes && (stats || !norm), CINE_EINVAL, "cine_normunet3d_pack: null pointer");
    CINE_REQUIRE(n > 0 && t > 0 && h > 0 && w > 0 && (long)t * h * w > 1, CINE_EINVAL, "cine_normunet3d_pack: bad sizes");
    int tp, pt, hp, ph, wp, pw;
    pad_split(t, tp, pt, norm != 0); pad_split(h, hp, ph, norm != 0); pad_split(w, wp, pw, norm != 0);
    ProfScope prof(F_PACK, as_stream(stream));
    if (!norm) {        // no padding either (pad_split leaves the sizes alone): a straight (re, im) de-interleave
        hipLaunchKernelGGL(normunet3d_repack_kernel, dim3(grid_for((long)t * h * w, 256, 2048), n), dim3(256), 0, as_stream(stream),
                           x, planes, t, h, w);
        return check_launch("normunet3d_repack_kernel");
    }
    hipLaunchKernelGGL(normunet3d_pack_kernel, dim3(n), dim3(1024), 0, as_stream(stream), x, planes, stats, t, h, w,
                       tp, hp, wp, pt, ph, pw, norm);
    return check_launch("normunet3d_pack_kernel");
}

extern "C" int cine_normunet3d_unpack(const float* planes, const float* stats, float* y, int n, int t, int h, int w, void* stream) {
    CINE_REQUIRE(planes && y, CINE_EINVAL, "cine_normunet3d_unpack: null pointer");
    CINE_REQUIRE(n > 0 && n <= 65535 && t > 0 && h > 0 && w > 0, CINE_EINVAL, "cine_normunet3d_unpack: bad sizes");
    int tp, pt, hp, ph, wp, pw;
    const bool nrm = stats != nullptr;
    pad_split(t, tp, pt, nrm); pad_split(h, hp, ph, nrm); pad_split(w, wp, pw, nrm);
    ProfScope prof(F_PACK, as_stream(stream));
    hipLaunchKernelGGL(normunet3d_unpack_kernel, dim3(grid_for((long)t * h * w, 256, 1024), n), dim3(256), 0, as_stream(stream),
                       planes, stats, y, t, h, w, tp, hp, wp, pt, ph, pw);
    return check_launch("normunet3d_unpack_kernel");
}
