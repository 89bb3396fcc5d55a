// train_kernels.hip -- adjoints of the byte-moving steps around the regulariser (training through the HIP path):
// NormUnet group norm / pad / un-norm (denoisers/norm_unet.py:59-96), the XT / XF rotations with the temporal mean and
// centered DFT (models/varnet.py:196-241), the sensitivity normalisation (varnet.py:58-59), complex_abs (utils/math.py:48-62)
// and the coil-sum of sens_reduce with respect to the maps (varnet.py:187-194).  Small HBM-bound kernels; every reduction
// is done in a fixed order (no atomics).
#include <algorithm>
#include "common.h"
#include "fft_core.h"

namespace cine {

__device__ __forceinline__ float wave_sum_t(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float block_sum_t(float v, float* red) {
    v = wave_sum_t(v);
    const int wave = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[wave] = v;
    __syncthreads();
    float s = 0.f;
    for (int i = 0; i < nw; ++i) s += red[i];
    return s;
}

// strided view of sample n, element (i, j) of a complex tensor (see normunet_pack_kernel in pack_kernels.hip)
struct View {
    float* x; int ninner; long s_outer, s_inner, si, sj;
    __device__ __forceinline__ float2* at(int n, int i, int j) const {
        return reinterpret_cast<float2*>(x + (long)(n / ninner) * s_outer + (long)(n % ninner) * s_inner + i * si + j * sj);
    }
};

// ---------------------------------------------------------------- NormUnet back half, adjoint
// forward (norm_unet.py:88-96, 71-74): out[n](i, j) = q[n][ch][i + pad_i][j + pad_j] * std_ch + mean_ch
// given gout: gq = gout * std inside the window, 0 on the pad frame;  dstats[n] = {sum gout_re, sum gout_re q_re, sum gout_im,
// sum gout_im q_im} (the direct gradients of mean and std).  gscale multiplies gout on load (the 0.5 of varnet.py:232).
struct UnpackBwdArgs {
    View gout; const float* q; const float* stats; float* gq; float* dstats;
    int n, I, J, Ip, Jp, pad_i, pad_j; float gscale;
};
struct UnpackBwdArgs2 { UnpackBwdArgs s[2]; };

__global__ __launch_bounds__(256) void normunet_unpack_bwd_kernel(UnpackBwdArgs2 two) {
    __shared__ float red[16];
    const UnpackBwdArgs& a = two.s[blockIdx.y];
    const int n = blockIdx.x;
    if (n >= a.n) return;
    const long pp = (long)a.Ip * a.Jp;
    const float* qr = a.q + (long)n * 2 * pp;
    const float* qi = qr + pp;
    float* gr = a.gq + (long)n * 2 * pp;
    float* gi = gr + pp;
    float sdr = 1.f, sdi = 1.f;
    if (a.stats) { sdr = a.stats[(long)n * 4 + 1]; sdi = a.stats[(long)n * 4 + 3]; }
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    for (int e = threadIdx.x; e < a.Ip * a.Jp; e += blockDim.x) {
        const int ip = e / a.Jp, jp = e - ip * a.Jp;
        const int i = ip - a.pad_i, j = jp - a.pad_j;
        float vr = 0.f, vi = 0.f;
        if (i >= 0 && i < a.I && j >= 0 && j < a.J) {
            const float2 g = *a.gout.at(n, i, j);
            const float gre = g.x * a.gscale, gim = g.y * a.gscale;
            s0 += gre; s1 = fmaf(gre, qr[e], s1); s2 += gim; s3 = fmaf(gim, qi[e], s3);
            vr = gre * sdr; vi = gim * sdi;
        }
        gr[e] = vr; gi[e] = vi;
    }
    if (a.dstats) {
        s0 = block_sum_t(s0, red); s1 = block_sum_t(s1, red); s2 = block_sum_t(s2, red); s3 = block_sum_t(s3, red);
        if (threadIdx.x == 0) { float* d = a.dstats + (long)n * 4; d[0] = s0; d[1] = s1; d[2] = s2; d[3] = s3; }
    }
}

// ---------------------------------------------------------------- NormUnet front half, adjoint
// forward (norm_unet.py:59-69, 76-86): p = (z - mean) / std over the window, std unbiased over the N = I J window values.
//   gz = gp / std + (dmean - sum(gp) / std) / N + phat (dstd - sum(gp phat) / std) / (N - 1)
// accumulate != 0 adds into gz (the y-f planes onto the x-f planes' result).
struct PackBwdArgs {
    const float* gp; const float* p; const float* stats; const float* dstats; View gz;
    int n, I, J, Ip, Jp, pad_i, pad_j, accumulate;
};
struct PackBwdArgs2 { PackBwdArgs s[2]; };

__global__ __launch_bounds__(256) void normunet_pack_bwd_kernel(PackBwdArgs a) {
    __shared__ float red[16];
    const int n = blockIdx.x;
    const long pp = (long)a.Ip * a.Jp;
    const float* gr = a.gp + (long)n * 2 * pp;
    const float* gi = gr + pp;
    const int cnt = a.I * a.J;
    if (!a.stats) {       // plain repack (cinenet.py:242): the adjoint is the inverse repack
        for (int e = threadIdx.x; e < cnt; e += blockDim.x) {
            const int i = e / a.J, j = e - i * a.J;
            const long qd = (long)(i + a.pad_i) * a.Jp + j + a.pad_j;
            float2* o = a.gz.at(n, i, j);
            float2 v = make_float2(gr[qd], gi[qd]);
            if (a.accumulate) { const float2 t = *o; v.x += t.x; v.y += t.y; }
            *o = v;
        }
        return;
    }
    const float* pr = a.p + (long)n * 2 * pp;
    const float* pi = pr + pp;
    const float sdr = a.stats[(long)n * 4 + 1], sdi = a.stats[(long)n * 4 + 3];
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    for (int e = threadIdx.x; e < cnt; e += blockDim.x) {
        const int i = e / a.J, j = e - i * a.J;
        const long qd = (long)(i + a.pad_i) * a.Jp + j + a.pad_j;
        s0 += gr[qd]; s1 = fmaf(gr[qd], pr[qd], s1); s2 += gi[qd]; s3 = fmaf(gi[qd], pi[qd], s3);
    }
    s0 = block_sum_t(s0, red); s1 = block_sum_t(s1, red); s2 = block_sum_t(s2, red); s3 = block_sum_t(s3, red);
    const float* d = a.dstats + (long)n * 4;
    const float cmr = (d[0] - s0 / sdr) / cnt, csr = (d[1] - s1 / sdr) / (cnt - 1);
    const float cmi = (d[2] - s2 / sdi) / cnt, csi = (d[3] - s3 / sdi) / (cnt - 1);
    for (int e = threadIdx.x; e < cnt; e += blockDim.x) {
        const int i = e / a.J, j = e - i * a.J;
        const long qd = (long)(i + a.pad_i) * a.Jp + j + a.pad_j;
        float2 v = make_float2(gr[qd] / sdr + cmr + pr[qd] * csr, gi[qd] / sdi + cmi + pi[qd] * csi);
        float2* o = a.gz.at(n, i, j);
        if (a.accumulate) { const float2 t = *o; v.x += t.x; v.y += t.y; }
        *o = v;
    }
}

// ---------------------------------------------------------------- temporal halves, adjoint
constexpr int kPixT = 64;
__device__ __forceinline__ void temporal_table_t(cf* tw, int T) {
    const double s = 1.0 / sqrt((double)T);
    for (int j = threadIdx.x; j < T; j += blockDim.x) {
        double sn, cs;
        sincospi(2.0 * (double)j / (double)T, &sn, &cs);
        tw[j] = mk((float)(cs * s), (float)(-sn * s));
    }
}
// centered ortho DFT of the T values buf[g * kPixT + p], output index i (DIR +1 forward, -1 inverse): fftc.py:5-56
template <int DIR>
__device__ __forceinline__ cf centered_dft(const cf* buf, const cf* tw, int T, int p, int i) {
    const int s_in = (T + 1) / 2, s_out = T / 2;
    int k = i - s_out; if (k < 0) k += T;
    float ax = 0.f, ay = 0.f;
    int idx = (s_in * k) % T;
    for (int g = 0; g < T; ++g) {
        const cf w = tw[idx];
        const cf x = buf[g * kPixT + p];
        if (DIR > 0) { ax += x.x * w.x - x.y * w.y; ay += x.x * w.y + x.y * w.x; }
        else { ax += x.x * w.x + x.y * w.y; ay += x.y * w.x - x.x * w.y; }
        idx += k; if (idx >= T) idx -= T;
    }
    return mk(ax, ay);
}

// adjoint of xfyf_unpack's tail (varnet.py:232-241): gout (b, t, h, w) -> GA[b][h][w][t] = fft1c_t(gout) (XF) or gout (XT);
// gmean[b][h][w] = sum_t gout (the temporal mean is added to every frame)
__global__ __launch_bounds__(256) void temporal_out_bwd_kernel(const cf* gout, cf* GA, cf* gmean, int T, long HW, int xf) {
    extern __shared__ __align__(16) unsigned char smem_t[];
    cf* buf = reinterpret_cast<cf*>(smem_t);       // [T][kPixT]
    cf* tw = buf + T * kPixT;
    const int b = blockIdx.y;
    const long p0 = (long)blockIdx.x * kPixT;
    const int np = (int)min((long)kPixT, HW - p0);
    if (xf) temporal_table_t(tw, T);
    for (int e = threadIdx.x; e < T * kPixT; e += blockDim.x) {
        const int t = e / kPixT, p = e - t * kPixT;
        buf[e] = p < np ? gout[((long)b * T + t) * HW + p0 + p] : mk(0.f, 0.f);
    }
    __syncthreads();
    if (threadIdx.x < np) {
        const int p = threadIdx.x;
        float sx = 0.f, sy = 0.f;
        for (int t = 0; t < T; ++t) { sx += buf[t * kPixT + p].x; sy += buf[t * kPixT + p].y; }
        gmean[(long)b * HW + p0 + p] = mk(sx, sy);
    }
    for (int e = threadIdx.x; e < np * T; e += blockDim.x) {
        const int p = e / T, i = e - p * T;
        GA[((long)b * HW + p0 + p) * T + i] = xf ? centered_dft<1>(buf, tw, T, p, i) : buf[i * kPixT + p];
    }
}

// adjoint of xfyf_pack's head (varnet.py:202-213): GX[b][h][w][t] -> gxc = ifft1c_t(GX) (XF) or GX (XT);
// gimg[b][t] = gxc - mean_t(gxc) + gmean / T
__global__ __launch_bounds__(256) void temporal_in_bwd_kernel(const cf* GX, const cf* gmean, cf* gimg, int T, long HW, int xf) {
    extern __shared__ __align__(16) unsigned char smem_t[];
    cf* buf = reinterpret_cast<cf*>(smem_t);       // [T][kPixT]
    cf* out = buf + T * kPixT;                      // [T][kPixT]
    cf* tw = out + T * kPixT;
    const int b = blockIdx.y;
    const long p0 = (long)blockIdx.x * kPixT;
    const int np = (int)min((long)kPixT, HW - p0);
    if (xf) temporal_table_t(tw, T);
    for (int e = threadIdx.x; e < np * T; e += blockDim.x) {       // lanes over t: contiguous reads
        const int p = e / T, t = e - p * T;
        buf[t * kPixT + p] = GX[((long)b * HW + p0 + p) * T + t];
    }
    __syncthreads();
    for (int e = threadIdx.x; e < T * np; e += blockDim.x) {
        const int i = e / np, p = e - i * np;
        out[i * kPixT + p] = xf ? centered_dft<-1>(buf, tw, T, p, i) : buf[i * kPixT + p];
    }
    __syncthreads();
    if (threadIdx.x < np) {
        const int p = threadIdx.x;
        float sx = 0.f, sy = 0.f;
        for (int t = 0; t < T; ++t) { sx += out[t * kPixT + p].x; sy += out[t * kPixT + p].y; }
        const cf gm = gmean[(long)b * HW + p0 + p];
        buf[p] = mk((gm.x - sx) / T, (gm.y - sy) / T);            // buf row 0 is free again: per-pixel correction
    }
    __syncthreads();
    for (int e = threadIdx.x; e < T * np; e += blockDim.x) {
        const int i = e / np, p = e - i * np;                     // lanes over pixels: coalesced stores
        gimg[((long)b * T + i) * HW + p0 + p] = cadd(out[i * kPixT + p], buf[p]);
    }
}

// ---------------------------------------------------------------- element-wise adjoints
// y = x / rss(x) over the coil axis (varnet.py:58-59): gx = gy / r - x (sum_c <gy_c, x_c>) / r^3, r = sqrt(sum_c |x_c|^2)
__global__ __launch_bounds__(256) void rss_normalise_bwd_kernel(const cf* gy, const cf* x, cf* gx, int C, long HW) {
    const int b = blockIdx.y;
    for (long p = (long)blockIdx.x * blockDim.x + threadIdx.x; p < HW; p += (long)gridDim.x * blockDim.x) {
        float s = 0.f, d = 0.f;
        for (int c = 0; c < C; ++c) {
            const cf v = x[((long)b * C + c) * HW + p], g = gy[((long)b * C + c) * HW + p];
            s += v.x * v.x + v.y * v.y;
            d += g.x * v.x + g.y * v.y;
        }
        const float r = sqrtf(s), ir = 1.f / r, k = d / (s * r);
        for (int c = 0; c < C; ++c) {
            const cf v = x[((long)b * C + c) * HW + p], g = gy[((long)b * C + c) * HW + p];
            gx[((long)b * C + c) * HW + p] = mk(g.x * ir - v.x * k, g.y * ir - v.y * k);
        }
    }
}

// y = |x| (math.py:48-62): gx = gy x / |x|
__global__ __launch_bounds__(256) void complex_abs_bwd_kernel(const float* gy, const cf* x, cf* gx, long n) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const cf v = x[i];
        const float s = gy[i] / sqrtf(v.x * v.x + v.y * v.y);
        gx[i] = mk(v.x * s, v.y * s);
    }
}

// gs[b][c][p] (+)= sum_t conj(g[b][t][p]) z[b][t][c][p]: the gradient of sum_c conj(S_c) z_c with respect to S
// (real-pair convention: d loss = Re(conj(gs) dS)); with z == NULL the summand is part[b][t][c][p] itself
__global__ __launch_bounds__(256) void coil_accum_kernel(const cf* g, const cf* z, cf* gs, int T, int C, long HW, int accumulate) {
    const int b = blockIdx.y;
    const long total = (long)C * HW;
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
        const int c = (int)(e / HW);
        const long p = e - (long)c * HW;
        float sx = 0.f, sy = 0.f;
        for (int t = 0; t < T; ++t) {
            cf v = z[(((long)b * T + t) * C + c) * HW + p];
            if (g) { const cf gg = g[((long)b * T + t) * HW + p]; v = cmulc(v, gg); }   // z * conj(g) = conj(g) z
            sx += v.x; sy += v.y;
        }
        cf* o = gs + (long)b * total + e;
        if (accumulate) { sx += o->x; sy += o->y; }
        *o = mk(sx, sy);
    }
}

// out = a + sign * f(softplus(*lam)) * b; a may be NULL.  kind 0: f = v; 1: v / (1 + v); 2: 1 / (1 + v)^2; 3: 1 / (1 + v)
__global__ __launch_bounds__(256) void axpby_lam_kernel(float* out, const float* a, const float* b, long n, const float* lam, int kind, float sign) {
    const float l = *lam;
    const float v = l > 20.f ? l : log1pf(expf(l));
    float s = kind == 0 ? v : kind == 1 ? v / (1.f + v) : kind == 2 ? 1.f / ((1.f + v) * (1.f + v)) : 1.f / (1.f + v);
    s *= sign;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
        out[i] = a ? fmaf(s, b[i], a[i]) : s * b[i];
}

static unsigned grid_t(long n, int threads, long cap = 4096) {
    long g = ceil_div(n, (long)threads);
    if (g > cap) g = cap;
    if (g < 1) g = 1;
    return (unsigned)g;
}

}  // namespace cine

using namespace cine;

static void pad_split_t(int n, int& np, int& lo, bool norm) {
    np = norm ? cine_pad16(n) : n;
    lo = (np - n) / 2;
}

extern "C" int cine_normunet_unpack_bwd(const float* gout, const float* planes_q, const float* stats, float* gq, float* dstats,
                                        int n, int h, int w, void* stream) {
    CINE_REQUIRE(gout && planes_q && gq && (!stats == !dstats), CINE_EINVAL, "cine_normunet_unpack_bwd: null pointer");
    CINE_REQUIRE(n > 0 && h > 0 && w > 0, CINE_EINVAL, "cine_normunet_unpack_bwd: bad sizes");
    UnpackBwdArgs a{};
    a.gout = View{const_cast<float*>(gout), 1, (long)h * w * 2, 0, (long)w * 2, 2};
    a.q = planes_q; a.stats = stats; a.gq = gq; a.dstats = dstats; a.n = n; a.I = h; a.J = w; a.gscale = 1.f;
    pad_split_t(h, a.Ip, a.pad_i, stats != nullptr); pad_split_t(w, a.Jp, a.pad_j, stats != nullptr);
    UnpackBwdArgs2 two{}; two.s[0] = a; two.s[1] = a;
    ProfScope prof(F_PACK, as_stream(stream));
    hipLaunchKernelGGL(normunet_unpack_bwd_kernel, dim3(n, 1), dim3(256), 0, as_stream(stream), two);
    return check_launch("normunet_unpack_bwd_kernel");
}

extern "C" int cine_normunet_pack_bwd(const float* gp, const float* planes_p, const float* stats, const float* dstats, float* gz,
                                      int n, int h, int w, void* stream) {
    CINE_REQUIRE(gp && gz && (!stats || (planes_p && dstats)), CINE_EINVAL, "cine_normunet_pack_bwd: null pointer");
    CINE_REQUIRE(n > 0 && h > 0 && w > 0 && (long)h * w > 1, CINE_EINVAL, "cine_normunet_pack_bwd: bad sizes");
    PackBwdArgs a{};
    a.gp = gp; a.p = planes_p; a.stats = stats; a.dstats = dstats;
    a.gz = View{gz, 1, (long)h * w * 2, 0, (long)w * 2, 2};
    a.n = n; a.I = h; a.J = w; a.accumulate = 0;
    pad_split_t(h, a.Ip, a.pad_i, stats != nullptr); pad_split_t(w, a.Jp, a.pad_j, stats != nullptr);
    ProfScope prof(F_PACK, as_stream(stream));
    hipLaunchKernelGGL(normunet_pack_bwd_kernel, dim3(n), dim3(256), 0, as_stream(stream), a);
    return check_launch("normunet_pack_bwd_kernel");
}

extern "C" size_t cine_xfyf_bwd_ws_bytes(int b, int t, int h, int w) { return (size_t)b * t * h * w * 2 * sizeof(float); }

extern "C" int cine_xfyf_unpack_bwd(const float* gout, const float* q_xf, const float* q_yf, const float* stats_xf, const float* stats_yf,
                                    float* gq_xf, float* gq_yf, float* dstats_xf, float* dstats_yf, float* gmean,
                                    int b, int t, int h, int w, int xf, void* ws, size_t ws_bytes, void* stream) {
    CINE_REQUIRE(gout && q_xf && q_yf && gq_xf && gq_yf && gmean && ws, CINE_EINVAL, "cine_xfyf_unpack_bwd: null pointer");
    CINE_REQUIRE((!stats_xf == !stats_yf) && (!stats_xf == !dstats_xf) && (!stats_yf == !dstats_yf), CINE_EINVAL,
                 "cine_xfyf_unpack_bwd: statistics pointers must be all set or all null");
    CINE_REQUIRE(b > 0 && b <= 65535 && t > 1 && t <= 64 && h > 0 && w > 0, CINE_EINVAL, "cine_xfyf_unpack_bwd: bad sizes");
    CINE_REQUIRE(ws_bytes >= cine_xfyf_bwd_ws_bytes(b, t, h, w), CINE_EWORKSPACE, "cine_xfyf_unpack_bwd: workspace too small");
    hipStream_t st = as_stream(stream);
    const long HW = (long)h * w;
    cf* GA = reinterpret_cast<cf*>(ws);
    ProfScope prof(F_PACK, st);
    hipLaunchKernelGGL(temporal_out_bwd_kernel, dim3((unsigned)ceil_div(HW, (long)kPixT), b), dim3(256), ((size_t)t * kPixT + t) * sizeof(cf), st,
                       reinterpret_cast<const cf*>(gout), GA, reinterpret_cast<cf*>(gmean), t, HW, xf);
    if (int e = check_launch("temporal_out_bwd_kernel")) return e;
    const bool nrm = stats_xf != nullptr;
    UnpackBwdArgs2 two{};
    UnpackBwdArgs& ax = two.s[0];
    ax.gout = View{reinterpret_cast<float*>(GA), h, HW * t * 2, (long)w * t * 2, (long)t * 2, 2};
    ax.q = q_xf; ax.stats = stats_xf; ax.gq = gq_xf; ax.dstats = dstats_xf; ax.n = b * h; ax.I = w; ax.J = t; ax.gscale = 0.5f;
    pad_split_t(w, ax.Ip, ax.pad_i, nrm); pad_split_t(t, ax.Jp, ax.pad_j, nrm);
    UnpackBwdArgs& ay = two.s[1];
    ay.gout = View{reinterpret_cast<float*>(GA), w, HW * t * 2, (long)t * 2, (long)w * t * 2, 2};
    ay.q = q_yf; ay.stats = stats_yf; ay.gq = gq_yf; ay.dstats = dstats_yf; ay.n = b * w; ay.I = h; ay.J = t; ay.gscale = 0.5f;
    pad_split_t(h, ay.Ip, ay.pad_i, nrm); pad_split_t(t, ay.Jp, ay.pad_j, nrm);
    hipLaunchKernelGGL(normunet_unpack_bwd_kernel, dim3(std::max(ax.n, ay.n), 2), dim3(256), 0, st, two);
    return check_launch("normunet_unpack_bwd_kernel");
}

extern "C" int cine_xfyf_pack_bwd(const float* gp_xf, const float* gp_yf, const float* p_xf, const float* p_yf,
                                  const float* stats_xf, const float* stats_yf, const float* dstats_xf, const float* dstats_yf,
                                  const float* gmean, float* gimg, int b, int t, int h, int w, int xf,
                                  void* ws, size_t ws_bytes, void* stream) {
    CINE_REQUIRE(gp_xf && gp_yf && gmean && gimg && ws, CINE_EINVAL, "cine_xfyf_pack_bwd: null pointer");
    CINE_REQUIRE((!stats_xf == !stats_yf) && (!stats_xf || (p_xf && p_yf && dstats_xf && dstats_yf)), CINE_EINVAL,
                 "cine_xfyf_pack_bwd: statistics pointers must be all set or all null");
    CINE_REQUIRE(b > 0 && b <= 65535 && t > 1 && t <= 64 && h > 0 && w > 0, CINE_EINVAL, "cine_xfyf_pack_bwd: bad sizes");
    CINE_REQUIRE(ws_bytes >= cine_xfyf_bwd_ws_bytes(b, t, h, w), CINE_EWORKSPACE, "cine_xfyf_pack_bwd: workspace too small");
    hipStream_t st = as_stream(stream);
    const long HW = (long)h * w;
    float* GX = reinterpret_cast<float*>(ws);
    const bool nrm = stats_xf != nullptr;
    ProfScope prof(F_PACK, st);
    PackBwdArgs a{};
    a.gp = gp_xf; a.p = p_xf; a.stats = stats_xf; a.dstats = dstats_xf;
    a.gz = View{GX, h, HW * t * 2, (long)w * t * 2, (long)t * 2, 2};
    a.n = b * h; a.I = w; a.J = t; a.accumulate = 0;
    pad_split_t(w, a.Ip, a.pad_i, nrm); pad_split_t(t, a.Jp, a.pad_j, nrm);
    hipLaunchKernelGGL(normunet_pack_bwd_kernel, dim3(a.n), dim3(256), 0, st, a);
    a.gp = gp_yf; a.p = p_yf; a.stats = stats_yf; a.dstats = dstats_yf;
    a.gz = View{GX, w, HW * t * 2, (long)t * 2, (long)w * t * 2, 2};
    a.n = b * w; a.I = h; a.J = t; a.accumulate = 1;
    pad_split_t(h, a.Ip, a.pad_i, nrm); pad_split_t(t, a.Jp, a.pad_j, nrm);
    hipLaunchKernelGGL(normunet_pack_bwd_kernel, dim3(a.n), dim3(256), 0, st, a);
    if (int e = check_launch("normunet_pack_bwd_kernel")) return e;
    hipLaunchKernelGGL(temporal_in_bwd_kernel, dim3((unsigned)ceil_div(HW, (long)kPixT), b), dim3(256), ((size_t)2 * t * kPixT + t) * sizeof(cf), st,
                       reinterpret_cast<const cf*>(GX), reinterpret_cast<const cf*>(gmean), reinterpret_cast<cf*>(gimg), t, HW, xf);
    return check_launch("temporal_in_bwd_kernel");
}

extern "C" int cine_rss_normalise_bwd(const float* gy, const float* x, float* gx, int b, int c, int h, int w, void* stream) {
    CINE_REQUIRE(gy && x && gx, CINE_EINVAL, "cine_rss_normalise_bwd: null pointer");
    CINE_REQUIRE(b > 0 && b <= 65535 && c > 0 && h > 0 && w > 0, CINE_EINVAL, "cine_rss_normalise_bwd: bad sizes");
    ProfScope prof(F_MISC, as_stream(stream));
    hipLaunchKernelGGL(rss_normalise_bwd_kernel, dim3(grid_t((long)h * w, 256), b), dim3(256), 0, as_stream(stream),
                       reinterpret_cast<const cf*>(gy), reinterpret_cast<const cf*>(x), reinterpret_cast<cf*>(gx), c, (long)h * w);
    return check_launch("rss_normalise_bwd_kernel");
}

extern "C" int cine_complex_abs_bwd(const float* gy, const float* x, float* gx, long n, void* stream) {
    CINE_REQUIRE(gy && x && gx && n >= 0, CINE_EINVAL, "cine_complex_abs_bwd: bad arguments");
    if (n == 0) return CINE_OK;
    ProfScope prof(F_MISC, as_stream(stream));
    hipLaunchKernelGGL(complex_abs_bwd_kernel, dim3(grid_t(n, 256)), dim3(256), 0, as_stream(stream), gy,
                       reinterpret_cast<const cf*>(x), reinterpret_cast<cf*>(gx), n);
    return check_launch("complex_abs_bwd_kernel");
}

extern "C" int cine_coil_accum(const float* g, const float* z, float* gs, int b, int t, int c, int h, int w, int accumulate, void* stream) {
    CINE_REQUIRE(z && gs, CINE_EINVAL, "cine_coil_accum: null pointer");
    CINE_REQUIRE(b > 0 && b <= 65535 && t > 0 && c > 0 && h > 0 && w > 0, CINE_EINVAL, "cine_coil_accum: bad sizes");
    ProfScope prof(F_MISC, as_stream(stream));
    hipLaunchKernelGGL(coil_accum_kernel, dim3(grid_t((long)c * h * w, 256), b), dim3(256), 0, as_stream(stream),
                       reinterpret_cast<const cf*>(g), reinterpret_cast<const cf*>(z), reinterpret_cast<cf*>(gs), t, c, (long)h * w, accumulate);
    return check_launch("coil_accum_kernel");
}

extern "C" int cine_axpby_lam(float* out, const float* a, const float* b, long n, const float* lambda_dev, int kind, float sign, void* stream) {
    CINE_REQUIRE(out && b && lambda_dev && n > 0 && kind >= 0 && kind <= 3, CINE_EINVAL, "cine_axpby_lam: bad arguments");
    ProfScope prof(F_MISC, as_stream(stream));
    hipLaunchKernelGGL(axpby_lam_kernel, dim3(grid_t(n, 256, 2048)), dim3(256), 0, as_stream(stream), out, a, b, n, lambda_dev, kind, sign);
    return check_launch("axpby_lam_kernel");
}

// ---------------------------------------------------------------- SSIMLoss forward + backward (utils/losses.py:25-58)
// loss = mean_t (1 - mean_windows S_t), S from win x win uniform windows (valid region), sample covariance, data range = max of the
// TARGET frame (losses.py:34).  The window sums run in float64 (the reference's float32 conv2d cancels catastrophically in
// uxx - ux^2; double keeps the loss and its gradient at float32 resolution).  Forward keeps dS/d(ux), dS/d(uxx), dS/d(uxy) per window;
// backward spreads them back over the windows that contain a pixel.
namespace cine {
constexpr int kSsimTile = 16;          // windows per tile side
constexpr int kSsimMaxWin = 11;

__global__ __launch_bounds__(256) void ssim_frame_max_kernel(const float* y, long hw, float* fmax) {
    __shared__ float red[4];
    const float* p = y + (long)blockIdx.x * hw;
    float m = -INFINITY;
    for (long e = threadIdx.x; e < hw; e += 256) m = fmaxf(m, p[e]);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) fmax[blockIdx.x] = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
}

// one workgroup = one 16 x 16 tile of windows of one frame; part[frame][tile] = sum of S over the tile's windows
__global__ __launch_bounds__(256) void ssim_loss_fwd_kernel(const float* x, const float* y, int H, int W, int win, double k1, double k2,
                                                            const float* fmax, float* da, float* db, float* dc, double* part) {
    __shared__ float sx[(kSsimTile + kSsimMaxWin - 1) * (kSsimTile + kSsimMaxWin - 1)];
    __shared__ float sy[(kSsimTile + kSsimMaxWin - 1) * (kSsimTile + kSsimMaxWin - 1)];
    __shared__ double red[4];
    const int Ho = H - win + 1, Wo = W - win + 1;
    const int tiles_w = (Wo + kSsimTile - 1) / kSsimTile;
    const int t = blockIdx.y, ty = blockIdx.x / tiles_w, tx = blockIdx.x % tiles_w;
    const int y0 = ty * kSsimTile, x0 = tx * kSsimTile;
    const int PS = kSsimTile + win - 1;
    const float* xp = x + (long)t * H * W;
    const float* yp = y + (long)t * H * W;
    for (int e = threadIdx.x; e < PS * PS; e += 256) {
        const int r = e / PS, c = e - r * PS;
        const int gy = min(y0 + r, H - 1), gx = min(x0 + c, W - 1);
        sx[e] = xp[(long)gy * W + gx]; sy[e] = yp[(long)gy * W + gx];
    }
    __syncthreads();
    const int wy = threadIdx.x / kSsimTile, wx = threadIdx.x % kSsimTile;
    double S = 0.0;
    if (y0 + wy < Ho && x0 + wx < Wo) {
        double a = 0, b = 0, aa = 0, bb = 0, ab = 0;
        for (int i = 0; i < win; ++i)
            for (int j = 0; j < win; ++j) {
                const double u = sx[(wy + i) * PS + wx + j], v = sy[(wy + i) * PS + wx + j];
                a += u; b += v; aa += u * u; bb += v * v; ab += u * v;
            }
        const double np = (double)win * win, cn = np / (np - 1.0);
        const double ux = a / np, uy = b / np, uxx = aa / np, uyy = bb / np, uxy = ab / np;
        const double L = fmax[t], C1 = (k1 * L) * (k1 * L), C2 = (k2 * L) * (k2 * L);
        const double vx = cn * (uxx - ux * ux), vy = cn * (uyy - uy * uy), vxy = cn * (uxy - ux * uy);
        const double A1 = 2 * ux * uy + C1, A2 = 2 * vxy + C2, B1 = ux * ux + uy * uy + C1, B2 = vx + vy + C2;
        S = (A1 * A2) / (B1 * B2);
        const long o = ((long)t * Ho + y0 + wy) * Wo + x0 + wx;
        da[o] = (float)(S * (2 * uy / A1 - 2 * cn * uy / A2 - 2 * ux / B1 + 2 * cn * ux / B2));
        db[o] = (float)(-S * cn / B2);
        dc[o] = (float)(S * 2 * cn / A2);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) S += __shfl_xor(S, o, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = S;
    __syncthreads();
    if (threadIdx.x == 0) part[(long)t * gridDim.x + blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}

__global__ __launch_bounds__(64) void ssim_loss_final_kernel(const double* part, int T, int ntiles, long nwin, float* loss) {
    double acc = 0.0;
    for (int t = 0; t < T; ++t) {               // fixed order: frames, then tiles lane-strided + butterfly
        double s = 0.0;
        for (int i = threadIdx.x; i < ntiles; i += 64) s += part[(long)t * ntiles + i];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
        acc += 1.0 - s / (double)nwin;
    }
    if (threadIdx.x == 0) *loss = (float)(acc / T);
}

// gx[t][p] = gloss * (-1 / (T Nw)) / win^2 * sum over the windows containing p of (da + 2 x_p db + y_p dc)
__global__ __launch_bounds__(256) void ssim_loss_bwd_kernel(const float* x, const float* y, int T, int H, int W, int win, const float* gloss,
                                                            const float* da, const float* db, const float* dc, float* gx) {
    const int Ho = H - win + 1, Wo = W - win + 1;
    const long total = (long)T * H * W;
    const double k = -(double)*gloss / ((double)T * Ho * Wo) / ((double)win * win);
    for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
        const int px = (int)(e % W); const long r = e / W;
        const int py = (int)(r % H), t = (int)(r / H);
        const int wy0 = max(py - win + 1, 0), wy1 = min(py, Ho - 1), wx0 = max(px - win + 1, 0), wx1 = min(px, Wo - 1);
        double sa = 0, sb = 0, sc = 0;
        for (int wy = wy0; wy <= wy1; ++wy)
            for (int wx = wx0; wx <= wx1; ++wx) {
                const long o = ((long)t * Ho + wy) * Wo + wx;
                sa += da[o]; sb += db[o]; sc += dc[o];
            }
        gx[e] = (float)(k * (sa + 2.0 * x[e] * sb + y[e] * sc));
    }
}
}  // namespace cine

extern "C" size_t cine_ssim_loss_ws_bytes(int t, int h, int w, int win) {
    if (t <= 0 || h < win || w < win || win < 1 || win > kSsimMaxWin) return 0;
    const long nw = (long)t * (h - win + 1) * (w - win + 1);
    const long ntiles = (long)ceil_div(h - win + 1, kSsimTile) * ceil_div(w - win + 1, kSsimTile);
    return (size_t)(3 * nw + t + 16) * sizeof(float) + (size_t)(t * ntiles + 2) * sizeof(double);
}

static void ssim_ws_layout(void* ws, int t, int h, int w, int win, float*& fmax, float*& da, float*& db, float*& dc, double*& part) {
    const long nw = (long)t * (h - win + 1) * (w - win + 1);
    part = reinterpret_cast<double*>(ws);
    const long ntiles = (long)ceil_div(h - win + 1, kSsimTile) * ceil_div(w - win + 1, kSsimTile);
    float* f = reinterpret_cast<float*>(part + (long)t * ntiles + 2);
    fmax = f; da = f + t; db = da + nw; dc = db + nw;
}

extern "C" int cine_ssim_loss(const float* x, const float* y, int t, int h, int w, int win, double k1, double k2,
                              float* loss_dev, void* ws, size_t ws_bytes, void* stream) {
    CINE_REQUIRE(x && y && loss_dev && ws, CINE_EINVAL, "cine_ssim_loss: null pointer");
    CINE_REQUIRE(t > 0 && t <= 65535 && win >= 1 && win <= kSsimMaxWin && h >= win && w >= win, CINE_EINVAL, "cine_ssim_loss: bad sizes");
    CINE_REQUIRE(ws_bytes >= cine_ssim_loss_ws_bytes(t, h, w, win), CINE_EWORKSPACE, "cine_ssim_loss: workspace too small");
    float *fmax, *da, *db, *dc; double* part;
    ssim_ws_layout(ws, t, h, w, win, fmax, da, db, dc, part);
    hipStream_t st = as_stream(stream);
    const int ntiles = ceil_div(h - win + 1, kSsimTile) * ceil_div(w - win + 1, kSsimTile);
    ProfScope prof(F_MISC, st);
    hipLaunchKernelGGL(ssim_frame_max_kernel, dim3(t), dim3(256), 0, st, y, (long)h * w, fmax);
    hipLaunchKernelGGL(ssim_loss_fwd_kernel, dim3(ntiles, t), dim3(256), 0, st, x, y, h, w, win, k1, k2, fmax, da, db, dc, part);
    hipLaunchKernelGGL(ssim_loss_final_kernel, dim3(1), dim3(64), 0, st, part, t, ntiles, (long)(h - win + 1) * (w - win + 1), loss_dev);
    return check_launch("ssim_loss kernels");
}

extern "C" int cine_ssim_loss_bwd(const float* x, const float* y, int t, int h, int w, int win, const float* gloss_dev,
                                  const void* ws, size_t ws_bytes, float* gx, void* stream) {
    CINE_REQUIRE(x && y && gloss_dev && ws && gx, CINE_EINVAL, "cine_ssim_loss_bwd: null pointer");
    CINE_REQUIRE(t > 0 && win >= 1 && win <= kSsimMaxWin && h >= win && w >= win, CINE_EINVAL, "cine_ssim_loss_bwd: bad sizes");
    CINE_REQUIRE(ws_bytes >= cine_ssim_loss_ws_bytes(t, h, w, win), CINE_EWORKSPACE, "cine_ssim_loss_bwd: workspace too small");
    float *fmax, *da, *db, *dc; double* part;
    ssim_ws_layout(const_cast<void*>(ws), t, h, w, win, fmax, da, db, dc, part);
    ProfScope prof(F_MISC, as_stream(stream));
    hipLaunchKernelGGL(ssim_loss_bwd_kernel, dim3(grid_t((long)t * h * w, 256)), dim3(256), 0, as_stream(stream), x, y, t, h, w, win, gloss_dev,
                       da, db, dc, gx);
    return check_launch("ssim_loss_bwd_kernel");
}

// ---------------------------------------------------------------- XPDNet I-step halves, adjoint (models/xpdnet.py:424-509)
namespace cine {
constexpr int kXPixT = 32;
// centered ortho DFT over the T values tile[g * cols + r] with explicit shifts: sum_g x[g] W^(+-(i - s_out)(g + s_in))
template <int DIR>
__device__ __forceinline__ cf shifted_dft(const cf* tile, const cf* tw, int T, int cols, int r, int i, int s_in, int s_out) {
    int k = i - s_out; if (k < 0) k += T;
    float ax = 0.f, ay = 0.f;
    int idx = (s_in * k) % T;
    for (int g = 0; g < T; ++g) {
        const cf w = tw[idx], x = tile[g * cols + r];
        if (DIR > 0) { ax += x.x * w.x - x.y * w.y; ay += x.x * w.y + x.y * w.x; }
        else { ax += x.x * w.x + x.y * w.y; ay += x.y * w.x - x.x * w.y; }
        idx += k; if (idx >= T) idx -= T;
    }
    return mk(ax, ay);
}

struct XpdUnpackBwdArgs {
    const float* gout; float* gpxf; float* gpyf; cf* gmean;
    int n, T, H, W, Wpx, Tp, pad_wx, pad_t, Hpy, pad_hy, xf;
};
// adjoint of xpd_unpack_kernel: gout (b, t, h, w, 2n) -> 0.5 * T2^H(gout) scattered into both plane sets (their pad frames are zeroed
// beforehand); gmean[b][pix][k < n] = sum_t gout (the temporal mean of channels 0..n-1 is added to every frame, :504-509), channel n: 0.
// T2 = fftshift(ifft(ifftshift(.))) (:500) is unitary: its adjoint is the forward transform with the same shifts.
__global__ __launch_bounds__(256) void xpd_unpack_bwd_kernel(XpdUnpackBwdArgs a) {
    extern __shared__ __align__(16) unsigned char smem_x[];
    const int n = a.n, T = a.T, H = a.H, W = a.W;
    cf* tile = reinterpret_cast<cf*>(smem_x);          // [T][kXPixT * n]
    cf* tw = tile + (size_t)T * kXPixT * n;
    const int b = blockIdx.z, h = blockIdx.y, w0 = blockIdx.x * kXPixT;
    const int np = min(kXPixT, W - w0);
    const int cols = kXPixT * n;
    const long HW = (long)H * W;
    if (a.xf) temporal_table_t(tw, T);
    for (int e = threadIdx.x; e < T * np * n; e += blockDim.x) {
        const int t = e / (np * n), r = e - t * (np * n), p = r / n, k = r - p * n;
        const float* g = a.gout + (((long)b * T + t) * HW + (long)h * W + w0 + p) * 2 * n;
        tile[t * cols + r] = mk(g[k], g[n + k]);
    }
    __syncthreads();
    for (int r = threadIdx.x; r < np * (n + 1); r += blockDim.x) {
        const int p = r / (n + 1), k = r - p * (n + 1);
        float sx = 0.f, sy = 0.f;
        if (k < n) for (int t = 0; t < T; ++t) { sx += tile[t * cols + p * n + k].x; sy += tile[t * cols + p * n + k].y; }
        a.gmean[((long)b * HW + (long)h * W + w0 + p) * (n + 1) + k] = mk(sx, sy);
    }
    float* px = a.gpxf + ((long)b * H + h) * 2 * n * a.Wpx * a.Tp;
    const long chx = (long)a.Wpx * a.Tp, chy = (long)a.Hpy * a.Tp;
    for (int e = threadIdx.x; e < np * n * T; e += blockDim.x) {
        const int r = e / T, i = e - r * T, p = r / n, k = r - p * n;
        const cf v = a.xf ? shifted_dft<1>(tile, tw, T, cols, r, i, (T + 1) / 2, T / 2) : tile[i * cols + r];
        const int w = w0 + p;
        const long qx = (long)(w + a.pad_wx) * a.Tp + i + a.pad_t;
        float* py = a.gpyf + ((long)b * W + w) * 2 * n * a.Hpy * a.Tp;
        const long qy = (long)(h + a.pad_hy) * a.Tp + i + a.pad_t;
        px[k * chx + qx] = 0.5f * v.x; px[(n + k) * chx + qx] = 0.5f * v.y;
        py[k * chy + qy] = 0.5f * v.x; py[(n + k) * chy + qy] = 0.5f * v.y;
    }
}

struct XpdPackBwdArgs {
    const float* gpxf; const float* gpyf; const cf* gmean; float* gbuf; cf* gextra;
    int n, T, H, W, Wpx, Tp, pad_wx, pad_t, Hpy, pad_hy, xf;
};
// adjoint of cine_xpd_pack: the gradients of the two plane sets (2 (n + 1) channels each) gathered per pixel, T1^H over the frames
// (T1 = ifftshift(fft(fftshift(.))), :466: the inverse transform with the same shifts), the temporal-mean subtraction's adjoint, then split
// into the buffer's gradient (channels < n) and the backward-operator image's (channel n).
__global__ __launch_bounds__(256) void xpd_pack_bwd_kernel(XpdPackBwdArgs a) {
    extern __shared__ __align__(16) unsigned char smem_x[];
    const int n = a.n, nc = n + 1, T = a.T, H = a.H, W = a.W;
    const int cols = kXPixT * nc;
    cf* tile = reinterpret_cast<cf*>(smem_x);          // [T][cols]
    cf* out = tile + (size_t)T * cols;                 // [T][cols]
    cf* tw = out + (size_t)T * cols;
    const int b = blockIdx.y;
    const long HW = (long)H * W, p0 = (long)blockIdx.x * kXPixT;
    const int np = (int)min((long)kXPixT, HW - p0);
    if (a.xf) temporal_table_t(tw, T);
    const long chx = (long)a.Wpx * a.Tp, chy = (long)a.Hpy * a.Tp;
    for (int e = threadIdx.x; e < np * nc * T; e += blockDim.x) {
        const int r = e / T, i = e - r * T, p = r / nc, k = r - p * nc;
        const long pix = p0 + p;
        const int h = (int)(pix / W), w = (int)(pix - (long)h * W);
        const float* px = a.gpxf + ((long)b * H + h) * 2 * nc * chx + (long)(w + a.pad_wx) * a.Tp + i + a.pad_t;
        const float* py = a.gpyf + ((long)b * W + w) * 2 * nc * chy + (long)(h + a.pad_hy) * a.Tp + i + a.pad_t;
        tile[i * cols + r] = mk(px[k * chx] + py[k * chy], px[(nc + k) * chx] + py[(nc + k) * chy]);
    }
    __syncthreads();
    for (int e = threadIdx.x; e < T * np * nc; e += blockDim.x) {
        const int i = e / (np * nc), r = e - i * (np * nc);
        out[i * cols + r] = a.xf ? shifted_dft<-1>(tile, tw, T, cols, r, i, T / 2, (T + 1) / 2) : tile[i * cols + r];
    }
    __syncthreads();
    for (int r = threadIdx.x; r < np * nc; r += blockDim.x) {
        const int p = r / nc, k = r - p * nc;
        float sx = 0.f, sy = 0.f;
        for (int t = 0; t < T; ++t) { sx += out[t * cols + r].x; sy += out[t * cols + r].y; }
        const cf gm = a.gmean[((long)b * HW + p0 + p) * nc + k];
        tile[r] = mk((gm.x - sx) / T, (gm.y - sy) / T);               // row 0 of `tile` is free again: per (pixel, channel) correction
    }
    __syncthreads();
    for (int e = threadIdx.x; e < T * np * nc; e += blockDim.x) {
        const int t = e / (np * nc), r = e - t * (np * nc), p = r / nc, k = r - p * nc;
        const cf v = cadd(out[t * cols + r], tile[r]);
        const long bt_pix = ((long)b * T + t) * HW + p0 + p;
        if (k < n) { float* q = a.gbuf + bt_pix * 2 * n; q[k] = v.x; q[n + k] = v.y; }
        else a.gextra[bt_pix] = v;
    }
}
}  // namespace cine

static void xpd_pads(int t, int h, int w, int n_scales, int& tp, int& lt, int& wp, int& lw, int& hp, int& lh) {
    int r;
    tp = cine_mwcnn_pad(t, n_scales, &lt, &r); wp = cine_mwcnn_pad(w, n_scales, &lw, &r); hp = cine_mwcnn_pad(h, n_scales, &lh, &r);
}

extern "C" int cine_xpd_unpack_bwd(const float* gout, float* gplanes_xf, float* gplanes_yf, float* gmean,
                                   int b, int t, int h, int w, int n_primal, int n_scales, int xf, void* stream) {
    CINE_REQUIRE(gout && gplanes_xf && gplanes_yf && gmean, CINE_EINVAL, "cine_xpd_unpack_bwd: null pointer");
    CINE_REQUIRE(b > 0 && t > 1 && t <= 64 && h > 0 && w > 0 && n_primal >= 1 && n_primal <= 15 && h <= 65535 && b <= 65535, CINE_EINVAL,
                 "cine_xpd_unpack_bwd: bad sizes");
    XpdUnpackBwdArgs a{};
    a.gout = gout; a.gpxf = gplanes_xf; a.gpyf = gplanes_yf; a.gmean = reinterpret_cast<cf*>(gmean);
    a.n = n_primal; a.T = t; a.H = h; a.W = w; a.xf = xf;
    xpd_pads(t, h, w, n_scales, a.Tp, a.pad_t, a.Wpx, a.pad_wx, a.Hpy, a.pad_hy);
    hipStream_t st = as_stream(stream);
    const size_t lds = ((size_t)t * kXPixT * n_primal + t) * sizeof(cf);
    CINE_REQUIRE(lds <= 64 * 1024, CINE_EUNSUPPORTED, "cine_xpd_unpack_bwd: tile does not fit LDS");
    ProfScope prof(F_PACK, st);
    CINE_REQUIRE(hipMemsetAsync(gplanes_xf, 0, (size_t)b * h * 2 * n_primal * a.Wpx * a.Tp * sizeof(float), st) == hipSuccess &&
                 hipMemsetAsync(gplanes_yf, 0, (size_t)b * w * 2 * n_primal * a.Hpy * a.Tp * sizeof(float), st) == hipSuccess, CINE_EHIP,
                 "cine_xpd_unpack_bwd: hipMemsetAsync failed");
    hipLaunchKernelGGL(xpd_unpack_bwd_kernel, dim3(ceil_div(w, kXPixT), h, b), dim3(256), lds, st, a);
    return check_launch("xpd_unpack_bwd_kernel");
}

extern "C" int cine_xpd_pack_bwd(const float* gplanes_xf, const float* gplanes_yf, const float* gmean, float* gbuf, float* gextra,
                                 int b, int t, int h, int w, int n_primal, int n_scales, int xf, void* stream) {
    CINE_REQUIRE(gplanes_xf && gplanes_yf && gmean && gbuf && gextra, CINE_EINVAL, "cine_xpd_pack_bwd: null pointer");
    CINE_REQUIRE(b > 0 && t > 1 && t <= 64 && h > 0 && w > 0 && n_primal >= 1 && n_primal <= 15 && b <= 65535, CINE_EINVAL,
                 "cine_xpd_pack_bwd: bad sizes");
    XpdPackBwdArgs a{};
    a.gpxf = gplanes_xf; a.gpyf = gplanes_yf; a.gmean = reinterpret_cast<const cf*>(gmean); a.gbuf = gbuf; a.gextra = reinterpret_cast<cf*>(gextra);
    a.n = n_primal; a.T = t; a.H = h; a.W = w; a.xf = xf;
    xpd_pads(t, h, w, n_scales, a.Tp, a.pad_t, a.Wpx, a.pad_wx, a.Hpy, a.pad_hy);
    const size_t lds = ((size_t)2 * t * kXPixT * (n_primal + 1) + t) * sizeof(cf);
    CINE_REQUIRE(lds <= 64 * 1024, CINE_EUNSUPPORTED, "cine_xpd_pack_bwd: tile does not fit LDS");
    ProfScope prof(F_PACK, as_stream(stream));
    hipLaunchKernelGGL(xpd_pack_bwd_kernel, dim3((unsigned)ceil_div((long)h * w, (long)kXPixT), b), dim3(256), lds, as_stream(stream), a);
    return check_launch("xpd_pack_bwd_kernel");
}
