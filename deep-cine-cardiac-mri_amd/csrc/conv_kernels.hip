// conv_kernels.hip -- U-Net building blocks for gfx950.
//
// conv3x3 (unet.py:160,164) is an implicit GEMM on the fp32 matrix cores
// (v_mfma_f32_16x16x4_f32: exact fp32, same numerics class as the reference's fp32 conv):
//     D[cout][pixel] += W[cout][(tap, cin)] * X[(tap, cin)][pixel]
//   M = 16 output channels, N = 16 pixels (one "fragment": 16/TW rows x TW columns),
//   K = 4 input channels of one tap.
// One workgroup = WM x WN waves; it owns 16*CT*WM output channels x WN*MT fragments of one
// sample and walks the input channels in chunks of CK: the chunk's input tile (with halo)
// and weight slab are staged in LDS, every wave then issues 9 * CK/4 * CT * MT MFMAs with
// operands fetched by ds_read_b32 at compile-time offsets.
// InstanceNorm + LeakyReLU of the PREVIOUS layer, the 2x2 average pool and the skip concat
// are applied while staging (cine_hip.h, cine_conv3x3_in), so normalised activations never
// round-trip through HBM.
#include <mutex>
#include "common.h"

namespace cine {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int kCK = 8;    // input channels per LDS chunk (2 MFMA k-steps per tap)

struct ConvSrc {
    const float* x; const float* stats;
    int c, mode, h, w;
};
struct ConvArgs {
    ConvSrc s0, s1;
    const float* wp; float* y;
    int n, cin, cout, coutp, H, W;
    float slope;
    int tiles_w, nchunks;
};

__device__ __forceinline__ float act(float x, float mean, float rstd, float slope) {
    const float v = (x - mean) * rstd;
    return v > 0.f ? v : v * slope;
}

// value of concatenated-input channel ci at (gy, gx) of sample n, after the source's transform
__device__ __forceinline__ float fetch_src(const ConvSrc& s, int n, int cl, int gy, int gx, float slope) {
    const long plane = (long)n * s.c + cl;
    if (s.mode == 2) {
        if (2 * gy + 1 >= s.h || 2 * gx + 1 >= s.w) return 0.f;      // outside the pooled extent
        const float mean = s.stats[plane * 2], rstd = s.stats[plane * 2 + 1];
        const float* p = s.x + (plane * s.h + 2 * gy) * s.w + 2 * gx;
        return 0.25f * (act(p[0], mean, rstd, slope) + act(p[1], mean, rstd, slope) +
                        act(p[s.w], mean, rstd, slope) + act(p[s.w + 1], mean, rstd, slope));
    }
    if (gy >= s.h || gx >= s.w) return 0.f;                           // up-path zero pad
    const float v = s.x[(plane * s.h + gy) * s.w + gx];
    if (s.mode == 0) return v;
    return act(v, s.stats[plane * 2], s.stats[plane * 2 + 1], slope);
}

template <int CK, int CT, int WM, int WN, int MT, int TW>
struct ConvCfg {
    static constexpr int NT = 64 * WM * WN;
    static constexpr int RPF = 16 / TW;            // rows per fragment
    static constexpr int NF = WN * MT;             // fragments per workgroup
    static constexpr int TH = NF * RPF;            // tile rows
    static constexpr int ROWS = TH + 2, COLS = TW + 2;
    static constexpr int PS = ((ROWS * COLS + 31) / 32) * 32 + 16;   // plane stride == 16 mod 32
    static constexpr int COT = 16 * CT * WM;
    static constexpr int COTP = (COT % 32 == 0) ? COT + 16 : COT;
    static constexpr size_t LDS = (size_t)(CK * PS + 9 * CK * COTP) * sizeof(float);
};

template <int CK, int CT, int WM, int WN, int MT, int TW>
__global__ __launch_bounds__(64 * WM * WN) void conv3x3_mfma_kernel(ConvArgs a) {
    using C = ConvCfg<CK, CT, WM, WN, MT, TW>;
    extern __shared__ __align__(16) float smem_f[];
    float* in_lds = smem_f;
    float* w_lds = smem_f + CK * C::PS;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int ty = blockIdx.x / a.tiles_w, tx = blockIdx.x % a.tiles_w;
    const int r0 = ty * C::TH, c0 = tx * TW;
    const int co0 = blockIdx.y * C::COT;
    const int n = blockIdx.z;
    const int q = lane & 15, kk = lane >> 4;
    const int qr = q / TW, qc = q % TW;
    const int base_in = kk * C::PS + (wn * MT * C::RPF + qr) * C::COLS + qc;
    const int base_w = kk * C::COTP + 16 * (wm * CT) + q;

    f32x4 acc[CT][MT];
#pragma unroll
    for (int ct = 0; ct < CT; ++ct)
#pragma unroll
        for (int f = 0; f < MT; ++f) acc[ct][f] = (f32x4){0.f, 0.f, 0.f, 0.f};

    for (int chunk = 0; chunk < a.nchunks; ++chunk) {
        __syncthreads();
        // ---- stage the weight slab [tap][ck][COT] (packed layout [chunk][tap][ck][coutp])
        {
            const float* wsrc = a.wp + (long)chunk * 9 * CK * a.coutp;
            for (int e = tid; e < 9 * CK * (C::COT / 4); e += C::NT) {
                const int row = e / (C::COT / 4), c4 = (e % (C::COT / 4)) * 4;
                float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                if (co0 + c4 < a.coutp) v = *reinterpret_cast<const float4*>(wsrc + (long)row * a.coutp + co0 + c4);
                *reinterpret_cast<float4*>(w_lds + row * C::COTP + c4) = v;
            }
        }
        // ---- stage the input tile with halo, transformed
        for (int e = tid; e < CK * C::ROWS * C::COLS; e += C::NT) {
            const int ck = e / (C::ROWS * C::COLS);
            const int rem = e - ck * (C::ROWS * C::COLS);
            const int row = rem / C::COLS, col = rem - row * C::COLS;
            const int ci = chunk * CK + ck;
            const int gy = r0 - 1 + row, gx = c0 - 1 + col;
            float v = 0.f;
            if (ci < a.cin && gy >= 0 && gy < a.H && gx >= 0 && gx < a.W)
                v = ci < a.s0.c ? fetch_src(a.s0, n, ci, gy, gx, a.slope)
                                : fetch_src(a.s1, n, ci - a.s0.c, gy, gx, a.slope);
            in_lds[ck * C::PS + row * C::COLS + col] = v;
        }
        __syncthreads();
        // ---- MFMA sweep
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const int dy = tap / 3, dx = tap % 3;
#pragma unroll
            for (int s = 0; s < CK / 4; ++s) {
                float af[CT], bf[MT];
#pragma unroll
                for (int ct = 0; ct < CT; ++ct) af[ct] = w_lds[base_w + (tap * CK + 4 * s) * C::COTP + 16 * ct];
#pragma unroll
                for (int f = 0; f < MT; ++f)
                    bf[f] = in_lds[base_in + (4 * s) * C::PS + (f * C::RPF + dy) * C::COLS + dx];
#pragma unroll
                for (int ct = 0; ct < CT; ++ct)
#pragma unroll
                    for (int f = 0; f < MT; ++f)
                        acc[ct][f] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[ct], bf[f], acc[ct][f], 0, 0, 0);
            }
        }
    }
    // ---- store raw conv output (NCHW)
#pragma unroll
    for (int ct = 0; ct < CT; ++ct)
#pragma unroll
        for (int f = 0; f < MT; ++f) {
            const int fg = wn * MT + f;
            const int gy = r0 + fg * C::RPF + qr, gx = c0 + qc;
            if (gy >= a.H || gx >= a.W) continue;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int co = co0 + 16 * (wm * CT + ct) + 4 * kk + j;
                if (co < a.cout) a.y[(((long)n * a.cout + co) * a.H + gy) * a.W + gx] = acc[ct][f][j];
            }
        }
}

// ---------------------------------------------------------------- weight packing
// (cout, cin, 3, 3) -> [chunk][tap][ck][coutp], zero padded
__global__ void pack_conv3x3_kernel(const float* w, float* p, int cout, int cin, int coutp, int nchunks) {
    const long total = (long)nchunks * 9 * kCK * coutp;
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
        const int co = (int)(e % coutp);
        long r = e / coutp;
        const int ck = (int)(r % kCK); r /= kCK;
        const int tap = (int)(r % 9);
        const int chunk = (int)(r / 9);
        const int ci = chunk * kCK + ck;
        p[e] = (co < cout && ci < cin) ? w[((long)co * cin + ci) * 9 + tap] : 0.f;
    }
}

// ---------------------------------------------------------------- InstanceNorm statistics
// stats[plane] = {mean, 1/sqrt(biased var + eps)}; exact two-pass; one wave per plane
// (plane_elems <= 8192) or one workgroup per plane.
__device__ __forceinline__ float wave_sum_f(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

__global__ void instnorm_stats_wave_kernel(const float* x, float* stats, long planes, int pe, float eps) {
    const long plane = (long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (plane >= planes) return;
    const int lane = threadIdx.x & 63;
    const float* p = x + plane * pe;
    float s = 0.f;
    for (int i = lane; i < pe; i += 64) s += p[i];
    const float mean = wave_sum_f(s) / pe;
    float qv = 0.f;
    for (int i = lane; i < pe; i += 64) { const float d = p[i] - mean; qv += d * d; }
    const float var = wave_sum_f(qv) / pe;
    if (lane == 0) { stats[plane * 2] = mean; stats[plane * 2 + 1] = 1.0f / sqrtf(var + eps); }
}

__global__ void instnorm_stats_block_kernel(const float* x, float* stats, long pe, float eps) {
    __shared__ float red[16];
    const long plane = blockIdx.x;
    const float* p = x + plane * pe;
    const int wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    float s = 0.f;
    for (long i = threadIdx.x; i < pe; i += blockDim.x) s += p[i];
    s = wave_sum_f(s);
    if ((threadIdx.x & 63) == 0) red[wave] = s;
    __syncthreads();
    float tot = 0.f;
    for (int i = 0; i < nw; ++i) tot += red[i];
    const float mean = tot / pe;
    __syncthreads();
    float qv = 0.f;
    for (long i = threadIdx.x; i < pe; i += blockDim.x) { const float d = p[i] - mean; qv += d * d; }
    qv = wave_sum_f(qv);
    if ((threadIdx.x & 63) == 0) red[wave] = qv;
    __syncthreads();
    if (threadIdx.x == 0) {
        float t2 = 0.f;
        for (int i = 0; i < nw; ++i) t2 += red[i];
        stats[plane * 2] = mean;
        stats[plane * 2 + 1] = 1.0f / sqrtf(t2 / pe + eps);
    }
}

__global__ void instnorm_lrelu_apply_kernel(const float* x, const float* stats, float* y, long planes, long pe, float slope) {
    const long total = planes * pe;
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
        const long plane = e / pe;
        y[e] = act(x[e], stats[plane * 2], stats[plane * 2 + 1], slope);
    }
}

// ---------------------------------------------------------------- transpose conv k2 s2
// One workgroup: 64 input pixels of one sample x all cin in LDS; wave g handles output
// channels g, g+4, ...; weights (cin, cout, 2, 2) read through the scalar cache.
constexpr int kTPix = 64;
__global__ __launch_bounds__(256) void tconv2x2_kernel(const float* x, const float* stats, int mode, const float* wt,
                                                        float* y, int cin, int cout, int H, int W, float slope) {
    extern __shared__ __align__(16) float xs[];        // [cin][kTPix]
    const int n = blockIdx.y;
    const int p0 = blockIdx.x * kTPix;
    const int HW = H * W;
    for (int e = threadIdx.x; e < cin * kTPix; e += blockDim.x) {
        const int ci = e / kTPix, p = e % kTPix;
        float v = 0.f;
        if (p0 + p < HW) {
            const long plane = (long)n * cin + ci;
            v = x[plane * HW + p0 + p];
            if (mode == 1) v = act(v, stats[plane * 2], stats[plane * 2 + 1], slope);
        }
        xs[e] = v;
    }
    __syncthreads();
    const int p = threadIdx.x & 63;
    const int g = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int pix = p0 + p;
    const int h = pix / W, w = pix - h * W;
    const int OW = 2 * W;
    for (int co = g; co < cout; co += 4) {
        float a00 = 0.f, a01 = 0.f, a10 = 0.f, a11 = 0.f;
        const float* wc = wt + (long)co * 4;
        for (int ci = 0; ci < cin; ++ci) {
            const float xv = xs[ci * kTPix + p];
            const float4 wv = *reinterpret_cast<const float4*>(wc + (long)ci * cout * 4);
            a00 += xv * wv.x; a01 += xv * wv.y; a10 += xv * wv.z; a11 += xv * wv.w;
        }
        if (pix < HW) {
            float* o = y + (((long)n * cout + co) * 2 * H + 2 * h) * OW + 2 * w;
            *reinterpret_cast<float2*>(o) = make_float2(a00, a01);
            *reinterpret_cast<float2*>(o + OW) = make_float2(a10, a11);
        }
    }
}

// ---------------------------------------------------------------- 1x1 conv + bias
__global__ void conv1x1_bias_kernel(const float* x, const float* stats, int mode, const float* wt, const float* bias,
                                    float* y, int cin, int cout, long HW, float slope) {
    const int n = blockIdx.y;
    for (long p = (long)blockIdx.x * blockDim.x + threadIdx.x; p < HW; p += (long)gridDim.x * blockDim.x) {
        for (int co = 0; co < cout; ++co) {
            float acc = bias[co];
            for (int ci = 0; ci < cin; ++ci) {
                const long plane = (long)n * cin + ci;
                float v = x[plane * HW + p];
                if (mode == 1) v = act(v, stats[plane * 2], stats[plane * 2 + 1], slope);
                acc += v * wt[co * cin + ci];
            }
            y[((long)n * cout + co) * HW + p] = acc;
        }
    }
}

// ---------------------------------------------------------------- host dispatch
template <int CK, int CT, int WM, int WN, int MT, int TW>
static int launch_conv(ConvArgs a, hipStream_t st) {
    using C = ConvCfg<CK, CT, WM, WN, MT, TW>;
    static std::once_flag once;
    auto kern = conv3x3_mfma_kernel<CK, CT, WM, WN, MT, TW>;
    std::call_once(once, [&] {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)C::LDS);
    });
    a.tiles_w = ceil_div(a.W, TW);
    const int tiles_h = ceil_div(a.H, C::TH);
    dim3 grid(a.tiles_w * tiles_h, ceil_div(a.coutp, C::COT), a.n);
    ProfScope prof(F_CONV3, st);
    hipLaunchKernelGGL(kern, grid, dim3(C::NT), C::LDS, st, a);
    return check_launch("conv3x3_mfma_kernel");
}

template <int TW>
static int dispatch_conv_tw(const ConvArgs& a, hipStream_t st) {
    const long frags = (long)ceil_div(a.H * TW, 16) * ceil_div(a.W, TW);   // fragments per sample
    if (a.coutp <= 16) return launch_conv<kCK, 1, 1, 4, 13, TW>(a, st);
    if (a.coutp <= 32) return launch_conv<kCK, 2, 1, 4, 13, TW>(a, st);
    if (a.coutp <= 64 || frags > 8) return launch_conv<kCK, 1, 4, 1, 13, TW>(a, st);
    return launch_conv<kCK, 2, 4, 1, 4, TW>(a, st);
}

int conv3x3_dispatch(const ConvArgs& a, hipStream_t st) {
    if (a.W > 8) return dispatch_conv_tw<16>(a, st);
    if (a.W > 4) return dispatch_conv_tw<8>(a, st);
    if (a.W > 2) return dispatch_conv_tw<4>(a, st);
    return dispatch_conv_tw<2>(a, st);
}

static unsigned grid1d(long n, int threads, long cap = 8192) {
    long g = ceil_div(n, (long)threads);
    if (g > cap) g = cap;
    return (unsigned)(g < 1 ? 1 : g);
}

int instnorm_stats(const float* x, float* stats, long planes, long pe, float eps, hipStream_t st) {
    ProfScope prof(F_STATS, st);
    if (pe <= 8192) {
        hipLaunchKernelGGL(instnorm_stats_wave_kernel, dim3((unsigned)ceil_div(planes, 4L)), dim3(256), 0, st,
                           x, stats, planes, (int)pe, eps);
    } else {
        hipLaunchKernelGGL(instnorm_stats_block_kernel, dim3((unsigned)planes), dim3(256), 0, st, x, stats, pe, eps);
    }
    return check_launch("instnorm_stats");
}

}  // namespace cine

using namespace cine;

extern "C" size_t cine_conv3x3_packed_floats(int cout, int cin) {
    if (cout <= 0 || cin <= 0) return 0;
    const int coutp = ceil_div(cout, 16) * 16, nchunks = ceil_div(cin, kCK);
    return (size_t)nchunks * 9 * kCK * coutp;
}

extern "C" int cine_pack_conv3x3(const float* w, float* packed, int cout, int cin, void* stream) {
    CINE_REQUIRE(w && packed && cout > 0 && cin > 0, CINE_EINVAL, "cine_pack_conv3x3: bad arguments");
    const int coutp = ceil_div(cout, 16) * 16, nchunks = ceil_div(cin, kCK);
    const long total = (long)nchunks * 9 * kCK * coutp;
    ProfScope prof(F_MISC, as_stream(stream));
    hipLaunchKernelGGL(pack_conv3x3_kernel, dim3(grid1d(total, 256)), dim3(256), 0, as_stream(stream),
                       w, packed, cout, cin, coutp, nchunks);
    return check_launch("pack_conv3x3_kernel");
}

static int check_src(const float* x, const float* stats, int c, int mode, const char* what) {
    CINE_REQUIRE(c >= 0, CINE_EINVAL, "%s: negative channel count", what);
    if (c == 0) return CINE_OK;
    CINE_REQUIRE(x, CINE_EINVAL, "%s: null source", what);
    CINE_REQUIRE(mode >= 0 && mode <= 2, CINE_EINVAL, "%s: mode %d", what, mode);
    CINE_REQUIRE(mode == 0 || stats, CINE_EINVAL, "%s: mode %d needs stats", what, mode);
    return CINE_OK;
}

extern "C" int cine_conv3x3_in(const float* x0, const float* stats0, int c0, int mode0, int h0, int w0,
                               const float* x1, const float* stats1, int c1, int mode1, int h1, int w1,
                               const float* wpacked, float* y, float* stats_y,
                               int n, int cout, int h, int w, float eps, float slope, void* stream) {
    CINE_REQUIRE(wpacked && y, CINE_EINVAL, "cine_conv3x3_in: null pointer");
    CINE_REQUIRE(n > 0 && n <= 65535 && cout > 0 && h > 0 && w > 0 && c0 > 0, CINE_EINVAL, "cine_conv3x3_in: bad sizes");
    if (int e = check_src(x0, stats0, c0, mode0, "cine_conv3x3_in(src0)")) return e;
    if (int e = check_src(x1, stats1, c1, mode1, "cine_conv3x3_in(src1)")) return e;
    ConvArgs a{};
    a.s0 = ConvSrc{x0, stats0, c0, mode0, h0, w0};
    a.s1 = ConvSrc{x1, stats1, c1, c1 > 0 ? mode1 : 0, h1, w1};
    a.wp = wpacked; a.y = y; a.n = n; a.cin = c0 + c1; a.cout = cout;
    a.coutp = ceil_div(cout, 16) * 16; a.H = h; a.W = w; a.slope = slope;
    a.nchunks = ceil_div(a.cin, kCK);
    hipStream_t st = as_stream(stream);
    if (int e = conv3x3_dispatch(a, st)) return e;
    if (stats_y) return instnorm_stats(y, stats_y, (long)n * cout, (long)h * w, eps, st);
    return CINE_OK;
}

extern "C" int cine_tconv2x2_in(const float* x, const float* stats_x, int mode, const float* wt,
                                float* y, float* stats_y, int n, int cin, int cout, int h, int w,
                                float eps, float slope, void* stream) {
    CINE_REQUIRE(x && wt && y, CINE_EINVAL, "cine_tconv2x2_in: null pointer");
    CINE_REQUIRE(n > 0 && n <= 65535 && cin > 0 && cout > 0 && h > 0 && w > 0, CINE_EINVAL, "cine_tconv2x2_in: bad sizes");
    CINE_REQUIRE(mode == 0 || (mode == 1 && stats_x), CINE_EINVAL, "cine_tconv2x2_in: mode %d", mode);
    const size_t lds = (size_t)cin * kTPix * sizeof(float);
    CINE_REQUIRE(lds <= 64 * 1024, CINE_EUNSUPPORTED, "cine_tconv2x2_in: cin %d too large", cin);
    hipStream_t st = as_stream(stream);
    { ProfScope prof(F_TCONV, st);
    hipLaunchKernelGGL(tconv2x2_kernel, dim3(ceil_div(h * w, kTPix), n), dim3(256), lds, st,
                       x, stats_x, mode, wt, y, cin, cout, h, w, slope); }
    if (int e = check_launch("tconv2x2_kernel")) return e;
    if (stats_y) return instnorm_stats(y, stats_y, (long)n * cout, (long)4 * h * w, eps, st);
    return CINE_OK;
}

extern "C" int cine_conv1x1_bias(const float* x, const float* stats_x, int mode, const float* wt, const float* bias,
                                 float* y, int n, int cin, int cout, int h, int w, float slope, void* stream) {
    CINE_REQUIRE(x && wt && bias && y, CINE_EINVAL, "cine_conv1x1_bias: null pointer");
    CINE_REQUIRE(n > 0 && n <= 65535 && cin > 0 && cout > 0 && h > 0 && w > 0, CINE_EINVAL, "cine_conv1x1_bias: bad sizes");
    CINE_REQUIRE(mode == 0 || (mode == 1 && stats_x), CINE_EINVAL, "cine_conv1x1_bias: mode %d", mode);
    ProfScope prof(F_CONV1, as_stream(stream));
    hipLaunchKernelGGL(conv1x1_bias_kernel, dim3(grid1d((long)h * w, 256, 64), n), dim3(256), 0, as_stream(stream),
                       x, stats_x, mode, wt, bias, y, cin, cout, (long)h * w, slope);
    return check_launch("conv1x1_bias_kernel");
}

extern "C" int cine_instnorm_stats(const float* x, float* stats, long planes, long plane_elems, float eps, void* stream) {
    CINE_REQUIRE(x && stats && planes > 0 && plane_elems > 0, CINE_EINVAL, "cine_instnorm_stats: bad arguments");
    return instnorm_stats(x, stats, planes, plane_elems, eps, as_stream(stream));
}

extern "C" int cine_instnorm_lrelu_apply(const float* x, const float* stats, float* y, long planes, long plane_elems,
                                         float slope, void* stream) {
    CINE_REQUIRE(x && stats && y && planes > 0 && plane_elems > 0, CINE_EINVAL, "cine_instnorm_lrelu_apply: bad arguments");
    ProfScope prof(F_MISC, as_stream(stream));
    hipLaunchKernelGGL(instnorm_lrelu_apply_kernel, dim3(grid1d(planes * plane_elems, 256)), dim3(256), 0,
                       as_stream(stream), x, stats, y, planes, plane_elems, slope);
    return check_launch("instnorm_lrelu_apply_kernel");
}
