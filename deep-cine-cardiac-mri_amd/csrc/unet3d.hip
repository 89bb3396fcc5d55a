// unet3d.hip -- launch sequence of one 3-D U-Net pass (reference denoisers/unet.py:73-125 with dims = 3:
// Conv3d 3x3x3 + InstanceNorm3d + LeakyReLU, avg_pool3d 2, ConvTranspose3d k2 s2, 1x1x1 conv).
// Same storage discipline as unet.hip.  A volume emits one statistics record per (tile, depth slice); they are
// merged into one record per (sample, channel) right after each launch.
#include <algorithm>
#include "common.h"
#include "grad.h"

using namespace cine;

extern "C" int cine_conv_stat_partials3d(int cout, int d, int h, int w, int is_tconv);
extern "C" int cine_conv3d_in(const float* x0, const float* part0, int np0, int c0, int mode0, int d0, int h0, int w0,
                              const float* x1, const float* part1, int np1, int c1, int mode1, int d1, int h1, int w1,
                              const float* wpacked, const float* bias, const float* addend, int relu,
                              float* y, float* part_y, int n, int cout, int d, int h, int w, float eps, float slope, void* stream);
extern "C" int cine_tconv3d_in(const float* x, const float* part_x, int np_x, int mode, const float* wpacked,
                               float* y, float* part_y, int n, int cin, int cout, int d, int h, int w,
                               float eps, float slope, void* stream);
extern "C" int cine_conv1x1x1_bias(const float* x, const float* part_x, int np_x, int mode, const float* wpacked,
                                   const float* bias, float* y, int n, int cin, int cout, int d, int h, int w,
                                   float eps, float slope, void* stream);
extern "C" int cine_instnorm_merge(const float* part, float* out, long planes, int np, void* stream);
extern "C" int cine_pool3d_act(const float* x, const float* part, int np, float* y, long planes, int d, int h, int w,
                               float eps, float slope, void* stream);
extern "C" int cine_conv3d_pools_on_load(int cout, int d, int h, int w);

namespace {
constexpr float kEps = 1e-5f;
struct Bump {
    char* base; size_t off;
    float* take(size_t floats) {
        const size_t bytes = (floats * sizeof(float) + 255) & ~size_t(255);
        float* p = base ? reinterpret_cast<float*>(base + off) : nullptr;
        off += bytes;
        return p;
    }
};
// Per level l = 0..P (P = bottleneck): mid = first conv of the block, out = its second conv (the skip tensor for l < P, the bottleneck output
// for l = P); per level l < P of the up path: up = transpose-conv output (2 x the extents of level l + 1), ca / cb = the block's two convs.
// Every tensor has ONE merged statistics record per (sample, channel) beside it.
struct Plan {
    int P; int ds[8], hs[8], wsz[8], ch[8];
    float *mid[8], *pmid[8], *out[8], *pout[8], *up[8], *pup[8], *ca[8], *pca[8], *cb[8], *pcb[8], *raw_part;
    float* pooled;                      // the materialised avg_pool3d(act(.)) input of the levels whose conv kernel does not pool on load
    size_t vol(int l) const { return (size_t)ds[l] * hs[l] * wsz[l]; }
    size_t upvol(int l) const { return (size_t)8 * vol(l + 1); }            // extents of up[l]
};
// priv == false: three rotating scratch buffers sized for the largest layer.  priv == true (training): every layer output owns its memory -- the
// backward pass reads all of them.
void build(Plan& p, Bump& b, int n, int d, int h, int w, int chans, int pools, bool priv) {
    p.P = pools;
    size_t big = 0, bigc = 0, bignp = 0;
    for (int l = 0; l <= pools; ++l) {
        p.ds[l] = l ? p.ds[l - 1] / 2 : d; p.hs[l] = l ? p.hs[l - 1] / 2 : h; p.wsz[l] = l ? p.wsz[l - 1] / 2 : w;
        p.ch[l] = chans << l;
        const size_t e = (size_t)p.ch[l] * p.vol(l);
        if (e > big) big = e;
        if ((size_t)p.ch[l] > bigc) bigc = p.ch[l];
        size_t np = (size_t)p.ch[l] * cine_conv_stat_partials3d(p.ch[l], p.ds[l], p.hs[l], p.wsz[l], 0);
        if (np > bignp) bignp = np;
        if (l > 0) {
            np = (size_t)p.ch[l - 1] * cine_conv_stat_partials3d(p.ch[l - 1], p.ds[l], p.hs[l], p.wsz[l], 1);
            if (np > bignp) bignp = np;
        }
    }
    auto elems = [&](int l) { return (size_t)n * p.ch[l] * p.vol(l); };
    auto rec = [&](int l) { return (size_t)n * p.ch[l] * 3; };
    for (int l = 0; l < pools; ++l) { p.out[l] = b.take(elems(l)); p.pout[l] = b.take(rec(l)); }
    if (priv) {
        for (int l = 0; l <= pools; ++l) { p.mid[l] = b.take(elems(l)); p.pmid[l] = b.take(rec(l)); }
        p.out[pools] = b.take(elems(pools)); p.pout[pools] = b.take(rec(pools));
        for (int l = 0; l < pools; ++l) {
            p.up[l] = b.take((size_t)n * p.ch[l] * p.upvol(l)); p.pup[l] = b.take(rec(l));
            p.ca[l] = b.take(elems(l)); p.pca[l] = b.take(rec(l));
            p.cb[l] = b.take(elems(l)); p.pcb[l] = b.take(rec(l));
        }
    } else {
        float *scr[3], *pscr[3];
        for (int i = 0; i < 3; ++i) { scr[i] = b.take((size_t)n * big); pscr[i] = b.take((size_t)n * bigc * 3); }
        for (int l = 0; l <= pools; ++l) { p.mid[l] = scr[0]; p.pmid[l] = pscr[0]; }
        p.out[pools] = scr[1]; p.pout[pools] = pscr[1];
        int cur = 1;
        for (int u = 0; u < pools; ++u) {
            const int l = pools - 1 - u;
            const int a = (cur + 1) % 3, c = (cur + 2) % 3;
            p.up[l] = scr[a]; p.pup[l] = pscr[a];
            p.ca[l] = scr[c]; p.pca[l] = pscr[c];
            p.cb[l] = scr[a]; p.pcb[l] = pscr[a];
            cur = a;
        }
    }
    p.raw_part = b.take((size_t)n * bignp * 3);
    size_t pool = 0;
    for (int l = 1; l <= pools; ++l)
        if (!cine_conv3d_pools_on_load(p.ch[l], p.ds[l], p.hs[l], p.wsz[l]))
            pool = std::max(pool, (size_t)p.ch[l - 1] * p.vol(l));
    p.pooled = pool ? b.take((size_t)n * pool) : nullptr;
}
bool sizes_ok(int n, int d, int h, int w, int chans, int pools) { return n > 0 && d > 0 && h > 0 && w > 0 && chans > 0 && pools > 0 && pools <= 6; }
}  // namespace

extern "C" size_t cine_unet3d_ws_bytes(int n, int d, int h, int w, int in_ch, int out_ch, int chans, int pools) {
    if (!sizes_ok(n, d, h, w, chans, pools)) return 0;
    (void)in_ch; (void)out_ch;
    Plan p; Bump b{nullptr, 0};
    build(p, b, n, d, h, w, chans, pools, false);
    return b.off;
}
// training: every layer output keeps its own memory (cine_unet3d_backward reads all of them)
extern "C" size_t cine_unet3d_train_ws_bytes(int n, int d, int h, int w, int in_ch, int out_ch, int chans, int pools) {
    if (!sizes_ok(n, d, h, w, chans, pools)) return 0;
    (void)in_ch; (void)out_ch;
    Plan p; Bump b{nullptr, 0};
    build(p, b, n, d, h, w, chans, pools, true);
    return b.off;
}

static int unet3d_forward_impl(const float* x, float* y, const void* const* weights, int n, int d, int h, int w,
                               int in_ch, int out_ch, int chans, int pools, float kSlope, void* ws, size_t ws_bytes, void* stream, bool train,
                               const float* drop = nullptr) {
    CINE_REQUIRE(x && y && weights && ws, CINE_EINVAL, "cine_unet3d_forward: null pointer");
    CINE_REQUIRE(kSlope >= 0.f && kSlope <= 1.f, CINE_EINVAL, "cine_unet3d_forward: LeakyReLU slope %g outside [0, 1]", (double)kSlope);
    CINE_REQUIRE(sizes_ok(n, d, h, w, chans, pools) && in_ch > 0 && out_ch > 0, CINE_EINVAL, "cine_unet3d_forward: bad sizes");
    CINE_REQUIRE((d >> pools) >= 1 && (h >> pools) >= 1 && (w >> pools) >= 1, CINE_EUNSUPPORTED,
                 "cine_unet3d_forward: %dx%dx%d too small for %d pools", d, h, w, pools);
    const size_t need = train ? cine_unet3d_train_ws_bytes(n, d, h, w, in_ch, out_ch, chans, pools) : cine_unet3d_ws_bytes(n, d, h, w, in_ch, out_ch, chans, pools);
    CINE_REQUIRE(ws_bytes >= need, CINE_EWORKSPACE, "cine_unet3d_forward: workspace %zu < %zu", ws_bytes, need);
    const int nptr = 5 * pools + 4;
    for (int i = 0; i < nptr; ++i) CINE_REQUIRE(weights[i], CINE_EINVAL, "cine_unet3d_forward: weights[%d] is null", i);
    Plan p; Bump b{reinterpret_cast<char*>(ws), 0};
    build(p, b, n, d, h, w, chans, pools, train);
    int wi = 0, e, ci = 0;
    auto W = [&]() { return reinterpret_cast<const float*>(weights[wi++]); };
    const DropMap dm{drop, n, 0, chans, pools};          // Dropout3d behind every ConvBlock activation (unet.py:163,167): the 2-D layout, one (n, ch) block per 3x3x3 conv
    // conv + merge of its per-tile statistics into one record per (sample, channel) (+ the Dropout multipliers folded into that record)
    auto conv = [&](const float* x0, const float* p0, int c0, int m0, int d0, int h0, int w0,
                    const float* x1, const float* p1, int c1, int m1, int d1, int h1, int w1,
                    const float* wp, float* yo, float* po, int cout, int l) -> int {
        const int np = cine_conv_stat_partials3d(cout, p.ds[l], p.hs[l], p.wsz[l], 0);
        int r = cine_conv3d_in(x0, p0, 1, c0, m0, d0, h0, w0, x1, p1, 1, c1, m1, d1, h1, w1, wp, nullptr, nullptr, 0,
                               yo, p.raw_part, n, cout, p.ds[l], p.hs[l], p.wsz[l], kEps, kSlope, stream);
        if (r) return r;
        const int conv_id = ci++;          // launch order == DropMap's: down / bottleneck convs 0 .. 2 P + 1, then the up path from the coarsest level
        if ((r = cine_instnorm_merge(p.raw_part, po, (long)n * cout, np, stream))) return r;
        return drop ? launch_dropout_stats(po, 1, (long)n * cout, dm.at(conv_id), as_stream(stream)) : CINE_OK;
    };
    for (int l = 0; l <= pools; ++l) {                        // down path + bottleneck (unet.py:94-99)
        const float* w1 = W();
        if (l == 0) e = conv(x, nullptr, in_ch, 0, d, h, w, nullptr, nullptr, 0, 0, 0, 0, 0, w1, p.mid[0], p.pmid[0], p.ch[0], 0);
        else if (p.pooled && !cine_conv3d_pools_on_load(p.ch[l], p.ds[l], p.hs[l], p.wsz[l])) {
            // the 16-wide tile kernels stage a pooled source element by element (80 us for cfg 4's level 1): pool once, then a plain source
            if ((e = cine_pool3d_act(p.out[l - 1], p.pout[l - 1], 1, p.pooled, (long)n * p.ch[l - 1], p.ds[l - 1], p.hs[l - 1], p.wsz[l - 1],
                                     kEps, kSlope, stream))) return e;
            e = conv(p.pooled, nullptr, p.ch[l - 1], 0, p.ds[l], p.hs[l], p.wsz[l], nullptr, nullptr, 0, 0, 0, 0, 0, w1, p.mid[l], p.pmid[l], p.ch[l], l);
        } else e = conv(p.out[l - 1], p.pout[l - 1], p.ch[l - 1], 2, p.ds[l - 1], p.hs[l - 1], p.wsz[l - 1],
                        nullptr, nullptr, 0, 0, 0, 0, 0, w1, p.mid[l], p.pmid[l], p.ch[l], l);
        if (e) return e;
        if ((e = conv(p.mid[l], p.pmid[l], p.ch[l], 1, p.ds[l], p.hs[l], p.wsz[l], nullptr, nullptr, 0, 0, 0, 0, 0, W(),
                      p.out[l], p.pout[l], p.ch[l], l))) return e;
    }
    const float* cur = p.out[pools]; const float* pcur = p.pout[pools];
    for (int l = pools - 1; l >= 0; --l) {                    // up path (unet.py:102-123)
        const int npt = cine_conv_stat_partials3d(p.ch[l], p.ds[l + 1], p.hs[l + 1], p.wsz[l + 1], 1);
        if ((e = cine_tconv3d_in(cur, pcur, 1, 1, W(), p.up[l], p.raw_part, n, p.ch[l + 1], p.ch[l],
                                 p.ds[l + 1], p.hs[l + 1], p.wsz[l + 1], kEps, kSlope, stream))) return e;
        if ((e = cine_instnorm_merge(p.raw_part, p.pup[l], (long)n * p.ch[l], npt, stream))) return e;
        if ((e = conv(p.up[l], p.pup[l], p.ch[l], 1, 2 * p.ds[l + 1], 2 * p.hs[l + 1], 2 * p.wsz[l + 1],
                      p.out[l], p.pout[l], p.ch[l], 1, p.ds[l], p.hs[l], p.wsz[l], W(), p.ca[l], p.pca[l], p.ch[l], l))) return e;
        if ((e = conv(p.ca[l], p.pca[l], p.ch[l], 1, p.ds[l], p.hs[l], p.wsz[l], nullptr, nullptr, 0, 0, 0, 0, 0, W(),
                      p.cb[l], p.pcb[l], p.ch[l], l))) return e;
        cur = p.cb[l]; pcur = p.pcb[l];
    }
    const float* wf = W(); const float* bf = W();
    return cine_conv1x1x1_bias(cur, pcur, 1, 1, wf, bf, y, n, chans, out_ch, d, h, w, kEps, kSlope, stream);
}

// weights: same order as cine_unet2d_forward, one set: conv3d weights packed with cine_pack_conv3d, transpose convs with
// cine_pack_tconv3d, the final 1x1x1 with cine_pack_conv1x1, then its bias.
extern "C" int cine_unet3d_forward(const float* x, float* y, const void* const* weights, int n, int d, int h, int w,
                                   int in_ch, int out_ch, int chans, int pools, float slope, void* ws, size_t ws_bytes, void* stream) {
    return unet3d_forward_impl(x, y, weights, n, d, h, w, in_ch, out_ch, chans, pools, slope, ws, ws_bytes, stream, false);
}
// The same launches with every raw layer output and its statistics record kept in `ws` (cine_unet3d_train_ws_bytes) for cine_unet3d_backward.
extern "C" int cine_unet3d_forward_train(const float* x, float* y, const void* const* weights, int n, int d, int h, int w,
                                         int in_ch, int out_ch, int chans, int pools, float slope, void* ws, size_t ws_bytes, void* stream) {
    return unet3d_forward_impl(x, y, weights, n, d, h, w, in_ch, out_ch, chans, pools, slope, ws, ws_bytes, stream, true);
}
// ... with Dropout3d (training mode, drop_prob > 0): `drop` as for cine_unet2d_forward_branches (cine_unet2d_drop_floats(n, chans, pools) multipliers)
extern "C" int cine_unet3d_forward_train_drop(const float* x, float* y, const void* const* weights, int n, int d, int h, int w,
                                              int in_ch, int out_ch, int chans, int pools, float slope, void* ws, size_t ws_bytes, const float* drop, void* stream) {
    return unet3d_forward_impl(x, y, weights, n, d, h, w, in_ch, out_ch, chans, pools, slope, ws, ws_bytes, stream, true, drop);
}

// ---------------------------------------------------------------- backward pass (training)
// The 2-D sequence of unet.hip on volumes.  What differs:
//  * InstanceNorm + LeakyReLU backward sees a volume as a plane of (d h, w); the depth crop of the zero-padded transpose-conv output is a
//    shorter plane, an in-plane crop (odd extents) and the 2x2x2 pool adjoint are the volume pieces of grad.h (types 5 / 6);
//  * the 3x3x3 weight gradient is three 3x3 weight gradients -- one per depth tap kz, over the slice pairs (x[z + kz - 1], g[z]) with the depth
//    slices as the samples of the MFMA weight-gradient kernel, and one reduction of the three taps' partial sums (launch_wgrad27).
//    The kernel addresses (sample, channel, h, w): the (re-activated, concatenated, zero-padded, pooled) conv input and the output gradient are
//    written once per layer in depth-major order (vol_slices_kernel) -- on the side stream, beside the input-gradient chain;
//  * the k2 s2 transpose conv: one space-to-depth copy of its output gradient (rows 8 c + 4 dz + 2 dy + dx), then the 1x1x1 kernels.
namespace {
struct VolSliceArgs { Src s0, s1; float* out; int C, D, H, W, vol; float eps, slope; };

// out[z][cg][y][x] = channel cg of cat(s0, s1) after its on-load transform (mode 0 as is, 1 InstanceNorm + LeakyReLU, 2 the same + avg_pool3d 2),
// zero outside a source's extents; sources are volumes (n, c, d, h, w) with one merged record per (n, c); this call: sample a.vol.
template <bool VEC>
__global__ __launch_bounds__(256) void vol_slices_kernel(VolSliceArgs a) {
    __shared__ float st[2];
    const int cg = blockIdx.y, z = blockIdx.z;
    const bool f0 = cg < a.s0.c;
    const Src& s = f0 ? a.s0 : a.s1;
    const int cl = f0 ? cg : cg - a.s0.c;
    if (threadIdx.x == 0) {
        float2 mr = make_float2(0.f, 1.f);
        if (s.mode != 0) mr = merge_partials(s.part + ((long)a.vol * s.c + cl) * s.np * 3, s.np, a.eps);
        st[0] = mr.y; st[1] = -mr.x * mr.y;
    }
    __syncthreads();
    float* o = a.out + ((long)z * a.C + cg) * a.H * a.W;
    if constexpr (VEC) {        // modes 0 / 1, rows of whole aligned float4 pieces, the source as wide as the output
        const int w4 = a.W >> 2;
        const float sc = st[0], sh = st[1];
        const float* p = s.x + (((long)a.vol * s.c + cl) * s.d + min(z, s.d - 1)) * (long)s.h * s.w;
        for (int e = blockIdx.x * 256 + threadIdx.x; e < a.H * w4; e += gridDim.x * 256) {
            const int y = e / w4, x = (e - y * w4) << 2;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (z < s.d && y < s.h) {
                v = *reinterpret_cast<const float4*>(p + (long)y * s.w + x);
                if (s.mode == 1) { v.x = act(v.x, sc, sh, a.slope); v.y = act(v.y, sc, sh, a.slope); v.z = act(v.z, sc, sh, a.slope); v.w = act(v.w, sc, sh, a.slope); }
            }
            *reinterpret_cast<float4*>(o + (long)y * a.W + x) = v;
        }
    } else {
        const float* tab = st - 2 * cl;                     // fetch_scalar indexes its table with the channel
        for (int e = blockIdx.x * 256 + threadIdx.x; e < a.H * a.W; e += gridDim.x * 256) {
            const int y = e / a.W, x = e - y * a.W;
            o[e] = fetch_scalar(s, a.vol, cl, z, y, x, tab, a.slope);
        }
    }
}
int launch_vol_slices(const Src& s0, const Src& s1, float* out, int vol, int C, int D, int H, int W, float eps, float slope, hipStream_t st) {
    CINE_REQUIRE(s0.x && out && s0.c + s1.c == C && C > 0 && C <= 65535 && D > 0 && D <= 65535 && H > 0 && W > 0, CINE_EINVAL, "vol_slices: bad arguments");
    VolSliceArgs a{s0, s1, out, C, D, H, W, vol, eps, slope};
    auto vec_ok = [&](const Src& s) {
        return s.c == 0 || (s.mode <= 1 && s.w == W && reinterpret_cast<uintptr_t>(s.x) % 16 == 0 && ((long)s.h * s.w) % 4 == 0);
    };
    const bool vec = W % 4 == 0 && vec_ok(s0) && vec_ok(s1) && reinterpret_cast<uintptr_t>(out) % 16 == 0;
    const long per = vec ? (long)H * (W / 4) : (long)H * W;
    const dim3 grid((unsigned)std::max(1L, std::min(64L, ceil_div(per, 1024L))), (unsigned)C, (unsigned)D);
    ProfScope prof(F_MISC, st);
    if (vec) hipLaunchKernelGGL(vol_slices_kernel<true>, grid, dim3(256), 0, st, a);
    else hipLaunchKernelGGL(vol_slices_kernel<false>, grid, dim3(256), 0, st, a);
    return check_launch("vol_slices_kernel");
}

// space-to-depth of the transpose conv's output gradient: out (n, 8 c, d, h, w)[8 c + 4 dz + 2 dy + dx][z][y][x] = g (n, c, 2d, 2h, 2w)[c][2z + dz][2y + dy][2x + dx].
// One thread = one pair of input columns (both dx).
__global__ __launch_bounds__(256) void s2d3d_kernel(const float* __restrict__ g, float* __restrict__ out, long planes, int d, int h, int w) {
    const long total = planes * 2 * d * 2 * h * w;
    for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
        const int x = (int)(e % w); long r = e / w;
        const int yy = (int)(r % (2 * h)); r /= 2 * h;
        const int zz = (int)(r % (2 * d)); const long pl = r / (2 * d);
        const float2 v = *reinterpret_cast<const float2*>(g + ((pl * 2 * d + zz) * 2 * h + yy) * 2 * w + 2 * x);
        float* o = out + (((pl * 8 + 4 * (zz & 1) + 2 * (yy & 1)) * d + (zz >> 1)) * h + (yy >> 1)) * (long)w + x;
        o[0] = v.x; o[(long)d * h * w] = v.y;
    }
}

struct BwdPlan {
    float *A, *B[2], *cat[8], *pool, *wg, *xs, *gs, *inb, *zero;
    size_t wg_floats, inb_floats;
};
void build_bwd(BwdPlan& q, const Plan& p, Bump& b, int n, int in_ch, int out_ch) {
    const int P = p.P;
    auto elems = [&](int l) { return (size_t)n * p.ch[l] * p.vol(l); };
    size_t big = (size_t)n * in_ch * p.vol(0);
    for (int l = 0; l <= P; ++l) big = std::max(big, elems(l));
    q.A = b.take(big); q.B[0] = b.take(big); q.B[1] = b.take(big);
    for (int l = 0; l < P; ++l) q.cat[l] = b.take(2 * elems(l));
    size_t pool = 16;
    for (int l = 1; l <= P; ++l) pool = std::max(pool, (size_t)n * p.ch[l - 1] * p.vol(l));
    q.pool = b.take(pool);
    size_t wg = (size_t)n * out_ch, xs = 16, gs = 16, inb = 16;
    for (int l = 0; l <= P; ++l) {
        const int cin1 = l ? p.ch[l - 1] : in_ch;
        wg = std::max(wg, 3 * wgrad_ws_floats(p.ch[l], cin1, 9, p.ds[l]));            // (three depth taps: launch_wgrad27)
        wg = std::max(wg, 3 * wgrad_ws_floats(p.ch[l], p.ch[l], 9, p.ds[l]));
        xs = std::max(xs, (size_t)std::max(cin1, p.ch[l]) * p.vol(l));
        gs = std::max(gs, (size_t)p.ch[l] * p.vol(l));
        if (l < P) {
            wg = std::max(wg, 3 * wgrad_ws_floats(p.ch[l], 2 * p.ch[l], 9, p.ds[l]));
            wg = std::max(wg, wgrad_ws_floats(8 * p.ch[l], p.ch[l + 1], 1, n));
            xs = std::max(xs, (size_t)2 * p.ch[l] * p.vol(l));
        }
        inb = std::max(inb, in_lrelu_bwd_ws_floats(n, p.ch[l], p.ds[l] * p.hs[l], p.wsz[l]));
    }
    wg = std::max(wg, wgrad_ws_floats(out_ch, p.ch[0], 1, n));
    q.wg_floats = wg; q.wg = b.take(wg);
    q.xs = b.take(xs); q.gs = b.take(gs);                    // one sample's conv input / output gradient in depth-major order (side stream only)
    q.inb_floats = inb; q.inb = b.take(inb);
    q.zero = b.take((size_t)p.ch[P] + 16);                   // the zero bias of the 1x1x1 input-gradient convolutions
}
}  // namespace

extern "C" size_t cine_unet3d_backward_ws_bytes(int n, int d, int h, int w, int in_ch, int out_ch, int chans, int pools) {
    if (!sizes_ok(n, d, h, w, chans, pools) || in_ch <= 0 || out_ch <= 0) return 0;
    Plan p; Bump b0{nullptr, 0};
    build(p, b0, n, d, h, w, chans, pools, true);
    BwdPlan q; Bump b{nullptr, 0};
    build_bwd(q, p, b, n, in_ch, out_ch);
    return b.off;
}

// Gradients of cine_unet3d_forward_train.  `fwd_ws`: the workspace that call filled; x its input, gy = d loss / d y (n, out_ch, d, h, w).
// `wdgrad`: host array of device pointers in the order of `weights`, holding the INPUT-GRADIENT packings: cine_pack_conv3d of the weight with its
// taps flipped and (cout, cin) transposed for the 3x3x3 convs; cine_pack_conv1x1 of the (cin, 8 cout) matrix of a transpose conv (its weight
// (cin, cout, 2, 2, 2) as is); cine_pack_conv1x1 of the transposed (chans, out_ch) matrix of the final conv; the bias slot is unused.
// `grads`: host array in the same order of device pointers to the weight gradients in the parameters' own layouts ((cout, cin, 3, 3, 3),
// (cin, cout, 2, 2, 2), (out_ch, chans), (out_ch)); they are ACCUMULATED into (+=).  gx (n, in_ch, d, h, w) may be NULL.
// Weight gradients run on the calling thread's side stream (cine_set_side_stream) when it has one.
extern "C" int cine_unet3d_backward_drop(const float* x, const float* gy, const void* const* wdgrad, void* const* grads,
                                         int n, int d, int h, int w, int in_ch, int out_ch, int chans, int pools, float kSlope,
                                         const void* fwd_ws, size_t fwd_ws_bytes, void* ws, size_t ws_bytes, float* gx, const float* drop, void* stream);
extern "C" int cine_unet3d_backward(const float* x, const float* gy, const void* const* wdgrad, void* const* grads,
                                    int n, int d, int h, int w, int in_ch, int out_ch, int chans, int pools, float kSlope,
                                    const void* fwd_ws, size_t fwd_ws_bytes, void* ws, size_t ws_bytes, float* gx, void* stream) {
    return cine_unet3d_backward_drop(x, gy, wdgrad, grads, n, d, h, w, in_ch, out_ch, chans, pools, kSlope, fwd_ws, fwd_ws_bytes, ws, ws_bytes, gx, nullptr, stream);
}
extern "C" int cine_unet3d_backward_drop(const float* x, const float* gy, const void* const* wdgrad, void* const* grads,
                                         int n, int d, int h, int w, int in_ch, int out_ch, int chans, int pools, float kSlope,
                                         const void* fwd_ws, size_t fwd_ws_bytes, void* ws, size_t ws_bytes, float* gx, const float* drop, void* stream) {
    CINE_REQUIRE(x && gy && wdgrad && grads && fwd_ws && ws, CINE_EINVAL, "cine_unet3d_backward: null pointer");
    CINE_REQUIRE(kSlope >= 0.f && kSlope <= 1.f, CINE_EINVAL, "cine_unet3d_backward: LeakyReLU slope %g outside [0, 1]", (double)kSlope);
    CINE_REQUIRE(sizes_ok(n, d, h, w, chans, pools) && in_ch > 0 && out_ch > 0 && n <= 65535, CINE_EINVAL, "cine_unet3d_backward: bad sizes");
    CINE_REQUIRE((d >> pools) >= 1 && (h >> pools) >= 1 && (w >> pools) >= 1, CINE_EUNSUPPORTED,
                 "cine_unet3d_backward: %dx%dx%d too small for %d pools", d, h, w, pools);
    CINE_REQUIRE(fwd_ws_bytes >= cine_unet3d_train_ws_bytes(n, d, h, w, in_ch, out_ch, chans, pools), CINE_EWORKSPACE,
                 "cine_unet3d_backward: forward workspace too small");
    CINE_REQUIRE(ws_bytes >= cine_unet3d_backward_ws_bytes(n, d, h, w, in_ch, out_ch, chans, pools), CINE_EWORKSPACE,
                 "cine_unet3d_backward: workspace too small");
    const int nptr = 5 * pools + 4;
    for (int i = 0; i < nptr; ++i) {
        CINE_REQUIRE(grads[i], CINE_EINVAL, "cine_unet3d_backward: grads[%d] is null", i);
        CINE_REQUIRE(wdgrad[i] || i == nptr - 1, CINE_EINVAL, "cine_unet3d_backward: wdgrad[%d] is null", i);
    }
    Plan p; Bump bf{const_cast<char*>(reinterpret_cast<const char*>(fwd_ws)), 0};
    build(p, bf, n, d, h, w, chans, pools, true);
    BwdPlan q; Bump bb{reinterpret_cast<char*>(ws), 0};
    build_bwd(q, p, bb, n, in_ch, out_ch);
    hipStream_t st = as_stream(stream);
    const int P = pools;
    int e;
    auto wd = [&](int i) { return reinterpret_cast<const float*>(wdgrad[i]); };
    auto gr = [&](int i) { return reinterpret_cast<float*>(grads[i]); };
    auto i_down = [&](int l, int k) { return 2 * l + k; };
    auto i_up = [&](int l, int k) { return 2 * P + 2 + 3 * (P - 1 - l) + k; };   // k = 0 tconv, 1 conv1, 2 conv2
    const int i_fin = 5 * P + 2, i_bias = 5 * P + 3;
    const Src none{nullptr, nullptr, 0, 0, 0, 0, 0, 0, 1};
    auto vsrc = [&](const float* t, const float* part, int c, int mode, int dd, int hh, int ww) { return Src{t, part, c, mode, hh, ww, 1, 2, dd}; };
    CINE_REQUIRE(hipMemsetAsync(q.zero, 0, ((size_t)p.ch[P] + 16) * sizeof(float), st) == hipSuccess, CINE_EHIP, "cine_unet3d_backward: hipMemsetAsync failed");
    SideLane lane(st);              // weight gradients on the side stream (grad.h); g alternates between q.B[0] / q.B[1]
    auto next_g = [&]() { lane.before_write(); return q.B[lane.slot()]; };
    auto on_side = [&](auto&& launch) { hipStream_t sw = lane.fork(); const int err = launch(sw); lane.launched(); return err; };
    // weight gradient of a 3x3x3 conv of level l over cat(s0, s1) from g (n, rows, dims of l)
    auto wgrad27 = [&](const Src& s0, const Src& s1, const float* g, int rows, int l, int wi) {
        const int D = p.ds[l], H = p.hs[l], W = p.wsz[l], cin = s0.c + s1.c;
        const long hw = (long)H * W;
        return on_side([&](hipStream_t sw) -> int {
            for (int v = 0; v < n; ++v) {
                if (int r = launch_vol_slices(s0, s1, q.xs, v, cin, D, H, W, kEps, kSlope, sw)) return r;
                const Src gsrc{g + (long)v * rows * D * hw, nullptr, rows, 0, H, W, 0, 2, D};
                if (int r = launch_vol_slices(gsrc, none, q.gs, 0, rows, D, H, W, kEps, kSlope, sw)) return r;
                WgArgs a[3] = {};
                for (int kz = 0; kz < 3; ++kz) {
                    const int dz = kz - 1, z0 = std::max(0, -dz), z1 = D - std::max(0, dz);
                    if (z1 <= z0) continue;
                    a[kz].s0 = Src{q.xs + (long)(z0 + dz) * cin * hw, nullptr, cin, 0, H, W, 0, 0, 1}; a[kz].s1 = none; a[kz].cin = cin;
                    a[kz].g = q.gs + (long)z0 * rows * hw; a[kz].g_mode = 0; a[kz].rows = rows;
                    a[kz].n = z1 - z0; a[kz].H = H; a[kz].W = W; a[kz].set_split = z1 - z0; a[kz].eps = kEps; a[kz].slope = kSlope;
                }
                if (int r = launch_wgrad27(a, gr(wi), q.wg, q.wg_floats, sw)) return r;
            }
            return CINE_OK;
        });
    };
    auto dgrad27 = [&](const float* g, int wi, float* out, int cout, int cin, int l) {
        return cine_conv3d_in(g, nullptr, 0, cout, 0, p.ds[l], p.hs[l], p.wsz[l], nullptr, nullptr, 0, 0, 0, 0, 0, 0, wd(wi), nullptr, nullptr, 0,
                              out, nullptr, n, cin, p.ds[l], p.hs[l], p.wsz[l], kEps, kSlope, stream);
    };
    // a consumer's gradient tensor (n, c_total, gd, gh, gw) as a piece of the tensor (c, td, th, tw)
    auto window = [&](const float* g, int c_total, int c_off, int gd, int gh, int gw, int th, int tw) {
        if (gh == th && gw == tw) return GradPiece{g, 1, c_total, c_off, gd * gh, gw, 0, 0};
        return GradPiece{g, 5, c_total, c_off, gh, gw, gd, th};
    };
    const GradPiece nopiece{nullptr, 0, 0, 0, 0, 0, 0, 0};
    const DropMap dm{drop, n, 0, chans, pools};
    auto inbwd = [&](const float* r, const float* part, int c, int td, int th, int tw, GradPiece pa, GradPiece pb, float* out, int conv = -1) {
        InBwdArgs a{r, part, 1, pa, pb, out, n, c, td * th, tw, kEps, kSlope, conv >= 0 ? dm.at(conv) : nullptr};
        return launch_in_lrelu_bwd_split(a, q.inb, q.inb_floats, st);
    };

    // ---- final 1x1x1 conv + bias (unet.py:69)
    const long vol0 = (long)p.vol(0);
    if ((e = launch_bias_grad(gy, n, out_ch, vol0, n, gr(i_bias), nullptr, q.wg, q.wg_floats, st))) return e;
    {
        WgArgs a{}; a.s0 = Src{p.cb[0], p.pcb[0], chans, 1, d * h, w, 1, 0, 1}; a.s1 = none; a.cin = chans;
        a.g = gy; a.g_mode = 0; a.rows = out_ch; a.n = n; a.H = d * h; a.W = w; a.set_split = n; a.eps = kEps; a.slope = kSlope;
        if ((e = on_side([&](hipStream_t sw) { return launch_wgrad(a, 1, 2, gr(i_fin), nullptr, q.wg, q.wg_floats, sw); }))) return e;
    }
    if ((e = cine_conv1x1x1_bias(gy, nullptr, 0, 0, wd(i_fin), q.zero, q.A, n, out_ch, chans, d, h, w, kEps, kSlope, stream))) return e;   // A = d / d act(cb_0)

    // ---- up path, level 0 first (reverse of unet.py:102-123)
    for (int l = 0; l < P; ++l) {
        const int c = p.ch[l], D = p.ds[l], H = p.hs[l], W = p.wsz[l];
        const int Du = 2 * p.ds[l + 1], Hu = 2 * p.hs[l + 1], Wu = 2 * p.wsz[l + 1];      // extents of the transpose-conv output
        float* B = next_g();                                   // second conv of the block: cb = conv(act(ca))
        if ((e = inbwd(p.cb[l], p.pcb[l], c, D, H, W, window(q.A, c, 0, D, H, W, H, W), nopiece, B, dm.up(l, 1)))) return e;
        if ((e = wgrad27(vsrc(p.ca[l], p.pca[l], c, 1, D, H, W), none, B, c, l, i_up(l, 2)))) return e;
        if ((e = dgrad27(B, i_up(l, 2), q.A, c, c, l))) return e;
        B = next_g();                                          // first conv: ca = conv(cat(act(up) zero-padded, act(skip)))
        if ((e = inbwd(p.ca[l], p.pca[l], c, D, H, W, window(q.A, c, 0, D, H, W, H, W), nopiece, B, dm.up(l, 0)))) return e;
        if ((e = wgrad27(vsrc(p.up[l], p.pup[l], c, 1, Du, Hu, Wu), vsrc(p.out[l], p.pout[l], c, 1, D, H, W), B, c, l, i_up(l, 1)))) return e;
        if ((e = dgrad27(B, i_up(l, 1), q.cat[l], c, 2 * c, l))) return e;
        // transpose conv: up = tconv(act(cur)), cur = cb[l + 1] or the bottleneck output
        if ((e = inbwd(p.up[l], p.pup[l], c, Du, Hu, Wu, window(q.cat[l], 2 * c, 0, D, H, W, Hu, Wu), nopiece, q.A))) return e;
        const int c1 = p.ch[l + 1], D1 = p.ds[l + 1], H1 = p.hs[l + 1], W1 = p.wsz[l + 1];
        B = next_g();
        {
            ProfScope prof(F_MISC, st);
            const long total = (long)n * c * Du * Hu * W1;
            hipLaunchKernelGGL(s2d3d_kernel, dim3((unsigned)std::max(1L, std::min(ceil_div(total, 256L), 8192L))), dim3(256), 0, st, q.A, B, (long)n * c, D1, H1, W1);
            if ((e = check_launch("s2d3d_kernel"))) return e;
        }
        const bool bott = l + 1 == P;
        const float* cur = bott ? p.out[P] : p.cb[l + 1];
        const float* pcur = bott ? p.pout[P] : p.pcb[l + 1];
        {
            WgArgs a{}; a.s0 = Src{cur, pcur, c1, 1, D1 * H1, W1, 1, 0, 1}; a.s1 = none; a.cin = c1;
            a.g = B; a.g_mode = 0; a.rows = 8 * c; a.n = n; a.H = D1 * H1; a.W = W1; a.set_split = n; a.eps = kEps; a.slope = kSlope;
            if ((e = on_side([&](hipStream_t sw) { return launch_wgrad(a, 1, 1, gr(i_up(l, 0)), nullptr, q.wg, q.wg_floats, sw); }))) return e;
        }
        if ((e = cine_conv1x1x1_bias(B, nullptr, 0, 0, wd(i_up(l, 0)), q.zero, q.A, n, 8 * c, c1, D1, H1, W1, kEps, kSlope, stream))) return e;   // A = d / d act(cur)
    }
    // ---- bottleneck and down path (reverse of unet.py:94-99)
    for (int l = P; l >= 0; --l) {
        const int c = p.ch[l], D = p.ds[l], H = p.hs[l], W = p.wsz[l];
        float* B = next_g();
        if (l == P) {
            if ((e = inbwd(p.out[l], p.pout[l], c, D, H, W, window(q.A, c, 0, D, H, W, H, W), nopiece, B, DropMap::down(l, 1)))) return e;
        } else {    // the skip tensor feeds the concat (second half of cat[l]) and the 2x2x2 average pool
            const GradPiece pool{q.pool, 6, c, 0, p.hs[l + 1], p.wsz[l + 1], p.ds[l + 1], H};
            if ((e = inbwd(p.out[l], p.pout[l], c, D, H, W, window(q.cat[l], 2 * c, c, D, H, W, H, W), pool, B, DropMap::down(l, 1)))) return e;
        }
        if ((e = wgrad27(vsrc(p.mid[l], p.pmid[l], c, 1, D, H, W), none, B, c, l, i_down(l, 1)))) return e;
        if ((e = dgrad27(B, i_down(l, 1), q.A, c, c, l))) return e;
        B = next_g();
        if ((e = inbwd(p.mid[l], p.pmid[l], c, D, H, W, window(q.A, c, 0, D, H, W, H, W), nopiece, B, DropMap::down(l, 0)))) return e;
        if (l > 0) {
            const int cp = p.ch[l - 1];
            if ((e = wgrad27(vsrc(p.out[l - 1], p.pout[l - 1], cp, 2, p.ds[l - 1], p.hs[l - 1], p.wsz[l - 1]), none, B, c, l, i_down(l, 0)))) return e;
            if ((e = dgrad27(B, i_down(l, 0), q.pool, c, cp, l))) return e;
        } else {
            if ((e = wgrad27(vsrc(x, nullptr, in_ch, 0, d, h, w), none, B, c, 0, i_down(0, 0)))) return e;
            if (gx && (e = dgrad27(B, i_down(0, 0), gx, c, in_ch, 0))) return e;
        }
    }
    lane.join();
    return CINE_OK;
}
