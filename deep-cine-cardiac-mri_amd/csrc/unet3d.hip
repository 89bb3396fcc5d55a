// unet3d.hip -- launch sequence of one 3-D U-Net pass (reference denoisers/unet.py:73-125 with dims = 3:
// Conv3d 3x3x3 + InstanceNorm3d + LeakyReLU, avg_pool3d 2, ConvTranspose3d k2 s2, 1x1x1 conv).
// Same storage discipline as unet.hip.  A volume emits one statistics record per (tile, depth slice); they are
// merged into one record per (sample, channel) right after each launch.
#include <algorithm>
#include "common.h"

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
struct Plan {
    int P; int ds[8], hs[8], wsz[8], ch[8];
    float *skip[8], *pskip[8], *scr[3], *pscr[3], *raw_part;
    float* pooled;                      // the materialised avg_pool3d(act(.)) input of the levels whose conv kernel does not pool on load
};
void build(Plan& p, Bump& b, int n, int d, int h, int w, int chans, int pools) {
    p.P = pools;
    size_t big = 0, bigc = 0, bignp = 0;
    for (int l = 0; l <= pools; ++l) {
        p.ds[l] = l ? p.ds[l - 1] / 2 : d; p.hs[l] = l ? p.hs[l - 1] / 2 : h; p.wsz[l] = l ? p.wsz[l - 1] / 2 : w;
        p.ch[l] = chans << l;
        const size_t e = (size_t)p.ch[l] * p.ds[l] * p.hs[l] * p.wsz[l];
        if (e > big) big = e;
        if ((size_t)p.ch[l] > bigc) bigc = p.ch[l];
        size_t np = (size_t)p.ch[l] * cine_conv_stat_partials3d(p.ch[l], p.ds[l], p.hs[l], p.wsz[l], 0);
        if (np > bignp) bignp = np;
        if (l > 0) {
            np = (size_t)p.ch[l - 1] * cine_conv_stat_partials3d(p.ch[l - 1], p.ds[l], p.hs[l], p.wsz[l], 1);
            if (np > bignp) bignp = np;
        }
    }
    for (int l = 0; l < pools; ++l) {
        p.skip[l] = b.take((size_t)n * p.ch[l] * p.ds[l] * p.hs[l] * p.wsz[l]);
        p.pskip[l] = b.take((size_t)n * p.ch[l] * 3);
    }
    for (int i = 0; i < 3; ++i) { p.scr[i] = b.take((size_t)n * big); p.pscr[i] = b.take((size_t)n * bigc * 3); }
    p.raw_part = b.take((size_t)n * bignp * 3);
    size_t pool = 0;
    for (int l = 1; l <= pools; ++l)
        if (!cine_conv3d_pools_on_load(p.ch[l], p.ds[l], p.hs[l], p.wsz[l]))
            pool = std::max(pool, (size_t)p.ch[l - 1] * p.ds[l] * p.hs[l] * p.wsz[l]);
    p.pooled = pool ? b.take((size_t)n * pool) : nullptr;
}
}  // namespace

extern "C" size_t cine_unet3d_ws_bytes(int n, int d, int h, int w, int in_ch, int out_ch, int chans, int pools) {
    if (n <= 0 || d <= 0 || h <= 0 || w <= 0 || chans <= 0 || pools <= 0 || pools > 6) return 0;
    (void)in_ch; (void)out_ch;
    Plan p; Bump b{nullptr, 0};
    build(p, b, n, d, h, w, chans, pools);
    return b.off;
}

// weights: same order as cine_unet2d_forward, one set: conv3d weights packed with cine_pack_conv3d, transpose convs with
// cine_pack_tconv3d, the final 1x1x1 with cine_pack_conv1x1, then its bias.
extern "C" int cine_unet3d_forward(const float* x, float* y, const void* const* weights, int n, int d, int h, int w,
                                   int in_ch, int out_ch, int chans, int pools, float kSlope, void* ws, size_t ws_bytes, void* stream) {
    CINE_REQUIRE(x && y && weights && ws, CINE_EINVAL, "cine_unet3d_forward: null pointer");
    CINE_REQUIRE(kSlope >= 0.f && kSlope <= 1.f, CINE_EINVAL, "cine_unet3d_forward: LeakyReLU slope %g outside [0, 1]", (double)kSlope);
    CINE_REQUIRE(n > 0 && d > 0 && h > 0 && w > 0 && in_ch > 0 && out_ch > 0 && chans > 0 && pools > 0 && pools <= 6, CINE_EINVAL,
                 "cine_unet3d_forward: bad sizes");
    CINE_REQUIRE((d >> pools) >= 1 && (h >> pools) >= 1 && (w >> pools) >= 1, CINE_EUNSUPPORTED,
                 "cine_unet3d_forward: %dx%dx%d too small for %d pools", d, h, w, pools);
    const size_t need = cine_unet3d_ws_bytes(n, d, h, w, in_ch, out_ch, chans, pools);
    CINE_REQUIRE(ws_bytes >= need, CINE_EWORKSPACE, "cine_unet3d_forward: workspace %zu < %zu", ws_bytes, need);
    const int nptr = 5 * pools + 4;
    for (int i = 0; i < nptr; ++i) CINE_REQUIRE(weights[i], CINE_EINVAL, "cine_unet3d_forward: weights[%d] is null", i);
    Plan p; Bump b{reinterpret_cast<char*>(ws), 0};
    build(p, b, n, d, h, w, chans, pools);
    int wi = 0, e;
    auto W = [&]() { return reinterpret_cast<const float*>(weights[wi++]); };
    // conv + merge of its per-tile statistics into one record per (sample, channel)
    auto conv = [&](const float* x0, const float* p0, int c0, int m0, int d0, int h0, int w0,
                    const float* x1, const float* p1, int c1, int m1, int d1, int h1, int w1,
                    const float* wp, float* yo, float* po, int cout, int l) -> int {
        const int np = cine_conv_stat_partials3d(cout, p.ds[l], p.hs[l], p.wsz[l], 0);
        int r = cine_conv3d_in(x0, p0, 1, c0, m0, d0, h0, w0, x1, p1, 1, c1, m1, d1, h1, w1, wp, nullptr, nullptr, 0,
                               yo, p.raw_part, n, cout, p.ds[l], p.hs[l], p.wsz[l], kEps, kSlope, stream);
        if (r) return r;
        return cine_instnorm_merge(p.raw_part, po, (long)n * cout, np, stream);
    };
    for (int l = 0; l <= pools; ++l) {                        // down path + bottleneck (unet.py:94-99)
        const bool last = l == pools;
        float* out = last ? p.scr[1] : p.skip[l];
        float* pout = last ? p.pscr[1] : p.pskip[l];
        const float* w1 = W();
        if (l == 0) e = conv(x, nullptr, in_ch, 0, d, h, w, nullptr, nullptr, 0, 0, 0, 0, 0, w1, p.scr[0], p.pscr[0], p.ch[0], 0);
        else if (p.pooled && !cine_conv3d_pools_on_load(p.ch[l], p.ds[l], p.hs[l], p.wsz[l])) {
            // the 16-wide tile kernels stage a pooled source element by element (80 us for cfg 4's level 1): pool once, then a plain source
            if ((e = cine_pool3d_act(p.skip[l - 1], p.pskip[l - 1], 1, p.pooled, (long)n * p.ch[l - 1], p.ds[l - 1], p.hs[l - 1], p.wsz[l - 1],
                                     kEps, kSlope, stream))) return e;
            e = conv(p.pooled, nullptr, p.ch[l - 1], 0, p.ds[l], p.hs[l], p.wsz[l], nullptr, nullptr, 0, 0, 0, 0, 0, w1, p.scr[0], p.pscr[0], p.ch[l], l);
        } else e = conv(p.skip[l - 1], p.pskip[l - 1], p.ch[l - 1], 2, p.ds[l - 1], p.hs[l - 1], p.wsz[l - 1],
                        nullptr, nullptr, 0, 0, 0, 0, 0, w1, p.scr[0], p.pscr[0], p.ch[l], l);
        if (e) return e;
        if ((e = conv(p.scr[0], p.pscr[0], p.ch[l], 1, p.ds[l], p.hs[l], p.wsz[l], nullptr, nullptr, 0, 0, 0, 0, 0, W(),
                      out, pout, p.ch[l], l))) return e;
    }
    int cur = 1;
    for (int u = 0; u < pools; ++u) {                         // up path (unet.py:102-123)
        const int l = pools - 1 - u;
        const int a = (cur + 1) % 3, c = (cur + 2) % 3;
        const int npt = cine_conv_stat_partials3d(p.ch[l], p.ds[l + 1], p.hs[l + 1], p.wsz[l + 1], 1);
        if ((e = cine_tconv3d_in(p.scr[cur], p.pscr[cur], 1, 1, W(), p.scr[a], p.raw_part, n, p.ch[l + 1], p.ch[l],
                                 p.ds[l + 1], p.hs[l + 1], p.wsz[l + 1], kEps, kSlope, stream))) return e;
        if ((e = cine_instnorm_merge(p.raw_part, p.pscr[a], (long)n * p.ch[l], npt, stream))) return e;
        if ((e = conv(p.scr[a], p.pscr[a], p.ch[l], 1, 2 * p.ds[l + 1], 2 * p.hs[l + 1], 2 * p.wsz[l + 1],
                      p.skip[l], p.pskip[l], p.ch[l], 1, p.ds[l], p.hs[l], p.wsz[l], W(), p.scr[c], p.pscr[c], p.ch[l], l))) return e;
        if ((e = conv(p.scr[c], p.pscr[c], p.ch[l], 1, p.ds[l], p.hs[l], p.wsz[l], nullptr, nullptr, 0, 0, 0, 0, 0, W(),
                      p.scr[a], p.pscr[a], p.ch[l], l))) return e;
        cur = a;
    }
    const float* wf = W(); const float* bf = W();
    return cine_conv1x1x1_bias(p.scr[cur], p.pscr[cur], 1, 1, wf, bf, y, n, chans, out_ch, d, h, w, kEps, kSlope, stream);
}
