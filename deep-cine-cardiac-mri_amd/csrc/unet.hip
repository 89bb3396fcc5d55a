// unet.hip -- launch sequence of one 2-D U-Net pass (reference denoisers/unet.py:73-125).
//
// Activations stay raw (pre-InstanceNorm) in HBM together with their per-(sample, channel)
// {mean, rstd}; every consumer normalises + LeakyReLUs (and pools / concatenates) on load.
// Workspace: one raw buffer per skip level plus three rotating scratch buffers.
#include "common.h"

using namespace cine;

namespace {

constexpr float kEps = 1e-5f;     // nn.InstanceNorm2d default (unet.py:161)
constexpr float kSlope = 0.2f;    // nn.LeakyReLU(0.2)        (unet.py:162)

struct Bump {
    char* base; size_t off, cap;
    float* take(size_t floats) {
        const size_t bytes = (floats * sizeof(float) + 255) & ~size_t(255);
        float* p = base ? reinterpret_cast<float*>(base + off) : nullptr;
        off += bytes;
        return p;
    }
};

struct Plan {
    int P;
    int hs[8], wsz[8], ch[8];          // per level 0..P (P = bottleneck)
    float *skip[8], *sskip[8];
    float *scr[3], *sscr[3];
};

void build(Plan& p, Bump& b, int n, int h, int w, int chans, int pools) {
    p.P = pools;
    for (int d = 0; d <= pools; ++d) {
        p.hs[d] = d ? p.hs[d - 1] / 2 : h;
        p.wsz[d] = d ? p.wsz[d - 1] / 2 : w;
        p.ch[d] = chans << d;
    }
    for (int d = 0; d < pools; ++d) {
        p.skip[d] = b.take((size_t)n * p.ch[d] * p.hs[d] * p.wsz[d]);
        p.sskip[d] = b.take((size_t)n * p.ch[d] * 2);
    }
    size_t big = 0, bigc = 0;
    for (int d = 0; d <= pools; ++d) {
        const size_t e = (size_t)p.ch[d] * p.hs[d] * p.wsz[d];
        if (e > big) big = e;
        if ((size_t)p.ch[d] > bigc) bigc = p.ch[d];
    }
    for (int i = 0; i < 3; ++i) {
        p.scr[i] = b.take((size_t)n * big);
        p.sscr[i] = b.take((size_t)n * bigc * 2);
    }
}

}  // namespace

extern "C" size_t cine_unet2d_ws_bytes(int n, int h, int w, int in_ch, int out_ch, int chans, int pools) {
    if (n <= 0 || h <= 0 || w <= 0 || chans <= 0 || pools <= 0 || pools > 6) return 0;
    (void)in_ch; (void)out_ch;
    Plan p; Bump b{nullptr, 0, 0};
    build(p, b, n, h, w, chans, pools);
    return b.off;
}

static int unet_one(const float* x, float* y, const void* const* wts, int n, int h, int w, int in_ch, int out_ch,
                    int chans, int pools, void* ws, void* stream) {
    Plan p; Bump b{reinterpret_cast<char*>(ws), 0, 0};
    build(p, b, n, h, w, chans, pools);
    int wi = 0;
    auto W = [&](void) { return reinterpret_cast<const float*>(wts[wi++]); };
    int e;
    // ---- down path (unet.py:94-97) + bottleneck (:99)
    for (int d = 0; d <= pools; ++d) {
        const float* w1 = W(); const float* w2 = W();
        const bool last = d == pools;
        float* mid = p.scr[0]; float* smid = p.sscr[0];
        float* out = last ? p.scr[1] : p.skip[d];
        float* sout = last ? p.sscr[1] : p.sskip[d];
        if (d == 0)
            e = cine_conv3x3_in(x, nullptr, in_ch, 0, h, w, nullptr, nullptr, 0, 0, 0, 0, w1, mid, smid,
                                n, p.ch[0], h, w, kEps, kSlope, stream);
        else
            e = cine_conv3x3_in(p.skip[d - 1], p.sskip[d - 1], p.ch[d - 1], 2, p.hs[d - 1], p.wsz[d - 1],
                                nullptr, nullptr, 0, 0, 0, 0, w1, mid, smid,
                                n, p.ch[d], p.hs[d], p.wsz[d], kEps, kSlope, stream);
        if (e) return e;
        e = cine_conv3x3_in(mid, smid, p.ch[d], 1, p.hs[d], p.wsz[d], nullptr, nullptr, 0, 0, 0, 0, w2, out, sout,
                            n, p.ch[d], p.hs[d], p.wsz[d], kEps, kSlope, stream);
        if (e) return e;
    }
    // ---- up path (unet.py:102-123)
    int cur = 1;
    for (int u = 0; u < pools; ++u) {
        const int d = pools - 1 - u;
        const float* wt = W(); const float* w1 = W(); const float* w2 = W();
        const int a = (cur + 1) % 3, c = (cur + 2) % 3;
        // transpose conv: level d+1 -> (2 h_{d+1}, 2 w_{d+1}), ch_d channels
        e = cine_tconv2x2_in(p.scr[cur], p.sscr[cur], 1, wt, p.scr[a], p.sscr[a], n, p.ch[d + 1], p.ch[d],
                             p.hs[d + 1], p.wsz[d + 1], kEps, kSlope, stream);
        if (e) return e;
        // cat([up, skip]) -> conv1; `up` reads as zero beyond its extent (zero pad, :106-120)
        e = cine_conv3x3_in(p.scr[a], p.sscr[a], p.ch[d], 1, 2 * p.hs[d + 1], 2 * p.wsz[d + 1],
                            p.skip[d], p.sskip[d], p.ch[d], 1, p.hs[d], p.wsz[d], w1, p.scr[c], p.sscr[c],
                            n, p.ch[d], p.hs[d], p.wsz[d], kEps, kSlope, stream);
        if (e) return e;
        e = cine_conv3x3_in(p.scr[c], p.sscr[c], p.ch[d], 1, p.hs[d], p.wsz[d], nullptr, nullptr, 0, 0, 0, 0, w2,
                            p.scr[a], p.sscr[a], n, p.ch[d], p.hs[d], p.wsz[d], kEps, kSlope, stream);
        if (e) return e;
        cur = a;
    }
    const float* wf = W(); const float* bf = W();
    return cine_conv1x1_bias(p.scr[cur], p.sscr[cur], 1, wf, bf, y, n, chans, out_ch, h, w, kSlope, stream);   // :69
}

extern "C" int cine_unet2d_forward(const float* x, float* y, const void* const* weights, int nsets,
                                   int n, int h, int w, int in_ch, int out_ch, int chans, int pools,
                                   void* ws, size_t ws_bytes, void* stream) {
    CINE_REQUIRE(x && y && weights && ws, CINE_EINVAL, "cine_unet2d_forward: null pointer");
    CINE_REQUIRE(nsets >= 1 && n > 0 && n % nsets == 0, CINE_EINVAL, "cine_unet2d_forward: n=%d not divisible by nsets=%d", n, nsets);
    CINE_REQUIRE(h > 0 && w > 0 && in_ch > 0 && out_ch > 0 && chans > 0 && pools > 0 && pools <= 6, CINE_EINVAL,
                 "cine_unet2d_forward: bad sizes");
    CINE_REQUIRE((h >> pools) >= 1 && (w >> pools) >= 1, CINE_EUNSUPPORTED,
                 "cine_unet2d_forward: %dx%d too small for %d pools", h, w, pools);
    const int per = n / nsets;
    const size_t need = cine_unet2d_ws_bytes(per, h, w, in_ch, out_ch, chans, pools);
    CINE_REQUIRE(ws_bytes >= need, CINE_EWORKSPACE, "cine_unet2d_forward: workspace %zu < %zu", ws_bytes, need);
    const int nptr = 5 * pools + 4;
    for (int s = 0; s < nsets; ++s) {
        for (int i = 0; i < nptr; ++i)
            CINE_REQUIRE(weights[s * nptr + i], CINE_EINVAL, "cine_unet2d_forward: weights[%d] is null", s * nptr + i);
        if (int e = unet_one(x + (size_t)s * per * in_ch * h * w, y + (size_t)s * per * out_ch * h * w,
                             weights + s * nptr, per, h, w, in_ch, out_ch, chans, pools, ws, stream))
            return e;
    }
    return CINE_OK;
}
