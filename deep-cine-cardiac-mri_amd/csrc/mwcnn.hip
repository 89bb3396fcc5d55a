// mwcnn.hip -- launch sequence of one multi-level wavelet CNN pass (reference denoisers/mwcnn.py:135-179).
//
// Same storage discipline as unet.hip: every feature map is kept RAW with its InstanceNorm partial
// statistics; normalise + LeakyReLU, the Haar DWT that replaces pooling (mwcnn.py:216-236), the Haar IWT
// that replaces up-sampling (:240-263) and the additive skips (:164,172) are applied while the next conv
// stages its operands.  Only the default topology of the reference is handled on this path
// (n_first_convs = 1, res = False -- what XPDNet builds, xpdnet.py:251-262); anything else is rejected.
#include "common.h"

using namespace cine;

extern "C" int cine_conv_stat_partials(int cout, int h, int w, int is_tconv);
extern "C" int cine_conv3x3_ex2(const float* x0, const float* part0, int np0, int c0, int mode0, int h0, int w0,
                                const float* x1, const float* part1, int np1, int c1, int mode1, int h1, int w1, int add_src1,
                                const float* wpacked, const float* bias, const float* wpacked2, const float* bias2, int set_split,
                                const float* addend, int relu,
                                float* y, float* part_y, int n, int cout, int h, int w, float eps, float slope, void* stream);

namespace {
constexpr float kEps = 1e-5f;        // (the LeakyReLU slope of the conv blocks, mwcnn.py:204 = 0.2, is an argument of every entry point)
constexpr int kMaxScales = 6, kMaxConvs = 8;
enum { M_PLAIN = 0, M_ACT = 1, M_DWT_ACT = 3 | 8, M_IWT_ACT = 4 | 8 };

struct Bump {
    char* base; size_t off;
    float* take(size_t floats) {
        const size_t bytes = (floats * sizeof(float) + 255) & ~size_t(255);
        float* p = base ? reinterpret_cast<float*>(base + off) : nullptr;
        off += bytes;
        return p;
    }
};

struct Feat { float* x; float* part; int c, h, w, np; };

struct Plan {
    int S, first, out_ch, in_ch;
    int nf[kMaxScales], nc[kMaxScales];
    Feat first_feat, skip[kMaxScales], scr[3];
};

// channel plan of mwcnn.py:110-132
void chans(const Plan& p, int s, int i, int& ci, int& co) {
    ci = co = p.nf[s];
    if (i == 0) ci = 4 * (s == 0 ? p.first : p.nf[s - 1]);
    if (i == 2 * p.nc[s] - 1) co = s == 0 ? (4 * p.first > 4 * p.out_ch ? 4 * p.first : 4 * p.out_ch) : 4 * p.nf[s - 1];
}

Feat alloc(Bump& b, int n, int c, int h, int w) {
    Feat f; f.c = c; f.h = h; f.w = w; f.np = cine_conv_stat_partials(c, h, w, 0);
    f.x = b.take((size_t)n * c * h * w); f.part = b.take((size_t)n * c * f.np * 3);
    return f;
}

void build(Plan& p, Bump& b, int n, int h, int w) {
    p.first_feat = alloc(b, n, p.first, h, w);
    size_t big = 0, bigc = 0; int bigh = 0, bigw = 0;
    for (int s = 0; s < p.S; ++s) {
        const int hs = h >> (s + 1), ws = w >> (s + 1);
        p.skip[s] = alloc(b, n, p.nf[s], hs, ws);
        for (int i = 0; i < 2 * p.nc[s]; ++i) {
            int ci, co; chans(p, s, i, ci, co);
            const size_t e = (size_t)co * hs * ws;
            if (e > big) { big = e; }
            if ((size_t)co > bigc) bigc = co;
            if (hs > bigh) { bigh = hs; bigw = ws; }
        }
    }
    const int npmax = cine_conv_stat_partials(16, bigh, bigw, 0);      // most tiles per plane occur at the finest scale
    for (int i = 0; i < 3; ++i) {
        p.scr[i].x = b.take((size_t)n * big);
        p.scr[i].part = b.take((size_t)n * bigc * (npmax > 0 ? npmax : 1) * 3 * 4);
    }
}

int check_topology(int n_scales, const int* nf, const int* nc, int n_first_convs, int res) {
    CINE_REQUIRE(n_scales >= 1 && n_scales <= kMaxScales, CINE_EUNSUPPORTED, "cine_mwcnn: n_scales %d", n_scales);
    CINE_REQUIRE(n_first_convs == 1 && !res, CINE_EUNSUPPORTED,
                 "cine_mwcnn: only n_first_convs = 1, res = False (the XPDNet topology) is on the HIP path");
    for (int s = 0; s < n_scales; ++s)
        CINE_REQUIRE(nf[s] > 0 && nc[s] >= 1 && 2 * nc[s] <= kMaxConvs, CINE_EUNSUPPORTED, "cine_mwcnn: scale %d plan", s);
    return CINE_OK;
}
}  // namespace

extern "C" size_t cine_mwcnn_ws_bytes(int n, int h, int w, int in_ch, int out_ch, int n_scales, const int* n_filters,
                                      const int* n_convs, int first_filters) {
    if (n <= 0 || h <= 0 || w <= 0 || n_scales < 1 || n_scales > kMaxScales || !n_filters || !n_convs) return 0;
    Plan p{}; p.S = n_scales; p.first = first_filters; p.out_ch = out_ch; p.in_ch = in_ch;
    for (int s = 0; s < n_scales; ++s) { p.nf[s] = n_filters[s]; p.nc[s] = n_convs[s]; }
    Bump b{nullptr, 0};
    build(p, b, n, h, w);
    return b.off;
}

// weights (host array of device pointers): first_convs[0] (packed 3x3), then for every scale s and conv i
// conv_blocks_per_scale[s][i] (packed 3x3), then first_convs[1] weight (packed 3x3) and its bias.
static int mwcnn_impl(const float* x, float* y, const void* const* weights, const void* const* weights2, int set_split, int n, int h, int w,
                      int in_ch, int out_ch, int n_scales, const int* n_filters, const int* n_convs,
                      int n_first_convs, int first_filters, int res, float kSlope, void* ws, size_t ws_bytes, void* stream);

extern "C" int cine_mwcnn_forward(const float* x, float* y, const void* const* weights, int n, int h, int w,
                                  int in_ch, int out_ch, int n_scales, const int* n_filters, const int* n_convs,
                                  int n_first_convs, int first_filters, int res, float slope, void* ws, size_t ws_bytes, void* stream) {
    return mwcnn_impl(x, y, weights, nullptr, n, n, h, w, in_ch, out_ch, n_scales, n_filters, n_convs, n_first_convs, first_filters, res,
                      slope, ws, ws_bytes, stream);
}

// two MWCNNs of the same topology in one launch sequence: samples [0, set_split) go through `weights`, the rest through `weights2`
// (XPDNet's x-t and y-t networks, xpdnet.py:424-446, on planes of equal shape)
extern "C" int cine_mwcnn_forward2(const float* x, float* y, const void* const* weights, const void* const* weights2, int set_split,
                                   int n, int h, int w, int in_ch, int out_ch, int n_scales, const int* n_filters, const int* n_convs,
                                   int n_first_convs, int first_filters, int res, float slope, void* ws, size_t ws_bytes, void* stream) {
    CINE_REQUIRE(weights2 && set_split > 0 && set_split < n, CINE_EINVAL, "cine_mwcnn_forward2: needs a second weight set and 0 < set_split < n");
    return mwcnn_impl(x, y, weights, weights2, set_split, n, h, w, in_ch, out_ch, n_scales, n_filters, n_convs, n_first_convs, first_filters, res,
                      slope, ws, ws_bytes, stream);
}

static int mwcnn_impl(const float* x, float* y, const void* const* weights, const void* const* weights2, int set_split, int n, int h, int w,
                      int in_ch, int out_ch, int n_scales, const int* n_filters, const int* n_convs,
                      int n_first_convs, int first_filters, int res, float kSlope, void* ws, size_t ws_bytes, void* stream) {
    CINE_REQUIRE(x && y && weights && ws && n_filters && n_convs, CINE_EINVAL, "cine_mwcnn_forward: null pointer");
    CINE_REQUIRE(kSlope >= 0.f && kSlope <= 1.f, CINE_EINVAL, "cine_mwcnn_forward: LeakyReLU slope %g outside [0, 1]", (double)kSlope);
    CINE_REQUIRE(n > 0 && h > 0 && w > 0 && in_ch > 0 && out_ch > 0 && first_filters > 0, CINE_EINVAL, "cine_mwcnn_forward: bad sizes");
    if (int e = check_topology(n_scales, n_filters, n_convs, n_first_convs, res)) return e;
    CINE_REQUIRE(h % (1 << n_scales) == 0 && w % (1 << n_scales) == 0, CINE_EINVAL,
                 "cine_mwcnn_forward: %dx%d is not a multiple of 2^%d (pad_for_mwcnn first)", h, w, n_scales);
    const size_t need = cine_mwcnn_ws_bytes(n, h, w, in_ch, out_ch, n_scales, n_filters, n_convs, first_filters);
    CINE_REQUIRE(ws_bytes >= need, CINE_EWORKSPACE, "cine_mwcnn_forward: workspace %zu < %zu", ws_bytes, need);
    Plan p{}; p.S = n_scales; p.first = first_filters; p.out_ch = out_ch; p.in_ch = in_ch;
    for (int s = 0; s < n_scales; ++s) { p.nf[s] = n_filters[s]; p.nc[s] = n_convs[s]; }
    Bump b{reinterpret_cast<char*>(ws), 0};
    build(p, b, n, h, w);
    // pointer index of conv_blocks_per_scale[s][i] in module order (after first_convs[0])
    int woff[kMaxScales + 1]; woff[0] = 1;
    for (int s = 0; s < p.S; ++s) woff[s + 1] = woff[s] + 2 * p.nc[s];
    auto WB = [&](int s, int i) { return reinterpret_cast<const float*>(weights[woff[s] + i]); };
    for (int i = 0; i < woff[p.S] + 2; ++i) CINE_REQUIRE(weights[i] && (!weights2 || weights2[i]), CINE_EINVAL, "cine_mwcnn_forward: weights[%d] is null", i);
    // the second set's pointer for the same slot (weights and weights2 are parallel arrays)
    auto second = [&](const float* first) -> const float* {
        if (!weights2 || !first) return nullptr;
        for (int i = 0; i < woff[p.S] + 2; ++i) if (weights[i] == first) return reinterpret_cast<const float*>(weights2[i]);
        return nullptr;
    };
    auto conv = [&](const Feat& s0, int mode0, const Feat* s1, int mode1, int add, const float* wp, const float* bias,
                    float* yo, float* po, int cout, int ho, int wo) {
        return cine_conv3x3_ex2(s0.x, s0.part, s0.np, s0.c, mode0, s0.h, s0.w,
                                s1 ? s1->x : nullptr, s1 ? s1->part : nullptr, s1 ? s1->np : 0, s1 ? s1->c : 0, mode1,
                                s1 ? s1->h : 0, s1 ? s1->w : 0, add, wp, bias, second(wp), second(bias), set_split,
                                nullptr, 0, yo, po, n, cout, ho, wo, kEps, kSlope, stream);
    };
    int e;
    // first conv block (mwcnn.py:143-146): in_ch -> first filters at full resolution
    Feat in{const_cast<float*>(x), nullptr, in_ch, h, w, 0};
    if ((e = conv(in, M_PLAIN, nullptr, 0, 0, reinterpret_cast<const float*>(weights[0]), nullptr, p.first_feat.x,
                  p.first_feat.part, p.first, h, w))) return e;
    // ---- analysis path (:148-154): DWT on load, n_convs blocks per scale
    Feat cur = p.first_feat;
    int scr_i = 0;
    for (int s = 0; s < p.S; ++s) {
        const int hs = h >> (s + 1), wsz = w >> (s + 1);
        for (int i = 0; i < p.nc[s]; ++i) {
            int ci, co; chans(p, s, i, ci, co);
            const bool last = i == p.nc[s] - 1;
            Feat out = last ? p.skip[s] : p.scr[scr_i];
            out.c = co; out.h = hs; out.w = wsz; out.np = cine_conv_stat_partials(co, hs, wsz, 0);
            if ((e = conv(cur, i == 0 ? M_DWT_ACT : M_ACT, nullptr, 0, 0, WB(s, i), nullptr, out.x, out.part, co, hs, wsz))) return e;
            if (!last) scr_i = (scr_i + 1) % 3;
            cur = out;
        }
    }
    // ---- synthesis path (:156-168)
    for (int s = p.S - 1; s >= 0; --s) {
        const int hs = h >> (s + 1), wsz = w >> (s + 1);
        for (int i = p.nc[s]; i < 2 * p.nc[s]; ++i) {
            int ci, co; chans(p, s, i, ci, co);
            Feat out = p.scr[scr_i];
            out.c = co; out.h = hs; out.w = wsz; out.np = cine_conv_stat_partials(co, hs, wsz, 0);
            if (i == p.nc[s] && s != p.S - 1)   // IWT of the coarser scale + this scale's last analysis feature
                e = conv(cur, M_IWT_ACT, &p.skip[s], M_ACT, 1, WB(s, i), nullptr, out.x, out.part, co, hs, wsz);
            else
                e = conv(cur, M_ACT, nullptr, 0, 0, WB(s, i), nullptr, out.x, out.part, co, hs, wsz);
            if (e) return e;
            scr_i = (scr_i + 1) % 3;
            cur = out;
        }
    }
    // ---- final IWT + first feature, last conv with bias and no norm (:170-174, 77-83)
    const float* wl = reinterpret_cast<const float*>(weights[woff[p.S]]);
    const float* bl = reinterpret_cast<const float*>(weights[woff[p.S] + 1]);
    return conv(cur, M_IWT_ACT, &p.first_feat, M_ACT, 1, wl, bl, y, nullptr, out_ch, h, w);
}

// ---------------------------------------------------------------- training: forward that keeps every feature map, and the backward pass
// Same launch sequence as mwcnn_impl, but every conv output owns its memory (the backward pass reads all of them).  One weight set or
// two (samples >= set_split through the second network).
#include "grad.h"
extern "C" int cine_conv3x3_dgrad(const float* gy, const float* wpacked, const float* wpacked2, int set_split,
                                  float* gx, int n, int cout, int cin, int h, int w, void* stream);
namespace {
struct TrainPlan {
    Plan p;
    Feat first_feat, feat[kMaxScales][kMaxConvs];
};
void build_train(TrainPlan& t, Bump& b, int n, int h, int w) {
    const Plan& p = t.p;
    t.first_feat = alloc(b, n, p.first, h, w);
    for (int s = 0; s < p.S; ++s)
        for (int i = 0; i < 2 * p.nc[s]; ++i) {
            int ci, co; chans(p, s, i, ci, co);
            t.feat[s][i] = alloc(b, n, co, h >> (s + 1), w >> (s + 1));
        }
}
int plan_from(Plan& p, int in_ch, int out_ch, int n_scales, const int* nf, const int* nc, int first) {
    p = Plan{}; p.S = n_scales; p.first = first; p.out_ch = out_ch; p.in_ch = in_ch;
    for (int s = 0; s < n_scales; ++s) { p.nf[s] = nf[s]; p.nc[s] = nc[s]; }
    return 0;
}
int woffsets(const Plan& p, int* woff) { woff[0] = 1; for (int s = 0; s < p.S; ++s) woff[s + 1] = woff[s] + 2 * p.nc[s]; return woff[p.S] + 2; }
// what conv (s, i) reads: source 0 (+ mode), optional added source 1
struct ConvIn { Feat s0; int m0; Feat s1; int m1; int add; };
ConvIn conv_input(const TrainPlan& t, int s, int i) {
    const Plan& p = t.p;
    ConvIn c{}; c.add = 0;
    if (i == 0) { c.s0 = s == 0 ? t.first_feat : t.feat[s - 1][p.nc[s - 1] - 1]; c.m0 = M_DWT_ACT; return c; }
    if (i == p.nc[s] && s != p.S - 1) {        // IWT of the coarser scale's last feature + this scale's last analysis feature (:162-164)
        c.s0 = t.feat[s + 1][2 * p.nc[s + 1] - 1]; c.m0 = M_IWT_ACT; c.s1 = t.feat[s][p.nc[s] - 1]; c.m1 = M_ACT; c.add = 1; return c;
    }
    c.s0 = t.feat[s][i - 1]; c.m0 = M_ACT; return c;
}
}  // namespace

extern "C" size_t cine_mwcnn_train_ws_bytes(int n, int h, int w, int in_ch, int out_ch, int n_scales, const int* n_filters,
                                            const int* n_convs, int first_filters) {
    if (n <= 0 || h <= 0 || w <= 0 || n_scales < 1 || n_scales > kMaxScales || !n_filters || !n_convs) return 0;
    TrainPlan t{}; plan_from(t.p, in_ch, out_ch, n_scales, n_filters, n_convs, first_filters);
    Bump b{nullptr, 0};
    build_train(t, b, n, h, w);
    return b.off;
}

extern "C" int cine_mwcnn_forward_train(const float* x, float* y, const void* const* weights, const void* const* weights2, int set_split,
                                        int n, int h, int w, int in_ch, int out_ch, int n_scales, const int* n_filters, const int* n_convs,
                                        int n_first_convs, int first_filters, int res, float kSlope, void* ws, size_t ws_bytes, void* stream) {
    CINE_REQUIRE(x && y && weights && ws && n_filters && n_convs, CINE_EINVAL, "cine_mwcnn_forward_train: null pointer");
    CINE_REQUIRE(kSlope >= 0.f && kSlope <= 1.f, CINE_EINVAL, "cine_mwcnn_forward_train: LeakyReLU slope %g outside [0, 1]", (double)kSlope);
    CINE_REQUIRE(n > 0 && h > 0 && w > 0 && in_ch > 0 && out_ch > 0 && first_filters > 0, CINE_EINVAL, "cine_mwcnn_forward_train: bad sizes");
    if (int e = check_topology(n_scales, n_filters, n_convs, n_first_convs, res)) return e;
    CINE_REQUIRE(h % (1 << n_scales) == 0 && w % (1 << n_scales) == 0, CINE_EINVAL, "cine_mwcnn_forward_train: %dx%d is not a multiple of 2^%d", h, w, n_scales);
    CINE_REQUIRE(ws_bytes >= cine_mwcnn_train_ws_bytes(n, h, w, in_ch, out_ch, n_scales, n_filters, n_convs, first_filters), CINE_EWORKSPACE,
                 "cine_mwcnn_forward_train: workspace too small");
    CINE_REQUIRE(!weights2 || (set_split > 0 && set_split < n), CINE_EINVAL, "cine_mwcnn_forward_train: set_split");
    TrainPlan t{}; plan_from(t.p, in_ch, out_ch, n_scales, n_filters, n_convs, first_filters);
    const Plan& p = t.p;
    Bump b{reinterpret_cast<char*>(ws), 0};
    build_train(t, b, n, h, w);
    int woff[kMaxScales + 1];
    const int nptr = woffsets(p, woff);
    for (int i = 0; i < nptr; ++i) CINE_REQUIRE(weights[i] && (!weights2 || weights2[i]), CINE_EINVAL, "cine_mwcnn_forward_train: weights[%d] is null", i);
    auto W = [&](int idx, int set) { return reinterpret_cast<const float*>((set && weights2 ? weights2 : weights)[idx]); };
    const int sp = weights2 ? set_split : n;
    auto conv = [&](const Feat& s0, int mode0, const Feat* s1, int mode1, int add, int widx, bool bias, float* yo, float* po, int cout, int ho, int wo) {
        return cine_conv3x3_ex2(s0.x, s0.part, s0.np, s0.c, mode0, s0.h, s0.w,
                                s1 ? s1->x : nullptr, s1 ? s1->part : nullptr, s1 ? s1->np : 0, s1 ? s1->c : 0, mode1, s1 ? s1->h : 0, s1 ? s1->w : 0, add,
                                W(widx, 0), bias ? W(widx + 1, 0) : nullptr, weights2 ? W(widx, 1) : nullptr, (bias && weights2) ? W(widx + 1, 1) : nullptr, sp,
                                nullptr, 0, yo, po, n, cout, ho, wo, kEps, kSlope, stream);
    };
    int e;
    Feat in{const_cast<float*>(x), nullptr, in_ch, h, w, 0};
    if ((e = conv(in, M_PLAIN, nullptr, 0, 0, 0, false, t.first_feat.x, t.first_feat.part, p.first, h, w))) return e;
    for (int s = 0; s < p.S; ++s)                                     // analysis
        for (int i = 0; i < p.nc[s]; ++i) {
            const ConvIn c = conv_input(t, s, i);
            const Feat& o = t.feat[s][i];
            if ((e = conv(c.s0, c.m0, nullptr, 0, 0, woff[s] + i, false, o.x, o.part, o.c, o.h, o.w))) return e;
        }
    for (int s = p.S - 1; s >= 0; --s)                                // synthesis
        for (int i = p.nc[s]; i < 2 * p.nc[s]; ++i) {
            const ConvIn c = conv_input(t, s, i);
            const Feat& o = t.feat[s][i];
            if ((e = conv(c.s0, c.m0, c.add ? &c.s1 : nullptr, c.m1, c.add, woff[s] + i, false, o.x, o.part, o.c, o.h, o.w))) return e;
        }
    const Feat& last = t.feat[0][2 * p.nc[0] - 1];
    return conv(last, M_IWT_ACT, &t.first_feat, M_ACT, 1, woff[p.S], true, y, nullptr, out_ch, h, w);
}

// the largest conv input (n, cin, h_s, w_s) over the layers: scratch for launch_wgrad's materialised sources
static size_t mwcnn_mat_floats(const Plan& p, int n, int h, int w, int in_ch) {
    size_t m = (size_t)n * std::max(p.first, in_ch) * h * w;
    for (int s = 0; s < p.S; ++s)
        for (int i = 0; i < 2 * p.nc[s]; ++i) {
            int ci, co; chans(p, s, i, ci, co);
            m = std::max(m, (size_t)n * ci * (h >> (s + 1)) * (w >> (s + 1)));
        }
    return m;
}

extern "C" size_t cine_mwcnn_backward_ws_bytes(int n, int h, int w, int in_ch, int out_ch, int n_scales, const int* n_filters,
                                               const int* n_convs, int first_filters) {
    if (n <= 0 || h <= 0 || w <= 0 || n_scales < 1 || n_scales > kMaxScales || !n_filters || !n_convs) return 0;
    Plan p; plan_from(p, in_ch, out_ch, n_scales, n_filters, n_convs, first_filters);
    Bump b{nullptr, 0};
    size_t big = (size_t)n * p.first * h * w, wg = wgrad_ws_floats(p.first, in_ch, 9, n);
    b.take((size_t)n * p.first * h * w);                              // input gradient of the final conv
    wg = std::max(wg, std::max(wgrad_ws_floats(out_ch, p.first, 9, n), (size_t)n * out_ch));
    for (int s = 0; s < p.S; ++s)
        for (int i = 0; i < 2 * p.nc[s]; ++i) {
            int ci, co; chans(p, s, i, ci, co);
            const size_t hw = (size_t)(h >> (s + 1)) * (w >> (s + 1));
            b.take((size_t)n * ci * hw);                              // input gradient of conv (s, i)
            big = std::max(big, (size_t)n * co * hw);
            wg = std::max(wg, wgrad_ws_floats(co, ci, 9, n));
        }
    b.take(big); b.take(big);                                         // d/d(raw) of the tensor in hand, two alternating buffers (SideLane)
    b.take(wg);
    b.take(mwcnn_mat_floats(p, n, h, w, in_ch));                      // materialised conv inputs of the weight gradients
    return b.off;
}

// Gradients of cine_mwcnn_forward_train.  wdgrad / grads as for cine_unet2d_backward, in the order of `weights` (first conv, the conv blocks in
// module order, the final conv's weight and bias); weights2-style second lists for the second network when set_split < n.
extern "C" int cine_mwcnn_backward(const float* x, const float* gy, const void* const* wdgrad, const void* const* wdgrad2, void* const* grads,
                                   void* const* grads2, int set_split, int n, int h, int w, int in_ch, int out_ch, int n_scales,
                                   const int* n_filters, const int* n_convs, int first_filters, float kSlope, const void* fwd_ws, size_t fwd_ws_bytes,
                                   void* ws, size_t ws_bytes, float* gx, void* stream) {
    CINE_REQUIRE(x && gy && wdgrad && grads && fwd_ws && ws && n_filters && n_convs, CINE_EINVAL, "cine_mwcnn_backward: null pointer");
    CINE_REQUIRE(kSlope >= 0.f && kSlope <= 1.f, CINE_EINVAL, "cine_mwcnn_backward: LeakyReLU slope %g outside [0, 1]", (double)kSlope);
    CINE_REQUIRE(n > 0 && n <= 65535 && h > 0 && w > 0, CINE_EINVAL, "cine_mwcnn_backward: bad sizes");
    if (int e = check_topology(n_scales, n_filters, n_convs, 1, 0)) return e;
    const bool two = wdgrad2 != nullptr;
    CINE_REQUIRE(!two || (grads2 && set_split > 0 && set_split < n), CINE_EINVAL, "cine_mwcnn_backward: second weight set");
    CINE_REQUIRE(fwd_ws_bytes >= cine_mwcnn_train_ws_bytes(n, h, w, in_ch, out_ch, n_scales, n_filters, n_convs, first_filters) &&
                 ws_bytes >= cine_mwcnn_backward_ws_bytes(n, h, w, in_ch, out_ch, n_scales, n_filters, n_convs, first_filters), CINE_EWORKSPACE,
                 "cine_mwcnn_backward: workspace too small");
    TrainPlan t{}; plan_from(t.p, in_ch, out_ch, n_scales, n_filters, n_convs, first_filters);
    const Plan& p = t.p;
    Bump bf{const_cast<char*>(reinterpret_cast<const char*>(fwd_ws)), 0};
    build_train(t, bf, n, h, w);
    int woff[kMaxScales + 1];
    const int nptr = woffsets(p, woff);
    for (int i = 0; i < nptr; ++i) {
        CINE_REQUIRE(grads[i] && (!two || grads2[i]), CINE_EINVAL, "cine_mwcnn_backward: grads[%d] is null", i);
        CINE_REQUIRE(i == nptr - 1 || (wdgrad[i] && (!two || wdgrad2[i])), CINE_EINVAL, "cine_mwcnn_backward: wdgrad[%d] is null", i);
    }
    // scratch (same order as cine_mwcnn_backward_ws_bytes)
    Bump bb{reinterpret_cast<char*>(ws), 0};
    float* g_final = bb.take((size_t)n * p.first * h * w);
    float* gin[kMaxScales][kMaxConvs];
    size_t big = (size_t)n * p.first * h * w, wgf = wgrad_ws_floats(p.first, in_ch, 9, n);
    wgf = std::max(wgf, std::max(wgrad_ws_floats(out_ch, p.first, 9, n), (size_t)n * out_ch));
    for (int s = 0; s < p.S; ++s)
        for (int i = 0; i < 2 * p.nc[s]; ++i) {
            int ci, co; chans(p, s, i, ci, co);
            const size_t hw = (size_t)(h >> (s + 1)) * (w >> (s + 1));
            gin[s][i] = bb.take((size_t)n * ci * hw);
            big = std::max(big, (size_t)n * co * hw);
            wgf = std::max(wgf, wgrad_ws_floats(co, ci, 9, n));
        }
    float* grb[2] = {bb.take(big), nullptr};
    grb[1] = bb.take(big);
    float* wgs = bb.take(wgf);
    const size_t matf = mwcnn_mat_floats(p, n, h, w, in_ch);
    float* mat = bb.take(matf);
    hipStream_t st = as_stream(stream);
    SideLane lane(st);              // weight gradients on the side stream (grad.h)
    float* gr = nullptr;
    auto next_g = [&]() { lane.before_write(); gr = grb[lane.slot()]; };
    const int sp = two ? set_split : n;
    auto WD = [&](int idx, int set) { return reinterpret_cast<const float*>((set ? wdgrad2 : wdgrad)[idx]); };
    auto GR = [&](int idx, int set) { return set ? (two ? reinterpret_cast<float*>(grads2[idx]) : nullptr) : reinterpret_cast<float*>(grads[idx]); };
    auto src_of = [&](const Feat& f, int mode) { return Src{f.x, f.part, f.c, mode & 7, f.h, f.w, f.np, (mode >> 3) & 1, 1}; };
    const Src none{nullptr, nullptr, 0, 0, 0, 0, 0, 0, 1};
    auto wgrad = [&](const ConvIn& c, const float* g, int rows, int cin, int hh, int ww, int widx) {
        WgArgs a{}; a.s0 = src_of(c.s0, c.m0); a.s1 = c.add ? src_of(c.s1, c.m1) : none; a.add_src1 = c.add; a.cin = cin;
        a.g = g; a.g_mode = 0; a.rows = rows; a.n = n; a.H = hh; a.W = ww; a.set_split = sp; a.eps = kEps; a.slope = kSlope;
        a.mat = mat; a.mat_floats = matf;
        hipStream_t sw = lane.fork();
        const int err = launch_wgrad(a, 9, 0, GR(widx, 0), GR(widx, 1), wgs, wgf, sw);
        lane.launched();
        return err;
    };
    auto inbwd = [&](const Feat& f, GradPiece pa, GradPiece pb) {
        next_g();
        InBwdArgs a{f.x, f.part, f.np, pa, pb, gr, n, f.c, f.h, f.w, kEps, kSlope};
        return launch_in_lrelu_bwd(a, st);
    };
    const GradPiece nopiece{nullptr, 0, 0, 0, 0, 0};
    int e;
    // ---- final conv: y = conv(IWT(act(last)) + act(first_feat)) + b  (:170-174)
    const Feat& last = t.feat[0][2 * p.nc[0] - 1];
    if ((e = launch_bias_grad(gy, n, out_ch, (long)h * w, sp, GR(woff[p.S] + 1, 0), GR(woff[p.S] + 1, 1), wgs, wgf, st))) return e;
    { ConvIn c{last, M_IWT_ACT, t.first_feat, M_ACT, 1};
      if ((e = wgrad(c, gy, out_ch, p.first, h, w, woff[p.S]))) return e; }
    if ((e = cine_conv3x3_dgrad(gy, WD(woff[p.S], 0), two ? WD(woff[p.S], 1) : nullptr, sp, g_final, n, out_ch, p.first, h, w, stream))) return e;
    // the gradient piece the consumer(s) of feature (s, i) hand back
    auto consumer_piece = [&](int s, int i) -> GradPiece {
        const int hs = h >> (s + 1), ws2 = w >> (s + 1);
        if (i == 2 * p.nc[s] - 1) {                                   // read through an IWT by scale s - 1's first synthesis conv, or by the final conv
            if (s == 0) return GradPiece{g_final, 4, p.first, 0, h, w};
            int ci, co; chans(p, s - 1, p.nc[s - 1], ci, co);
            return GradPiece{gin[s - 1][p.nc[s - 1]], 4, ci, 0, 2 * hs, 2 * ws2};
        }
        int ci, co; chans(p, s, i + 1, ci, co);                       // the next conv of the same scale reads it plainly
        return GradPiece{gin[s][i + 1], 1, ci, 0, hs, ws2};
    };
    // ---- synthesis path, reversed: scale 0 first
    for (int s = 0; s < p.S; ++s)
        for (int i = 2 * p.nc[s] - 1; i >= p.nc[s]; --i) {
            const Feat& f = t.feat[s][i];
            int ci, co; chans(p, s, i, ci, co);
            if ((e = inbwd(f, consumer_piece(s, i), nopiece))) return e;
            const ConvIn c = conv_input(t, s, i);
            if ((e = wgrad(c, gr, co, ci, f.h, f.w, woff[s] + i))) return e;
            if ((e = cine_conv3x3_dgrad(gr, WD(woff[s] + i, 0), two ? WD(woff[s] + i, 1) : nullptr, sp, gin[s][i], n, co, ci, f.h, f.w, stream))) return e;
        }
    // ---- analysis path, reversed: the coarsest scale first
    for (int s = p.S - 1; s >= 0; --s)
        for (int i = p.nc[s] - 1; i >= 0; --i) {
            const Feat& f = t.feat[s][i];
            int ci, co; chans(p, s, i, ci, co);
            GradPiece pa, pb = nopiece;
            if (i == p.nc[s] - 1 && s != p.S - 1) {                   // the scale's last analysis feature: DWT of the next scale + the synthesis skip
                int c2, o2; chans(p, s + 1, 0, c2, o2);
                pa = GradPiece{gin[s + 1][0], 3, c2, 0, f.h / 2, f.w / 2};
                int c3, o3; chans(p, s, p.nc[s], c3, o3);
                pb = GradPiece{gin[s][p.nc[s]], 1, c3, 0, f.h, f.w};
            } else pa = consumer_piece(s, i);
            if ((e = inbwd(f, pa, pb))) return e;
            const ConvIn c = conv_input(t, s, i);
            if ((e = wgrad(c, gr, co, ci, f.h, f.w, woff[s] + i))) return e;
            if ((e = cine_conv3x3_dgrad(gr, WD(woff[s] + i, 0), two ? WD(woff[s] + i, 1) : nullptr, sp, gin[s][i], n, co, ci, f.h, f.w, stream))) return e;
        }
    // ---- first conv block: consumers = scale 0's DWT conv and the final conv's skip
    {
        int c2, o2; chans(p, 0, 0, c2, o2);
        if ((e = inbwd(t.first_feat, GradPiece{gin[0][0], 3, c2, 0, h / 2, w / 2}, GradPiece{g_final, 1, p.first, 0, h, w}))) return e;
        ConvIn c{}; c.s0 = Feat{const_cast<float*>(x), nullptr, in_ch, h, w, 0}; c.m0 = M_PLAIN;
        if ((e = wgrad(c, gr, p.first, in_ch, h, w, 0))) return e;
        if (gx && (e = cine_conv3x3_dgrad(gr, WD(0, 0), two ? WD(0, 1) : nullptr, sp, gx, n, p.first, in_ch, h, w, stream))) return e;
    }
    lane.join();
    return CINE_OK;
}
