// unet.hip -- launch sequence of one 2-D U-Net pass (reference denoisers/unet.py:73-125).
//
// Activations stay raw (pre-InstanceNorm) in HBM together with per-tile partial statistics
// {count, mean, M2}; every consumer merges the partials, normalises + LeakyReLUs (and pools /
// concatenates) while staging its operands.  Two weight sets (the x-f and y-f U-Nets of one
// cascade, varnet.py:224-226) run in the SAME launches: samples [0, n/2) use set 0, the rest set 1.
// Workspace: one raw buffer per skip level plus three rotating scratch buffers.
#include <algorithm>
#include "common.h"

using namespace cine;

extern "C" int cine_conv_stat_partials(int cout, int h, int w, int is_tconv);

namespace cine {
// unet_bottom.hip: levels P-1 and P of the U as one kernel per plane
struct BottomArgs {
    const float* x1; const float* px1; int np1;
    const float* w[7][2];
    int set_split;
    float* skip2;
    float* y; float* py;
    float eps, slope;
};
bool unet_bottom_applies(int chans2, int h2, int w2);
int launch_unet_bottom(const BottomArgs& a, int n, hipStream_t st);
}  // namespace cine

namespace {

constexpr float kEps = 1e-5f;     // nn.InstanceNorm2d default (unet.py:161)
constexpr float kSlope = 0.2f;    // nn.LeakyReLU(0.2)        (unet.py:162)

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
    int P;
    int hs[8], wsz[8], ch[8];          // per level 0..P (P = bottleneck)
    int np_conv[8], np_tconv[8];       // partial records per (sample, channel) of a conv / tconv output at level d
    float *skip[8], *pskip[8];         // ConvBlock outputs of the down path (the concat sources)
    float *mid[8], *pmid[8];           // first conv of the down-path ConvBlock at level d (and of the bottleneck at d = P)
    float *bott, *pbott;               // bottleneck output
    float *up[8], *pup[8];             // transpose-conv output at level d
    float *ca[8], *pca[8];             // up-path ConvBlock at level d: first conv ...
    float *cb[8], *pcb[8];             // ... and second conv
    float* prog;                       // device copy of the plane-persistent kernel's program (private layout only)
};

// private == false: three rotating scratch buffers sized for the largest layer (layers are separated by kernel boundaries, so
// a buffer may hold different shapes over time).  private == true: every layer output owns its memory, sample n of every
// tensor sits at n * (its dense size) -- what the plane-persistent kernel needs, where the samples are at different layers
// at the same time.
void build(Plan& p, Bump& b, int n, int h, int w, int chans, int pools, bool priv) {
    p.P = pools;
    for (int d = 0; d <= pools; ++d) {
        p.hs[d] = d ? p.hs[d - 1] / 2 : h;
        p.wsz[d] = d ? p.wsz[d - 1] / 2 : w;
        p.ch[d] = chans << d;
        p.np_conv[d] = cine_conv_stat_partials(p.ch[d], p.hs[d], p.wsz[d], 0);
    }
    for (int d = 0; d < pools; ++d)    // tconv from level d+1 producing ch[d] channels on 2*hs[d+1] x 2*wsz[d+1]
        p.np_tconv[d] = cine_conv_stat_partials(p.ch[d], p.hs[d + 1], p.wsz[d + 1], 1);
    auto elems = [&](int d) { return (size_t)n * p.ch[d] * p.hs[d] * p.wsz[d]; };
    auto pelems = [&](int d, bool tconv) { return (size_t)n * p.ch[d] * (tconv ? p.np_tconv[d] : p.np_conv[d]) * 3; };
    for (int d = 0; d < pools; ++d) {
        p.skip[d] = b.take(elems(d));
        p.pskip[d] = b.take(pelems(d, false));
    }
    p.prog = nullptr;
    if (priv) {
        p.prog = b.take((plane_program_bytes() + 3) / 4);
        for (int d = 0; d <= pools; ++d) { p.mid[d] = b.take(elems(d)); p.pmid[d] = b.take(pelems(d, false)); }
        p.bott = b.take(elems(pools)); p.pbott = b.take(pelems(pools, false));
        for (int d = 0; d < pools; ++d) {
            p.up[d] = b.take(elems(d)); p.pup[d] = b.take(pelems(d, true));
            p.ca[d] = b.take(elems(d)); p.pca[d] = b.take(pelems(d, false));
            p.cb[d] = b.take(elems(d)); p.pcb[d] = b.take(pelems(d, false));
        }
        return;
    }
    size_t big = 0, bigp = 0;
    for (int d = 0; d <= pools; ++d) {
        big = std::max(big, elems(d));
        bigp = std::max(bigp, std::max(pelems(d, false), d < pools ? pelems(d, true) : (size_t)0));
    }
    float *scr[3], *pscr[3];
    for (int i = 0; i < 3; ++i) { scr[i] = b.take(big); pscr[i] = b.take(bigp); }
    for (int d = 0; d <= pools; ++d) { p.mid[d] = scr[0]; p.pmid[d] = pscr[0]; }
    p.bott = scr[1]; p.pbott = pscr[1];
    int cur = 1;
    for (int u = 0; u < pools; ++u) {
        const int d = pools - 1 - u;
        const int a = (cur + 1) % 3, c = (cur + 2) % 3;
        p.up[d] = scr[a]; p.pup[d] = pscr[a];
        p.ca[d] = scr[c]; p.pca[d] = pscr[c];
        p.cb[d] = scr[a]; p.pcb[d] = pscr[a];
        cur = a;
    }
}

// samples from which the plane-persistent kernel is worth trying (conv_kernels.hip applies the same threshold)
constexpr int kPlaneMinSamples = 128;

}  // namespace

extern "C" size_t cine_unet2d_ws_bytes(int n, int h, int w, int in_ch, int out_ch, int chans, int pools) {
    if (n <= 0 || h <= 0 || w <= 0 || chans <= 0 || pools <= 0 || pools > 6) return 0;
    (void)in_ch; (void)out_ch;
    Plan p; Bump b{nullptr, 0};
    build(p, b, n, h, w, chans, pools, plane_kernel_enabled() && n >= kPlaneMinSamples);
    return b.off;
}

extern "C" int cine_unet2d_forward(const float* x, float* y, const void* const* weights, int nsets,
                                   int n, int h, int w, int in_ch, int out_ch, int chans, int pools,
                                   void* ws, size_t ws_bytes, void* stream) {
    CINE_REQUIRE(x && y && weights && ws, CINE_EINVAL, "cine_unet2d_forward: null pointer");
    CINE_REQUIRE(nsets == 1 || nsets == 2, CINE_EINVAL, "cine_unet2d_forward: nsets must be 1 or 2");
    CINE_REQUIRE(n > 0 && n % nsets == 0, CINE_EINVAL, "cine_unet2d_forward: n=%d not divisible by nsets=%d", n, nsets);
    CINE_REQUIRE(h > 0 && w > 0 && in_ch > 0 && out_ch > 0 && chans > 0 && pools > 0 && pools <= 6, CINE_EINVAL,
                 "cine_unet2d_forward: bad sizes");
    CINE_REQUIRE((h >> pools) >= 1 && (w >> pools) >= 1, CINE_EUNSUPPORTED,
                 "cine_unet2d_forward: %dx%d too small for %d pools", h, w, pools);
    const size_t need = cine_unet2d_ws_bytes(n, h, w, in_ch, out_ch, chans, pools);
    CINE_REQUIRE(ws_bytes >= need, CINE_EWORKSPACE, "cine_unet2d_forward: workspace %zu < %zu", ws_bytes, need);
    const int nptr = 5 * pools + 4;
    for (int i = 0; i < nsets * nptr; ++i)
        CINE_REQUIRE(weights[i], CINE_EINVAL, "cine_unet2d_forward: weights[%d] is null", i);

    Plan p; Bump b{reinterpret_cast<char*>(ws), 0};
    build(p, b, n, h, w, chans, pools, plane_kernel_enabled() && n >= kPlaneMinSamples);
    const int split = n / nsets;
    // every step below reads only its own sample's data, so the launches are recorded and issued together: as one
    // plane-persistent kernel when the layer shapes are the ones it is built for, else layer by layer (conv_kernels.hip)
    // (recording only for the opt-in plane-persistent kernel; otherwise every launch is issued where it stands)
    struct Guard {
        PlaneRecorder* r;
        ~Guard() { if (r) plane_record_abort(r); }
    } guard{plane_kernel_enabled() ? plane_record_begin() : nullptr};
    int wi = 0;
    const float *w0, *w1;
    auto next = [&]() {
        w0 = reinterpret_cast<const float*>(weights[wi]);
        w1 = nsets == 2 ? reinterpret_cast<const float*>(weights[nptr + wi]) : nullptr;
        ++wi;
    };
    int e;
    // levels P-1 / P (second-lowest ConvBlock, pool, bottleneck, transpose conv, first up-path ConvBlock): one fused kernel per
    // plane when the planes are cfg 2's 52 x 4 with 64 channels (unet_bottom.hip); its output carries ONE statistics record
    const bool fuse = pools >= 2 && !plane_kernel_enabled() && unet_bottom_applies(p.ch[pools - 1], p.hs[pools - 1], p.wsz[pools - 1]) &&
                      p.ch[pools - 2] * 2 == p.ch[pools - 1] && p.hs[pools - 2] == 2 * p.hs[pools - 1] && p.wsz[pools - 2] == 2 * p.wsz[pools - 1];
    // ---- down path (unet.py:94-97) + bottleneck (:99)
    for (int d = 0; d <= pools; ++d) {
        if (fuse && d >= pools - 1) { wi += 2; continue; }
        const bool last = d == pools;
        float* mid = p.mid[d]; float* pmid = p.pmid[d];
        float* out = last ? p.bott : p.skip[d];
        float* pout = last ? p.pbott : p.pskip[d];
        next();
        if (d == 0)
            e = cine_conv3x3_in(x, nullptr, 0, in_ch, 0, h, w, nullptr, nullptr, 0, 0, 0, 0, 0, w0, w1, split,
                                mid, pmid, n, p.ch[0], h, w, kEps, kSlope, stream);
        else
            e = cine_conv3x3_in(p.skip[d - 1], p.pskip[d - 1], p.np_conv[d - 1], p.ch[d - 1], 2, p.hs[d - 1], p.wsz[d - 1],
                                nullptr, nullptr, 0, 0, 0, 0, 0, w0, w1, split,
                                mid, pmid, n, p.ch[d], p.hs[d], p.wsz[d], kEps, kSlope, stream);
        if (e) return e;
        next();
        e = cine_conv3x3_in(mid, pmid, p.np_conv[d], p.ch[d], 1, p.hs[d], p.wsz[d], nullptr, nullptr, 0, 0, 0, 0, 0,
                            w0, w1, split, out, pout, n, p.ch[d], p.hs[d], p.wsz[d], kEps, kSlope, stream);
        if (e) return e;
    }
    // ---- up path (unet.py:102-123)
    const float* cur = p.bott; const float* pcur = p.pbott;
    int np_cur = p.np_conv[pools];
    for (int u = 0; u < pools; ++u) {
        const int d = pools - 1 - u;
        if (fuse && u == 0) {
            // weights in module order: [2(P-1)], [2(P-1)+1] level P-1 block; [2P], [2P+1] bottleneck; then tconv, conv, conv
            BottomArgs ba{};
            ba.x1 = p.skip[d - 1]; ba.px1 = p.pskip[d - 1]; ba.np1 = p.np_conv[d - 1];
            const int base = 2 * d;
            for (int l = 0; l < 7; ++l) {
                ba.w[l][0] = reinterpret_cast<const float*>(weights[base + l]);
                ba.w[l][1] = nsets == 2 ? reinterpret_cast<const float*>(weights[nptr + base + l]) : ba.w[l][0];
            }
            ba.set_split = split; ba.skip2 = p.skip[d]; ba.y = p.cb[d]; ba.py = p.pcb[d]; ba.eps = kEps; ba.slope = kSlope;
            if ((e = launch_unet_bottom(ba, n, as_stream(stream)))) return e;
            wi = base + 7;
            cur = p.cb[d]; pcur = p.pcb[d]; np_cur = 1;
            continue;
        }
        next();   // transpose conv: level d+1 -> (2 h_{d+1}, 2 w_{d+1}), ch_d channels
        e = cine_tconv2x2_in(cur, pcur, np_cur, 1, w0, w1, split, p.up[d], p.pup[d], n,
                             p.ch[d + 1], p.ch[d], p.hs[d + 1], p.wsz[d + 1], kEps, kSlope, stream);
        if (e) return e;
        next();   // cat([up, skip]) -> conv1; `up` reads as zero beyond its extent (zero pad, :106-120)
        e = cine_conv3x3_in(p.up[d], p.pup[d], p.np_tconv[d], p.ch[d], 1, 2 * p.hs[d + 1], 2 * p.wsz[d + 1],
                            p.skip[d], p.pskip[d], p.np_conv[d], p.ch[d], 1, p.hs[d], p.wsz[d], w0, w1, split,
                            p.ca[d], p.pca[d], n, p.ch[d], p.hs[d], p.wsz[d], kEps, kSlope, stream);
        if (e) return e;
        next();
        e = cine_conv3x3_in(p.ca[d], p.pca[d], p.np_conv[d], p.ch[d], 1, p.hs[d], p.wsz[d], nullptr, nullptr, 0, 0, 0, 0, 0,
                            w0, w1, split, p.cb[d], p.pcb[d], n, p.ch[d], p.hs[d], p.wsz[d], kEps, kSlope, stream);
        if (e) return e;
        cur = p.cb[d]; pcur = p.pcb[d]; np_cur = p.np_conv[d];
    }
    // ---- final 1x1 conv + bias (unet.py:69)
    const float* wf0 = reinterpret_cast<const float*>(weights[wi]);
    const float* bf0 = reinterpret_cast<const float*>(weights[wi + 1]);
    const float* wf1 = nsets == 2 ? reinterpret_cast<const float*>(weights[nptr + wi]) : nullptr;
    const float* bf1 = nsets == 2 ? reinterpret_cast<const float*>(weights[nptr + wi + 1]) : nullptr;
    e = cine_conv1x1_bias(cur, pcur, np_cur, 1, wf0, bf0, wf1, bf1, split, y, n, chans, out_ch, h, w,
                          kEps, kSlope, stream);
    if (e) return e;
    PlaneRecorder* rec = guard.r;
    guard.r = nullptr;
    return rec ? plane_record_end(rec, as_stream(stream), p.prog) : CINE_OK;
}
