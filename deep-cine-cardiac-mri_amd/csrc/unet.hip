// unet.hip -- launch sequence of one 2-D U-Net pass (reference denoisers/unet.py:73-125).
//
// Activations stay raw (pre-InstanceNorm) in HBM together with per-tile partial statistics
// {count, mean, M2}; every consumer merges the partials, normalises + LeakyReLUs (and pools /
// concatenates) while staging its operands.  Two weight sets (the x-f and y-f U-Nets of one
// cascade, varnet.py:224-226) run in the SAME launches: samples [0, n/2) use set 0, the rest set 1.
// Workspace: one raw buffer per skip level plus three rotating scratch buffers.
#include <algorithm>
#include <cstdlib>
#include "common.h"
#include "grad.h"
#include "conv_cfg.h"

using namespace cine;

extern "C" int cine_conv_stat_partials(int cout, int h, int w, int is_tconv);
extern "C" int cine_instnorm_merge(const float* part, float* out, long planes, int np, void* stream);

namespace {

constexpr float kEps = 1e-5f;     // nn.InstanceNorm2d default (unet.py:161)

// Dropout2d behind a ConvBlock's LeakyReLU (unet.py:163,167; training mode, drop_prob > 0): the multiplier d of a (sample, channel) plane is 0 or
// 1 / (1 - p), and d LeakyReLU(v) = LeakyReLU(d v) for d >= 0 -- so it is folded into the plane's statistics records: every consumer
// (conv staging, weight-gradient re-activation, the pooled / concatenated reads) computes rstd from the merged {count, mean, M2}, and adding
// (M2 + count eps) (1 / d^2 - 1) to ONE record's M2 makes that rstd' = d rstd (the merge is linear in the M2 entries); d = 0: M2 = +inf -> rstd' = 0.
__global__ void dropout_stats_kernel(float* part, int np, long planes, const float* __restrict__ drop, float eps) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= planes) return;
    const float d = drop[i];
    if (d == 1.f) return;
    float* p = part + i * np * 3;
    if (!(d > 0.f)) { p[2] = __builtin_inff(); return; }
    float cnt = 0.f, mean = 0.f;
    for (int k = 0; k < np; ++k) { cnt += p[3 * k]; mean += p[3 * k] * p[3 * k + 1]; }
    mean /= cnt;
    float m2 = 0.f;
    for (int k = 0; k < np; ++k) { const float dl = p[3 * k + 1] - mean; m2 += p[3 * k + 2] + p[3 * k] * dl * dl; }
    p[2] += (m2 + cnt * eps) * (1.f / (d * d) - 1.f);
}
}  // namespace
namespace cine {
int launch_dropout_stats(float* part, int np, long planes, const float* drop, hipStream_t st) {
    hipLaunchKernelGGL(dropout_stats_kernel, dim3((unsigned)ceil_div(planes, 256L)), dim3(256), 0, st, part, np, planes, drop, kEps);
    return check_launch("dropout_stats_kernel");
}
}  // namespace cine
namespace {
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
    // inference only: one MERGED statistics record per (sample, channel) beside the per-tile records of the layers that emit many (wide planes: the
    // sensitivity network's 208 x 208 level has 52 per channel, and every consumer workgroup used to merge them again in its prologue -- 39 % of
    // such a workgroup's life); NULL where the layer's records stay as they are
    float *mskip[8], *mmid[8], *mbott, *mup[8], *mca[8], *mcb[8];
};
constexpr int kMergeMin = 12;          // records per channel from which a layer's statistics are merged once by cine_instnorm_merge

// private == false: three rotating scratch buffers sized for the largest layer (layers are separated by kernel boundaries, so
// a buffer may hold different shapes over time).  private == true: every layer output owns its memory, sample n of every
// tensor sits at n * (its dense size) -- what the training path needs (the backward pass reads all of them).
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
    for (int d = 0; d < 8; ++d) p.mskip[d] = p.mmid[d] = p.mup[d] = p.mca[d] = p.mcb[d] = nullptr;
    p.mbott = nullptr;
    if (priv) {
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
    float *scr[3], *pscr[3], *mscr[3];
    size_t bigm = 0;                    // merged records: (n, ch, 3) of the widest layer that merges
    for (int d = 0; d <= pools; ++d)
        if (p.np_conv[d] >= kMergeMin || (d < pools && p.np_tconv[d] >= kMergeMin)) bigm = std::max(bigm, (size_t)n * p.ch[d] * 3);
    for (int i = 0; i < 3; ++i) { scr[i] = b.take(big); pscr[i] = b.take(bigp); mscr[i] = bigm ? b.take(bigm) : nullptr; }
    for (int d = 0; d < pools; ++d) p.mskip[d] = p.np_conv[d] >= kMergeMin ? b.take((size_t)n * p.ch[d] * 3) : nullptr;
    auto mif = [&](float* m, int np) { return np >= kMergeMin ? m : nullptr; };
    for (int d = 0; d <= pools; ++d) { p.mid[d] = scr[0]; p.pmid[d] = pscr[0]; p.mmid[d] = mif(mscr[0], p.np_conv[d]); }
    p.bott = scr[1]; p.pbott = pscr[1]; p.mbott = mif(mscr[1], p.np_conv[pools]);
    int cur = 1;
    for (int u = 0; u < pools; ++u) {
        const int d = pools - 1 - u;
        const int a = (cur + 1) % 3, c = (cur + 2) % 3;
        p.up[d] = scr[a]; p.pup[d] = pscr[a]; p.mup[d] = mif(mscr[a], p.np_tconv[d]);
        p.ca[d] = scr[c]; p.pca[d] = pscr[c]; p.mca[d] = mif(mscr[c], p.np_conv[d]);
        p.cb[d] = scr[a]; p.pcb[d] = pscr[a]; p.mcb[d] = mif(mscr[a], p.np_conv[d]);
        cur = a;
    }
}

}  // namespace

extern "C" size_t cine_unet2d_ws_bytes(int n, int h, int w, int in_ch, int out_ch, int chans, int pools) {
    if (n <= 0 || h <= 0 || w <= 0 || chans <= 0 || pools <= 0 || pools > 6) return 0;
    (void)in_ch; (void)out_ch;
    Plan p; Bump b{nullptr, 0};
    build(p, b, n, h, w, chans, pools, false);
    return b.off;
}

// training: every layer output keeps its own memory (the backward pass reads all of them)
extern "C" size_t cine_unet2d_train_ws_bytes(int n, int h, int w, int in_ch, int out_ch, int chans, int pools) {
    if (n <= 0 || h <= 0 || w <= 0 || chans <= 0 || pools <= 0 || pools > 6) return 0;
    (void)in_ch; (void)out_ch;
    Plan p; Bump b{nullptr, 0};
    build(p, b, n, h, w, chans, pools, true);
    return b.off;
}

// One launch sequence of the U-Net over the `n` planes at x / y with the buffers of plan `p` (built for exactly these planes): `wa` = the
// pointer list of the weight set of samples [0, split), `wb` (may be NULL: one set) of the rest.  Only launches [l0, l1) of the sequence
// are enqueued (5 * pools + 3 in all), so that concurrent branches can be enqueued layer by layer.
static int unet_launches(int pools) { return 5 * pools + 3; }
static int run_unet(const Plan& p, const float* x, float* y, const void* const* wa, const void* const* wb, int split,
                    int n, int h, int w, int in_ch, int out_ch, int chans, int pools, float kSlope, void* stream, int l0 = 0, int l1 = 1 << 30,
                    const DropMap* dm = nullptr) {
    int wi = 0, li = 0;
    auto drop = [&](int conv, float* part, int np, int ch) {          // fold this conv's Dropout2d multipliers into its statistics records
        if (!dm || !dm->base) return (int)CINE_OK;
        return launch_dropout_stats(part, np, (long)n * ch, dm->at(conv), as_stream(stream));
    };
    const float *w0, *w1;
    auto next = [&]() {
        w0 = reinterpret_cast<const float*>(wa[wi]);
        w1 = wb ? reinterpret_cast<const float*>(wb[wi]) : nullptr;
        ++wi;
        const bool on = li >= l0 && li < l1;
        ++li;
        return on;
    };
    int e;
    // a layer with many per-tile records gets them merged ONCE (one small launch) instead of once per consumer workgroup; (ptr, np) = what the consumers read
    struct Rec { const float* p; int np; };
    auto fold = [&](float* part, int np, float* merged, int ch, Rec& r) -> int {
        r = Rec{part, np};
        if (!merged) return CINE_OK;
        r = Rec{merged, 1};
        return cine_instnorm_merge(part, merged, (long)n * ch, np, stream);
    };
    Rec rskip[8], rmid{nullptr, 0}, rcur{nullptr, 0}, rup{nullptr, 0}, rca{nullptr, 0};
    for (int d = 0; d < 8; ++d) rskip[d] = Rec{p.pskip[d], p.np_conv[d]};
    // ---- down path (unet.py:94-97) + bottleneck (:99)
    for (int d = 0; d <= pools; ++d) {
        const bool last = d == pools;
        float* mid = p.mid[d]; float* pmid = p.pmid[d];
        float* out = last ? p.bott : p.skip[d];
        float* pout = last ? p.pbott : p.pskip[d];
        rmid = Rec{pmid, p.np_conv[d]};
        if (next()) {
            if (d == 0)
                e = cine_conv3x3_in(x, nullptr, 0, in_ch, 0, h, w, nullptr, nullptr, 0, 0, 0, 0, 0, w0, w1, split,
                                    mid, pmid, n, p.ch[0], h, w, kEps, kSlope, stream);
            else
                e = cine_conv3x3_in(p.skip[d - 1], rskip[d - 1].p, rskip[d - 1].np, p.ch[d - 1], 2, p.hs[d - 1], p.wsz[d - 1],
                                    nullptr, nullptr, 0, 0, 0, 0, 0, w0, w1, split,
                                    mid, pmid, n, p.ch[d], p.hs[d], p.wsz[d], kEps, kSlope, stream);
            if (e || (e = drop(DropMap::down(d, 0), pmid, p.np_conv[d], p.ch[d])) || (e = fold(pmid, p.np_conv[d], p.mmid[d], p.ch[d], rmid))) return e;
        } else if (p.mmid[d]) rmid = Rec{p.mmid[d], 1};
        Rec rout{pout, p.np_conv[d]};
        if (next()) {
            e = cine_conv3x3_in(mid, rmid.p, rmid.np, p.ch[d], 1, p.hs[d], p.wsz[d], nullptr, nullptr, 0, 0, 0, 0, 0,
                                w0, w1, split, out, pout, n, p.ch[d], p.hs[d], p.wsz[d], kEps, kSlope, stream);
            if (e || (e = drop(DropMap::down(d, 1), pout, p.np_conv[d], p.ch[d])) || (e = fold(pout, p.np_conv[d], last ? p.mbott : p.mskip[d], p.ch[d], rout))) return e;
        } else if (last ? p.mbott : p.mskip[d]) rout = Rec{last ? p.mbott : p.mskip[d], 1};
        if (last) rcur = rout; else rskip[d] = rout;
    }
    // ---- up path (unet.py:102-123)
    const float* cur = p.bott;
    for (int u = 0; u < pools; ++u) {
        const int d = pools - 1 - u;
        rup = Rec{p.pup[d], p.np_tconv[d]};
        if (next()) {   // transpose conv: level d+1 -> (2 h_{d+1}, 2 w_{d+1}), ch_d channels
            e = cine_tconv2x2_in(cur, rcur.p, rcur.np, 1, w0, w1, split, p.up[d], p.pup[d], n,
                                 p.ch[d + 1], p.ch[d], p.hs[d + 1], p.wsz[d + 1], kEps, kSlope, stream);
            if (e || (e = fold(p.pup[d], p.np_tconv[d], p.mup[d], p.ch[d], rup))) return e;
        } else if (p.mup[d]) rup = Rec{p.mup[d], 1};
        rca = Rec{p.pca[d], p.np_conv[d]};
        if (next()) {   // cat([up, skip]) -> conv1; `up` reads as zero beyond its extent (zero pad, :106-120)
            e = cine_conv3x3_in(p.up[d], rup.p, rup.np, p.ch[d], 1, 2 * p.hs[d + 1], 2 * p.wsz[d + 1],
                                p.skip[d], rskip[d].p, rskip[d].np, p.ch[d], 1, p.hs[d], p.wsz[d], w0, w1, split,
                                p.ca[d], p.pca[d], n, p.ch[d], p.hs[d], p.wsz[d], kEps, kSlope, stream);
            if (e || (e = drop(dm ? dm->up(d, 0) : 0, p.pca[d], p.np_conv[d], p.ch[d])) || (e = fold(p.pca[d], p.np_conv[d], p.mca[d], p.ch[d], rca))) return e;
        } else if (p.mca[d]) rca = Rec{p.mca[d], 1};
        rcur = Rec{p.pcb[d], p.np_conv[d]};
        if (next()) {
            e = cine_conv3x3_in(p.ca[d], rca.p, rca.np, p.ch[d], 1, p.hs[d], p.wsz[d], nullptr, nullptr, 0, 0, 0, 0, 0,
                                w0, w1, split, p.cb[d], p.pcb[d], n, p.ch[d], p.hs[d], p.wsz[d], kEps, kSlope, stream);
            if (e || (e = drop(dm ? dm->up(d, 1) : 0, p.pcb[d], p.np_conv[d], p.ch[d])) || (e = fold(p.pcb[d], p.np_conv[d], p.mcb[d], p.ch[d], rcur))) return e;
        } else if (p.mcb[d]) rcur = Rec{p.mcb[d], 1};
        cur = p.cb[d];
    }
    const float* pcur = rcur.p;
    const int np_cur = rcur.np;
    // ---- final 1x1 conv + bias (unet.py:69)
    const float* wf0 = reinterpret_cast<const float*>(wa[wi]);
    const float* bf0 = reinterpret_cast<const float*>(wa[wi + 1]);
    const float* wf1 = wb ? reinterpret_cast<const float*>(wb[wi]) : nullptr;
    const float* bf1 = wb ? reinterpret_cast<const float*>(wb[wi + 1]) : nullptr;
    if (li >= l0 && li < l1)
        return cine_conv1x1_bias(cur, pcur, np_cur, 1, wf0, bf0, wf1, bf1, split, y, n, chans, out_ch, h, w,
                                 kEps, kSlope, stream);
    return CINE_OK;
}

static int unet2d_check(const float* x, float* y, const void* const* weights, int nsets, int n, int h, int w, int in_ch, int out_ch,
                        int chans, int pools, float kSlope, void* ws) {
    CINE_REQUIRE(x && y && weights && ws, CINE_EINVAL, "cine_unet2d_forward: null pointer");
    CINE_REQUIRE(kSlope >= 0.f && kSlope <= 1.f, CINE_EINVAL, "cine_unet2d_forward: LeakyReLU slope %g outside [0, 1]", (double)kSlope);
    CINE_REQUIRE(nsets == 1 || nsets == 2, CINE_EINVAL, "cine_unet2d_forward: nsets must be 1 or 2");
    CINE_REQUIRE(n > 0 && n % nsets == 0, CINE_EINVAL, "cine_unet2d_forward: n=%d not divisible by nsets=%d", n, nsets);
    CINE_REQUIRE(h > 0 && w > 0 && in_ch > 0 && out_ch > 0 && chans > 0 && pools > 0 && pools <= 6, CINE_EINVAL,
                 "cine_unet2d_forward: bad sizes");
    CINE_REQUIRE((h >> pools) >= 1 && (w >> pools) >= 1, CINE_EUNSUPPORTED,
                 "cine_unet2d_forward: %dx%d too small for %d pools", h, w, pools);
    const int nptr = 5 * pools + 4;
    for (int i = 0; i < nsets * nptr; ++i)
        CINE_REQUIRE(weights[i], CINE_EINVAL, "cine_unet2d_forward: weights[%d] is null", i);
    return CINE_OK;
}

static int unet2d_forward_impl(const float* x, float* y, const void* const* weights, int nsets,
                               int n, int h, int w, int in_ch, int out_ch, int chans, int pools, float kSlope,
                               void* ws, size_t ws_bytes, void* stream, bool train) {
    if (int e = unet2d_check(x, y, weights, nsets, n, h, w, in_ch, out_ch, chans, pools, kSlope, ws)) return e;
    const size_t need = train ? cine_unet2d_train_ws_bytes(n, h, w, in_ch, out_ch, chans, pools)
                              : cine_unet2d_ws_bytes(n, h, w, in_ch, out_ch, chans, pools);
    CINE_REQUIRE(ws_bytes >= need, CINE_EWORKSPACE, "cine_unet2d_forward: workspace %zu < %zu", ws_bytes, need);
    const int nptr = 5 * pools + 4;
    Plan p; Bump b{reinterpret_cast<char*>(ws), 0};
    build(p, b, n, h, w, chans, pools, train);
    return run_unet(p, x, y, weights, nsets == 2 ? weights + nptr : nullptr, n / nsets, n, h, w, in_ch, out_ch, chans, pools, kSlope, stream);
}

extern "C" int cine_unet2d_forward(const float* x, float* y, const void* const* weights, int nsets,
                                   int n, int h, int w, int in_ch, int out_ch, int chans, int pools, float slope,
                                   void* ws, size_t ws_bytes, void* stream) {
    return unet2d_forward_impl(x, y, weights, nsets, n, h, w, in_ch, out_ch, chans, pools, slope, ws, ws_bytes, stream, false);
}
extern "C" int cine_unet2d_forward_train(const float* x, float* y, const void* const* weights, int nsets,
                                         int n, int h, int w, int in_ch, int out_ch, int chans, int pools, float slope,
                                         void* ws, size_t ws_bytes, void* stream) {
    return unet2d_forward_impl(x, y, weights, nsets, n, h, w, in_ch, out_ch, chans, pools, slope, ws, ws_bytes, stream, true);
}

// ---------------------------------------------------------------- the same U-Net pass as CONCURRENT BRANCHES
// The planes of a U-Net pass are independent (varnet.py:216-232: the x-f and the y-f network of a cascade meet only in the sum behind
// them), yet on ONE stream every layer is a kernel boundary at which the whole chip drains: a cfg-2 level-1 layer is 800 workgroups on 768
// resident slots, a level-2 layer 400 workgroups on 512 -- a round plus a sliver.  Here the planes are cut into nside + 1
// contiguous runs (never across the two weight sets), and each run goes through the SAME launch sequence -- same kernels, same tiles, same
// statistics records per plane: bit-identical outputs -- on its own stream: branch 0 on `stream`, branch k on side[k - 1], forked from and
// joined back into `stream` with events, so a branch's next layer starts in the slots its sibling's last round leaves empty.
namespace {
struct Cut { int a, n, set; };       // planes [a, a + n) of weight set `set`
int cut_planes(int n, int nsets, int nbranch, Cut* c) {
    if (nbranch == 1) { c[0] = Cut{0, n, -1}; return 1; }          // one run over every plane (set -1: both weight sets when there are two)
    const int per_set = nbranch / nsets, ns = n / nsets;
    int k = 0;
    for (int s = 0; s < nsets; ++s)
        for (int j = 0; j < per_set; ++j) {
            const int lo = (int)((long)ns * j / per_set), hi = (int)((long)ns * (j + 1) / per_set);
            if (hi > lo) c[k++] = Cut{s * ns + lo, hi - lo, s};
        }
    return k;
}
bool branches_ok(int n, int nsets, int nbranch) {
    return nbranch == 1 || (nbranch >= 2 && nbranch <= 8 && nbranch % nsets == 0 && n / nbranch >= 1);
}
}  // namespace

extern "C" size_t cine_unet2d_drop_floats(int n, int chans, int pools) {
    if (n <= 0 || chans <= 0 || pools <= 0 || pools > 6) return 0;
    const DropMap dm{nullptr, n, 0, chans, pools};
    return (size_t)dm.off(4 * pools + 2);
}

extern "C" size_t cine_unet2d_branch_ws_bytes(int n, int h, int w, int in_ch, int out_ch, int chans, int pools, int nsets, int nbranch, int train) {
    if (n <= 0 || (nsets != 1 && nsets != 2) || n % nsets || !branches_ok(n, nsets, nbranch)) return 0;
    if (train) return cine_unet2d_train_ws_bytes(n, h, w, in_ch, out_ch, chans, pools);      // the backward pass reads ONE layout: the branches write their planes of it
    Cut c[8];
    const int k = cut_planes(n, nsets, nbranch, c);
    size_t tot = 0;
    for (int i = 0; i < k; ++i) {
        const size_t one = cine_unet2d_ws_bytes(c[i].n, h, w, in_ch, out_ch, chans, pools);
        if (!one) return 0;
        tot += one;
    }
    return tot;
}

extern "C" int cine_unet2d_forward_branches(const float* x, float* y, const void* const* weights, int nsets,
                                            int n, int h, int w, int in_ch, int out_ch, int chans, int pools, float slope,
                                            void* ws, size_t ws_bytes, void* stream, void* const* side, int nside, int train, const float* drop) {
    if (int e = unet2d_check(x, y, weights, nsets, n, h, w, in_ch, out_ch, chans, pools, slope, ws)) return e;
    const bool interleave = (train & 2) != 0;
    train &= 1;
    CINE_REQUIRE(!drop || train, CINE_EINVAL, "cine_unet2d_forward_branches: Dropout multipliers belong to the training forward (eval mode has no dropout)");
    const int nbranch = nside + 1;
    CINE_REQUIRE(nside >= 0 && (nside == 0 || side), CINE_EINVAL, "cine_unet2d_forward_branches: side streams missing");
    CINE_REQUIRE(branches_ok(n, nsets, nbranch), CINE_EINVAL, "cine_unet2d_forward_branches: %d planes of %d set(s) do not cut into %d branches", n, nsets, nbranch);
    for (int i = 0; i < nside; ++i)
        CINE_REQUIRE(side[i] && side[i] != stream, CINE_EINVAL, "cine_unet2d_forward_branches: side[%d] is NULL or the main stream", i);
    const size_t need = cine_unet2d_branch_ws_bytes(n, h, w, in_ch, out_ch, chans, pools, nsets, nbranch, train);
    CINE_REQUIRE(need && ws_bytes >= need, CINE_EWORKSPACE, "cine_unet2d_forward_branches: workspace %zu < %zu", ws_bytes, need);
    const int nptr = 5 * pools + 4;
    Cut c[8];
    const int k = cut_planes(n, nsets, nbranch, c);
    Plan full;
    size_t off = 0;
    if (train) { Bump b{reinterpret_cast<char*>(ws), 0}; build(full, b, n, h, w, chans, pools, true); }
    hipStream_t main = as_stream(stream);
    hipEvent_t fork = nullptr, done[8] = {};
    bool ok = k <= 1 || hipEventCreateWithFlags(&fork, hipEventDisableTiming) == hipSuccess;
    for (int i = 1; i < k && ok; ++i) ok = hipEventCreateWithFlags(&done[i], hipEventDisableTiming) == hipSuccess;
    auto cleanup = [&]() { if (fork) (void)hipEventDestroy(fork); for (auto ev : done) if (ev) (void)hipEventDestroy(ev); };
    if (!ok) { cleanup(); set_error("cine_unet2d_forward_branches: hipEventCreate failed"); return CINE_EHIP; }
    if (k > 1) { (void)hipEventRecord(fork, main); diag_count(D_UNET_BRANCHED); }
    const AloneScope alone(k > 1 && !train);    // side streams = the caller runs this slice beside nothing else (conv_cfg.h)
    int err = CINE_OK;
    const long xs = (long)in_ch * h * w, ys = (long)out_ch * h * w;
    Plan plans[8];
    for (int i = 0; i < k; ++i) {
        Plan& p = plans[i];
        if (train) {
            // this branch's planes inside the one layout the backward pass reads (sample s of a tensor sits at s * its dense size)
            p = full;
            const long a = c[i].a;
            for (int d = 0; d <= pools; ++d) {
                const long e_ = (long)p.ch[d] * p.hs[d] * p.wsz[d], pe = (long)p.ch[d] * p.np_conv[d] * 3;
                p.mid[d] += a * e_; p.pmid[d] += a * pe;
                if (d < pools) {
                    p.skip[d] += a * e_; p.pskip[d] += a * pe;
                    p.ca[d] += a * e_; p.pca[d] += a * pe;
                    p.cb[d] += a * e_; p.pcb[d] += a * pe;
                    p.up[d] += a * (long)p.ch[d] * (2 * p.hs[d + 1]) * (2 * p.wsz[d + 1]);
                    p.pup[d] += a * (long)p.ch[d] * p.np_tconv[d] * 3;
                }
            }
            p.bott += a * (long)p.ch[pools] * p.hs[pools] * p.wsz[pools];
            p.pbott += a * (long)p.ch[pools] * p.np_conv[pools] * 3;
        } else {
            Bump b{reinterpret_cast<char*>(ws) + off, 0};
            build(p, b, c[i].n, h, w, chans, pools, false);
            off += b.off;
        }
        if (i > 0) (void)hipStreamWaitEvent(as_stream(side[i - 1]), fork, 0);
    }
    // enqueue order: a whole sequence per branch (the later branches start some 20 launches of host time behind the first, which keeps the
    // branches in DIFFERENT layers: measured faster than lockstep), or -- train bit 1 -- layer by layer over the branches (diagnostics)
    const int nl = unet_launches(pools), lstep = interleave ? 1 : nl;
    for (int l = 0; l < nl && !err; l += lstep)
        for (int i = 0; i < k && !err; ++i) {
            hipStream_t st = i == 0 ? main : as_stream(side[i - 1]);
            const bool whole = c[i].set < 0;
            const void* const* wa = weights + (c[i].set > 0 ? nptr : 0);
            const void* const* wb = whole && nsets == 2 ? weights + nptr : nullptr;
            const DropMap dm{drop, n, c[i].a, chans, pools};
            err = run_unet(plans[i], x + c[i].a * xs, y + c[i].a * ys, wa, wb, whole ? n / nsets : c[i].n, c[i].n, h, w, in_ch, out_ch, chans, pools, slope,
                           st, l, l + lstep, &dm);
        }
    for (int i = 1; i < k; ++i) {       // joined even after an error: no dangling fork
        (void)hipEventRecord(done[i], as_stream(side[i - 1])); (void)hipStreamWaitEvent(main, done[i], 0);
    }
    cleanup();
    return err;
}

// ---------------------------------------------------------------- backward pass (training)
// Scratch of the backward pass: two rotating gradient buffers (the largest is the input gradient of a level-0 conv over
// the 2 x chans concat), the concat gradients of every level (their skip halves are consumed on the way down), the pooled
// gradient, and the partial sums of the weight-gradient kernel.
namespace {
struct BwdPlan {
    float *A, *B[2], *cat[8], *pool, *wg, *mat, *inb;
    size_t wg_floats, mat_floats, inb_floats;
};
void build_bwd(BwdPlan& q, const Plan& p, Bump& b, int n, int in_ch, int out_ch) {
    const int P = p.P;
    auto elems = [&](int d) { return (size_t)n * p.ch[d] * p.hs[d] * p.wsz[d]; };
    size_t big = 0;
    for (int d = 0; d <= P; ++d) big = std::max(big, elems(d));
    big = std::max(big, (size_t)n * std::max(in_ch, 16) * p.hs[0] * p.wsz[0]);
    q.A = b.take(big); q.B[0] = b.take(big); q.B[1] = b.take(big);
    for (int d = 0; d < P; ++d) q.cat[d] = b.take(2 * elems(d));
    q.pool = b.take(P > 0 ? elems(1) / 2 + 16 : 16);          // (n, ch[d], hs[d+1], wsz[d+1]) <= elems(d) / 4
    for (int d = 0; d < P; ++d) (void)0;
    size_t wg = 0;
    for (int d = 0; d <= P; ++d) {
        const int cin1 = d ? p.ch[d - 1] : in_ch;
        wg = std::max(wg, wgrad_ws_floats(p.ch[d], cin1, 9, n));
        wg = std::max(wg, wgrad_ws_floats(p.ch[d], p.ch[d], 9, n));
        if (d < P) {
            wg = std::max(wg, wgrad_ws_floats(p.ch[d], 2 * p.ch[d], 9, n));
            wg = std::max(wg, wgrad_ws_floats(4 * p.ch[d], p.ch[d + 1], 1, n));
        }
    }
    wg = std::max(wg, wgrad_ws_floats(out_ch, p.ch[0], 1, n));
    wg = std::max(wg, (size_t)n * out_ch);                  // bias-gradient partial sums
    q.wg_floats = wg;
    q.wg = b.take(wg);
    size_t mat = 16;                                          // pooled conv inputs (first conv of every level below the top), materialised for the weight gradient
    for (int d = 1; d <= P; ++d) mat = std::max(mat, (size_t)n * p.ch[d - 1] * p.hs[d] * p.wsz[d]);
    q.mat_floats = mat;
    q.mat = b.take(mat);
    size_t inb = 16;                                          // chunk sums of the InstanceNorm backward on few, large planes (the sens-net's 200 x 200 ones)
    for (int d = 0; d <= P; ++d) inb = std::max(inb, in_lrelu_bwd_ws_floats(n, p.ch[d], p.hs[d], p.wsz[d]));
    q.inb_floats = inb;
    q.inb = b.take(inb);
}
}  // namespace

extern "C" size_t cine_unet2d_backward_ws_bytes(int n, int h, int w, int in_ch, int out_ch, int chans, int pools) {
    if (n <= 0 || h <= 0 || w <= 0 || chans <= 0 || pools <= 0 || pools > 6 || in_ch <= 0 || out_ch <= 0) return 0;
    Plan p; Bump b0{nullptr, 0};
    build(p, b0, n, h, w, chans, pools, true);
    BwdPlan q; Bump b{nullptr, 0};
    build_bwd(q, p, b, n, in_ch, out_ch);
    return b.off;
}

// Gradients of cine_unet2d_forward_train.  `fwd_ws` is the workspace that call filled (all raw layer outputs + statistics);
// x its input, gy = d loss / d y.  `wdgrad`: host array of device pointers ordered like `weights` of the forward, holding the
// INPUT-GRADIENT packings (cine_pack_conv3x3_dgrad / _tconv2x2_dgrad / _conv1x1_dgrad; the bias slot is unused).  `grads`:
// host array in the same order of device pointers to the weight gradients in the parameters' own layouts ((cout, cin, 3, 3),
// (cin, cout, 2, 2), (cout, cin), (cout)); they are ACCUMULATED into (+=).  gx (n, in_ch, h, w) may be NULL.
extern "C" int cine_unet2d_backward_drop(const float* x, const float* gy, const void* const* wdgrad, void* const* grads, int nsets,
                                         int n, int h, int w, int in_ch, int out_ch, int chans, int pools, float kSlope,
                                         const void* fwd_ws, size_t fwd_ws_bytes, void* ws, size_t ws_bytes, float* gx, const float* drop, void* stream);
extern "C" int cine_unet2d_backward(const float* x, const float* gy, const void* const* wdgrad, void* const* grads, int nsets,
                                    int n, int h, int w, int in_ch, int out_ch, int chans, int pools, float kSlope,
                                    const void* fwd_ws, size_t fwd_ws_bytes, void* ws, size_t ws_bytes, float* gx, void* stream) {
    return cine_unet2d_backward_drop(x, gy, wdgrad, grads, nsets, n, h, w, in_ch, out_ch, chans, pools, kSlope, fwd_ws, fwd_ws_bytes, ws, ws_bytes, gx, nullptr, stream);
}
// `drop`: the Dropout2d multipliers the training forward applied (cine_unet2d_forward_branches; NULL = none)
extern "C" int cine_unet2d_backward_drop(const float* x, const float* gy, const void* const* wdgrad, void* const* grads, int nsets,
                                         int n, int h, int w, int in_ch, int out_ch, int chans, int pools, float kSlope,
                                         const void* fwd_ws, size_t fwd_ws_bytes, void* ws, size_t ws_bytes, float* gx, const float* drop, void* stream) {
    CINE_REQUIRE(x && gy && wdgrad && grads && fwd_ws && ws, CINE_EINVAL, "cine_unet2d_backward: null pointer");
    CINE_REQUIRE(kSlope >= 0.f && kSlope <= 1.f, CINE_EINVAL, "cine_unet2d_backward: LeakyReLU slope %g outside [0, 1]", (double)kSlope);
    CINE_REQUIRE(nsets == 1 || nsets == 2, CINE_EINVAL, "cine_unet2d_backward: nsets must be 1 or 2");
    CINE_REQUIRE(n > 0 && n % nsets == 0 && n <= 65535, CINE_EINVAL, "cine_unet2d_backward: n=%d", n);
    CINE_REQUIRE(h > 0 && w > 0 && in_ch > 0 && out_ch > 0 && chans > 0 && pools > 0 && pools <= 6, CINE_EINVAL, "cine_unet2d_backward: bad sizes");
    CINE_REQUIRE(fwd_ws_bytes >= cine_unet2d_train_ws_bytes(n, h, w, in_ch, out_ch, chans, pools), CINE_EWORKSPACE,
                 "cine_unet2d_backward: forward workspace too small");
    CINE_REQUIRE(ws_bytes >= cine_unet2d_backward_ws_bytes(n, h, w, in_ch, out_ch, chans, pools), CINE_EWORKSPACE,
                 "cine_unet2d_backward: workspace too small");
    const int nptr = 5 * pools + 4;
    for (int i = 0; i < nsets * nptr; ++i) {
        CINE_REQUIRE(grads[i], CINE_EINVAL, "cine_unet2d_backward: grads[%d] is null", i);
        CINE_REQUIRE(wdgrad[i] || (i % nptr) == nptr - 1, CINE_EINVAL, "cine_unet2d_backward: wdgrad[%d] is null", i);
    }
    Plan p; Bump bf{const_cast<char*>(reinterpret_cast<const char*>(fwd_ws)), 0};
    build(p, bf, n, h, w, chans, pools, true);
    BwdPlan q; Bump bb{reinterpret_cast<char*>(ws), 0};
    build_bwd(q, p, bb, n, in_ch, out_ch);
    hipStream_t st = as_stream(stream);
    const int P = pools, split = n / nsets;
    auto wd = [&](int i, int set) { return reinterpret_cast<const float*>(wdgrad[(set && nsets == 2 ? nptr : 0) + i]); };
    auto gr = [&](int i, int set) { return nsets == 2 || set == 0 ? reinterpret_cast<float*>(grads[(set ? nptr : 0) + i]) : nullptr; };
    auto wd2 = [&](int i) { return nsets == 2 ? wd(i, 1) : nullptr; };
    const int sp = nsets == 2 ? split : n;
    int e;
    // weight-list indices (module order, as in the forward)
    auto i_down = [&](int d, int k) { return 2 * d + k; };                  // d = 0..P (P = bottleneck), k = 0 | 1
    auto i_up = [&](int d, int k) { return 2 * P + 2 + 3 * (P - 1 - d) + k; };   // k = 0 tconv, 1 conv1, 2 conv2
    const int i_fin = 5 * P + 2, i_bias = 5 * P + 3;
    auto src = [&](const float* t, const float* part, int c, int mode, int hh, int ww, int np) { return Src{t, part, c, mode, hh, ww, np, 0, 1}; };
    const Src none{nullptr, nullptr, 0, 0, 0, 0, 0, 0, 1};
    SideLane lane(st);              // weight gradients on the side stream (grad.h); g alternates between q.B[0] / q.B[1]
    auto next_g = [&]() { lane.before_write(); return q.B[lane.slot()]; };
    auto on_side = [&](auto&& launch) { hipStream_t sw = lane.fork(); const int err = launch(sw); lane.launched(); return err; };
    auto wgrad3 = [&](const Src& s0, const Src& s1, const float* g, int rows, int hh, int ww, int wi) {
        WgArgs a{}; a.s0 = s0; a.s1 = s1; a.cin = src_cin(s0) + src_cin(s1); a.g = g; a.g_mode = 0; a.rows = rows;
        a.n = n; a.H = hh; a.W = ww; a.set_split = sp; a.eps = kEps; a.slope = kSlope;
        a.mat = q.mat; a.mat_floats = q.mat_floats;
        return on_side([&](hipStream_t sw) { return launch_wgrad(a, 9, 0, gr(wi, 0), gr(wi, 1), q.wg, q.wg_floats, sw); });
    };
    const DropMap dm{drop, n, 0, chans, pools};
    auto inbwd = [&](const float* r, const float* part, int np, int c, int hh, int ww, const float* ga, int ca_total, int ca_off,
                     int ha, int wa, const float* gb, int hb, int wb, float* out, int conv = -1) {
        InBwdArgs a{r, part, np, GradPiece{ga, 1, ca_total, ca_off, ha, wa}, GradPiece{gb, gb ? 2 : 0, c, 0, hb, wb}, out, n, c, hh, ww, kEps, kSlope,
                    conv >= 0 ? dm.at(conv) : nullptr};
        return launch_in_lrelu_bwd_split(a, q.inb, q.inb_floats, st);      // one workgroup per plane unless the planes are few and large
    };

    // ---- final 1x1 conv + bias (unet.py:69): y = W act(cb_0) + b
    const long hw0 = (long)h * w;
    if ((e = launch_bias_grad(gy, n, out_ch, hw0, sp, gr(i_bias, 0), gr(i_bias, 1), q.wg, q.wg_floats, st))) return e;
    {
        WgArgs a{}; a.s0 = src(p.cb[0], p.pcb[0], chans, 1, h, w, p.np_conv[0]); a.s1 = none; a.cin = chans;
        a.g = gy; a.g_mode = 0; a.rows = out_ch; a.n = n; a.H = h; a.W = w; a.set_split = sp; a.eps = kEps; a.slope = kSlope;
        if ((e = on_side([&](hipStream_t sw) { return launch_wgrad(a, 1, 2, gr(i_fin, 0), gr(i_fin, 1), q.wg, q.wg_floats, sw); }))) return e;
    }
    if ((e = cine_conv1x1_dgrad(gy, wd(i_fin, 0), wd2(i_fin), sp, q.A, n, out_ch, chans, h, w, stream))) return e;   // A = d/d act(cb_0)

    // ---- up path, level 0 first (reverse of unet.py:102-123)
    for (int d = 0; d < P; ++d) {
        const int c = p.ch[d], hh = p.hs[d], ww = p.wsz[d], npc = p.np_conv[d];
        const int hu = 2 * p.hs[d + 1], wu = 2 * p.wsz[d + 1];           // extent of the transpose-conv output
        // second conv of the block: cb = conv(act(ca))
        float* B = next_g();
        if ((e = inbwd(p.cb[d], p.pcb[d], npc, c, hh, ww, q.A, c, 0, hh, ww, nullptr, 0, 0, B, dm.up(d, 1)))) return e;
        if ((e = wgrad3(src(p.ca[d], p.pca[d], c, 1, hh, ww, npc), none, B, c, hh, ww, i_up(d, 2)))) return e;
        if ((e = cine_conv3x3_dgrad(B, wd(i_up(d, 2), 0), wd2(i_up(d, 2)), sp, q.A, n, c, c, hh, ww, stream))) return e;
        // first conv: ca = conv(cat(act(up) zero-padded, act(skip)))
        B = next_g();
        if ((e = inbwd(p.ca[d], p.pca[d], npc, c, hh, ww, q.A, c, 0, hh, ww, nullptr, 0, 0, B, dm.up(d, 0)))) return e;
        if ((e = wgrad3(src(p.up[d], p.pup[d], c, 1, hu, wu, p.np_tconv[d]), src(p.skip[d], p.pskip[d], c, 1, hh, ww, npc), B, c, hh, ww, i_up(d, 1)))) return e;
        if ((e = cine_conv3x3_dgrad(B, wd(i_up(d, 1), 0), wd2(i_up(d, 1)), sp, q.cat[d], n, c, 2 * c, hh, ww, stream))) return e;
        // transpose conv: up = tconv(act(cur)), cur = cb[d+1] or the bottleneck output
        B = next_g();
        if ((e = inbwd(p.up[d], p.pup[d], p.np_tconv[d], c, hu, wu, q.cat[d], 2 * c, 0, hh, ww, nullptr, 0, 0, B))) return e;
        const bool bott = d + 1 == P;
        const float* cur = bott ? p.bott : p.cb[d + 1];
        const float* pcur = bott ? p.pbott : p.pcb[d + 1];
        const int c1 = p.ch[d + 1], h1 = p.hs[d + 1], w1 = p.wsz[d + 1];
        {
            WgArgs a{}; a.s0 = src(cur, pcur, c1, 1, h1, w1, p.np_conv[d + 1]); a.s1 = none; a.cin = c1;
            a.g = B; a.g_mode = 5; a.g_c = c; a.g_h = hu; a.g_w = wu; a.rows = 4 * c;
            a.n = n; a.H = h1; a.W = w1; a.set_split = sp; a.eps = kEps; a.slope = kSlope;
            if ((e = on_side([&](hipStream_t sw) { return launch_wgrad(a, 1, 1, gr(i_up(d, 0), 0), gr(i_up(d, 0), 1), q.wg, q.wg_floats, sw); }))) return e;
        }
        if ((e = cine_tconv2x2_dgrad(B, wd(i_up(d, 0), 0), wd2(i_up(d, 0)), sp, q.A, n, c1, c, h1, w1, stream))) return e;   // A = d/d act(cur)
    }
    // ---- bottleneck and down path (reverse of unet.py:94-99)
    for (int d = P; d >= 0; --d) {
        const int c = p.ch[d], hh = p.hs[d], ww = p.wsz[d], npc = p.np_conv[d];
        const float* out = d == P ? p.bott : p.skip[d];
        const float* pout = d == P ? p.pbott : p.pskip[d];
        float* B = next_g();
        if (d == P) {
            if ((e = inbwd(out, pout, npc, c, hh, ww, q.A, c, 0, hh, ww, nullptr, 0, 0, B, DropMap::down(d, 1)))) return e;
        } else {    // the skip tensor feeds the concat (second half of cat[d]) and the 2x2 average pool
            if ((e = inbwd(out, pout, npc, c, hh, ww, q.cat[d], 2 * c, c, hh, ww, q.pool, p.hs[d + 1], p.wsz[d + 1], B, DropMap::down(d, 1)))) return e;
        }
        if ((e = wgrad3(src(p.mid[d], p.pmid[d], c, 1, hh, ww, npc), none, B, c, hh, ww, i_down(d, 1)))) return e;
        if ((e = cine_conv3x3_dgrad(B, wd(i_down(d, 1), 0), wd2(i_down(d, 1)), sp, q.A, n, c, c, hh, ww, stream))) return e;
        B = next_g();
        if ((e = inbwd(p.mid[d], p.pmid[d], npc, c, hh, ww, q.A, c, 0, hh, ww, nullptr, 0, 0, B, DropMap::down(d, 0)))) return e;
        if (d > 0) {
            const int cp = p.ch[d - 1];
            if ((e = wgrad3(src(p.skip[d - 1], p.pskip[d - 1], cp, 2, p.hs[d - 1], p.wsz[d - 1], p.np_conv[d - 1]), none, B, c, hh, ww, i_down(d, 0)))) return e;
            if ((e = cine_conv3x3_dgrad(B, wd(i_down(d, 0), 0), wd2(i_down(d, 0)), sp, q.pool, n, c, cp, hh, ww, stream))) return e;
        } else {
            if ((e = wgrad3(src(x, nullptr, in_ch, 0, h, w, 0), none, B, c, hh, ww, i_down(0, 0)))) return e;
            if (gx && (e = cine_conv3x3_dgrad(B, wd(i_down(0, 0), 0), wd2(i_down(0, 0)), sp, gx, n, c, in_ch, hh, ww, stream))) return e;
        }
    }
    lane.join();
    return CINE_OK;
}
