// grad.h -- launchers of the gradient kernels (grad_kernels.hip) used by the U-Net backward sequence (unet.hip).
#pragma once
#include "common.h"
#include "conv_src.h"

namespace cine {

// InstanceNorm + LeakyReLU backward of one raw tensor r (n, c, h, w) with statistics records `part`:
//   gr = d loss / d r   from   g = d loss / d act(r),   act = LeakyReLU((r - mean) * rstd)
// g is the sum of up to two pieces, each the input gradient of one consumer of act(r), gathered on load:
//   type 1  window : a (n, c_total, gh, gw) tensor of which channels [c_off, c_off + c) and the top-left (h, w) window belong to this
//                    tensor (a conv over a channel concat / a zero-padded source, unet.py:106-122; an added skip, mwcnn.py:164,172)
//   type 2  pool   : the consumer read the 2x2 average pool of act(r) (unet.py:97): g (n, c, gh, gw), 0.25 g[y/2][x/2]
//   type 3  DWT    : the consumer read the Haar DWT of act(r) (mwcnn.py:224-236): g (n, c_total >= 4 c, h/2, w/2), bands [LL, HL, LH, HH]
//                    of channel ch at c_off + band * c + ch -- the adjoint is the inverse transform of the four band gradients
//   type 4  IWT    : the consumer read the Haar IWT of act(r) (mwcnn.py:252-261): g (n, c_total >= c / 4, 2 h, 2 w); channel ch = k c/4 + cc
//                    collects the 2x2 block of output channel c_off + cc with the signs of sub-band k
//   Volumes (the 3-D U-Net, unet.py with dims = 3) pass their tensors as planes of (d h, w); a piece whose in-plane extents equal the tensor's is a
//   type 1 window of (gd gh, gw), and
//   type 5  window3: a (n, c_total, gd, gh, gw) tensor whose front-top-left (d, vh, w) window belongs to this tensor (vh = the tensor's own height)
//   type 6  pool3  : the consumer read the 2x2x2 average pool of act(r) (unet.py:88,97): g (n, c, gd, gh, gw), 0.125 g[z/2][y/2][x/2]
struct GradPiece { const float* g; int type, c_total, c_off, gh, gw, gd, vh; };
struct InBwdArgs {
    const float* r; const float* part; int np;
    GradPiece a, b;
    float* gr;
    int n, c, h, w;
    float eps, slope;
    // Dropout2d behind the activation (unet.py:163,167): drop[n * c + ch] = the channel's multiplier d (0 or 1 / (1 - p)), NULL = none.  The forward
    // folds d into the statistics records (d LeakyReLU(v) = LeakyReLU(d v): the records then give rstd' = d rstd, cine_dropout_stats), so every
    // consumer sees z = d xhat; the backward of the normalisation needs d once more: gr = rstd' (g' - mean g' - z mean(g' z) / d^2)
    const float* drop;
};
__device__ __forceinline__ float drop_k2(const float* drop, long plane) {
    if (!drop) return 1.f;
    const float d = drop[plane];
    return d > 0.f ? 1.f / (d * d) : 0.f;
}
int launch_in_lrelu_bwd(const InBwdArgs& a, hipStream_t st);
// Dropout2d / Dropout3d multipliers (cine_unet2d_forward_branches' `drop`): fold drop[plane] into the statistics records of `planes` (sample, channel)
// planes with np records each (unet.hip)
int launch_dropout_stats(float* part, int np, long planes, const float* drop, hipStream_t st);
// layout of the multipliers: one (n, ch) block per 3x3 conv in launch order -- down path / bottleneck level d: convs 2 d and 2 d + 1 (ch[d] channels),
// up path level d: convs 2 (P + 1) + 2 (P - 1 - d) and the next one (ch[d] channels); the transpose convs have no dropout (unet.py:204-219)
struct DropMap {
    const float* base; int n_total, a, chans, pools;
    long off(int conv) const {         // floats in front of conv's block
        long o = 0;
        for (int j = 0; j < conv; ++j) o += (long)n_total * ch_of(j);
        return o;
    }
    int ch_of(int conv) const {
        const int P = pools;
        const int d = conv < 2 * (P + 1) ? conv / 2 : P - 1 - (conv - 2 * (P + 1)) / 2;
        return chans << d;
    }
    const float* at(int conv) const { return base ? base + off(conv) + (long)a * ch_of(conv) : nullptr; }
    static int down(int d, int k) { return 2 * d + k; }
    int up(int d, int k) const { return 2 * (pools + 1) + 2 * (pools - 1 - d) + k; }
};

// few, large planes (volumes; the sens-net's 200 x 200 coil planes): two passes over chunks of a plane with `ws` holding the chunk sums;
// falls back to launch_in_lrelu_bwd for every other shape or without a workspace
size_t in_lrelu_bwd_ws_floats(int n, int c, int h, int w);
int launch_in_lrelu_bwd_split(const InBwdArgs& a, float* ws, size_t ws_floats, hipStream_t st);
void set_wgrad_plane(int on);        // diagnostics: 0 routes the plane-wide weight gradients through the general kernel (cine_set_conv_plane bit 4)
int launch_in_lrelu_bwd_fast(const InBwdArgs& a, hipStream_t st, bool* handled);      // inbwd_fast.hip: the U-Nets' plane shapes, one pass over HBM

// Weight gradient of a convolution y = conv(X) whose input X is described like the forward's sources (modes 0 / 1 / 2 vectorised; the
// Haar modes 3 / 4 and added sources element by element; channel concat or sum of two sources):  dW[row][ci][tap] += sum_{n, pixels} G[n][row][p] * X[n][ci][p + tap offset].
//   taps 9: 3x3 pad 1; taps 1: 1x1 (also the k2 s2 transpose conv, whose rows are the 4 sub-positions x cout of G's
//   space-to-depth view: g_mode 5, g (n, g_c, 2H, 2W), rows = 4 g_c).
// Samples [0, set_split) accumulate into grad0, the rest into grad1 (two networks in one launch).  grad layouts (natural,
// `+=`): kind 0 (rows, cin, 3, 3); kind 1 transpose conv (cin, rows / 4, 2, 2) -- in general (cin, rows); kind 2 (rows, cin).
struct WgArgs {
    Src s0, s1;
    int add_src1;          // 1: source 1 is ADDED to source 0 channel-wise (the MWCNN skips) instead of concatenated
    int cin;
    const float* g; int g_mode, g_c, g_h, g_w;
    int rows;
    int n, H, W;
    int set_split;
    float eps, slope;
    float* mat; size_t mat_floats;   // optional scratch (n, cin, H, W): sources the vectorised staging cannot read (wavelet modes, summed or pooled
                                     // sources) are materialised there first and the launch then reads them as one plain tensor
};
size_t wgrad_ws_floats(int rows, int cin, int taps, int n);
int launch_wgrad(const WgArgs& a, int taps, int kind, float* grad0, float* grad1, float* ws, size_t ws_floats, hipStream_t st);
// grad (rows, cin, 3, 3, 3) += the weight gradient of a 3x3x3 conv as its three depth taps: a[kz] = the one-set 3x3 problem of tap kz over the
// depth-shifted slice pairs (x[z + kz - 1], g[z]), depth slices as samples (a[kz].n == 0: a dead tap); ws: 3 x wgrad_ws_floats(rows, cin, 9, .)
int launch_wgrad27(const WgArgs (&a)[3], float* grad, float* ws, size_t ws_floats, hipStream_t st);

// gb[co] += sum_{n in set, pixels} g[n][co][p]   (bias of the final 1x1 conv, unet.py:69)
int launch_bias_grad(const float* g, int n, int cout, long hw, int set_split, float* gb0, float* gb1, float* ws, size_t ws_floats, hipStream_t st);   // ws: n * cout floats

// Weight gradients beside the input-gradient chain.  Per conv layer the backward pass is in_lrelu_bwd (writes g) -> {wgrad(g), dgrad(g)}:
// the two consumers are independent, and a backward pass on ONE stream pays every launch's half-empty last round.  SideLane
// puts the weight gradients on a second stream supplied by the caller (cine_set_side_stream; none: everything on the caller's stream): fork() after the producer of g has been enqueued, launched() after the wgrad;
// g alternates between two buffers and before_write(slot) makes the main stream wait for the wgrad that still reads that slot; join()
// before the call returns.  The weight gradients stay serial among themselves (they share the partial-sum workspace), and nothing about
// the results depends on timing.
void set_side_stream(hipStream_t s);      // thread-local
class SideLane {
public:
    explicit SideLane(hipStream_t main);
    ~SideLane();
    SideLane(const SideLane&) = delete;
    SideLane& operator=(const SideLane&) = delete;
    int slot() const { return k_ & 1; }
    void before_write();            // main: wait for the wgrad that read buffer slot()
    hipStream_t fork();             // main: record "g ready"; side: wait for it.  Returns the stream for the wgrad
    void launched();                // side: record "wgrad done" for slot(); advances to the other slot
    void join();                    // main: wait for everything on the side lane
private:
    hipStream_t main_, side_ = nullptr;
    hipEvent_t ready_ = nullptr, done_[2] = {nullptr, nullptr};
    bool rec_[2] = {false, false};
    int k_ = 0;
};

// dgrad entry points of conv_kernels.hip
extern "C" int cine_conv3x3_dgrad(const float* gy, const float* wpacked, const float* wpacked2, int set_split,
                                  float* gx, int n, int cout, int cin, int h, int w, void* stream);
extern "C" int cine_tconv2x2_dgrad(const float* gy, const float* wpacked, const float* wpacked2, int set_split,
                                   float* gx, int n, int cin, int cout, int h, int w, void* stream);
extern "C" int cine_conv1x1_dgrad(const float* gy, const float* wpacked, const float* wpacked2, int set_split,
                                  float* gx, int n, int cout, int cin, int h, int w, void* stream);

}  // namespace cine
