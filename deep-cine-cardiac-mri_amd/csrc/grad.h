// grad.h -- launchers of the gradient kernels (grad_kernels.hip) used by the U-Net backward sequence (unet.hip).
#pragma once
#include "common.h"
#include "conv_src.h"

namespace cine {

// InstanceNorm + LeakyReLU backward of one raw tensor r (n, c, h, w) with statistics records `part`:
//   gr = d loss / d r   from   g = d loss / d act(r),   act = LeakyReLU((r - mean) * rstd)
// g is the sum of up to two pieces: `ga`, a (n, ca_total, ha, wa) tensor of which channels [ca_off, ca_off + c) and the
// top-left (h, w) window belong to this tensor (the input gradient of a conv over a channel concat / a zero-padded source,
// unet.py:106-122), and `gb`, the gradient of the 2x2 average pool of act(r) (n, c, hb, wb) (unet.py:97).
struct InBwdArgs {
    const float* r; const float* part; int np;
    const float* ga; int ca_total, ca_off, ha, wa;
    const float* gb; int hb, wb;
    float* gr;
    int n, c, h, w;
    float eps, slope;
};
int launch_in_lrelu_bwd(const InBwdArgs& a, hipStream_t st);

// Weight gradient of a convolution y = conv(X) whose input X is described like the forward's sources (modes 0 / 1 / 2,
// channel concat of two sources):  dW[row][ci][tap] += sum_{n, pixels} G[n][row][p] * X[n][ci][p + tap offset].
//   taps 9: 3x3 pad 1; taps 1: 1x1 (also the k2 s2 transpose conv, whose rows are the 4 sub-positions x cout of G's
//   space-to-depth view: g_mode 5, g (n, g_c, 2H, 2W), rows = 4 g_c).
// Samples [0, set_split) accumulate into grad0, the rest into grad1 (two networks in one launch).  grad layouts (natural,
// `+=`): kind 0 (rows, cin, 3, 3); kind 1 transpose conv (cin, rows / 4, 2, 2); kind 2 (rows, cin).
struct WgArgs {
    Src s0, s1;
    int cin;
    const float* g; int g_mode, g_c, g_h, g_w;
    int rows;
    int n, H, W;
    int set_split;
    float eps, slope;
};
size_t wgrad_ws_floats(int rows, int cin, int taps, int n);
int launch_wgrad(const WgArgs& a, int taps, int kind, float* grad0, float* grad1, float* ws, size_t ws_floats, hipStream_t st);

// gb[co] += sum_{n in set, pixels} g[n][co][p]   (bias of the final 1x1 conv, unet.py:69)
int launch_bias_grad(const float* g, int n, int cout, long hw, int set_split, float* gb0, float* gb1, float* ws, size_t ws_floats, hipStream_t st);   // ws: n * cout floats

// dgrad entry points of conv_kernels.hip
extern "C" int cine_conv3x3_dgrad(const float* gy, const float* wpacked, const float* wpacked2, int set_split,
                                  float* gx, int n, int cout, int cin, int h, int w, void* stream);
extern "C" int cine_tconv2x2_dgrad(const float* gy, const float* wpacked, const float* wpacked2, int set_split,
                                   float* gx, int n, int cin, int cout, int h, int w, void* stream);
extern "C" int cine_conv1x1_dgrad(const float* gy, const float* wpacked, const float* wpacked2, int set_split,
                                  float* gx, int n, int cout, int cin, int h, int w, void* stream);

}  // namespace cine
