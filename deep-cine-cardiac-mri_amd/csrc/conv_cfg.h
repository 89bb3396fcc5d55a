// conv_cfg.h -- launch arguments and tile geometry of the MFMA convolution kernels (conv_kernels.hip: every shape and source
// mode; conv_plane.hip: the lean kernel of the cfg-2 U-Net's plane-wide tiles).  Both kernels share one geometry, one weight
// packing and one statistics-record layout, so their outputs are interchangeable (and bit-identical).
#pragma once
#include "conv_src.h"

namespace cine {

struct ConvArgs {
    Src s0, s1;
    const float* wp0; const float* wp1; int set_split;   // samples >= set_split use wp1
    const float* bias; const float* bias1;   // bias1: samples >= set_split (two weight sets in one launch)
    const float* addend; int relu;       // epilogue: y = [relu](conv + bias + addend), addend shaped like y
    float* accum;                        // optional second output, shaped like y: accum += y (BCRNN: output_f + output_b, recurrent_varnet.py:254)
    float* y; float* ypart;
    int n, cin, rows, rowsp, H, W;       // GEMM rows (cout, or 4*cout / 8*cout for tconv), padded to 16
    int D, tiles_hw;                     // output depth (1 in 2-D); tiles per depth slice
    int vol;                             // 1: 3-D entry point (a depth-1 volume is still a volume: 8-way transpose conv)
    int tconv_cout;                      // > 0: transpose-conv store mapping with this many channels
    int add_src1;                        // 1: source 1 is ADDED to source 0 channel-wise (MWCNN skips, mwcnn.py:164,172)
                                         //    instead of concatenated
    float slope, eps;
    int tiles_w, tiles, nchunks, fast;
    int vfast;                           // volumes (TAPS 27): row-wise 16-byte staging of plain / normalised / 2x2x2-pooled sources
    int wav;                             // fast staging of a Haar DWT / IWT source (modes 3 / 4), optionally + an added plain / normalised skip
    int tvec;                            // transpose conv: paired 16-byte stores (W multiple of the lane's pixel run, aligned y)
    int ncc;                             // V3 kernels: channel chunks per depth offset (nchunks = 3 * ncc, chunk = (dz + 1) * ncc + cc)
    int accum_store;                     // 1: the second output is written (accum = y), not added to
    // pair launches (conv_mfma_pair_kernel, the two directions of a BCRNN time sweep in one grid): samples >= pair_n take
    // these pointers instead (and count from 0 again)
    int pair_n, accum_store_b;
    const float* x_b; const float* addend_b; float* y_b; float* accum_b;
    // gate (shaped like y): y = gate > 0 ? (conv + bias + addend) : 0 -- one step of back-propagation through a ReLU recurrence (the
    // adjoint of h_t = ReLU(conv(h_{t-1}) + P_t): g_{t-1} = [h_{t-1} > 0] (conv^T(g_t) + gout_{t-1})); gate_b: the second set of a pair launch
    const float* gate; const float* gate_b;
};

template <int CK, int CT, int WM, int WN, int MT, int TW, int TAPS>
struct ConvCfg {
    static constexpr int NT = 64 * WM * WN;
    static constexpr int HALO = TAPS == 1 ? 0 : 1;
    static constexpr int ZP = TAPS == 27 ? 3 : 1;  // input depth slices held in LDS
    static constexpr int RPF = 16 / TW;            // rows per fragment
    static constexpr int NF = WN * MT;             // fragments per workgroup
    static constexpr int TH = NF * RPF;            // tile rows
    static constexpr int ROWS = TH + 2 * HALO;
    // LDS row: [0, TW) interior, then (3x3 only) right halo at TW and left halo at COLS-1, i.e. image
    // column x lives at (x + COLS) % COLS.  COLS is a multiple of the staging piece so interior
    // pieces are 16-byte aligned (ds_write_b128).
    static constexpr int COLS = HALO ? (TW >= 4 ? TW + 4 : TW + 2) : TW;
    static constexpr int ZS = ROWS * COLS;         // stride between the depth slices of one channel
    static constexpr int PS = ((ZP * ROWS * COLS + 31) / 32) * 32 + 16;   // channel stride == 16 (mod 32)
    static constexpr int COT = 16 * CT * WM;
    static constexpr int COTP = (COT % 32 == 0) ? COT + 16 : COT;
    static constexpr int PW = TW >= 4 ? 4 : TW;    // floats per staging piece
    static constexpr int PR = TW / PW;             // pieces per row
    static constexpr int NPIECE = CK * ROWS * PR;
    static constexpr int NPT = (NPIECE + NT - 1) / NT;
    // staging map of the vectorised path: a thread owns one (row, piece) slot of the tile -- KR of them when
    // there are more slots than threads -- in channels g, g + G, ...: slot arithmetic happens once, the
    // pieces of a thread differ by compile-time strides
    static constexpr int RP = ROWS * PR;
    static constexpr int KR = (RP + NT - 1) / NT;
    static constexpr int gsel() { int g = 1; while (2 * g <= CK && 2 * g * RP <= NT) g *= 2; return g; }
    static constexpr int G = gsel();
    static constexpr int NCI = CK / G, NPF = KR * NCI;
    static constexpr int IN_FLOATS = CK * PS;
    static constexpr int W_FLOATS = TAPS * CK * COTP;
    static constexpr int RED_FLOATS = 3 * WN * COT;
    // waves per SIMD to ask the register allocator for (= workgroups per CU for 256-thread groups), from an
    // estimate of the live registers: accumulators + two operand groups + the chunk prefetched during the sweep
    static constexpr int NWT = (TAPS * CK * (COT / 4) + NT - 1) / NT;   // weight float4 per thread and chunk
    static constexpr int REGS = 4 * CT * MT + 2 * (CT + MT) + PW * NPF + 4 * NWT + 48;
    static constexpr int MINW = REGS <= 128 ? 4 : (REGS <= 168 ? 3 : 2);
    static_assert(RED_FLOATS <= IN_FLOATS, "reduction scratch must fit in the input tile");
    static size_t lds_bytes(int src_chans) { return (size_t)(IN_FLOATS + W_FLOATS + 2 * src_chans) * sizeof(float); }
};


template <int PW> struct Piece;
template <> struct Piece<4> { typedef float4 T; };
template <> struct Piece<2> { typedef float2 T; };

// conv_plane.hip: 3x3 convolutions whose tile spans the plane's width (W == TW: the x-f / y-f planes of the cascade U-Nets).
// Called by the general dispatcher with the tile configuration it chose; returns with *handled = true when the layer was launched
// there, *handled = false (and CINE_OK) when it is not one of that kernel's shapes.
int launch_conv_plane(const ConvArgs& a, int ck, int ct, int wm, int wn, int mt, int tw, hipStream_t st, bool* handled);
// Set (per calling thread, for the duration of the call) by entry points whose ARGUMENTS say the launch sequence runs beside nothing but
// its own branches (cine_unet2d_forward_branches with side streams: one slice alone on the chip).  Layers whose one-workgroup-per-plane
// grid under-fills the chip then split their output rows over two workgroups; the pixel tiling, the statistics records and every output
// bit stay the same, and with many slices in flight (one stream each, no side streams) the unsplit grids remain the faster ones.
extern thread_local int g_alone_on_chip;
struct AloneScope { int prev; explicit AloneScope(bool on) : prev(g_alone_on_chip) { if (on) g_alone_on_chip = 1; } ~AloneScope() { g_alone_on_chip = prev; } };

// the k2 s2 transpose conv of plane-wide tiles (pixel tile = mt fragments of 16 pixels, plane width tw)
int launch_tconv_plane(const ConvArgs& a, int mt, int tw, hipStream_t st, bool* handled);

// input gradient of the k2 s2 transpose conv (1x1 GEMM over the space-to-depth view of the output gradient, source mode 5)
int launch_tconv_dgrad_plane(const ConvArgs& a, int mt, int tw, hipStream_t st, bool* handled);
// 3x3 convolutions over 16-wide column tiles of wider planes (v3 = 0) / volumes in the three-pass form (v3 = 1)
int launch_conv_wide(const ConvArgs& a, int ct, int wm, int wn, int mt, int v3, hipStream_t st, bool* handled);

// conv_coarse.hip: 3x3(x3) convolutions of small volumes / planes with >= 48 output rows (the coarse levels of the 3-D U-Net, the 26 x 26
// level of the sensitivity network) -- flattened
// positions, K split over the waves of a workgroup.  coarse_tiles() = its statistics records per (sample, channel).
int launch_conv_coarse(const ConvArgs& a, hipStream_t st, bool* handled);
int coarse_tiles(int rowsp, int d, int h, int w, bool vol);

}  // namespace cine
