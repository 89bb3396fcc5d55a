// conv_plane.hip -- the lean 3x3 convolution of the cascade U-Nets (reference denoisers/unet.py:159-168 on x-f / y-f planes).
//
// conv_kernels.hip's conv_tile serves every shape, source mode and epilogue of the repository from one body: 13 k instructions
// (100 KB of code per instantiation), run-time widths in every address, wavelet / volume / ragged-edge branches the U-Nets never
// take.  Its profile on the cfg-2 layers is 2.1 - 3.0 vector instructions per fp32 MFMA -- and the fp32 MFMA shares the SIMD's
// vector issue, so those instructions are paid in matrix time.  This kernel is the same algorithm for the one shape class that
// carries 95 % of cfg 2's FLOPs, with everything that can be a compile-time constant made one:
//   * the tile spans the plane's width (W == TW in {16, 8, 4, 2}): no halo columns to fetch, a tile's rows are ONE contiguous
//     run of memory per channel, every store offset of the epilogue is an instruction immediate;
//   * sources: one plain tensor (first layer, <= 4 channels), one or two InstanceNorm + LeakyReLU-on-load tensors of the plane's
//     own extent (conv 2 of a block, the up path's concat), or one 2x2-average-pooled tensor (first conv below a pool);
//   * rows outside the image and the halo columns are zeroed ONCE (the thread -> slot map is the same for every chunk), so the
//     staging phase has no per-element selects;
//   * wave-level sums of the statistics epilogue use gfx950's v_permlane16/32_swap instead of ds_bpermute round trips.
// Same tile geometry (ConvCfg), weight packing, accumulation order and statistics arithmetic as conv_tile: the outputs and the
// {count, mean, M2} records are BIT-IDENTICAL to the general kernel's (tests/test_hip_parity.py::test_conv_plane_bit_identical),
// so producers and consumers of either kind mix freely.
#include <atomic>
#include <mutex>
#include <type_traits>
#include "common.h"
#include "conv_cfg.h"
#include "grad.h"

namespace cine {
namespace {

thread_local int g_plane_on = 7;          // bit 0: plane-wide 3x3 convs, bit 1: transpose convs, bit 2: wide planes / volumes

struct PlaneArgs {
    const float* x0; const float* part0; int c0, np0;
    const float* x1; const float* part1; int c1, np1;
    long cs0;                            // MODE 2: floats between the channels of the pooled source
    int sh0;                             // MODE 2: source height (2H or 2H + 1); its row pitch is 2 TW
    const float* wp0; const float* wp1; int set_split;
    const float* bias; const float* bias1; int relu;      // optional epilogue y = [ReLU](conv + bias) (the MWCNN's conv blocks, mwcnn.py:60-75)
    float* y; float* ypart;
    int cin, rows, rowsp, H, nchunks, tiles;
    float slope, eps;
};

// x + (x of lane ^ 16) and x + (x of lane ^ 32): the swaps hand every lane both halves, the add is commutative -> the same
// bits as x + __shfl_xor(x, 16 / 32)
__device__ __forceinline__ float add_xor16(float x) {
    const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(x), __float_as_uint(x), false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);          // {own, partner} or {partner, own}: IEEE addition commutes
}
__device__ __forceinline__ float add_xor32(float x) {
    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}

// MODE 0: plain source (no statistics); 1: InstanceNorm + LeakyReLU on load, one source or the concat of two; 2: the same + 2x2 average pool;
// 3: the same + Haar DWT (mwcnn.py:224-236): conv input channel band * c0 + c = that band of source channel c, whole 8-channel chunks of one band;
// 4: Haar IWT of source 0 (4 cin channels at half the extent, mwcnn.py:252-261) PLUS source 1 (cin channels of the plane's extent), both
//    InstanceNorm + LeakyReLU on load (the additive skips of mwcnn.py:164,172): source 1 rides in the register prefetch, the IWT inputs are
//    read when the chunk is committed
template <int CK, int CT, int WM, int WN, int MT, int TW, int MODE>
__global__ __launch_bounds__(64 * WM * WN, (ConvCfg<CK, CT, WM, WN, MT, TW, 9>::MINW)) void conv_plane_kernel(PlaneArgs a) {
    using C = ConvCfg<CK, CT, WM, WN, MT, TW, 9>;
    constexpr int PW = C::PW, NT = C::NT, PR = C::PR, RP = C::RP, G = C::G, NCI = C::NCI, NWT = C::NWT;
    static_assert(C::KR == 1, "one (row, piece) slot per thread");
    typedef typename Piece<PW>::T piece_t;
    extern __shared__ __align__(16) float smem_f[];
    float* in_lds = smem_f;
    float* w_lds = smem_f + C::IN_FLOATS;
    float* st_lds = w_lds + C::W_FLOATS;            // {scale, shift} per input channel

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int tile = blockIdx.x, n = blockIdx.z;
    const int r0 = tile * C::TH, co0 = blockIdx.y * C::COT;
    const float* wp = n >= a.set_split ? a.wp1 : a.wp0;
    const int q = lane & 15, kk = lane >> 4;
    const int qr = q / TW, qc = q % TW;
    int base_in[3];
#pragma unroll
    for (int dx = 0; dx < 3; ++dx)
        base_in[dx] = kk * C::PS + (wn * MT * C::RPF + qr) * C::COLS + (qc + dx - 1 + C::COLS) % C::COLS;
    const int base_w = kk * C::COTP + 16 * (wm * CT) + q;

    __builtin_amdgcn_s_setprio(2);
    CINE_STAMP_RT(9);
    CINE_STAMP(0);
    // ---- my slot of the staging map: (row, piece) srp of channel group sg; fixed for the whole kernel
    const int sg = tid / RP, srp = tid - sg * RP;
    const bool slot = sg < G;
    const int sgc = G == 1 ? 0 : min(sg, G - 1);
    const int srow = srp / PR, sj = srp % PR;
    const int gy = r0 - 1 + srow;                                   // image row of my slot
    const bool rowok = gy >= 0 && gy < a.H;
    // modes 0 / 1: both sources have the plane's extent, so one channel stride; my byte offset inside a chunk's first channel
    const unsigned cstride = (unsigned)(a.H * TW) * 4u;
    const unsigned voff = (unsigned)(min(max(gy, 0), a.H - 1) * TW + PW * sj) * 4u + (unsigned)sgc * cstride;
    float* const lrow = in_lds + sgc * C::PS + srow * C::COLS + PW * sj;

    float4 wraw[NWT];
    constexpr bool HALF = MODE == 2 || MODE == 3;          // the source has twice the plane's extent: 2 x 2 input values per staged value
    // MODE 4 (Haar IWT of the scale below + additive skip): a slot is a PAIR of image rows (2 sy, 2 sy + 1) x one piece -- both rows come from the
    // same source row sy of the four bands, which round 4's (row, piece) slots read and activated once per row.  RPAIR pair slots cover the tile's
    // ROWS rows (the first / last one only half: the tile starts on an odd image row); G4 channel groups, NCI4 channels per thread.
    constexpr int RPAIR = C::ROWS / 2 + 1, SLOTS4 = RPAIR * PR;
    constexpr int G4 = SLOTS4 * 8 <= NT ? 8 : SLOTS4 * 4 <= NT ? 4 : SLOTS4 * 2 <= NT ? 2 : 1, NCI4 = CK / G4;
    static_assert(MODE != 4 || (C::ROWS % 2 == 0 && SLOTS4 <= NT && PW == 4), "MODE 4: even tile rows, one slot per thread, 16-byte pieces");
    const int sg4 = tid / SLOTS4, sp4 = tid - sg4 * SLOTS4;
    const bool slot4 = sg4 < G4;
    const int sg4c = min(sg4, G4 - 1);
    const int ps4 = sp4 / PR, sj4 = sp4 % PR;
    const int sy4 = (r0 >> 1) - 1 + ps4;                    // source row; image rows 2 sy4, 2 sy4 + 1 = LDS rows 2 ps4 - 1, 2 ps4 (r0 is even)
    const bool srcok4 = slot4 && sy4 >= 0 && 2 * sy4 < a.H;
    piece_t xraw[HALF ? 1 : (MODE == 4 ? 2 * NCI4 : NCI)];
    // pooled source: the first NPRE pieces of a chunk are prefetched like the plain ones (2 PW floats of two source rows each)
    // MODE 3 (Haar DWT on load): a chunk = the FOUR BANDS of TWO source channels, LDS channel ck = 2 band + s -- the weights of conv channel
    // band * C + 2 chunk + s are gathered to match -- so a thread reads each 2 x 2 PW source block once and activates it once for all the
    // bands it stages (round 4: a chunk = 8 source channels of one band, every block read and activated in four chunks; 540 vector
    // instructions per MFMA against 300 for a plain source).  NB3 = source blocks per thread and chunk, all prefetched.
    constexpr int NB3 = G == 1 ? 2 : 1;
    constexpr int NPRE = MODE == 3 ? NB3 : (HALF ? (NCI < 2 ? NCI : 2) : 0);
    float4 praw[NPRE > 0 ? NPRE : 1][PW == 4 ? 4 : 2];
    const float* const pbase = HALF ? a.x0 + ((long)n * a.c0 + (MODE == 3 ? 0 : sgc)) * a.cs0 + (long)(2 * min(max(gy, 0), a.H - 1)) * (2 * TW) + 2 * PW * sj : nullptr;
    auto issue = [&](int chunk) {
        const float* wsrc = wp + (long)chunk * 9 * CK * a.rowsp;
#pragma unroll
        for (int i = 0; i < NWT; ++i) {
            const int e = tid + i * NT;
            int row = e / (C::COT / 4);
            const int c4 = (e % (C::COT / 4)) * 4;
            const bool v = e < 9 * CK * (C::COT / 4) && co0 + c4 < a.rowsp;
            if (CK == 4) row = (row / CK) * 8 + row % CK;           // 4-channel chunk over the 8-channel packing
            if constexpr (MODE == 3) {      // LDS row (tap, ck) <- packed row of conv channel (ck >> 1) C + 2 chunk + (ck & 1)
                const int tap = row >> 3, ckk = row & 7;
                const int ci = (ckk >> 1) * a.c0 + 2 * chunk + (ckk & 1);
                wraw[i] = *reinterpret_cast<const float4*>(v ? wp + (long)(((ci >> 3) * 9 + tap) * 8 + (ci & 7)) * a.rowsp + co0 + c4 : wp);
            } else
            wraw[i] = *reinterpret_cast<const float4*>(v ? wsrc + (long)row * a.rowsp + co0 + c4 : wp);
        }
        if constexpr (MODE == 4) {      // the skip's pieces of my two rows (source 1, conv channel = its channel)
            const char* sb = reinterpret_cast<const char*>(a.x1) + ((size_t)n * a.c1 + chunk * CK + sg4c) * cstride;
            const int g0 = min(max(2 * sy4, 0), a.H - 2);
#pragma unroll
            for (int r = 0; r < 2; ++r)
#pragma unroll
                for (int i = 0; i < NCI4; ++i)
                    xraw[r * NCI4 + i] = *reinterpret_cast<const piece_t*>(sb + (size_t)(i * G4) * cstride + (unsigned)((g0 + r) * TW + PW * sj4) * 4u);
        } else if constexpr (!HALF) {
            const int ci0 = chunk * CK;
            const bool first = ci0 < a.c0;
            const int cl0 = first ? ci0 : ci0 - a.c0;
            const int sc = first ? a.c0 : a.c1;
            const char* sb = reinterpret_cast<const char*>(first ? a.x0 : a.x1) + ((size_t)n * sc + cl0) * cstride;   // uniform
            const int cmax = sc - 1 - cl0;
#pragma unroll
            for (int i = 0; i < NCI; ++i) {
                const int cku = G == 1 ? min(i, cmax) : i * G;       // uniform part of the channel (G > 1: whole chunks only, host check)
                xraw[i] = *reinterpret_cast<const piece_t*>(sb + (size_t)cku * cstride + voff);
            }
        } else {
#pragma unroll
            for (int i = 0; i < NPRE; ++i) {
                const float* src = pbase + (long)(MODE == 3 ? 2 * chunk + (G == 1 ? i : (sgc & 1)) : chunk * CK + i * G) * a.cs0;
                if constexpr (PW == 4) {
                    praw[i][0] = *reinterpret_cast<const float4*>(src); praw[i][1] = *reinterpret_cast<const float4*>(src + 4);
                    praw[i][2] = *reinterpret_cast<const float4*>(src + 2 * TW); praw[i][3] = *reinterpret_cast<const float4*>(src + 2 * TW + 4);
                } else {
                    praw[i][0] = *reinterpret_cast<const float4*>(src); praw[i][1] = *reinterpret_cast<const float4*>(src + 2 * TW);
                }
            }
        }
    };

    // ---- statistics records of the input channels first (their latency is not paid behind the first chunk's 16-byte loads)
    const int nch = a.c0 + a.c1;
    constexpr int NPQ = 16;
    float prec[MODE == 0 ? 1 : 3 * NPQ];
    const bool pfirst = tid < a.c0;
    const int pnp = pfirst ? a.np0 : a.np1;
    const int npm = max(a.np0, a.c1 > 0 ? a.np1 : 0);
    if constexpr (MODE != 0) {
        if (tid < nch) {
            const float* pp = (pfirst ? a.part0 + ((long)n * a.c0 + tid) * pnp * 3 : a.part1 + ((long)n * a.c1 + (tid - a.c0)) * pnp * 3);
            if (npm <= 4) {
#pragma unroll
                for (int i = 0; i < 12; ++i) prec[i] = pp[min(i, 3 * pnp - 1)];
            } else load_partials<NPQ>(pp, pnp, prec);
        }
    }
    issue(0);
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (MODE != 0) {
        if (tid < nch) {
            float2 mr;
            if (npm <= 4) { float r4[12];
#pragma unroll
                for (int i = 0; i < 12; ++i) r4[i] = prec[i];
                mr = merge_loaded<4>(r4, pnp, a.eps);
            } else mr = merge_loaded<NPQ>(prec, pnp, a.eps);
            st_lds[2 * tid] = mr.y; st_lds[2 * tid + 1] = -mr.x * mr.y;
        }
    }
    // ---- what no chunk ever writes is zeroed once: the halo columns, the rows outside the image, and (a chunk wider than the
    // layer's input, i.e. the 2-channel first layer) the channels that do not exist
    if (a.cin % CK != 0) {
        for (int e = tid; e < C::IN_FLOATS; e += NT) in_lds[e] = 0.f;
    } else {
        for (int e = tid; e < CK * C::ROWS * 2; e += NT) {
            const int ck = e / (C::ROWS * 2), rem = e % (C::ROWS * 2);
            in_lds[ck * C::PS + (rem >> 1) * C::COLS + ((rem & 1) ? C::COLS - 1 : TW)] = 0.f;
        }
        if (slot && !rowok) {
            piece_t z;
            float* zf = reinterpret_cast<float*>(&z);
#pragma unroll
            for (int u = 0; u < PW; ++u) zf[u] = 0.f;
#pragma unroll
            for (int i = 0; i < NCI; ++i) *reinterpret_cast<piece_t*>(lrow + i * G * C::PS) = z;
        }
    }

    f32x4 acc[CT][MT];
#pragma unroll
    for (int ct = 0; ct < CT; ++ct)
#pragma unroll
        for (int f = 0; f < MT; ++f) acc[ct][f] = (f32x4){0.f, 0.f, 0.f, 0.f};

    CINE_STAMP(1);
    for (int chunk = 0; chunk < a.nchunks; ++chunk) {
        __syncthreads();                              // stats table + zero fill ready / previous sweep done with LDS
        if (chunk == 0) CINE_STAMP(2);
#pragma unroll
        for (int i = 0; i < NWT; ++i) {
            const int e = tid + i * NT;
            if (e >= 9 * CK * (C::COT / 4)) break;
            const int row = e / (C::COT / 4), c4 = (e % (C::COT / 4)) * 4;
            *reinterpret_cast<float4*>(w_lds + row * C::COTP + c4) = co0 + c4 < a.rowsp ? wraw[i] : make_float4(0.f, 0.f, 0.f, 0.f);
        }
        const int ci0 = chunk * CK;
        if constexpr (MODE == 4) {
            if (srcok4) {
                constexpr int NS = PW / 2;
                const int cq = a.c0 >> 2, hs = a.H >> 1;
                const float* ib = a.x0 + (((long)n * a.c0 + ci0 + sg4c) * hs + sy4) * (TW / 2) + ((PW * sj4) >> 1);
                const float* stp = st_lds + 2 * (a.c0 + ci0 + sg4c);
                const bool w0 = ps4 >= 1, w1 = 2 * ps4 < C::ROWS;                 // which of my two LDS rows exist in this tile
                float* const l0 = in_lds + sg4c * C::PS + (2 * ps4 - 1) * C::COLS + PW * sj4;
#pragma unroll
                for (int i = 0; i < NCI4; ++i) {
                    float v[4][NS];
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const int c = ci0 + sg4c + i * G4 + k * cq;
                        const float2 t2 = *reinterpret_cast<const float2*>(ib + (long)(i * G4 + k * cq) * hs * (TW / 2));
                        const float2 sk = *reinterpret_cast<const float2*>(st_lds + 2 * c);
                        v[k][0] = 0.5f * act(t2.x, sk.x, sk.y, a.slope); v[k][1] = 0.5f * act(t2.y, sk.x, sk.y, a.slope);
                    }
                    const float2 ss = *reinterpret_cast<const float2*>(stp + 2 * i * G4);
                    piece_t o0 = xraw[i], o1 = xraw[NCI4 + i];
                    float* a0 = reinterpret_cast<float*>(&o0);
                    float* a1 = reinterpret_cast<float*>(&o1);
#pragma unroll
                    for (int e = 0; e < NS; ++e) {          // conv_tile's operation order per output: the IWT value, then += the activated skip (mwcnn.py:252-261)
                        a0[2 * e] = (v[0][e] - v[1][e] - v[2][e] + v[3][e]) + act(a0[2 * e], ss.x, ss.y, a.slope);
                        a0[2 * e + 1] = (v[0][e] + v[1][e] - v[2][e] - v[3][e]) + act(a0[2 * e + 1], ss.x, ss.y, a.slope);
                        a1[2 * e] = (v[0][e] - v[1][e] + v[2][e] - v[3][e]) + act(a1[2 * e], ss.x, ss.y, a.slope);
                        a1[2 * e + 1] = (v[0][e] + v[1][e] + v[2][e] + v[3][e]) + act(a1[2 * e + 1], ss.x, ss.y, a.slope);
                    }
                    if (w0) *reinterpret_cast<piece_t*>(l0 + i * G4 * C::PS) = o0;
                    if (w1) *reinterpret_cast<piece_t*>(l0 + C::COLS + i * G4 * C::PS) = o1;
                }
            }
        } else if constexpr (!HALF) {
            if (slot && rowok) {
                const float* stp = st_lds + 2 * (ci0 + sgc);
#pragma unroll
                for (int i = 0; i < NCI; ++i) {
                    piece_t o = xraw[i];
                    if (G == 1 && ci0 + i >= a.cin) {               // uniform: a chunk wider than what is left of the layer's input (2-channel first
                        float* zf = reinterpret_cast<float*>(&o);   // layer; the MWCNN's 10 = 8 + 2 channels) -- zeros over the previous chunk's values
#pragma unroll
                        for (int u = 0; u < PW; ++u) zf[u] = 0.f;
                    } else if constexpr (MODE == 1) {
                        const float2 ss = *reinterpret_cast<const float2*>(stp + 2 * i * G);
                        act_piece<PW>(reinterpret_cast<float*>(&o), ss.x, ss.y, a.slope);
                    }
                    *reinterpret_cast<piece_t*>(lrow + i * G * C::PS) = o;
                }
            }
        } else {
            // pooled source (extent sh0 x 2 TW): 2 PW floats from each of two rows per piece; pieces >= NPRE are loaded here
            if constexpr (MODE == 3) {
                if (slot && rowok) {
#pragma unroll
                    for (int b = 0; b < NB3; ++b) {
                        const int sch = 2 * chunk + (G == 1 ? b : (sgc & 1));       // this block's source channel
                        const float2 ss = *reinterpret_cast<const float2*>(st_lds + 2 * sch);
                        float t0[2 * PW], t1[2 * PW];
                        if constexpr (PW == 4) {
                            *reinterpret_cast<float4*>(t0) = praw[b][0]; *reinterpret_cast<float4*>(t0 + 4) = praw[b][1];
                            *reinterpret_cast<float4*>(t1) = praw[b][2]; *reinterpret_cast<float4*>(t1 + 4) = praw[b][3];
                        } else {
                            *reinterpret_cast<float4*>(t0) = praw[b][0]; *reinterpret_cast<float4*>(t1) = praw[b][1];
                        }
                        float x1[PW], x2[PW], x3[PW], x4[PW];                       // mwcnn.py:224-236: x1 even/even, x2 odd row, x3 odd column, x4 odd/odd, halved
#pragma unroll
                        for (int u = 0; u < PW; ++u) {
                            x1[u] = 0.5f * act(t0[2 * u], ss.x, ss.y, a.slope); x3[u] = 0.5f * act(t0[2 * u + 1], ss.x, ss.y, a.slope);
                            x2[u] = 0.5f * act(t1[2 * u], ss.x, ss.y, a.slope); x4[u] = 0.5f * act(t1[2 * u + 1], ss.x, ss.y, a.slope);
                        }
#pragma unroll
                        for (int i = 0; i < NCI; ++i) {
                            if (G == 1 && (i & 1) != b) continue;                   // G == 1: LDS channel i = 2 band + s
                            const int band = (sgc + i * G) >> 1;                    // LL, HL, LH, HH: (+ + + +), (- - + +), (- + - +), (+ - - +)
                            const float s1 = (band == 0 || band == 3) ? 1.f : -1.f, s2 = (band == 0 || band == 2) ? 1.f : -1.f,
                                        s3 = (band == 0 || band == 1) ? 1.f : -1.f;
                            piece_t o;
                            float* ov = reinterpret_cast<float*>(&o);
#pragma unroll
                            for (int u = 0; u < PW; ++u) ov[u] = fmaf(s3, x3[u], fmaf(s2, x2[u], s1 * x1[u])) + x4[u];      // ((+-x1 +- x2) +- x3) + x4: fetch_scalar's order
                            *reinterpret_cast<piece_t*>(lrow + i * G * C::PS) = o;
                        }
                    }
                }
            } else
            if (slot && rowok && 2 * gy + 1 < a.sh0) {
                const int cs = ci0;                                   // first source channel of the chunk (uniform)
                constexpr int band = 0;
                const float* sb = pbase + (long)cs * a.cs0;
                const float* stp = st_lds + 2 * (cs + sgc);
                auto pooled = [&](const float* t0, const float* t1, int i) {
                    const float2 ss = *reinterpret_cast<const float2*>(stp + 2 * i * G);
                    piece_t o;
                    float* ov = reinterpret_cast<float*>(&o);
#pragma unroll
                    for (int u = 0; u < PW; ++u) {
                        if constexpr (MODE == 3) {                    // conv_src.h fetch_scalar's operation order (bit-identical)
                            const float x1 = 0.5f * act(t0[2 * u], ss.x, ss.y, a.slope), x3 = 0.5f * act(t0[2 * u + 1], ss.x, ss.y, a.slope);
                            const float x2 = 0.5f * act(t1[2 * u], ss.x, ss.y, a.slope), x4 = 0.5f * act(t1[2 * u + 1], ss.x, ss.y, a.slope);
                            ov[u] = band == 0 ? x1 + x2 + x3 + x4 : band == 1 ? -x1 - x2 + x3 + x4 : band == 2 ? -x1 + x2 - x3 + x4 : x1 - x2 - x3 + x4;
                        } else {
                            ov[u] = 0.25f * (act(t0[2 * u], ss.x, ss.y, a.slope) + act(t0[2 * u + 1], ss.x, ss.y, a.slope) +
                                             act(t1[2 * u], ss.x, ss.y, a.slope) + act(t1[2 * u + 1], ss.x, ss.y, a.slope));
                        }
                    }
                    *reinterpret_cast<piece_t*>(lrow + i * G * C::PS) = o;
                };
#pragma unroll
                for (int i = 0; i < NPRE; ++i) {
                    float t0[2 * PW], t1[2 * PW];
                    if constexpr (PW == 4) {
                        *reinterpret_cast<float4*>(t0) = praw[i][0]; *reinterpret_cast<float4*>(t0 + 4) = praw[i][1];
                        *reinterpret_cast<float4*>(t1) = praw[i][2]; *reinterpret_cast<float4*>(t1 + 4) = praw[i][3];
                    } else {
                        *reinterpret_cast<float4*>(t0) = praw[i][0]; *reinterpret_cast<float4*>(t1) = praw[i][1];
                    }
                    pooled(t0, t1, i);
                }
#pragma unroll 2
                for (int i = NPRE; i < NCI; ++i) {
                    const float* src = sb + (long)i * G * a.cs0;
                    float t0[2 * PW], t1[2 * PW];
                    if constexpr (PW == 4) {
#pragma unroll
                        for (int u = 0; u < 8; u += 4) {
                            *reinterpret_cast<float4*>(t0 + u) = *reinterpret_cast<const float4*>(src + u);
                            *reinterpret_cast<float4*>(t1 + u) = *reinterpret_cast<const float4*>(src + 2 * TW + u);
                        }
                    } else {
                        *reinterpret_cast<float4*>(t0) = *reinterpret_cast<const float4*>(src);
                        *reinterpret_cast<float4*>(t1) = *reinterpret_cast<const float4*>(src + 2 * TW);
                    }
                    pooled(t0, t1, i);
                }
            }
        }
        if (chunk == 0) CINE_STAMP(3);
        __syncthreads();
        if (chunk == 0) CINE_STAMP(4);
        if (chunk + 1 < a.nchunks) issue(chunk + 1);
        __builtin_amdgcn_sched_barrier(0);
        // ---- MFMA sweep: 9 taps x CK/4 operand groups, software-pipelined one group ahead (as conv_tile: same accumulation order)
        {
            constexpr int KS = CK / 4, NG = 9 * KS;
            float af[2][CT], bf[2][MT];
            auto load_group = [&](int g, float (&wa)[CT], float (&xa)[MT]) {
                const int tap = g / KS, ks = g % KS;
                const int dy = tap / 3, dx = tap % 3;
#pragma unroll
                for (int ct = 0; ct < CT; ++ct) wa[ct] = w_lds[base_w + (tap * CK + 4 * ks) * C::COTP + 16 * ct];
#pragma unroll
                for (int f = 0; f < MT; ++f) xa[f] = in_lds[base_in[dx] + (4 * ks) * C::PS + (f * C::RPF + dy) * C::COLS];
            };
            __builtin_amdgcn_s_setprio(0);
            load_group(0, af[0], bf[0]);
#pragma unroll
            for (int g = 0; g < NG; ++g) {
                if (g + 1 < NG) load_group(g + 1, af[(g + 1) & 1], bf[(g + 1) & 1]);
#pragma unroll
                for (int ct = 0; ct < CT; ++ct)
#pragma unroll
                    for (int f = 0; f < MT; ++f)
                        acc[ct][f] = __builtin_amdgcn_mfma_f32_16x16x4f32(bf[g & 1][f], af[g & 1][ct], acc[ct][f], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
            __builtin_amdgcn_s_setprio(2);
        }
        if (chunk == 0) CINE_STAMP(5);
    }
    CINE_STAMP(6);

    // ---- epilogue.  Lane holds output row m = co0 + 16 (wm CT + ct) + q at pixels 4 kk .. 4 kk + 3 of fragment f; a fragment is
    // 16 consecutive floats of the plane (RPF rows of TW), so fragment f sits 64 f bytes behind fragment 0: instruction immediates
    constexpr int PPR = TW >= 4 ? 4 : TW;
    const int pr0 = (4 * kk) / TW, pc0 = (4 * kk) % TW;
    const int fr0 = r0 + wn * MT * C::RPF;                          // first image row of my fragments
    if (a.bias) {
        const float* bsel = n >= a.set_split ? a.bias1 : a.bias;
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) {
            const int m = co0 + 16 * (wm * CT + ct) + q;
            const float bv = m < a.rows ? bsel[m] : 0.f;
#pragma unroll
            for (int f = 0; f < MT; ++f)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[ct][f][j] += bv;
        }
    }
    if (a.relu) {
#pragma unroll
        for (int ct = 0; ct < CT; ++ct)
#pragma unroll
            for (int f = 0; f < MT; ++f)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[ct][f][j] = fmaxf(acc[ct][f][j], 0.f);
    }
    const bool full = r0 + C::TH <= a.H;
    unsigned long long vmask = ~0ull;
    if (!full) {
        vmask = 0;
#pragma unroll
        for (int f = 0; f < MT; ++f)
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if (fr0 + f * C::RPF + (4 * kk + j) / TW < a.H) vmask |= 1ull << (4 * f + j);
    }
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) {
        const int m = co0 + 16 * (wm * CT + ct) + q;
        if (m >= a.rows) continue;
        float* yb = a.y + ((long)n * a.rows + m) * a.H * TW + (fr0 + pr0) * TW + pc0;
        if (full) {
#pragma unroll
            for (int f = 0; f < MT; ++f) {
                if constexpr (PPR == 4) {
                    *reinterpret_cast<float4*>(yb + 16 * f) = make_float4(acc[ct][f][0], acc[ct][f][1], acc[ct][f][2], acc[ct][f][3]);
                } else {
                    *reinterpret_cast<float2*>(yb + 16 * f) = make_float2(acc[ct][f][0], acc[ct][f][1]);
                    *reinterpret_cast<float2*>(yb + 16 * f + TW) = make_float2(acc[ct][f][2], acc[ct][f][3]);
                }
            }
        } else {
#pragma unroll
            for (int f = 0; f < MT; ++f) {
                if constexpr (PPR == 4) {
                    if ((vmask >> (4 * f)) & 1ull)
                        *reinterpret_cast<float4*>(yb + 16 * f) = make_float4(acc[ct][f][0], acc[ct][f][1], acc[ct][f][2], acc[ct][f][3]);
                } else {
                    if ((vmask >> (4 * f)) & 1ull) *reinterpret_cast<float2*>(yb + 16 * f) = make_float2(acc[ct][f][0], acc[ct][f][1]);
                    if ((vmask >> (4 * f + 2)) & 1ull) *reinterpret_cast<float2*>(yb + 16 * f + TW) = make_float2(acc[ct][f][2], acc[ct][f][3]);
                }
            }
        }
    }
    CINE_STAMP(7);
    if (a.ypart) {
        // InstanceNorm partial {count, mean, M2} of this workgroup's pixels per output row: exact two-pass per WAVE in registers,
        // the WN wave records merged with Chan's formula by one thread per row (conv_tile's arithmetic, operation for operation)
        const int rows_w = min(max(a.H - fr0, 0), MT * C::RPF);
        const float cnt_w = (float)(rows_w * TW);
        float mean_w[CT], m2_w[CT];
        auto wave_stats = [&](auto fullc) {
            constexpr bool FULL = decltype(fullc)::value;
#pragma unroll
            for (int ct = 0; ct < CT; ++ct) {
                float sacc = 0.f;
#pragma unroll
                for (int f = 0; f < MT; ++f)
#pragma unroll
                    for (int j = 0; j < 4; ++j) sacc += (FULL || ((vmask >> (4 * f + j)) & 1ull)) ? acc[ct][f][j] : 0.f;
                sacc = add_xor16(sacc);
                sacc = add_xor32(sacc);
                mean_w[ct] = cnt_w > 0.f ? sacc / cnt_w : 0.f;
                float qacc = 0.f;
#pragma unroll
                for (int f = 0; f < MT; ++f)
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const float d = acc[ct][f][j] - mean_w[ct];
                        qacc += (FULL || ((vmask >> (4 * f + j)) & 1ull)) ? d * d : 0.f;
                    }
                qacc = add_xor16(qacc);
                qacc = add_xor32(qacc);
                m2_w[ct] = qacc;
            }
        };
        if (full) wave_stats(std::true_type{}); else wave_stats(std::false_type{});
        // the WN wave records of a row are merged with Chan's formula (conv_tile's expressions verbatim: same bits)
        auto merge_store = [&](const float (&rc)[WN], const float (&rm)[WN], const float (&rq)[WN], int row) {
            float cnt = 0.f, mean = 0.f;
#pragma unroll
            for (int w = 0; w < WN; ++w) { cnt += rc[w]; mean += rc[w] * rm[w]; }
            mean /= cnt;
            float m2 = 0.f;
#pragma unroll
            for (int w = 0; w < WN; ++w) {
                const float d = rm[w] - mean;
                m2 += rq[w] + rc[w] * d * d;
            }
            float* o = a.ypart + (((long)n * a.rows + co0 + row) * a.tiles + tile) * 3;
            o[0] = cnt; o[1] = mean; o[2] = m2;
        };
        if constexpr (WN == 1) {
            // one wave holds all pixels of its rows: no exchange, no barrier
            if (kk == 0) {
#pragma unroll
                for (int ct = 0; ct < CT; ++ct) {
                    const int row = 16 * (wm * CT + ct) + q;
                    if (co0 + row < a.rows) { const float rc[1] = {cnt_w}, rm[1] = {mean_w[ct]}, rq[1] = {m2_w[ct]}; merge_store(rc, rm, rq, row); }
                }
            }
        } else {
            float* red = st_lds + 2 * nch;              // [WN][COT][3], a region of its own: no barrier against the last sweep's LDS reads
            if (kk == 0) {
#pragma unroll
                for (int ct = 0; ct < CT; ++ct) {
                    float* o = red + (wn * C::COT + 16 * (wm * CT + ct) + q) * 3;
                    o[0] = cnt_w; o[1] = mean_w[ct]; o[2] = m2_w[ct];
                }
            }
            __syncthreads();
            if (tid < C::COT && co0 + tid < a.rows) {
                float rc[WN], rm[WN], rq[WN];
#pragma unroll
                for (int w = 0; w < WN; ++w) { const float* r = red + (w * C::COT + tid) * 3; rc[w] = r[0]; rm[w] = r[1]; rq[w] = r[2]; }
                merge_store(rc, rm, rq, tid);
            }
        }
    }
    CINE_STAMP(8);
    CINE_STAMP_RT(10);
}

template <int CK, int CT, int WM, int WN, int MT, int TW, int MODE>
int launch_plane(const PlaneArgs& p, int n, hipStream_t st) {
    using C = ConvCfg<CK, CT, WM, WN, MT, TW, 9>;
    auto kern = conv_plane_kernel<CK, CT, WM, WN, MT, TW, MODE>;
    const size_t lds = C::lds_bytes(p.c0 + p.c1) + (WN > 1 ? C::RED_FLOATS * sizeof(float) : 0);      // + the statistics exchange of the epilogue
    static std::once_flag once[64];
    static hipError_t status[64];
    if (lds > 64 * 1024) {
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess) dev = 0;
        CINE_REQUIRE(dev >= 0 && dev < 64, CINE_EUNSUPPORTED, "conv_plane_kernel: device index %d", dev);
        std::call_once(once[dev], [&] {
            status[dev] = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        });
        CINE_REQUIRE(status[dev] == hipSuccess, CINE_EHIP, "conv_plane_kernel: hipFuncSetAttribute: %s", hipGetErrorString(status[dev]));
    }
    CINE_REQUIRE(lds <= 160 * 1024, CINE_EUNSUPPORTED, "conv_plane_kernel: %d input channels need %zu bytes of LDS", p.cin, lds);
    CINE_REQUIRE(C::G == 1 || p.cin % CK == 0, CINE_EUNSUPPORTED, "conv_plane_kernel: %d input channels on a shape with %d channel groups", p.cin, C::G);
    const dim3 grid(p.tiles, ceil_div(p.rowsp, C::COT), n);
    ProfScope prof(F_CONV3, st);
    hipLaunchKernelGGL(kern, grid, dim3(C::NT), lds, st, p);
    return check_launch("conv_plane_kernel");
}

}  // namespace

// The general dispatcher (conv_kernels.hip: launch_cfg) has chosen the tile configuration (ck, ct, wm, wn, mt, tw); take the
// launch when the layer is one of this kernel's shapes.
int launch_conv_plane(const ConvArgs& a, int ck, int ct, int wm, int wn, int mt, int tw, hipStream_t st, bool* handled) {
    *handled = false;
    if (!(g_plane_on & 1)) return CINE_OK;
    if (a.vol || a.D != 1 || a.addend || a.accum || a.gate || a.pair_n > 0 || a.tconv_cout > 0) return CINE_OK;
    if (a.W != tw || a.n <= 0 || a.n > 65535) return CINE_OK;
    const Src& s0 = a.s0; const Src& s1 = a.s1;
    auto al16 = [](const void* p) { return reinterpret_cast<uintptr_t>(p) % 16 == 0; };
    if (!al16(a.y) || !al16(s0.x) || (s1.c > 0 && !al16(s1.x)) || !al16(a.wp0) || !al16(a.wp1)) return CINE_OK;
    int mode;
    if (a.add_src1) {                                                   // the MWCNN's IWT + additive skip, nothing else
        if (!(s0.mode == 4 && (s0.act & 1) && s1.mode == 1 && ck == 8 && s0.c == 4 * s1.c && s1.c == a.cin && a.cin % 8 == 0)) return CINE_OK;
        mode = 4;
    } else if (s0.mode == 0 && s1.c == 0) mode = 0;                            // the 2-channel first layer (4-channel chunk); the input-gradient convs of training
    else if (s0.mode == 1 && (s1.c == 0 || s1.mode == 1) && ck == 8) mode = 1;
    else if (s0.mode == 2 && s1.c == 0 && ck == 8) mode = 2;
    else if (s0.mode == 3 && (s0.act & 1) && s1.c == 0 && ck == 8 && s0.c % 8 == 0) mode = 3;
    else return CINE_OK;
    if (mode == 3) {
        if (s0.w != 2 * a.W || s0.h != 2 * a.H) return CINE_OK;
    } else if (mode == 4) {
        if (2 * s0.w != a.W || 2 * s0.h != a.H || s1.w != a.W || s1.h != a.H || reinterpret_cast<uintptr_t>(s0.x) % 8 != 0) return CINE_OK;
    } else if (mode != 2) {
        if (s0.w != a.W || s0.h != a.H || (s1.c > 0 && (s1.w != a.W || s1.h != a.H))) return CINE_OK;
        if (s1.c > 0 && s0.c % ck != 0) return CINE_OK;
    } else {
        if (s0.w != 2 * a.W || (s0.h != 2 * a.H && s0.h != 2 * a.H + 1)) return CINE_OK;
    }
    if (mode != 0) {
        if (s0.np < 1 || s0.np > 16 || !s0.part || (s1.c > 0 && (s1.np < 1 || s1.np > 16 || !s1.part))) return CINE_OK;
        if (s0.c + s1.c > 256) return CINE_OK;
    }
    const bool whole = a.cin % ck == 0;             // a narrower last chunk (2 of 8 channels, 10 = 8 + 2): on the G == 1 shapes, one source
    PlaneArgs p{};
    p.x0 = s0.x; p.part0 = s0.part; p.c0 = s0.c; p.np0 = s0.np;
    p.x1 = s1.c > 0 ? s1.x : nullptr; p.part1 = s1.c > 0 ? s1.part : nullptr; p.c1 = s1.c; p.np1 = s1.c > 0 ? s1.np : 0;
    p.cs0 = (long)s0.h * s0.w; p.sh0 = s0.h;
    p.wp0 = a.wp0; p.wp1 = a.wp1; p.set_split = a.set_split; p.bias = a.bias; p.bias1 = a.bias1; p.relu = a.relu;
    p.y = a.y; p.ypart = a.ypart; p.cin = a.cin; p.rows = a.rows; p.rowsp = a.rowsp; p.H = a.H; p.nchunks = a.nchunks; p.tiles = a.tiles;
    p.slope = a.slope; p.eps = a.eps;
    // G > 1 configurations (planes narrower than 16) stage whole chunks only
#define CINE_PLANE_CASE(CK_, CT_, WM_, WN_, MT_, TW_, MODE_, NEEDWHOLE)                                                   \
    if (ck == CK_ && ct == CT_ && wm == WM_ && wn == WN_ && mt == MT_ && tw == TW_ && mode == MODE_ && (whole || !(NEEDWHOLE))) { \
        *handled = true;                                                                                                  \
        return launch_plane<CK_, CT_, WM_, WN_, MT_, TW_, MODE_>(p, a.n, st);                                              \
    }
    CINE_PLANE_CASE(4, 1, 1, 4, 13, 16, 0, false)
    CINE_PLANE_CASE(8, 1, 1, 4, 13, 16, 1, false)
    CINE_PLANE_CASE(8, 1, 1, 4, 13, 16, 0, false)
    CINE_PLANE_CASE(8, 1, 2, 2, 13, 16, 0, true)
    CINE_PLANE_CASE(8, 1, 2, 2, 13, 8, 0, true)
    CINE_PLANE_CASE(8, 1, 4, 1, 13, 8, 0, true)
    CINE_PLANE_CASE(8, 1, 4, 1, 13, 4, 0, true)
    CINE_PLANE_CASE(8, 2, 4, 1, 4, 2, 0, true)
    CINE_PLANE_CASE(8, 1, 1, 4, 13, 8, 0, true)           // the MWCNN's plane shapes (XT / XF planes at its coarser scales)
    CINE_PLANE_CASE(8, 1, 4, 1, 4, 2, 0, true)
    CINE_PLANE_CASE(8, 1, 2, 2, 7, 4, 0, true)
    CINE_PLANE_CASE(8, 1, 1, 4, 13, 8, 1, true)           // ... and their InstanceNorm + LeakyReLU inner convs (mwcnn.py:143-168)
    CINE_PLANE_CASE(8, 1, 4, 1, 4, 2, 1, true)
    CINE_PLANE_CASE(8, 1, 2, 2, 7, 4, 1, true)
    CINE_PLANE_CASE(8, 1, 4, 1, 13, 8, 1, true)           // 16 -> 64 before an IWT
    CINE_PLANE_CASE(8, 1, 1, 4, 13, 16, 4, true)          // first conv of an MWCNN synthesis scale / its last conv: IWT of the scale below + additive skip
    CINE_PLANE_CASE(8, 1, 1, 4, 13, 8, 4, true)
    CINE_PLANE_CASE(8, 1, 2, 2, 7, 4, 4, true)
    CINE_PLANE_CASE(8, 1, 1, 4, 13, 8, 3, true)           // first conv of an MWCNN scale: Haar DWT of the scale above on load
    CINE_PLANE_CASE(8, 1, 2, 2, 7, 4, 3, true)
    CINE_PLANE_CASE(8, 1, 4, 1, 4, 2, 3, true)
    CINE_PLANE_CASE(8, 1, 2, 2, 13, 8, 1, true)
    CINE_PLANE_CASE(8, 1, 2, 2, 13, 8, 2, true)
    CINE_PLANE_CASE(8, 1, 4, 1, 13, 4, 1, true)
    CINE_PLANE_CASE(8, 1, 4, 1, 13, 4, 2, true)
    CINE_PLANE_CASE(8, 2, 4, 1, 4, 2, 1, true)
    CINE_PLANE_CASE(8, 2, 4, 1, 4, 2, 2, true)
    CINE_PLANE_CASE(8, 1, 4, 1, 4, 2, 2, true)            // the coarsest U-Net level as two 64-row workgroups per plane
#undef CINE_PLANE_CASE
    return CINE_OK;
}

}  // namespace cine

// Diagnostics: route the plane-wide 3x3 convolutions through the general kernel (0) or the lean one (1, the default).  The two are
// bit-identical; the switch exists for that test and for A/B timing.  State of the CALLING THREAD (like cine_set_side_stream): launches
// enqueued by other threads keep their own setting (default 7).
extern "C" int cine_set_conv_plane(int on) {
    cine::set_wgrad_plane((on & 16) ? 0 : 1);       // bit 4 SET: the plane-wide weight gradients (training) on the general kernel
    cine::g_plane_on = on & 7;       // bit 0 plane-wide 3x3 convs, bit 1 transpose convs, bit 2 wide planes / volumes (7 = all, the default)
    return CINE_OK;
}

// ================================================================ transpose conv k2 s2 of plane-wide tiles
// unet.py:212-218 as a GEMM with 4 cout rows over the input pixels (row m = 2 (s cout + co) + b -> output (2y + s, 2x + b)), for the
// cascade U-Nets' shapes: ALL input channels of the tile are staged at once (InstanceNorm + LeakyReLU on load; one barrier in the
// whole kernel instead of a barrier pair per 16-channel chunk), the weights never touch LDS -- every lane streams its own B operand
// W[k][row] from L2 two k-steps ahead of the MFMAs that use it.  Same pixel tiles, accumulation order (k ascending), paired 16-byte
// stores and statistics records as conv_tile's TAPS = 1 path: bit-identical outputs.
namespace cine {
namespace {

struct TconvPlaneArgs {
    const float* x; const float* part; int np, mode;
    const float* wp0; const float* wp1; int set_split;
    float* y; float* ypart;
    int cout, rows, rowsp, H, tiles;
    float slope, eps;
};

template <int CIN, int CT, int WM, int MT, int TW>
struct TconvCfg {
    static constexpr int NT = 64 * WM;
    static constexpr int TPX = 16 * MT;                    // pixels of a tile: TH rows of TW, contiguous in the plane
    static constexpr int TH = TPX / TW;
    static constexpr int PS = ((TPX + 31) / 32) * 32 + 16; // channel stride == 16 (mod 32)
    static constexpr int NV = TPX / 4;                     // 16-byte pieces per channel
    static constexpr int G = NT / NV;                      // channel groups staged in parallel
    static constexpr int NCI = (CIN + G - 1) / G;          // pieces per thread
    static constexpr int COT = 16 * CT * WM;
    static constexpr int KSN = CIN / 4;
    static_assert(TPX % 4 == 0 && G >= 1 && CIN % 4 == 0, "tile shape");
    static size_t lds_bytes() { return (size_t)(CIN * PS + 2 * CIN) * sizeof(float); }
};

template <int CIN, int CT, int WM, int MT, int TW>
__global__ __launch_bounds__(64 * WM, (WM > 4 ? 4 : 2)) void tconv_plane_kernel(TconvPlaneArgs a) {
    using C = TconvCfg<CIN, CT, WM, MT, TW>;
    constexpr int G = C::G, NCI = C::NCI, NV = C::NV, KSN = C::KSN;
    extern __shared__ __align__(16) float smem_f[];
    float* in_lds = smem_f;
    float* st_lds = smem_f + CIN * C::PS;
    const int tid = threadIdx.x, lane = tid & 63, wm = tid >> 6;
    const int tile = blockIdx.x, n = blockIdx.z;
    const int co0 = blockIdx.y * C::COT;
    const int q = lane & 15, kk = lane >> 4;
    const float* wp = n >= a.set_split ? a.wp1 : a.wp0;
    __builtin_amdgcn_s_setprio(2);
    // ---- staging map: piece sv of channel group sg; rows beyond the image (a last tile that overhangs) read as zero
    const int sg = tid / NV, sv = tid - sg * NV;
    const bool slot = sg < G;
    const int sgc = min(sg, G - 1);
    const int hw = a.H * TW;                                // plane size in pixels
    const int p0 = tile * C::TPX + 4 * sv;                  // my first pixel in the plane
    const bool pok = p0 < hw;                               // (TW divides 4 or 4 divides TW, hw % 4 == 0: a piece is inside or outside as a whole)
    const float* xb = a.x + (long)n * CIN * hw + (long)sgc * hw + min(p0, hw - 4);
    // statistics records first, then all the input pieces, then the first weights
    float2 mr = make_float2(0.f, 1.f);
    constexpr int NPQ = 16;
    float prec[3 * NPQ];
    const bool need = a.mode == 1 && tid < CIN;
    if (need) {
        const float* pp = a.part + ((long)n * CIN + tid) * a.np * 3;
        if (a.np <= 4) {
#pragma unroll
            for (int i = 0; i < 12; ++i) prec[i] = pp[min(i, 3 * a.np - 1)];
        } else load_partials<NPQ>(pp, a.np, prec);
    }
    float4 xraw[NCI];
#pragma unroll
    for (int i = 0; i < NCI; ++i) xraw[i] = *reinterpret_cast<const float4*>(xb + (long)min(i * G, CIN - 1 - sgc) * hw);
    // B operand (k x rows): lane = output row q of a 16-row block, channel kk of the k-step
    const float* wl = wp + (long)kk * a.rowsp + co0 + 16 * (wm * CT) + q;
    constexpr int PD = 2;                                   // k-steps of weights in flight
    float wreg[PD][CT];
    auto wload = [&](int ks, float (&w)[CT]) {
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) w[ct] = co0 + 16 * (wm * CT + ct) < a.rowsp ? wl[(long)(4 * ks) * a.rowsp + 16 * ct] : 0.f;
    };
#pragma unroll
    for (int d = 0; d < PD; ++d) wload(min(d, KSN - 1), wreg[d]);
    __builtin_amdgcn_sched_barrier(0);
    if (tid < CIN) {
        if (need) {
            if (a.np <= 4) { float r4[12];
#pragma unroll
                for (int i = 0; i < 12; ++i) r4[i] = prec[i];
                mr = merge_loaded<4>(r4, a.np, a.eps);
            } else mr = merge_loaded<NPQ>(prec, a.np, a.eps);
        }
        st_lds[2 * tid] = mr.y; st_lds[2 * tid + 1] = -mr.x * mr.y;
    }
    __syncthreads();
    if (slot) {
#pragma unroll
        for (int i = 0; i < NCI; ++i) {
            const int c = sgc + i * G;
            if (c >= CIN) break;
            float4 o = xraw[i];
            if (a.mode == 1) {
                const float2 ss = *reinterpret_cast<const float2*>(st_lds + 2 * c);
                act_piece<4>(reinterpret_cast<float*>(&o), ss.x, ss.y, a.slope);
            }
            if (!pok) o = make_float4(0.f, 0.f, 0.f, 0.f);
            *reinterpret_cast<float4*>(in_lds + c * C::PS + 4 * sv) = o;
        }
    }
    __syncthreads();

    f32x4 acc[CT][MT];
#pragma unroll
    for (int ct = 0; ct < CT; ++ct)
#pragma unroll
        for (int f = 0; f < MT; ++f) acc[ct][f] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const float* ain = in_lds + kk * C::PS + q;             // A operand (pixels x k): lane = pixel q of the fragment, channel kk
    __builtin_amdgcn_s_setprio(0);
#pragma unroll 2
    for (int ks = 0; ks < KSN; ++ks) {
        float wcur[CT];
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) wcur[ct] = wreg[ks % PD][ct];
        if (ks + PD < KSN) wload(ks + PD, wreg[ks % PD]);
        float xa[MT];
#pragma unroll
        for (int f = 0; f < MT; ++f) xa[f] = ain[(4 * ks) * C::PS + 16 * f];
#pragma unroll
        for (int ct = 0; ct < CT; ++ct)
#pragma unroll
            for (int f = 0; f < MT; ++f)
                acc[ct][f] = __builtin_amdgcn_mfma_f32_16x16x4f32(xa[f], wcur[ct], acc[ct][f], 0, 0, 0);
    }
    __builtin_amdgcn_s_setprio(2);

    // ---- epilogue: lanes q, q ^ 1 hold the two x-parities of the same 4 pixels; swapping halves gives each lane 4 consecutive output floats
    const int r0 = tile * C::TH;
    const bool full = r0 + C::TH <= a.H;
    unsigned long long vmask = ~0ull;
    if (!full) {
        vmask = 0;
#pragma unroll
        for (int f = 0; f < MT; ++f)
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if (r0 + (16 * f + 4 * kk + j) / TW < a.H) vmask |= 1ull << (4 * f + j);
    }
    const bool odd = q & 1;
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) {
        const int m = co0 + 16 * (wm * CT + ct) + q;
        if (m >= a.rows) continue;
        const int me = m >> 1, co = me % a.cout, ay = me / a.cout;
        float* yb = a.y + ((long)n * a.cout + co) * (2 * a.H) * (2 * TW);
#pragma unroll
        for (int f = 0; f < MT; ++f) {
            const float s0 = odd ? acc[ct][f][0] : acc[ct][f][2], s1 = odd ? acc[ct][f][1] : acc[ct][f][3];
            const float t0 = __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(s0), 0xB1, 0xf, 0xf, true));   // quad_perm [1,0,3,2]
            const float t1 = __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(s1), 0xB1, 0xf, 0xf, true));
            const float4 o = odd ? make_float4(t0, acc[ct][f][2], t1, acc[ct][f][3]) : make_float4(acc[ct][f][0], t0, acc[ct][f][1], t1);
            const int p = 16 * f + 4 * kk + (odd ? 2 : 0);
            const int gy = r0 + p / TW, gx = p % TW;
            if (gy < a.H) *reinterpret_cast<float4*>(yb + (long)(2 * gy + ay) * (2 * TW) + 2 * gx) = o;
        }
    }
    if (a.ypart) {
        const int rows_w = min(max(a.H - r0, 0), C::TH);
        const float cnt_w = (float)(rows_w * TW);
        float mean_w[CT], m2_w[CT];
        auto wave_stats = [&](auto fullc) {
            constexpr bool FULL = decltype(fullc)::value;
#pragma unroll
            for (int ct = 0; ct < CT; ++ct) {
                float sacc = 0.f;
#pragma unroll
                for (int f = 0; f < MT; ++f)
#pragma unroll
                    for (int j = 0; j < 4; ++j) sacc += (FULL || ((vmask >> (4 * f + j)) & 1ull)) ? acc[ct][f][j] : 0.f;
                sacc = add_xor16(sacc);
                sacc = add_xor32(sacc);
                mean_w[ct] = cnt_w > 0.f ? sacc / cnt_w : 0.f;
                float qacc = 0.f;
#pragma unroll
                for (int f = 0; f < MT; ++f)
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const float d = acc[ct][f][j] - mean_w[ct];
                        qacc += (FULL || ((vmask >> (4 * f + j)) & 1ull)) ? d * d : 0.f;
                    }
                qacc = add_xor16(qacc);
                qacc = add_xor32(qacc);
                m2_w[ct] = qacc;
            }
        };
        if (full) wave_stats(std::true_type{}); else wave_stats(std::false_type{});
        if (kk == 0) {
#pragma unroll
            for (int ct = 0; ct < CT; ++ct) {
                const int m = co0 + 16 * (wm * CT + ct) + q;
                if (m >= a.rows) continue;
                // one wave holds all pixels of its rows; conv_tile's single-record merge, expression for expression
                float cnt = 0.f, mean = 0.f;
                cnt += cnt_w; mean += cnt_w * mean_w[ct];
                mean /= cnt;
                float m2 = 0.f;
                const float d = mean_w[ct] - mean;
                m2 += m2_w[ct] + cnt_w * d * d;
                const int co = (m >> 1) % a.cout, ab = 2 * ((m >> 1) / a.cout) + (m & 1);
                float* o = a.ypart + (((long)n * a.cout + co) * (a.tiles * 4) + tile * 4 + ab) * 3;
                o[0] = cnt; o[1] = mean; o[2] = m2;
            }
        }
    }
}

template <int CIN, int CT, int WM, int MT, int TW>
int launch_tconv(const TconvPlaneArgs& p, int n, hipStream_t st) {
    using C = TconvCfg<CIN, CT, WM, MT, TW>;
    auto kern = tconv_plane_kernel<CIN, CT, WM, MT, TW>;
    const size_t lds = C::lds_bytes();
    static std::once_flag once[64];
    static hipError_t status[64];
    if (lds > 64 * 1024) {
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess) dev = 0;
        CINE_REQUIRE(dev >= 0 && dev < 64, CINE_EUNSUPPORTED, "tconv_plane_kernel: device index %d", dev);
        std::call_once(once[dev], [&] {
            status[dev] = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        });
        CINE_REQUIRE(status[dev] == hipSuccess, CINE_EHIP, "tconv_plane_kernel: hipFuncSetAttribute: %s", hipGetErrorString(status[dev]));
    }
    const dim3 grid(p.tiles, ceil_div(p.rowsp, C::COT), n);
    ProfScope prof(F_TCONV, st);
    hipLaunchKernelGGL(kern, grid, dim3(C::NT), lds, st, p);
    return check_launch("tconv_plane_kernel");
}

}  // namespace

// Called by the general dispatcher for a 2-D transpose conv with the pixel-tile shape (mt fragments, tw) it chose.
int launch_tconv_plane(const ConvArgs& a, int mt, int tw, hipStream_t st, bool* handled) {
    *handled = false;
    if (!(g_plane_on & 2)) return CINE_OK;
    if (a.vol || a.D != 1 || a.tconv_cout <= 0 || !a.tvec || a.bias || a.addend || a.relu || a.accum || a.s1.c > 0 || a.add_src1) return CINE_OK;
    if (a.W != tw || a.n <= 0 || a.n > 65535 || a.s0.mode > 1 || a.s0.w != a.W || a.s0.h != a.H || (a.H * a.W) % 4 != 0) return CINE_OK;
    if (a.s0.mode == 1 && (!a.s0.part || a.s0.np < 1 || a.s0.np > 16)) return CINE_OK;
    auto al16 = [](const void* p) { return reinterpret_cast<uintptr_t>(p) % 16 == 0; };
    if (!al16(a.y) || !al16(a.s0.x)) return CINE_OK;
    TconvPlaneArgs p{};
    p.x = a.s0.x; p.part = a.s0.part; p.np = a.s0.np; p.mode = a.s0.mode;
    p.wp0 = a.wp0; p.wp1 = a.wp1; p.set_split = a.set_split; p.y = a.y; p.ypart = a.ypart;
    p.cout = a.tconv_cout; p.rows = a.rows; p.rowsp = a.rowsp; p.H = a.H; p.tiles = a.tiles; p.slope = a.slope; p.eps = a.eps;
#define CINE_TCONV_CASE(CIN_, CT_, WM_, MT_, TW_)                                   \
    if (a.cin == CIN_ && mt == MT_ && tw == TW_) {                                   \
        *handled = true;                                                            \
        return launch_tconv<CIN_, CT_, WM_, MT_, TW_>(p, a.n, st);                   \
    }
    CINE_TCONV_CASE(32, 1, 4, 13, 8)
    CINE_TCONV_CASE(64, 1, 8, 13, 4)         // eight waves: all 128 rows of a plane in one workgroup, the 64-channel input staged once
    CINE_TCONV_CASE(128, 2, 8, 4, 2)
#undef CINE_TCONV_CASE
    return CINE_OK;
}

}  // namespace cine

// ================================================================ 3x3 (x3) convolutions of planes / volumes wider than one tile
// The sensitivity network's 208 x 208 planes, the CRNN hybrids' all-frame convolutions on 200 x 200 planes and the 3-D U-Net's
// 15 x 200 x 200 / 7 x 100 x 100 volumes (3x3x3 as three 3x3 passes per depth offset, the V3 form of conv_tile).  Same idea as
// conv_plane_kernel; what the column tiling adds: the two halo columns of a tile row ride in the SAME prefetch as the row's
// 16-byte pieces (one extra 4-byte load per channel in the threads that own a row's first / last piece) instead of conv_tile's
// synchronous element-wise fetch after every chunk's staging, and columns / rows / slices that lie outside the source are
// zeroed once or by a wave-uniform branch.  Optional bias / addend / ReLU epilogue (the CRNN cells, recurrent_varnet.py:122-134).
// Bit-identical to conv_tile (same chunk order, accumulation order, epilogue and statistics arithmetic).
namespace cine {
namespace {

struct WideArgs {
    const float* x0; const float* part0; int c0, np0, d0;
    const float* x1; const float* part1; int c1, np1, d1;
    const float* wp0; const float* wp1; int set_split;
    const float* bias; const float* bias1; const float* addend; int relu;
    float* y; float* ypart;
    int cin, rows, rowsp, D, H, W, nchunks, ncc, tiles, tiles_w, tiles_hw;
    float slope, eps;
    // CRNN time-sweep steps: optional second output (accum += y, or = y), and the second sample set of a pair launch
    float* accum; int accum_store;
    int pair_n, accum_store_b; const float* x_b; const float* addend_b; float* y_b; float* accum_b;
    const float* gate; const float* gate_b;       // y = gate > 0 ? v : 0 (ConvArgs::gate: the adjoint step of a ReLU recurrence)
    int xcd_bands;                                // > 0: 1-D grid, bands (sample, depth slice, tile row) dealt over the XCDs (see the kernel)
};

// MODE 0: plain sources; 1: InstanceNorm + LeakyReLU on load.  V3: volumes, chunk = (depth offset, 8 channels).
// waves per SIMD asked of the register allocator: ConvCfg's estimate, except the 40-row shape, which exists to have ALL its workgroups resident at once
// (975 of them for 15 frames of 200 x 200 on 1 024 slots) and so needs four
template <int CT, int WM, int WN, int MT> constexpr int wide_minw() { return MT == 10 ? 4 : ConvCfg<8, CT, WM, WN, MT, 16, 9>::MINW; }
template <int CT, int WM, int WN, int MT, int MODE, int V3>
__global__ __launch_bounds__(64 * WM * WN, (wide_minw<CT, WM, WN, MT>())) void conv_wide_kernel(WideArgs a) {       // (by value: a pair launch swaps pointers)
    constexpr int CK = 8, TW = 16;
    using C = ConvCfg<CK, CT, WM, WN, MT, TW, 9>;
    constexpr int NT = C::NT, PR = C::PR, RP = C::RP, G = C::G, NCI = C::NCI, NWT = C::NWT;
    static_assert(C::KR == 1 && C::PW == 4, "one (row, 16-byte piece) slot per thread");
    extern __shared__ __align__(16) float smem_f[];
    float* in_lds = smem_f;
    float* w_lds = smem_f + C::IN_FLOATS;
    float* st_lds = w_lds + C::W_FLOATS;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    // (measured and rejected: a grid of D x (tiles per slice rounded up to 8) blocks, so that the three depth offsets of an (x, y)
    // tile land on one XCD and share its L2 -- cfg 4 150.6 vs 153.2 slices/s, the dummy blocks and the changed dispatch order cost more)
    int n = blockIdx.z;
    int tile = blockIdx.x;
    if (a.xcd_bands > 0) {
        // XCD-aware decode of a 1-D grid (launch_wide).  A tile row is 64 bytes of a 128-byte line, and its halo columns sit in the lines beside it: with tiles
        // dealt round-robin over the 8 XCDs (each with an L2 of its own) every tile row pulls two whole lines through its XCD's L2 for 64 bytes of interior --
        // four times the algorithmic reads (PMC FETCH_SIZE: 9.4 GB per cfg-5 slice for 2.4 GB of operands).  Workgroup ids go round-robin over the XCDs, so
        // id = xcd + 8 slot: a BAND (one row of tiles of one sample / depth slice) lives on ONE XCD, its tiles in consecutive slots -- every line of the
        // band is fetched once.  Whole bands are dealt over the XCDs eight at a time; the tiles of the last (bands mod 8) bands are cut into eight contiguous
        // runs, one per XCD, so that no XCD gets a band more than another (15 frames x 5 tile rows = 75 bands: 9 bands + 5 tiles each = 122 workgroups per XCD
        // on its 128 resident slots; a tenth band on three of the XCDs would be 130) and the runs still share their lines (ids past a run's end: <= 7, idle).
        // (volumes keep WHOLE bands to the end -- the last (bands mod 8) bands go to the first XCDs, the others idle through those ids: a depth slice's bands stay
        //  with the XCDs that hold its neighbours' lines; measured at cfg 4: 8.96 ms per slice against 9.17 with runs, the same in flight)
        const int i = blockIdx.x, full = V3 ? (a.xcd_bands + 7) & ~7 : a.xcd_bands & ~7, seg1 = full * a.tiles_w;
        int band, txb;
        if (i < seg1) {
            const int xcd = i & 7, slot = i >> 3, k = slot / a.tiles_w;
            txb = slot - k * a.tiles_w; band = xcd + 8 * k;
            if (V3 && band >= a.xcd_bands) return;                    // (uniform: before any barrier)
        } else {
            const int rest = (a.xcd_bands - full) * a.tiles_w, run = (rest + 7) >> 3;
            const int j = i - seg1, t = (j & 7) * run + (j >> 3);
            if ((j >> 3) >= run || t >= rest) return;                 // (uniform: before any barrier)
            const int kb = t / a.tiles_w;
            txb = t - kb * a.tiles_w; band = full + kb;
        }
        const int th = a.tiles_hw / a.tiles_w, per_n = th * a.D;  // bands per sample
        n = band / per_n;
        const int r = band - n * per_n;                            // = z * th + ty
        tile = (r / th) * a.tiles_hw + (r % th) * a.tiles_w + txb;
    }
    const int z0 = tile / a.tiles_hw, t2 = tile - z0 * a.tiles_hw;
    if (a.pair_n > 0 && n >= a.pair_n) {               // second sample set of a pair launch (both directions of a BCRNN time sweep in one grid)
        n -= a.pair_n;
        a.x0 = a.x_b; a.addend = a.addend_b; a.y = a.y_b; a.accum = a.accum_b; a.accum_store = a.accum_store_b; a.gate = a.gate_b;
    }
    const int ty = t2 / a.tiles_w, tx = t2 - ty * a.tiles_w;
    const int r0 = ty * C::TH, c0 = tx * TW, co0 = blockIdx.y * C::COT;
    const float* wp = n >= a.set_split ? a.wp1 : a.wp0;
    const int q = lane & 15, kk = lane >> 4;
    int base_in[3];
#pragma unroll
    for (int dx = 0; dx < 3; ++dx) base_in[dx] = kk * C::PS + (wn * MT) * C::COLS + (q + dx - 1 + C::COLS) % C::COLS;
    const int base_w = kk * C::COTP + 16 * (wm * CT) + q;
    __builtin_amdgcn_s_setprio(2);
    CINE_STAMP_RT(9);
    CINE_STAMP(0);
    // ---- my slot: row srow, piece sj of channel group sg; the first / last piece of a row also carries the row's left / right halo column
    const int sg = tid / RP, srp = tid - sg * RP;
    const bool slot = sg < G;
    const int sgc = G == 1 ? 0 : min(sg, G - 1);
    const int srow = srp / PR, sj = srp % PR;
    const int gy = r0 - 1 + srow, gx = c0 + 4 * sj;
    const bool rowok = gy >= 0 && gy < a.H;
    const bool pok = rowok && gx < a.W;                              // my piece lies inside the plane (W % 4 == 0: as a whole)
    const int hx = sj == 0 ? c0 - 1 : c0 + TW;                       // my halo column (sj 0 / PR - 1 only)
    const bool hslot = slot && (sj == 0 || sj == PR - 1);
    const bool hok = rowok && hx >= 0 && hx < a.W;
    const long hwl = (long)a.H * a.W;
    const unsigned voff = (unsigned)(min(max(gy, 0), a.H - 1) * a.W + min(gx, a.W - 4)) * 4u;
    const unsigned hoff = (unsigned)(min(max(gy, 0), a.H - 1) * a.W + min(max(hx, 0), a.W - 1)) * 4u;
    float* const lrow = in_lds + sgc * C::PS + srow * C::COLS + 4 * sj;
    float* const lhalo = in_lds + sgc * C::PS + srow * C::COLS + (sj == 0 ? C::COLS - 1 : TW);

    auto chunk_cc = [&](int chunk) { return V3 ? chunk % a.ncc : chunk; };
    auto chunk_zs = [&](int chunk) { return V3 ? z0 + chunk / a.ncc - 1 : 0; };
    auto chunk_live = [&](int chunk) { return !V3 || (chunk_zs(chunk) >= 0 && chunk_zs(chunk) < a.D); };
    float4 wraw[NWT];
    float4 xraw[NCI];
    float hraw[NCI];
    auto issue = [&](int chunk) {
        if (V3 && !chunk_live(chunk)) return;
        const float* wsrc = wp + (long)chunk * 9 * CK * a.rowsp;
#pragma unroll
        for (int i = 0; i < NWT; ++i) {
            const int e = tid + i * NT;
            const int row = e / (C::COT / 4), c4 = (e % (C::COT / 4)) * 4;
            const bool v = e < 9 * CK * (C::COT / 4) && co0 + c4 < a.rowsp;
            wraw[i] = *reinterpret_cast<const float4*>(v ? wsrc + (long)row * a.rowsp + co0 + c4 : wp);
        }
        const int ci0 = chunk_cc(chunk) * CK;
        const bool first = ci0 < a.c0;
        const int cl0 = first ? ci0 : ci0 - a.c0;
        const int sc = first ? a.c0 : a.c1, sd = first ? a.d0 : a.d1;
        const int zsc = V3 ? min(chunk_zs(chunk), sd - 1) : 0;
        const size_t cstride = (size_t)sd * hwl * 4u;
        const char* sb = reinterpret_cast<const char*>(first ? a.x0 : a.x1) + (((size_t)n * sc + cl0) * sd + zsc) * hwl * 4u + (size_t)sgc * cstride;
        const int cmax = sc - 1 - cl0;
#pragma unroll
        for (int i = 0; i < NCI; ++i) {
            const int cku = G == 1 ? min(i, cmax) : i * G;
            xraw[i] = *reinterpret_cast<const float4*>(sb + (size_t)cku * cstride + voff);
            if (hslot) hraw[i] = *reinterpret_cast<const float*>(sb + (size_t)cku * cstride + hoff);
        }
    };
    const int nch = a.c0 + a.c1;
    float prec[MODE == 0 ? 1 : 12];
    const bool pfirst = tid < a.c0;
    const int pnp = pfirst ? a.np0 : a.np1;
    const int npm = max(a.np0, a.c1 > 0 ? a.np1 : 0);
    const float* const pp = MODE == 0 || tid >= nch ? nullptr
                            : (pfirst ? a.part0 + ((long)n * a.c0 + tid) * pnp * 3 : a.part1 + ((long)n * a.c1 + (tid - a.c0)) * pnp * 3);
    if constexpr (MODE != 0) {
        if (tid < nch && npm <= 4) {
#pragma unroll
            for (int i = 0; i < 12; ++i) prec[i] = pp[min(i, 3 * pnp - 1)];
        }
    }
    int first_live = 0;
    if (V3) while (first_live < a.nchunks && !chunk_live(first_live)) ++first_live;
    issue(first_live);
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (MODE != 0) {
        if (tid < nch) {
            float2 mr;
            if (npm <= 4) { float r4[12];
#pragma unroll
                for (int i = 0; i < 12; ++i) r4[i] = prec[i];
                mr = merge_loaded<4>(r4, pnp, a.eps);
            } else mr = merge_partials(pp, pnp, a.eps);          // one record per tile of a wide plane (52 for 208 x 208): the loop form
            st_lds[2 * tid] = mr.y; st_lds[2 * tid + 1] = -mr.x * mr.y;
        }
    }
    // ---- zeroed once: everything when the layer's only chunk is narrower than 8 channels; else my slots that lie outside the plane
    const bool narrow = a.cin % CK != 0;
    if (narrow) {
        for (int e = tid; e < C::IN_FLOATS; e += NT) in_lds[e] = 0.f;
    } else if (slot) {
        if (!pok) {
#pragma unroll
            for (int i = 0; i < NCI; ++i) *reinterpret_cast<float4*>(lrow + i * G * C::PS) = make_float4(0.f, 0.f, 0.f, 0.f);
        }
        if (hslot && !hok) {
#pragma unroll
            for (int i = 0; i < NCI; ++i) lhalo[i * G * C::PS] = 0.f;
        }
    }

    f32x4 acc[CT][MT];
#pragma unroll
    for (int ct = 0; ct < CT; ++ct)
#pragma unroll
        for (int f = 0; f < MT; ++f) acc[ct][f] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // The small-tile shape (the CRNN time-sweep steps: 4 fragments per wave, a dependent chain of 10-us launches) fetches its epilogue operands --
    // the addend P_t and, in back-propagation through time, the ReLU gate -- NOW, under the staging and the MFMA sweep, instead of behind them
    // (one exposed memory latency per step less); the 13-fragment shapes have no registers to spare for that.
    constexpr bool PRE = MT <= 4 && CT == 1 && !V3;
    float4 pre_add[PRE ? MT : 1], pre_gate[PRE ? MT : 1];
    if constexpr (PRE) {
        const int pfr0 = r0 + wn * MT, pgx0 = c0 + 4 * kk, pm = co0 + 16 * wm + q;
        const bool pok2 = pgx0 < a.W && pm < a.rows;
        const long pbase = (((long)n * a.rows + min(pm, a.rows - 1)) * a.D + z0) * hwl + min(pgx0, a.W - 4);
#pragma unroll
        for (int f = 0; f < MT; ++f) {
            const long off = pbase + (long)min(pfr0 + f, a.H - 1) * a.W;
            pre_add[f] = (a.addend && pok2) ? *reinterpret_cast<const float4*>(a.addend + off) : make_float4(0.f, 0.f, 0.f, 0.f);
            pre_gate[f] = (a.gate && pok2) ? *reinterpret_cast<const float4*>(a.gate + off) : make_float4(1.f, 1.f, 1.f, 1.f);
        }
    }
    CINE_STAMP(1);
    for (int chunk = first_live; chunk < a.nchunks; ++chunk) {
        if (V3 && !chunk_live(chunk)) break;           // the live chunks of a tile are one run: the dead ones behind it are skipped
        __syncthreads();
        if (chunk == first_live) CINE_STAMP(2);
#pragma unroll
        for (int i = 0; i < NWT; ++i) {
            const int e = tid + i * NT;
            if (e >= 9 * CK * (C::COT / 4)) break;
            const int row = e / (C::COT / 4), c4 = (e % (C::COT / 4)) * 4;
            *reinterpret_cast<float4*>(w_lds + row * C::COTP + c4) = co0 + c4 < a.rowsp ? wraw[i] : make_float4(0.f, 0.f, 0.f, 0.f);
        }
        const int ci0 = chunk_cc(chunk) * CK;
        const bool first = ci0 < a.c0;
        const bool srcok = !V3 || chunk_zs(chunk) < (first ? a.d0 : a.d1);       // a shorter `up` volume reads as zero behind its end (unet.py:106-120)
        if (slot) {
            const float* stp = st_lds + 2 * (ci0 + sgc);
            if (pok) {
#pragma unroll
                for (int i = 0; i < NCI; ++i) {
                    float4 o = xraw[i];
                    const bool gone = G == 1 && ci0 + i >= a.cin;       // uniform: the last chunk of a layer whose input is not a multiple of 8 channels
                    if constexpr (MODE == 1) {                          // (2-channel first layers; the CRNN's 16 + 2) -- zeros over the previous chunk's values
                        const float2 ss = *reinterpret_cast<const float2*>(stp + 2 * i * G);
                        act_piece<4>(reinterpret_cast<float*>(&o), ss.x, ss.y, a.slope);
                    }
                    if (!srcok || gone) o = make_float4(0.f, 0.f, 0.f, 0.f);
                    *reinterpret_cast<float4*>(lrow + i * G * C::PS) = o;
                }
            }
            if (hslot && hok) {
#pragma unroll
                for (int i = 0; i < NCI; ++i) {
                    float v = hraw[i];
                    if constexpr (MODE == 1) {
                        const float2 ss = *reinterpret_cast<const float2*>(stp + 2 * i * G);
                        v = act(v, ss.x, ss.y, a.slope);
                    }
                    lhalo[i * G * C::PS] = (srcok && !(G == 1 && ci0 + i >= a.cin)) ? v : 0.f;
                }
            }
        }
        if (chunk == first_live) CINE_STAMP(3);
        __syncthreads();
        if (chunk == first_live) CINE_STAMP(4);
        if (chunk + 1 < a.nchunks) issue(chunk + 1);
        __builtin_amdgcn_sched_barrier(0);
        {
            constexpr int KS = CK / 4, NG = 9 * KS;
            float af[2][CT], bf[2][MT];
            auto load_group = [&](int g, float (&wa)[CT], float (&xa)[MT]) {
                const int tap = g / KS, ks = g % KS;
                const int dy = tap / 3, dx = tap % 3;
#pragma unroll
                for (int ct = 0; ct < CT; ++ct) wa[ct] = w_lds[base_w + (tap * CK + 4 * ks) * C::COTP + 16 * ct];
#pragma unroll
                for (int f = 0; f < MT; ++f) xa[f] = in_lds[base_in[dx] + (4 * ks) * C::PS + (f + dy) * C::COLS];
            };
            __builtin_amdgcn_s_setprio(0);
            load_group(0, af[0], bf[0]);
#pragma unroll
            for (int g = 0; g < NG; ++g) {
                if (g + 1 < NG) load_group(g + 1, af[(g + 1) & 1], bf[(g + 1) & 1]);
#pragma unroll
                for (int ct = 0; ct < CT; ++ct)
#pragma unroll
                    for (int f = 0; f < MT; ++f)
                        acc[ct][f] = __builtin_amdgcn_mfma_f32_16x16x4f32(bf[g & 1][f], af[g & 1][ct], acc[ct][f], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
            __builtin_amdgcn_s_setprio(2);
        }
        if (chunk == first_live) CINE_STAMP(5);
    }
    CINE_STAMP(6);

    // ---- epilogue: fragment f = image row fr0 + f, my 4 pixels at columns gx0 .. gx0 + 3 (inside or outside the plane as a whole)
    const int fr0 = r0 + wn * MT, gx0 = c0 + 4 * kk;
    const bool colok = gx0 < a.W;
    if (a.bias) {
        const float* bsel = n >= a.set_split ? a.bias1 : a.bias;
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) {
            const int m = co0 + 16 * (wm * CT + ct) + q;
            const float bv = m < a.rows ? bsel[m] : 0.f;
#pragma unroll
            for (int f = 0; f < MT; ++f)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[ct][f][j] += bv;
        }
    }
    const bool full = r0 + C::TH <= a.H && c0 + TW <= a.W;
    if (a.addend || a.relu || a.gate) {
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) {
            const int m = co0 + 16 * (wm * CT + ct) + q;
            if (m >= a.rows) continue;
            const float* ab = a.addend ? a.addend + (((long)n * a.rows + m) * a.D + z0) * hwl : nullptr;
            const float* gt = a.gate ? a.gate + (((long)n * a.rows + m) * a.D + z0) * hwl : nullptr;
#pragma unroll
            for (int f = 0; f < MT; ++f) {
                if (fr0 + f >= a.H || !colok) continue;
                float4 t = make_float4(0.f, 0.f, 0.f, 0.f), gv = make_float4(1.f, 1.f, 1.f, 1.f);
                if constexpr (PRE) { t = pre_add[f]; gv = pre_gate[f]; }
                else {
                    if (ab) t = *reinterpret_cast<const float4*>(ab + (long)(fr0 + f) * a.W + gx0);
                    if (gt) gv = *reinterpret_cast<const float4*>(gt + (long)(fr0 + f) * a.W + gx0);
                }
                const float tv[4] = {t.x, t.y, t.z, t.w};
                const float gg[4] = {gv.x, gv.y, gv.z, gv.w};
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    float v = acc[ct][f][j];
                    if (ab) v += tv[j];
                    if (gt) v = gg[j] > 0.f ? v : 0.f;
                    acc[ct][f][j] = a.relu ? fmaxf(v, 0.f) : v;
                }
            }
        }
    }
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) {
        const int m = co0 + 16 * (wm * CT + ct) + q;
        if (m >= a.rows || !colok) continue;
        const long yoff = (((long)n * a.rows + m) * a.D + z0) * hwl + (long)fr0 * a.W + gx0;
        float* yb = a.y + yoff;
        float* ab2 = a.accum ? a.accum + yoff : nullptr;
#pragma unroll
        for (int f = 0; f < MT; ++f)
            if (full || fr0 + f < a.H) {
                const float4 o = make_float4(acc[ct][f][0], acc[ct][f][1], acc[ct][f][2], acc[ct][f][3]);
                *reinterpret_cast<float4*>(yb + (long)f * a.W) = o;
                if (ab2) {
                    float4 t = o;
                    if (!a.accum_store) { t = *reinterpret_cast<float4*>(ab2 + (long)f * a.W); t.x += o.x; t.y += o.y; t.z += o.z; t.w += o.w; }
                    *reinterpret_cast<float4*>(ab2 + (long)f * a.W) = t;
                }
            }
    }
    CINE_STAMP(7);
    if (a.ypart) {
        const int rows_w = min(max(a.H - fr0, 0), MT);
        const float cnt_w = (float)(rows_w * min(TW, a.W - c0));
        float mean_w[CT], m2_w[CT];
        auto wave_stats = [&](auto fullc) {
            constexpr bool FULL = decltype(fullc)::value;
#pragma unroll
            for (int ct = 0; ct < CT; ++ct) {
                float sacc = 0.f;
#pragma unroll
                for (int f = 0; f < MT; ++f)
#pragma unroll
                    for (int j = 0; j < 4; ++j) sacc += (FULL || (colok && fr0 + f < a.H)) ? acc[ct][f][j] : 0.f;
                sacc = add_xor16(sacc);
                sacc = add_xor32(sacc);
                mean_w[ct] = cnt_w > 0.f ? sacc / cnt_w : 0.f;
                float qacc = 0.f;
#pragma unroll
                for (int f = 0; f < MT; ++f)
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const float d = acc[ct][f][j] - mean_w[ct];
                        qacc += (FULL || (colok && fr0 + f < a.H)) ? d * d : 0.f;
                    }
                qacc = add_xor16(qacc);
                qacc = add_xor32(qacc);
                m2_w[ct] = qacc;
            }
        };
        if (full) wave_stats(std::true_type{}); else wave_stats(std::false_type{});
        auto merge_store = [&](const float (&rc)[WN], const float (&rm)[WN], const float (&rq)[WN], int row) {
            float cnt = 0.f, mean = 0.f;
#pragma unroll
            for (int w = 0; w < WN; ++w) { cnt += rc[w]; mean += rc[w] * rm[w]; }
            mean /= cnt;
            float m2 = 0.f;
#pragma unroll
            for (int w = 0; w < WN; ++w) {
                const float d = rm[w] - mean;
                m2 += rq[w] + rc[w] * d * d;
            }
            float* o = a.ypart + (((long)n * a.rows + co0 + row) * a.tiles + tile) * 3;
            o[0] = cnt; o[1] = mean; o[2] = m2;
        };
        if constexpr (WN == 1) {
            if (kk == 0) {
#pragma unroll
                for (int ct = 0; ct < CT; ++ct) {
                    const int row = 16 * (wm * CT + ct) + q;
                    if (co0 + row < a.rows) { const float rc[1] = {cnt_w}, rm[1] = {mean_w[ct]}, rq[1] = {m2_w[ct]}; merge_store(rc, rm, rq, row); }
                }
            }
        } else {
            float* red = st_lds + 2 * nch;
            if (kk == 0) {
#pragma unroll
                for (int ct = 0; ct < CT; ++ct) {
                    float* o = red + (wn * C::COT + 16 * (wm * CT + ct) + q) * 3;
                    o[0] = cnt_w; o[1] = mean_w[ct]; o[2] = m2_w[ct];
                }
            }
            __syncthreads();
            if (tid < C::COT && co0 + tid < a.rows) {
                float rc[WN], rm[WN], rq[WN];
#pragma unroll
                for (int w = 0; w < WN; ++w) { const float* r = red + (w * C::COT + tid) * 3; rc[w] = r[0]; rm[w] = r[1]; rq[w] = r[2]; }
                merge_store(rc, rm, rq, tid);
            }
        }
    }
    CINE_STAMP(8);
    CINE_STAMP_RT(10);
}

template <int CT, int WM, int WN, int MT, int MODE, int V3>
int launch_wide(const WideArgs& p, int n, hipStream_t st) {
    using C = ConvCfg<8, CT, WM, WN, MT, 16, 9>;
    auto kern = conv_wide_kernel<CT, WM, WN, MT, MODE, V3>;
    const size_t lds = C::lds_bytes(p.c0 + p.c1) + (WN > 1 ? C::RED_FLOATS * sizeof(float) : 0);
    static std::once_flag once[64];
    static hipError_t status[64];
    if (lds > 64 * 1024) {
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess) dev = 0;
        CINE_REQUIRE(dev >= 0 && dev < 64, CINE_EUNSUPPORTED, "conv_wide_kernel: device index %d", dev);
        std::call_once(once[dev], [&] {
            status[dev] = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        });
        CINE_REQUIRE(status[dev] == hipSuccess, CINE_EHIP, "conv_wide_kernel: hipFuncSetAttribute: %s", hipGetErrorString(status[dev]));
    }
    CINE_REQUIRE(lds <= 160 * 1024, CINE_EUNSUPPORTED, "conv_wide_kernel: %d input channels need %zu bytes of LDS", p.cin, lds);
    // the staging map reads channels g, g + G, ... of a chunk unconditionally when G > 1: such shapes take whole 8-channel chunks only
    CINE_REQUIRE(C::G == 1 || (p.cin % 8 == 0 && p.c0 % 8 == 0), CINE_EUNSUPPORTED,
                 "conv_wide_kernel: %d (+%d) input channels on a shape with %d channel groups", p.c0, p.c1, C::G);
    dim3 grid(p.tiles, ceil_div(p.rowsp, C::COT), n);
    WideArgs q = p;
#ifndef CINE_NO_XCD_BANDS
    if (grid.y == 1) {                 // (every shape this kernel runs today: its row block covers the layer's rows)
        const long bands = (long)n * p.D * (p.tiles_hw / p.tiles_w);
        const long ids = V3 ? 8L * ceil_div(bands, 8L) * p.tiles_w : (bands & ~7L) * p.tiles_w + 8L * ceil_div((bands & 7L) * p.tiles_w, 8L);
        if (bands > 0 && ids < (1L << 30)) { q.xcd_bands = (int)bands; grid = dim3((unsigned)ids, 1, 1); }
    }
#endif
    ProfScope prof(F_CONV3, st);
    hipLaunchKernelGGL(kern, grid, dim3(C::NT), lds, st, q);
    return check_launch("conv_wide_kernel");
}

}  // namespace

// general dispatcher -> wide kernel: 16-wide column tiles of 2-D planes (v3 = 0) or of volumes in the three-pass form (v3 = 1)
int launch_conv_wide(const ConvArgs& a, int ct, int wm, int wn, int mt, int v3, hipStream_t st, bool* handled) {
    *handled = false;
    if (!(g_plane_on & 4)) return CINE_OK;
    if (a.add_src1 || a.tconv_cout > 0 || a.n <= 0 || a.n > 65535) return CINE_OK;
    if ((a.accum || a.pair_n > 0) && (v3 || a.s1.c > 0 || a.s0.mode != 0)) return CINE_OK;
    if ((v3 != 0) != (a.vol != 0) || (!v3 && a.D != 1)) return CINE_OK;
    if (a.W <= 16 || a.W % 4 != 0) return CINE_OK;
    const Src& s0 = a.s0; const Src& s1 = a.s1;
    auto al16 = [](const void* p) { return reinterpret_cast<uintptr_t>(p) % 16 == 0; };
    if (!al16(a.y) || !al16(s0.x) || (s1.c > 0 && !al16(s1.x)) || (a.addend && !al16(a.addend)) || (a.accum && !al16(a.accum)) || (a.gate && !al16(a.gate))) return CINE_OK;
    if (a.pair_n > 0 && (!al16(a.x_b) || !al16(a.y_b) || (a.addend_b && !al16(a.addend_b)) || (a.accum_b && !al16(a.accum_b)) || !a.addend == !!a.addend_b || !a.gate == !!a.gate_b || (a.gate_b && !al16(a.gate_b)))) return CINE_OK;
    int mode;
    if (s0.mode == 0 && (s1.c == 0 || s1.mode == 0)) mode = 0;
    else if (s0.mode == 1 && (s1.c == 0 || s1.mode == 1)) mode = 1;
    else return CINE_OK;
    if (s0.w != a.W || s0.h != a.H || (s1.c > 0 && (s1.w != a.W || s1.h != a.H))) return CINE_OK;
    const int d0 = v3 ? s0.d : 1, d1 = v3 && s1.c > 0 ? s1.d : 1;
    if (v3 && (d0 < 1 || d0 > a.D || d1 < 1 || d1 > a.D)) return CINE_OK;
    if (mode == 1) {
        if (!s0.part || s0.np < 1 || (s1.c > 0 && (!s1.part || s1.np < 1))) return CINE_OK;
        if (s0.c + s1.c > 256) return CINE_OK;
    }
    const int ncc = v3 ? a.ncc : a.nchunks;
    // whole 8-channel chunks (two sources: the first one too); a narrower layer only as ONE chunk on the one-channel-group shape
    const bool whole = a.cin % 8 == 0 && (s1.c == 0 || s0.c % 8 == 0);
    // (mt == 13: ONE channel group, every thread stages channels 0 .. 7 of its slot and clamps the ones that do not exist; the 16-row
    //  shape has two groups whose second would read channels past the end of the tensor)
    const bool narrow_ok = ncc == 1 && s1.c == 0 && ct == 1 && wm == 1 && wn == 4 && (mt == 13 || mt == 10);
    // ... or as the LAST chunk of several on the 52-row shape (one channel group: every thread stages all 8 channels of its slot), the
    // first source ending on a chunk boundary: the CRNN's all-frame conv over cat(hidden 16, image 2)
    const bool ragged_ok = ct == 1 && wm == 1 && wn == 4 && (mt == 13 || mt == 10) && (s1.c == 0 || s0.c % 8 == 0);
    if (!whole && !narrow_ok && !ragged_ok) return CINE_OK;
    WideArgs p{};
    p.x0 = s0.x; p.part0 = s0.part; p.c0 = s0.c; p.np0 = s0.np; p.d0 = d0;
    p.x1 = s1.c > 0 ? s1.x : nullptr; p.part1 = s1.c > 0 ? s1.part : nullptr; p.c1 = s1.c; p.np1 = s1.c > 0 ? s1.np : 0; p.d1 = d1;
    p.wp0 = a.wp0; p.wp1 = a.wp1; p.set_split = a.set_split;
    p.bias = a.bias; p.bias1 = a.bias1; p.addend = a.addend; p.relu = a.relu;
    p.y = a.y; p.ypart = a.ypart; p.cin = a.cin; p.rows = a.rows; p.rowsp = a.rowsp; p.D = a.D; p.H = a.H; p.W = a.W;
    p.nchunks = a.nchunks; p.ncc = ncc; p.tiles = a.tiles; p.tiles_w = a.tiles_w; p.tiles_hw = a.tiles_hw;
    p.slope = a.slope; p.eps = a.eps;
    p.accum = a.accum; p.accum_store = a.accum_store; p.pair_n = a.pair_n; p.accum_store_b = a.accum_store_b;
    p.x_b = a.x_b; p.addend_b = a.addend_b; p.y_b = a.y_b; p.accum_b = a.accum_b; p.gate = a.gate; p.gate_b = a.gate_b;
#define CINE_WIDE_CASE(CT_, WM_, WN_, MT_)                                                                     \
    if (ct == CT_ && wm == WM_ && wn == WN_ && mt == MT_) {                                                     \
        *handled = true;                                                                                        \
        if (v3) return mode ? launch_wide<CT_, WM_, WN_, MT_, 1, 1>(p, a.n, st) : launch_wide<CT_, WM_, WN_, MT_, 0, 1>(p, a.n, st); \
        return mode ? launch_wide<CT_, WM_, WN_, MT_, 1, 0>(p, a.n, st) : launch_wide<CT_, WM_, WN_, MT_, 0, 0>(p, a.n, st);         \
    }
    CINE_WIDE_CASE(1, 1, 4, 13)
    CINE_WIDE_CASE(1, 2, 2, 13)
    CINE_WIDE_CASE(1, 2, 2, 7)                      // volumes of <= 32 rows whose 26-fragment tiling would leave the chip under-filled (vol_mid_tiles)
    CINE_WIDE_CASE(1, 4, 1, 13)
    CINE_WIDE_CASE(1, 1, 4, 4)                      // 16-row tiles of few planes: the CRNN cells' single-plane steps
    CINE_WIDE_CASE(1, 1, 4, 10)                     // 40-row tiles: the CRNN cells' all-frame convs when the 52-row tiling is one resident round plus a sliver
#undef CINE_WIDE_CASE
    return CINE_OK;
}

}  // namespace cine

// ================================================================ input gradient of the k2 s2 transpose conv (training)
// gx[ci][y][x] = sum_{co, a, b} W[ci][co][a][b] gy[co][2y + a][2x + b] (what autograd derives for unet.py:212-215): a 1x1 GEMM whose K
// dimension is the space-to-depth view of the output gradient (source mode 5 of conv_tile, which stages it element by element:
// 126 - 178 us per launch at cfg 2 for 1.4 GFLOP).  Here a K-chunk of 64 s2d channels is staged with two 16-byte loads per
// (channel, row parity) -- the even floats are the b = 0 piece, the odd ones the b = 1 piece -- prefetched one chunk ahead, and the
// weights stream from L2 as in tconv_plane_kernel.  Same K order as conv_tile: bit-identical.
namespace cine {
namespace {

struct S2dArgs {
    const float* g; const float* wp0; const float* wp1; int set_split;
    float* y;
    int cout, rows, rowsp, H, nk;            // cout: channels of g (K = 4 cout); rows: channels of y; H: rows of the LOW-resolution plane; nk: K chunks
};

template <int CT, int WM, int MT, int TW, int KC>       // KC: s2d channels per chunk = KC / 2 (channel, row parity) pairs
__global__ __launch_bounds__(64 * WM, 2) void s2d_gemm_plane_kernel(S2dArgs a) {
    constexpr int NT = 64 * WM;
    constexpr int TPX = 16 * MT, PS = ((TPX + 31) / 32) * 32 + 16, NV = TPX / 4, G = NT / NV;
    constexpr int NPAIR = KC / 2, NCI = (NPAIR + G - 1) / G;
    static_assert(G >= 1 && TPX % 4 == 0, "tile shape");
    extern __shared__ __align__(16) float smem_f[];
    float* in_lds = smem_f;                                // [KC][PS]
    const int tid = threadIdx.x, lane = tid & 63, wm = tid >> 6;
    const int tile = blockIdx.x, n = blockIdx.z;
    const int co0 = blockIdx.y * (16 * CT * WM);
    const int q = lane & 15, kk = lane >> 4;
    const float* wp = n >= a.set_split ? a.wp1 : a.wp0;
    __builtin_amdgcn_s_setprio(2);
    const int sg = tid / NV, sv = tid - sg * NV;
    const bool slot = sg < G;
    const int sgc = min(sg, G - 1);
    const int hw = a.H * TW;
    const int p0 = tile * TPX + 4 * sv;
    const bool pok = p0 < hw;
    const int pc = min(p0, hw - 4);
    // source offsets of my 4 low-resolution pixels inside one (channel, row parity 0) plane of g (2H x 2TW): TW >= 4: one row, 8 floats;
    // TW = 2: two rows of 4 floats
    const int y0 = pc / TW, x0 = pc % TW;
    const long o0 = (long)(2 * y0) * (2 * TW) + 2 * x0;
    const long o1 = TW >= 4 ? o0 + 4 : o0 + 2 * (2 * TW);
    const long gplane = (long)(2 * a.H) * (2 * TW);
    const float* gb = a.g + (long)n * a.cout * gplane;
    float4 xraw[NCI][2];
    auto issue = [&](int kc) {
#pragma unroll
        for (int i = 0; i < NCI; ++i) {
            const int pr = min(kc * NPAIR + sgc + i * G, 2 * a.cout - 1);        // (channel c, row parity s): pr = 2 c + s
            const float* src = gb + (long)(pr >> 1) * gplane + (long)(pr & 1) * (2 * TW);
            xraw[i][0] = *reinterpret_cast<const float4*>(src + o0);
            xraw[i][1] = *reinterpret_cast<const float4*>(src + o1);
        }
    };
    issue(0);
    const float* wl = wp + (long)kk * a.rowsp + co0 + 16 * (wm * CT) + q;
    constexpr int PD = 2;
    float wreg[PD][CT];
    const int ksn = 4 * a.cout / 4;                          // k-steps in all
    auto wload = [&](int ks, float (&w)[CT]) {
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) w[ct] = co0 + 16 * (wm * CT + ct) < a.rowsp ? wl[(long)(4 * ks) * a.rowsp + 16 * ct] : 0.f;
    };
#pragma unroll
    for (int d = 0; d < PD; ++d) wload(min(d, ksn - 1), wreg[d]);
    f32x4 acc[CT][MT];
#pragma unroll
    for (int ct = 0; ct < CT; ++ct)
#pragma unroll
        for (int f = 0; f < MT; ++f) acc[ct][f] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const float* ain = in_lds + kk * PS + q;
    for (int kc = 0; kc < a.nk; ++kc) {
        if (kc) __syncthreads();                            // the previous chunk's sweep is done with the tile
        if (slot) {
#pragma unroll
            for (int i = 0; i < NCI; ++i) {
                const int prl = sgc + i * G;                 // pair inside the chunk
                if (prl >= NPAIR) break;
                const bool ok = pok && kc * NPAIR + prl < 2 * a.cout;
                const float4 u = xraw[i][0], v = xraw[i][1];
                const float4 e = ok ? make_float4(u.x, u.z, v.x, v.z) : make_float4(0.f, 0.f, 0.f, 0.f);     // b = 0: even source columns
                const float4 o = ok ? make_float4(u.y, u.w, v.y, v.w) : make_float4(0.f, 0.f, 0.f, 0.f);     // b = 1
                *reinterpret_cast<float4*>(in_lds + (2 * prl) * PS + 4 * sv) = e;
                *reinterpret_cast<float4*>(in_lds + (2 * prl + 1) * PS + 4 * sv) = o;
            }
        }
        __syncthreads();
        if (kc + 1 < a.nk) issue(kc + 1);
        __builtin_amdgcn_s_setprio(0);
        const int ks0 = kc * (KC / 4), ks1 = min(ks0 + KC / 4, ksn);        // (both even: K = 4 cout, cout a multiple of 16)
        auto kstep = [&](int ks, float (&w)[CT]) {
            float wcur[CT];
#pragma unroll
            for (int ct = 0; ct < CT; ++ct) wcur[ct] = w[ct];
            if (ks + PD < ksn) wload(ks + PD, w);
            float xa[MT];
#pragma unroll
            for (int f = 0; f < MT; ++f) xa[f] = ain[(4 * (ks - ks0)) * PS + 16 * f];
#pragma unroll
            for (int ct = 0; ct < CT; ++ct)
#pragma unroll
                for (int f = 0; f < MT; ++f)
                    acc[ct][f] = __builtin_amdgcn_mfma_f32_16x16x4f32(xa[f], wcur[ct], acc[ct][f], 0, 0, 0);
        };
        for (int ks = ks0; ks < ks1; ks += 2) { kstep(ks, wreg[0]); kstep(ks + 1, wreg[1]); }
        __builtin_amdgcn_s_setprio(2);
    }
    // lane: output channel m, 4 consecutive pixels of fragment f
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) {
        const int m = co0 + 16 * (wm * CT + ct) + q;
        if (m >= a.rows) continue;
        float* yb = a.y + ((long)n * a.rows + m) * hw + (long)tile * TPX + 4 * kk;
#pragma unroll
        for (int f = 0; f < MT; ++f)
            if (tile * TPX + 16 * f + 4 * kk < hw)
                *reinterpret_cast<float4*>(yb + 16 * f) = make_float4(acc[ct][f][0], acc[ct][f][1], acc[ct][f][2], acc[ct][f][3]);
    }
}

template <int CT, int WM, int MT, int TW, int KC>
int launch_s2d(const S2dArgs& p, int n, int tiles, hipStream_t st) {
    constexpr int PS = ((16 * MT + 31) / 32) * 32 + 16;
    auto kern = s2d_gemm_plane_kernel<CT, WM, MT, TW, KC>;
    const size_t lds = (size_t)KC * PS * sizeof(float);
    const dim3 grid(tiles, ceil_div(p.rowsp, 16 * CT * WM), n);
    ProfScope prof(F_TCONV, st);
    hipLaunchKernelGGL(kern, grid, dim3(64 * WM), lds, st, p);
    return check_launch("s2d_gemm_plane_kernel");
}

}  // namespace

// general dispatcher -> input gradient of a 2-D transpose conv (TAPS = 1, source mode 5) with the pixel tile (mt, tw) it chose
int launch_tconv_dgrad_plane(const ConvArgs& a, int mt, int tw, hipStream_t st, bool* handled) {
    *handled = false;
    if (!(g_plane_on & 2)) return CINE_OK;
    if (a.vol || a.D != 1 || a.tconv_cout > 0 || a.bias || a.addend || a.relu || a.accum || a.ypart || a.s1.c > 0 || a.add_src1 || a.s0.mode != 5) return CINE_OK;
    if (a.W != tw || a.n <= 0 || a.n > 65535 || a.s0.w != 2 * a.W || a.s0.h != 2 * a.H || (a.H * a.W) % 4 != 0 || a.s0.c % 16 != 0) return CINE_OK;
    auto al16 = [](const void* p) { return reinterpret_cast<uintptr_t>(p) % 16 == 0; };
    if (!al16(a.y) || !al16(a.s0.x)) return CINE_OK;
    S2dArgs p{};
    p.g = a.s0.x; p.wp0 = a.wp0; p.wp1 = a.wp1; p.set_split = a.set_split; p.y = a.y;
    p.cout = a.s0.c; p.rows = a.rows; p.rowsp = a.rowsp; p.H = a.H;
    (void)mt;                                   // no statistics records: the pixel tiling is this kernel's own
#define CINE_S2D_CASE(ROWSP_, CT_, WM_, MT_, TW_, KC_)                                                      \
    if (a.rowsp == ROWSP_ && tw == TW_) {                                                                   \
        *handled = true;                                                                                   \
        p.nk = ceil_div(4 * a.s0.c, KC_);                                                                  \
        return launch_s2d<CT_, WM_, MT_, TW_, KC_>(p, a.n, ceil_div(a.H * a.W, 16 * MT_), st);              \
    }
    CINE_S2D_CASE(32, 1, 2, 13, 8, 32)         // level 1 <- level 0: 32 output channels = two waves of 16 rows, 4 tiles of 208 pixels
    CINE_S2D_CASE(64, 1, 4, 13, 4, 64)         // 64 channels, one tile per plane
    CINE_S2D_CASE(128, 2, 4, 4, 2, 64)         // 128
#undef CINE_S2D_CASE
    return CINE_OK;
}

}  // namespace cine
