// conv_kernels.hip -- U-Net building blocks for gfx950.
//
// One implicit-GEMM kernel on the fp32 matrix cores (v_mfma_f32_16x16x4_f32: exact fp32, the
// same numerics class as the reference's fp32 convolutions) serves
//   TAPS = 9 : conv3x3, pad 1, no bias             (unet.py:160,164)
//   TAPS = 1 : conv-transpose k2 s2 as a 1x1 GEMM with 4*cout (8*cout in 3-D) rows (unet.py:212-215),
//              and the final 1x1 conv + bias         (unet.py:69)
//   TAPS = 27: conv3x3x3 over (depth, h, w) volumes  (unet.py:48-49, dims = 3): one workgroup works on one
//              depth slice and stages the three input slices z-1, z, z+1
//     D[row][pixel] += W[row][(tap, cin)] * X[(tap, cin)][pixel]
//   M = 16 output rows, N = 16 pixels (one "fragment" = 16/TW rows x TW columns), K = 4 input
//   channels of one tap.
// One workgroup = WM x WN waves; it owns 16*CT*WM output rows x WN*MT fragments of ONE sample and
// walks the input channels in chunks of CK.  Per chunk the input tile (+halo) and the weight slab
// are staged in LDS, then every wave issues TAPS * CK/4 * CT * MT MFMAs whose operands come from
// ds_read_b32 at compile-time offsets.
//
// What is fused around the GEMM (so normalised activations never touch HBM):
//   on load  : InstanceNorm + LeakyReLU of the producer layer, 2x2 average pool (unet.py:97),
//              channel concat of two sources (unet.py:122), up-path zero pad (unet.py:106-120)
//   epilogue : InstanceNorm statistics of THIS layer's raw output as per-tile partials
//              {count, mean, M2} (exact two-pass in registers; consumers merge them with Chan's
//              formula -- deterministic, no atomics), bias for the 1x1 conv.
#include <algorithm>
#include <memory>
#include <mutex>
#include <type_traits>
#include <vector>
#include "common.h"
#include "conv_src.h"
#include "conv_cfg.h"

namespace cine {

constexpr int kPrio = 2;     // s_setprio level of the non-MFMA phases of conv_tile (the sweeps run at 0)
// One workgroup's share of one layer: tile `tile` (depth slice, tile row, tile column) of sample n, output-row block
// `coblk`.  Called once per workgroup by conv_mfma_kernel.
// V3 = 1 (with TAPS = 9): a 3x3x3 convolution over (depth, h, w) volumes as three 3x3 passes -- chunk = (depth offset dz, 8 input
// channels), staged from slice z + dz exactly like a 2-D plane (same LDS footprint, prefetch and vectorised transforms as the 2-D
// kernel), accumulated into the same MFMA accumulators; chunks whose slice lies outside the volume are skipped.
template <int CK, int CT, int WM, int WN, int MT, int TW, int TAPS, int V3 = 0>
__device__ __forceinline__ void conv_tile(const ConvArgs& a, const int tile, const int coblk, const int n, float* smem_f) {
    static_assert(V3 != 1 || TAPS == 9, "V3 = 1 rides on the 3x3 kernel");      // V3 = 2: volume addressing only (1x1x1 kinds)
    using C = ConvCfg<CK, CT, WM, WN, MT, TW, TAPS>;
    constexpr int HALO = C::HALO, PW = C::PW;
    typedef typename Piece<PW>::T piece_t;
    float* in_lds = smem_f;
    float* w_lds = smem_f + C::IN_FLOATS;
    float* st_lds = w_lds + C::W_FLOATS;            // {scale, shift} per input channel (see act())

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int z0 = tile / a.tiles_hw, t2 = tile - z0 * a.tiles_hw;
    const int ty = t2 / a.tiles_w, tx = t2 % a.tiles_w;
    const int r0 = ty * C::TH, c0 = tx * TW;
    const int co0 = coblk * C::COT;
    const float* wp = n >= a.set_split ? a.wp1 : a.wp0;
    const int q = lane & 15, kk = lane >> 4;
    const int qr = q / TW, qc = q % TW;
    // A operand (pixels x k): lane = pixel q of the fragment, channel kk of the k-step; one base per dx
    int base_in[3];
#pragma unroll
    for (int dx = 0; dx < 3; ++dx)
        base_in[dx] = kk * C::PS + (wn * MT * C::RPF + qr) * C::COLS + (HALO ? (qc + dx - 1 + C::COLS) % C::COLS : qc);
    // B operand (k x rows): lane = output row q of the 16-row tile, channel kk
    const int base_w = kk * C::COTP + 16 * (wm * CT) + q;

    __builtin_amdgcn_s_setprio(kPrio);
    CINE_STAMP_RT(9);
    CINE_STAMP(0);
    // ---- chunk pipeline.  issue(c) puts the global loads of chunk c (weight slab + raw input pieces) in
    // flight into registers; commit(c) applies the on-load transform and writes them to LDS.  issue(c+1)
    // is placed before the MFMA sweep of chunk c, so the memory latency of the next chunk hides under it.
    constexpr int NWT = C::NWT;
    float4 wraw[NWT];
    constexpr int KR = C::KR, G = C::G, NCI = C::NCI;
    piece_t xraw[C::NPF];
    const int sg = KR == 1 ? tid / C::RP : 0;         // my channel group of the staging map (a slot if < G)
    const int srp = KR == 1 ? tid - sg * C::RP : tid; // my (row, piece) slot
    const int sgc = G == 1 ? 0 : min(sg, G - 1);
    auto chunk_cc = [&](int chunk) { return V3 == 1 ? chunk % a.ncc : chunk; };     // channel chunk of a pipeline chunk
    auto chunk_zs = [&](int chunk) { return V3 == 1 ? z0 + chunk / a.ncc - 1 : (V3 == 2 ? z0 : 0); };     // the input slice it stages
    auto chunk_live = [&](int chunk) { return V3 != 1 || (chunk_zs(chunk) >= 0 && chunk_zs(chunk) < a.D); };
    auto chunk_src = [&](int chunk, int& cl0) -> const Src& {
        const int ci0 = chunk_cc(chunk) * CK;
        const bool first = ci0 < a.s0.c;
        cl0 = first ? ci0 : ci0 - a.s0.c;
        return first ? a.s0 : a.s1;
    };
    auto issue = [&](int chunk) {
        if (V3 == 1 && !chunk_live(chunk)) return;
        const float* wsrc = wp + (long)chunk * TAPS * CK * a.rowsp;
#pragma unroll
        for (int i = 0; i < NWT; ++i) {
            const int e = tid + i * C::NT;
            int row = e / (C::COT / 4);
            const int c4 = (e % (C::COT / 4)) * 4;
            const bool v = e < TAPS * CK * (C::COT / 4) && co0 + c4 < a.rowsp;
            if (TAPS == 27) {
                // 3-D weights are packed once, in the V3 order [dz][8-channel chunk][3x3 tap][8 channels][rows]: this kernel's
                // chunk of CK = 4 channels is half of an 8-channel chunk
                const int tap = row / CK, ck = row % CK, c = chunk * CK + ck;
                row = (((tap / 9) * a.ncc + c / 8) * 9 + tap % 9) * 8 + c % 8;
            }
            if (TAPS == 9 && CK == 4) row = (row / CK) * 8 + row % CK;      // 4-channel chunks (layers with <= 4 input channels) over the 8-channel packing
            wraw[i] = *reinterpret_cast<const float4*>(v ? (TAPS == 27 ? wp : wsrc) + (long)row * a.rowsp + co0 + c4 : wp);
        }
        int cl0;
        const Src& s = chunk_src(chunk, cl0);
        if (a.fast && !a.wav && s.mode != 2) {
            const int zsc = V3 ? min(chunk_zs(chunk), s.d - 1) : 0;     // V3: a plain / normalised source slice of a volume
            const char* sb = reinterpret_cast<const char*>(s.x + (((long)n * s.c + cl0) * (V3 ? s.d : 1) + zsc) * s.h * a.W);
            const unsigned cstride = (unsigned)((V3 ? s.d : 1) * s.h * a.W) * 4u;        // bytes between channels
            const int cmax = s.c - 1 - cl0;                              // last channel of this source, chunk-relative
#pragma unroll
            for (int k = 0; k < KR; ++k) {
                const int rp = srp + k * C::NT, row = rp / C::PR, j = rp % C::PR;
                const int gy = min(max(r0 - HALO + row, 0), s.h - 1), gx = min(c0 + PW * j, a.W - PW);
                const unsigned voff = (unsigned)(gy * a.W + gx) * 4u;
#pragma unroll
                for (int i = 0; i < NCI; ++i) {
                    const int ck = min(sgc + i * G, cmax);              // uniform when G == 1
                    if constexpr (V3 && PW == 4) {      // volume rows of any width (50, 25): 4-byte aligned 16-byte loads
                        const f4u t = *reinterpret_cast<const f4u*>(sb + (size_t)((unsigned)ck * cstride) + voff);
                        xraw[k * NCI + i] = make_float4(t.v[0], t.v[1], t.v[2], t.v[3]);
                    } else
                    xraw[k * NCI + i] = *reinterpret_cast<const piece_t*>(sb + (size_t)((unsigned)ck * cstride) + voff);
                }
            }
        }
    };
    // ---- the statistics records of the input channels go out FIRST (loads return in order: behind the first chunk's 16-byte
    // loads their latency would be paid twice), then the first chunk, then the merge arithmetic
    auto src_needs_stats = [](const Src& s) { return (s.mode == 1 || s.mode == 2) || (s.mode >= 3 && (s.act & 1)); };
    const int nch = a.s0.c + a.s1.c;
    const int npm = max(src_needs_stats(a.s0) ? a.s0.np : 0, a.s1.c > 0 && src_needs_stats(a.s1) ? a.s1.np : 0);   // uniform
    constexpr int NPQ = 16;                           // records per plane the early path holds in registers
    const bool early = npm <= NPQ && nch <= C::NT;
    float prec[3 * NPQ];
    const bool pfirst = tid < a.s0.c;
    const Src& psrc = pfirst ? a.s0 : a.s1;
    const bool pneed = early && tid < nch && src_needs_stats(psrc);
    if (pneed) {
        const float* pp = psrc.part + ((long)n * psrc.c + (pfirst ? tid : tid - a.s0.c)) * psrc.np * 3;
        if (npm <= 4) {
#pragma unroll
            for (int i = 0; i < 12; ++i) prec[i] = pp[min(i, 3 * psrc.np - 1)];
        } else load_partials<NPQ>(pp, psrc.np, prec);
    }
    CINE_STAMP(11);
    issue(0);
    __builtin_amdgcn_sched_barrier(0);
    CINE_STAMP(12);

    // ---- prologue: merged InstanceNorm stats of every input channel; zero the tile once
    // table layout: source 0's channels, then source 1's (for modes 0/1/2 that is the concat channel order)
    if (early) {
        if (tid < nch) {
            float2 mr = make_float2(0.f, 1.f);
            if (pneed) {
                if (npm <= 4) { float r4[12];
#pragma unroll
                    for (int i = 0; i < 12; ++i) r4[i] = prec[i];
                    mr = merge_loaded<4>(r4, psrc.np, a.eps);
                } else mr = merge_loaded<NPQ>(prec, psrc.np, a.eps);
            }
            st_lds[2 * tid] = mr.y; st_lds[2 * tid + 1] = -mr.x * mr.y;   // {scale, shift} of act()
        }
    } else
    for (int ci = tid; ci < nch; ci += C::NT) {
        const bool first = ci < a.s0.c;
        const Src& s = first ? a.s0 : a.s1;
        const int cl = first ? ci : ci - a.s0.c;
        float2 mr = make_float2(0.f, 1.f);
        if (src_needs_stats(s)) mr = merge_partials(s.part + ((long)n * s.c + cl) * s.np * 3, s.np, a.eps);
        st_lds[2 * ci] = mr.y; st_lds[2 * ci + 1] = -mr.x * mr.y;      // {scale, shift} of act()
    }
    CINE_STAMP(13);
    if (HALO) {   // halo columns stay zero for the whole kernel when the image is no wider than the tile
        for (int e = tid; e < CK * C::ZP * C::ROWS * 2; e += C::NT) {
            const int ck = e / (C::ZP * C::ROWS * 2), rem = e % (C::ZP * C::ROWS * 2);
            in_lds[ck * C::PS + (rem >> 1) * C::COLS + ((rem & 1) ? C::COLS - 1 : TW)] = 0.f;
        }
    }

    f32x4 acc[CT][MT];   // acc[ct][f][j] = out[row 16*ct + q][pixel 4*kk + j of fragment f]
#pragma unroll
    for (int ct = 0; ct < CT; ++ct)
#pragma unroll
        for (int f = 0; f < MT; ++f) acc[ct][f] = (f32x4){0.f, 0.f, 0.f, 0.f};

    CINE_STAMP(1);
    for (int chunk = 0; chunk < a.nchunks; ++chunk) {
        if (V3 == 1 && !chunk_live(chunk)) {               // slice outside the volume: nothing to add; keep the load pipeline going
            if (chunk + 1 < a.nchunks) issue(chunk + 1);
            continue;
        }
        __syncthreads();                              // stats table ready / previous sweep done with LDS
        if (chunk == 0) CINE_STAMP(2);
        // ---- commit: weight slab [tap][ck][COT] (packed layout [chunk][tap][ck][rowsp])
#pragma unroll
        for (int i = 0; i < NWT; ++i) {
            const int e = tid + i * C::NT;
            if (e >= TAPS * CK * (C::COT / 4)) break;
            const int row = e / (C::COT / 4), c4 = (e % (C::COT / 4)) * 4;
            *reinterpret_cast<float4*>(w_lds + row * C::COTP + c4) = co0 + c4 < a.rowsp ? wraw[i] : make_float4(0.f, 0.f, 0.f, 0.f);
        }
        const int ci0 = chunk_cc(chunk) * CK;
        const bool first = ci0 < a.s0.c;
        const int zs = chunk_zs(chunk);                 // V3: the volume slice this chunk stages (0 otherwise)
        int cl0;
        const Src& s = chunk_src(chunk, cl0);
        if (a.fast) {
            // ---- vectorised staging: every piece = PW consecutive floats of one (channel, row)
            if (a.wav) {
                // Haar wavelets on load (mwcnn.py:216-263).  DWT: a piece of PW outputs of band b, channel c comes from the
                // 2 x 2PW block of source channel c; IWT: from PW/2 pixels of the four source channels c + k C/4.  16 / 8-byte
                // loads, the butterflies in registers, same operation order as fetch_scalar (bit-identical results).
                const Src& w0 = a.s0;
#pragma unroll 2
                for (int i = 0; i < C::NPT; ++i) {
                    const int p = tid + i * C::NT;
                    if (p >= C::NPIECE) break;
                    const int ck = p / (C::ROWS * C::PR), rem = p % (C::ROWS * C::PR);
                    const int row = rem / C::PR, j = rem % C::PR;
                    const int gy = r0 - HALO + row, gx = c0 + PW * j;
                    const int ci = ci0 + ck;
                    float o[PW];
#pragma unroll
                    for (int u = 0; u < PW; ++u) o[u] = 0.f;
                    if (ci < a.cin && gy >= 0 && gy < a.H && gx < a.W) {
                        const bool actv = w0.act & 1;
                        if (w0.mode == 3) {
                            const int band = ci / w0.c, c = ci - band * w0.c;
                            const float sc = st_lds[2 * c], sh = st_lds[2 * c + 1];
                            const float* src = w0.x + (((long)n * w0.c + c) * w0.h + 2 * gy) * w0.w + 2 * gx;
                            float t0[2 * PW], t1[2 * PW];
#pragma unroll
                            for (int u = 0; u < 2 * PW; u += 4) {
                                *reinterpret_cast<float4*>(t0 + u) = *reinterpret_cast<const float4*>(src + u);
                                *reinterpret_cast<float4*>(t1 + u) = *reinterpret_cast<const float4*>(src + w0.w + u);
                            }
#pragma unroll
                            for (int u = 0; u < PW; ++u) {
                                float x1 = t0[2 * u], x3 = t0[2 * u + 1], x2 = t1[2 * u], x4 = t1[2 * u + 1];
                                if (actv) { x1 = act(x1, sc, sh, a.slope); x2 = act(x2, sc, sh, a.slope); x3 = act(x3, sc, sh, a.slope); x4 = act(x4, sc, sh, a.slope); }
                                x1 *= 0.5f; x2 *= 0.5f; x3 *= 0.5f; x4 *= 0.5f;
                                o[u] = band == 0 ? x1 + x2 + x3 + x4 : band == 1 ? -x1 - x2 + x3 + x4 : band == 2 ? -x1 + x2 - x3 + x4 : x1 - x2 - x3 + x4;
                            }
                        } else {
                            const int cq = w0.c / 4, sy = gy >> 1, sx = gx >> 1;
                            const bool ry = gy & 1;
                            constexpr int NS = PW / 2;
                            float v[4][NS];
#pragma unroll
                            for (int k = 0; k < 4; ++k) {
                                const int c = ci + k * cq;
                                const float* src = w0.x + (((long)n * w0.c + c) * w0.h + sy) * w0.w + sx;
                                if (NS == 2) { const float2 t = *reinterpret_cast<const float2*>(src); v[k][0] = t.x; v[k][NS - 1] = t.y; }
                                else v[k][0] = src[0];
                                const float sc = st_lds[2 * c], sh = st_lds[2 * c + 1];
#pragma unroll
                                for (int e = 0; e < NS; ++e) v[k][e] = 0.5f * (actv ? act(v[k][e], sc, sh, a.slope) : v[k][e]);
                            }
#pragma unroll
                            for (int e = 0; e < NS; ++e) {
                                o[2 * e] = ry ? v[0][e] - v[1][e] + v[2][e] - v[3][e] : v[0][e] - v[1][e] - v[2][e] + v[3][e];
                                o[2 * e + 1] = ry ? v[0][e] + v[1][e] + v[2][e] + v[3][e] : v[0][e] + v[1][e] - v[2][e] - v[3][e];
                            }
                        }
                        if (a.add_src1) {                           // additive skip (mwcnn.py:164,172): same channel, same pixel
                            const Src& w1 = a.s1;
                            const float* q1 = w1.x + (((long)n * w1.c + ci) * w1.h + gy) * w1.w + gx;
                            piece_t t = *reinterpret_cast<const piece_t*>(q1);
                            const float* tv = reinterpret_cast<const float*>(&t);
                            const float sc = st_lds[2 * (w0.c + ci)], sh = st_lds[2 * (w0.c + ci) + 1];
#pragma unroll
                            for (int u = 0; u < PW; ++u) o[u] += w1.mode == 0 ? tv[u] : act(tv[u], sc, sh, a.slope);
                        }
                    }
                    piece_t ov;
                    float* ovf = reinterpret_cast<float*>(&ov);
#pragma unroll
                    for (int u = 0; u < PW; ++u) ovf[u] = o[u];
                    *reinterpret_cast<piece_t*>(in_lds + ck * C::PS + row * C::COLS + PW * j) = ov;
                }
            } else if (s.mode != 2) {
                const bool plain = s.mode == 0, fullchunk = ci0 + CK <= a.cin;
#pragma unroll
                for (int k = 0; k < KR; ++k) {
                    const int rp = srp + k * C::NT, row = rp / C::PR, j = rp % C::PR;
                    const int gy = r0 - HALO + row, gx = c0 + PW * j;
                    const bool slot = rp < C::RP && sg < G;
                    const bool rowok = gy >= 0 && gy < s.h && gx < a.W && (!V3 || zs < s.d);
                    // V3, width not a multiple of the piece: the last piece of a row was loaded from W - PW (issue() clamps the
                    // start), i.e. shifted by rsh elements; its tail beyond the row is zero
                    const int rsh = V3 ? max(gx + PW - a.W, 0) : 0;
                    float* lrow = in_lds + sgc * C::PS + row * C::COLS + PW * j;
                    const float* stp = st_lds + 2 * (ci0 + sgc);
                    // wave-uniform: every slot of this wave is inside the image and the chunk has all CK channels
                    if (fullchunk && __builtin_amdgcn_ballot_w64(slot && (!rowok || rsh > 0)) == 0) {
                        if (slot) {
#pragma unroll
                            for (int i = 0; i < NCI; ++i) {
                                piece_t o = xraw[k * NCI + i];
                                if (!plain) {
                                    const float2 ss = *reinterpret_cast<const float2*>(stp + 2 * i * G);
                                    float* ov = reinterpret_cast<float*>(&o);
                                    act_piece<PW>(ov, ss.x, ss.y, a.slope);
                                }
                                *reinterpret_cast<piece_t*>(lrow + i * G * C::PS) = o;
                            }
                        }
                    } else if (slot) {
#pragma unroll
                        for (int i = 0; i < NCI; ++i) {
                            const bool ok = rowok && ci0 + sgc + i * G < a.cin;
                            piece_t o = xraw[k * NCI + i];
                            float* ov = reinterpret_cast<float*>(&o);
                            const float2 ss = *reinterpret_cast<const float2*>(stp + 2 * i * G);
                            if (V3 && rsh > 0) {            // element u of the piece sits at position u + rsh of the loaded one
                                float t[PW];
#pragma unroll
                                for (int u = 0; u < PW; ++u) t[u] = ov[u];
#pragma unroll
                                for (int u = 0; u < PW; ++u) {
                                    float v = 0.f;
#pragma unroll
                                    for (int q2 = u; q2 < PW; ++q2) v = (q2 == u + rsh) ? t[q2] : v;
                                    ov[u] = v;
                                }
                            }
#pragma unroll
                            for (int u = 0; u < PW; ++u) ov[u] = ok && (!V3 || gx + u < a.W) ? (plain ? ov[u] : act(ov[u], ss.x, ss.y, a.slope)) : 0.f;
                            *reinterpret_cast<piece_t*>(lrow + i * G * C::PS) = o;
                        }
                    }
                }
            } else {
                // pooled source (extent 2H x 2W): 2*PW floats from each of two rows per piece
#pragma unroll 2
                for (int i = 0; i < C::NPT; ++i) {
                    const int p = tid + i * C::NT;
                    if (p >= C::NPIECE) break;
                    const int ck = p / (C::ROWS * C::PR), rem = p % (C::ROWS * C::PR);
                    const int row = rem / C::PR, j = rem % C::PR;
                    const int gy = r0 - HALO + row, gx = c0 + PW * j;
                    const bool ok = ci0 + ck < a.cin && gy >= 0 && 2 * gy + 1 < s.h && gx < a.W && (!V3 || 2 * zs + 1 < s.d);
                    float* dst = in_lds + ck * C::PS + row * C::COLS + PW * j;
                    if (V3 && ok && gx + PW > a.W) {        // ragged last piece of a row: element by element
#pragma unroll
                        for (int u = 0; u < PW; ++u)
                            dst[u] = gx + u < a.W ? fetch_scalar(s, n, cl0 + ck, zs, gy, gx + u, st_lds + (first ? 0 : 2 * a.s0.c), a.slope) : 0.f;
                    } else if (ok) {
                        const float mean = st_lds[2 * (ci0 + ck)], rstd = st_lds[2 * (ci0 + ck) + 1];
                        if (V3) {       // avg_pool3d 2x2x2 (unet.py:88,97): two source slices, same summation order as fetch_scalar
                            float acc8[PW];
#pragma unroll
                            for (int u = 0; u < PW; ++u) acc8[u] = 0.f;
#pragma unroll
                            for (int dzz = 0; dzz < 2; ++dzz) {
                                const float* src = s.x + ((((long)n * s.c + cl0 + ck) * s.d + 2 * zs + dzz) * s.h + 2 * gy) * s.w + 2 * gx;
                                float t0[2 * PW], t1[2 * PW];
#pragma unroll
                                for (int u = 0; u < 2 * PW; u += 4) {
                                    *reinterpret_cast<f4u*>(t0 + u) = *reinterpret_cast<const f4u*>(src + u);
                                    *reinterpret_cast<f4u*>(t1 + u) = *reinterpret_cast<const f4u*>(src + s.w + u);
                                }
#pragma unroll
                                for (int u = 0; u < PW; ++u)
                                    acc8[u] += act(t0[2 * u], mean, rstd, a.slope) + act(t0[2 * u + 1], mean, rstd, a.slope) +
                                               act(t1[2 * u], mean, rstd, a.slope) + act(t1[2 * u + 1], mean, rstd, a.slope);
                            }
#pragma unroll
                            for (int u = 0; u < PW; ++u) dst[u] = 0.125f * acc8[u];
                        } else {
                        const float* src = s.x + (((long)n * s.c + cl0 + ck) * s.h + 2 * gy) * s.w + 2 * gx;
                        float t0[2 * PW], t1[2 * PW];
#pragma unroll
                        for (int u = 0; u < 2 * PW; u += 4) {
                            *reinterpret_cast<float4*>(t0 + u) = *reinterpret_cast<const float4*>(src + u);
                            *reinterpret_cast<float4*>(t1 + u) = *reinterpret_cast<const float4*>(src + s.w + u);
                        }
#pragma unroll
                        for (int u = 0; u < PW; ++u)
                            dst[u] = 0.25f * (act(t0[2 * u], mean, rstd, a.slope) + act(t0[2 * u + 1], mean, rstd, a.slope) +
                                              act(t1[2 * u], mean, rstd, a.slope) + act(t1[2 * u + 1], mean, rstd, a.slope));
                        }
                    } else {
#pragma unroll
                        for (int u = 0; u < PW; ++u) dst[u] = 0.f;
                    }
                }
            }
            // halo columns (only exist as data when the image is wider than the tile)
            if (HALO && a.W > TW) {
                for (int e = tid; e < CK * C::ROWS * 2; e += C::NT) {
                    const int ck = e / (C::ROWS * 2), rem = e % (C::ROWS * 2);
                    const int row = rem >> 1, side = rem & 1;
                    const int gy = r0 - 1 + row, gx = side ? c0 + TW : c0 - 1;
                    float v = 0.f;
                    if (ci0 + ck < a.cin && gy >= 0 && gy < a.H && gx >= 0 && gx < a.W)
                        v = fetch_scalar(s, n, cl0 + ck, V3 ? zs : 0, gy, gx, st_lds + (first ? 0 : 2 * a.s0.c), a.slope);
                    in_lds[ck * C::PS + row * C::COLS + (side ? TW : C::COLS - 1)] = v;
                }
            }
        } else if (TAPS == 27 && a.vfast) {
            // ---- volumes: one unit = 4 consecutive voxels of one (channel, depth slice, row) -- a 16-byte load (eight for a
            // 2x2x2-pooled source) -- or that row's two halo columns.  Same operation order as fetch_scalar (bit-identical).
            // (Measured and rejected: batches of four units with all loads issued first, and prefetching the pieces one chunk
            // ahead like the 2-D path -- both spill 100-1000 registers in the 27-tap instantiations and run slower.)
            constexpr int NU = CK * C::ZP * C::ROWS * (C::PR + 1);
            const int c0n = src_cin(a.s0);
            for (int un = tid; un < NU; un += C::NT) {
                const int j = un % (C::PR + 1), r2 = un / (C::PR + 1);
                const int ck = r2 / (C::ZP * C::ROWS), rem = r2 - ck * (C::ZP * C::ROWS);
                const int zp = rem / C::ROWS, row = rem - zp * C::ROWS;
                const int ci = ci0 + ck, gz = z0 + zp - 1, gy = r0 - HALO + row;
                const bool f0 = ci < c0n;
                const Src& v = f0 ? a.s0 : a.s1;
                const int cl = f0 ? ci : ci - c0n;
                const float* stp = st_lds + (f0 ? 0 : 2 * a.s0.c);
                const bool inb = ci < a.cin && gz >= 0 && gz < a.D && gy >= 0 && gy < a.H;
                float* lrow = in_lds + ck * C::PS + zp * C::ZS + row * C::COLS;
                if (j < C::PR) {
                    const int gx = c0 + 4 * j;
                    float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
                    if (inb && gx < a.W) {
                        const float sc = stp[2 * cl], sh = stp[2 * cl + 1];
                        const long plane = (long)n * v.c + cl;
                        const bool whole = v.mode == 2 ? (2 * gx + 8 <= v.w && gx + 4 <= a.W) : (gx + 4 <= a.W && gx + 4 <= v.w);
                        if (!whole) {                               // ragged right edge (width not a multiple of 4, or a narrower
                            float ov[4];                            // `up` source): element by element
#pragma unroll
                            for (int u = 0; u < 4; ++u) ov[u] = gx + u < a.W ? fetch_scalar(v, n, cl, gz, gy, gx + u, stp, a.slope) : 0.f;
                            o = make_float4(ov[0], ov[1], ov[2], ov[3]);
                        } else if (v.mode == 2) {
                            if (2 * gz + 1 < v.d && 2 * gy + 1 < v.h) {       // avg_pool3d(2, 2) floors (unet.py:88,97)
                                float acc8[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                                for (int dz = 0; dz < 2; ++dz) {
                                    const float* p0 = v.x + ((plane * v.d + 2 * gz + dz) * v.h + 2 * gy) * v.w + 2 * gx;
                                    float t0[8], t1[8];
#pragma unroll
                                    for (int u = 0; u < 8; u += 4) {
                                        *reinterpret_cast<f4u*>(t0 + u) = *reinterpret_cast<const f4u*>(p0 + u);
                                        *reinterpret_cast<f4u*>(t1 + u) = *reinterpret_cast<const f4u*>(p0 + v.w + u);
                                    }
#pragma unroll
                                    for (int u = 0; u < 4; ++u)
                                        acc8[u] += act(t0[2 * u], sc, sh, a.slope) + act(t0[2 * u + 1], sc, sh, a.slope) +
                                                   act(t1[2 * u], sc, sh, a.slope) + act(t1[2 * u + 1], sc, sh, a.slope);
                                }
                                o = make_float4(0.125f * acc8[0], 0.125f * acc8[1], 0.125f * acc8[2], 0.125f * acc8[3]);
                            }
                        } else if (gz < v.d && gy < v.h) {
                            const f4u t = *reinterpret_cast<const f4u*>(v.x + ((plane * v.d + gz) * v.h + gy) * v.w + gx);
                            o = make_float4(t.v[0], t.v[1], t.v[2], t.v[3]);
                            if (v.mode == 1)
                                o = make_float4(act(o.x, sc, sh, a.slope), act(o.y, sc, sh, a.slope), act(o.z, sc, sh, a.slope), act(o.w, sc, sh, a.slope));
                        }
                    }
                    *reinterpret_cast<float4*>(lrow + 4 * j) = o;
                } else if (HALO) {
#pragma unroll
                    for (int side = 0; side < 2; ++side) {
                        const int gx = side ? c0 + TW : c0 - 1;
                        float hv = 0.f;
                        if (inb && gx >= 0 && gx < a.W) hv = fetch_scalar(v, n, cl, gz, gy, gx, stp, a.slope);
                        lrow[side ? TW : C::COLS - 1] = hv;
                    }
                }
            }
        } else {
            // ---- generic scalar staging (odd widths, mixed-source chunks, narrow `up` extents)
            constexpr int XC = TW + 2 * HALO;               // image columns c0-HALO .. c0+TW-1+HALO
            for (int e = tid; e < CK * C::ZP * C::ROWS * XC; e += C::NT) {
                const int ck = e / (C::ZP * C::ROWS * XC);
                const int rem = e - ck * (C::ZP * C::ROWS * XC);
                const int zp = rem / (C::ROWS * XC), rem2 = rem - zp * (C::ROWS * XC);
                const int row = rem2 / XC, xcol = rem2 - row * XC;
                const int col = (xcol - HALO + C::COLS) % C::COLS;
                const int ci = ci0 + ck;
                const int gz = V3 ? zs : z0 + zp - (C::ZP == 3 ? 1 : 0), gy = r0 - HALO + row, gx = c0 - HALO + xcol;
                float v = 0.f;
                if (ci < a.cin && gz >= 0 && gz < a.D && gy >= 0 && gy < a.H && gx >= 0 && gx < a.W) {
                    const int c0n = src_cin(a.s0);
                    if (a.add_src1) {
                        v = fetch_scalar(a.s0, n, ci, gz, gy, gx, st_lds, a.slope) +
                            fetch_scalar(a.s1, n, ci, gz, gy, gx, st_lds + 2 * a.s0.c, a.slope);
                    } else {
                        const bool f0 = ci < c0n;
                        v = fetch_scalar(f0 ? a.s0 : a.s1, n, f0 ? ci : ci - c0n, gz, gy, gx, st_lds + (f0 ? 0 : 2 * a.s0.c), a.slope);
                    }
                }
                in_lds[ck * C::PS + zp * C::ZS + row * C::COLS + col] = v;
            }
        }
        if (chunk == 0) CINE_STAMP(3);
        __syncthreads();
        if (chunk == 0) CINE_STAMP(4);
        if (chunk + 1 < a.nchunks) issue(chunk + 1);
        __builtin_amdgcn_sched_barrier(0);
        // ---- MFMA sweep: TAPS * CK/4 operand groups, software-pipelined one group ahead.  The
        // scheduling barriers keep the compiler from hoisting every group's ds_reads to the top
        // (which costs > 100 extra VGPRs and with them half the occupancy).
        auto sweep = [&](auto ksc) {
            constexpr int KS = decltype(ksc)::value;   // k-steps (groups of 4 input channels) of this chunk that hold data
            constexpr int NG = TAPS * KS;
            float af[2][CT], bf[2][MT];
            auto load_group = [&](int g, float (&wa)[CT], float (&xa)[MT]) {
                const int tap = g / KS, ks = g % KS;
                const int dz = TAPS == 27 ? tap / 9 : 0;
                const int dy = TAPS == 1 ? 0 : (tap / 3) % 3, dx = TAPS == 1 ? 0 : tap % 3;
#pragma unroll
                for (int ct = 0; ct < CT; ++ct) wa[ct] = w_lds[base_w + (tap * CK + 4 * ks) * C::COTP + 16 * ct];
#pragma unroll
                for (int f = 0; f < MT; ++f)
                    xa[f] = in_lds[base_in[dx] + (4 * ks) * C::PS + dz * C::ZS + (f * C::RPF + dy) * C::COLS];
            };
            load_group(0, af[0], bf[0]);
#pragma unroll
            for (int g = 0; g < NG; ++g) {
                if (g + 1 < NG) load_group(g + 1, af[(g + 1) & 1], bf[(g + 1) & 1]);
#pragma unroll
                for (int ct = 0; ct < CT; ++ct)
#pragma unroll
                    for (int f = 0; f < MT; ++f)   // M = pixels (bf), N = output rows (af)
                        acc[ct][f] = __builtin_amdgcn_mfma_f32_16x16x4f32(bf[g & 1][f], af[g & 1][ct], acc[ct][f], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
        };
        // The fp32 MFMA shares the SIMD's vector issue with ordinary vector instructions, and at equal priority a wave in a
        // vector-instruction phase (prologue, staging, epilogue) gets one instruction in per 32-cycle MFMA of a sweeping wave
        // on the same SIMD.  Sweeps run at the lowest priority, everything else above it: the short phases finish at their own
        // pace and the workgroup is back in a sweep sooner (the sweeping wave loses 4 cycles per such instruction either way).
        __builtin_amdgcn_s_setprio(0);
        sweep(std::integral_constant<int, CK / 4>{});
        __builtin_amdgcn_s_setprio(kPrio);
        if (chunk == 0) CINE_STAMP(5);
    }

    CINE_STAMP(6);
    // ---- epilogue.  Lane holds output row m = co0 + 16*(wm*CT+ct) + q at pixels 4*kk .. 4*kk+3 of
    // fragment f (4 consecutive pixels of one image row when TW >= 4).
    constexpr int PPR = TW >= 4 ? 4 : TW;             // consecutive pixels per image row held by a lane
    const int pr0 = (4 * kk) / TW, pc0 = (4 * kk) % TW;
    const int gx0 = c0 + pc0;
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
    // validity of my 4 pixels per fragment: bit (4 f + j); tiles inside the image skip the masking altogether
    const bool full = r0 + C::TH <= a.H && c0 + TW <= a.W;
    unsigned long long vmask = ~0ull;
    if (!full) {
        vmask = 0;
#pragma unroll
        for (int f = 0; f < MT; ++f)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int p = 4 * kk + j;
                const int gy = r0 + (wn * MT + f) * C::RPF + p / TW, gx = c0 + p % TW;
                if (gy < a.H && gx < a.W) vmask |= 1ull << (4 * f + j);
            }
    }
    if (a.addend || a.relu || a.gate) {          // CRNN cells: sum with a precomputed term, then ReLU (recurrent_varnet.py:172-178); gate: its adjoint step
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) {
            const int m = co0 + 16 * (wm * CT + ct) + q;
            if (m >= a.rows) continue;
            const float* ab = a.addend ? a.addend + (((long)n * a.rows + m) * a.D + z0) * a.H * a.W : nullptr;
            const float* gt = a.gate ? a.gate + (((long)n * a.rows + m) * a.D + z0) * a.H * a.W : nullptr;
#pragma unroll
            for (int f = 0; f < MT; ++f)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    if (!((vmask >> (4 * f + j)) & 1ull)) continue;
                    const int p = 4 * kk + j;
                    const int gy = r0 + (wn * MT + f) * C::RPF + p / TW, gx = c0 + p % TW;
                    float v = acc[ct][f][j];
                    if (ab) v += ab[(long)gy * a.W + gx];
                    if (gt) v = gt[(long)gy * a.W + gx] > 0.f ? v : 0.f;
                    acc[ct][f][j] = a.relu ? fmaxf(v, 0.f) : v;
                }
        }
    }
    // ---- the raw output goes out first: the stores drain while the statistics below are reduced
    // ---- store raw output: PPR consecutive pixels of one row per (ct, f [, half])
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) {
        const int m = co0 + 16 * (wm * CT + ct) + q;
        if (m >= a.rows) continue;
        if (a.tconv_cout > 0) {
            // row m = 2*(s*cout + co) + b with b the x-parity and s = a (2-D: output row parity) or 2*az + ay (3-D):
            // output (2z+az, 2y+ay, 2x+b)
            const int me = m >> 1, co = me % a.tconv_cout, sub = me / a.tconv_cout;
            const int sz = a.vol ? (sub >> 1) : 0, ay = a.vol ? (sub & 1) : sub;
            float* yb = a.y + (((long)n * a.tconv_cout + co) * (a.vol ? 2 * a.D : 1) + (a.vol ? 2 * z0 + sz : 0)) * (2 * a.H) * (2 * a.W);
            const bool odd = q & 1;
            if (a.tvec) {
                // lanes q, q^1 hold the two x-parities of the same 4 pixels: swap halves so that each lane owns
                // 4 consecutive output floats (even lane: pixels 0,1; odd lane: pixels 2,3 of the quad)
#pragma unroll
                for (int f = 0; f < MT; ++f) {
                    const float s0 = odd ? acc[ct][f][0] : acc[ct][f][2], s1 = odd ? acc[ct][f][1] : acc[ct][f][3];
                    const float t0 = __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(s0), 0xB1, 0xf, 0xf, true));   // quad_perm [1,0,3,2]
                    const float t1 = __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(s1), 0xB1, 0xf, 0xf, true));
                    const float4 o = odd ? make_float4(t0, acc[ct][f][2], t1, acc[ct][f][3]) : make_float4(acc[ct][f][0], t0, acc[ct][f][1], t1);
                    const int p = 4 * kk + (odd ? 2 : 0);
                    const int gy = r0 + (wn * MT + f) * C::RPF + p / TW, gx = c0 + p % TW;
                    if (gy < a.H && gx < a.W) *reinterpret_cast<float4*>(yb + (long)(2 * gy + ay) * (2 * a.W) + 2 * gx) = o;
                }
            } else {
#pragma unroll
                for (int f = 0; f < MT; ++f)
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        if (!((vmask >> (4 * f + j)) & 1ull)) continue;
                        const int p = 4 * kk + j;
                        const int gy = r0 + (wn * MT + f) * C::RPF + p / TW, gx = c0 + p % TW;
                        yb[(long)(2 * gy + ay) * (2 * a.W) + 2 * gx + (odd ? 1 : 0)] = acc[ct][f][j];
                    }
            }
            continue;
        }
        const long ybase = (((long)n * a.rows + m) * a.D + z0) * a.H * a.W;
        float* yb = a.y + ybase;
        float* ab2 = a.accum ? a.accum + ybase : nullptr;
        const bool vec = (a.W % PPR) == 0 && ((long)a.H * a.W) % PPR == 0;   // rows stay 16-byte (8-byte) aligned
#pragma unroll
        for (int f = 0; f < MT; ++f) {
#pragma unroll
            for (int hrow = 0; hrow < 4 / PPR; ++hrow) {
                const int gy = r0 + (wn * MT + f) * C::RPF + pr0 + hrow;
                const int j0 = hrow * PPR;
                if (!full && !((vmask >> (4 * f + j0)) & 1ull)) continue;
                const long off = (long)gy * a.W + gx0;
                float* dst = yb + off;
                if (vec) {
                    if (PPR == 4) {
                        const float4 o = make_float4(acc[ct][f][0], acc[ct][f][1], acc[ct][f][2], acc[ct][f][3]);
                        *reinterpret_cast<float4*>(dst) = o;
                        if (ab2) {
                            float4 t = o;
                            if (!a.accum_store) { t = *reinterpret_cast<float4*>(ab2 + off); t.x += o.x; t.y += o.y; t.z += o.z; t.w += o.w; }
                            *reinterpret_cast<float4*>(ab2 + off) = t;
                        }
                    } else {
                        const float2 o = make_float2(acc[ct][f][j0], acc[ct][f][j0 + 1]);
                        *reinterpret_cast<float2*>(dst) = o;
                        if (ab2) {
                            float2 t = o;
                            if (!a.accum_store) { t = *reinterpret_cast<float2*>(ab2 + off); t.x += o.x; t.y += o.y; }
                            *reinterpret_cast<float2*>(ab2 + off) = t;
                        }
                    }
                } else {
#pragma unroll
                    for (int u = 0; u < PPR; ++u)
                        if ((vmask >> (4 * f + j0 + u)) & 1ull) { dst[u] = acc[ct][f][j0 + u]; if (ab2) ab2[off + u] = a.accum_store ? acc[ct][f][j0 + u] : ab2[off + u] + acc[ct][f][j0 + u]; }
                }
            }
        }
    }
    CINE_STAMP(7);
    if (a.ypart) {
        // InstanceNorm partial {count, mean, M2} of this workgroup's pixels per output row: exact two-pass
        // per WAVE in registers (sum -> wave mean -> squared deviations), the WN wave records are merged
        // with Chan's formula by one thread per row.
        const int rows_w = min(max(a.H - (r0 + wn * MT * C::RPF), 0), MT * C::RPF);
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
                    for (int j = 0; j < 4; ++j) sacc += (FULL || ((vmask >> (4 * f + j)) & 1ull)) ? acc[ct][f][j] : 0.f;
                sacc += __shfl_xor(sacc, 16, 64);
                sacc += __shfl_xor(sacc, 32, 64);
                mean_w[ct] = cnt_w > 0.f ? sacc / cnt_w : 0.f;
                float qacc = 0.f;
#pragma unroll
                for (int f = 0; f < MT; ++f)
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const float d = acc[ct][f][j] - mean_w[ct];
                        qacc += (FULL || ((vmask >> (4 * f + j)) & 1ull)) ? d * d : 0.f;
                    }
                qacc += __shfl_xor(qacc, 16, 64);
                qacc += __shfl_xor(qacc, 32, 64);
                m2_w[ct] = qacc;
            }
        };
        if (full) wave_stats(std::true_type{}); else wave_stats(std::false_type{});
        __syncthreads();                            // everyone is done reading in_lds
        float* red = in_lds;                        // [WN][COT][3]
        if (kk == 0) {
#pragma unroll
            for (int ct = 0; ct < CT; ++ct) {
                float* o = red + (wn * C::COT + 16 * (wm * CT + ct) + q) * 3;
                o[0] = cnt_w; o[1] = mean_w[ct]; o[2] = m2_w[ct];
            }
        }
        __syncthreads();
        if (tid < C::COT && co0 + tid < a.rows) {
            const int m = co0 + tid;
            float cnt = 0.f, mean = 0.f;
#pragma unroll
            for (int w = 0; w < WN; ++w) { const float* r = red + (w * C::COT + tid) * 3; cnt += r[0]; mean += r[0] * r[1]; }
            mean /= cnt;
            float m2 = 0.f;
#pragma unroll
            for (int w = 0; w < WN; ++w) {
                const float* r = red + (w * C::COT + tid) * 3;
                const float d = r[1] - mean;
                m2 += r[2] + r[0] * d * d;
            }
            long slot;
            if (a.tconv_cout > 0) {
                const int nsub = a.vol ? 8 : 4;         // sub-positions of the k2 s2 transpose conv
                const int co = (m >> 1) % a.tconv_cout, ab = 2 * ((m >> 1) / a.tconv_cout) + (m & 1);
                slot = ((long)n * a.tconv_cout + co) * (a.tiles * nsub) + tile * nsub + ab;
            } else {
                slot = ((long)n * a.rows + m) * a.tiles + tile;
            }
            float* o = a.ypart + slot * 3;
            o[0] = cnt; o[1] = mean; o[2] = m2;
        }
    }
    CINE_STAMP(8);
    CINE_STAMP_RT(10);
}

// the volume forms' staging (depth offsets, ragged 4-byte-aligned pieces) needs more registers than ConvCfg's estimate: with the small-tile
// shapes' four waves per SIMD (128 VGPRs) the compiler spilled 172 B per thread in the chunk loop of the coarse 3-D U-Net levels, which
// run 12 - 156 workgroups per launch and gain nothing from occupancy
template <int CK, int CT, int WM, int WN, int MT, int TW, int TAPS, int V3>
constexpr int conv_minw() {
    constexpr int m = ConvCfg<CK, CT, WM, WN, MT, TW, TAPS>::MINW, cap = WM == 4 ? 2 : 3;      // WM == 4: the 64 / 128-row shapes of the coarse levels
    return (V3 != 0 && m > cap) ? cap : m;
}
template <int CK, int CT, int WM, int WN, int MT, int TW, int TAPS, int V3 = 0>
__global__ __launch_bounds__(64 * WM * WN, (conv_minw<CK, CT, WM, WN, MT, TW, TAPS, V3>())) void conv_mfma_kernel(ConvArgs a) {
    extern __shared__ __align__(16) float smem_f[];
    conv_tile<CK, CT, WM, WN, MT, TW, TAPS, V3>(a, blockIdx.x, blockIdx.y, blockIdx.z, smem_f);
}

// Two independent sample sets in one grid: samples [0, pair_n) use the ordinary pointers, samples [pair_n, n) the *_b set.
template <int CK, int CT, int WM, int WN, int MT, int TW, int TAPS>
__global__ __launch_bounds__(64 * WM * WN, (ConvCfg<CK, CT, WM, WN, MT, TW, TAPS>::MINW)) void conv_mfma_pair_kernel(ConvArgs a) {
    extern __shared__ __align__(16) float smem_f[];
    int n = blockIdx.z;
    if (n >= a.pair_n) {
        n -= a.pair_n;
        a.s0.x = a.x_b; a.addend = a.addend_b; a.y = a.y_b; a.accum = a.accum_b; a.accum_store = a.accum_store_b; a.gate = a.gate_b;
    }
    conv_tile<CK, CT, WM, WN, MT, TW, TAPS>(a, blockIdx.x, blockIdx.y, n, smem_f);
}

// ---------------------------------------------------------------- final 1x1 conv, few output channels
// The U-Net's last layer (unet.py:69) maps chans -> 2: no matrix-core work to speak of, one streaming read
// of the level-0 tensor.  Lanes run over 4-pixel groups, the channel loop keeps 8 16-byte loads in flight.
struct Conv1Args {
    const float* x; const float* part; int np, mode;
    const float* wp0; const float* wp1; const float* b0; const float* b1; int set_split;
    float* y; int cin, rowsp; long hw; float eps, slope;
};
// One sample's share of the layer: 4-pixel groups g0, g0 + gstride, ... of sample n.
template <int COUT>
__device__ __forceinline__ void conv1x1_tile(const Conv1Args& a, const int n, const long g0, const long gstride, float* smem_c1) {
    float* st = smem_c1;                 // {scale, shift} per input channel
    float* wl = smem_c1 + 2 * a.cin;     // [cin][COUT]
    const int tid = threadIdx.x;
    const bool second = n >= a.set_split;
    const float* wp = second ? a.wp1 : a.wp0;
    const float* bias = second ? a.b1 : a.b0;
    for (int ci = tid; ci < a.cin; ci += 256) {
        float2 mr = make_float2(0.f, 1.f);
        if (a.mode == 1) mr = merge_partials(a.part + ((long)n * a.cin + ci) * a.np * 3, a.np, a.eps);
        st[2 * ci] = mr.y; st[2 * ci + 1] = -mr.x * mr.y;
#pragma unroll
        for (int co = 0; co < COUT; ++co) wl[ci * COUT + co] = wp[(long)ci * a.rowsp + co];
    }
    __syncthreads();
    for (long g = g0; 4 * g < a.hw; g += gstride) {
        const float* xp = a.x + (long)n * a.cin * a.hw + 4 * g;
        float4 acc[COUT];
#pragma unroll
        for (int co = 0; co < COUT; ++co) { const float b = bias[co]; acc[co] = make_float4(b, b, b, b); }
        constexpr int U = 8;
        for (int c0 = 0; c0 < a.cin; c0 += U) {
            float4 v[U];
#pragma unroll
            for (int u = 0; u < U; ++u) v[u] = *reinterpret_cast<const float4*>(xp + (long)min(c0 + u, a.cin - 1) * a.hw);
#pragma unroll
            for (int u = 0; u < U; ++u) {
                if (c0 + u >= a.cin) break;
                const float sc = st[2 * (c0 + u)], sh = st[2 * (c0 + u) + 1];
                float4 t = v[u];
                if (a.mode == 1) { t.x = act(t.x, sc, sh, a.slope); t.y = act(t.y, sc, sh, a.slope); t.z = act(t.z, sc, sh, a.slope); t.w = act(t.w, sc, sh, a.slope); }
#pragma unroll
                for (int co = 0; co < COUT; ++co) {
                    const float w = wl[(c0 + u) * COUT + co];
                    acc[co].x = fmaf(w, t.x, acc[co].x); acc[co].y = fmaf(w, t.y, acc[co].y);
                    acc[co].z = fmaf(w, t.z, acc[co].z); acc[co].w = fmaf(w, t.w, acc[co].w);
                }
            }
        }
#pragma unroll
        for (int co = 0; co < COUT; ++co) *reinterpret_cast<float4*>(a.y + ((long)n * COUT + co) * a.hw + 4 * g) = acc[co];
    }
}
template <int COUT>
__global__ __launch_bounds__(256) void conv1x1_stream_kernel(Conv1Args a) {
    extern __shared__ __align__(16) float smem_c1[];
    conv1x1_tile<COUT>(a, blockIdx.y, (long)blockIdx.x * 256 + threadIdx.x, a.hw, smem_c1);
}

// ---------------------------------------------------------------- weight packing
// conv3x3 (cout, cin, 3, 3)        -> [chunk][tap][ck][rowsp]     rows = cout
// tconv   (cin, cout, 2, 2)        -> [chunk][1][ck][rowsp]       rows = 4*cout, row = 2*(a*cout + co) + b: the two x-parities
//                                     of one output channel sit in adjacent lanes, which pair up for 16-byte stores
// conv1x1 (cout, cin)              -> [chunk][1][ck][rowsp]       rows = cout
__device__ __forceinline__ void pack_weights_body(const float* w, float* p, int kind, int cout, int cin, int rows, int rowsp,
                                                  int taps, int ck_, int nchunks, long e0, long stride) {
    const long total = (long)nchunks * taps * ck_ * rowsp;
    for (long e = e0; e < total; e += stride) {
        const int m = (int)(e % rowsp);
        long r = e / rowsp;
        const int ck = (int)(r % ck_); r /= ck_;
        const int tap = (int)(r % taps);
        const int chunk = (int)(r / taps);
        const int ci = chunk * ck_ + ck;
        float v = 0.f;
        if (m < rows && ci < cin) {
            if (kind == 0) v = w[((long)m * cin + ci) * 9 + tap];
            else if (kind == 1) { const int b = m & 1, co = (m >> 1) % cout, a_ = (m >> 1) / cout; v = w[((long)ci * cout + co) * 4 + 2 * a_ + b]; }
            else if (kind == 3) v = w[((long)m * cin + ci) * 27 + tap];                                   // (conv3d: pack_conv3d_kernel)
            else if (kind == 4) { const int c_ = m & 1, co = (m >> 1) % cout, ab = (m >> 1) / cout; v = w[((long)ci * cout + co) * 8 + 2 * ab + c_]; }   // tconv3d
            else if (kind == 5) v = w[((long)ci * rows + m) * 9 + (8 - tap)];       // conv3x3 input gradient: w is (cin = K, rows, 3, 3), taps flipped
            else if (kind == 7) v = w[(long)ci * rows + m];                         // 1x1 input gradient: w is (K, rows)
            else v = w[(long)m * cin + ci];
        }
        p[e] = v;
    }
}
__global__ void pack_weights_kernel(const float* w, float* p, int kind, int cout, int cin, int rows, int rowsp,
                                    int taps, int ck_, int nchunks) {
    pack_weights_body(w, p, kind, cout, cin, rows, rowsp, taps, ck_, nchunks, (long)blockIdx.x * blockDim.x + threadIdx.x, (long)gridDim.x * blockDim.x);
}
// Every 2-D conv weight of a network in ONE launch (training re-packs after every optimiser step: 596 launches of 4 us per cfg-3 step): blockIdx.y
// picks a descriptor -- the arguments pack_weights_kernel would get for that tensor -- from device memory (cine_pack_desc / cine_pack_batch).
struct PackDesc { const float* w; float* p; int kind, cout, cin, rows, rowsp, taps, ck, nchunks; };
__global__ void pack_weights_batch_kernel(const PackDesc* __restrict__ descs) {
    const PackDesc d = descs[blockIdx.y];
    pack_weights_body(d.w, d.p, d.kind, d.cout, d.cin, d.rows, d.rowsp, d.taps, d.ck, d.nchunks, (long)blockIdx.x * blockDim.x + threadIdx.x, (long)gridDim.x * blockDim.x);
}

// conv3d (cout, cin, 3, 3, 3) -> [dz][8-channel chunk][3x3 tap][8 channels][rowsp]: the chunk order of the V3 kernels (the
// 27-tap kernel reads the same buffer through an index map)
__global__ void pack_conv3d_kernel(const float* w, float* p, int cout, int cin, int rowsp, int ncc) {
    const long total = 3L * ncc * 9 * 8 * rowsp;
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
        const int m = (int)(e % rowsp);
        long r = e / rowsp;
        const int ck = (int)(r % 8); r /= 8;
        const int t9 = (int)(r % 9); r /= 9;
        const int cc = (int)(r % ncc), dz = (int)(r / ncc);
        const int ci = cc * 8 + ck;
        p[e] = (m < cout && ci < cin) ? w[((long)m * cin + ci) * 27 + dz * 9 + t9] : 0.f;
    }
}

// ---------------------------------------------------------------- stand-alone statistics
// partial record {count, mean, M2} per plane (np = 1); exact two-pass.
__device__ __forceinline__ float wave_sum_f(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

__global__ void instnorm_partial_wave_kernel(const float* x, float* part, long planes, int pe) {
    const long plane = (long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (plane >= planes) return;
    const int lane = threadIdx.x & 63;
    const float* p = x + plane * pe;
    float s = 0.f;
    for (int i = lane; i < pe; i += 64) s += p[i];
    const float mean = wave_sum_f(s) / pe;
    float qv = 0.f;
    for (int i = lane; i < pe; i += 64) { const float d = p[i] - mean; qv += d * d; }
    qv = wave_sum_f(qv);
    if (lane == 0) { part[plane * 3] = (float)pe; part[plane * 3 + 1] = mean; part[plane * 3 + 2] = qv; }
}

__global__ void instnorm_partial_block_kernel(const float* x, float* part, long pe) {
    __shared__ float red[16];
    const long plane = blockIdx.x;
    const float* p = x + plane * pe;
    const int wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    float s = 0.f;
    for (long i = threadIdx.x; i < pe; i += blockDim.x) s += p[i];
    s = wave_sum_f(s);
    if ((threadIdx.x & 63) == 0) red[wave] = s;
    __syncthreads();
    float tot = 0.f;
    for (int i = 0; i < nw; ++i) tot += red[i];
    const float mean = tot / pe;
    __syncthreads();
    float qv = 0.f;
    for (long i = threadIdx.x; i < pe; i += blockDim.x) { const float d = p[i] - mean; qv += d * d; }
    qv = wave_sum_f(qv);
    if ((threadIdx.x & 63) == 0) red[wave] = qv;
    __syncthreads();
    if (threadIdx.x == 0) {
        float t2 = 0.f;
        for (int i = 0; i < nw; ++i) t2 += red[i];
        part[plane * 3] = (float)pe; part[plane * 3 + 1] = mean; part[plane * 3 + 2] = t2;
    }
}

// {mean, rstd} per plane from np partials
__global__ void instnorm_finalize_kernel(const float* part, float* stats, long planes, int np, float eps) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= planes) return;
    const float2 mr = merge_partials(part + i * np * 3, np, eps);
    stats[2 * i] = mr.x; stats[2 * i + 1] = mr.y;
}

__global__ void instnorm_lrelu_apply_kernel(const float* x, const float* part, int np, float* y, long planes, long pe,
                                            float eps, float slope) {
    const long total = planes * pe;
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
        const long plane = e / pe;
        const float2 mr = merge_partials(part + plane * np * 3, np, eps);
        y[e] = act(x[e], mr.y, -mr.x * mr.y, slope);
    }
}


// y (planes, d/2, h/2, w/2) = avg_pool3d(LeakyReLU(InstanceNorm(x)), 2) (unet.py:88,97 with dims = 3), fetch_scalar's summation order: the
// pooled level-1 input of the 3-D U-Net materialised once (38 MB read, 4.5 MB written at cfg 4) so that its consumer stages a plain
// tensor with 16-byte pieces; the coarse levels pool inside conv_coarse.hip and do not use this.  VEC: 4 outputs per thread.
template <bool VEC>
__global__ __launch_bounds__(256) void pool3d_act_kernel(const float* x, const float* part, int np, float* y, long planes,
                                                         int d, int h, int w, float eps, float slope) {
    const int od = d / 2, oh = h / 2, ow = w / 2;
    const int owv = VEC ? ow / 4 : ow;
    const long per = (long)od * oh * owv, total = planes * per;
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
        const long plane = e / per;
        long r = e - plane * per;
        const int ox = (int)(r % owv); r /= owv;
        const int oy = (int)(r % oh), oz = (int)(r / oh);
        const float2 mr = merge_partials(part + plane * np * 3, np, eps);
        const float sc = mr.y, sh = -mr.x * mr.y;
        const float* src = x + ((plane * d + 2 * oz) * h + 2 * oy) * (long)w + (VEC ? 8 * ox : 2 * ox);
        if constexpr (VEC) {
            float acc8[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int dz = 0; dz < 2; ++dz) {
                const float* p0 = src + (long)dz * h * w;
                float t0[8], t1[8];
#pragma unroll
                for (int u = 0; u < 8; u += 4) {
                    *reinterpret_cast<float4*>(t0 + u) = *reinterpret_cast<const float4*>(p0 + u);
                    *reinterpret_cast<float4*>(t1 + u) = *reinterpret_cast<const float4*>(p0 + w + u);
                }
#pragma unroll
                for (int u = 0; u < 4; ++u)
                    acc8[u] += act(t0[2 * u], sc, sh, slope) + act(t0[2 * u + 1], sc, sh, slope) + act(t1[2 * u], sc, sh, slope) + act(t1[2 * u + 1], sc, sh, slope);
            }
            *reinterpret_cast<float4*>(y + ((plane * od + oz) * oh + oy) * (long)ow + 4 * ox) =
                make_float4(0.125f * acc8[0], 0.125f * acc8[1], 0.125f * acc8[2], 0.125f * acc8[3]);
        } else {
            float acc8 = 0.f;
#pragma unroll
            for (int dz = 0; dz < 2; ++dz) {
                const float* p0 = src + (long)dz * h * w;
                acc8 += act(p0[0], sc, sh, slope) + act(p0[1], sc, sh, slope) + act(p0[w], sc, sh, slope) + act(p0[w + 1], sc, sh, slope);
            }
            y[((plane * od + oz) * oh + oy) * (long)ow + ox] = 0.125f * acc8;
        }
    }
}

// hipFuncAttributeMaxDynamicSharedMemorySize is per device: raise it once per (kernel, device) and KEEP the result -- a failed
// first call must fail every later launch with its own message instead of a generic launch error
template <typename K>
static int allow_big_lds(K kern, std::once_flag (&once)[64], const char* what) {
    static hipError_t status[64];                 // per template instance = per kernel
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) dev = 0;
    CINE_REQUIRE(dev >= 0 && dev < 64, CINE_EUNSUPPORTED, "%s: device index %d", what, dev);
    std::call_once(once[dev], [&] {
        status[dev] = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    });
    CINE_REQUIRE(status[dev] == hipSuccess, CINE_EHIP, "%s: hipFuncSetAttribute(MaxDynamicSharedMemorySize): %s", what, hipGetErrorString(status[dev]));
    return CINE_OK;
}


// ---------------------------------------------------------------- host dispatch
constexpr int kSmallMT = 4;  // 16-fragment tiles: 2 is ~5 % faster with ONE slice in flight, 4 is ~8 % faster with 12 (cfg 5: 286 -> 309 slices/s)
constexpr int kMT16 = 13;   // pixel fragments per wave of the <= 16-row configurations
constexpr int kWN16 = 4;   // waves (each 13 pixel fragments) per workgroup for <= 16 / <= 32 output rows
constexpr int kCK3 = 8;     // conv3x3: input channels per chunk
constexpr int kCK1 = 16;    // 1x1 / tconv: input channels per chunk
constexpr int kCK27 = 4;    // conv3x3x3: three input depth slices per channel live in LDS

template <int CK, int CT, int WM, int WN, int MT, int TW, int TAPS, bool PAIR = false, int V3 = 0>
static int launch_cfg(ConvArgs a, hipStream_t st) {
    using C = ConvCfg<CK, CT, WM, WN, MT, TW, TAPS>;
    static std::once_flag once[64];
    void (*kern)(ConvArgs);
    if constexpr (PAIR) kern = conv_mfma_pair_kernel<CK, CT, WM, WN, MT, TW, TAPS>;
    else kern = conv_mfma_kernel<CK, CT, WM, WN, MT, TW, TAPS, V3>;
    const size_t lds = C::lds_bytes(a.s0.c + a.s1.c);
    CINE_REQUIRE(lds <= 160 * 1024, CINE_EUNSUPPORTED, "conv: %d input channels need %zu bytes of LDS", a.cin, lds);
    if (lds > 64 * 1024)
        if (int e = allow_big_lds(kern, once, "conv_mfma_kernel")) return e;
    a.tiles_w = ceil_div(a.W, TW);
    a.tiles_hw = a.tiles_w * ceil_div(a.H, C::TH);
    a.tiles = a.tiles_hw * a.D;
    // vectorised staging preconditions (see kernel): widths multiple of the piece, one source per chunk
    const int PW = C::PW;
    auto src_ok = [&](const Src& s) {
        if (s.c == 0) return true;
        if (s.mode >= 3) return false;
        if (s.mode == 2) return (s.w == 2 * a.W || (V3 && s.w == 2 * a.W + 1)) && s.h >= 2 * a.H && (V3 || (s.w % 4) == 0);
        return s.w == a.W && s.h <= a.H;
    };
    // (V3: rows of any width >= one piece -- 4-byte aligned 16-byte loads, ragged last pieces element-wise)
    a.fast = (!a.vol || V3) && !a.add_src1 && (V3 ? a.W >= PW : a.W % PW == 0) && src_ok(a.s0) && src_ok(a.s1) && (a.s1.c == 0 || a.s0.c % CK == 0) &&
             (V3 || ((reinterpret_cast<uintptr_t>(a.s0.x) % 16 == 0) && (a.s1.c == 0 || reinterpret_cast<uintptr_t>(a.s1.x) % 16 == 0)));
    // volumes: 4-float row pieces (4-byte aligned loads: rows of any width), ragged right edges element by element
    a.vfast = a.vol && TAPS == 27 && PW == 4 && !a.add_src1 && a.s0.mode <= 2 && (a.s1.c == 0 || a.s1.mode <= 2);
    // Haar DWT / IWT source (+ added skip): whole-plane tiles in x (no halo columns to fetch), exact 2:1 extents, aligned rows
    a.wav = 0;
    if (!a.fast && !a.vol && TAPS == 9 && a.s0.mode >= 3 && a.W % PW == 0 && a.W <= TW && reinterpret_cast<uintptr_t>(a.s0.x) % 16 == 0) {
        const bool dwt_ok = a.s0.mode == 3 && a.s0.w == 2 * a.W && a.s0.h == 2 * a.H && (a.s0.w % 4) == 0;
        const bool iwt_ok = a.s0.mode == 4 && 2 * a.s0.w == a.W && 2 * a.s0.h == a.H && a.s0.c % 4 == 0 && (PW == 2 || a.s0.w % 2 == 0);
        const bool skip_ok = a.s1.c == 0 || (a.add_src1 && a.s1.mode <= 1 && a.s1.w == a.W && a.s1.h == a.H &&
                                            reinterpret_cast<uintptr_t>(a.s1.x) % 16 == 0);
        if ((dwt_ok || iwt_ok) && skip_ok) a.fast = a.wav = 1;
    }
    // (each lane stores the 4 output floats of 2 adjacent input pixels: an even width keeps the pairs whole and the rows 16-byte aligned)
    a.tvec = a.tconv_cout > 0 && a.W % 2 == 0 && a.H % 2 == 0 && reinterpret_cast<uintptr_t>(a.y) % 16 == 0;
    dim3 grid(a.tiles, ceil_div(a.rowsp, C::COT), a.n);
    const int fam = TAPS != 1 ? F_CONV3 : (a.tconv_cout > 0 ? F_TCONV : F_CONV1);
    if constexpr (TAPS == 1 && !PAIR && V3 == 0) {      // 2-D transpose conv of plane-wide tiles and its input gradient: conv_plane.hip, bit-identical
        if (a.tconv_cout > 0) {
            bool handled = false;
            const int e = launch_tconv_plane(a, MT, TW, st, &handled);
            if (e || handled) return e;
        } else if (a.s0.mode == 5) {
            bool handled = false;
            const int e = launch_tconv_dgrad_plane(a, MT, TW, st, &handled);
            if (e || handled) return e;
        }
    }
    if constexpr (TAPS == 9 && V3 <= 1 && CK == 8 && TW == 16) {      // wider planes / volumes in 16-wide column tiles (pair launches too): conv_plane.hip, bit-identical
        bool handled = false;
        const int e = launch_conv_wide(a, CT, WM, WN, MT, V3, st, &handled);
        if (e || handled) return e;
    }
    if constexpr (TAPS == 9 && !PAIR && V3 == 0) {       // plane-wide tiles of the U-Nets: the lean kernel (conv_plane.hip), bit-identical
        bool handled = false;
        const int e = launch_conv_plane(a, CK, CT, WM, WN, MT, TW, st, &handled);
        if (e || handled) return e;
    }
    ProfScope prof(fam, st);
    hipLaunchKernelGGL(kern, grid, dim3(C::NT), lds, st, a);
    return check_launch("conv_mfma_kernel");
}

// fragments per tile of the regular configurations (mirrored by tiles_for)
// plane3x3: a 2-D 3x3 convolution, whose small planes get tiles that fit them (MWCNN's coarse scales: a 50 x 4 plane is 13
// fragments, a 25 x 2 plane 4 -- in the 26- and 13-fragment tiles most MFMAs worked on padding)
static int regular_nf(int rowsp, long frags, bool plane3x3 = false) {
    if (plane3x3 && rowsp > 16 && rowsp <= 32 && frags <= 14) return 14;
    if (plane3x3 && rowsp > 32 && rowsp <= 64 && frags <= 4) return 4;
    if (rowsp <= 16) return kMT16 * kWN16;
    if (rowsp <= 32) return 26;
    if (rowsp <= 64 || frags > 8) return 13;
    return 4;
}
// volumes whose regular tiling gives fewer than 128 workgroups per sample (the rule depends on the layer shape only, so the
// statistics-record count of cine_conv_stat_partials3d stays a function of the shape)
constexpr int kVolSmall = 128;      // regular tiling with fewer workgroups than this -> 4-fragment tiles
static bool vol_small_tiles(int rowsp, int h, int w, int d) {
    const int TW = w > 8 ? 16 : w > 4 ? 8 : w > 2 ? 4 : 2;
    const long frags = (long)ceil_div(h * TW, 16) * ceil_div(w, TW);
    const int TH = regular_nf(rowsp, frags) * 16 / TW;
    return (long)ceil_div(w, TW) * ceil_div(h, TH) * d * ceil_div(rowsp, rowsp <= 64 ? rowsp : 128) < kVolSmall;
}

// volumes of 17 .. 32 output rows whose regular 26-fragment tiling gives 128 .. 255 workgroups per sample -- cfg 4's level 1 (32 channels on 7 x 100 x 100): 196
// workgroups on 768 resident slots, PMC MFMA 0.35 -- take 14-fragment tiles (twice the workgroups, 8 % more halo rows); again a rule on the shape only
constexpr int kVolMid = 256;
static bool vol_mid_tiles(int rowsp, int h, int w, int d) {
    if (rowsp <= 16 || rowsp > 32 || w <= 8 || vol_small_tiles(rowsp, h, w, d)) return false;
    return (long)ceil_div(w, 16) * ceil_div(h, 26) * d < kVolMid;
}

// 3x3x3 layers that run on conv_coarse.hip: a rule on the layer SHAPE only (the record count of cine_conv_stat_partials3d follows it)
static bool coarse_shape(int rowsp, int h, int w, int d) {
    return w > 8 && rowsp > 32 && vol_small_tiles(rowsp, h, w, d) && 32 + 2 * (w + 2) <= 192;
}
// ... and the 2-D planes that do: wider than the plane-wide tiles of conv_plane.hip, small, > 32 output rows (the sensitivity network's
// 26 x 26 x 64-channel level: round 4's conv_mfma_kernel<8, 1, 4, 1, 13, 16, 9, 0>, PMC MFMA 0.039)
static bool coarse_shape2d(int rowsp, int h, int w) {
    return w > 16 && rowsp > 32 && 32 + 2 * (w + 2) <= 192 && (long)h * (w + 1) <= 4096;
}

template <int TW, int TAPS, int CK>
static int dispatch_tw(const ConvArgs& a, hipStream_t st) {
    const long frags = (long)ceil_div(a.H * TW, 16) * ceil_div(a.W, TW);   // fragments per sample
    // few samples, no statistics record (the CRNN cells' single-plane convolutions, recurrent_varnet.py:241-252): 52-fragment
    // tiles would give 52 workgroups for a 200 x 200 plane; 16-fragment tiles give 169
    if (a.rowsp <= 16 && !a.ypart && TAPS == 9 && !a.vol && (long)a.n * ceil_div(frags, 52L) < 200) {
        if (a.pair_n > 0) return launch_cfg<CK, 1, 1, 4, kSmallMT, TW, 9, true>(a, st);
        return launch_cfg<CK, 1, 1, 4, kSmallMT, TW, TAPS>(a, st);
    }
    CINE_REQUIRE(a.pair_n == 0, CINE_EUNSUPPORTED, "conv: pair launches exist for the single-plane CRNN configuration only");
    if (TAPS == 27 && vol_small_tiles(a.rowsp, a.H, a.W, a.D)) {
        // a volume level with only a handful of 13-fragment tiles (cfg 4: 3 x 50 x 50 -> 48 workgroups, 1 x 25 x 25 -> 8): 4-fragment tiles
        if (a.rowsp <= 16) return launch_cfg<CK, 1, 1, 4, 1, TW, TAPS>(a, st);
        if (a.rowsp <= 32) return launch_cfg<CK, 1, 2, 2, 2, TW, TAPS>(a, st);
        // > 64 rows: two workgroups of 64 rows per tile (cfg 4's 1 x 25 x 25 level has 14 tiles; finer row splits are faster
        // still with one slice in flight -- 41.2 / 44.0 / 44.8 / 45.4 slices/s for 128 / 64 / 32 / 16 rows per workgroup -- but
        // not with twelve: 88.0 / 89.3 / 87.9 / 87.3)
        return launch_cfg<CK, 1, 4, 1, 4, TW, TAPS>(a, st);
    }
    if constexpr (TAPS == 27 && TW == 16) {      // the 27-tap fallback of a layer that vol_mid_tiles gives 14-fragment tiles (sources the three-pass form cannot stage): same geometry
        if (vol_mid_tiles(a.rowsp, a.H, a.W, a.D)) return launch_cfg<CK, 1, 2, 2, 7, TW, TAPS>(a, st);
    }
    if constexpr (TAPS == 9) {
        if (!a.vol && regular_nf(a.rowsp, frags, true) == 14) return launch_cfg<CK, 1, 2, 2, 7, TW, 9>(a, st);
        if (!a.vol && regular_nf(a.rowsp, frags, true) == 4 && a.rowsp <= 64) return launch_cfg<CK, 1, 4, 1, 4, TW, 9>(a, st);
    }
    if constexpr (TAPS == 9 && CK == 8) {
        // the first layer of every U-Net (2 -> chans, unet.py:51): a 4-channel chunk and one k-step per tap instead of an 8-channel
        // chunk that is three quarters zeros (half the MFMAs and half the staging of that layer)
        if (a.rowsp <= 16 && !a.vol && a.cin <= 4 && a.s1.c == 0 && a.nchunks == 1) return launch_cfg<4, 1, 1, kWN16, kMT16, TW, 9>(a, st);
    }
    if constexpr (TAPS == 9 && CK == 8 && TW == 16) {
        // <= 16 rows, no statistics record (the CRNN cells' all-frame convolutions, recurrent_varnet.py:172-176: any tiling gives the same bits): the 52-row
        // tiles of 15 frames of 200 x 200 are 780 workgroups on 768 resident slots (150 registers: three per CU) -- one round plus a sliver of 12, 92 us.
        // 40-row tiles at four workgroups per CU are 975 on 1 024 slots, and 200 = 5 x 40 rows leave no ragged row tile.  A rule on resident rounds x the
        // MFMA work of a round
#ifndef CINE_NO_MT10        // (A/B builds: tools/build_variant.sh nomt10 conv_kernels.hip -DCINE_NO_MT10)
        if (a.rowsp <= 16 && !a.ypart && !a.vol && a.pair_n == 0 && a.W > 16) {      // (W <= 16: the plane-wide kernels of conv_plane.hip, which have no 40-row shape)
            const long w13 = (long)a.n * ceil_div(a.W, 16) * ceil_div(a.H, 52), w10 = (long)a.n * ceil_div(a.W, 16) * ceil_div(a.H, 40);
            if (ceil_div(w10, 1024L) * 40 < ceil_div(w13, 768L) * 39) return launch_cfg<CK, 1, 1, 4, 10, TW, TAPS>(a, st);
        }
#endif
    }
    if (a.rowsp <= 16) return launch_cfg<CK, 1, 1, kWN16, kMT16, TW, TAPS>(a, st);
    // <= 32 rows: two waves split the rows, two split the pixel fragments (52 accumulator registers per wave, three
    // workgroups per CU): a touch slower than one 4-wave workgroup per plane in isolation, but it packs better with the
    // kernels of the other slices in flight (134.8 -> 138.0 slices/s on cfg 2)
    if (a.rowsp <= 32) return launch_cfg<CK, 1, 2, 2, 13, TW, TAPS>(a, st);
    // transpose convs of the narrow levels with ALL their 4 x cout rows in one workgroup (the input tile is staged once instead of
    // once per 64- / 128-row block: 0.925 -> 0.84 ms of transpose conv per cfg-2 slice, 153.3 -> 155.2 slices/s)
    if constexpr (TAPS == 1 && (TW == 4 || TW == 2)) {
        if (a.tconv_cout > 0 && !a.vol && a.rowsp == 128 && frags > 8) return launch_cfg<CK, 2, 4, 1, 13, TW, TAPS>(a, st);
        if (a.tconv_cout > 0 && !a.vol && a.rowsp == 256 && frags <= 8) return launch_cfg<CK, 4, 4, 1, 4, TW, TAPS>(a, st);
    }
    if (a.rowsp <= 64 || frags > 8) return launch_cfg<CK, 1, 4, 1, 13, TW, TAPS>(a, st);
    if constexpr (TAPS == 9) {
        // the U-Nets' coarsest level (128 output rows, ONE workgroup per plane) when the caller's arguments say the slice is alone on the chip
        // (AloneScope): two workgroups of 64 rows per plane -- the pixel tiling, and with it the statistics records, are unchanged.  Measured at
        // cfg 2: one slice 7.43 -> 7.35 ms; with 10 slices in flight the split costs 0.7 % (172.7 vs 173.9 slices/s), hence the condition
        if (!a.vol && a.rowsp == 128 && g_alone_on_chip) return launch_cfg<CK, 1, 4, 1, 4, TW, TAPS>(a, st);
    }
    return launch_cfg<CK, 2, 4, 1, 4, TW, TAPS>(a, st);
}

// 3x3x3 convolutions of volumes wider than 8 voxels with 16-byte rows: the regular 16-wide configurations of the 3x3 kernel in V3
// form (the tile geometry -- and with it the statistics-record count -- is that of the 27-tap configurations it replaces)
static bool conv3d_v3_ok(const ConvArgs& a) {
    auto src_ok = [&](const Src& s) {
        if (s.c == 0) return true;
        if (s.mode == 2) return (s.w == 2 * a.W || s.w == 2 * a.W + 1) && s.h >= 2 * a.H;
        return s.w == a.W && s.h <= a.H;
    };
    if (!(a.vol && a.W > 8 && !a.add_src1 && src_ok(a.s0) && src_ok(a.s1) && (a.s1.c == 0 || a.s0.c % kCK3 == 0))) return false;
    // levels with a handful of tiles keep the 4-fragment tile geometry: V3 has it for > 32 rows
    return !vol_small_tiles(a.rowsp, a.H, a.W, a.D) || a.rowsp > 32;
}
static int dispatch_v3(const ConvArgs& a, hipStream_t st) {
    if (vol_small_tiles(a.rowsp, a.H, a.W, a.D)) return launch_cfg<kCK3, 1, 4, 1, 4, 16, 9, false, 1>(a, st);    // 64 rows per workgroup
    if (vol_mid_tiles(a.rowsp, a.H, a.W, a.D)) return launch_cfg<kCK3, 1, 2, 2, 7, 16, 9, false, 1>(a, st);      // 14-fragment tiles: twice the workgroups of an under-filled launch
    const long frags = (long)ceil_div(a.H * 16, 16) * ceil_div(a.W, 16);
    if (a.rowsp <= 16) return launch_cfg<kCK3, 1, 1, kWN16, kMT16, 16, 9, false, 1>(a, st);
    if (a.rowsp <= 32) return launch_cfg<kCK3, 1, 2, 2, 13, 16, 9, false, 1>(a, st);
    if (a.rowsp <= 64 || frags > 8) return launch_cfg<kCK3, 1, 4, 1, 13, 16, 9, false, 1>(a, st);
    return launch_cfg<kCK3, 2, 4, 1, 4, 16, 9, false, 1>(a, st);
}

// transpose conv k2 s2 / 1x1x1 conv of volumes wider than 8 voxels: the 16-wide 1x1 configurations with volume addressing (V3 = 2)
// and the vectorised staging (the generic path stages volumes element by element)
static bool vol1x1_fast_ok(const ConvArgs& a) {
    return a.vol && a.W > 8 && a.s1.c == 0 && a.s0.mode <= 1 && a.s0.w == a.W && a.s0.h <= a.H && a.s0.d == a.D;
}
static int dispatch_vol1x1(const ConvArgs& a, hipStream_t st) {
    const long frags = (long)ceil_div(a.H * 16, 16) * ceil_div(a.W, 16);
    if (a.rowsp <= 16) return launch_cfg<kCK1, 1, 1, kWN16, kMT16, 16, 1, false, 2>(a, st);
    if (a.rowsp <= 32) return launch_cfg<kCK1, 1, 2, 2, 13, 16, 1, false, 2>(a, st);
    if (a.rowsp <= 64 || frags > 8) return launch_cfg<kCK1, 1, 4, 1, 13, 16, 1, false, 2>(a, st);
    return launch_cfg<kCK1, 2, 4, 1, 4, 16, 1, false, 2>(a, st);
}

template <int TAPS, int CK>
static int dispatch(const ConvArgs& a, hipStream_t st) {
#ifdef CINE_FAST_BUILD      // diagnostic builds (tools/conv_stamps.hip): only the 16-wide 3x3 / transpose-conv instantiations
    if constexpr (TAPS == 27) return CINE_EUNSUPPORTED;
    else return dispatch_tw<16, TAPS, CK>(a, st);
#endif
    if (a.W > 8) return dispatch_tw<16, TAPS, CK>(a, st);
    if (a.W > 4) return dispatch_tw<8, TAPS, CK>(a, st);
    if (a.W > 2) return dispatch_tw<4, TAPS, CK>(a, st);
    return dispatch_tw<2, TAPS, CK>(a, st);
}

// tiles per sample of the configuration dispatch() picks (must mirror dispatch_tw)
int tiles_for(int rowsp, int h, int w, int d = 1, bool vol3 = false, bool plane3x3 = false) {
    const int TW = w > 8 ? 16 : w > 4 ? 8 : w > 2 ? 4 : 2;
    const long frags = (long)ceil_div(h * TW, 16) * ceil_div(w, TW);
    int nf = regular_nf(rowsp, frags, plane3x3);
    if (vol3 && coarse_shape(rowsp, h, w, d)) return coarse_tiles(rowsp, d, h, w, true);
    if (plane3x3 && d == 1 && coarse_shape2d(rowsp, h, w)) return coarse_tiles(rowsp, 1, h, w, false);
    if (vol3 && vol_small_tiles(rowsp, h, w, d)) nf = 4;          // every small-tile volume configuration has 4 fragments
    else if (vol3 && vol_mid_tiles(rowsp, h, w, d)) nf = 14;
    const int TH = nf * 16 / TW;
    return ceil_div(w, TW) * ceil_div(h, TH) * d;
}

static unsigned grid1d(long n, int threads, long cap = 8192) {
    long g = ceil_div(n, (long)threads);
    if (g > cap) g = cap;
    return (unsigned)(g < 1 ? 1 : g);
}

}  // namespace cine

using namespace cine;

// number of partial-statistics records per (sample, channel) that the conv / tconv kernels emit
extern "C" int cine_conv_stat_partials(int cout, int h, int w, int is_tconv) {
    if (cout <= 0 || h <= 0 || w <= 0) return 0;
    const int rows = is_tconv ? 4 * cout : cout;
    const int rowsp = ceil_div(rows, 16) * 16;
    return tiles_for(rowsp, h, w, 1, false, !is_tconv) * (is_tconv ? 4 : 1);
}

extern "C" int cine_conv_stat_partials3d(int cout, int d, int h, int w, int is_tconv) {
    if (cout <= 0 || d <= 0 || h <= 0 || w <= 0) return 0;
    const int rows = is_tconv ? 8 * cout : cout;
    const int rowsp = ceil_div(rows, 16) * 16;
    return tiles_for(rowsp, h, w, d, !is_tconv) * (is_tconv ? 8 : 1);
}

static size_t packed_floats(int rows, int cin, int taps, int ck) {
    return (size_t)ceil_div(cin, ck) * taps * ck * (ceil_div(rows, 16) * 16);
}
extern "C" size_t cine_conv3x3_packed_floats(int cout, int cin) {
    return (cout <= 0 || cin <= 0) ? 0 : packed_floats(cout, cin, 9, kCK3);
}
extern "C" size_t cine_tconv2x2_packed_floats(int cin, int cout) {
    return (cout <= 0 || cin <= 0) ? 0 : packed_floats(4 * cout, cin, 1, kCK1);
}
extern "C" size_t cine_conv1x1_packed_floats(int cout, int cin) {
    return (cout <= 0 || cin <= 0) ? 0 : packed_floats(cout, cin, 1, kCK1);
}

static int pack(const float* w, float* packed, int kind, int cout, int cin, void* stream, const char* what) {
    CINE_REQUIRE(w && packed && cout > 0 && cin > 0, CINE_EINVAL, "%s: bad arguments", what);
    const int rows = kind == 1 ? 4 * cout : (kind == 4 ? 8 * cout : cout);
    const int taps = kind == 0 ? 9 : (kind == 3 ? 27 : 1), ck = kind == 0 ? kCK3 : (kind == 3 ? kCK27 : kCK1);
    const int rowsp = ceil_div(rows, 16) * 16, nchunks = ceil_div(cin, ck);
    const long total = (long)nchunks * taps * ck * rowsp;
    ProfScope prof(F_MISC, as_stream(stream));
    hipLaunchKernelGGL(pack_weights_kernel, dim3(grid1d(total, 256)), dim3(256), 0, as_stream(stream),
                       w, packed, kind, cout, cin, rows, rowsp, taps, ck, nchunks);
    return check_launch("pack_weights_kernel");
}
extern "C" int cine_pack_conv3x3(const float* w, float* packed, int cout, int cin, void* stream) {
    return pack(w, packed, 0, cout, cin, stream, "cine_pack_conv3x3");
}
extern "C" int cine_pack_tconv2x2(const float* w, float* packed, int cin, int cout, void* stream) {
    return pack(w, packed, 1, cout, cin, stream, "cine_pack_tconv2x2");
}
extern "C" int cine_pack_conv1x1(const float* w, float* packed, int cout, int cin, void* stream) {
    return pack(w, packed, 2, cout, cin, stream, "cine_pack_conv1x1");
}
extern "C" size_t cine_conv3d_packed_floats(int cout, int cin) {
    return (cout <= 0 || cin <= 0) ? 0 : packed_floats(cout, cin, 27, kCK3);
}
extern "C" size_t cine_tconv3d_packed_floats(int cin, int cout) {
    return (cout <= 0 || cin <= 0) ? 0 : packed_floats(8 * cout, cin, 1, kCK1);
}
extern "C" int cine_pack_conv3d(const float* w, float* packed, int cout, int cin, void* stream) {
    CINE_REQUIRE(w && packed && cout > 0 && cin > 0, CINE_EINVAL, "cine_pack_conv3d: bad arguments");
    const int rowsp = ceil_div(cout, 16) * 16, ncc = ceil_div(cin, kCK3);
    ProfScope prof(F_MISC, as_stream(stream));
    hipLaunchKernelGGL(pack_conv3d_kernel, dim3(grid1d(27L * ncc * 8 * rowsp, 256)), dim3(256), 0, as_stream(stream), w, packed, cout, cin, rowsp, ncc);
    return check_launch("pack_conv3d_kernel");
}
extern "C" int cine_pack_tconv3d(const float* w, float* packed, int cin, int cout, void* stream) {
    return pack(w, packed, 4, cout, cin, stream, "cine_pack_tconv3d");
}

static int check_src(const float* x, const float* part, int c, int mode, int np, const char* what) {
    CINE_REQUIRE(c >= 0, CINE_EINVAL, "%s: negative channel count", what);
    if (c == 0) return CINE_OK;
    CINE_REQUIRE(x, CINE_EINVAL, "%s: null source", what);
    CINE_REQUIRE(mode >= 0 && mode <= 2, CINE_EINVAL, "%s: mode %d", what, mode);
    CINE_REQUIRE(mode == 0 || (part && np > 0), CINE_EINVAL, "%s: mode %d needs partial stats", what, mode);
    return CINE_OK;
}

// leaky_relu(v) is evaluated as max(v, v * slope), which equals the reference's (v > 0 ? v : v * slope) for 0 <= slope <= 1
static int check_slope(float slope, const char* what) {
    CINE_REQUIRE(slope >= 0.f && slope <= 1.f, CINE_EINVAL, "%s: LeakyReLU slope %g outside [0, 1]", what, (double)slope);
    return CINE_OK;
}

static int conv3x3_full(const float* x0, const float* part0, int np0, int c0, int mode0, int h0, int w0,
                        const float* x1, const float* part1, int np1, int c1, int mode1, int h1, int w1, int add_src1,
                        const float* wpacked, const float* wpacked2, int set_split, const float* bias,
                        const float* addend, int relu, float* accum,
                        float* y, float* part_y, int n, int cout, int h, int w, float eps, float slope, void* stream,
                        const float* bias2 = nullptr);

extern "C" int cine_conv3x3_in(const float* x0, const float* part0, int np0, int c0, int mode0, int h0, int w0,
                               const float* x1, const float* part1, int np1, int c1, int mode1, int h1, int w1,
                               const float* wpacked, const float* wpacked2, int set_split,
                               float* y, float* part_y, int n, int cout, int h, int w, float eps, float slope, void* stream) {
    return conv3x3_full(x0, part0, np0, c0, mode0, h0, w0, x1, part1, np1, c1, mode1, h1, w1, 0, wpacked, wpacked2, set_split,
                        nullptr, nullptr, 0, nullptr, y, part_y, n, cout, h, w, eps, slope, stream);
}

extern "C" int cine_conv3x3_ex(const float* x0, const float* part0, int np0, int c0, int mode0, int h0, int w0,
                               const float* x1, const float* part1, int np1, int c1, int mode1, int h1, int w1, int add_src1,
                               const float* wpacked, const float* bias, const float* addend, int relu,
                               float* y, float* part_y, int n, int cout, int h, int w, float eps, float slope, void* stream) {
    return conv3x3_full(x0, part0, np0, c0, mode0, h0, w0, x1, part1, np1, c1, mode1, h1, w1, add_src1, wpacked, nullptr, 0,
                        bias, addend, relu, nullptr, y, part_y, n, cout, h, w, eps, slope, stream);
}

// cine_conv3x3_ex with two weight / bias sets: samples [0, set_split) use (wpacked, bias), the rest (wpacked2, bias2)
extern "C" int cine_conv3x3_ex2(const float* x0, const float* part0, int np0, int c0, int mode0, int h0, int w0,
                                const float* x1, const float* part1, int np1, int c1, int mode1, int h1, int w1, int add_src1,
                                const float* wpacked, const float* bias, const float* wpacked2, const float* bias2, int set_split,
                                const float* addend, int relu,
                                float* y, float* part_y, int n, int cout, int h, int w, float eps, float slope, void* stream) {
    CINE_REQUIRE(!wpacked2 || (set_split >= 0 && set_split <= n), CINE_EINVAL, "cine_conv3x3_ex2: set_split outside [0, n]");
    CINE_REQUIRE(!bias == !bias2 || !wpacked2, CINE_EINVAL, "cine_conv3x3_ex2: both sets need a bias, or neither");
    return conv3x3_full(x0, part0, np0, c0, mode0, h0, w0, x1, part1, np1, c1, mode1, h1, w1, add_src1, wpacked, wpacked2, set_split,
                        bias, addend, relu, nullptr, y, part_y, n, cout, h, w, eps, slope, stream, wpacked2 ? bias2 : nullptr);
}

// one step of a convolutional-RNN time sweep (reference recurrent_varnet.py:241-254): y = ReLU(conv3x3(x; w) + addend) and,
// when accum != NULL, accum += y in the same epilogue (the backward sweep adding onto the forward sweep's outputs)
extern "C" int cine_crnn_step(const float* x, const float* wpacked, const float* addend, float* y, float* accum,
                              int n, int c, int h, int w, int relu, void* stream) {
    CINE_REQUIRE(x && wpacked && addend && y, CINE_EINVAL, "cine_crnn_step: null pointer");
    CINE_REQUIRE(y != x && accum != x && accum != y, CINE_EINVAL, "cine_crnn_step: outputs must not alias the input or each other");
    return conv3x3_full(x, nullptr, 0, c, 0, h, w, nullptr, nullptr, 0, 0, 0, 0, 0, 0, wpacked, nullptr, 0,
                        nullptr, addend, relu ? 1 : 0, accum, y, nullptr, n, c, h, w, 1e-5f, 0.2f, stream);
}

// both directions of a BCRNN time sweep (recurrent_varnet.py:241-252: the forward and the backward pass over time are independent
// chains; only their sum couples them, :254) in ONE launch: set f and set b are two cine_crnn_step calls on different tensors.
static int crnn_step2_impl(const float* x_f, const float* addend_f, float* y_f, float* accum_f, int store_f, const float* gate_f,
                           const float* x_b, const float* addend_b, float* y_b, float* accum_b, int store_b, const float* gate_b,
                           const float* wpacked, int n, int c, int h, int w, int relu, void* stream) {
    CINE_REQUIRE(x_f && addend_f && y_f && wpacked, CINE_EINVAL, "cine_crnn_step2: null pointer");
    CINE_REQUIRE(n > 0 && 2 * n <= 65535 && c > 0 && h > 0 && w > 0, CINE_EINVAL, "cine_crnn_step2: bad sizes");
    CINE_REQUIRE(y_f != x_f && accum_f != x_f && accum_f != y_f, CINE_EINVAL, "cine_crnn_step2: outputs must not alias the input or each other");
    ConvArgs a{};
    a.s0 = Src{x_f, nullptr, c, 0, h, w, 0, 0, 1};
    a.s1 = Src{nullptr, nullptr, 0, 0, 0, 0, 0, 0, 1};
    a.addend = addend_f; a.relu = relu ? 1 : 0; a.accum = accum_f; a.accum_store = store_f && accum_f; a.gate = gate_f;
    a.wp0 = a.wp1 = wpacked; a.set_split = 2 * n;
    a.y = y_f; a.ypart = nullptr; a.n = n; a.cin = c; a.rows = c; a.rowsp = ceil_div(c, 16) * 16;
    a.H = h; a.W = w; a.D = 1; a.slope = 0.2f; a.eps = 1e-5f; a.nchunks = ceil_div(c, kCK3);
    if (!x_b) return dispatch<9, kCK3>(a, as_stream(stream));
    CINE_REQUIRE(addend_b && y_b, CINE_EINVAL, "cine_crnn_step2: null pointer in set b");
    CINE_REQUIRE(y_b != x_b && accum_b != x_b && accum_b != y_b && y_b != y_f && y_b != x_f && y_f != x_b, CINE_EINVAL,
                 "cine_crnn_step2: the two sets must not write what the other reads");
    CINE_REQUIRE(!accum_f || accum_f != accum_b, CINE_EINVAL, "cine_crnn_step2: both sets accumulate into the same tensor (run them one after the other)");
    const int TW = w > 8 ? 16 : w > 4 ? 8 : w > 2 ? 4 : 2;
    const long frags = (long)ceil_div(h * TW, 16) * ceil_div(w, TW);
    if (a.rowsp <= 16 && 2L * n * ceil_div(frags, 52L) < 200 && !gate_f == !gate_b) {
        a.n = 2 * n; a.pair_n = n;
        a.x_b = x_b; a.addend_b = addend_b; a.y_b = y_b; a.accum_b = accum_b; a.accum_store_b = store_b && accum_b; a.gate_b = gate_b;
        return dispatch<9, kCK3>(a, as_stream(stream));
    }
    if (int e = dispatch<9, kCK3>(a, as_stream(stream))) return e;      // shapes outside the pair configuration: two launches
    a.s0.x = x_b; a.addend = addend_b; a.y = y_b; a.accum = accum_b; a.accum_store = store_b && accum_b; a.gate = gate_b;
    return dispatch<9, kCK3>(a, as_stream(stream));
}

extern "C" int cine_crnn_step2(const float* x_f, const float* addend_f, float* y_f, float* accum_f, int store_f,
                               const float* x_b, const float* addend_b, float* y_b, float* accum_b, int store_b,
                               const float* wpacked, int n, int c, int h, int w, int relu, void* stream) {
    return crnn_step2_impl(x_f, addend_f, y_f, accum_f, store_f, nullptr, x_b, addend_b, y_b, accum_b, store_b, nullptr, wpacked, n, c, h, w, relu, stream);
}

// Both time sweeps of a BCRNN layer (recurrent_varnet.py:236-254, batch 1) in ONE call: h_t = [ReLU](conv3x3(h_prev; W_h2h) + P_t) forward in time
// (frames 0 .. T-1) and backward in time (T-1 .. 0), both chains advanced by one pair launch per step (T launches, T + 1 for odd T: the
// middle frame is reached by both in the same step), out = hidden_f + hidden_b -- the first chain to reach a frame stores, the second adds.
// P, out (T, c, h, w); `zero` (c, h, w) zeros = hid_init (:236), read only; keep != 0: hf / hb (T, c, h, w) receive EVERY hidden state (what
// cine_bcrnn_sweep_bwd reads), keep == 0: hf / hb are two-frame ping-pong buffers (2, c, h, w).
extern "C" int cine_bcrnn_sweep(const float* P, const float* wpacked_hh, const float* zero, float* hf, float* hb, float* out,
                                int T, int c, int h, int w, int relu, int keep, void* stream) {
    CINE_REQUIRE(P && wpacked_hh && zero && hf && hb && out, CINE_EINVAL, "cine_bcrnn_sweep: null pointer");
    CINE_REQUIRE(T > 0 && c > 0 && h > 0 && w > 0, CINE_EINVAL, "cine_bcrnn_sweep: bad sizes");
    const long fr = (long)c * h * w;
    auto slot = [&](int i) { return (long)(keep ? i : (i & 1)) * fr; };
    const float *prev_f = zero, *prev_b = zero;
    diag_count(D_CRNN_SWEEP_C);
    for (int s = 0; s < T; ++s) {
        const int i_f = s, i_b = T - 1 - s;
        // ping-pong: the two chains write slots s & 1 of their own buffers; keep: slot = the frame they reach
        float* yf = hf + (keep ? (long)i_f * fr : slot(s));
        float* yb = hb + (keep ? (long)i_b * fr : slot(s));
        int e;
        if (i_f == i_b) {
            if ((e = crnn_step2_impl(prev_f, P + i_f * fr, yf, out + i_f * fr, 1, nullptr, nullptr, nullptr, nullptr, nullptr, 0, nullptr, wpacked_hh, 1, c, h, w, relu, stream))) return e;
            e = crnn_step2_impl(prev_b, P + i_b * fr, yb, out + i_b * fr, 0, nullptr, nullptr, nullptr, nullptr, nullptr, 0, nullptr, wpacked_hh, 1, c, h, w, relu, stream);
        } else {
            const int first = i_f < i_b;          // before the chains cross each of them is the first to reach its frame
            e = crnn_step2_impl(prev_f, P + i_f * fr, yf, out + i_f * fr, first, nullptr, prev_b, P + i_b * fr, yb, out + i_b * fr, first, nullptr,
                                wpacked_hh, 1, c, h, w, relu, stream);
        }
        if (e) return e;
        prev_f = yf; prev_b = yb;
    }
    return CINE_OK;
}

// Back-propagation through time of cine_bcrnn_sweep (keep != 0): from gout = d loss / d out (T, c, h, w),
//   gf[t] = [hf[t] > 0] (conv(gf[t + 1]; Wd) + gout[t]),  t = T-1 .. 0      (the chain that ran forward in time)
//   gb[t] = [hb[t] > 0] (conv(gb[t - 1]; Wd) + gout[t]),  t = 0 .. T-1      (the chain that ran backward in time)
// with Wd = the input-gradient packing of W_h2h (cine_pack_conv3x3_dgrad) -- the forward conv kernel, gout riding in as the addend and the ReLU
// mask as the epilogue's gate (relu == 0: identity activation, no gate) -- and gP = gf + gb = d loss / d P through the second output.  The two
// chains are independent: one pair launch per step, as in the forward sweep.  gf, gb, gP (T, c, h, w); `zero` as above.
extern "C" int cine_bcrnn_sweep_bwd(const float* gout, const float* wpacked_hh_dgrad, const float* zero, const float* hf, const float* hb,
                                    float* gf, float* gb, float* gP, int T, int c, int h, int w, int relu, void* stream) {
    CINE_REQUIRE(gout && wpacked_hh_dgrad && zero && hf && hb && gf && gb && gP, CINE_EINVAL, "cine_bcrnn_sweep_bwd: null pointer");
    CINE_REQUIRE(T > 0 && c > 0 && h > 0 && w > 0, CINE_EINVAL, "cine_bcrnn_sweep_bwd: bad sizes");
    const long fr = (long)c * h * w;
    const float *prev_f = zero, *prev_b = zero;
    for (int s = 0; s < T; ++s) {
        const int t_f = T - 1 - s, t_b = s;
        const float* gate_f = relu ? hf + t_f * fr : nullptr;
        const float* gate_b = relu ? hb + t_b * fr : nullptr;
        int e;
        if (t_f == t_b) {
            if ((e = crnn_step2_impl(prev_f, gout + t_f * fr, gf + t_f * fr, gP + t_f * fr, 1, gate_f, nullptr, nullptr, nullptr, nullptr, 0, nullptr,
                                     wpacked_hh_dgrad, 1, c, h, w, 0, stream))) return e;
            e = crnn_step2_impl(prev_b, gout + t_b * fr, gb + t_b * fr, gP + t_b * fr, 0, gate_b, nullptr, nullptr, nullptr, nullptr, 0, nullptr,
                                wpacked_hh_dgrad, 1, c, h, w, 0, stream);
        } else {
            const int first = t_f > t_b;
            e = crnn_step2_impl(prev_f, gout + t_f * fr, gf + t_f * fr, gP + t_f * fr, first, gate_f, prev_b, gout + t_b * fr, gb + t_b * fr, gP + t_b * fr, first, gate_b,
                                wpacked_hh_dgrad, 1, c, h, w, 0, stream);
        }
        if (e) return e;
        prev_f = gf + t_f * fr; prev_b = gb + t_b * fr;
    }
    return CINE_OK;
}

// mode encoding of the extended entry: low 3 bits = mode (0..4), bit 3 set = source is raw and gets
// InstanceNorm + LeakyReLU before the wavelet transform (modes 3/4)
static int conv3x3_full(const float* x0, const float* part0, int np0, int c0, int mode0, int h0, int w0,
                        const float* x1, const float* part1, int np1, int c1, int mode1, int h1, int w1, int add_src1,
                        const float* wpacked, const float* wpacked2, int set_split, const float* bias,
                        const float* addend, int relu, float* accum,
                        float* y, float* part_y, int n, int cout, int h, int w, float eps, float slope, void* stream,
                        const float* bias2) {
    CINE_REQUIRE(wpacked && y, CINE_EINVAL, "cine_conv3x3_in: null pointer");
    if (int e = check_slope(slope, "cine_conv3x3_in")) return e;
    CINE_REQUIRE(n > 0 && n <= 65535 && cout > 0 && h > 0 && w > 0 && c0 > 0, CINE_EINVAL, "cine_conv3x3_in: bad sizes");
    const int act0 = (mode0 >> 3) & 1, act1 = (mode1 >> 3) & 1;
    mode0 &= 7; mode1 &= 7;
    CINE_REQUIRE(mode0 <= 4 && mode1 <= 4, CINE_EINVAL, "cine_conv3x3_in: bad mode");
    CINE_REQUIRE(mode0 != 4 || c0 % 4 == 0, CINE_EINVAL, "cine_conv3x3_in: IWT source needs 4k channels");
    if (int e = check_src(x0, part0, c0, (mode0 >= 3 ? act0 : mode0), np0, "cine_conv3x3_in(src0)")) return e;
    if (int e = check_src(x1, part1, c1, (mode1 >= 3 ? act1 : mode1), np1, "cine_conv3x3_in(src1)")) return e;
    ConvArgs a{};
    a.s0 = Src{x0, part0, c0, mode0, h0, w0, np0, act0, 1};
    a.s1 = Src{x1, part1, c1, c1 > 0 ? mode1 : 0, h1, w1, np1, act1, 1};
    a.add_src1 = add_src1 && c1 > 0;
    a.bias = bias; a.bias1 = bias2 ? bias2 : bias; a.addend = addend; a.relu = relu; a.accum = accum;
    if (a.add_src1) CINE_REQUIRE(src_cin(a.s0) == src_cin(a.s1), CINE_EINVAL, "cine_conv3x3_in: added sources differ in channels");
    a.wp0 = wpacked; a.wp1 = wpacked2 ? wpacked2 : wpacked; a.set_split = wpacked2 ? set_split : n;
    a.y = y; a.ypart = part_y; a.n = n;
    a.cin = a.add_src1 ? src_cin(a.s0) : src_cin(a.s0) + src_cin(a.s1);
    a.rows = cout; a.rowsp = ceil_div(cout, 16) * 16;
    a.H = h; a.W = w; a.D = 1; a.slope = slope; a.eps = eps; a.nchunks = ceil_div(a.cin, kCK3);
    if (coarse_shape2d(a.rowsp, h, w)) {
        bool handled = false;
        const int e = launch_conv_coarse(a, as_stream(stream), &handled);
        if (e || handled) return e;
    }
    return dispatch<9, kCK3>(a, as_stream(stream));
}

extern "C" int cine_tconv2x2_in(const float* x, const float* part_x, int np_x, int mode,
                                const float* wpacked, const float* wpacked2, int set_split,
                                float* y, float* part_y, int n, int cin, int cout, int h, int w,
                                float eps, float slope, void* stream) {
    if (int e = check_slope(slope, "conv entry point")) return e;
    CINE_REQUIRE(x && wpacked && y, CINE_EINVAL, "cine_tconv2x2_in: null pointer");
    CINE_REQUIRE(n > 0 && n <= 65535 && cin > 0 && cout > 0 && h > 0 && w > 0, CINE_EINVAL, "cine_tconv2x2_in: bad sizes");
    if (int e = check_src(x, part_x, cin, mode, np_x, "cine_tconv2x2_in")) return e;
    CINE_REQUIRE(mode != 2, CINE_EINVAL, "cine_tconv2x2_in: mode 2 not supported");
    ConvArgs a{};
    a.s0 = Src{x, part_x, cin, mode, h, w, np_x, 0, 1};
    a.wp0 = wpacked; a.wp1 = wpacked2 ? wpacked2 : wpacked; a.set_split = wpacked2 ? set_split : n;
    a.y = y; a.ypart = part_y; a.n = n; a.cin = cin; a.rows = 4 * cout; a.rowsp = ceil_div(4 * cout, 16) * 16;
    a.tconv_cout = cout; a.H = h; a.W = w; a.D = 1; a.slope = slope; a.eps = eps; a.nchunks = ceil_div(cin, kCK1);
    return dispatch<1, kCK1>(a, as_stream(stream));
}

extern "C" int cine_conv1x1_bias(const float* x, const float* part_x, int np_x, int mode,
                                 const float* wpacked, const float* bias, const float* wpacked2, const float* bias2,
                                 int set_split, float* y, int n, int cin, int cout, int h, int w,
                                 float eps, float slope, void* stream) {
    if (int e = check_slope(slope, "conv entry point")) return e;
    CINE_REQUIRE(x && wpacked && bias && y, CINE_EINVAL, "cine_conv1x1_bias: null pointer");
    CINE_REQUIRE(n > 0 && n <= 65535 && cin > 0 && cout > 0 && h > 0 && w > 0, CINE_EINVAL, "cine_conv1x1_bias: bad sizes");
    if (int e = check_src(x, part_x, cin, mode, np_x, "cine_conv1x1_bias")) return e;
    CINE_REQUIRE(mode != 2, CINE_EINVAL, "cine_conv1x1_bias: mode 2 not supported");
    const bool two = wpacked2 && bias2 && set_split < n;
    const long hw = (long)h * w;
    if (cout <= 4 && cin <= 1024 && (mode == 0 || mode == 1) && hw % 4 == 0 &&
        reinterpret_cast<uintptr_t>(x) % 16 == 0 && reinterpret_cast<uintptr_t>(y) % 16 == 0) {
        Conv1Args c{x, part_x, np_x, mode, wpacked, two ? wpacked2 : wpacked, bias, two ? bias2 : bias, two ? set_split : n,
                    y, cin, ceil_div(cout, 16) * 16, hw, eps, slope};
        const dim3 grid((unsigned)ceil_div(hw / 4, 256L), n);
        const size_t lds = (size_t)(2 + cout) * cin * sizeof(float);
        hipStream_t st = as_stream(stream);
        ProfScope prof(F_CONV1, st);
        switch (cout) {
            case 1: hipLaunchKernelGGL(conv1x1_stream_kernel<1>, grid, dim3(256), lds, st, c); break;
            case 2: hipLaunchKernelGGL(conv1x1_stream_kernel<2>, grid, dim3(256), lds, st, c); break;
            case 3: hipLaunchKernelGGL(conv1x1_stream_kernel<3>, grid, dim3(256), lds, st, c); break;
            default: hipLaunchKernelGGL(conv1x1_stream_kernel<4>, grid, dim3(256), lds, st, c); break;
        }
        return check_launch("conv1x1_stream_kernel");
    }
    for (int s = 0; s < (two ? 2 : 1); ++s) {     // the bias pointer is per launch: one launch per weight set
        const int n0 = s ? set_split : 0, n1 = two ? (s ? n : set_split) : n;
        if (n1 <= n0) continue;
        ConvArgs a{};
        a.s0 = Src{x + (size_t)n0 * cin * h * w, part_x ? part_x + (size_t)n0 * cin * np_x * 3 : nullptr, cin, mode, h, w, np_x, 0, 1};
        a.wp0 = a.wp1 = s ? wpacked2 : wpacked; a.set_split = n1 - n0; a.bias = a.bias1 = s ? bias2 : bias;
        a.y = y + (size_t)n0 * cout * h * w; a.ypart = nullptr; a.n = n1 - n0; a.cin = cin; a.rows = cout;
        a.rowsp = ceil_div(cout, 16) * 16; a.H = h; a.W = w; a.D = 1; a.slope = slope; a.eps = eps; a.nchunks = ceil_div(cin, kCK1);
        if (int e = dispatch<1, kCK1>(a, as_stream(stream))) return e;
    }
    return CINE_OK;
}

extern "C" int cine_instnorm_partials(const float* x, float* part, long planes, long plane_elems, void* stream) {
    CINE_REQUIRE(x && part && planes > 0 && plane_elems > 0, CINE_EINVAL, "cine_instnorm_partials: bad arguments");
    hipStream_t st = as_stream(stream);
    ProfScope prof(F_STATS, st);
    if (plane_elems <= 8192)
        hipLaunchKernelGGL(instnorm_partial_wave_kernel, dim3((unsigned)ceil_div(planes, 4L)), dim3(256), 0, st,
                           x, part, planes, (int)plane_elems);
    else
        hipLaunchKernelGGL(instnorm_partial_block_kernel, dim3((unsigned)planes), dim3(256), 0, st, x, part, plane_elems);
    return check_launch("instnorm_partials");
}

extern "C" int cine_instnorm_finalize(const float* part, float* stats, long planes, int np, float eps, void* stream) {
    CINE_REQUIRE(part && stats && planes > 0 && np > 0, CINE_EINVAL, "cine_instnorm_finalize: bad arguments");
    ProfScope prof(F_STATS, as_stream(stream));
    hipLaunchKernelGGL(instnorm_finalize_kernel, dim3((unsigned)ceil_div(planes, 256L)), dim3(256), 0, as_stream(stream),
                       part, stats, planes, np, eps);
    return check_launch("instnorm_finalize_kernel");
}

extern "C" int cine_instnorm_lrelu_apply(const float* x, const float* part, int np, float* y, long planes, long plane_elems,
                                         float eps, float slope, void* stream) {
    if (int e = check_slope(slope, "conv entry point")) return e;
    CINE_REQUIRE(x && part && y && planes > 0 && plane_elems > 0 && np > 0, CINE_EINVAL, "cine_instnorm_lrelu_apply: bad arguments");
    ProfScope prof(F_MISC, as_stream(stream));
    hipLaunchKernelGGL(instnorm_lrelu_apply_kernel, dim3(grid1d(planes * plane_elems, 256)), dim3(256), 0,
                       as_stream(stream), x, part, np, y, planes, plane_elems, eps, slope);
    return check_launch("instnorm_lrelu_apply_kernel");
}

extern "C" int cine_pool3d_act(const float* x, const float* part, int np, float* y, long planes, int d, int h, int w,
                               float eps, float slope, void* stream) {
    if (int e = check_slope(slope, "cine_pool3d_act")) return e;
    CINE_REQUIRE(x && part && y && planes > 0 && np > 0 && d >= 2 && h >= 2 && w >= 2, CINE_EINVAL, "cine_pool3d_act: bad arguments");
    const bool vec = w % 8 == 0 && reinterpret_cast<uintptr_t>(x) % 16 == 0 && reinterpret_cast<uintptr_t>(y) % 16 == 0;
    const long total = planes * (long)(d / 2) * (h / 2) * (vec ? w / 8 : w / 2);
    ProfScope prof(F_PACK, as_stream(stream));
    if (vec) hipLaunchKernelGGL(pool3d_act_kernel<true>, dim3(grid1d(total, 256, 16384)), dim3(256), 0, as_stream(stream), x, part, np, y, planes, d, h, w, eps, slope);
    else hipLaunchKernelGGL(pool3d_act_kernel<false>, dim3(grid1d(total, 256, 16384)), dim3(256), 0, as_stream(stream), x, part, np, y, planes, d, h, w, eps, slope);
    return check_launch("pool3d_act_kernel");
}

// 1 when cine_conv3d_in pools a mode-2 source inside an efficient kernel for this layer shape (conv_coarse.hip); 0 when the layer runs on
// the 16-wide tile kernels, whose pooled staging is element-wise: a caller then does better to materialise the pooled tensor once
// (cine_pool3d_act) and hand it over as a plain source -- what cine_unet3d_forward does for its level 1
extern "C" int cine_conv3d_pools_on_load(int cout, int d, int h, int w) {
    if (cout <= 0 || d <= 0 || h <= 0 || w <= 0) return 0;
    return coarse_shape(ceil_div(cout, 16) * 16, h, w, d) ? 1 : 0;
}

// ---------------------------------------------------------------- input gradients (training)
// The gradient of a convolution with respect to its input is a convolution of the output gradient: conv3x3 (pad 1) with the
// weights transposed (cout <-> cin) and the taps flipped; the k2 s2 transpose conv with its weight matrix (cin, 4 cout) over
// the space-to-depth view of the output gradient (source mode 5); the 1x1 conv with the transposed matrix.  All three run on
// conv_mfma_kernel; only the weight packing differs.
static int pack_general(const float* w, float* packed, int kind, int rows, int kdim, int taps, int ck, void* stream, const char* what) {
    CINE_REQUIRE(w && packed && rows > 0 && kdim > 0, CINE_EINVAL, "%s: bad arguments", what);
    const int rowsp = ceil_div(rows, 16) * 16, nchunks = ceil_div(kdim, ck);
    const long total = (long)nchunks * taps * ck * rowsp;
    ProfScope prof(F_MISC, as_stream(stream));
    hipLaunchKernelGGL(pack_weights_kernel, dim3(grid1d(total, 256)), dim3(256), 0, as_stream(stream),
                       w, packed, kind, 0, kdim, rows, rowsp, taps, ck, nchunks);
    return check_launch("pack_weights_kernel");
}
extern "C" size_t cine_conv3x3_dgrad_packed_floats(int cout, int cin) { return (cout <= 0 || cin <= 0) ? 0 : packed_floats(cin, cout, 9, kCK3); }
extern "C" size_t cine_tconv2x2_dgrad_packed_floats(int cin, int cout) { return (cout <= 0 || cin <= 0) ? 0 : packed_floats(cin, 4 * cout, 1, kCK1); }
extern "C" size_t cine_conv1x1_dgrad_packed_floats(int cout, int cin) { return (cout <= 0 || cin <= 0) ? 0 : packed_floats(cin, cout, 1, kCK1); }
extern "C" int cine_pack_conv3x3_dgrad(const float* w, float* packed, int cout, int cin, void* stream) {
    return pack_general(w, packed, 5, cin, cout, 9, kCK3, stream, "cine_pack_conv3x3_dgrad");
}
extern "C" int cine_pack_tconv2x2_dgrad(const float* w, float* packed, int cin, int cout, void* stream) {
    return pack_general(w, packed, 2, cin, 4 * cout, 1, kCK1, stream, "cine_pack_tconv2x2_dgrad");   // (cin, cout, 2, 2) read as a (cin, 4 cout) matrix
}
extern "C" int cine_pack_conv1x1_dgrad(const float* w, float* packed, int cout, int cin, void* stream) {
    return pack_general(w, packed, 7, cin, cout, 1, kCK1, stream, "cine_pack_conv1x1_dgrad");
}

// Batched packing.  cine_pack_desc writes, into HOST memory, the descriptor of one tensor: op 0 = cine_pack_conv3x3 (n1 = cout, n2 = cin), 1 = cine_pack_tconv2x2
// (cin, cout), 2 = cine_pack_conv1x1 (cout, cin), 3 = cine_pack_conv3x3_dgrad (cout, cin), 4 = cine_pack_tconv2x2_dgrad (cin, cout), 5 = cine_pack_conv1x1_dgrad
// (cout, cin) -- the same arguments, the same packed layout and size.  The caller copies an array of them to the device once (the parameters and their
// packed buffers keep their addresses across optimiser steps) and cine_pack_batch re-packs all of them in one launch.
extern "C" size_t cine_pack_desc_bytes(void) { return sizeof(PackDesc); }
extern "C" int cine_pack_desc(void* desc_host, int op, const float* w, float* packed, int n1, int n2) {
    CINE_REQUIRE(desc_host && w && packed && n1 > 0 && n2 > 0 && op >= 0 && op <= 5, CINE_EINVAL, "cine_pack_desc: bad arguments");
    PackDesc d{};
    d.w = w; d.p = packed;
    if (op <= 2) {                       // pack(): kind = op, (cout, cin) = (n1, n2) except the transpose conv's (cin, cout)
        const int cout = op == 1 ? n2 : n1, cin = op == 1 ? n1 : n2;
        d.kind = op; d.cout = cout; d.cin = cin;
        d.rows = op == 1 ? 4 * cout : cout; d.taps = op == 0 ? 9 : 1; d.ck = op == 0 ? kCK3 : kCK1;
        d.rowsp = ceil_div(d.rows, 16) * 16; d.nchunks = ceil_div(cin, d.ck);
    } else {                             // pack_general(kind, rows, kdim): the kernel's `cin` is the K dimension
        if (op == 3) { d.kind = 5; d.rows = n2; d.cin = n1; d.taps = 9; d.ck = kCK3; }              // (cout, cin): rows = cin, K = cout
        else if (op == 4) { d.kind = 2; d.rows = n1; d.cin = 4 * n2; d.taps = 1; d.ck = kCK1; }     // (cin, cout): rows = cin, K = 4 cout
        else { d.kind = 7; d.rows = n2; d.cin = n1; d.taps = 1; d.ck = kCK1; }                      // (cout, cin): rows = cin, K = cout
        d.cout = 0;
        d.rowsp = ceil_div(d.rows, 16) * 16; d.nchunks = ceil_div(d.cin, d.ck);
    }
    *reinterpret_cast<PackDesc*>(desc_host) = d;
    return CINE_OK;
}
extern "C" int cine_pack_batch(const void* descs_dev, int n, long max_total, void* stream) {
    CINE_REQUIRE(descs_dev && n > 0 && n <= 65535 && max_total > 0, CINE_EINVAL, "cine_pack_batch: bad arguments");
    ProfScope prof(F_MISC, as_stream(stream));
    const unsigned gx = (unsigned)std::max(1L, std::min(64L, ceil_div(max_total, 256L)));
    hipLaunchKernelGGL(pack_weights_batch_kernel, dim3(gx, (unsigned)n), dim3(256), 0, as_stream(stream), reinterpret_cast<const PackDesc*>(descs_dev));
    return check_launch("pack_weights_batch_kernel");
}

extern "C" int cine_conv3x3_dgrad(const float* gy, const float* wpacked, const float* wpacked2, int set_split,
                                  float* gx, int n, int cout, int cin, int h, int w, void* stream) {
    CINE_REQUIRE(gy && wpacked && gx, CINE_EINVAL, "cine_conv3x3_dgrad: null pointer");
    return conv3x3_full(gy, nullptr, 0, cout, 0, h, w, nullptr, nullptr, 0, 0, 0, 0, 0, 0, wpacked, wpacked2, set_split,
                        nullptr, nullptr, 0, nullptr, gx, nullptr, n, cin, h, w, 1e-5f, 0.2f, stream);
}

// gx = gate > 0 ? (conv3x3(gy; wpacked) + addend) : 0 -- the input gradient of a conv whose input was a ReLU output `gate` that also fed other
// consumers (their gradient = addend): mask and sum ride in the conv's epilogue instead of two more passes over gx
extern "C" int cine_conv3x3_dgrad_gated(const float* gy, const float* wpacked, const float* addend, const float* gate,
                                        float* gx, int n, int cout, int cin, int h, int w, void* stream) {
    CINE_REQUIRE(gy && wpacked && gx, CINE_EINVAL, "cine_conv3x3_dgrad_gated: null pointer");
    CINE_REQUIRE(n > 0 && n <= 65535 && cout > 0 && cin > 0 && h > 0 && w > 0, CINE_EINVAL, "cine_conv3x3_dgrad_gated: bad sizes");
    CINE_REQUIRE(gx != gy && gx != addend && gx != gate, CINE_EINVAL, "cine_conv3x3_dgrad_gated: the output must not alias an input");
    ConvArgs a{};
    a.s0 = Src{gy, nullptr, cout, 0, h, w, 0, 0, 1};
    a.s1 = Src{nullptr, nullptr, 0, 0, 0, 0, 0, 0, 1};
    a.addend = addend; a.gate = gate;
    a.wp0 = a.wp1 = wpacked; a.set_split = n;
    a.y = gx; a.ypart = nullptr; a.n = n; a.cin = cout; a.rows = cin; a.rowsp = ceil_div(cin, 16) * 16;
    a.H = h; a.W = w; a.D = 1; a.slope = 0.2f; a.eps = 1e-5f; a.nchunks = ceil_div(a.cin, kCK3);
    return dispatch<9, kCK3>(a, as_stream(stream));
}

extern "C" int cine_tconv2x2_dgrad(const float* gy, const float* wpacked, const float* wpacked2, int set_split,
                                   float* gx, int n, int cin, int cout, int h, int w, void* stream) {
    CINE_REQUIRE(gy && wpacked && gx, CINE_EINVAL, "cine_tconv2x2_dgrad: null pointer");
    CINE_REQUIRE(n > 0 && n <= 65535 && cin > 0 && cout > 0 && h > 0 && w > 0, CINE_EINVAL, "cine_tconv2x2_dgrad: bad sizes");
    ConvArgs a{};
    a.s0 = Src{gy, nullptr, cout, 5, 2 * h, 2 * w, 0, 0, 1};
    a.wp0 = wpacked; a.wp1 = wpacked2 ? wpacked2 : wpacked; a.set_split = wpacked2 ? set_split : n;
    a.y = gx; a.ypart = nullptr; a.n = n; a.cin = 4 * cout; a.rows = cin; a.rowsp = ceil_div(cin, 16) * 16;
    a.H = h; a.W = w; a.D = 1; a.slope = 0.2f; a.eps = 1e-5f; a.nchunks = ceil_div(a.cin, kCK1);
    return dispatch<1, kCK1>(a, as_stream(stream));
}

extern "C" int cine_conv1x1_dgrad(const float* gy, const float* wpacked, const float* wpacked2, int set_split,
                                  float* gx, int n, int cout, int cin, int h, int w, void* stream) {
    CINE_REQUIRE(gy && wpacked && gx, CINE_EINVAL, "cine_conv1x1_dgrad: null pointer");
    CINE_REQUIRE(n > 0 && n <= 65535 && cin > 0 && cout > 0 && h > 0 && w > 0, CINE_EINVAL, "cine_conv1x1_dgrad: bad sizes");
    ConvArgs a{};
    a.s0 = Src{gy, nullptr, cout, 0, h, w, 0, 0, 1};
    a.wp0 = wpacked; a.wp1 = wpacked2 ? wpacked2 : wpacked; a.set_split = wpacked2 ? set_split : n;
    a.y = gx; a.ypart = nullptr; a.n = n; a.cin = cout; a.rows = cin; a.rowsp = ceil_div(cin, 16) * 16;
    a.H = h; a.W = w; a.D = 1; a.slope = 0.2f; a.eps = 1e-5f; a.nchunks = ceil_div(cout, kCK1);
    return dispatch<1, kCK1>(a, as_stream(stream));
}

// ---------------------------------------------------------------- 3-D entry points (unet.py dims = 3)
extern "C" int cine_conv3d_in(const float* x0, const float* part0, int np0, int c0, int mode0, int d0, int h0, int w0,
                              const float* x1, const float* part1, int np1, int c1, int mode1, int d1, int h1, int w1,
                              const float* wpacked, const float* bias, const float* addend, int relu,
                              float* y, float* part_y, int n, int cout, int d, int h, int w, float eps, float slope, void* stream) {
    if (int e = check_slope(slope, "conv entry point")) return e;
    CINE_REQUIRE(wpacked && y, CINE_EINVAL, "cine_conv3d_in: null pointer");
    CINE_REQUIRE(n > 0 && n <= 65535 && cout > 0 && d > 0 && h > 0 && w > 0 && c0 > 0, CINE_EINVAL, "cine_conv3d_in: bad sizes");
    CINE_REQUIRE(mode0 >= 0 && mode0 <= 2 && mode1 >= 0 && mode1 <= 2, CINE_EINVAL, "cine_conv3d_in: modes 0..2 only");
    if (int e = check_src(x0, part0, c0, mode0, np0, "cine_conv3d_in(src0)")) return e;
    if (int e = check_src(x1, part1, c1, mode1, np1, "cine_conv3d_in(src1)")) return e;
    ConvArgs a{};
    a.s0 = Src{x0, part0, c0, mode0, h0, w0, np0, 2, d0};          // act bit 1 marks a volume source
    a.s1 = Src{x1, part1, c1, c1 > 0 ? mode1 : 0, h1, w1, np1, 2, d1 > 0 ? d1 : 1};
    a.vol = 1;
    a.wp0 = a.wp1 = wpacked; a.set_split = n; a.bias = a.bias1 = bias; a.addend = addend; a.relu = relu;
    a.y = y; a.ypart = part_y; a.n = n; a.cin = c0 + c1; a.rows = cout; a.rowsp = ceil_div(cout, 16) * 16;
    a.H = h; a.W = w; a.D = d; a.slope = slope; a.eps = eps; a.ncc = ceil_div(a.cin, kCK3);
    if (coarse_shape(a.rowsp, h, w, d)) {
        bool handled = false;
        const int e = launch_conv_coarse(a, as_stream(stream), &handled);
        if (e || handled) return e;
    }
    if (conv3d_v3_ok(a)) { a.nchunks = 3 * a.ncc; return dispatch_v3(a, as_stream(stream)); }
    a.nchunks = ceil_div(a.cin, kCK27);
    return dispatch<27, kCK27>(a, as_stream(stream));
}

extern "C" int cine_tconv3d_in(const float* x, const float* part_x, int np_x, int mode, const float* wpacked,
                               float* y, float* part_y, int n, int cin, int cout, int d, int h, int w,
                               float eps, float slope, void* stream) {
    if (int e = check_slope(slope, "conv entry point")) return e;
    CINE_REQUIRE(x && wpacked && y, CINE_EINVAL, "cine_tconv3d_in: null pointer");
    CINE_REQUIRE(n > 0 && n <= 65535 && cin > 0 && cout > 0 && d > 0 && h > 0 && w > 0, CINE_EINVAL, "cine_tconv3d_in: bad sizes");
    CINE_REQUIRE(mode == 0 || mode == 1, CINE_EINVAL, "cine_tconv3d_in: mode %d", mode);
    if (int e = check_src(x, part_x, cin, mode, np_x, "cine_tconv3d_in")) return e;
    ConvArgs a{};
    a.s0 = Src{x, part_x, cin, mode, h, w, np_x, 2, d};
    a.vol = 1;
    a.wp0 = a.wp1 = wpacked; a.set_split = n;
    a.y = y; a.ypart = part_y; a.n = n; a.cin = cin; a.rows = 8 * cout; a.rowsp = ceil_div(8 * cout, 16) * 16;
    a.tconv_cout = cout; a.H = h; a.W = w; a.D = d; a.slope = slope; a.eps = eps; a.nchunks = ceil_div(cin, kCK1);
    if (vol1x1_fast_ok(a)) return dispatch_vol1x1(a, as_stream(stream));
    return dispatch<1, kCK1>(a, as_stream(stream));
}

extern "C" int cine_conv1x1x1_bias(const float* x, const float* part_x, int np_x, int mode, const float* wpacked,
                                   const float* bias, float* y, int n, int cin, int cout, int d, int h, int w,
                                   float eps, float slope, void* stream) {
    if (int e = check_slope(slope, "conv entry point")) return e;
    CINE_REQUIRE(x && wpacked && bias && y, CINE_EINVAL, "cine_conv1x1x1_bias: null pointer");
    CINE_REQUIRE(n > 0 && n <= 65535 && cin > 0 && cout > 0 && d > 0 && h > 0 && w > 0, CINE_EINVAL, "cine_conv1x1x1_bias: bad sizes");
    CINE_REQUIRE(mode == 0 || mode == 1, CINE_EINVAL, "cine_conv1x1x1_bias: mode %d", mode);
    if (int e = check_src(x, part_x, cin, mode, np_x, "cine_conv1x1x1_bias")) return e;
    ConvArgs a{};
    a.s0 = Src{x, part_x, cin, mode, h, w, np_x, 2, d};
    a.vol = 1;
    a.wp0 = a.wp1 = wpacked; a.set_split = n; a.bias = a.bias1 = bias;
    a.y = y; a.n = n; a.cin = cin; a.rows = cout; a.rowsp = ceil_div(cout, 16) * 16;
    a.H = h; a.W = w; a.D = d; a.slope = slope; a.eps = eps; a.nchunks = ceil_div(cin, kCK1);
    if (vol1x1_fast_ok(a)) return dispatch_vol1x1(a, as_stream(stream));
    return dispatch<1, kCK1>(a, as_stream(stream));
}

// np partial records per plane -> ONE record per plane (volumes emit one record per tile and depth slice; merging
// them once keeps the consumers' prologue short)
// one wave per plane: the records are summed lane-strided, then across the wave (same two-pass formula as merge_partials;
// a volume has hundreds of records per plane and only n * cout planes -- one thread per plane took 115 us per call at cfg 4)
__global__ __launch_bounds__(256) void instnorm_merge_kernel(const float* part, float* out, long planes, int np) {
    const long i = (long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (i >= planes) return;
    const int lane = threadIdx.x & 63;
    const float* p = part + i * np * 3;
    float cnt = 0.f, wsum = 0.f;
    for (int k = lane; k < np; k += 64) { cnt += p[3 * k]; wsum += p[3 * k] * p[3 * k + 1]; }
    cnt = wave_sum_f(cnt); wsum = wave_sum_f(wsum);
    const float mean = wsum / cnt;
    float m2 = 0.f;
    for (int k = lane; k < np; k += 64) { const float dlt = p[3 * k + 1] - mean; m2 += p[3 * k + 2] + p[3 * k] * dlt * dlt; }
    m2 = wave_sum_f(m2);
    if (lane == 0) { out[3 * i] = cnt; out[3 * i + 1] = mean; out[3 * i + 2] = m2; }
}
extern "C" int cine_instnorm_merge(const float* part, float* out, long planes, int np, void* stream) {
    CINE_REQUIRE(part && out && planes > 0 && np > 0, CINE_EINVAL, "cine_instnorm_merge: bad arguments");
    ProfScope prof(F_STATS, as_stream(stream));
    hipLaunchKernelGGL(instnorm_merge_kernel, dim3((unsigned)ceil_div(planes, 4L)), dim3(256), 0, as_stream(stream),
                       part, out, planes, np);
    return check_launch("instnorm_merge_kernel");
}
