// conv_coarse.hip -- 3x3x3 convolutions of the COARSE levels of the 3-D U-Net (reference denoisers/unet.py:46-49,149-157 with dims = 3:
// Conv3d 3x3x3, pad 1, no bias; cfg 4's 64-channel 3 x 50 x 50 and 128-channel 1 x 25 x 25 volumes).
//
// Such a level is a few thousand voxels x 64 - 128 rows x K = 27 * cin: the regular tilings give 12 - 156 workgroups that each walk
// 16 - 24 staged chunks between barrier pairs -- a latency chain on a fraction of the CUs (round 4: 72 us per layer, PMC MFMA 0.105).
// This kernel is laid out for exactly that regime:
//   * positions of one depth slice are FLATTENED with one shared zero column between the rows (P = y * (W + 1) + x; column W is the
//     right halo of row y and the left halo of row y + 1), so a 16-position MFMA fragment is any 16 consecutive P and tap (dy, dx)
//     reads position P + (dy - 1)(W + 1) + (dx - 1): no fragment is cut at the row end (W = 50: 2 % padding instead of the 22 % of
//     16-wide column tiles);
//   * one workgroup = 64 output rows x 16 MT positions of one slice; its four waves split the K dimension ((depth offset, 8-channel
//     chunk) units dealt round-robin) and each holds the whole 64 x 16 MT accumulator tile: per (tap, 4 channels) a wave reads
//     MT A operands from ITS OWN staged run in LDS (wave-private: no workgroup barrier in the chunk loop) and streams its 4 B operands
//     straight from the packed weights in L2 -- ONE 16-byte load (the row tiles interleave the rows: tile ct = rows 4 q + ct), requested
//     a whole unit (9 taps) ahead -- 4 MT MFMAs per 1 + MT operand loads;
//   * the four partial tiles meet in LDS in a fixed order (wave 0 + 1 + 2 + 3: deterministic), then bias / addend / ReLU, the
//     stores, and this layer's InstanceNorm record {count, mean, M2} per (row, tile).
// 240 workgroups for cfg 4's 3 x 50 x 50 level (was 156 with 24 barrier pairs each), 82 for 1 x 25 x 25 (was 14 - 28).
// Sources: plain / InstanceNorm + LeakyReLU / the same + 2x2x2 average pool (unet.py:88,97), one or two of them (concat, unet.py:122),
// extents that end before the output's (the up path's zero pad, unet.py:106-120) -- everything cine_conv3d_in accepts, staged
// position by position (the data is L2-resident; 4-byte loads in runs of 64 consecutive positions).
#include <mutex>
#include "common.h"
#include "conv_src.h"
#include "conv_cfg.h"

namespace cine {
namespace {

struct CoarseArgs {
    Src s0, s1;
    const float* wp; const float* wp1; const float* bias; const float* bias1; int set_split;     // samples >= set_split: the second weight set
    const float* addend; int relu;
    float* y; float* ypart;
    int cin, rows, rowsp, D, H, W, Wp, ncc, tiles_z, tiles;
    int wz_off;                          // 1: 2-D weights [chunk][tap][8][rowsp] (no depth-offset blocks); 0: the 3-D packing [dz][chunk][tap][8][rowsp]
    int add_src1;                        // GEN staging only: source 1 is added to source 0 (mwcnn.py:164,172)
    int nph;                             // staged positions per channel: 16 MT + 2 (Wp + 1)
    float slope, eps;
};

// NS: 64-position slots per lane of a staged run (2 or 3); POOL: some source is 2x2x2-pooled on load (its eight loads per value happen in
// commit); RAGGED: a chunk may hold fewer than 8 channels or channels of both sources (per-channel source selection).  The common
// instantiation (whole chunks, one source per chunk, no pooling) has a branch-free unit loop with ~150 vector instructions of staging
// per unit; the first version selected the source per channel and masked every LDS write: ~1 000 instructions per unit on a SIMD
// that runs ONE wave (240 workgroups on 256 CUs) -- as long as the unit's 144 MFMAs.
#ifndef CINE_COARSE_WAVES                // (diagnostic builds: tools/build_variant.sh ... -DCINE_COARSE_WAVES=8)
#define CINE_COARSE_WAVES 4
#endif
// K-split width = waves per workgroup.  Measured on cfg 4 (tools/ab_variants.sh): 8 waves (two per SIMD, 128 KB of LDS, one workgroup per CU)
// 23.4 / 43.6 us per 64 -> 64 / 128 -> 64 layer alone and 159.4 slices/s with ten slices in flight; 4 waves (55 KB: a second stream's
// workgroup fits beside it) 24.8 / 47.4 us and 163.3 slices/s -- the same latency for one slice (100.4 slices/s either way)
constexpr int kCoarseWaves = CINE_COARSE_WAVES;
// STAGE: 0 whole chunks of one source, 1 = RAGGED, 2 = GEN: any source cine_conv3x3_ex accepts (Haar DWT / IWT on load, added skips, narrower
// extents) element by element through fetch_scalar -- the fallback that keeps the statistics-record count a function of the layer SHAPE
template <int MT, int NS, bool POOL, int STAGE>
__global__ __launch_bounds__(64 * kCoarseWaves, (kCoarseWaves > 4 ? 1 : 2)) void conv_coarse_kernel(CoarseArgs a) {
    constexpr bool RAGGED = STAGE == 1, GEN = STAGE == 2;
    constexpr int NW = kCoarseWaves;
    constexpr int NP = 16 * MT;          // output positions of the workgroup
    constexpr int RS = NP + 4;           // row stride of the partial tiles in LDS
    constexpr int PPT = NP / 4;          // positions per thread in the final pass
    constexpr int PS = 64 * NS + 16;     // LDS channel stride of a staged unit: every slot of every lane exists, == 16 (mod 32)
    extern __shared__ __align__(16) float smem_c[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);      // uniform: the unit list, source selection and modes stay scalar
    const int q = lane & 15, kk = lane >> 4;
    const int n = blockIdx.z, co0 = blockIdx.y * 64, tile = blockIdx.x;
    const int z0 = tile / a.tiles_z, P0 = (tile - z0 * a.tiles_z) * NP;
    const int nch = a.s0.c + a.s1.c, nchp = (nch + 1) & ~1;
    float* st_lds = smem_c;                                  // {scale, shift} per input channel (act())
    float* xw = smem_c + 2 * nchp + wave * (8 * PS);         // this wave's staged unit: 8 channels x 64 NS positions
    float* red = smem_c + 2 * nchp + NW * 8 * PS;            // [NW waves][64 rows][RS]

    // ---- InstanceNorm table of the input channels (records are usually merged already: np = 1)
    for (int ci = tid; ci < nch; ci += 64 * NW) {
        const bool first = ci < a.s0.c;
        const Src& s = first ? a.s0 : a.s1;
        const int cl = first ? ci : ci - a.s0.c;
        float2 mr = make_float2(0.f, 1.f);
        if (s.mode == 1 || s.mode == 2 || (s.mode >= 3 && (s.act & 1))) mr = merge_partials(s.part + ((long)n * s.c + cl) * s.np * 3, s.np, a.eps);
        st_lds[2 * ci] = mr.y; st_lds[2 * ci + 1] = -mr.x * mr.y;
    }
    const float* const wpn = n >= a.set_split ? a.wp1 : a.wp;

    // ---- my position slots of a staged run: run position p <-> padded position P0 - (Wp + 1) + p of the slice
    int off0[NS], off1[NS];              // element offset inside a (channel, slice) plane of source 0 / 1; -1: reads as zero
    int pyy[GEN ? NS : 1], pxx[GEN ? NS : 1];                 // GEN: the slot's (row, column), -1: outside the plane
#pragma unroll
    for (int j = 0; j < NS; ++j) {
        const int p = lane + 64 * j;
        const int pin = P0 - (a.Wp + 1) + p + 2 * a.Wp;      // >= 0
        const int yy = pin / a.Wp - 2, xx = pin - (yy + 2) * a.Wp;
        const bool in = p < a.nph && yy >= 0 && yy < a.H && xx < a.W;
        if constexpr (GEN) { pyy[j] = in ? yy : -1; pxx[j] = xx; }
        auto soff = [&](const Src& s) {
            if (!in || s.c == 0) return -1;
            if (s.mode == 2) return (2 * yy + 1 < s.h && 2 * xx + 1 < s.w) ? 2 * yy * s.w + 2 * xx : -1;
            return (yy < s.h && xx < s.w) ? yy * s.w + xx : -1;
        };
        off0[j] = soff(a.s0); off1[j] = soff(a.s1);
    }

    // ---- units of this wave: the live (depth offset, 8-channel chunk) pairs k = dz * ncc + cc of the tile, dealt round-robin: k = wave, wave + NW, ...
    const int dz_lo = z0 == 0 ? 1 : 0, dz_hi = z0 == a.D - 1 ? 1 : 2;
    const int nk = (dz_hi - dz_lo + 1) * a.ncc;
    const int nunits = wave < nk ? (nk - wave + NW - 1) / NW : 0;
    auto unit_dz = [&](int u) { return dz_lo + (wave + NW * u) / a.ncc; };
    auto unit_cc = [&](int u) { return (wave + NW * u) % a.ncc; };

    // channel ci of the layer input -> source, channel inside it (uniform)
    auto chan = [&](int ci, int& cl) -> const Src& { const bool f = ci < a.s0.c; cl = f ? ci : ci - a.s0.c; return f ? a.s0 : a.s1; };

    float xraw[NS][8];
    // raw loads of unit u (plain / normalised sources; pooled ones are fetched in commit).  Every load is UNCONDITIONAL (clamped address,
    // the value is selected in commit): a conditional load is a branch, and a branch makes the compiler wait for every load in flight
    auto issue = [&](int u) {
        const int zs = z0 + unit_dz(u) - 1, ci0 = unit_cc(u) * 8;
        if constexpr (GEN) {
            (void)zs; (void)ci0;
        } else if constexpr (!RAGGED) {
            const bool f0 = ci0 < a.s0.c;                        // the whole chunk lies in one source
            const Src& s = f0 ? a.s0 : a.s1;
            const long plane = (long)s.h * s.w;
            const char* sb = reinterpret_cast<const char*>(s.x + (((long)n * s.c + (f0 ? ci0 : ci0 - a.s0.c)) * s.d + min(zs, s.d - 1)) * plane);
            const long cstr = (long)s.d * plane * 4;
            unsigned vo[NS];
#pragma unroll
            for (int j = 0; j < NS; ++j) vo[j] = (unsigned)max(f0 ? off0[j] : off1[j], 0) * 4u;
#pragma unroll
            for (int c = 0; c < 8; ++c)
#pragma unroll
                for (int j = 0; j < NS; ++j) xraw[j][c] = *reinterpret_cast<const float*>(sb + c * cstr + vo[j]);
        } else {
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                const int ci = min(ci0 + c, a.cin - 1);
                int cl;
                const Src& s = chan(ci, cl);
                const bool f0 = ci < a.s0.c;
                const float* sb = s.x + (((long)n * s.c + cl) * s.d + min(zs, s.d - 1)) * (long)s.h * s.w;     // (a pooled source: a valid address, the value is not used)
#pragma unroll
                for (int j = 0; j < NS; ++j) xraw[j][c] = sb[max(f0 ? off0[j] : off1[j], 0)];
            }
        }
    };
    // avg_pool3d 2x2x2 of act(x) (unet.py:88,97) for channel cl of source s into LDS channel c: two source slices x two rows x two
    // columns, fetch_scalar's summation order; invalid slots load one valid element eight times
    auto commit_pooled = [&](const Src& s, int cl, bool f0, int c, int zs, bool cok, float sc, float sh) {
        const bool vol = s.act & 2;                              // a 2-D plane pools 2 x 2 (unet.py:97 with dims = 2): the second "slice" is the first again, weight 1/8 each
        const bool zok = cok && (!vol || 2 * zs + 1 < s.d);
        const float* sb = s.x + (((long)n * s.c + cl) * s.d + (zok && vol ? 2 * zs : 0)) * (long)s.h * s.w;
        const long zstr = zok && vol ? (long)s.h * s.w : 0;
        float t[NS][8];
#pragma unroll
        for (int j = 0; j < NS; ++j) {
            const int o = f0 ? off0[j] : off1[j];
            const float* p = sb + max(o, 0);
            const int dw = o >= 0 ? s.w : 0, d1 = o >= 0 ? 1 : 0;
#pragma unroll
            for (int dzz = 0; dzz < 2; ++dzz) {
                t[j][4 * dzz] = p[dzz * zstr]; t[j][4 * dzz + 1] = p[dzz * zstr + d1];
                t[j][4 * dzz + 2] = p[dzz * zstr + dw]; t[j][4 * dzz + 3] = p[dzz * zstr + dw + d1];
            }
        }
#pragma unroll
        for (int j = 0; j < NS; ++j) {
            const int o = f0 ? off0[j] : off1[j];
            float acc8 = 0.f;
#pragma unroll
            for (int dzz = 0; dzz < 2; ++dzz)
                acc8 += act(t[j][4 * dzz], sc, sh, a.slope) + act(t[j][4 * dzz + 1], sc, sh, a.slope) +
                        act(t[j][4 * dzz + 2], sc, sh, a.slope) + act(t[j][4 * dzz + 3], sc, sh, a.slope);
            xw[c * PS + lane + 64 * j] = (zok && o >= 0) ? 0.125f * acc8 : 0.f;
        }
    };
    auto commit = [&](int u) {
        const int zs = z0 + unit_dz(u) - 1, ci0 = unit_cc(u) * 8;
        if constexpr (GEN) {
            const int c0n = src_cin(a.s0);
#pragma unroll 1
            for (int c = 0; c < 8; ++c) {
                const int ci = ci0 + c;
                const bool f0 = a.add_src1 || ci < c0n;
#pragma unroll
                for (int j = 0; j < NS; ++j) {
                    float v = 0.f;
                    if (ci < a.cin && pyy[j] >= 0) {
                        v = fetch_scalar(f0 ? a.s0 : a.s1, n, f0 ? ci : ci - c0n, zs, pyy[j], pxx[j], st_lds + (f0 ? 0 : 2 * a.s0.c), a.slope);
                        if (a.add_src1) v += fetch_scalar(a.s1, n, ci, zs, pyy[j], pxx[j], st_lds + 2 * a.s0.c, a.slope);
                    }
                    xw[c * PS + lane + 64 * j] = v;
                }
            }
        } else if constexpr (!RAGGED) {
            const bool f0 = ci0 < a.s0.c;
            const Src& s = f0 ? a.s0 : a.s1;
            if (POOL && s.mode == 2) {
#pragma unroll
                for (int c = 0; c < 8; ++c) {
                    const float2 ss = *reinterpret_cast<const float2*>(st_lds + 2 * (ci0 + c));
                    commit_pooled(s, (f0 ? ci0 : ci0 - a.s0.c) + c, f0, c, zs, true, ss.x, ss.y);
                }
                return;
            }
            const bool live = zs < s.d;                          // a shorter `up` volume reads as zero behind its end (unet.py:106-120)
            bool ok[NS];
#pragma unroll
            for (int j = 0; j < NS; ++j) ok[j] = live && (f0 ? off0[j] : off1[j]) >= 0;
            if (s.mode == 0) {
#pragma unroll
                for (int c = 0; c < 8; ++c)
#pragma unroll
                    for (int j = 0; j < NS; ++j) xw[c * PS + lane + 64 * j] = ok[j] ? xraw[j][c] : 0.f;
            } else {
#pragma unroll
                for (int c = 0; c < 8; ++c) {
                    const float2 ss = *reinterpret_cast<const float2*>(st_lds + 2 * (ci0 + c));
#pragma unroll
                    for (int j = 0; j < NS; ++j) xw[c * PS + lane + 64 * j] = ok[j] ? act(xraw[j][c], ss.x, ss.y, a.slope) : 0.f;
                }
            }
        } else {
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                const int ci = min(ci0 + c, a.cin - 1);
                const bool cok = ci0 + c < a.cin;
                int cl;
                const Src& s = chan(ci, cl);
                const bool f0 = ci < a.s0.c;
                const float sc = st_lds[2 * ci], sh = st_lds[2 * ci + 1];
                if (POOL && s.mode == 2) {
                    commit_pooled(s, cl, f0, c, zs, cok, sc, sh);
                } else {
                    const bool live = cok && zs < s.d;
                    const bool plain = s.mode == 0;
#pragma unroll
                    for (int j = 0; j < NS; ++j) {
                        const int o = f0 ? off0[j] : off1[j];
                        float v = xraw[j][c];
                        if (!plain) v = act(v, sc, sh, a.slope);
                        xw[c * PS + lane + 64 * j] = (live && o >= 0) ? v : 0.f;
                    }
                }
            }
        }
    };

    f32x4 acc[4][MT];
#pragma unroll
    for (int ct = 0; ct < 4; ++ct)
#pragma unroll
        for (int f = 0; f < MT; ++f) acc[ct][f] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // B operands from the packed weights [dz][8-channel chunk][3x3 tap][8 channels][rowsp] (cine_pack_conv3d): lane (q, kk) of row tile
    // ct works on output row co0 + 4 q + ct, so its four B operands of a k-step are ONE 16-byte load (rows 4 q .. 4 q + 3 of channel
    // 4 ks + kk; a wave-load = four 256-byte runs).  All nine taps of a unit sit in registers; tap t of the NEXT unit is requested as
    // soon as tap t of this one has been consumed -- a prefetch distance of a whole unit (288 MFMAs at MT = 2).
    // (row quads past the padded row count read the last quad again: their accumulators are never stored)
    const unsigned wlane = (unsigned)(kk * a.rowsp + min(co0 + 4 * q, a.rowsp - 4)) * 4u;     // my byte offset inside a (unit, tap) block; the block base is scalar
    const unsigned wks = (unsigned)(4 * a.rowsp) * 4u;
    float4 wr[9][2];
    auto loadw = [&](int u, int tap, float4 (&w)[2]) {
        const char* wu = reinterpret_cast<const char*>(wpn + (((long)(unit_dz(u) - a.wz_off) * a.ncc + unit_cc(u)) * 9 + tap) * 8 * a.rowsp);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) w[ks] = *reinterpret_cast<const float4*>(wu + (wlane + ks * wks));
    };

    __syncthreads();                     // the statistics table (the only workgroup-wide dependency before the final pass)
    if (nunits > 0) {
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) loadw(0, tap, wr[tap]);
        issue(0);
    }
    const int xbase = kk * PS + q;
    for (int u = 0; u < nunits; ++u) {
        const int un = min(u + 1, nunits - 1);      // the prefetches are unconditional (the last unit re-reads itself): no branch, no lost wait counts
        commit(u);
        issue(un);
        __builtin_amdgcn_sched_barrier(0);
        // 18 operand groups (tap, k-step); the A operands of group g + 1 are read from LDS before the MFMAs of group g are issued (this
        // wave is alone on its SIMD: nobody else hides the LDS latency)
        float xa[2][MT];
        auto loadx = [&](int g, float (&x)[MT]) {
            const int tap = g >> 1, ks = g & 1, toff = (tap / 3) * a.Wp + tap % 3;
#pragma unroll
            for (int f = 0; f < MT; ++f) x[f] = xw[xbase + (4 * ks) * PS + 16 * f + toff];
        };
        loadx(0, xa[0]);
#pragma unroll
        for (int g = 0; g < 18; ++g) {
            const int tap = g >> 1, ks = g & 1;
            if (g + 1 < 18) loadx(g + 1, xa[(g + 1) & 1]);
            __builtin_amdgcn_sched_barrier(0);          // (the scheduler otherwise sinks these reads behind the MFMAs, next to their use)
            const float wv[4] = {wr[tap][ks].x, wr[tap][ks].y, wr[tap][ks].z, wr[tap][ks].w};
#pragma unroll
            for (int ct = 0; ct < 4; ++ct)
#pragma unroll
                for (int f = 0; f < MT; ++f)
                    acc[ct][f] = __builtin_amdgcn_mfma_f32_16x16x4f32(xa[g & 1][f], wv[ct], acc[ct][f], 0, 0, 0);
            if (ks == 1) loadw(un, tap, wr[tap]);
            __builtin_amdgcn_sched_barrier(0);
        }
    }

    // ---- the K-split partial tiles meet in LDS; lane (q, kk) holds rows 4 q + ct, positions 16 f + 4 kk .. + 3
#pragma unroll
    for (int ct = 0; ct < 4; ++ct)
#pragma unroll
        for (int f = 0; f < MT; ++f)
            *reinterpret_cast<float4*>(red + (wave * 64 + 4 * q + ct) * RS + 16 * f + 4 * kk) =
                make_float4(acc[ct][f][0], acc[ct][f][1], acc[ct][f][2], acc[ct][f][3]);
    __syncthreads();
    if (tid >= 256) return;              // (no barrier behind this point)
    // ---- final pass: thread = (row, PPT consecutive positions); partials added in wave order
    const int row = tid >> 2, pos0 = (tid & 3) * PPT;
    const int m = co0 + row;
    const bool mok = m < a.rows;
    float v[PPT];
#pragma unroll
    for (int i = 0; i < PPT; ++i) v[i] = 0.f;
#pragma unroll
    for (int w = 0; w < NW; ++w)
#pragma unroll
        for (int i = 0; i < PPT; i += 4) {
            const float4 t = *reinterpret_cast<const float4*>(red + (w * 64 + row) * RS + pos0 + i);
            v[i] += t.x; v[i + 1] += t.y; v[i + 2] += t.z; v[i + 3] += t.w;
        }
    const float bv = (a.bias && mok) ? (n >= a.set_split ? a.bias1 : a.bias)[m] : 0.f;
    const long plane = (((long)n * a.rows + (mok ? m : 0)) * a.D + z0) * (long)a.H * a.W;
    float cnt = 0.f, sum = 0.f;
    bool ok[PPT];
#pragma unroll
    for (int i = 0; i < PPT; ++i) {
        const int P = P0 + pos0 + i;
        const int yy = P / a.Wp, xx = P - yy * a.Wp;
        ok[i] = yy < a.H && xx < a.W;
        const long o = plane + (long)yy * a.W + xx;
        float t = v[i] + bv;
        if (a.addend && ok[i] && mok) t += a.addend[o];
        if (a.relu) t = fmaxf(t, 0.f);
        v[i] = t;
        if (ok[i] && mok) a.y[o] = t;
        cnt += ok[i] ? 1.f : 0.f; sum += ok[i] ? t : 0.f;
    }
    if (a.ypart) {
        // InstanceNorm record of this (row, tile): exact two-pass over the 4 lanes that share the row
        cnt += __shfl_xor(cnt, 1, 64); cnt += __shfl_xor(cnt, 2, 64);
        sum += __shfl_xor(sum, 1, 64); sum += __shfl_xor(sum, 2, 64);
        const float mean = cnt > 0.f ? sum / cnt : 0.f;
        float m2 = 0.f;
#pragma unroll
        for (int i = 0; i < PPT; ++i) { const float d = v[i] - mean; m2 += ok[i] ? d * d : 0.f; }
        m2 += __shfl_xor(m2, 1, 64); m2 += __shfl_xor(m2, 2, 64);
        if (mok && (tid & 3) == 0) {
            float* o = a.ypart + (((long)n * a.rows + m) * a.tiles + tile) * 3;
            o[0] = cnt; o[1] = mean; o[2] = m2;
        }
    }
}

template <int MT, int NS, bool POOL, int STAGE>
int launch_coarse(const CoarseArgs& p, int n, hipStream_t st) {
    auto kern = conv_coarse_kernel<MT, NS, POOL, STAGE>;
    const int nch = p.s0.c + p.s1.c, nchp = (nch + 1) & ~1;
    const size_t lds = (size_t)(2 * nchp + kCoarseWaves * 8 * (64 * NS + 16) + kCoarseWaves * 64 * (16 * MT + 4)) * sizeof(float);
    CINE_REQUIRE(lds <= 160 * 1024, CINE_EUNSUPPORTED, "conv_coarse_kernel: %d input channels need %zu bytes of LDS", p.cin, lds);
    if (lds > 64 * 1024) {
        static std::once_flag once[64];
        static hipError_t status[64];
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess) dev = 0;
        CINE_REQUIRE(dev >= 0 && dev < 64, CINE_EUNSUPPORTED, "conv_coarse_kernel: device index %d", dev);
        std::call_once(once[dev], [&] {
            status[dev] = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        });
        CINE_REQUIRE(status[dev] == hipSuccess, CINE_EHIP, "conv_coarse_kernel: hipFuncSetAttribute: %s", hipGetErrorString(status[dev]));
    }
    const dim3 grid(p.tiles, ceil_div(p.rowsp, 64), n);
    ProfScope prof(F_CONV3, st);
    hipLaunchKernelGGL(kern, grid, dim3(64 * kCoarseWaves), lds, st, p);
    return check_launch("conv_coarse_kernel");
}
template <int MT, int NS>
int launch_coarse_variant(const CoarseArgs& p, int n, bool pool, int stage, hipStream_t st) {
    if (stage == 2) return launch_coarse<MT, NS, false, 2>(p, n, st);
    if (stage == 1) return pool ? launch_coarse<MT, NS, true, 1>(p, n, st) : launch_coarse<MT, NS, false, 1>(p, n, st);
    return pool ? launch_coarse<MT, NS, true, 0>(p, n, st) : launch_coarse<MT, NS, false, 0>(p, n, st);
}

}  // namespace

// fragments per workgroup of the coarse kernel for a layer shape: two unless that leaves fewer than 128 workgroups per volume (2-D planes
// come many to a launch -- the sensitivity network's 15 coil planes: always two)
int coarse_mt(int rowsp, int d, int h, int w, bool vol) {
    if (!vol) return 2;
    const long tiles2 = (long)ceil_div(h * (w + 1), 32) * d * ceil_div(rowsp, 64);
    return tiles2 >= 128 ? 2 : 1;
}
// statistics records per (sample, channel) = position tiles of the volume / plane
int coarse_tiles(int rowsp, int d, int h, int w, bool vol) {
    return ceil_div(h * (w + 1), 16 * coarse_mt(rowsp, d, h, w, vol)) * d;
}
// what the fast staging paths cover; everything else that reaches a coarse SHAPE runs the GEN staging (or, without a statistics record to
// keep in step, the general kernel: launch_conv_coarse returns *handled = false)
static bool coarse_fast_ok(const ConvArgs& a) {
    auto ok = [&](const Src& s) { return s.c == 0 || s.mode <= 2; };
    return !a.add_src1 && ok(a.s0) && ok(a.s1);
}

int launch_conv_coarse(const ConvArgs& a, hipStream_t st, bool* handled) {
    *handled = false;
    if (a.tconv_cout != 0 || a.accum || a.gate || a.pair_n != 0 || a.n <= 0 || a.n > 65535) {
        CINE_REQUIRE(!a.ypart, CINE_EUNSUPPORTED, "conv_coarse_kernel: this epilogue has no statistics-compatible fallback");
        return CINE_OK;                                   // (CRNN second outputs / pair launches on a coarse shape: the general kernel, no records involved)
    }
    const bool fast = coarse_fast_ok(a);
    if (!fast && !a.ypart) return CINE_OK;
    *handled = true;
    const bool vol = a.vol != 0;
    const int mt = coarse_mt(a.rowsp, a.D, a.H, a.W, vol);
    CoarseArgs p{};
    p.s0 = a.s0; p.s1 = a.s1;
    if (p.s1.c == 0) { p.s1 = p.s0; p.s1.c = 0; }
    p.wp = a.wp0; p.wp1 = a.wp1 ? a.wp1 : a.wp0; p.bias = a.bias; p.bias1 = a.bias1 ? a.bias1 : a.bias; p.set_split = a.set_split;
    p.addend = a.addend; p.relu = a.relu;
    p.y = a.y; p.ypart = a.ypart;
    p.cin = a.cin; p.rows = a.rows; p.rowsp = a.rowsp; p.D = a.D; p.H = a.H; p.W = a.W; p.Wp = a.W + 1;
    p.ncc = ceil_div(a.cin, 8); p.wz_off = vol ? 0 : 1; p.add_src1 = a.add_src1;
    p.tiles_z = ceil_div(a.H * p.Wp, 16 * mt); p.tiles = p.tiles_z * a.D;
    p.nph = 16 * mt + 2 * (p.Wp + 1);
    CINE_REQUIRE(p.nph <= 192, CINE_EUNSUPPORTED, "conv_coarse_kernel: rows of %d voxels are too wide", a.W);
    p.slope = a.slope; p.eps = a.eps;
    const bool pool = fast && (a.s0.mode == 2 || (a.s1.c > 0 && a.s1.mode == 2));
    const int stage = !fast ? 2 : ((a.cin % 8 != 0 || (a.s1.c > 0 && a.s0.c % 8 != 0)) ? 1 : 0);
    if (p.nph <= 128) return mt == 2 ? launch_coarse_variant<2, 2>(p, a.n, pool, stage, st) : launch_coarse_variant<1, 2>(p, a.n, pool, stage, st);
    return mt == 2 ? launch_coarse_variant<2, 3>(p, a.n, pool, stage, st) : launch_coarse_variant<1, 3>(p, a.n, pool, stage, st);
}

}  // namespace cine
