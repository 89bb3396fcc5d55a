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
//   * one workgroup = 64 output rows x 16 MT positions of one slice; its four waves split the K dimension (8-channel chunks
//     round-robin, every depth offset) and each holds the whole 64 x 16 MT accumulator tile: per (tap, 4 channels) a wave reads
//     MT A operands from ITS OWN staged run in LDS (wave-private: no workgroup barrier in the chunk loop) and streams 4 B operands
//     straight from the packed weights in L2, three taps ahead -- 4 MT MFMAs per 4 + MT operand loads;
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
    const float* wp; const float* bias; const float* addend; int relu;
    float* y; float* ypart;
    int cin, rows, rowsp, D, H, W, Wp, ncc, tiles_z, tiles;
    int nph, ps;                         // staged positions per channel (16 MT + 2 (Wp + 1)); LDS channel stride, == 16 (mod 32)
    float slope, eps;
};

constexpr int kNS = 3;                   // position slots per lane of a staged run (<= 192 positions)

template <int MT>
__global__ __launch_bounds__(256, 2) void conv_coarse_kernel(CoarseArgs a) {
    constexpr int NP = 16 * MT;          // output positions of the workgroup
    constexpr int RS = NP + 4;           // row stride of the partial tiles in LDS
    constexpr int PPT = NP / 4;          // positions per thread in the final pass
    extern __shared__ __align__(16) float smem_c[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int q = lane & 15, kk = lane >> 4;
    const int n = blockIdx.z, co0 = blockIdx.y * 64, tile = blockIdx.x;
    const int z0 = tile / a.tiles_z, P0 = (tile - z0 * a.tiles_z) * NP;
    const int nch = a.s0.c + a.s1.c, nchp = (nch + 1) & ~1;
    float* st_lds = smem_c;                                  // {scale, shift} per input channel (act())
    float* xw = smem_c + 2 * nchp + wave * (8 * a.ps);       // this wave's staged unit: 8 channels x nph positions
    float* red = smem_c + 2 * nchp + 4 * 8 * a.ps;           // [4 waves][64 rows][RS]

    // ---- InstanceNorm table of the input channels (records are usually merged already: np = 1)
    for (int ci = tid; ci < nch; ci += 256) {
        const bool first = ci < a.s0.c;
        const Src& s = first ? a.s0 : a.s1;
        const int cl = first ? ci : ci - a.s0.c;
        float2 mr = make_float2(0.f, 1.f);
        if (s.mode != 0) mr = merge_partials(s.part + ((long)n * s.c + cl) * s.np * 3, s.np, a.eps);
        st_lds[2 * ci] = mr.y; st_lds[2 * ci + 1] = -mr.x * mr.y;
    }

    // ---- my position slots of a staged run: run position p <-> padded position P0 - (Wp + 1) + p of the slice
    int off0[kNS], off1[kNS];            // element offset inside a (channel, slice) plane of source 0 / 1; -1: reads as zero
#pragma unroll
    for (int j = 0; j < kNS; ++j) {
        const int p = lane + 64 * j;
        const int pin = P0 - (a.Wp + 1) + p + 2 * a.Wp;      // >= 0
        const int yy = pin / a.Wp - 2, xx = pin - (yy + 2) * a.Wp;
        const bool in = p < a.nph && yy >= 0 && yy < a.H && xx < a.W;
        auto soff = [&](const Src& s) {
            if (!in || s.c == 0) return -1;
            if (s.mode == 2) return (2 * yy + 1 < s.h && 2 * xx + 1 < s.w) ? 2 * yy * s.w + 2 * xx : -1;
            return (yy < s.h && xx < s.w) ? yy * s.w + xx : -1;
        };
        off0[j] = soff(a.s0); off1[j] = soff(a.s1);
    }

    // ---- units of this wave: (depth offset, 8-channel chunk), chunks wave, wave + 4, ...; dead depth offsets are skipped
    const int dz_lo = z0 == 0 ? 1 : 0, dz_hi = z0 == a.D - 1 ? 1 : 2;
    const int nccw = wave < a.ncc ? (a.ncc - wave + 3) / 4 : 0;
    const int nunits = (dz_hi - dz_lo + 1) * nccw;
    auto unit_dz = [&](int u) { return dz_lo + u / nccw; };
    auto unit_cc = [&](int u) { return wave + 4 * (u % nccw); };

    // channel ci of the layer input -> source, channel inside it (uniform)
    auto chan = [&](int ci, int& cl) -> const Src& { const bool f = ci < a.s0.c; cl = f ? ci : ci - a.s0.c; return f ? a.s0 : a.s1; };

    float xraw[kNS][8];
    // raw loads of unit u (plain / normalised sources; pooled ones are fetched in commit)
    auto issue = [&](int u) {
        const int zs = z0 + unit_dz(u) - 1, ci0 = unit_cc(u) * 8;
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            const int ci = ci0 + c;
            int cl;
            const Src& s = chan(min(ci, a.cin - 1), cl);
            const bool live = ci < a.cin && s.mode != 2 && zs < s.d;
            const float* sb = s.x + (((long)n * s.c + cl) * s.d + min(zs, s.d - 1)) * (long)s.h * s.w;
            const bool f0 = ci < a.s0.c;
#pragma unroll
            for (int j = 0; j < kNS; ++j) {
                const int o = f0 ? off0[j] : off1[j];
                xraw[j][c] = (live && o >= 0) ? sb[o] : 0.f;
            }
        }
    };
    auto commit = [&](int u) {
        const int zs = z0 + unit_dz(u) - 1, ci0 = unit_cc(u) * 8;
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            const int ci = ci0 + c;
            int cl;
            const Src& s = chan(min(ci, a.cin - 1), cl);
            const bool f0 = ci < a.s0.c;
            const float sc = st_lds[2 * min(ci, a.cin - 1)], sh = st_lds[2 * min(ci, a.cin - 1) + 1];
            if (ci < a.cin && s.mode == 2) {
                // avg_pool3d 2x2x2 of act(x) (unet.py:88,97): two source slices x two rows x two columns, fetch_scalar's summation order
                const bool zok = 2 * zs + 1 < s.d;
                const float* sb = s.x + (((long)n * s.c + cl) * s.d + (zok ? 2 * zs : 0)) * (long)s.h * s.w;
                const long zstr = (long)s.h * s.w;
#pragma unroll
                for (int j = 0; j < kNS; ++j) {
                    const int o = f0 ? off0[j] : off1[j];
                    float v = 0.f;
                    if (zok && o >= 0) {
                        float acc8 = 0.f;
#pragma unroll
                        for (int dzz = 0; dzz < 2; ++dzz) {
                            const float* p = sb + dzz * zstr + o;
                            acc8 += act(p[0], sc, sh, a.slope) + act(p[1], sc, sh, a.slope) + act(p[s.w], sc, sh, a.slope) + act(p[s.w + 1], sc, sh, a.slope);
                        }
                        v = 0.125f * acc8;
                    }
                    if (lane + 64 * j < a.nph) xw[c * a.ps + lane + 64 * j] = v;
                }
            } else {
                const bool live = ci < a.cin && zs < s.d;
                const bool plain = s.mode == 0;
#pragma unroll
                for (int j = 0; j < kNS; ++j) {
                    const int o = f0 ? off0[j] : off1[j];
                    float v = xraw[j][c];
                    if (!plain) v = act(v, sc, sh, a.slope);
                    if (!(live && o >= 0)) v = 0.f;
                    if (lane + 64 * j < a.nph) xw[c * a.ps + lane + 64 * j] = v;
                }
            }
        }
    };

    f32x4 acc[4][MT];
#pragma unroll
    for (int ct = 0; ct < 4; ++ct)
#pragma unroll
        for (int f = 0; f < MT; ++f) acc[ct][f] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // B operands: lane (q, kk) of k-step ks, row tile ct reads W[row co0 + 16 ct + q][channel 4 ks + kk][tap] of the packed weights
    // [dz][8-channel chunk][3x3 tap][8 channels][rowsp] (cine_pack_conv3d)
    bool ctl[4];
#pragma unroll
    for (int ct = 0; ct < 4; ++ct) ctl[ct] = co0 + 16 * ct < a.rowsp;
    float wr[3][2][4];
    auto loadw = [&](int u, int tap, float (&w)[2][4]) {
        const float* wu = a.wp + ((((long)unit_dz(u) * a.ncc + unit_cc(u)) * 9 + tap) * 8 + kk) * a.rowsp + co0 + q;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int ct = 0; ct < 4; ++ct) w[ks][ct] = ctl[ct] ? wu[(long)(4 * ks) * a.rowsp + 16 * ct] : 0.f;
    };

    __syncthreads();                     // the statistics table (the only workgroup-wide dependency before the final pass)
    if (nunits > 0) {
        loadw(0, 0, wr[0]); loadw(0, 1, wr[1]);
        issue(0);
    }
    const int xbase = kk * a.ps + q;
    for (int u = 0; u < nunits; ++u) {
        commit(u);
        if (u + 1 < nunits) issue(u + 1);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            if (tap + 2 < 9) loadw(u, tap + 2, wr[(tap + 2) % 3]);
            else if (u + 1 < nunits) loadw(u + 1, tap + 2 - 9, wr[(tap + 2) % 3]);
            const int toff = (tap / 3) * a.Wp + tap % 3;
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                float xa[MT];
#pragma unroll
                for (int f = 0; f < MT; ++f) xa[f] = xw[xbase + (4 * ks) * a.ps + 16 * f + toff];
#pragma unroll
                for (int ct = 0; ct < 4; ++ct)
#pragma unroll
                    for (int f = 0; f < MT; ++f)
                        acc[ct][f] = __builtin_amdgcn_mfma_f32_16x16x4f32(xa[f], wr[tap % 3][ks][ct], acc[ct][f], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }

    // ---- the four K-split partial tiles meet in LDS; lane (q, kk) holds rows 16 ct + q, positions 16 f + 4 kk .. + 3
#pragma unroll
    for (int ct = 0; ct < 4; ++ct)
#pragma unroll
        for (int f = 0; f < MT; ++f)
            *reinterpret_cast<float4*>(red + (wave * 64 + 16 * ct + q) * RS + 16 * f + 4 * kk) =
                make_float4(acc[ct][f][0], acc[ct][f][1], acc[ct][f][2], acc[ct][f][3]);
    __syncthreads();
    // ---- final pass: thread = (row, PPT consecutive positions); partials added in wave order
    const int row = tid >> 2, pos0 = (tid & 3) * PPT;
    const int m = co0 + row;
    const bool mok = m < a.rows;
    float v[PPT];
#pragma unroll
    for (int i = 0; i < PPT; ++i) v[i] = 0.f;
#pragma unroll
    for (int w = 0; w < 4; ++w)
#pragma unroll
        for (int i = 0; i < PPT; i += 4) {
            const float4 t = *reinterpret_cast<const float4*>(red + (w * 64 + row) * RS + pos0 + i);
            v[i] += t.x; v[i + 1] += t.y; v[i + 2] += t.z; v[i + 3] += t.w;
        }
    const float bv = (a.bias && mok) ? a.bias[m] : 0.f;
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

template <int MT>
int launch_coarse(const CoarseArgs& p, int n, hipStream_t st) {
    auto kern = conv_coarse_kernel<MT>;
    const int nch = p.s0.c + p.s1.c, nchp = (nch + 1) & ~1;
    const size_t lds = (size_t)(2 * nchp + 4 * 8 * p.ps + 4 * 64 * (16 * MT + 4)) * sizeof(float);
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
    hipLaunchKernelGGL(kern, grid, dim3(256), lds, st, p);
    return check_launch("conv_coarse_kernel");
}

}  // namespace

// fragments per workgroup of the coarse kernel for a layer shape: two unless that leaves fewer than 128 workgroups
int coarse_mt(int rowsp, int d, int h, int w) {
    const long tiles2 = (long)ceil_div(h * (w + 1), 32) * d * ceil_div(rowsp, 64);
    return tiles2 >= 128 ? 2 : 1;
}
// statistics records per (sample, channel) = position tiles of the volume
int coarse_tiles(int rowsp, int d, int h, int w) {
    return ceil_div(h * (w + 1), 16 * coarse_mt(rowsp, d, h, w)) * d;
}

int launch_conv_coarse(const ConvArgs& a, hipStream_t st) {
    CINE_REQUIRE(a.vol && !a.add_src1 && a.tconv_cout == 0 && !a.accum && a.pair_n == 0, CINE_EUNSUPPORTED, "conv_coarse_kernel: not a plain 3x3x3 convolution");
    CINE_REQUIRE(a.s0.mode <= 2 && (a.s1.c == 0 || a.s1.mode <= 2), CINE_EUNSUPPORTED, "conv_coarse_kernel: source modes 0..2 only");
    CINE_REQUIRE(a.set_split >= a.n, CINE_EUNSUPPORTED, "conv_coarse_kernel: one weight set");
    const int mt = coarse_mt(a.rowsp, a.D, a.H, a.W);
    CoarseArgs p{};
    p.s0 = a.s0; p.s1 = a.s1;
    if (p.s1.c == 0) { p.s1 = p.s0; p.s1.c = 0; }
    p.wp = a.wp0; p.bias = a.bias; p.addend = a.addend; p.relu = a.relu;
    p.y = a.y; p.ypart = a.ypart;
    p.cin = a.cin; p.rows = a.rows; p.rowsp = a.rowsp; p.D = a.D; p.H = a.H; p.W = a.W; p.Wp = a.W + 1; p.ncc = a.ncc;
    p.tiles_z = ceil_div(a.H * p.Wp, 16 * mt); p.tiles = p.tiles_z * a.D;
    p.nph = 16 * mt + 2 * (p.Wp + 1);
    CINE_REQUIRE(p.nph <= 64 * kNS, CINE_EUNSUPPORTED, "conv_coarse_kernel: rows of %d voxels are too wide", a.W);
    p.ps = ((p.nph + 15) / 32) * 32 + 16;           // >= nph, == 16 (mod 32): the four channels of a k-step land on disjoint banks
    p.slope = a.slope; p.eps = a.eps;
    return mt == 2 ? launch_coarse<2>(p, a.n, st) : launch_coarse<1>(p, a.n, st);
}

}  // namespace cine
