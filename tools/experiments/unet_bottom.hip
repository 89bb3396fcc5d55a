// unet_bottom.hip -- the bottom of the U-Net (reference denoisers/unet.py:94-123, levels P-1 and P of a 3-pool, 16-channel
// U-Net on <= 16-wide planes) as ONE kernel per plane:
//
//   pool(L1 out) -> conv 32->64 -> IN+LReLU -> conv 64->64 -> IN+LReLU  (= skip, also pooled)
//                -> conv 64->128 -> IN+LReLU -> conv 128->128 -> IN+LReLU -> conv-transpose 128->64 -> IN+LReLU
//                -> conv cat(up, skip) 128->64 -> IN+LReLU -> conv 64->64 -> raw output + InstanceNorm record
//
// seven of the 17 launches of a U-Net pass.  A plane at these levels is 52 x 4 (64 channels: 53 KB) / 26 x 2 (128 channels:
// 27 KB), so ONE workgroup keeps the activation chain in LDS, in the halo layout the MFMA operand reads want: InstanceNorm
// statistics of a layer are known as soon as its accumulators are (no partial records, no second kernel), the normalised
// activations go from the accumulator registers straight into LDS as the next layer's operand, and only the skip tensor
// (53 KB, written once and read back by the same workgroup) and the packed weights (streamed from L2 in chunks of 8 input
// channels, prefetched into registers under the MFMA sweep of the previous chunk) touch memory.
// Arithmetic is the per-layer kernels' (conv_kernels.hip): v_mfma_f32_16x16x4_f32, the same chunk / tap / k-step order, the
// same scale-shift form of InstanceNorm + LeakyReLU, two-pass statistics.
#include <cstdlib>
#include <mutex>
#include "common.h"

namespace cine {

typedef float f32x4 __attribute__((ext_vector_type(4)));

namespace ub {
constexpr int C1 = 32, C2 = 64, C3 = 128;          // channels of the level above / this level / the bottleneck
constexpr int H2 = 52, W2 = 4, H3 = 26, W3 = 2;    // plane sizes
constexpr int NT = 512;                            // 8 waves: two per SIMD
// level-2 geometry: fragment = 16 pixels = 4 rows x 4 columns
constexpr int RPF2 = 4, NF2 = 13, COLS2 = 6, ROWS2 = H2 + 2;
constexpr int PS2 = ((ROWS2 * COLS2 + 31) / 32) * 32 + 16;            // channel stride == 16 (mod 32)
// level-3 geometry: fragment = 8 rows x 2 columns; 4 fragments cover 32 rows, rows 26..31 are padding (read as zeros)
constexpr int RPF3 = 8, NF3 = 4, COLS3 = 4, ROWS3 = NF3 * RPF3 + 2;
constexpr int PS3 = ((ROWS3 * COLS3 + 31) / 32) * 32 + 16;
constexpr int CK = 8, CKT = 16;                    // input channels per chunk: 3x3 convs / transpose conv (the packers' values)
constexpr int COTP2 = C2 + 16, COTP3 = C3 + 16, COTPT = 4 * C2 + 16;  // LDS row stride of a weight slab (== 16 mod 64)
constexpr int BUF_FLOATS = (C2 * PS2 > C3 * PS3) ? C2 * PS2 : C3 * PS3;
constexpr int WL_FLOATS = 9 * CK * COTP3;          // the largest slab (>= 9*CK*COTP2, >= CKT*COTPT)
constexpr int STG_FLOATS = CK * PS2;
constexpr int RED_FLOATS = 2 * C2 * 3;
constexpr int ST_FLOATS = 2 * C1;
constexpr int LDS_FLOATS = BUF_FLOATS + WL_FLOATS + STG_FLOATS + RED_FLOATS + ST_FLOATS;
constexpr int NWT = (9 * CK * (C3 / 4) + NT - 1) / NT;                // weight float4 per thread and chunk (largest slab)
static_assert(CKT * COTPT <= WL_FLOATS && 9 * CK * COTP2 <= WL_FLOATS, "weight slab");
}  // namespace ub

struct BottomArgs {
    const float* x1; const float* px1; int np1;     // level P-2 ConvBlock output: raw (n, 32, 104, 8) + partial stats (n, 32, np1, 3)
    const float* w[7][2];                           // packed weights [layer][weight set]: L2a L2b L3a L3b T L2c L2d
    int set_split;                                  // samples >= set_split use weight set 1
    float* skip2;                                   // (n, 64, 52, 4) scratch: the activated level-2 skip
    float* y; float* py;                            // level-2 up-block output: raw (n, 64, 52, 4) + one stats record per plane
    float eps, slope;
#ifdef CINE_UB_DEBUG
    float* dbg; int dbg_stop;                       // diagnostic build: dump the LDS activation buffer after layer dbg_stop and return
#endif
};

#ifdef CINE_UB_DEBUG
#define UB_DUMP(k)                                                                                          \
    if (a.dbg_stop == (k)) {                                                                                \
        __syncthreads();                                                                                    \
        for (int e = threadIdx.x; e < ub::BUF_FLOATS; e += ub::NT) a.dbg[(long)blockIdx.x * ub::BUF_FLOATS + e] = buf[e]; \
        return;                                                                                             \
    }
#else
#define UB_DUMP(k)
#endif

__device__ __forceinline__ float2 ub_merge(const float* p, int np, float eps) {
    float cnt = 0.f, mean = 0.f;
    for (int i = 0; i < np; ++i) { cnt += p[3 * i]; mean += p[3 * i] * p[3 * i + 1]; }
    mean /= cnt;
    float m2 = 0.f;
    for (int i = 0; i < np; ++i) { const float d = p[3 * i + 1] - mean; m2 += p[3 * i + 2] + p[3 * i] * d * d; }
    return make_float2(mean, 1.0f / sqrtf(m2 / cnt + eps));
}
__device__ __forceinline__ float ub_act(float x, float scale, float shift, float slope) {
    const float v = fmaf(x, scale, shift);
    return fmaxf(v, v * slope);
}

// ---- weight slab of one chunk: global [TAPS*CKC][rowsp] -> registers -> LDS [TAPS*CKC][COTP]
template <int ROWS_G, int ROWSP, int COTP>
__device__ __forceinline__ void slab_issue(float4 (&wraw)[ub::NWT], const float* wsrc) {
    constexpr int N4 = ROWS_G * (ROWSP / 4);
#pragma unroll
    for (int i = 0; i < ub::NWT; ++i) {
        const int e = threadIdx.x + i * ub::NT;
        if (i * ub::NT < N4) wraw[i] = *reinterpret_cast<const float4*>(wsrc + 4 * (e < N4 ? e : 0));
    }
}
template <int ROWS_G, int ROWSP, int COTP>
__device__ __forceinline__ void slab_commit(const float4 (&wraw)[ub::NWT], float* w_lds) {
    constexpr int N4 = ROWS_G * (ROWSP / 4);
#pragma unroll
    for (int i = 0; i < ub::NWT; ++i) {
        const int e = threadIdx.x + i * ub::NT;
        if (i * ub::NT < N4 && e < N4) {
            const int row = e / (ROWSP / 4), c4 = (e % (ROWSP / 4)) * 4;
            *reinterpret_cast<float4*>(w_lds + row * COTP + c4) = wraw[i];
        }
    }
}

// ---- MFMA sweep of one chunk.  A operand (pixels x k): in[(4 ks + kk) * PS + (frow[f] + dy) * COLS + cmap[dx]];
// B operand (k x rows): w[(tap * CKC + 4 ks + kk) * COTP + wcol[ct]].  One operand group ahead, as in conv_kernels.hip.
template <int TAPS, int CKC, int PS, int COLS, int COTP, int CT, int MT>
__device__ __forceinline__ void sweep(f32x4 (&acc)[CT][MT], const float* in, const float* w_lds, const int kk,
                                      const int (&frow)[MT], const int (&cmap)[3], const int (&wcol)[CT]) {
    constexpr int KS = CKC / 4, NG = TAPS * KS;
    float af[2][CT], bf[2][MT];
    auto load_group = [&](int g, float (&wa)[CT], float (&xa)[MT]) {
        const int tap = g / KS, ks = g % KS;
        const int dy = TAPS == 1 ? 0 : tap / 3, dx = TAPS == 1 ? 1 : tap % 3;
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) wa[ct] = w_lds[(tap * CKC + 4 * ks + kk) * COTP + wcol[ct]];
#pragma unroll
        for (int f = 0; f < MT; ++f) xa[f] = in[(4 * ks + kk) * PS + (frow[f] + dy) * COLS + cmap[dx]];
    };
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
}

__device__ __forceinline__ void zero_lds(float* p, int nfloats) {
    for (int e = threadIdx.x * 4; e < nfloats; e += ub::NT * 4) *reinterpret_cast<float4*>(p + e) = make_float4(0.f, 0.f, 0.f, 0.f);
}

__global__ __launch_bounds__(ub::NT, 2) void unet_bottom_kernel(BottomArgs a) {
    using namespace ub;
    extern __shared__ __align__(16) float smem_ub[];
    float* buf = smem_ub;
    float* wl = buf + BUF_FLOATS;
    float* stg = wl + WL_FLOATS;
    float* red = stg + STG_FLOATS;
    float* st1 = red + RED_FLOATS;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int q = lane & 15, kk = lane >> 4;
    const int n = blockIdx.x;
    const bool set = n >= a.set_split;
    const float slope = a.slope;
    float4 wraw[NWT];
    // weight pointers of this sample's set (selected with compile-time indices: no private copy of the argument block)
    const float* const wL2a = set ? a.w[0][1] : a.w[0][0];
    const float* const wL2b = set ? a.w[1][1] : a.w[1][0];
    const float* const wL3a = set ? a.w[2][1] : a.w[2][0];
    const float* const wL3b = set ? a.w[3][1] : a.w[3][0];
    const float* const wT = set ? a.w[4][1] : a.w[4][0];
    const float* const wL2c = set ? a.w[5][1] : a.w[5][0];
    const float* const wL2d = set ? a.w[6][1] : a.w[6][0];

    // ---- lane geometry.  Level 2 (conv 3x3 on 52 x 4): waves = 4 row blocks (16 output channels) x 2 fragment halves;
    // level 3 (26 x 2): 8 row blocks, 4 fragments of 8 rows x 2 columns.  The operand addresses derived from these are
    // rebuilt per layer from lane ids passed through an empty asm: hoisted to the top of the kernel as loop invariants
    // they were > 150 registers that the allocator spilled and re-read inside the sweeps.
    const int wm2 = wave & 3, wn2 = wave >> 2;
    const int nf2 = wn2 == 0 ? 7 : NF2 - 7;           // live fragments of this wave
#define UB_GEO2()                                                                                              \
    int q_ = q, kk_ = kk;                                                                                      \
    asm volatile("" : "+v"(q_), "+v"(kk_));                                                                    \
    const int qr2 = q_ / W2, qc2 = q_ % W2;                                                                    \
    int frow2[7], cmap2[3], wcol2[1];                                                                          \
    _Pragma("unroll") for (int f = 0; f < 7; ++f) frow2[f] = (wn2 * 7 + f) * RPF2 + qr2;                       \
    _Pragma("unroll") for (int dx = 0; dx < 3; ++dx) cmap2[dx] = (qc2 + dx - 1 + COLS2) % COLS2;               \
    wcol2[0] = 16 * wm2 + q_;
#define UB_GEO3()                                                                                              \
    int q_ = q, kk_ = kk;                                                                                      \
    asm volatile("" : "+v"(q_), "+v"(kk_));                                                                    \
    const int qr3 = q_ / W3, qc3 = q_ % W3;                                                                    \
    int frow3[4], cmap3[3];                                                                                    \
    _Pragma("unroll") for (int f = 0; f < 4; ++f) frow3[f] = f * RPF3 + qr3;                                   \
    _Pragma("unroll") for (int dx = 0; dx < 3; ++dx) cmap3[dx] = (qc3 + dx - 1 + COLS3) % COLS3;
    // frow: LDS row of tap dy = 0 (= image row); the 14th level-2 fragment (second half's f = 6) reads the channel's
    // padding and is discarded

    // ---- prologue: weights of the first chunk in flight; merged stats of the 32 input channels; zeroed LDS
    slab_issue<9 * CK, C2, COTP2>(wraw, wL2a);
    if (tid < C1) {
        const float2 mr = ub_merge(a.px1 + ((long)n * C1 + tid) * a.np1 * 3, a.np1, a.eps);
        st1[2 * tid] = mr.y; st1[2 * tid + 1] = -mr.x * mr.y;
    }
    zero_lds(buf, BUF_FLOATS);
    zero_lds(stg, STG_FLOATS);

    // staging unit of a streamed level-2 source: (channel of the chunk, image row) -> one row of 4 pixels
    const int su_ck = tid / H2, su_y = tid - su_ck * H2;
    const bool su_on = tid < CK * H2;

    f32x4 acc2[1][7];
    auto zero_acc2 = [&]() {
#pragma unroll
        for (int f = 0; f < 7; ++f) acc2[0][f] = (f32x4){0.f, 0.f, 0.f, 0.f};
    };
    // InstanceNorm statistics of a level-2 layer (64 channels, two waves per channel): two-pass per wave, Chan merge
    // of the two wave records through LDS.  Returns {scale, shift} of act() and the merged {mean, M2}.
    auto stats2 = [&](float& scale, float& shift, float& mean_o, float& m2_o) {
        float s = 0.f;
#pragma unroll
        for (int f = 0; f < 7; ++f)
#pragma unroll
            for (int j = 0; j < 4; ++j) s += f < nf2 ? acc2[0][f][j] : 0.f;
        s += __shfl_xor(s, 16, 64); s += __shfl_xor(s, 32, 64);
        const float cnt = (float)(nf2 * 16), mean = s / cnt;
        float qv = 0.f;
#pragma unroll
        for (int f = 0; f < 7; ++f)
#pragma unroll
            for (int j = 0; j < 4; ++j) { const float d = acc2[0][f][j] - mean; qv += f < nf2 ? d * d : 0.f; }
        qv += __shfl_xor(qv, 16, 64); qv += __shfl_xor(qv, 32, 64);
        if (kk == 0) { float* o = red + (wn2 * C2 + 16 * wm2 + q) * 3; o[0] = cnt; o[1] = mean; o[2] = qv; }
        __syncthreads();                               // also: every wave is done reading buf / stg / wl
        const float* r0 = red + (16 * wm2 + q) * 3; const float* r1 = r0 + C2 * 3;
        const float c = r0[0] + r1[0];
        const float m = (r0[0] * r0[1] + r1[0] * r1[1]) / c;
        const float d0 = r0[1] - m, d1 = r1[1] - m;
        const float m2 = (r0[2] + r0[0] * d0 * d0) + (r1[2] + r1[0] * d1 * d1);
        const float rstd = 1.0f / sqrtf(m2 / c + a.eps);
        scale = rstd; shift = -m * rstd; mean_o = m; m2_o = m2;
    };
    // normalised + activated accumulators -> buf in level-2 layout (this lane: channel 16 wm2 + q, rows 4 f + kk)
    auto store_act2 = [&](float scale, float shift) {
        float* cb = buf + (16 * wm2 + q) * PS2;
#pragma unroll
        for (int f = 0; f < 7; ++f) {
            if (f >= nf2) break;
            float* d = cb + ((wn2 * 7 + f) * RPF2 + kk + 1) * COLS2;
            *reinterpret_cast<float2*>(d) = make_float2(ub_act(acc2[0][f][0], scale, shift, slope), ub_act(acc2[0][f][1], scale, shift, slope));
            *reinterpret_cast<float2*>(d + 2) = make_float2(ub_act(acc2[0][f][2], scale, shift, slope), ub_act(acc2[0][f][3], scale, shift, slope));
        }
    };

    // ================================================================ L2a: pool(act(x1)) 32 -> 64
    {
        UB_GEO2();
        zero_acc2();
        const float* xb = a.x1 + (long)n * C1 * (2 * H2) * (2 * W2);
        float4 xr[4];
        auto issue_x = [&](int chunk) {
            if (!su_on) return;
            const float* p = xb + ((long)(chunk * CK + su_ck) * (2 * H2) + 2 * su_y) * (2 * W2);
            xr[0] = *reinterpret_cast<const float4*>(p); xr[1] = *reinterpret_cast<const float4*>(p + 4);
            xr[2] = *reinterpret_cast<const float4*>(p + 8); xr[3] = *reinterpret_cast<const float4*>(p + 12);
        };
        auto commit_x = [&](int chunk) {
            if (!su_on) return;
            const float sc = st1[2 * (chunk * CK + su_ck)], sh = st1[2 * (chunk * CK + su_ck) + 1];
            const float* r0 = reinterpret_cast<const float*>(&xr[0]);   // row 2y: 8 floats, row 2y+1: 8 floats
            float o[4];
#pragma unroll
            for (int u = 0; u < 4; ++u)                                   // unet.py:97 avg_pool2d(2, 2) of the activated tensor
                o[u] = 0.25f * (ub_act(r0[2 * u], sc, sh, slope) + ub_act(r0[2 * u + 1], sc, sh, slope) +
                                ub_act(r0[8 + 2 * u], sc, sh, slope) + ub_act(r0[8 + 2 * u + 1], sc, sh, slope));
            float* d = stg + su_ck * PS2 + (su_y + 1) * COLS2;
            *reinterpret_cast<float2*>(d) = make_float2(o[0], o[1]);
            *reinterpret_cast<float2*>(d + 2) = make_float2(o[2], o[3]);
        };
        issue_x(0);
        constexpr int NCH = C1 / CK;
        for (int chunk = 0; chunk < NCH; ++chunk) {
            __syncthreads();
            slab_commit<9 * CK, C2, COTP2>(wraw, wl);
            commit_x(chunk);
            __syncthreads();
            if (chunk + 1 < NCH) { slab_issue<9 * CK, C2, COTP2>(wraw, wL2a + (long)(chunk + 1) * 9 * CK * C2); issue_x(chunk + 1); }
            else slab_issue<9 * CK, C2, COTP2>(wraw, wL2b);
            __builtin_amdgcn_sched_barrier(0);
            sweep<9, CK, PS2, COLS2, COTP2, 1, 7>(acc2, stg, wl, kk_, frow2, cmap2, wcol2);
        }
        float sc, sh, m_, m2_;
        stats2(sc, sh, m_, m2_);
        (void)m_; (void)m2_;
        store_act2(sc, sh);
        UB_DUMP(0)
    }
    // ================================================================ L2b: 64 -> 64; output = skip (global, activated) + pooled -> level-3 input
    {
        UB_GEO2();
        zero_acc2();
        constexpr int NCH = C2 / CK;
        for (int chunk = 0; chunk < NCH; ++chunk) {
            __syncthreads();
            slab_commit<9 * CK, C2, COTP2>(wraw, wl);
            __syncthreads();
            if (chunk + 1 < NCH) slab_issue<9 * CK, C2, COTP2>(wraw, wL2b + (long)(chunk + 1) * 9 * CK * C2);
            else slab_issue<9 * CK, C3, COTP3>(wraw, wL3a);
            __builtin_amdgcn_sched_barrier(0);
            sweep<9, CK, PS2, COLS2, COTP2, 1, 7>(acc2, buf + chunk * CK * PS2, wl, kk_, frow2, cmap2, wcol2);
        }
        float sc, sh, m_, m2_;
        stats2(sc, sh, m_, m2_);                         // barrier inside: buf is dead now
        (void)m_; (void)m2_;
        zero_lds(buf, BUF_FLOATS);                       // level-3 layout next: its halo / padding rows must read zero
        __syncthreads();
        const int c = 16 * wm2 + q;
        float* sk = a.skip2 + ((long)n * C2 + c) * (H2 * W2);
        float* cb3 = buf + c * PS3;
#pragma unroll
        for (int f = 0; f < 7; ++f) {
            if (f >= nf2) break;
            const int row = (wn2 * 7 + f) * RPF2 + kk;
            float v[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] = ub_act(acc2[0][f][j], sc, sh, slope);
            *reinterpret_cast<float4*>(sk + row * W2) = make_float4(v[0], v[1], v[2], v[3]);
            // 2 x 2 average with the lane holding the next row (kk ^ 1): ((a00 + a01) + a10) + a11, the per-layer kernel's order
            float o[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) o[j] = __shfl_xor(v[j], 16, 64);
            if ((kk & 1) == 0) {
                const float p0 = 0.25f * (((v[0] + v[1]) + o[0]) + o[1]), p1 = 0.25f * (((v[2] + v[3]) + o[2]) + o[3]);
                *reinterpret_cast<float2*>(cb3 + (row / 2 + 1) * COLS3) = make_float2(p0, p1);
            }
        }
        UB_DUMP(1)
    }
    // ================================================================ level 3: 64 -> 128, 128 -> 128 (8 row blocks x 4 fragments)
    f32x4 acc3[1][4];
    auto stats3 = [&](float& scale, float& shift) {       // one wave holds all 52 pixels of its 16 channels
        float s = 0.f;
#pragma unroll
        for (int f = 0; f < 4; ++f)
#pragma unroll
            for (int j = 0; j < 4; ++j) s += (f * RPF3 + (4 * kk + j) / W3 < H3) ? acc3[0][f][j] : 0.f;
        s += __shfl_xor(s, 16, 64); s += __shfl_xor(s, 32, 64);
        const float cnt = (float)(H3 * W3), mean = s / cnt;
        float qv = 0.f;
#pragma unroll
        for (int f = 0; f < 4; ++f)
#pragma unroll
            for (int j = 0; j < 4; ++j) { const float d = acc3[0][f][j] - mean; qv += (f * RPF3 + (4 * kk + j) / W3 < H3) ? d * d : 0.f; }
        qv += __shfl_xor(qv, 16, 64); qv += __shfl_xor(qv, 32, 64);
        const float rstd = 1.0f / sqrtf(qv / cnt + a.eps);
        scale = rstd; shift = -mean * rstd;
    };
    auto store_act3 = [&](float scale, float shift) {
        float* cb = buf + (16 * wave + q) * PS3;
#pragma unroll
        for (int f = 0; f < 4; ++f)
#pragma unroll
            for (int hh = 0; hh < 2; ++hh) {                          // pixels 4 kk + 2 hh, + 1: one image row of 2 columns
                const int y = f * RPF3 + 2 * kk + hh;
                if (y < H3)
                    *reinterpret_cast<float2*>(cb + (y + 1) * COLS3) =
                        make_float2(ub_act(acc3[0][f][2 * hh], scale, shift, slope), ub_act(acc3[0][f][2 * hh + 1], scale, shift, slope));
            }
    };
#pragma unroll 1
    for (int layer = 0; layer < 2; ++layer) {
        UB_GEO3();
        int wcol3[1]; wcol3[0] = 16 * wave + q_;
        const int cin = layer == 0 ? C2 : C3;
#pragma unroll
        for (int f = 0; f < 4; ++f) acc3[0][f] = (f32x4){0.f, 0.f, 0.f, 0.f};
        const float* wp = layer == 0 ? wL3a : wL3b;
        const int nch = cin / CK;
        for (int chunk = 0; chunk < nch; ++chunk) {
            __syncthreads();
            slab_commit<9 * CK, C3, COTP3>(wraw, wl);
            __syncthreads();
            if (chunk + 1 < nch) slab_issue<9 * CK, C3, COTP3>(wraw, wp + (long)(chunk + 1) * 9 * CK * C3);
            else if (layer == 0) slab_issue<9 * CK, C3, COTP3>(wraw, wL3b);
            else slab_issue<CKT, 4 * C2, COTPT>(wraw, wT);
            __builtin_amdgcn_sched_barrier(0);
            sweep<9, CK, PS3, COLS3, COTP3, 1, 4>(acc3, buf + chunk * CK * PS3, wl, kk_, frow3, cmap3, wcol3);
        }
        float sc, sh;
        stats3(sc, sh);
        __syncthreads();                                  // every wave is done reading buf
        store_act3(sc, sh);                               // same layout, interior only: halo and padding rows stay zero
        UB_DUMP(2 + layer)
    }
    // ================================================================ transpose conv 128 -> 64 (k2 s2) as a 1x1 GEMM with 256 rows
    {
        f32x4 acct[2][4];
#pragma unroll
        for (int ct = 0; ct < 2; ++ct)
#pragma unroll
            for (int f = 0; f < 4; ++f) acct[ct][f] = (f32x4){0.f, 0.f, 0.f, 0.f};
        UB_GEO3();
        int wcolt[2]; wcolt[0] = 16 * wave + q_; wcolt[1] = 16 * (wave + 8) + q_;   // row blocks {w, w + 8}: both output-row parities of a channel
        int frowt[4];
#pragma unroll
        for (int f = 0; f < 4; ++f) frowt[f] = frow3[f] + 1;                        // 1x1: the pixel itself (LDS row = image row + 1)
        constexpr int NCH = C3 / CKT;
        for (int chunk = 0; chunk < NCH; ++chunk) {
            __syncthreads();
            slab_commit<CKT, 4 * C2, COTPT>(wraw, wl);
            __syncthreads();
            if (chunk + 1 < NCH) slab_issue<CKT, 4 * C2, COTPT>(wraw, wT + (long)(chunk + 1) * CKT * 4 * C2);
            else slab_issue<9 * CK, C2, COTP2>(wraw, wL2c);
            __builtin_amdgcn_sched_barrier(0);
            sweep<1, CKT, PS3, COLS3, COTPT, 2, 4>(acct, buf + chunk * CKT * PS3, wl, kk_, frowt, cmap3, wcolt);
        }
        // row m = 16 blk + q = 2 (a 64 + co) + b: co = 8 wave + q / 2, b = q & 1 (column parity), a = ct (row parity).
        // Statistics of channel co over its 4 sub-positions x 52 input pixels = 208 output pixels: lanes q, q ^ 1, all kk.
        float s = 0.f;
#pragma unroll
        for (int ct = 0; ct < 2; ++ct)
#pragma unroll
            for (int f = 0; f < 4; ++f)
#pragma unroll
                for (int j = 0; j < 4; ++j) s += (f * RPF3 + (4 * kk + j) / W3 < H3) ? acct[ct][f][j] : 0.f;
        s += __shfl_xor(s, 1, 64); s += __shfl_xor(s, 16, 64); s += __shfl_xor(s, 32, 64);
        const float cnt = (float)(H2 * W2), mean = s / cnt;
        float qv = 0.f;
#pragma unroll
        for (int ct = 0; ct < 2; ++ct)
#pragma unroll
            for (int f = 0; f < 4; ++f)
#pragma unroll
                for (int j = 0; j < 4; ++j) { const float d = acct[ct][f][j] - mean; qv += (f * RPF3 + (4 * kk + j) / W3 < H3) ? d * d : 0.f; }
        qv += __shfl_xor(qv, 1, 64); qv += __shfl_xor(qv, 16, 64); qv += __shfl_xor(qv, 32, 64);
        const float rstd = 1.0f / sqrtf(qv / cnt + a.eps), sc = rstd, sh = -mean * rstd;
        __syncthreads();                                  // buf (level-3 layout) is dead
        zero_lds(buf, BUF_FLOATS);
        __syncthreads();
        const int co = 8 * wave + (q >> 1), b = q & 1;
        float* cb = buf + co * PS2;
#pragma unroll
        for (int ct = 0; ct < 2; ++ct)
#pragma unroll
            for (int f = 0; f < 4; ++f)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int p = 4 * kk + j, y3 = f * RPF3 + p / W3, x3 = p % W3;
                    if (y3 < H3) cb[(2 * y3 + ct + 1) * COLS2 + 2 * x3 + b] = ub_act(acct[ct][f][j], sc, sh, slope);
                }
        UB_DUMP(4)
    }
    // ================================================================ L2c: conv on cat([up (buf), skip (global)]) 128 -> 64
    {
        UB_GEO2();
        zero_acc2();
        const float* sk = a.skip2 + (long)n * C2 * (H2 * W2);
        float4 sr;
        auto issue_s = [&](int chunk) {                   // chunk counts within the skip half
            if (su_on) sr = *reinterpret_cast<const float4*>(sk + ((long)(chunk * CK + su_ck) * H2 + su_y) * W2);
        };
        auto commit_s = [&]() {
            if (!su_on) return;
            float* d = stg + su_ck * PS2 + (su_y + 1) * COLS2;
            *reinterpret_cast<float2*>(d) = make_float2(sr.x, sr.y);
            *reinterpret_cast<float2*>(d + 2) = make_float2(sr.z, sr.w);
        };
        constexpr int NCH = 2 * C2 / CK, NUP = C2 / CK;
        for (int chunk = 0; chunk < NCH; ++chunk) {
            __syncthreads();
            slab_commit<9 * CK, C2, COTP2>(wraw, wl);
            if (chunk >= NUP) commit_s();
            __syncthreads();
            if (chunk + 1 < NCH) {
                slab_issue<9 * CK, C2, COTP2>(wraw, wL2c + (long)(chunk + 1) * 9 * CK * C2);
                if (chunk + 1 >= NUP) issue_s(chunk + 1 - NUP);
            } else {
                slab_issue<9 * CK, C2, COTP2>(wraw, wL2d);
            }
            __builtin_amdgcn_sched_barrier(0);
            sweep<9, CK, PS2, COLS2, COTP2, 1, 7>(acc2, chunk < NUP ? buf + chunk * CK * PS2 : stg, wl, kk_, frow2, cmap2, wcol2);
        }
        float sc, sh, m_, m2_;
        stats2(sc, sh, m_, m2_);
        (void)m_; (void)m2_;
        store_act2(sc, sh);
        UB_DUMP(5)
    }
    // ================================================================ L2d: 64 -> 64, raw output + one statistics record per plane
    {
        UB_GEO2();
        zero_acc2();
        constexpr int NCH = C2 / CK;
        for (int chunk = 0; chunk < NCH; ++chunk) {
            __syncthreads();
            slab_commit<9 * CK, C2, COTP2>(wraw, wl);
            __syncthreads();
            if (chunk + 1 < NCH) slab_issue<9 * CK, C2, COTP2>(wraw, wL2d + (long)(chunk + 1) * 9 * CK * C2);
            __builtin_amdgcn_sched_barrier(0);
            sweep<9, CK, PS2, COLS2, COTP2, 1, 7>(acc2, buf + chunk * CK * PS2, wl, kk_, frow2, cmap2, wcol2);
        }
        const int c = 16 * wm2 + q;
        float* yb = a.y + ((long)n * C2 + c) * (H2 * W2);
#pragma unroll
        for (int f = 0; f < 7; ++f) {
            if (f >= nf2) break;
            const int row = (wn2 * 7 + f) * RPF2 + kk;
            *reinterpret_cast<float4*>(yb + row * W2) = make_float4(acc2[0][f][0], acc2[0][f][1], acc2[0][f][2], acc2[0][f][3]);
        }
        float sc, sh, mean, m2;
        stats2(sc, sh, mean, m2);
        if (wn2 == 0 && kk == 0) { float* o = a.py + ((long)n * C2 + c) * 3; o[0] = (float)(H2 * W2); o[1] = mean; o[2] = m2; }
    }
}

// Opt-in (CINE_UNET_BOTTOM=1, read at every U-Net pass).  Measured on cfg 2: results match the per-layer launches
// to rounding (1.4e-6), but one 149 KB workgroup per CU means 400 planes take two rounds on 256 CUs and nothing else can share
// the CU: 615 us per U-Net pass against 539 us for the seven launches it replaces, 129 vs 143 slices/s with three slices in flight.
bool unet_bottom_enabled() {
    const char* e = getenv("CINE_UNET_BOTTOM");
    return e != nullptr && atoi(e) != 0;
}

// The fused kernel handles exactly this shape (cfg 2's x-f / y-f planes); anything else takes the per-layer launches.
bool unet_bottom_applies(int chans2, int h2, int w2) {
    return unet_bottom_enabled() && chans2 == ub::C2 && h2 == ub::H2 && w2 == ub::W2;
}

int launch_unet_bottom(const BottomArgs& a, int n, hipStream_t st) {
    constexpr size_t lds = (size_t)ub::LDS_FLOATS * sizeof(float);
    static_assert(lds <= 160 * 1024, "unet_bottom_kernel LDS");
    int dev = 0;
    (void)hipGetDevice(&dev);
    CINE_REQUIRE(dev >= 0 && dev < 64, CINE_EUNSUPPORTED, "unet_bottom_kernel: device index %d", dev);
    static std::once_flag once[64];
    static hipError_t status[64];       // kept per device: a failed first call fails every later launch with its own message
    std::call_once(once[dev], [&] {
        status[dev] = hipFuncSetAttribute(reinterpret_cast<const void*>(unet_bottom_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    });
    CINE_REQUIRE(status[dev] == hipSuccess, CINE_EHIP, "unet_bottom_kernel: hipFuncSetAttribute(MaxDynamicSharedMemorySize): %s", hipGetErrorString(status[dev]));
    ProfScope prof(F_CONV3, st);
    hipLaunchKernelGGL(unet_bottom_kernel, dim3(n), dim3(ub::NT), lds, st, a);
    return check_launch("unet_bottom_kernel");
}

}  // namespace cine
