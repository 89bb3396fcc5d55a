// conv_src.h -- device helpers shared by the convolution kernels (conv_kernels.hip) and their gradients (grad_kernels.hip):
// InstanceNorm statistics records, the on-load activation, and the description / scalar fetch of a convolution source.
#pragma once
#include "common.h"

namespace cine {

typedef float f32x4 __attribute__((ext_vector_type(4)));
struct __attribute__((packed, aligned(4))) f4u { float v[4]; };      // 4 floats at 4-byte alignment (rows of widths like 50 or 25)

// ---------------------------------------------------------------- statistics helpers
// partial record = {count, mean, M2}; merged InstanceNorm stats = {mean, 1/sqrt(M2/count + eps)}
// the usual case (a handful of records per plane) in two halves, so that a kernel can put other loads between them: one round
// of loads into registers, then the arithmetic (same summation order as the loop below: bit-identical)
template <int NPMAX>
__device__ __forceinline__ void load_partials(const float* p, int np, float (&r)[3 * NPMAX]) {
#pragma unroll
    for (int i = 0; i < 3 * NPMAX; ++i) r[i] = p[min(i, 3 * np - 1)];
}
template <int NPMAX>
__device__ __forceinline__ float2 merge_loaded(const float (&r)[3 * NPMAX], int np, float eps) {
    float cnt = 0.f, mean = 0.f;
#pragma unroll
    for (int i = 0; i < NPMAX; ++i) if (i < np) { cnt += r[3 * i]; mean += r[3 * i] * r[3 * i + 1]; }
    mean /= cnt;
    float m2 = 0.f;
#pragma unroll
    for (int i = 0; i < NPMAX; ++i) if (i < np) { const float d = r[3 * i + 1] - mean; m2 += r[3 * i + 2] + r[3 * i] * d * d; }
    return make_float2(mean, 1.0f / sqrtf(m2 / cnt + eps));
}
__device__ __forceinline__ float2 merge_partials(const float* p, int np, float eps) {
    if (np <= 8) {
        float r[24];
        load_partials<8>(p, np, r);
        return merge_loaded<8>(r, np, eps);
    }
    float cnt = 0.f, mean = 0.f;
    for (int i = 0; i < np; ++i) { cnt += p[3 * i]; mean += p[3 * i] * p[3 * i + 1]; }
    mean /= cnt;
    float m2 = 0.f;
    for (int i = 0; i < np; ++i) {
        const float d = p[3 * i + 1] - mean;
        m2 += p[3 * i + 2] + p[3 * i] * d * d;
    }
    return make_float2(mean, 1.0f / sqrtf(m2 / cnt + eps));
}
// InstanceNorm + LeakyReLU of one raw value: scale = rstd, shift = -mean * rstd (the x * alpha + beta form of
// ATen's batch-norm transform); 0 <= slope <= 1 so that leaky_relu(v) == max(v, v * slope)
__device__ __forceinline__ float act(float x, float scale, float shift, float slope) {
    const float v = fmaf(x, scale, shift);
    return fmaxf(v, v * slope);
}

// act() of PW (2 or 4) consecutive values with the packed fp32 instructions (v_pk_fma_f32, v_pk_mul_f32: two lanes of IEEE
// arithmetic per instruction, the same results as the scalar form) -- fewer vector instructions in the staging phase, which
// shares the SIMD's issue with the other workgroups' MFMA sweeps
typedef float f32x2 __attribute__((ext_vector_type(2)));
template <int PW>
__device__ __forceinline__ void act_piece(float* ov, float scale, float shift, float slope) {
#pragma unroll
    for (int u = 0; u < PW; u += 2) {
        const f32x2 x = {ov[u], ov[u + 1]};
        const f32x2 v = __builtin_elementwise_fma(x, (f32x2){scale, scale}, (f32x2){shift, shift});
        const f32x2 w = v * (f32x2){slope, slope};
        ov[u] = fmaxf(v.x, w.x); ov[u + 1] = fmaxf(v.y, w.y);
    }
}

struct Src {
    const float* x; const float* part;   // raw activations (n, c, h, w); partial stats (n, c, np, 3)
    int c, mode, h, w, np;               // mode 0 as-is, 1 norm+LReLU, 2 norm+LReLU+avgpool2,
                                         // 3 Haar DWT of act(x): 4c channels at (h/2, w/2)   (mwcnn.py:224-236)
                                         // 4 Haar IWT of act(x): c/4 channels at (2h, 2w)    (mwcnn.py:252-261)
                                         // 5 space-to-depth of x (as is): channel 4 c + 2 a + b at (y, x) = x[c][2y + a][2x + b]
                                         //   (gradient of the k2 s2 transpose conv, unet.py:212-215: its GEMM K dimension)
    int act;                             // modes 3/4: 1 = x is raw (normalise + LReLU first), 0 = use as is
    int d;                               // depth of the source volume (1 for 2-D planes)
};
// input channels a source contributes after its on-load transform
__host__ __device__ inline int src_cin(const Src& s) { return (s.mode == 3 || s.mode == 5) ? 4 * s.c : (s.mode == 4 ? s.c / 4 : s.c); }

// scalar (any shape) fetch of one transformed input value; st = {scale, shift} table (see act()) of THIS source's channels
__device__ __forceinline__ float fetch_scalar(const Src& s, int n, int cl, int gz, int gy, int gx, const float* st, float slope) {
    if (s.mode == 3) {                               // DWT: band = cl / c, source channel = cl % c
        const int band = cl / s.c, c = cl - band * s.c;
        if (2 * gy + 1 >= s.h || 2 * gx + 1 >= s.w) return 0.f;
        const float* p = s.x + (((long)n * s.c + c) * s.h + 2 * gy) * s.w + 2 * gx;
        float x1 = p[0], x3 = p[1], x2 = p[s.w], x4 = p[s.w + 1];          // x1 even/even, x2 odd row, x3 odd col
        if (s.act & 1) {
            const float m = st[2 * c], r = st[2 * c + 1];
            x1 = act(x1, m, r, slope); x2 = act(x2, m, r, slope); x3 = act(x3, m, r, slope); x4 = act(x4, m, r, slope);
        }
        x1 *= 0.5f; x2 *= 0.5f; x3 *= 0.5f; x4 *= 0.5f;
        switch (band) {
            case 0: return x1 + x2 + x3 + x4;        // LL
            case 1: return -x1 - x2 + x3 + x4;       // HL
            case 2: return -x1 + x2 - x3 + x4;       // LH
            default: return x1 - x2 - x3 + x4;       // HH
        }
    }
    if (s.mode == 5) {                               // space-to-depth: channel cl = 4 c + 2 a + b
        const int c = cl >> 2, yy = 2 * gy + ((cl >> 1) & 1), xx = 2 * gx + (cl & 1);
        if (yy >= s.h || xx >= s.w) return 0.f;
        return s.x[(((long)n * s.c + c) * s.h + yy) * s.w + xx];
    }
    if (s.mode == 4) {                               // IWT: channel cl from source channels cl + k * c/4
        const int cq = s.c / 4, sy = gy >> 1, sx = gx >> 1;
        if (sy >= s.h || sx >= s.w) return 0.f;
        float v[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int c = cl + k * cq;
            float t = s.x[(((long)n * s.c + c) * s.h + sy) * s.w + sx];
            if (s.act & 1) t = act(t, st[2 * c], st[2 * c + 1], slope);
            v[k] = 0.5f * t;
        }
        const int ry = gy & 1, rx = gx & 1;
        if (!ry && !rx) return v[0] - v[1] - v[2] + v[3];
        if (ry && !rx) return v[0] - v[1] + v[2] - v[3];
        if (!ry && rx) return v[0] + v[1] - v[2] - v[3];
        return v[0] + v[1] + v[2] + v[3];
    }
    const long plane = (long)n * s.c + cl;
    const float mean = st[2 * cl], rstd = st[2 * cl + 1];   // positional: {scale, shift}
    if (s.mode == 2) {
        if (2 * gy + 1 >= s.h || 2 * gx + 1 >= s.w) return 0.f;
        if (s.act & 2) {                             // volume source: avg_pool3d 2x2x2 (unet.py:88,97)
            if (2 * gz + 1 >= s.d) return 0.f;
            float acc8 = 0.f;
#pragma unroll
            for (int dz = 0; dz < 2; ++dz) {
                const float* p = s.x + ((plane * s.d + 2 * gz + dz) * s.h + 2 * gy) * s.w + 2 * gx;
                acc8 += act(p[0], mean, rstd, slope) + act(p[1], mean, rstd, slope) +
                        act(p[s.w], mean, rstd, slope) + act(p[s.w + 1], mean, rstd, slope);
            }
            return 0.125f * acc8;
        }
        const float* p = s.x + (plane * s.h + 2 * gy) * s.w + 2 * gx;
        return 0.25f * (act(p[0], mean, rstd, slope) + act(p[1], mean, rstd, slope) +
                        act(p[s.w], mean, rstd, slope) + act(p[s.w + 1], mean, rstd, slope));
    }
    if (gz >= s.d || gy >= s.h || gx >= s.w) return 0.f;
    const float v = s.x[((plane * s.d + gz) * s.h + gy) * s.w + gx];
    return s.mode == 0 ? v : act(v, mean, rstd, slope);
}

}  // namespace cine
