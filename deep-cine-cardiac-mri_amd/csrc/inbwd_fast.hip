// inbwd_fast.hip -- InstanceNorm + LeakyReLU backward (grad.h: InBwdArgs) for the U-Nets' plane shapes, one pass over HBM.
//
// d/d(raw) = rstd (g' - mean(g') - xhat mean(g' xhat)),  g' = g lrelu'(xhat),  per (sample, channel) plane (what autograd derives
// for unet.py:161-162).  The general kernel (grad_kernels.hip: in_lrelu_bwd_kernel) walks the plane twice element by element with
// an integer division per element; here a plane of <= 4096 elements lives in registers between the reduction and the apply pass:
// 16-byte loads of the raw tensor and of the incoming gradient (a window of a conv's input gradient over the concat, plus
// optionally the 2x2 average-pool gradient of the level below, unet.py:97), one 16-byte store -- 2 reads + 1 write per element.
// Planes of <= 1024 elements take one wave each (no LDS, no barrier), larger ones a workgroup.
#include "grad.h"

namespace cine {
namespace {

template <bool WAVE>
__global__ __launch_bounds__(256) void in_lrelu_bwd_fast_kernel(InBwdArgs a) {
    constexpr int K = 4;                                   // 16-byte pieces per thread
    __shared__ float red[2][4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long planes = (long)a.n * a.c;
    const long plane = WAVE ? (long)blockIdx.x * 4 + wave : blockIdx.x;
    const bool live = plane < planes;
    const long pl = live ? plane : planes - 1;
    const int n = (int)(pl / a.c), c = (int)(pl - (long)n * a.c);
    const int pe4 = (a.h * a.w) >> 2, w4 = a.w >> 2;
    const float2 mr = merge_partials(a.part + pl * a.np * 3, a.np, a.eps);
    const float scale = mr.y, shift = -mr.x * mr.y;
    const float4* r = reinterpret_cast<const float4*>(a.r + pl * (long)a.h * a.w);
    float4* gr = reinterpret_cast<float4*>(a.gr + pl * (long)a.h * a.w);
    const float4* qa = reinterpret_cast<const float4*>(a.a.g + ((long)n * a.a.c_total + a.a.c_off + c) * a.a.gh * a.a.gw);
    const float* qb = a.b.type ? a.b.g + ((long)n * a.b.c_total + a.b.c_off + c) * a.b.gh * a.b.gw : nullptr;
    const int t0 = WAVE ? lane : threadIdx.x, ts = WAVE ? 64 : 256;
    float4 xh[K], g[K];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int i = 0; i < K; ++i) {
        const int e4 = t0 + i * ts;
        if (e4 < pe4) {
            const float4 rv = r[e4];
            float4 gv = qa[e4];
            if (qb) {
                const int y = e4 / w4, x4 = e4 - y * w4, py = y >> 1;
                if (py < a.b.gh) {
                    const float2 p = *reinterpret_cast<const float2*>(qb + (long)py * a.b.gw + 2 * x4);
                    gv.x += 0.25f * p.x; gv.y += 0.25f * p.x; gv.z += 0.25f * p.y; gv.w += 0.25f * p.y;
                }
            }
            float4 h;
            h.x = fmaf(rv.x, scale, shift); h.y = fmaf(rv.y, scale, shift); h.z = fmaf(rv.z, scale, shift); h.w = fmaf(rv.w, scale, shift);
            gv.x = h.x > 0.f ? gv.x : gv.x * a.slope; gv.y = h.y > 0.f ? gv.y : gv.y * a.slope;
            gv.z = h.z > 0.f ? gv.z : gv.z * a.slope; gv.w = h.w > 0.f ? gv.w : gv.w * a.slope;
            s1 += (gv.x + gv.y) + (gv.z + gv.w);
            s2 = fmaf(gv.x, h.x, s2); s2 = fmaf(gv.y, h.y, s2); s2 = fmaf(gv.z, h.z, s2); s2 = fmaf(gv.w, h.w, s2);
            xh[i] = h; g[i] = gv;
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { s1 += __shfl_xor(s1, o, 64); s2 += __shfl_xor(s2, o, 64); }
    if (!WAVE) {
        if (lane == 0) { red[0][wave] = s1; red[1][wave] = s2; }
        __syncthreads();
        s1 = (red[0][0] + red[0][1]) + (red[0][2] + red[0][3]);
        s2 = (red[1][0] + red[1][1]) + (red[1][2] + red[1][3]);
    }
    if (!live) return;
    const float pe = (float)(a.h * a.w);
    const float m1 = s1 / pe, m2 = s2 / pe * drop_k2(a.drop, pl);
#pragma unroll
    for (int i = 0; i < K; ++i) {
        const int e4 = t0 + i * ts;
        if (e4 < pe4) {
            float4 o;
            o.x = scale * (g[i].x - m1 - xh[i].x * m2); o.y = scale * (g[i].y - m1 - xh[i].y * m2);
            o.z = scale * (g[i].z - m1 - xh[i].z * m2); o.w = scale * (g[i].w - m1 - xh[i].w * m2);
            gr[e4] = o;
        }
    }
}

}  // namespace

// Takes the launch when the shapes fit (else *handled = false and the general kernel runs): window piece of the tensor's own row
// length, optional pool piece of exactly half the width, rows of a multiple of 4 floats, planes of <= 4096 elements, 16-byte aligned.
int launch_in_lrelu_bwd_fast(const InBwdArgs& a, hipStream_t st, bool* handled) {
    *handled = false;
    const long pe = (long)a.h * a.w;
    if (a.a.type != 1 || a.a.gw != a.w || a.a.gh < a.h || (a.w & 3) || pe > 4096) return CINE_OK;
    if (a.b.type != 0 && (a.b.type != 2 || 2 * a.b.gw != a.w || ((long)a.b.gh * a.b.gw) % 2 != 0)) return CINE_OK;
    if (((long)a.a.gh * a.a.gw) % 4 != 0) return CINE_OK;
    auto al = [](const void* p, size_t n) { return reinterpret_cast<uintptr_t>(p) % n == 0; };
    if (!al(a.r, 16) || !al(a.gr, 16) || !al(a.a.g, 16) || (a.b.type && !al(a.b.g, 8))) return CINE_OK;
    const long planes = (long)a.n * a.c;
    if (planes <= 0 || planes > 0x7fffffffL) return CINE_OK;
    *handled = true;
    if (pe <= 1024) hipLaunchKernelGGL(in_lrelu_bwd_fast_kernel<true>, dim3((unsigned)ceil_div(planes, 4L)), dim3(256), 0, st, a);
    else hipLaunchKernelGGL(in_lrelu_bwd_fast_kernel<false>, dim3((unsigned)planes), dim3(256), 0, st, a);
    return check_launch("in_lrelu_bwd_fast_kernel");
}

}  // namespace cine
