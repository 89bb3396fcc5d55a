// grad_kernels.hip -- gradients of the U-Net building blocks (training through the HIP path).
//
// The forward pass keeps every layer's RAW output with its InstanceNorm statistics records; normalised / activated /
// pooled / concatenated tensors never exist in HBM (conv_kernels.hip).  The backward pass mirrors that:
//   in_lrelu_bwd_kernel : d/d(raw) from d/d(activated) for one (sample, channel) plane -- LeakyReLU mask, the two
//                         InstanceNorm reductions (sum g, sum g xhat) and the result in one kernel; the pieces of the
//                         incoming gradient (concat half, zero-pad crop, 2x2 average-pool gradient) are gathered on load
//   wgrad_mfma_kernel   : weight gradient as a GEMM on v_mfma_f32_16x16x4_f32 with K = pixels x samples:
//                         D[ci][co] (per tap) += X[ci][p + tap] * G[co][p]; the input operand is re-activated on load
//                         exactly like the forward's staging (modes 0 / 1 / 2, concat), partial sums per sample chunk,
//                         added in a fixed order by wgrad_reduce_kernel (deterministic, no atomics)
//   the input gradients (dgrad) are forward convolutions with re-packed weights (conv_kernels.hip: cine_*_dgrad).
// Reference: reconstruction/models/denoisers/unet.py:73-125 (what autograd differentiates there).
#include <algorithm>
#include "grad.h"
#include <atomic>
#include <mutex>
#include <cstdlib>

namespace cine {

__device__ __forceinline__ float wave_sum_g(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// ---------------------------------------------------------------- InstanceNorm + LeakyReLU backward
// one consumer's share of g(y, x) for channel ch of sample n (see grad.h)
__device__ __forceinline__ float grad_piece(const GradPiece& p, int n, int c, int ch, int y, int x) {
    if (p.type == 1) return p.g[(((long)n * p.c_total + p.c_off + ch) * p.gh + y) * p.gw + x];
    if (p.type == 2) {
        const int py = y >> 1, px = x >> 1;
        return (py < p.gh && px < p.gw) ? 0.25f * p.g[(((long)n * p.c_total + p.c_off + ch) * p.gh + py) * p.gw + px] : 0.f;
    }
    if (p.type == 3) {
        const int Y = y >> 1, X = x >> 1;
        if (Y >= p.gh || X >= p.gw) return 0.f;
        const long plane = (long)p.gh * p.gw;
        const float* q = p.g + (((long)n * p.c_total + p.c_off + ch) * p.gh + Y) * p.gw + X;
        const float ll = q[0], hl = q[(long)c * plane], lh = q[2L * c * plane], hh = q[3L * c * plane];
        const bool ry = y & 1, rx = x & 1;
        // forward: LL = x1+x2+x3+x4, HL = -x1-x2+x3+x4, LH = -x1+x2-x3+x4, HH = x1-x2-x3+x4 (x1 even/even, x2 odd row, x3 odd column), all / 2
        const float v = !ry && !rx ? ll - hl - lh + hh : (ry && !rx ? ll - hl + lh - hh : (!ry && rx ? ll + hl - lh - hh : ll + hl + lh + hh));
        return 0.5f * v;
    }
    if (p.type == 4) {
        const int cq = c >> 2, k = ch / cq, cc = ch - k * cq;
        if (2 * y + 1 >= p.gh || 2 * x + 1 >= p.gw) return 0.f;
        const float* q = p.g + (((long)n * p.c_total + p.c_off + cc) * p.gh + 2 * y) * p.gw + 2 * x;
        const float g00 = q[0], g01 = q[1], g10 = q[p.gw], g11 = q[p.gw + 1];      // g[row parity][column parity]
        const float v = k == 0 ? g00 + g10 + g01 + g11 : (k == 1 ? -g00 - g10 + g01 + g11 : (k == 2 ? -g00 + g10 - g01 + g11 : g00 - g10 - g01 + g11));
        return 0.5f * v;
    }
    return 0.f;
}
__device__ __forceinline__ float in_bwd_g(const InBwdArgs& a, int n, int ch, int y, int x) {
    return grad_piece(a.a, n, a.c, ch, y, x) + (a.b.type ? grad_piece(a.b, n, a.c, ch, y, x) : 0.f);
}
// window / pooled pieces (the U-Net's): the plane's base pointer once, then one multiply-add per element
__device__ __forceinline__ const float* piece_plane(const GradPiece& p, int n, int ch) {
    return p.type ? p.g + ((long)n * p.c_total + p.c_off + ch) * p.gh * p.gw * (p.type >= 5 ? p.gd : 1) : nullptr;
}
__device__ __forceinline__ float piece_at(const GradPiece& p, const float* q, int y, int x) {
    if (p.type == 1) return q[y * p.gw + x];
    if (p.type == 2) {
        const int py = y >> 1, px = x >> 1;
        return (py < p.gh && px < p.gw) ? 0.25f * q[py * p.gw + px] : 0.f;
    }
    // volumes as (d vh, w) planes: row y = z vh + yy
    const int z = y / p.vh, yy = y - z * p.vh;
    if (p.type == 5) return q[((long)z * p.gh + yy) * p.gw + x];
    const int pz = z >> 1, py = yy >> 1, px = x >> 1;
    return (pz < p.gd && py < p.gh && px < p.gw) ? 0.125f * q[((long)pz * p.gh + py) * p.gw + px] : 0.f;
}

// WAVE = true: one wave per plane (small planes), else one workgroup per plane.  HAAR: a piece is a wavelet adjoint (MWCNN)
template <bool WAVE, bool HAAR>
__global__ __launch_bounds__(256) void in_lrelu_bwd_kernel(InBwdArgs a) {
    __shared__ float red[2][4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long plane = WAVE ? (long)blockIdx.x * 4 + wave : blockIdx.x;
    const long planes = (long)a.n * a.c;
    const bool live = plane < planes;
    const long pl = live ? plane : planes - 1;
    const int n = (int)(pl / a.c), c = (int)(pl - (long)n * a.c);
    const int pe = a.h * a.w;
    const float2 mr = merge_partials(a.part + pl * a.np * 3, a.np, a.eps);
    const float scale = mr.y, shift = -mr.x * mr.y;
    const float* r = a.r + pl * pe;
    float* gr = a.gr + pl * pe;
    const float* qa = HAAR ? nullptr : piece_plane(a.a, n, c);
    const float* qb = HAAR ? nullptr : piece_plane(a.b, n, c);
    auto grad_at = [&](int y, int x) -> float {
        if constexpr (HAAR) return in_bwd_g(a, n, c, y, x);
        else return piece_at(a.a, qa, y, x) + (qb ? piece_at(a.b, qb, y, x) : 0.f);
    };
    const int t0 = WAVE ? lane : threadIdx.x, ts = WAVE ? 64 : 256;
    float s1 = 0.f, s2 = 0.f;
    for (int e = t0; e < pe; e += ts) {
        const int y = e / a.w, x = e - y * a.w;
        const float xh = fmaf(r[e], scale, shift);
        float g = grad_at(y, x);
        g = xh > 0.f ? g : g * a.slope;
        s1 += g; s2 = fmaf(g, xh, s2);
    }
    s1 = wave_sum_g(s1); s2 = wave_sum_g(s2);
    if (!WAVE) {
        if (lane == 0) { red[0][wave] = s1; red[1][wave] = s2; }
        __syncthreads();
        s1 = red[0][0] + red[0][1] + red[0][2] + red[0][3];
        s2 = red[1][0] + red[1][1] + red[1][2] + red[1][3];
    }
    if (!live) return;
    const float m1 = s1 / pe, m2 = s2 / pe * drop_k2(a.drop, pl);
    for (int e = t0; e < pe; e += ts) {
        const int y = e / a.w, x = e - y * a.w;
        const float xh = fmaf(r[e], scale, shift);
        float g = grad_at(y, x);
        g = xh > 0.f ? g : g * a.slope;
        gr[e] = scale * (g - m1 - xh * m2);
    }
}

int launch_in_lrelu_bwd(const InBwdArgs& a, hipStream_t st) {
    CINE_REQUIRE(a.r && a.part && a.gr && a.a.g && a.a.type >= 1 && a.a.type <= 6 && a.n > 0 && a.c > 0 && a.h > 0 && a.w > 0 && a.np > 0, CINE_EINVAL,
                 "in_lrelu_bwd: bad arguments");
    const bool haar = a.a.type == 3 || a.a.type == 4 || a.b.type == 3 || a.b.type == 4;
    for (const GradPiece* p : {&a.a, &a.b}) {
        if (!p->type) continue;
        CINE_REQUIRE(p->g && p->type >= 1 && p->type <= 6 && p->c_off >= 0, CINE_EINVAL, "in_lrelu_bwd: bad gradient piece");
        if (p->type >= 5) {
            CINE_REQUIRE(!haar && p->vh > 0 && a.h % p->vh == 0 && p->gd > 0, CINE_EINVAL, "in_lrelu_bwd: volume piece (vh %d, gd %d) of a (%d, %d) plane", p->vh, p->gd, a.h, a.w);
            CINE_REQUIRE(p->type != 5 || (p->gd >= a.h / p->vh && p->gh >= p->vh && p->gw >= a.w), CINE_EINVAL, "in_lrelu_bwd: volume window smaller than the tensor");
            CINE_REQUIRE((long)p->gd * p->gh * p->gw < (1L << 31), CINE_EUNSUPPORTED, "in_lrelu_bwd: volume piece too large");
        }
        const int need_c = p->type == 3 ? 3 * a.c + a.c : (p->type == 4 ? a.c / 4 : a.c);
        CINE_REQUIRE(p->c_off + need_c <= p->c_total && (p->type != 4 || a.c % 4 == 0), CINE_EINVAL, "in_lrelu_bwd: gradient piece channels");
        CINE_REQUIRE(p->type != 1 || (p->gh >= a.h && p->gw >= a.w), CINE_EINVAL,
                     "in_lrelu_bwd: gradient window (%d, %d) smaller than the tensor (%d, %d)", p->gh, p->gw, a.h, a.w);
    }
    const long planes = (long)a.n * a.c;
    if (!haar) {
        ProfScope prof(F_STATS, st);
        bool handled = false;
        const int e = launch_in_lrelu_bwd_fast(a, st, &handled);
        if (e || handled) return e;
    }
    CINE_REQUIRE(haar || ((long)a.a.gh * a.a.gw < (1L << 31) && (long)a.b.gh * a.b.gw < (1L << 31)), CINE_EUNSUPPORTED, "in_lrelu_bwd: plane too large");
    ProfScope prof(F_STATS, st);
    if ((long)a.h * a.w <= 1024) {
        CINE_REQUIRE(ceil_div(planes, 4L) <= 0x7fffffffL, CINE_EUNSUPPORTED, "in_lrelu_bwd: too many planes");
        if (haar) hipLaunchKernelGGL((in_lrelu_bwd_kernel<true, true>), dim3((unsigned)ceil_div(planes, 4L)), dim3(256), 0, st, a);
        else hipLaunchKernelGGL((in_lrelu_bwd_kernel<true, false>), dim3((unsigned)ceil_div(planes, 4L)), dim3(256), 0, st, a);
    } else {
        CINE_REQUIRE(planes <= 0x7fffffffL, CINE_EUNSUPPORTED, "in_lrelu_bwd: too many planes");
        if (haar) hipLaunchKernelGGL((in_lrelu_bwd_kernel<false, true>), dim3((unsigned)planes), dim3(256), 0, st, a);
        else hipLaunchKernelGGL((in_lrelu_bwd_kernel<false, false>), dim3((unsigned)planes), dim3(256), 0, st, a);
    }
    return check_launch("in_lrelu_bwd_kernel");
}

// Few, large planes (volumes: 16 channels of 15 x 200 x 200): a plane is cut into chunks of kInBwdChunk elements, one workgroup each.
// Pass 1 writes the chunk's two sums, pass 2 re-adds the plane's chunk sums in index order (deterministic) and applies.  Window / pooled
// pieces only.
constexpr int kInBwdChunk = 8192;
__global__ __launch_bounds__(256) void in_lrelu_bwd_sums_kernel(InBwdArgs a, float* __restrict__ ws, int nchunk) {
    __shared__ float red[2][4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long pl = blockIdx.y;
    const int n = (int)(pl / a.c), c = (int)(pl - (long)n * a.c);
    const int pe = a.h * a.w;
    const float2 mr = merge_partials(a.part + pl * a.np * 3, a.np, a.eps);
    const float scale = mr.y, shift = -mr.x * mr.y;
    const float* r = a.r + pl * pe;
    const float* qa = piece_plane(a.a, n, c);
    const float* qb = piece_plane(a.b, n, c);
    const int e0 = blockIdx.x * kInBwdChunk, e1 = min(pe, e0 + kInBwdChunk);
    float s1 = 0.f, s2 = 0.f;
    for (int e = e0 + threadIdx.x; e < e1; e += 256) {
        const int y = e / a.w, x = e - y * a.w;
        const float xh = fmaf(r[e], scale, shift);
        float g = piece_at(a.a, qa, y, x) + (qb ? piece_at(a.b, qb, y, x) : 0.f);
        g = xh > 0.f ? g : g * a.slope;
        s1 += g; s2 = fmaf(g, xh, s2);
    }
    s1 = wave_sum_g(s1); s2 = wave_sum_g(s2);
    if (lane == 0) { red[0][wave] = s1; red[1][wave] = s2; }
    __syncthreads();
    if (threadIdx.x == 0) {
        float* o = ws + (pl * nchunk + blockIdx.x) * 2;
        o[0] = red[0][0] + red[0][1] + red[0][2] + red[0][3];
        o[1] = red[1][0] + red[1][1] + red[1][2] + red[1][3];
    }
}
__global__ __launch_bounds__(256) void in_lrelu_bwd_apply_kernel(InBwdArgs a, const float* __restrict__ ws, int nchunk) {
    const long pl = blockIdx.y;
    const int n = (int)(pl / a.c), c = (int)(pl - (long)n * a.c);
    const int pe = a.h * a.w;
    const float2 mr = merge_partials(a.part + pl * a.np * 3, a.np, a.eps);
    const float scale = mr.y, shift = -mr.x * mr.y;
    float s1 = 0.f, s2 = 0.f;
    for (int k = 0; k < nchunk; ++k) { s1 += ws[(pl * nchunk + k) * 2]; s2 += ws[(pl * nchunk + k) * 2 + 1]; }
    const float m1 = s1 / pe, m2 = s2 / pe * drop_k2(a.drop, pl);
    const float* r = a.r + pl * pe;
    float* gr = a.gr + pl * pe;
    const float* qa = piece_plane(a.a, n, c);
    const float* qb = piece_plane(a.b, n, c);
    const int e0 = blockIdx.x * kInBwdChunk, e1 = min(pe, e0 + kInBwdChunk);
    for (int e = e0 + threadIdx.x; e < e1; e += 256) {
        const int y = e / a.w, x = e - y * a.w;
        const float xh = fmaf(r[e], scale, shift);
        float g = piece_at(a.a, qa, y, x) + (qb ? piece_at(a.b, qb, y, x) : 0.f);
        g = xh > 0.f ? g : g * a.slope;
        gr[e] = scale * (g - m1 - xh * m2);
    }
}
// The same two passes with 16-byte accesses for what the 3-D U-Net's large levels hand over: the consumer's gradient as wide as the tensor (a type 1
// window with gw == w: the plane is contiguous), optionally plus the 2x2x2 pool adjoint (type 6); w % 4 == 0.  One thread = four consecutive elements
// of one row (the scalar kernels moved 2.1 TB/s on cfg 4's level 0: 90 us per layer for 192 MB).
template <bool POOL>
__device__ __forceinline__ float4 in_bwd_g4(const InBwdArgs& a, const float4* qa4, const float* qb, int e4) {
    float4 g = qa4[e4];
    if constexpr (POOL) {
        const int e = 4 * e4, y = e / a.w, x = e - y * a.w;
        const int z = y / a.b.vh, yy = y - z * a.b.vh;
        const int pz = z >> 1, py = yy >> 1, px = x >> 1;
        const bool ok = pz < a.b.gd && py < a.b.gh;
        const float* q = qb + ((long)min(pz, a.b.gd - 1) * a.b.gh + min(py, a.b.gh - 1)) * a.b.gw;
        const float v0 = (ok && px < a.b.gw) ? 0.125f * q[min(px, a.b.gw - 1)] : 0.f;
        const float v1 = (ok && px + 1 < a.b.gw) ? 0.125f * q[min(px + 1, a.b.gw - 1)] : 0.f;
        g.x += v0; g.y += v0; g.z += v1; g.w += v1;
    }
    return g;
}
template <bool POOL>
__global__ __launch_bounds__(256) void in_lrelu_bwd_sums_vec_kernel(InBwdArgs a, float* __restrict__ ws, int nchunk) {
    __shared__ float red[2][4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long pl = blockIdx.y;
    const int n = (int)(pl / a.c), c = (int)(pl - (long)n * a.c);
    const int pe = a.h * a.w;
    const float2 mr = merge_partials(a.part + pl * a.np * 3, a.np, a.eps);
    const float scale = mr.y, shift = -mr.x * mr.y;
    const float4* r4 = reinterpret_cast<const float4*>(a.r + pl * pe);
    const float4* qa4 = reinterpret_cast<const float4*>(piece_plane(a.a, n, c));
    const float* qb = POOL ? piece_plane(a.b, n, c) : nullptr;
    const int e0 = blockIdx.x * (kInBwdChunk / 4), e1 = min(pe / 4, e0 + kInBwdChunk / 4);
    float s1 = 0.f, s2 = 0.f;
    for (int e4 = e0 + threadIdx.x; e4 < e1; e4 += 256) {
        const float4 rv = r4[e4];
        const float4 gv = in_bwd_g4<POOL>(a, qa4, qb, e4);
        const float rr[4] = {rv.x, rv.y, rv.z, rv.w}, gg[4] = {gv.x, gv.y, gv.z, gv.w};
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const float xh = fmaf(rr[u], scale, shift);
            const float g = xh > 0.f ? gg[u] : gg[u] * a.slope;
            s1 += g; s2 = fmaf(g, xh, s2);
        }
    }
    s1 = wave_sum_g(s1); s2 = wave_sum_g(s2);
    if (lane == 0) { red[0][wave] = s1; red[1][wave] = s2; }
    __syncthreads();
    if (threadIdx.x == 0) {
        float* o = ws + (pl * nchunk + blockIdx.x) * 2;
        o[0] = red[0][0] + red[0][1] + red[0][2] + red[0][3];
        o[1] = red[1][0] + red[1][1] + red[1][2] + red[1][3];
    }
}
template <bool POOL>
__global__ __launch_bounds__(256) void in_lrelu_bwd_apply_vec_kernel(InBwdArgs a, const float* __restrict__ ws, int nchunk) {
    const long pl = blockIdx.y;
    const int n = (int)(pl / a.c), c = (int)(pl - (long)n * a.c);
    const int pe = a.h * a.w;
    const float2 mr = merge_partials(a.part + pl * a.np * 3, a.np, a.eps);
    const float scale = mr.y, shift = -mr.x * mr.y;
    float s1 = 0.f, s2 = 0.f;
    for (int k = 0; k < nchunk; ++k) { s1 += ws[(pl * nchunk + k) * 2]; s2 += ws[(pl * nchunk + k) * 2 + 1]; }
    const float m1 = s1 / pe, m2 = s2 / pe * drop_k2(a.drop, pl);
    const float4* r4 = reinterpret_cast<const float4*>(a.r + pl * pe);
    float4* gr4 = reinterpret_cast<float4*>(a.gr + pl * pe);
    const float4* qa4 = reinterpret_cast<const float4*>(piece_plane(a.a, n, c));
    const float* qb = POOL ? piece_plane(a.b, n, c) : nullptr;
    const int e0 = blockIdx.x * (kInBwdChunk / 4), e1 = min(pe / 4, e0 + kInBwdChunk / 4);
    for (int e4 = e0 + threadIdx.x; e4 < e1; e4 += 256) {
        const float4 rv = r4[e4];
        const float4 gv = in_bwd_g4<POOL>(a, qa4, qb, e4);
        const float rr[4] = {rv.x, rv.y, rv.z, rv.w}, gg[4] = {gv.x, gv.y, gv.z, gv.w};
        float o[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const float xh = fmaf(rr[u], scale, shift);
            const float g = xh > 0.f ? gg[u] : gg[u] * a.slope;
            o[u] = scale * (g - m1 - xh * m2);
        }
        gr4[e4] = make_float4(o[0], o[1], o[2], o[3]);
    }
}
static bool in_bwd_vec_ok(const InBwdArgs& a) {
    auto al = [](const void* p) { return reinterpret_cast<uintptr_t>(p) % 16 == 0; };
    if (a.a.type != 1 || a.a.gw != a.w || (a.w & 3) || (((long)a.a.gh * a.a.gw) & 3)) return false;
    if (a.b.type != 0 && a.b.type != 6) return false;
    return al(a.r) && al(a.gr) && al(a.a.g) && (((long)a.a.c_off * a.a.gh * a.a.gw) & 3) == 0;
}
size_t in_lrelu_bwd_ws_floats(int n, int c, int h, int w) {
    const long pe = (long)h * w;
    return pe > 4 * kInBwdChunk ? (size_t)n * c * ceil_div(pe, (long)kInBwdChunk) * 2 : 0;
}
int launch_in_lrelu_bwd_split(const InBwdArgs& a, float* ws, size_t ws_floats, hipStream_t st) {
    const size_t need = in_lrelu_bwd_ws_floats(a.n, a.c, a.h, a.w);
    const bool haar = a.a.type == 3 || a.a.type == 4 || a.b.type == 3 || a.b.type == 4;
    if (!need || !ws || haar || (long)a.n * a.c >= 2048) return launch_in_lrelu_bwd(a, st);
    CINE_REQUIRE(ws_floats >= need, CINE_EWORKSPACE, "in_lrelu_bwd: workspace too small");
    CINE_REQUIRE(a.r && a.part && a.gr && a.a.g && a.np > 0 && (long)a.n * a.c <= 65535, CINE_EINVAL, "in_lrelu_bwd: bad arguments");
    CINE_REQUIRE(a.a.type != 1 || (a.a.gh >= a.h && a.a.gw >= a.w), CINE_EINVAL, "in_lrelu_bwd: gradient window smaller than the tensor");
    for (const GradPiece* p : {&a.a, &a.b})
        if (p->type >= 5) {
            CINE_REQUIRE(p->g && p->type <= 6 && p->vh > 0 && a.h % p->vh == 0 && p->gd > 0 && (long)p->gd * p->gh * p->gw < (1L << 31), CINE_EINVAL, "in_lrelu_bwd: bad volume piece");
            CINE_REQUIRE(p->type != 5 || (p->gd >= a.h / p->vh && p->gh >= p->vh && p->gw >= a.w), CINE_EINVAL, "in_lrelu_bwd: volume window smaller than the tensor");
        }
    const int nchunk = (int)ceil_div((long)a.h * a.w, (long)kInBwdChunk);
    ProfScope prof(F_STATS, st);
    const dim3 grid((unsigned)nchunk, (unsigned)((long)a.n * a.c));
    if (in_bwd_vec_ok(a)) {
        if (a.b.type == 6) {
            hipLaunchKernelGGL(in_lrelu_bwd_sums_vec_kernel<true>, grid, dim3(256), 0, st, a, ws, nchunk);
            hipLaunchKernelGGL(in_lrelu_bwd_apply_vec_kernel<true>, grid, dim3(256), 0, st, a, ws, nchunk);
        } else {
            hipLaunchKernelGGL(in_lrelu_bwd_sums_vec_kernel<false>, grid, dim3(256), 0, st, a, ws, nchunk);
            hipLaunchKernelGGL(in_lrelu_bwd_apply_vec_kernel<false>, grid, dim3(256), 0, st, a, ws, nchunk);
        }
        return check_launch("in_lrelu_bwd_apply_vec_kernel");
    }
    hipLaunchKernelGGL(in_lrelu_bwd_sums_kernel, grid, dim3(256), 0, st, a, ws, nchunk);
    hipLaunchKernelGGL(in_lrelu_bwd_apply_kernel, grid, dim3(256), 0, st, a, ws, nchunk);
    return check_launch("in_lrelu_bwd_apply_kernel");
}


// ---------------------------------------------------------------- weight gradient (MFMA)
// One workgroup = 4 waves = one 16-channel chunk of the conv input x one block of 16*CT*WM output rows x one chunk of
// samples.  Per tile of NPIX = TH x TW pixels it stages the (re-activated) input tile with its halo and the output-gradient
// tile in LDS, channel-major with a channel stride == 2 (mod 32) floats: the MFMA operands are read with ds_read_b32 by
// lanes (channel q, pixel kk) -> bank 2 q + kk, conflict-free.  WM waves split the row tiles, 4 / WM waves split the pixel
// groups (K); the K-split accumulators are added through LDS in a fixed order at the end.
struct WgLaunch {
    WgArgs a;
    int chunk, nchunks;        // samples per chunk, chunks per weight set
    int rowsp, cinp;           // padded to 16
    float* part;               // [2][nchunks][rowsp][cinp][TAPS]
    int fast_in, fast_g;       // vectorised + prefetched staging of the input / output-gradient tiles (see launch_wgrad)
};

template <int TAPS, int TW, int CT, int WM, int NPIX>
struct WgCfg {
    static constexpr int HALO = TAPS == 9 ? 1 : 0;
    static constexpr int TH = NPIX / TW;
    static constexpr int ROWS = TH + 2 * HALO, COLS = TW + 2 * HALO;
    // LDS row of the input tile: interior columns [0, TW), then (3x3 only) the right halo at TW and the left halo at COLS - 1,
    // so that interior pieces are written with aligned 8-byte stores; image column x of the tile lives at (x + COLS) % COLS
    static constexpr int pad2(int v) { return ((v + 29) / 32) * 32 + 2; }       // smallest >= v with == 2 (mod 32)
    static constexpr int PSI = pad2(ROWS * COLS), PSG = pad2(NPIX);
    static constexpr int COB = 16 * CT * WM, WK = 4 / WM;
    static constexpr int IN_FLOATS = 16 * PSI, G_FLOATS = COB * PSG;
    static constexpr int RED_FLOATS = WK > 1 ? TAPS * CT * WM * 256 : 0;       // the accumulators of one K-split slice
    static constexpr int LDS_FLOATS = (IN_FLOATS + G_FLOATS > RED_FLOATS ? IN_FLOATS + G_FLOATS : RED_FLOATS);   // + the statistics tables (launch_wg_cfg)
    static constexpr int PW = TW >= 4 ? 4 : 2, PR = TW / PW;                    // floats per staging piece, pieces per row
    static constexpr int NIU = 16 * ROWS * PR, NIP = (NIU + 255) / 256;         // input pieces per tile / per thread
    static constexpr int NGU = COB * NPIX / PW, NGP = (NGU + 255) / 256;        // output-gradient pieces
    static_assert(NPIX % TW == 0 && NPIX % 4 == 0 && COLS % 2 == 0, "tile shape");
};

template <int PW> struct WgPiece;
template <> struct WgPiece<4> { typedef float4 T; };
template <> struct WgPiece<2> { typedef float2 T; };

// ADD: the conv input is the SUM of the two sources (MWCNN's additive skips); its own instantiation, because the second inlined fetch per value
// costs the plain kernels their registers (the compiler then keeps the kernel arguments in scratch: 187 -> 212 us per level-0 layer)
template <int TAPS, int TW, int CT, int WM, int NPIX, bool ADD>
__global__ __launch_bounds__(256) void wgrad_mfma_kernel(WgLaunch L) {
    using C = WgCfg<TAPS, TW, CT, WM, NPIX>;
    constexpr int HALO = C::HALO, PW = C::PW, PR = C::PR;
    typedef typename WgPiece<PW>::T piece_t;
    extern __shared__ __align__(16) float smem_g[];
    float* in_lds = smem_g;
    float* g_lds = smem_g + C::IN_FLOATS;
    float* st_lds = smem_g + C::LDS_FLOATS;           // two tables (sample parity) of {scale, shift} of ALL source channels (source 0's, then source 1's)
    const WgArgs& a = L.a;
    const int nch = a.s0.c + a.s1.c;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int q = lane & 15, kk = lane >> 4;
    const int wm = wave % WM, wk = wave / WM;
    const int ci0 = blockIdx.x * 16, co0 = blockIdx.y * C::COB;
    const int set = blockIdx.z / L.nchunks, ch = blockIdx.z - set * L.nchunks;
    const int c0n = src_cin(a.s0);

    f32x4 acc[TAPS][CT];
#pragma unroll
    for (int t = 0; t < TAPS; ++t)
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) acc[t][ct] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const int tiles_w = (a.W + TW - 1) / TW, tiles_h = (a.H + C::TH - 1) / C::TH;
    const int ntile = tiles_w * tiles_h;
    // this workgroup's share: `chunk` consecutive (sample, tile) work items of its weight set
    const int sbase = set ? a.set_split : 0;
    const int items = ((set ? a.n : a.set_split) - sbase) * ntile;
    const int ibeg = min(ch * L.chunk, items), iend = min(ibeg + L.chunk, items);
    const int total = iend - ibeg;
    const Src gs{a.g, nullptr, a.g_c, a.g_mode, a.g_h, a.g_w, 0, 0, 1};

    // the fields the vectorised staging selects per lane, as values (a per-lane choice between the two structs themselves makes the compiler
    // keep a copy of the kernel arguments in scratch and re-load from it in the staging loops)
    const float* const s0x = a.s0.x; const float* const s1x = a.s1.x;
    const int s0c = a.s0.c, s1c = a.s1.c, s0h = a.s0.h, s1h = a.s1.h, s0mode = a.s0.mode, s1mode = a.s1.mode;
    // statistics table of one sample: {scale, shift} of every source channel (one loop per source: the source is uniform in each)
    auto table_of = [&](const Src& s, int n, float* st) {
        const bool stats = s.mode == 1 || s.mode == 2 || (s.mode >= 3 && (s.act & 1));
        for (int cl = tid; cl < s.c; cl += 256) {
            float2 mr = make_float2(0.f, 1.f);
            if (stats) mr = merge_partials(s.part + ((long)n * s.c + cl) * s.np * 3, s.np, a.eps);
            st[2 * cl] = mr.y; st[2 * cl + 1] = -mr.x * mr.y;
        }
    };
    auto table = [&](int n, float* st) { table_of(a.s0, n, st); table_of(a.s1, n, st + 2 * a.s0.c); };
    // one transformed input value of concat / summed channel cg (any source mode), st = this sample's table
    auto fetch_in = [&](int n, int cg, int gy, int gx, const float* st) -> float {
        if constexpr (ADD)
            return fetch_scalar(a.s0, n, cg, 0, gy, gx, st, a.slope) + fetch_scalar(a.s1, n, cg, 0, gy, gx, st + 2 * a.s0.c, a.slope);
        // (a source chosen per lane by reference or by value makes the compiler keep a copy of the kernel arguments in scratch)
        if (cg < c0n) return fetch_scalar(a.s0, n, cg, 0, gy, gx, st, a.slope);
        return fetch_scalar(a.s1, n, cg - c0n, 0, gy, gx, st + 2 * a.s0.c, a.slope);
    };
    // ---- the tile pipeline: issue(it) puts the raw pieces of tile `it` in flight into registers, commit(it) transforms them and
    // writes LDS, the sweep of tile `it` runs with the loads of tile it + 1 in flight
    piece_t xin[C::NIP], gin[C::NGP];
    auto issue = [&](int it) {
        const int n = sbase + (ibeg + it) / ntile, tile = (ibeg + it) % ntile;
        const int ty = tile / tiles_w, tx = tile - ty * tiles_w;
        const int r0 = ty * C::TH, c0 = tx * TW;
        if (L.fast_in) {
#pragma unroll
            for (int i = 0; i < C::NIP; ++i) {
                const int e = min(tid + i * 256, C::NIU - 1);
                const int k = e / (C::ROWS * PR), rem = e - k * (C::ROWS * PR);
                const int row = rem / PR, j = rem - row * PR;
                const int cg = min(ci0 + k, a.cin - 1);
                const bool f0 = cg < c0n;
                const float* sx = f0 ? s0x : s1x;
                const int sc_ = f0 ? s0c : s1c, sh_ = f0 ? s0h : s1h;
                const int cl = f0 ? cg : cg - c0n;
                const int gy = min(max(r0 - HALO + row, 0), sh_ - 1), gx = min(c0 + PW * j, a.W - PW);
                xin[i] = *reinterpret_cast<const piece_t*>(sx + (((long)n * sc_ + cl) * sh_ + gy) * a.W + gx);
            }
        }
        if (L.fast_g == 1) {
#pragma unroll
            for (int i = 0; i < C::NGP; ++i) {
                const int e = min(tid + i * 256, C::NGU - 1);
                const int co = min(co0 + e / (NPIX / PW), a.rows - 1), pp = (e % (NPIX / PW)) * PW;
                const int gy = min(r0 + pp / TW, a.H - 1), gx = min(c0 + pp % TW, a.W - PW);
                gin[i] = *reinterpret_cast<const piece_t*>(a.g + (((long)n * a.rows + co) * a.H + gy) * a.W + gx);
            }
        } else if (L.fast_g == 2) {
            // space-to-depth view (g_mode 5): rows 4 c + 2 a + b.  One unit = (c, a, PW pixels) = 2 PW consecutive floats of row 2 y + a
            // of channel c, which hold both column parities b: register pieces 2 u and 2 u + 1 (NGP is even)
#pragma unroll
            for (int i = 0; i < C::NGP / 2; ++i) {
                const int e = min(tid + i * 256, C::NGU / 2 - 1);
                const int ca = e / (NPIX / PW), pp = (e % (NPIX / PW)) * PW;       // ca = 2 c + a relative to the row block
                const int row0 = min(co0 + 2 * ca, a.rows - 2);                     // GEMM row of (c, a, b = 0)
                const int c = row0 >> 2, sa = (row0 >> 1) & 1;
                const int gy = min(r0 + pp / TW, a.H - 1), gx = min(c0 + pp % TW, a.W - PW);
                const float* src = a.g + (((long)n * a.g_c + c) * a.g_h + 2 * gy + sa) * a.g_w + 2 * gx;
                gin[2 * i] = *reinterpret_cast<const piece_t*>(src);
                gin[2 * i + 1] = *reinterpret_cast<const piece_t*>(src + PW);
            }
        }
    };
    auto commit = [&](int it) {
        const int n = sbase + (ibeg + it) / ntile, tile = (ibeg + it) % ntile;
        const int ty = tile / tiles_w, tx = tile - ty * tiles_w;
        const int r0 = ty * C::TH, c0 = tx * TW;
        const float* st = st_lds + 2 * nch * (((ibeg + it) / ntile) & 1);
        if (L.fast_in) {
#pragma unroll
            for (int i = 0; i < C::NIP; ++i) {
                const int e = tid + i * 256;
                if (e >= C::NIU) break;
                const int k = e / (C::ROWS * PR), rem = e - k * (C::ROWS * PR);
                const int row = rem / PR, j = rem - row * PR;
                const int cg = ci0 + k;
                const bool f0 = cg < c0n;
                const int smode = f0 ? s0mode : s1mode, sh_ = f0 ? s0h : s1h;
                const int gy = r0 - HALO + row, gx = c0 + PW * j;
                const bool ok = cg < a.cin && gy >= 0 && gy < sh_ && gx < a.W;
                piece_t o = xin[i];
                float* ov = reinterpret_cast<float*>(&o);
                const int cgc = min(cg, a.cin - 1);
                const float sc = st[2 * cgc], sh = st[2 * cgc + 1];
#pragma unroll
                for (int u = 0; u < PW; ++u) ov[u] = ok ? (smode == 0 ? ov[u] : act(ov[u], sc, sh, a.slope)) : 0.f;
                float* dst = in_lds + k * C::PSI + row * C::COLS + PW * j;
#pragma unroll
                for (int u = 0; u < PW; u += 2) *reinterpret_cast<float2*>(dst + u) = make_float2(ov[u], ov[u + 1]);
            }
            if (HALO) {       // the two halo columns: data only when the image is wider than the tile
                for (int e = tid; e < 16 * C::ROWS * 2; e += 256) {
                    const int k = e / (C::ROWS * 2), rem = e - k * (C::ROWS * 2);
                    const int row = rem >> 1, side = rem & 1;
                    const int gy = r0 - 1 + row, gx = side ? c0 + TW : c0 - 1, cg = ci0 + k;
                    float v = 0.f;
                    if (a.W > TW && cg < a.cin && gy >= 0 && gy < a.H && gx >= 0 && gx < a.W) v = fetch_in(n, cg, gy, gx, st);
                    in_lds[k * C::PSI + row * C::COLS + (side ? TW : C::COLS - 1)] = v;
                }
            }
        } else {
            // generic staging (pooled sources, odd widths): element by element, re-activated like the forward's operand staging
            for (int e = tid; e < 16 * C::ROWS * C::COLS; e += 256) {
                const int k = e / (C::ROWS * C::COLS), rem = e - k * (C::ROWS * C::COLS);
                const int row = rem / C::COLS, xc = rem - row * C::COLS;          // xc: tile column + HALO
                const int gy = r0 - HALO + row, gx = c0 - HALO + xc, cg = ci0 + k;
                float v = 0.f;
                if (cg < a.cin && gy >= 0 && gy < a.H && gx >= 0 && gx < a.W) v = fetch_in(n, cg, gy, gx, st);
                in_lds[k * C::PSI + row * C::COLS + (xc - HALO + C::COLS) % C::COLS] = v;
            }
        }
        if (L.fast_g == 1) {
#pragma unroll
            for (int i = 0; i < C::NGP; ++i) {
                const int e = tid + i * 256;
                if (e >= C::NGU) break;
                const int co = e / (NPIX / PW), pp = (e % (NPIX / PW)) * PW;
                const int gy = r0 + pp / TW, gx = c0 + pp % TW;
                const bool ok = co0 + co < a.rows && gy < a.H && gx < a.W;
                const piece_t o = gin[i];
                const float* ov = reinterpret_cast<const float*>(&o);
                float* dst = g_lds + co * C::PSG + pp;
#pragma unroll
                for (int u = 0; u < PW; u += 2) *reinterpret_cast<float2*>(dst + u) = ok ? make_float2(ov[u], ov[u + 1]) : make_float2(0.f, 0.f);
            }
        } else if (L.fast_g == 2) {
#pragma unroll
            for (int i = 0; i < C::NGP / 2; ++i) {
                const int e = tid + i * 256;
                if (e >= C::NGU / 2) break;
                const int ca = e / (NPIX / PW), pp = (e % (NPIX / PW)) * PW;
                const int gy = r0 + pp / TW, gx = c0 + pp % TW;
                const bool ok = co0 + 2 * ca + 1 < a.rows && gy < a.H && gx < a.W;
                float t[2 * PW];
                *reinterpret_cast<piece_t*>(t) = gin[2 * i];
                *reinterpret_cast<piece_t*>(t + PW) = gin[2 * i + 1];
                float* d0 = g_lds + (2 * ca) * C::PSG + pp;         // column parity 0
                float* d1 = d0 + C::PSG;                            // column parity 1
#pragma unroll
                for (int u = 0; u < PW; u += 2) {
                    *reinterpret_cast<float2*>(d0 + u) = ok ? make_float2(t[2 * u], t[2 * u + 2]) : make_float2(0.f, 0.f);
                    *reinterpret_cast<float2*>(d1 + u) = ok ? make_float2(t[2 * u + 1], t[2 * u + 3]) : make_float2(0.f, 0.f);
                }
            }
        } else {
            for (int e = tid; e < C::COB * NPIX; e += 256) {
                const int co = e / NPIX, p = e - co * NPIX;
                const int gy = r0 + p / TW, gx = c0 + p % TW;
                float v = 0.f;
                if (co0 + co < a.rows && gy < a.H && gx < a.W)
                    v = a.g_mode == 5 ? fetch_scalar(gs, n, co0 + co, 0, gy, gx, st_lds, a.slope)
                                      : a.g[(((long)n * a.rows + co0 + co) * a.H + gy) * a.W + gx];
                g_lds[co * C::PSG + p] = v;
            }
        }
    };

    if (total > 0) { table(sbase + ibeg / ntile, st_lds + 2 * nch * ((ibeg / ntile) & 1)); issue(0); }
    for (int it = 0; it < total; ++it) {
        __syncthreads();                                        // previous sweep done with LDS; this sample's table written
        if ((ibeg + it + 1) % ntile == 0 && it + 1 < total)     // the next item starts a new sample: its table goes into the other buffer
            table(sbase + (ibeg + it + 1) / ntile, st_lds + 2 * nch * (((ibeg + it + 1) / ntile) & 1));
        commit(it);
        __syncthreads();
        if (it + 1 < total) issue(it + 1);
        // ---- sweep: K = pixel groups of 4, software-pipelined one group ahead (the operands of group i + 1 are read from LDS
        // while the MFMAs of group i issue)
        constexpr int NSTEP = NPIX / 4 / C::WK;
        float gv[2][CT], xv[2][TAPS];
        auto load_step = [&](int i, float (&gq)[CT], float (&xq)[TAPS]) {
            const int p = 4 * (wk + i * C::WK) + kk;
            const int prow = p / TW, pcol = p % TW;
#pragma unroll
            for (int ct = 0; ct < CT; ++ct) gq[ct] = g_lds[(16 * (wm * CT + ct) + q) * C::PSG + p];
            const float* ib = in_lds + q * C::PSI + prow * C::COLS;
#pragma unroll
            for (int t = 0; t < TAPS; ++t) {
                int xc = pcol;
                if (TAPS == 9) { xc = pcol + (t % 3) - 1; xc = xc < 0 ? C::COLS - 1 : xc; }
                xq[t] = ib[(TAPS == 9 ? (t / 3) * C::COLS : 0) + xc];
            }
        };
        load_step(0, gv[0], xv[0]);
#pragma unroll
        for (int i = 0; i < NSTEP; ++i) {
            if (i + 1 < NSTEP) load_step(i + 1, gv[(i + 1) & 1], xv[(i + 1) & 1]);
#pragma unroll
            for (int t = 0; t < TAPS; ++t)
#pragma unroll
                for (int ct = 0; ct < CT; ++ct)
                    acc[t][ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(xv[i & 1][t], gv[i & 1][ct], acc[t][ct], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    // ---- the partial goes out.  lane (q, kk), register j of acc[t][ct]: D[ci = 4 kk + j][co = 16 (wm CT + ct) + q]
    float* part = L.part + (((long)blockIdx.z * L.rowsp + co0) * L.cinp + ci0) * TAPS;
    if (C::WK == 1) {
#pragma unroll
        for (int t = 0; t < TAPS; ++t)
#pragma unroll
            for (int ct = 0; ct < CT; ++ct)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    part[((long)(16 * (wm * CT + ct) + q) * L.cinp + 4 * kk + j) * TAPS + t] = acc[t][ct][j];
        return;
    }
    // K-split waves add their accumulators through LDS in a fixed order first
    __syncthreads();
    float* red = smem_g;                                          // [TAPS][CT * WM][16 ci][16 co]
    for (int turn = 0; turn < C::WK; ++turn) {
        if (wk == turn) {
#pragma unroll
            for (int t = 0; t < TAPS; ++t)
#pragma unroll
                for (int ct = 0; ct < CT; ++ct)
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        float* o = red + ((t * (CT * WM) + wm * CT + ct) * 16 + 4 * kk + j) * 16 + q;
                        *o = turn == 0 ? acc[t][ct][j] : *o + acc[t][ct][j];
                    }
        }
        __syncthreads();
    }
    for (int e = tid; e < TAPS * C::COB * 16; e += 256) {
        const int t = e % TAPS, r2 = e / TAPS;
        const int ci = r2 % 16, co = r2 / 16;
        part[((long)co * L.cinp + ci) * TAPS + t] = red[((t * (CT * WM) + co / 16) * 16 + ci) * 16 + (co % 16)];
    }
}

// ---------------------------------------------------------------- weight gradient of the U-Nets' plane-wide 3x3 convs (lean)
// wgrad_mfma_kernel serves every source mode, width and tap count from one body (230 - 250 VGPRs: one or two workgroups per CU) and
// re-derives every staging slot of every tile with integer divisions; a third of its instructions per tile go into a halo-column loop
// that writes zeros whenever the tile spans the plane's width.  Same algorithm for exactly that case (W == TW, plain or
// InstanceNorm + LeakyReLU sources, at most a concat of two whose first ends on a 16-channel boundary, plain output gradient):
// the slot tables are built once per thread, the halo columns are zeroed once, the statistics table holds this workgroup's
// 16 channels instead of all of them.  Same tile order, K split and reduction order: the partial sums are BIT-IDENTICAL.
template <int TW, int CT, int WM, int NPIX>
__global__ __launch_bounds__(256, 2) void wgrad_plane_kernel(WgLaunch L) {
    using C = WgCfg<9, TW, CT, WM, NPIX>;
    constexpr int PW = C::PW, PR = C::PR, NIP = C::NIP, NGP = C::NGP;
    typedef typename WgPiece<PW>::T piece_t;
    extern __shared__ __align__(16) float smem_g[];
    float* in_lds = smem_g;
    float* g_lds = smem_g + C::IN_FLOATS;
    float* st_lds = smem_g + C::LDS_FLOATS;           // two tables (sample parity) of {scale, shift} of this workgroup's 16 input channels
    const WgArgs& a = L.a;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int q = lane & 15, kk = lane >> 4;
    const int wm = wave % WM, wk = wave / WM;
    const int ci0 = blockIdx.x * 16, co0 = blockIdx.y * C::COB;
    const int set = blockIdx.z / L.nchunks, ch = blockIdx.z - set * L.nchunks;

    f32x4 acc[9][CT];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) acc[t][ct] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const int ntile = (a.H + C::TH - 1) / C::TH;
    const int sbase = set ? a.set_split : 0;
    const int items = ((set ? a.n : a.set_split) - sbase) * ntile;
    const int ibeg = min(ch * L.chunk, items), iend = min(ibeg + L.chunk, items);
    const int total = iend - ibeg;
    const int hw = a.H * TW;
    // the source that holds my 16-channel chunk (uniform: the first source ends on a chunk boundary)
    const bool f0 = ci0 < a.s0.c;
    const float* const sx = f0 ? a.s0.x : a.s1.x;
    const float* const spart = f0 ? a.s0.part : a.s1.part;
    const int sc_ = f0 ? a.s0.c : a.s1.c, snp = f0 ? a.s0.np : a.s1.np, smode = f0 ? a.s0.mode : a.s1.mode;
    const int cl0 = f0 ? ci0 : ci0 - a.s0.c;
    const int nci = min(16, a.cin - ci0);             // channels of the chunk that exist
    // ---- slot tables (fixed for the kernel)
    int xko[NIP], xrm[NIP], xl[NIP];                  // channel + column offset in the source, row - 1 in the tile, LDS offset (-1: no slot / no channel)
#pragma unroll
    for (int i = 0; i < NIP; ++i) {
        const int e = tid + i * 256;
        const int ec = min(e, C::NIU - 1);
        const int k = ec / (C::ROWS * PR), rem = ec - k * (C::ROWS * PR);
        const int row = rem / PR, j = rem - row * PR;
        xko[i] = min(k, nci - 1) * hw + PW * j; xrm[i] = row - 1;
        xl[i] = (e < C::NIU) ? ((k * C::PSI + row * C::COLS + PW * j) << 1) | (k < nci ? 1 : 0) : -1;      // bit 0: the channel exists
    }
    int gko[NGP], grw[NGP], gl[NGP];
#pragma unroll
    for (int i = 0; i < NGP; ++i) {
        const int e = tid + i * 256;
        const int ec = min(e, C::NGU - 1);
        const int co = ec / (NPIX / PW), pp = (ec - co * (NPIX / PW)) * PW;
        gko[i] = min(co0 + co, a.rows - 1) * hw + pp % TW; grw[i] = pp / TW;
        gl[i] = (e < C::NGU) ? ((co * C::PSG + pp) << 1) | (co0 + co < a.rows ? 1 : 0) : -1;
    }
    auto table = [&](int n, float* st) {
        if (tid < 16) {
            float2 mr = make_float2(0.f, 1.f);
            if (smode == 1 && tid < nci) mr = merge_partials(spart + ((long)n * sc_ + cl0 + tid) * snp * 3, snp, a.eps);
            st[2 * tid] = mr.y; st[2 * tid + 1] = -mr.x * mr.y;
        }
    };
    piece_t xin[NIP], gin[NGP];
    auto issue = [&](int it) {
        const int n = sbase + (ibeg + it) / ntile, r0 = ((ibeg + it) % ntile) * C::TH;
        const float* xb = sx + ((long)n * sc_ + cl0) * hw;
        const float* gb = a.g + (long)n * a.rows * hw;
#pragma unroll
        for (int i = 0; i < NIP; ++i) xin[i] = *reinterpret_cast<const piece_t*>(xb + xko[i] + min(max(r0 + xrm[i], 0), a.H - 1) * TW);
#pragma unroll
        for (int i = 0; i < NGP; ++i) gin[i] = *reinterpret_cast<const piece_t*>(gb + gko[i] + min(r0 + grw[i], a.H - 1) * TW);
    };
    auto commit = [&](int it) {
        const int r0 = ((ibeg + it) % ntile) * C::TH;
        const float* st = st_lds + 32 * (((ibeg + it) / ntile) & 1);
#pragma unroll
        for (int i = 0; i < NIP; ++i) {
            if (xl[i] < 0) continue;
            const int gy = r0 + xrm[i];
            const bool ok = (xl[i] & 1) && gy >= 0 && gy < a.H;
            piece_t o = xin[i];
            float* ov = reinterpret_cast<float*>(&o);
            const int k2 = 2 * min((xl[i] >> 1) / C::PSI, 15);
            const float sc = st[k2], sh = st[k2 + 1];
#pragma unroll
            for (int u = 0; u < PW; ++u) ov[u] = ok ? (smode == 0 ? ov[u] : act(ov[u], sc, sh, a.slope)) : 0.f;
            float* dst = in_lds + (xl[i] >> 1);
#pragma unroll
            for (int u = 0; u < PW; u += 2) *reinterpret_cast<float2*>(dst + u) = make_float2(ov[u], ov[u + 1]);
        }
#pragma unroll
        for (int i = 0; i < NGP; ++i) {
            if (gl[i] < 0) continue;
            const bool ok = (gl[i] & 1) && r0 + grw[i] < a.H;
            const piece_t o = gin[i];
            const float* ov = reinterpret_cast<const float*>(&o);
            float* dst = g_lds + (gl[i] >> 1);
#pragma unroll
            for (int u = 0; u < PW; u += 2) *reinterpret_cast<float2*>(dst + u) = ok ? make_float2(ov[u], ov[u + 1]) : make_float2(0.f, 0.f);
        }
    };
    // the halo columns of a plane-wide tile are zero for every item
    for (int e = tid; e < 16 * C::ROWS * 2; e += 256) {
        const int k = e / (C::ROWS * 2), rem = e - k * (C::ROWS * 2);
        in_lds[k * C::PSI + (rem >> 1) * C::COLS + ((rem & 1) ? TW : C::COLS - 1)] = 0.f;
    }
    if (total > 0) { table(sbase + ibeg / ntile, st_lds + 32 * ((ibeg / ntile) & 1)); issue(0); }
    for (int it = 0; it < total; ++it) {
        __syncthreads();
        if ((ibeg + it + 1) % ntile == 0 && it + 1 < total)
            table(sbase + (ibeg + it + 1) / ntile, st_lds + 32 * (((ibeg + it + 1) / ntile) & 1));
        commit(it);
        __syncthreads();
        if (it + 1 < total) issue(it + 1);
        constexpr int NSTEP = NPIX / 4 / C::WK;
        float gv[2][CT], xv[2][9];
        auto load_step = [&](int i, float (&gq)[CT], float (&xq)[9]) {
            const int p = 4 * (wk + i * C::WK) + kk;
            const int prow = p / TW, pcol = p % TW;
#pragma unroll
            for (int ct = 0; ct < CT; ++ct) gq[ct] = g_lds[(16 * (wm * CT + ct) + q) * C::PSG + p];
            const float* ib = in_lds + q * C::PSI + prow * C::COLS;
#pragma unroll
            for (int t = 0; t < 9; ++t) {
                int xc = pcol + (t % 3) - 1; xc = xc < 0 ? C::COLS - 1 : xc;
                xq[t] = ib[(t / 3) * C::COLS + xc];
            }
        };
        load_step(0, gv[0], xv[0]);
#pragma unroll
        for (int i = 0; i < NSTEP; ++i) {
            if (i + 1 < NSTEP) load_step(i + 1, gv[(i + 1) & 1], xv[(i + 1) & 1]);
#pragma unroll
            for (int t = 0; t < 9; ++t)
#pragma unroll
                for (int ct = 0; ct < CT; ++ct)
                    acc[t][ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(xv[i & 1][t], gv[i & 1][ct], acc[t][ct], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    float* part = L.part + (((long)blockIdx.z * L.rowsp + co0) * L.cinp + ci0) * 9;
    if (C::WK == 1) {
#pragma unroll
        for (int t = 0; t < 9; ++t)
#pragma unroll
            for (int ct = 0; ct < CT; ++ct)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    part[((long)(16 * (wm * CT + ct) + q) * L.cinp + 4 * kk + j) * 9 + t] = acc[t][ct][j];
        return;
    }
    __syncthreads();
    float* red = smem_g;
    for (int turn = 0; turn < C::WK; ++turn) {
        if (wk == turn) {
#pragma unroll
            for (int t = 0; t < 9; ++t)
#pragma unroll
                for (int ct = 0; ct < CT; ++ct)
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        float* o = red + ((t * (CT * WM) + wm * CT + ct) * 16 + 4 * kk + j) * 16 + q;
                        *o = turn == 0 ? acc[t][ct][j] : *o + acc[t][ct][j];
                    }
        }
        __syncthreads();
    }
    for (int e = tid; e < 9 * C::COB * 16; e += 256) {
        const int t = e % 9, r2 = e / 9;
        const int ci = r2 % 16, co = r2 / 16;
        part[((long)co * L.cinp + ci) * 9 + t] = red[((t * (CT * WM) + co / 16) * 16 + ci) * 16 + (co % 16)];
    }
}

// grad += sum over the partial sums of one weight set.  A workgroup owns 64 consecutive weights (coalesced 256-byte reads of
// every partial) and splits the chunks over its 16 waves; the 16 wave sums are added in a fixed order (deterministic).
// One plane (sample n, conv-input channel cg) of the transformed conv input -- concat or sum of the two sources, any on-load mode -- written
// out as a plain tensor (launch_wgrad: sources the vectorised staging cannot read directly).
__global__ __launch_bounds__(256) void wgrad_materialise_kernel(WgArgs a, float* __restrict__ out) {
    extern __shared__ float st_m[];                      // {scale, shift} of every source channel of this sample
    const int n = blockIdx.x / a.cin, cg = blockIdx.x - n * a.cin;
    auto table_of = [&](const Src& s, float* st) {
        const bool stats = s.mode == 1 || s.mode == 2 || (s.mode >= 3 && (s.act & 1));
        for (int cl = threadIdx.x; cl < s.c; cl += 256) {
            float2 mr = make_float2(0.f, 1.f);
            if (stats) mr = merge_partials(s.part + ((long)n * s.c + cl) * s.np * 3, s.np, a.eps);
            st[2 * cl] = mr.y; st[2 * cl + 1] = -mr.x * mr.y;
        }
    };
    table_of(a.s0, st_m); table_of(a.s1, st_m + 2 * a.s0.c);
    __syncthreads();
    const int c0n = src_cin(a.s0);
    float* o = out + (long)blockIdx.x * a.H * a.W;
    for (int e = blockIdx.y * 256 + threadIdx.x; e < a.H * a.W; e += gridDim.y * 256) {
        const int gy = e / a.W, gx = e - gy * a.W;
        float v;
        if (a.add_src1) v = fetch_scalar(a.s0, n, cg, 0, gy, gx, st_m, a.slope) + fetch_scalar(a.s1, n, cg, 0, gy, gx, st_m + 2 * a.s0.c, a.slope);
        else if (cg < c0n) v = fetch_scalar(a.s0, n, cg, 0, gy, gx, st_m, a.slope);
        else v = fetch_scalar(a.s1, n, cg - c0n, 0, gy, gx, st_m + 2 * a.s0.c, a.slope);
        o[e] = v;
    }
}

// The same for the shapes the MWCNN and the U-Net down paths produce -- 2-D planes whose width is a multiple of 4 and whose wavelet /
// pooled sources have exactly twice (DWT, pool) or half (IWT) the extents -- with 16-byte stores and vector loads: one thread = four
// consecutive outputs of one row.  Same operations in the same order as fetch_scalar: bit-identical values.  (The scalar kernel
// moved 0.9 TB/s: 11.5 ms of the cfg-3 training step.)
static inline bool mat_vec_host(const Src& s, int H, int W) {
    if (s.c == 0) return true;
    if (s.d != 1 || (s.act & 2)) return false;
    auto al = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; };
    if (s.mode <= 1) return s.w == W && s.h <= H && al(s.x);
    if (s.mode == 2 || s.mode == 3) return s.w == 2 * W && s.h == 2 * H && al(s.x);
    if (s.mode == 4) return 2 * s.w == W && 2 * s.h == H && s.c % 4 == 0 && (reinterpret_cast<uintptr_t>(s.x) & 7) == 0;
    return false;
}
__device__ __forceinline__ float4 fetch_vec4(const Src& s, int n, int cl, int gy, int gx, const float* st, float slope) {
    float o[4];
    if (s.mode == 3) {
        const int band = cl / s.c, c = cl - band * s.c;
        const float* p = s.x + (((long)n * s.c + c) * s.h + 2 * gy) * s.w + 2 * gx;
        const float4 a0 = *reinterpret_cast<const float4*>(p), a1 = *reinterpret_cast<const float4*>(p + 4);
        const float4 b0 = *reinterpret_cast<const float4*>(p + s.w), b1 = *reinterpret_cast<const float4*>(p + s.w + 4);
        const float r0[8] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w}, r1[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
        const float m = st[2 * c], r = st[2 * c + 1];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            float x1 = r0[2 * u], x3 = r0[2 * u + 1], x2 = r1[2 * u], x4 = r1[2 * u + 1];
            if (s.act & 1) { x1 = act(x1, m, r, slope); x2 = act(x2, m, r, slope); x3 = act(x3, m, r, slope); x4 = act(x4, m, r, slope); }
            x1 *= 0.5f; x2 *= 0.5f; x3 *= 0.5f; x4 *= 0.5f;
            o[u] = band == 0 ? x1 + x2 + x3 + x4 : band == 1 ? -x1 - x2 + x3 + x4 : band == 2 ? -x1 + x2 - x3 + x4 : x1 - x2 - x3 + x4;
        }
    } else if (s.mode == 4) {
        const int cq = s.c / 4, sy = gy >> 1, sx = gx >> 1, ry = gy & 1;
        float v[2][4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int c = cl + k * cq;
            const float2 t = *reinterpret_cast<const float2*>(s.x + (((long)n * s.c + c) * s.h + sy) * s.w + sx);
            float t0 = t.x, t1 = t.y;
            if (s.act & 1) { t0 = act(t0, st[2 * c], st[2 * c + 1], slope); t1 = act(t1, st[2 * c], st[2 * c + 1], slope); }
            v[0][k] = 0.5f * t0; v[1][k] = 0.5f * t1;
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const float* w = v[u >> 1];
            const int rx = u & 1;
            o[u] = (!ry && !rx) ? w[0] - w[1] - w[2] + w[3] : (ry && !rx) ? w[0] - w[1] + w[2] - w[3] : (!ry && rx) ? w[0] + w[1] - w[2] - w[3]
                                                                                                                     : w[0] + w[1] + w[2] + w[3];
        }
    } else if (s.mode == 2) {
        const float* p = s.x + (((long)n * s.c + cl) * s.h + 2 * gy) * s.w + 2 * gx;
        const float4 a0 = *reinterpret_cast<const float4*>(p), a1 = *reinterpret_cast<const float4*>(p + 4);
        const float4 b0 = *reinterpret_cast<const float4*>(p + s.w), b1 = *reinterpret_cast<const float4*>(p + s.w + 4);
        const float r0[8] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w}, r1[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
        const float m = st[2 * cl], r = st[2 * cl + 1];
#pragma unroll
        for (int u = 0; u < 4; ++u)
            o[u] = 0.25f * (act(r0[2 * u], m, r, slope) + act(r0[2 * u + 1], m, r, slope) + act(r1[2 * u], m, r, slope) + act(r1[2 * u + 1], m, r, slope));
    } else {
        if (gy >= s.h) return make_float4(0.f, 0.f, 0.f, 0.f);
        const float4 t = *reinterpret_cast<const float4*>(s.x + (((long)n * s.c + cl) * s.h + gy) * s.w + gx);
        o[0] = t.x; o[1] = t.y; o[2] = t.z; o[3] = t.w;
        if (s.mode == 1) {
            const float m = st[2 * cl], r = st[2 * cl + 1];
#pragma unroll
            for (int u = 0; u < 4; ++u) o[u] = act(o[u], m, r, slope);
        }
    }
    return make_float4(o[0], o[1], o[2], o[3]);
}
__global__ __launch_bounds__(256) void wgrad_materialise_vec_kernel(WgArgs a, float* __restrict__ out) {
    extern __shared__ float st_m[];
    const int n = blockIdx.x / a.cin, cg = blockIdx.x - n * a.cin;
    auto table_of = [&](const Src& s, float* st) {
        const bool stats = s.mode == 1 || s.mode == 2 || (s.mode >= 3 && (s.act & 1));
        for (int cl = threadIdx.x; cl < s.c; cl += 256) {
            float2 mr = make_float2(0.f, 1.f);
            if (stats) mr = merge_partials(s.part + ((long)n * s.c + cl) * s.np * 3, s.np, a.eps);
            st[2 * cl] = mr.y; st[2 * cl + 1] = -mr.x * mr.y;
        }
    };
    table_of(a.s0, st_m); table_of(a.s1, st_m + 2 * a.s0.c);
    __syncthreads();
    const int c0n = src_cin(a.s0), w4 = a.W >> 2;
    float* o = out + (long)blockIdx.x * a.H * a.W;
    for (int e = blockIdx.y * 256 + threadIdx.x; e < a.H * w4; e += gridDim.y * 256) {
        const int gy = e / w4, gx = (e - gy * w4) << 2;
        float4 v;
        if (a.add_src1) {
            const float4 p = fetch_vec4(a.s0, n, cg, gy, gx, st_m, a.slope), q = fetch_vec4(a.s1, n, cg, gy, gx, st_m + 2 * a.s0.c, a.slope);
            v = make_float4(p.x + q.x, p.y + q.y, p.z + q.z, p.w + q.w);
        } else if (cg < c0n) v = fetch_vec4(a.s0, n, cg, gy, gx, st_m, a.slope);
        else v = fetch_vec4(a.s1, n, cg - c0n, gy, gx, st_m + 2 * a.s0.c, a.slope);
        *reinterpret_cast<float4*>(o + (long)gy * a.W + gx) = v;
    }
}

__global__ __launch_bounds__(1024) void wgrad_reduce_kernel(const float* part, int nchunks, int rows, int cin, int rowsp, int cinp,
                                                            int taps, int kind, float* grad0, float* grad1) {
    __shared__ float red[16][64];
    const int set = blockIdx.y;
    float* grad = set ? grad1 : grad0;
    if (!grad) return;
    const long total = (long)rows * cin * taps;
    const long pstride = (long)rowsp * cinp * taps;
    const float* p0 = part + (long)set * nchunks * pstride;
    const int lane = threadIdx.x & 63, sub = threadIdx.x >> 6;
    const long e = (long)blockIdx.x * 64 + lane;
    float s = 0.f;
    long o = 0;
    if (e < total) {
        const int t = (int)(e % taps);
        const long r2 = e / taps;
        const int ci = (int)(r2 % cin), row = (int)(r2 / cin);
        const float* p = p0 + ((long)row * cinp + ci) * taps + t;
        for (int c = sub; c < nchunks; c += 16) s += p[c * pstride];
        o = kind == 1 ? (long)ci * rows + row : e;            // transpose conv (cin, cout, 2, 2): row = 4 co + 2 a + b
    }
    red[sub][lane] = s;
    __syncthreads();
    if (sub == 0 && e < total) {
        float tsum = 0.f;
#pragma unroll
        for (int i = 0; i < 16; ++i) tsum += red[i][lane];
        grad[o] += tsum;
    }
}

// Runs of work items per launch: ONE resident round of workgroups.  The kernel is register-bound to two workgroups per CU for row blocks of
// 16 / 32 and one for 64 / 128; a run that covers ~10 tiles amortises its prologue and its K-split reduction, and every extra run is another
// partial sum to write and re-read (measured on the cfg-2 step, conv family: 2048 runs per set 32.7 ms, 512: 30.4, 256: 28.6, 128: 29.5,
// 64: 40.0).
static int wgrad_runs(int rows, int cin, int nsets) {
    const int cob = rows <= 16 ? 16 : rows <= 32 ? 32 : rows <= 64 ? 64 : 128;
    const long slots = 256L * (cob <= 32 ? 2 : 1);
    const long wgs = (long)ceil_div(cin, 16) * ceil_div(rows, cob) * nsets;
    const long per_run = (long)ceil_div(rows, cob) * cob * ceil_div(cin, 16) * 16 * 9 * 4;       // bytes of one partial (3x3)
    const long cap = std::max(8L, (32L << 20) / per_run);                                          // <= 32 MB of partials per weight set
    return (int)std::max(1L, std::min(cap, ceil_div(slots, wgs)));
}

size_t wgrad_ws_floats(int rows, int cin, int taps, int n) {
    const int rowsp = ceil_div(rows, 16) * 16, cinp = ceil_div(cin, 16) * 16;
    const int cob = rows <= 16 ? 16 : rows <= 32 ? 32 : rows <= 64 ? 64 : 128;
    const int rowsb = ceil_div(rowsp, cob) * cob;
    (void)n;
    return (size_t)std::max(wgrad_runs(rows, cin, 1), 2 * wgrad_runs(rows, cin, 2)) * rowsb * cinp * taps;
}

static thread_local int g_wgrad_plane = 1;      // the CALLING THREAD's diagnostic choice (cine_set_conv_plane bit 4), like its side stream
bool wgrad_plane_enabled() { return g_wgrad_plane != 0; }
void set_wgrad_plane(int on) { g_wgrad_plane = on; }

template <int TAPS, int TW, int CT, int WM, int NPIX>
static int launch_wg_cfg(const WgLaunch& L, dim3 grid, hipStream_t st) {
    using C = WgCfg<TAPS, TW, CT, WM, NPIX>;
    const size_t lds = (size_t)(C::LDS_FLOATS + std::max(4 * (L.a.s0.c + L.a.s1.c), 64)) * sizeof(float);     // (wgrad_plane_kernel: two tables of 16 channels)
    static_assert(C::LDS_FLOATS * sizeof(float) <= 60 * 1024, "wgrad tile exceeds the default LDS limit");
    CINE_REQUIRE(lds <= 64 * 1024, CINE_EUNSUPPORTED, "wgrad: %d source channels need %zu bytes of LDS", L.a.s0.c + L.a.s1.c, lds);
    if constexpr (TAPS == 9) {
        const WgArgs& a = L.a;
        auto same = [&](const Src& s) { return s.c == 0 || (s.mode <= 1 && s.h == a.H && s.w == a.W); };
        if (wgrad_plane_enabled() && L.fast_in && L.fast_g == 1 && a.W == TW && !a.add_src1 && same(a.s0) && same(a.s1) && (a.s1.c == 0 || a.s0.c % 16 == 0) &&
            (a.s0.mode == 0 || (a.s0.part && a.s0.np > 0)) && (a.s1.c == 0 || a.s1.mode == 0 || (a.s1.part && a.s1.np > 0))) {
            diag_count(D_WGRAD_PLANE);
            hipLaunchKernelGGL((wgrad_plane_kernel<TW, CT, WM, NPIX>), grid, dim3(256), lds, st, L);
            return check_launch("wgrad_plane_kernel");
        }
        diag_count(D_WGRAD_GENERAL);
        if (L.a.add_src1) {
            hipLaunchKernelGGL((wgrad_mfma_kernel<TAPS, TW, CT, WM, NPIX, true>), grid, dim3(256), lds, st, L);
            return check_launch("wgrad_mfma_kernel");
        }
    }
    CINE_REQUIRE(!L.a.add_src1, CINE_EUNSUPPORTED, "wgrad: summed sources only for 3x3 convs");
    hipLaunchKernelGGL((wgrad_mfma_kernel<TAPS, TW, CT, WM, NPIX, false>), grid, dim3(256), lds, st, L);
    return check_launch("wgrad_mfma_kernel");
}

template <int TAPS, int TW>
static int launch_wg_tw(const WgLaunch& L, int cob, dim3 grid, hipStream_t st) {
    // pixels per tile: as many as LDS allows for the row block (the fewer rows, the more pixels amortise the staging)
    if (cob == 16) return launch_wg_cfg<TAPS, TW, 1, 1, 256>(L, grid, st);
    if (cob == 32) return launch_wg_cfg<TAPS, TW, 1, 2, 128>(L, grid, st);
    if (cob == 64) return launch_wg_cfg<TAPS, TW, 1, 4, 128>(L, grid, st);
    return launch_wg_cfg<TAPS, TW, 2, 4, 64>(L, grid, st);
}

// the partial sums of one weight-gradient launch into `ws` (everything but the reduction); L: the launch's geometry for the reduction
static int launch_wgrad_partials(const WgArgs& a, int taps, int kind, float* ws, size_t ws_floats, hipStream_t st, WgLaunch& L);

int launch_wgrad(const WgArgs& a, int taps, int kind, float* grad0, float* grad1, float* ws, size_t ws_floats, hipStream_t st) {
    CINE_REQUIRE(a.g && a.s0.x && ws && grad0 && a.n > 0 && a.rows > 0 && a.cin > 0 && a.H > 0 && a.W > 0, CINE_EINVAL, "wgrad: bad arguments");
    CINE_REQUIRE(taps == 9 || taps == 1, CINE_EINVAL, "wgrad: taps %d", taps);
    CINE_REQUIRE(a.s0.mode <= 4 && (a.s1.c == 0 || a.s1.mode <= 4), CINE_EUNSUPPORTED, "wgrad: source modes 0..4 only");
    CINE_REQUIRE((a.add_src1 && a.s1.c > 0) ? (src_cin(a.s0) == a.cin && src_cin(a.s1) == a.cin) : (src_cin(a.s0) + src_cin(a.s1) == a.cin), CINE_EINVAL,
                 "wgrad: channel counts");
    CINE_REQUIRE(a.set_split >= 0 && a.set_split <= a.n && (a.set_split == a.n || grad1), CINE_EINVAL, "wgrad: second weight set without a gradient");
    const int TW = a.W > 8 ? 16 : a.W > 4 ? 8 : a.W > 2 ? 4 : 2;
    {   // sources the vectorised staging cannot read: materialise the transformed conv input once, then read it as a plain tensor
        const int PW0 = TW >= 4 ? 4 : 2;
        auto plain = [&](const Src& s) { return s.c == 0 || (s.mode <= 1 && s.w == a.W && s.h <= a.H); };
        const size_t elems = (size_t)a.n * a.cin * a.H * a.W;
        if (a.mat && (a.add_src1 || !plain(a.s0) || !plain(a.s1)) && a.W % PW0 == 0 && a.mat_floats >= elems && elems >= (1u << 16) &&
            (long)a.n * a.cin <= 0x7fffffffL) {
            const size_t lds = (size_t)2 * (a.s0.c + a.s1.c) * sizeof(float);
            CINE_REQUIRE(lds <= 48 * 1024, CINE_EUNSUPPORTED, "wgrad: too many source channels");
            {
                ProfScope prof(F_MISC, st);
                const int gy = std::max(1, std::min(64, (int)ceil_div((long)a.H * a.W, 2048L)));
                const bool vec = a.W % 4 == 0 && mat_vec_host(a.s0, a.H, a.W) && mat_vec_host(a.s1, a.H, a.W) &&
                                 reinterpret_cast<uintptr_t>(a.mat) % 16 == 0;
                if (vec) hipLaunchKernelGGL(wgrad_materialise_vec_kernel, dim3((unsigned)(a.n * a.cin), (unsigned)gy), dim3(256), lds, st, a, a.mat);
                else hipLaunchKernelGGL(wgrad_materialise_kernel, dim3((unsigned)(a.n * a.cin), (unsigned)gy), dim3(256), lds, st, a, a.mat);
                if (int e = check_launch("wgrad_materialise_kernel")) return e;
            }
            WgArgs b = a;
            b.s0 = Src{a.mat, nullptr, a.cin, 0, a.H, a.W, 0, 0, 1};
            b.s1 = Src{nullptr, nullptr, 0, 0, 0, 0, 0, 0, 1};
            b.add_src1 = 0; b.mat = nullptr; b.mat_floats = 0;
            return launch_wgrad(b, taps, kind, grad0, grad1, ws, ws_floats, st);
        }
    }
    WgLaunch L{};
    if (int e = launch_wgrad_partials(a, taps, kind, ws, ws_floats, st, L)) return e;
    const int nsets = a.n - a.set_split > 0 ? 2 : 1;
    const long total = (long)a.rows * a.cin * taps;
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((unsigned)ceil_div(total, 64L), nsets), dim3(1024), 0, st,
                       ws, L.nchunks, a.rows, a.cin, L.rowsp, L.cinp, taps, kind, grad0, grad1);
    return check_launch("wgrad_reduce_kernel");
}

static int launch_wgrad_partials(const WgArgs& a, int taps, int kind, float* ws, size_t ws_floats, hipStream_t st, WgLaunch& L) {
    const int TW = a.W > 8 ? 16 : a.W > 4 ? 8 : a.W > 2 ? 4 : 2;
    L = WgLaunch{};
    L.a = a;
    L.rowsp = ceil_div(a.rows, 16) * 16; L.cinp = ceil_div(a.cin, 16) * 16;
    const int cob = a.rows <= 16 ? 16 : a.rows <= 32 ? 32 : a.rows <= 64 ? 64 : 128;
    const int rowsb = ceil_div(L.rowsp, cob) * cob;
    L.rowsp = rowsb;                                      // partial rows padded to whole row blocks
    const int n0 = a.set_split, n1 = a.n - a.set_split;
    {   // work items = (sample, tile); the tile shape follows the row block (launch_wg_tw)
        const int npix = cob == 16 ? 256 : cob == 128 ? 64 : 128;
        const long items = (long)std::max(n0, n1) * ceil_div(a.W, TW) * ceil_div(a.H, npix / TW);
        L.nchunks = (int)std::min<long>(wgrad_runs(a.rows, a.cin, n1 > 0 ? 2 : 1), items);
        L.chunk = (int)ceil_div(items, (long)L.nchunks);
        L.nchunks = (int)ceil_div(items, (long)L.chunk);
    }
    L.part = ws;
    // vectorised staging: rows of whole 16 / 8-byte pieces at aligned addresses, sources addressed with the conv's row stride
    const int PW = TW >= 4 ? 4 : 2;
    auto src_fast = [&](const Src& s) {
        return s.c == 0 || (s.mode <= 1 && s.w == a.W && s.h <= a.H && reinterpret_cast<uintptr_t>(s.x) % 16 == 0);
    };
    L.fast_in = !a.add_src1 && a.W % PW == 0 && a.W >= PW && src_fast(a.s0) && src_fast(a.s1);
    L.fast_g = a.g_mode == 0 && a.W % PW == 0 && a.W >= PW && reinterpret_cast<uintptr_t>(a.g) % 16 == 0;
    // the transpose conv's space-to-depth view: whole 2 x 2 blocks (exact 2:1 extents), rows in whole row blocks of 4 c + 2 a + b
    if (a.g_mode == 5 && a.W % PW == 0 && a.W >= PW && a.g_w == 2 * a.W && a.g_h == 2 * a.H && a.rows == 4 * a.g_c && a.rows % 4 == 0 &&
        (a.rows >= cob || a.rows % 4 == 0) && reinterpret_cast<uintptr_t>(a.g) % 16 == 0)
        L.fast_g = 2;
    const int nsets = n1 > 0 ? 2 : 1;
    CINE_REQUIRE(ws_floats >= (size_t)nsets * L.nchunks * L.rowsp * L.cinp * taps, CINE_EWORKSPACE, "wgrad: workspace too small");
    const dim3 grid(L.cinp / 16, rowsb / cob, nsets * L.nchunks);
    ProfScope prof(taps == 9 ? F_CONV3 : (kind == 1 ? F_TCONV : F_CONV1), st);
    int e;
    if (taps == 9) {
        if (TW == 16) e = launch_wg_tw<9, 16>(L, cob, grid, st);
        else if (TW == 8) e = launch_wg_tw<9, 8>(L, cob, grid, st);
        else if (TW == 4) e = launch_wg_tw<9, 4>(L, cob, grid, st);
        else e = launch_wg_tw<9, 2>(L, cob, grid, st);
    } else {
        if (TW == 16) e = launch_wg_tw<1, 16>(L, cob, grid, st);
        else if (TW == 8) e = launch_wg_tw<1, 8>(L, cob, grid, st);
        else if (TW == 4) e = launch_wg_tw<1, 4>(L, cob, grid, st);
        else e = launch_wg_tw<1, 2>(L, cob, grid, st);
    }
    return e;
}

// grad (rows, cin, 3, 3, 3) += the three depth taps' partial sums (regions of `stride` floats, nchunks[kz] partials each; 0: a dead tap)
__global__ __launch_bounds__(1024) void wgrad_reduce27_kernel(const float* part, long stride, int nc0, int nc1, int nc2, int rows, int cin, int rowsp, int cinp, float* grad) {
    __shared__ float red[16][64];
    const long total = (long)rows * cin * 27;
    const long pstride = (long)rowsp * cinp * 9;
    const int lane = threadIdx.x & 63, sub = threadIdx.x >> 6;
    const long e = (long)blockIdx.x * 64 + lane;
    float s = 0.f;
    if (e < total) {
        const int t = (int)(e % 9), kz = (int)((e / 9) % 3);
        const long r2 = e / 27;
        const int ci = (int)(r2 % cin), row = (int)(r2 / cin);
        const int nchunks = kz == 0 ? nc0 : (kz == 1 ? nc1 : nc2);
        const float* p = part + kz * stride + ((long)row * cinp + ci) * 9 + t;
        for (int c = sub; c < nchunks; c += 16) s += p[c * pstride];
    }
    red[sub][lane] = s;
    __syncthreads();
    if (sub == 0 && e < total) {
        float tsum = 0.f;
#pragma unroll
        for (int i = 0; i < 16; ++i) tsum += red[i][lane];
        grad[e] += tsum;
    }
}

// The 3x3x3 weight gradient from its three depth taps: a[kz] = the 3x3 weight-gradient problem of tap kz (a[kz].n == 0: a dead tap), one set each;
// three launches of partial sums into thirds of `ws`, ONE reduction (the per-tap form paid three: 15 us each on cfg 4's level 0).
int launch_wgrad27(const WgArgs (&a)[3], float* grad, float* ws, size_t ws_floats, hipStream_t st) {
    const WgArgs* first = nullptr;
    for (const WgArgs& t : a) if (t.n > 0 && !first) first = &t;
    CINE_REQUIRE(first && grad && ws, CINE_EINVAL, "wgrad27: bad arguments");
    const size_t stride = ws_floats / 3;
    int nc[3] = {0, 0, 0}, rowsp = 0, cinp = 0;
    for (int kz = 0; kz < 3; ++kz) {
        if (a[kz].n <= 0) continue;
        CINE_REQUIRE(a[kz].g && a[kz].s0.x && a[kz].rows == first->rows && a[kz].cin == first->cin && a[kz].set_split == a[kz].n && !a[kz].mat && !a[kz].add_src1 &&
                     a[kz].H > 0 && a[kz].W > 0, CINE_EINVAL, "wgrad27: the taps are one-set problems of one layer");
        WgLaunch L{};
        if (int e = launch_wgrad_partials(a[kz], 9, 0, ws + kz * stride, stride, st, L)) return e;
        nc[kz] = L.nchunks; rowsp = L.rowsp; cinp = L.cinp;
    }
    const long total = (long)first->rows * first->cin * 27;
    hipLaunchKernelGGL(wgrad_reduce27_kernel, dim3((unsigned)ceil_div(total, 64L)), dim3(1024), 0, st, ws, (long)stride, nc[0], nc[1], nc[2],
                       first->rows, first->cin, rowsp, cinp, grad);
    return check_launch("wgrad_reduce27_kernel");
}

// ---------------------------------------------------------------- bias gradient
// one workgroup per (channel, sample): the sample's pixel sum into ws[co][n]; then one wave per (set, channel) adds the samples of
// its set in a fixed order
// (large planes with few samples -- volumes -- are cut into `chunks` pieces per plane: ws[co][n][chunk])
__global__ __launch_bounds__(256) void bias_partial_kernel(const float* g, int cout, long hw, float* ws, int n, int chunks) {
    __shared__ float red[4];
    const int co = blockIdx.x, i = blockIdx.y, k = blockIdx.z;
    const float* p = g + ((long)i * cout + co) * hw;
    const long per = (hw + chunks - 1) / chunks, e0 = k * per, e1 = e0 + per < hw ? e0 + per : hw;
    float s = 0.f;
    for (long e = e0 + threadIdx.x; e < e1; e += 256) s += p[e];
    s = wave_sum_g(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) ws[((long)co * n + i) * chunks + k] = red[0] + red[1] + red[2] + red[3];
}
__global__ __launch_bounds__(64) void bias_final_kernel(const float* ws, int n, int set_split, float* gb0, float* gb1) {
    const int co = blockIdx.x, set = blockIdx.y;
    float* gb = set ? gb1 : gb0;
    const int nb = set ? set_split : 0, ne = set ? n : set_split;
    if (!gb || ne <= nb) return;
    float s = 0.f;
    for (int i = nb + (int)threadIdx.x; i < ne; i += 64) s += ws[(long)co * n + i];
    s = wave_sum_g(s);
    if (threadIdx.x == 0) gb[co] += s;
}

int launch_bias_grad(const float* g, int n, int cout, long hw, int set_split, float* gb0, float* gb1, float* ws, size_t ws_floats, hipStream_t st) {
    CINE_REQUIRE(g && gb0 && ws && n > 0 && n <= 65535 && cout > 0 && hw > 0, CINE_EINVAL, "bias_grad: bad arguments");
    CINE_REQUIRE(ws_floats >= (size_t)n * cout, CINE_EWORKSPACE, "bias_grad: workspace too small");
    ProfScope prof(F_STATS, st);
    const long want = (long)n * cout >= 512 ? 1 : std::min(64L, ceil_div(hw, 8192L));          // few planes: spread each over workgroups
    const int chunks = (int)std::max(1L, std::min(want, (long)(ws_floats / ((size_t)n * cout))));
    hipLaunchKernelGGL(bias_partial_kernel, dim3(cout, n, chunks), dim3(256), 0, st, g, cout, hw, ws, n, chunks);
    hipLaunchKernelGGL(bias_final_kernel, dim3(cout, set_split < n ? 2 : 1), dim3(64), 0, st, ws, n * chunks, set_split * chunks, gb0, gb1);
    return check_launch("bias_grad_kernel");
}

// ---- plain convolutions with bias / ReLU epilogues (the convolutional-RNN cells, recurrent_varnet.py:153-259) -------------------------
__global__ __launch_bounds__(256) void relu_mask_kernel(float4* __restrict__ g, const float4* __restrict__ y, long n4, float* gs, const float* ys, long tail0, long n) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i < n4) {
        float4 a = g[i]; const float4 b = y[i];
        a.x = b.x > 0.f ? a.x : 0.f; a.y = b.y > 0.f ? a.y : 0.f; a.z = b.z > 0.f ? a.z : 0.f; a.w = b.w > 0.f ? a.w : 0.f;
        g[i] = a;
    }
    if (i == 0) for (long k = tail0; k < n; ++k) gs[k] = ys[k] > 0.f ? gs[k] : 0.f;
}
__global__ __launch_bounds__(256) void relu_mask_scalar_kernel(float* __restrict__ g, const float* __restrict__ y, long n) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i < n) g[i] = y[i] > 0.f ? g[i] : 0.f;
}

}  // namespace cine
using namespace cine;

// g *= (y > 0): the gradient of y = ReLU(pre) with respect to pre, from the stored output (in place)
extern "C" int cine_relu_mask(float* g, const float* y, long n, void* stream) {
    CINE_REQUIRE(g && y && n > 0, CINE_EINVAL, "cine_relu_mask: bad arguments");
    if (reinterpret_cast<uintptr_t>(g) % 16 || reinterpret_cast<uintptr_t>(y) % 16) {
        CINE_REQUIRE(ceil_div(n, 256L) <= 0x7fffffffL, CINE_EUNSUPPORTED, "cine_relu_mask: too large");
        hipLaunchKernelGGL(relu_mask_scalar_kernel, dim3((unsigned)ceil_div(n, 256L)), dim3(256), 0, as_stream(stream), g, y, n);
        return check_launch("relu_mask_scalar_kernel");
    }
    const long n4 = n / 4;
    CINE_REQUIRE(ceil_div(std::max(n4, 1L), 256L) <= 0x7fffffffL, CINE_EUNSUPPORTED, "cine_relu_mask: too large");
    hipLaunchKernelGGL(relu_mask_kernel, dim3((unsigned)ceil_div(std::max(n4, 1L), 256L)), dim3(256), 0, as_stream(stream),
                       reinterpret_cast<float4*>(g), reinterpret_cast<const float4*>(y), n4, g, y, n4 * 4, n);
    return check_launch("relu_mask_kernel");
}

extern "C" size_t cine_conv3x3_wgrad_ws_bytes(int cout, int cin, int n) {
    if (cout <= 0 || cin <= 0 || n <= 0) return 0;
    return std::max(wgrad_ws_floats(cout, cin, 9, n), (size_t)n * cout) * sizeof(float);
}

// gw (cout, c0 + c1, 3, 3) += d loss / d W of y = conv3x3(cat(x0, x1); W) from g = d loss / d y; x0 (n, c0, h, w), x1 (n, c1, h, w) or NULL
// (the summed convolutions of the CRNN cells are one convolution over concatenated inputs, recurrent_varnet.py:172-178, 122-134);
// gb (cout) += sum of g over samples and pixels when not NULL.
extern "C" int cine_conv3x3_wgrad(const float* x0, int c0, const float* x1, int c1, const float* g, float* gw, float* gb,
                                  int n, int cout, int h, int w, void* ws, size_t ws_bytes, void* stream) {
    CINE_REQUIRE(x0 && g && gw && ws && c0 > 0 && c1 >= 0 && (c1 == 0 || x1) && n > 0 && cout > 0 && h > 0 && w > 0, CINE_EINVAL,
                 "cine_conv3x3_wgrad: bad arguments");
    CINE_REQUIRE(ws_bytes >= cine_conv3x3_wgrad_ws_bytes(cout, c0 + c1, n), CINE_EWORKSPACE, "cine_conv3x3_wgrad: workspace too small");
    hipStream_t st = as_stream(stream);
    WgArgs a{};
    a.s0 = Src{x0, nullptr, c0, 0, h, w, 0, 0, 1};
    a.s1 = c1 ? Src{x1, nullptr, c1, 0, h, w, 0, 0, 1} : Src{nullptr, nullptr, 0, 0, 0, 0, 0, 0, 1};
    a.cin = c0 + c1; a.g = g; a.g_mode = 0; a.rows = cout; a.n = n; a.H = h; a.W = w; a.set_split = n; a.eps = 1e-5f; a.slope = 0.2f;
    float* wsf = reinterpret_cast<float*>(ws);
    const size_t wsn = ws_bytes / sizeof(float);
    if (gb) if (int e = launch_bias_grad(g, n, cout, (long)h * w, n, gb, nullptr, wsf, wsn, st)) return e;
    return launch_wgrad(a, 9, 0, gw, nullptr, wsf, wsn, st);
}

// gw (cout, cin) += the weight gradient of a 1x1 (x1) convolution, gb += its bias gradient when not NULL; x (n, cin, h, w), g (n, cout, h, w).
// Volumes: pass (d h, w) -- a 1x1x1 convolution does not see the shape.
extern "C" size_t cine_conv1x1_wgrad_ws_bytes(int cout, int cin, int n) {
    if (cout <= 0 || cin <= 0 || n <= 0) return 0;
    return std::max(wgrad_ws_floats(cout, cin, 1, n), (size_t)n * cout) * sizeof(float);
}
extern "C" int cine_conv1x1_wgrad(const float* x, int cin, const float* g, float* gw, float* gb, int n, int cout, int h, int w,
                                  void* ws, size_t ws_bytes, void* stream) {
    CINE_REQUIRE(x && g && gw && ws && cin > 0 && n > 0 && cout > 0 && h > 0 && w > 0, CINE_EINVAL, "cine_conv1x1_wgrad: bad arguments");
    CINE_REQUIRE(ws_bytes >= cine_conv1x1_wgrad_ws_bytes(cout, cin, n), CINE_EWORKSPACE, "cine_conv1x1_wgrad: workspace too small");
    hipStream_t st = as_stream(stream);
    WgArgs a{};
    a.s0 = Src{x, nullptr, cin, 0, h, w, 0, 0, 1};
    a.s1 = Src{nullptr, nullptr, 0, 0, 0, 0, 0, 0, 1};
    a.cin = cin; a.g = g; a.g_mode = 0; a.rows = cout; a.n = n; a.H = h; a.W = w; a.set_split = n; a.eps = 1e-5f; a.slope = 0.2f;
    float* wsf = reinterpret_cast<float*>(ws);
    const size_t wsn = ws_bytes / sizeof(float);
    if (gb) if (int e = launch_bias_grad(g, n, cout, (long)h * w, n, gb, nullptr, wsf, wsn, st)) return e;
    return launch_wgrad(a, 1, 2, gw, nullptr, wsf, wsn, st);
}

// gr = d loss / d raw from g = d loss / d act(raw), act = LeakyReLU(InstanceNorm(raw)) of the planes (n, c) of h * w elements with the
// statistics records part (n, c, np, 3) (unet.py:159-168); g has the tensor's own shape.  Volumes: pass (d h, w).
extern "C" size_t cine_in_lrelu_bwd_ws_bytes(int n, int c, int h, int w) {
    return (n <= 0 || c <= 0 || h <= 0 || w <= 0) ? 0 : in_lrelu_bwd_ws_floats(n, c, h, w) * sizeof(float);
}
extern "C" int cine_in_lrelu_bwd(const float* r, const float* part, int np, const float* g, float* gr, int n, int c, int h, int w,
                                 float eps, float slope, void* ws, size_t ws_bytes, void* stream) {
    InBwdArgs a{r, part, np, GradPiece{g, 1, c, 0, h, w}, GradPiece{nullptr, 0, 0, 0, 0, 0}, gr, n, c, h, w, eps, slope};
    return launch_in_lrelu_bwd_split(a, reinterpret_cast<float*>(ws), ws_bytes / sizeof(float), as_stream(stream));
}

namespace cine {
// ---- SideLane -------------------------------------------------------------------------------------------------------------------------
namespace {
thread_local hipStream_t g_side_tls = nullptr;      // the caller's second stream (cine_set_side_stream); never created by the library
}  // namespace
void set_side_stream(hipStream_t s) { g_side_tls = s; }

SideLane::SideLane(hipStream_t main) : main_(main) {
    side_ = g_side_tls == main ? nullptr : g_side_tls;
    if (!side_) return;
    bool ok = hipEventCreateWithFlags(&ready_, hipEventDisableTiming) == hipSuccess;
    for (int i = 0; i < 2 && ok; ++i) ok = hipEventCreateWithFlags(&done_[i], hipEventDisableTiming) == hipSuccess;
    if (!ok) side_ = nullptr;       // everything on the caller's stream
}
SideLane::~SideLane() {
    join();
    if (ready_) (void)hipEventDestroy(ready_);
    for (auto e : done_) if (e) (void)hipEventDestroy(e);
}
void SideLane::before_write() {
    if (side_ && rec_[slot()]) { (void)hipStreamWaitEvent(main_, done_[slot()], 0); rec_[slot()] = false; }
}
hipStream_t SideLane::fork() {
    if (!side_) return main_;
    (void)hipEventRecord(ready_, main_);
    (void)hipStreamWaitEvent(side_, ready_, 0);
    return side_;
}
void SideLane::launched() {
    if (side_) { (void)hipEventRecord(done_[slot()], side_); rec_[slot()] = true; }
    ++k_;
}
void SideLane::join() {
    if (!side_) return;
    for (int i = 0; i < 2; ++i)
        if (rec_[i]) { (void)hipStreamWaitEvent(main_, done_[i], 0); rec_[i] = false; }
}

}  // namespace cine

// The backward entry points that overlap weight gradients with the input-gradient chain (cine_unet2d_backward, cine_mwcnn_backward)
// put the weight-gradient launches on this stream of the CALLING THREAD; NULL (the default) keeps everything on `stream`.
extern "C" int cine_set_side_stream(void* side_stream) {
    cine::set_side_stream(reinterpret_cast<hipStream_t>(side_stream));
    return CINE_OK;
}
