// fft_kernels.hip -- batched centered ortho FFTs fused with the coil operators.
//
// A 2-D centered FFT is two line passes over LDS tiles (fft_core.h):
//   column pass: one workgroup = one image x LINES adjacent columns, transform along h
//   row pass   : one workgroup = LINES rows, transform along w (contiguous)
// The reference's ifftshift / fftshift copies (fftc.py:119-213) become index
// rotations on the tile load / store; the ortho scale rides on the twiddles.
// Fusions (reference lines in include/cine_hip.h):
//   row pass load  : S * img                (sens_expand, varnet.py:181-185)
//   col pass store : soft / hard DC blend   (varnet.py:281-282, cinenet.py:129)
//   row pass store : conj(S) * x, coil sum, optional magnitude (varnet.py:187-194, 150-151)
// N == 200 uses the 10 x 20 Cooley-Tukey engine; any other N = 2^a 3^b 5^c <= 512 the mixed-radix Stockham engine
// (radices 4 / 2 / 3 / 5, two tiles ping-pong); lengths with another prime factor, <= 400, a direct DFT.
#include <algorithm>
#include <map>
#include <mutex>
#include <utility>
#include "common.h"
#include "fft_core.h"

namespace cine {

__device__ const float2 TW200[200] = {
#include "tw200.inc"
};

constexpr int kLines200 = 32, kThreads200 = 320;   // 640 r10 items = 2 rounds, 320 r20 items = 1 round
constexpr int kLinesGen = 8, kThreadsGen = 256;
constexpr int kMaxGenericN = 400;                  // direct DFT, O(n^2): any length
constexpr int kMaxSmoothN = 512;                   // mixed radix: 2^a 3^b 5^c (two 512 x 9 tiles + twiddles = 78 KB of LDS, opted in per kernel)
constexpr int kMaxOut = 4;                          // reduce outputs per thread

enum { POST_NONE = 0, POST_DC = 1, POST_HARD = 2, POST_RESID = 3 };   // RESID: mask ? k - kref : 0 (xpdnet.py:128-131,295-298)
enum { PRE_NONE = 0, PRE_SMUL = 1 };
enum { RPOST_NONE = 0, RPOST_REDUCE = 1, RPOST_REDUCE_ABS = 2, RPOST_RSS = 3 };   // RSS: sqrt(sum_c |x_c|^2) (coil_combine.py:21-34)

template <bool F200> __device__ __forceinline__ void load_twiddles(cf* tw, int n) {
    if (F200) {
        for (int j = threadIdx.x; j < 200; j += blockDim.x) tw[j] = TW200[j];
    } else {
        // the mixed-radix engine scales in its last stage (unit twiddles keep the butterflies exact rotations); the direct one here
        const double s = MixedRadix::smooth(n) ? 1.0 : 1.0 / sqrt((double)n);
        for (int j = threadIdx.x; j < n; j += blockDim.x) {
            double sn, cs;
            sincospi(2.0 * (double)j / (double)n, &sn, &cs);
            tw[j] = mk((float)(cs * s), (float)(-sn * s));
        }
    }
}

// Transform every line of the tile along p; returns the tile that holds the result
// (natural order for the direct and mixed-radix engines -- t0 or t1 by the parity of the stage count --, Fft200::pos_of order for
// the 200 engine).  The other tile is free afterwards.
template <bool F200, int DIR, int LINES>
__device__ __forceinline__ cf* run_lines(cf* t0, cf* t1, int n, const cf* tw) {
    constexpr int LP = LINES + 1;
    const int tid = threadIdx.x, nt = blockDim.x;
    if (F200) {
        for (int i = tid; i < Fft200::items_r10(LINES); i += nt)
            Fft200::stage_r10<DIR, false, true>(t0, LP, i, LINES, tw);
        __syncthreads();
        for (int i = tid; i < Fft200::items_r20(LINES); i += nt) Fft200::stage_r20<DIR>(t0, LP, i, LINES);
        __syncthreads();
        return t0;
    } else if (MixedRadix::smooth(n)) {           // uniform over the workgroup
        int radix[MixedRadix::kMaxStages];
        const int ns = MixedRadix::plan(n, radix);
        const float scale = 1.0f / sqrtf((float)n);
        cf* src = t0; cf* dst = t1;
        int Ns = 1;
        for (int s = 0; s < ns; ++s) {
            const int R = radix[s];
            const float sc = s == ns - 1 ? scale : 1.f;
            for (int i = tid; i < MixedRadix::items(LINES, n, R); i += nt) MixedRadix::stage<DIR>(src, dst, LP, i, LINES, n, R, Ns, tw, sc);
            __syncthreads();
            Ns *= R;
            cf* x = src; src = dst; dst = x;
        }
        return src;
    } else {
        for (int i = tid; i < DirectDft::items(LINES, n); i += nt)
            DirectDft::stage<DIR>(t0, t1, LP, i, LINES, n, tw);
        __syncthreads();
        return t1;
    }
}
template <bool F200> __device__ __forceinline__ int res_pos(int k) { return F200 ? Fft200::pos_of(k) : k; }

__device__ __forceinline__ float softplus1(float x) { return x > 20.f ? x : log1pf(expf(x)); }

// ------------------------------------------------------------------ column pass
struct ColArgs {
    const cf* in; cf* out;
    int H, W;
    int s_in, s_out;
    const cf* kref; const uint8_t* mask; const float* lam;
    int coils;            // images per mask row-set (mask index = img / coils)
    const uint8_t* premask;   // optional: rows with premask == 0 enter the transform as zeros and are not read
};

template <bool F200, int DIR, int POST, int LINES>
__global__ void col_pass_kernel(ColArgs a) {
    constexpr int LP = LINES + 1;
    extern __shared__ __align__(16) unsigned char smem[];
    const int H = F200 ? 200 : a.H;
    cf* t0 = reinterpret_cast<cf*>(smem);
    cf* t1 = t0 + (F200 ? 0 : H * LP);
    cf* tw = t1 + H * LP;
    const int tid = threadIdx.x, nt = blockDim.x;
    const int w0 = blockIdx.x * LINES;
    const long img = blockIdx.y;
    const cf* in = a.in + img * H * a.W;
    cf* out = a.out + img * H * a.W;

    load_twiddles<F200>(tw, H);
    for (int e = tid; e < H * LINES; e += nt) {
        const int g = e / LINES, l = e % LINES, col = w0 + l;
        cf v = mk(0.f, 0.f);
        if (col < a.W && (!a.premask || a.premask[(img / a.coils) * H + g])) v = in[(long)g * a.W + col];
        int n = g + a.s_in; if (n >= H) n -= H;
        t0[n * LP + l] = v;
    }
    __syncthreads();
    cf* res = run_lines<F200, DIR, LINES>(t0, t1, H, tw);

    float v = 0.f, inv1v = 1.f;
    if (POST == POST_DC) { v = softplus1(*a.lam); }
    const uint8_t* mrow = (POST != POST_NONE) ? a.mask + (img / a.coils) * H : nullptr;
    const cf* kref = (POST == POST_DC || POST == POST_RESID) ? a.kref + img * H * a.W : nullptr;
    for (int e = tid; e < H * LINES; e += nt) {
        const int i = e / LINES, l = e % LINES, col = w0 + l;
        if (col >= a.W) continue;
        int k = i - a.s_out; if (k < 0) k += H;
        cf val = res[res_pos<F200>(k) * LP + l];
        if (POST == POST_DC) {
            if (mrow[i]) {
                cf r = kref[(long)i * a.W + col];
                val = mk((val.x + v * r.x) / (1.f + v), (val.y + v * r.y) / (1.f + v));
            }
        } else if (POST == POST_HARD) {
            if (!mrow[i]) val = mk(0.f, 0.f);
        } else if (POST == POST_RESID) {
            val = mrow[i] ? csub(val, kref[(long)i * a.W + col]) : mk(0.f, 0.f);
        }
        out[(long)i * a.W + col] = val;
    }
    (void)inv1v;
}

// ------------------------------------------------------------------ row pass
struct RowArgs {
    const cf* in; cf* out; float* out_abs;
    long nlines;          // plain mode: total lines
    int W;
    int s_in, s_out;
    // coil modes
    const cf* sens; const cf* img;
    int T, C, H, rpw, cc;  // rows per workgroup, coils per chunk
};

template <bool F200, int DIR, int PRE, int POST, int LINES>
__global__ void row_pass_kernel(RowArgs a) {
    constexpr int LP = LINES + 1;
    extern __shared__ __align__(16) unsigned char smem[];
    const int W = F200 ? 200 : a.W;
    cf* t0 = reinterpret_cast<cf*>(smem);
    cf* t1 = t0 + (F200 ? 0 : W * LP);
    cf* tw = t1 + W * LP;
    const int tid = threadIdx.x, nt = blockDim.x;
    load_twiddles<F200>(tw, W);

    if (PRE == PRE_NONE && POST == RPOST_NONE) {
        // ---- plain: LINES consecutive lines of W points
        const long L0 = (long)blockIdx.x * LINES;
        for (int e = tid; e < LINES * W; e += nt) {
            const int l = e / W, g = e - l * W;
            cf v = mk(0.f, 0.f);
            if (L0 + l < a.nlines) v = a.in[(L0 + l) * W + g];
            int n = g + a.s_in; if (n >= W) n -= W;
            t0[n * LP + l] = v;
        }
        __syncthreads();
        cf* res = run_lines<F200, DIR, LINES>(t0, t1, W, tw);
        for (int e = tid; e < LINES * W; e += nt) {
            const int l = e / W, i = e - l * W;
            if (L0 + l >= a.nlines) continue;
            int k = i - a.s_out; if (k < 0) k += W;
            a.out[(L0 + l) * W + i] = res[res_pos<F200>(k) * LP + l];
        }
        return;
    }

    // ---- coil modes: lines = (coil within chunk, row within group)
    const int bt = blockIdx.y, b = bt / a.T;
    const int h0 = blockIdx.x * a.rpw;
    const long HW = (long)a.H * W;
    cf acc[kMaxOut];
#pragma unroll
    for (int o = 0; o < kMaxOut; ++o) acc[o] = mk(0.f, 0.f);

    for (int c0 = 0; c0 < a.C; c0 += a.cc) {
        const int nc = min(a.cc, a.C - c0);
        // load
        for (int e = tid; e < LINES * W; e += nt) {
            const int l = e / W, g = e - l * W;
            const int cl = l / a.rpw, r = l - cl * a.rpw;
            const int h = h0 + r;
            cf v = mk(0.f, 0.f);
            if (cl < nc && h < a.H) {
                const int c = c0 + cl;
                if (PRE == PRE_SMUL) {
                    cf s = a.sens[((long)b * a.C + c) * HW + (long)h * W + g];
                    cf x = a.img[(long)bt * HW + (long)h * W + g];
                    v = cmul(x, s);
                } else {
                    v = a.in[((long)bt * a.C + c) * HW + (long)h * W + g];
                }
            }
            int n = g + a.s_in; if (n >= W) n -= W;
            t0[n * LP + l] = v;
        }
        __syncthreads();
        cf* res = run_lines<F200, DIR, LINES>(t0, t1, W, tw);
        if (POST == RPOST_NONE) {
            for (int e = tid; e < LINES * W; e += nt) {
                const int l = e / W, i = e - l * W;
                const int cl = l / a.rpw, r = l - cl * a.rpw;
                const int h = h0 + r;
                if (cl >= nc || h >= a.H) continue;
                int k = i - a.s_out; if (k < 0) k += W;
                a.out[((long)bt * a.C + c0 + cl) * HW + (long)h * W + i] = res[res_pos<F200>(k) * LP + l];
            }
        } else {
#pragma unroll
            for (int o = 0; o < kMaxOut; ++o) {
                const int e = tid + o * nt;
                if (e >= a.rpw * W) break;
                const int r = e / W, i = e - r * W;
                const int h = h0 + r;
                if (h >= a.H) continue;
                int k = i - a.s_out; if (k < 0) k += W;
                const int p = res_pos<F200>(k) * LP;
                cf s_acc = acc[o];
                for (int cl = 0; cl < nc; ++cl) {
                    cf x = res[p + cl * a.rpw + r];
                    if (POST == RPOST_RSS) { s_acc.x += x.x * x.x + x.y * x.y; continue; }
                    cf s = a.sens[((long)b * a.C + c0 + cl) * HW + (long)h * W + i];
                    cf m = cmulc(x, s);       // x * conj(s)
                    s_acc.x += m.x; s_acc.y += m.y;
                }
                acc[o] = s_acc;
            }
        }
        __syncthreads();
    }
    if (POST != RPOST_NONE) {
#pragma unroll
        for (int o = 0; o < kMaxOut; ++o) {
            const int e = tid + o * nt;
            if (e >= a.rpw * W) break;
            const int r = e / W, i = e - r * W;
            const int h = h0 + r;
            if (h >= a.H) continue;
            if (POST == RPOST_REDUCE_ABS)
                a.out_abs[(long)bt * HW + (long)h * W + i] = sqrtf(acc[o].x * acc[o].x + acc[o].y * acc[o].y);
            else if (POST == RPOST_RSS)
                a.out_abs[(long)bt * HW + (long)h * W + i] = sqrtf(acc[o].x);
            else
                a.out[(long)bt * HW + (long)h * W + i] = acc[o];
        }
    }
}

// ================================================================== N = 200 fast passes
// First butterfly stage straight from global memory into registers, last stage straight back:
// ONE LDS exchange per transform (the generic kernels above need three passes over the tile).
// 32 lines per workgroup, 320 threads: 640 radix-10 items (2 per thread), 320 radix-20 items.
constexpr int kFL = 16;                 // lines per workgroup of the fast passes (small tiles -> 5-6 workgroups per CU)
constexpr int kFT = 10 * kFL;           // threads: one radix-20 item each, two radix-10 items each
constexpr int kLP200c = kFL;            // column passes: lanes run over lines, contiguous LDS rows
constexpr int kLP200r = kFL + 1;        // row passes: lanes run over points, odd stride spreads banks

__device__ __forceinline__ int wrap200(int x) { return x >= 200 ? x - 200 : (x < 0 ? x + 200 : x); }
// (20 j + c +- 100) mod 200 for 0 <= c < 20 and (10 k + g +- 100) mod 200 for 0 <= g < 10: with j / k a compile-time
// constant of an unrolled loop the wrap is decided at compile time (no compare + select per access)
__device__ __forceinline__ int rot20(int j, int c) { return 20 * j + c + (j < 5 ? 100 : -100); }
__device__ __forceinline__ int rot10(int k, int g) { return 10 * k + g + (k < 10 ? 100 : -100); }

// Column pass along h (H == 200).  DIR forward/inverse, optional DC on the centered rows, and with
// INV_AFTER the inverse transform of the blended column right away (k-space -> DC -> hybrid space
// without the k-space ever leaving the CU):
//   global -> r10 -> LDS -> r20 [-> DC -> r20^-1 -> LDS -> r10^-1] -> global
template <int DIR, int POST, bool INV_AFTER, bool PREMASK = false>
__global__ __launch_bounds__(kFT, 3) void col200_kernel(ColArgs a) {
    constexpr int LP = kLP200c;
    extern __shared__ __align__(16) unsigned char smem[];
    cf* t = reinterpret_cast<cf*>(smem);
    const int tid = threadIdx.x;
    const int w0 = blockIdx.x * kFL;
    const long img = blockIdx.y;
    const cf* in = a.in + img * 200 * a.W;
    cf* out = a.out + img * 200 * a.W;
    CINE_STAMP(0);
    // DC operands of the rows this thread will own after the radix-20 stage (k = g + 10 k2, centered
    // row (k + 100) % 200).  Fetched FIRST so their latency overlaps the stage-1 stream instead of
    // adding two dependent round trips behind the LDS exchange.  Unsampled rows read kref[0]
    // (one cached line): no branches, no HBM traffic for the rows the mask drops.
    unsigned mbits = 0;
    cf rr[20];
    if (POST != POST_NONE) {
        const int g2 = tid / kFL;
        const uint8_t* mrow = a.mask + (img / a.coils) * 200;
#pragma unroll
        for (int k2 = 0; k2 < 20; ++k2) mbits |= (mrow[rot10(k2, g2)] ? 1u : 0u) << k2;
        if (POST == POST_DC || POST == POST_RESID) {
            const cf* kref = a.kref + img * 200 * a.W;
            const int colc = min(w0 + tid % kFL, a.W - 1);
#pragma unroll
            for (int k2 = 0; k2 < 20; ++k2) {
                const int off = ((mbits >> k2) & 1u) ? rot10(k2, g2) * a.W + colc : 0;
                rr[k2] = kref[off];
            }
        }
    }
#pragma unroll 1
    for (int r = 0; r < 2; ++r) {
        const int item = tid + r * kFT;
        const int line = item % kFL, c = item / kFL;
        const int col = w0 + line;
        const int colc = min(col, a.W - 1);                     // clamped: lanes past the edge load a valid
        cf v[10];                                               // column and are never stored (no branches)
        if (PREMASK) {
            // rows the mask drops enter as zeros and are not fetched (they re-read one cached line instead: no branches)
            const uint8_t* pm = a.premask + (img / a.coils) * 200;
            bool keep[10];
#pragma unroll
            for (int j = 0; j < 10; ++j) keep[j] = pm[rot20(j, c)] != 0;
#pragma unroll
            for (int j = 0; j < 10; ++j) v[j] = in[keep[j] ? rot20(j, c) * a.W + colc : 0];
#pragma unroll
            for (int j = 0; j < 10; ++j) v[j] = keep[j] ? v[j] : mk(0.f, 0.f);
        } else {
#pragma unroll
            for (int j = 0; j < 10; ++j) v[j] = in[rot20(j, c) * a.W + colc];   // x'[n] = x[(n - 100) mod N]
        }
        Fft200::r10_regs<DIR, false, true>(v, c, TW200);
#pragma unroll
        for (int j = 0; j < 10; ++j) t[(20 * j + c) * LP + line] = v[j];
        CINE_STAMP(1 + r);
    }
    __syncthreads();
    CINE_STAMP(3);
    {
        const int line = tid % kFL, g = tid / kFL;
        const int col = w0 + line;
        cf v[20];
#pragma unroll
        for (int j = 0; j < 20; ++j) v[j] = t[(20 * g + j) * LP + line];
        dft20<DIR>(v);                                          // v[k2] = X'[g + 10 k2] -> centered row (k + 100) % 200
        CINE_STAMP(4);
        if (POST == POST_DC) {
            const float vv = softplus1(*a.lam), inv = 1.0f / (1.f + vv);
#pragma unroll
            for (int k2 = 0; k2 < 20; ++k2) {
                const cf b = mk((v[k2].x + vv * rr[k2].x) * inv, (v[k2].y + vv * rr[k2].y) * inv);
                v[k2] = ((mbits >> k2) & 1u) ? b : v[k2];
            }
        } else if (POST == POST_HARD) {
#pragma unroll
            for (int k2 = 0; k2 < 20; ++k2) v[k2] = ((mbits >> k2) & 1u) ? v[k2] : mk(0.f, 0.f);
        } else if (POST == POST_RESID) {
#pragma unroll
            for (int k2 = 0; k2 < 20; ++k2) v[k2] = ((mbits >> k2) & 1u) ? csub(v[k2], rr[k2]) : mk(0.f, 0.f);
        }
        if (!INV_AFTER) {
            if (col < a.W) {
#pragma unroll
                for (int k2 = 0; k2 < 20; ++k2) out[rot10(k2, g) * a.W + col] = v[k2];
            }
            return;
        }
        CINE_STAMP(5);
        dft20<-DIR>(v);                                         // PN flavour, stage A on the same registers
#pragma unroll
        for (int j = 0; j < 20; ++j) t[(20 * g + j) * LP + line] = v[j];
        CINE_STAMP(6);
    }
    __syncthreads();
    CINE_STAMP(7);
#pragma unroll 1
    for (int r = 0; r < 2; ++r) {
        const int item = tid + r * kFT;
        const int line = item % kFL, c = item / kFL;
        const int col = w0 + line;
        cf v[10];
#pragma unroll
        for (int j = 0; j < 10; ++j) v[j] = t[(20 * j + c) * LP + line];
        Fft200::r10_regs<-DIR, true, false>(v, c, TW200);
        if (col < a.W) {
#pragma unroll
            for (int j = 0; j < 10; ++j) out[rot20(j, c) * a.W + col] = v[j];
        }
        CINE_STAMP(8 + r);
    }
}

// Row pass along w (W == 200), inverse, fused with conj(S) multiply + coil sum (+ magnitude):
//   global -> r10^-1 -> LDS -> r20^-1 -> LDS -> sum_c conj(S) x -> global
template <int POST>
__global__ __launch_bounds__(kFT, 3) void row200_reduce_kernel(RowArgs a) {
    constexpr int LP = kLP200r;
    extern __shared__ __align__(16) unsigned char smem[];
    cf* t = reinterpret_cast<cf*>(smem);
    const int tid = threadIdx.x;
    const int bt = blockIdx.y, b = bt / a.T;
    const int h0 = blockIdx.x * a.rpw;
    const long HW = (long)a.H * 200;
    cf acc[kMaxOut];
#pragma unroll
    for (int o = 0; o < kMaxOut; ++o) acc[o] = mk(0.f, 0.f);
    for (int c0 = 0; c0 < a.C; c0 += a.cc) {
        const int nc = min(a.cc, a.C - c0);
        // sensitivities of this thread's first output pixel, fetched up front (up to kSPre coils)
        constexpr int kSPre = 16;
        cf spre[kSPre];
        if (POST != RPOST_RSS) {
            const int rr0 = tid / 200, i0 = tid - rr0 * 200;
            const cf* sb = a.sens + ((long)b * a.C + c0) * HW + (long)min(h0 + rr0, a.H - 1) * 200 + i0;
#pragma unroll
            for (int u = 0; u < kSPre; ++u) spre[u] = sb[(long)min(u, nc - 1) * HW];
        }
#pragma unroll 1
        for (int r = 0; r < 2; ++r) {
            const int item = tid + r * kFT;
            const int c = item % 20, line = item / 20;
            const int cl = line / a.rpw, rr = line - cl * a.rpw;
            const int h = h0 + rr;
            // lines past the coil chunk / image edge read a clamped (valid) row; their results are never used
            const cf* src = a.in + ((long)bt * a.C + c0 + min(cl, nc - 1)) * HW + (long)min(h, a.H - 1) * 200;
            cf v[10];
#pragma unroll
            for (int j = 0; j < 10; ++j) v[j] = src[rot20(j, c)];
            Fft200::r10_regs<-1, false, true>(v, c, TW200);
#pragma unroll
            for (int j = 0; j < 10; ++j) t[(20 * j + c) * LP + line] = v[j];
        }
        __syncthreads();
        {
            const int line = tid % kFL, g = tid / kFL;
            cf v[20];
#pragma unroll
            for (int j = 0; j < 20; ++j) v[j] = t[(20 * g + j) * LP + line];
            dft20<-1>(v);
#pragma unroll
            for (int j = 0; j < 20; ++j) t[(20 * g + j) * LP + line] = v[j];     // pos 20 g + k2 <-> k = g + 10 k2
        }
        __syncthreads();
#pragma unroll
        for (int o = 0; o < kMaxOut; ++o) {
            const int e = tid + o * kFT;
            if (e >= a.rpw * 200) break;
            const int rr = e / 200, i = e - rr * 200;
            const int h = h0 + rr;
            if (h >= a.H) continue;
            const int k = wrap200(i - 100);
            const int p = Fft200::pos_of(k) * LP;
            cf s_acc = acc[o];
            if (POST == RPOST_RSS) {
                for (int cl = 0; cl < nc; ++cl) { const cf x = t[p + cl * a.rpw + rr]; s_acc.x += x.x * x.x + x.y * x.y; }
                acc[o] = s_acc;
                continue;
            }
            const cf* sbase = a.sens + ((long)b * a.C + c0) * HW + (long)h * 200 + i;
            int cg0 = 0;
            if (o == 0) {                                   // first output: sensitivities already in registers
#pragma unroll
                for (int u = 0; u < kSPre; ++u) {
                    const cf m = cmulc(t[p + min(u, nc - 1) * a.rpw + rr], spre[u]);
                    if (u < nc) { s_acc.x += m.x; s_acc.y += m.y; }
                }
                cg0 = kSPre;
            }
            for (int cg = cg0; cg < nc; cg += 5) {
                cf sv[5], xv[5];
#pragma unroll
                for (int u = 0; u < 5; ++u) {
                    const int cl = min(cg + u, nc - 1);
                    sv[u] = sbase[(long)cl * HW];
                    xv[u] = t[p + cl * a.rpw + rr];
                }
#pragma unroll
                for (int u = 0; u < 5; ++u) {
                    const cf m = cmulc(xv[u], sv[u]);
                    if (cg + u < nc) { s_acc.x += m.x; s_acc.y += m.y; }
                }
            }
            acc[o] = s_acc;
        }
        __syncthreads();
    }
#pragma unroll
    for (int o = 0; o < kMaxOut; ++o) {
        const int e = tid + o * kFT;
        if (e >= a.rpw * 200) break;
        const int rr = e / 200, i = e - rr * 200;
        const int h = h0 + rr;
        if (h >= a.H) continue;
        if (POST == RPOST_REDUCE_ABS)
            a.out_abs[(long)bt * HW + (long)h * 200 + i] = sqrtf(acc[o].x * acc[o].x + acc[o].y * acc[o].y);
        else if (POST == RPOST_RSS)
            a.out_abs[(long)bt * HW + (long)h * 200 + i] = sqrtf(acc[o].x);
        else
            a.out[(long)bt * HW + (long)h * 200 + i] = acc[o];
    }
}

// Row pass along w (W == 200), forward, fused with the sensitivity multiply:
//   S x (L2-resident reads) -> r20 -> LDS -> r10 -> global, natural order out (160-byte runs)
__global__ __launch_bounds__(kFT, 3) void row200_expand_kernel(RowArgs a) {
    constexpr int LP = kLP200r;
    extern __shared__ __align__(16) unsigned char smem[];
    cf* t = reinterpret_cast<cf*>(smem);
    const int tid = threadIdx.x;
    const int bt = blockIdx.y, b = bt / a.T;
    const int h0 = blockIdx.x * a.rpw;
    const long HW = (long)a.H * 200;
    for (int c0 = 0; c0 < a.C; c0 += a.cc) {
        const int nc = min(a.cc, a.C - c0);
        {
            const int g = tid % 10, line = tid / 10;
            const int cl = line / a.rpw, rr = line - cl * a.rpw;
            const int h = h0 + rr;
            const int hc = min(h, a.H - 1);                     // clamped: out-of-range lines are never stored
            const cf* sp = a.sens + ((long)b * a.C + c0 + min(cl, nc - 1)) * HW + (long)hc * 200;
            const cf* xp = a.img + (long)bt * HW + (long)hc * 200;
            cf xs[20], ss[20], v[20];
#pragma unroll
            for (int j = 0; j < 20; ++j) {
                const int gi = rot10(j, g);
                xs[j] = xp[gi]; ss[j] = sp[gi];
            }
#pragma unroll
            for (int j = 0; j < 20; ++j) v[j] = cmul(xs[j], ss[j]);
            dft20<1>(v);
#pragma unroll
            for (int j = 0; j < 20; ++j) t[(20 * g + j) * LP + line] = v[j];
        }
        __syncthreads();
#pragma unroll 1
        for (int r = 0; r < 2; ++r) {
            const int item = tid + r * kFT;
            const int c = item % 20, line = item / 20;
            const int cl = line / a.rpw, rr = line - cl * a.rpw;
            const int h = h0 + rr;
            cf v[10];
#pragma unroll
            for (int j = 0; j < 10; ++j) v[j] = t[(20 * j + c) * LP + line];
            Fft200::r10_regs<1, true, false>(v, c, TW200);
            if (cl < nc && h < a.H) {
                cf* dst = a.out + ((long)bt * a.C + c0 + cl) * HW + (long)h * 200;
#pragma unroll
                for (int j = 0; j < 10; ++j) dst[rot20(j, c)] = v[j];
            }
        }
        __syncthreads();
    }
}

// ================================================================== image-space data consistency
// The reference's cascade step  x -> sens_expand -> FFT2 -> DC -> (next cascade) IFFT2 -> sens_reduce  (varnet.py:181-194,
// 253, 281-282) with a Cartesian ROW mask (b, t, 1, h, 1, 1) never needs the transform along w: the mask and the blend
// weights depend on the k-space row only, so they commute with the row FFT and
//     IFFT2( wgt(ky) * FFT2(y) ) = IFFT_h( wgt(ky) * FFT_h(y) )          (all transforms centered, ortho)
// Hence, with y_c = S_c x,  v = softplus(lambda),  zf = sum_c conj(S_c) IFFT2(mask * k_ref)  (constant over the cascades):
//     x_next = sum_c conj(S_c) IFFT_h[ (mask ? 1/(1+v) : 1) * FFT_h(S_c x) ]  +  v/(1+v) * zf
// which is exactly sens_reduce(DC(sens_expand(x))) of the reference -- and one kernel that reads x, S and zf (14.4 MB at
// cfg 2) instead of three passes over the 72 MB coil-wise k-space.  The same operator with weights (1, 0) and beta 0 is
// CineNet's normal operator A^H M A (cinenet.py:121-133, 255-257); with (1, 0), beta -1 it is XPDNet's backward image
// A^H M (A x - k_ref) (xpdnet.py:128-131, 161-167, 295-298).
struct ImgDcArgs {
    const cf* img;          // (b, t, h, w)
    const cf* sens;         // (b, c, h, w)
    const cf* sens_t;       // optional (H == 200): the same maps in column-tile-major order [b][c][ceil(w / 5)][200][5] (cine_sens_tile_pack):
                            // a workgroup's 200 rows of one coil are ONE contiguous 8 KB run instead of 40 bytes out of every 128-byte line
    const cf* zf;           // (b, t, h, w) or null
    const uint8_t* mask;    // (b, t, h)
    const float* lam;       // device scalar (soft DC weights) or null (w1 / w0 / beta below)
    int lam_beta;           // lam != null: 0 = soft DC weights from softplus(*lam); 1 = w1 / w0 as given, beta = softplus(*lam)
    float w1, w0, beta;     // weight of sampled / unsampled rows, factor of zf
    cf* out; float* out_abs;
    int T, C, H, W;
    cf* partial; long part_stride;   // H == 200 with more than one coil group: per-group partial sums (workspace)
    float* pd_part;                  // optional: 256 partial sums of <img, out> (the p.d of a conjugate-gradient step, cinenet.py:155), one per workgroup of imgdc_sum_kernel
    int BT, ntx, nz;                 // H == 200: frames x batch, column tiles, coil groups
    float* pd_wg;                    // optional (H == 200, nz > 1): one partial sum of <img, sum_z partial_z + beta img> per WORKGROUP of imgdc200_kernel
                                     // (cine_normal_op_cg_fused: the conjugate-gradient step then needs no imgdc_sum pass)
    // imgdc200_kernel<true> (cine_conj_grad): the operator's input is the NEW conjugate-gradient direction p = r + (sum num / sum den) p_old
    // (cinenet.py:165-169), formed on load from the interleaved {p_old, r} pairs the update kernel left; coil group 0 writes it to cg_p_out
    const float4* cg_pr; const float* cg_num; const float* cg_den; cf* cg_p_out;
};

__device__ __forceinline__ void imgdc_weights(const ImgDcArgs& a, float& w1, float& w0, float& beta) {
    w1 = a.w1; w0 = a.w0; beta = a.beta;
    if (a.lam) {                                 // varnet.py:281-282: (1 - m) K + m (K + v K_ref) / (1 + v)
        const float v = softplus1(*a.lam);
        if (a.lam_beta) beta = v;                // cinenet.py:133: A^H M A x + v x (zf = x)
        else { w1 = 1.0f / (1.f + v); w0 = 1.f; beta = v * w1; }
    }
}

// H == 200.  The arithmetic is small (0.6 GFLOP per launch at cfg 2); the kernel is laid out for occupancy and parallelism:
// one workgroup = one frame x 5 adjacent columns x ONE group of 5 coils = 25 lines, 10 threads per line = 250 of 256
// threads: four full waves (one per SIMD; the 5-wave shapes of 32 lines packed 1-2 workgroups per CU), one 40 KB LDS tile,
// no coil loop.  Coil groups are separate workgroups (blockIdx.z) whose partial sums a small second
// kernel adds in a fixed order (deterministic: no atomics).  Measured alternatives at cfg 2: all coils in one workgroup
// over 2 columns (16-byte segments, every one drags a 128-byte line from L2: 102 us), a coil loop inside the workgroup
// (375 workgroups: 77 us), 32 lines x 4 coils in 5 waves (50 us); this shape 41 us + 7 us for the partial sums.  Ablation
// of this shape: global loads 20 us, radix-10 math 8 us, radix-20 math 3 us, the rest LDS traffic, barriers and stores.
//   P1  S x -> r10 -> tile                                  (two radix-10 items per thread)
//   P2  tile -> r20 -> row weights -> r20^-1 -> tile        (one radix-20 item per thread)
//   P3  tile -> r10^-1 -> conj(S) -> tile (in place)
//   P4  sum over the coil slots -> partial[z] (or, with a single group, + beta * zf -> out)
#ifndef CINE_DC_MINW
#define CINE_DC_MINW 3
#endif
constexpr int kDcCS = 5, kDcCW = 5, kDcL = kDcCS * kDcCW, kDcT = 256, kDcAct = 10 * kDcL;
__device__ __forceinline__ void imgdc_store(const ImgDcArgs& a, long o, cf s, float beta) {
    if (a.zf) { const cf z = a.zf[o]; s.x = fmaf(beta, z.x, s.x); s.y = fmaf(beta, z.y, s.y); }
    if (a.out_abs) a.out_abs[o] = sqrtf(s.x * s.x + s.y * s.y);
    else a.out[o] = s;
}
template <bool CG>
__global__ __launch_bounds__(kDcT, CINE_DC_MINW) void imgdc200_kernel(ImgDcArgs a) {
    constexpr int CS = kDcCS, CW = kDcCW, LP = kDcL;
    constexpr int NOUT = (200 * CW + kDcT - 1) / kDcT;         // outputs per thread
    extern __shared__ __align__(16) unsigned char smem[];
    cf* t = reinterpret_cast<cf*>(smem);
    const int tid = threadIdx.x;
    const bool active = tid < kDcAct;
    // XCD-aware decode of a 1-D grid.  Workgroup ids go round-robin over the 8 XCDs, and the sensitivities a workgroup
    // reads depend on (column tile, coil group) but not on the frame: all frames of one (tile, group) pair are given to
    // ONE XCD, so every XCD's L2 holds 1/8 of the maps instead of each of them streaming all 4.8 MB (twice).
    const int xcd = blockIdx.x & 7, kq = blockIdx.x >> 3;
    const int pair = (kq / a.BT) * 8 + xcd, bt = kq % a.BT;
    if (pair >= a.ntx * a.nz) {                                 // padding of the pair count to a multiple of 8
        if (a.pd_wg && tid == 0) a.pd_wg[blockIdx.x] = 0.f;
        return;
    }
    const int zg = pair / a.ntx;
    const int w0c = (pair - zg * a.ntx) * CW;
    const int b = bt / a.T;
    const int c0 = zg * CS;
    const long HW = 200L * a.W;
    float w1, w0, beta;
    imgdc_weights(a, w1, w0, beta);
    // P2 geometry: mask bits of the rows k = g + 10 k2 -> centered row rot10(k2, g)
    const int line2 = tid % kDcL, g2 = min(tid / kDcL, 9);
    unsigned mbits = 0;
    {
        const uint8_t* mrow = a.mask + (long)bt * 200;
#pragma unroll
        for (int k2 = 0; k2 < 20; ++k2) mbits |= (mrow[rot10(k2, g2)] ? 1u : 0u) << k2;
    }
    const cf* xp = CG ? nullptr : a.img + (long)bt * HW;
    float bcg = 0.f;                 // CG: beta = r.r (new) / r.r (old), both from the update kernels' 256 partial sums (block_sum's order: what cg_direction_kernel computes)
    if constexpr (CG) {
        __shared__ float cgred[8];
        float vn = a.cg_num[tid], vd = a.cg_den[tid];
#pragma unroll
        for (int o2 = 32; o2 > 0; o2 >>= 1) { vn += __shfl_xor(vn, o2, 64); vd += __shfl_xor(vd, o2, 64); }
        if ((tid & 63) == 0) { cgred[tid >> 6] = vn; cgred[4 + (tid >> 6)] = vd; }
        __syncthreads();
        float sn = 0.f, sd = 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i) { sn += cgred[i]; sd += cgred[4 + i]; }
        bcg = sn / sd;
    }
    const float4* prp = CG ? a.cg_pr + (long)bt * HW : nullptr;
    CINE_STAMP(0);
    cf svk[2][10];          // the sensitivities stay in registers for P3: 140 VGPRs, three workgroups per CU (48 vs 56 us re-reading them)
    // ---- P1: the loads of BOTH items first (40 per thread in flight: one memory latency instead of two), then the two radix-10 items
    if (active) {
        cf v2[2][10];
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            const int item = tid + r * kDcAct;
            const int line = item % kDcL, c = item / kDcL;                  // c = 0..19: rows rot20(j, c)
            const int slot = line / CW;
            const int colc = min(w0c + line % CW, a.W - 1);                 // clamped: lanes past the edge are never stored
            const cf* sp = a.sens_t ? a.sens_t + (((long)b * a.C + min(c0 + slot, a.C - 1)) * a.ntx + w0c / CW) * (200L * CW) + line % CW
                                    : a.sens + ((long)b * a.C + min(c0 + slot, a.C - 1)) * HW + colc;
            const int sstr = a.sens_t ? CW : a.W;                           // (the tiled copy is zero past the last column)
            if constexpr (CG) {
                const float4* pq = prp + colc;
#pragma unroll
                for (int j = 0; j < 10; ++j) {
                    svk[r][j] = sp[rot20(j, c) * sstr];
                    const float4 u = pq[rot20(j, c) * a.W];                 // {p_old, r}
                    v2[r][j] = mk(u.z + bcg * u.x, u.w + bcg * u.y);        // p = r + beta p_old (the expression of cg_direction_kernel)
                }
            } else {
                const cf* xq = xp + colc;
#pragma unroll
                for (int j = 0; j < 10; ++j) { svk[r][j] = sp[rot20(j, c) * sstr]; v2[r][j] = xq[rot20(j, c) * a.W]; }
            }
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            const int item = tid + r * kDcAct;
            const int line = item % kDcL, c = item / kDcL;
            cf v[10];
#pragma unroll
            for (int j = 0; j < 10; ++j) v[j] = cmul(v2[r][j], svk[r][j]);
            Fft200::r10_regs<1, false, true>(v, c, TW200);
#pragma unroll
            for (int j = 0; j < 10; ++j) t[(20 * j + c) * LP + line] = v[j];
            CINE_STAMP(1 + r);
        }
    }
    __syncthreads();
    CINE_STAMP(3);
    // ---- P2
    if (active) {
        cf v[20];
#pragma unroll
        for (int j = 0; j < 20; ++j) v[j] = t[(20 * g2 + j) * LP + line2];
        dft20<1>(v);                                            // v[k2] = X'[g + 10 k2] -> centered row (k + 100) % 200
#pragma unroll
        for (int k2 = 0; k2 < 20; ++k2) v[k2] = cscale(v[k2], ((mbits >> k2) & 1u) ? w1 : w0);
        dft20<-1>(v);
#pragma unroll
        for (int j = 0; j < 20; ++j) t[(20 * g2 + j) * LP + line2] = v[j];
    }
    CINE_STAMP(4);
    __syncthreads();
    CINE_STAMP(5);
    // ---- P3
    if (active) {
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            const int item = tid + r * kDcAct;
            const int line = item % kDcL, c = item / kDcL;
            const int slot = line / CW;
            const bool live = c0 + slot < a.C;
            cf v[10], sv[10];
#pragma unroll
            for (int j = 0; j < 10; ++j) sv[j] = svk[r][j];
#pragma unroll
            for (int j = 0; j < 10; ++j) v[j] = t[(20 * j + c) * LP + line];
            Fft200::r10_regs<-1, true, false>(v, c, TW200);
#pragma unroll
            for (int j = 0; j < 10; ++j) {
                const cf m = cmulc(v[j], sv[j]);
                t[(20 * j + c) * LP + line] = live ? m : mk(0.f, 0.f);
            }
            CINE_STAMP(6 + r);
        }
    }
    __syncthreads();
    CINE_STAMP(8);
    // ---- P4: this thread's outputs, summed over the coil slots
    const bool single = a.nz == 1;
    cf* part = a.partial + (long)zg * a.part_stride;
    float pdl = 0.f;
#pragma unroll
    for (int k = 0; k < NOUT; ++k) {
        const int e = tid + k * kDcT;
        const int row = e / CW, cl = e - row * CW, col = w0c + cl;
        if (e >= 200 * CW || col >= a.W) continue;
        const int pos = wrap200(row - 100);                     // row = rot20(j, c) <-> position 20 j + c
        cf s = t[pos * LP + cl];
#pragma unroll
        for (int sl = 1; sl < CS; ++sl) { const cf u = t[pos * LP + sl * CW + cl]; s.x += u.x; s.y += u.y; }
        const long o = (long)bt * HW + (long)row * a.W + col;
        if (single) imgdc_store(a, o, s, beta);
        else part[o] = s;
        if (a.pd_wg) {                  // <p, this group's share of H p>; the regulariser term beta <p, p> rides with group 0
            cf pv;
            if constexpr (CG) {
                const float4 u = a.cg_pr[o];
                pv = mk(u.z + bcg * u.x, u.w + bcg * u.y);
                if (zg == 0) a.cg_p_out[o] = pv;                            // the new direction, for the update kernel that follows
            } else pv = a.img[o];
            pdl += pv.x * s.x + pv.y * s.y;
            if (zg == 0) pdl += beta * (pv.x * pv.x + pv.y * pv.y);
        }
    }
    if (a.pd_wg) {                      // uniform
        __shared__ float pdred[4];
#pragma unroll
        for (int o2 = 32; o2 > 0; o2 >>= 1) pdl += __shfl_xor(pdl, o2, 64);
        if ((tid & 63) == 0) pdred[tid >> 6] = pdl;
        __syncthreads();
        if (tid == 0) a.pd_wg[blockIdx.x] = pdred[0] + pdred[1] + pdred[2] + pdred[3];
    }
    CINE_STAMP(9);
}

// out = sum_z partial[z] (fixed order) + beta * zf
__global__ __launch_bounds__(256) void imgdc_sum_kernel(ImgDcArgs a, int nz, long n) {
    __shared__ float red[4];
    float w1, w0, beta;
    imgdc_weights(a, w1, w0, beta);
    float pd = 0.f;
    for (long o = (long)blockIdx.x * blockDim.x + threadIdx.x; o < n; o += (long)gridDim.x * blockDim.x) {
        cf s = a.partial[o];
        for (int z = 1; z < nz; ++z) { const cf u = a.partial[z * a.part_stride + o]; s.x += u.x; s.y += u.y; }
        if (a.pd_part) {        // the value imgdc_store writes (complex output only), times the operator's input
            cf v = s;
            if (a.zf) { const cf zz = a.zf[o]; v.x = fmaf(beta, zz.x, v.x); v.y = fmaf(beta, zz.y, v.y); }
            const cf p = a.img[o];
            pd += p.x * v.x + p.y * v.y;
        }
        imgdc_store(a, o, s, beta);
    }
    if (a.pd_part) {            // uniform
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) pd += __shfl_xor(pd, o, 64);
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = pd;
        __syncthreads();
        if (threadIdx.x == 0) a.pd_part[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
    }
}

// Any other supported H: mixed-radix or direct DFT in LDS, kLinesGen columns per workgroup, one coil at a time.
constexpr int kDcAcc = (kMaxSmoothN * kLinesGen + kThreadsGen - 1) / kThreadsGen;
__global__ __launch_bounds__(kThreadsGen) void imgdc_generic_kernel(ImgDcArgs a) {
    constexpr int LINES = kLinesGen, LP = LINES + 1;
    extern __shared__ __align__(16) unsigned char smem[];
    const int H = a.H;
    cf* t0 = reinterpret_cast<cf*>(smem);
    cf* t1 = t0 + H * LP;
    cf* tw = t1 + H * LP;
    const int tid = threadIdx.x, nt = blockDim.x;
    const int w0c = blockIdx.x * LINES;
    const int bt = blockIdx.y, b = bt / a.T;
    const long HW = (long)H * a.W;
    float w1, w0, beta;
    imgdc_weights(a, w1, w0, beta);
    load_twiddles<false>(tw, H);
    const uint8_t* mrow = a.mask + (long)bt * H;
    const int s_in = (H + 1) / 2, s_out = H / 2;               // ifftshift before, fftshift after (fftc.py:75-81)
    const cf* xp = a.img + (long)bt * HW;
    cf acc[kDcAcc];
#pragma unroll
    for (int m = 0; m < kDcAcc; ++m) acc[m] = mk(0.f, 0.f);
    for (int coil = 0; coil < a.C; ++coil) {
        const cf* sp = a.sens + ((long)b * a.C + coil) * HW;
        __syncthreads();
        for (int e = tid; e < H * LINES; e += nt) {
            const int g = e / LINES, l = e - g * LINES, col = w0c + l;
            cf v = mk(0.f, 0.f);
            if (col < a.W) v = cmul(xp[(long)g * a.W + col], sp[(long)g * a.W + col]);
            int n = g + s_in; if (n >= H) n -= H;
            t0[n * LP + l] = v;
        }
        __syncthreads();
        cf* res = run_lines<false, 1, LINES>(t0, t1, H, tw);
        cf* other = res == t0 ? t1 : t0;
        for (int e = tid; e < H * LINES; e += nt) {
            const int i = e / LINES, l = e - i * LINES;
            int k = i - s_out; if (k < 0) k += H;
            int n = i + s_in; if (n >= H) n -= H;
            other[n * LP + l] = cscale(res[k * LP + l], mrow[i] ? w1 : w0);
        }
        __syncthreads();
        const cf* res2 = run_lines<false, -1, LINES>(other, res, H, tw);
#pragma unroll
        for (int m = 0; m < kDcAcc; ++m) {
            const int e = tid + m * nt;
            if (e >= H * LINES) break;
            const int i = e / LINES, l = e - i * LINES, col = w0c + l;
            if (col >= a.W) continue;
            int k = i - s_out; if (k < 0) k += H;
            const cf mm = cmulc(res2[k * LP + l], sp[(long)i * a.W + col]);
            acc[m].x += mm.x; acc[m].y += mm.y;
        }
    }
#pragma unroll
    for (int m = 0; m < kDcAcc; ++m) {
        const int e = tid + m * nt;
        if (e >= H * LINES) break;
        const int i = e / LINES, l = e - i * LINES, col = w0c + l;
        if (col >= a.W) continue;
        const long o = (long)bt * HW + (long)i * a.W + col;
        cf s = acc[m];
        if (a.zf) { const cf z = a.zf[o]; s.x = fmaf(beta, z.x, s.x); s.y = fmaf(beta, z.y, s.y); }
        if (a.out_abs) a.out_abs[o] = sqrtf(s.x * s.x + s.y * s.y);
        else a.out[o] = s;
    }
}

// ------------------------------------------------------------------ image-space DC, gradient with respect to the maps
// out = sum_c conj(S_c) T(S_c m) with T = IFFT_h W FFT_h (Hermitian).  For the output gradient g (real-pair convention,
// d loss = Re(conj(g) d out)) the maps receive, per frame,
//     gS_c = conj(g) T(S_c m)  +  T(S_c g) conj(m)
// (first term: the conj(S_c) factor; second: the S_c factor inside T).  One workgroup = one (frame, coil) x kLinesGen columns:
// two transform chains through LDS, result written per frame into `part` (b, t, c, h, w); cine_coil_accum adds the frames.
struct DcGradArgs {
    const cf* m; const cf* g; const cf* sens; const uint8_t* mask; const float* lam; float w1, w0;
    cf* part; int T, C, H, W;
};
template <bool F200>
__global__ __launch_bounds__(kThreadsGen) void imgdc_sgrad_kernel(DcGradArgs a) {
    constexpr int LINES = kLinesGen, LP = LINES + 1;
    extern __shared__ __align__(16) unsigned char smem[];
    const int H = F200 ? 200 : a.H;
    cf* t0 = reinterpret_cast<cf*>(smem);
    cf* t1 = t0 + H * LP;
    cf* tw = t1 + H * LP;
    const int tid = threadIdx.x, nt = blockDim.x;
    const int w0c = blockIdx.x * LINES, coil = blockIdx.y;
    const int bt = blockIdx.z, b = bt / a.T;
    const long HW = (long)H * a.W;
    float w1 = a.w1, w0 = a.w0;
    if (a.lam) { const float v = softplus1(*a.lam); w1 = 1.0f / (1.f + v); w0 = 1.f; }
    load_twiddles<F200>(tw, H);
    const uint8_t* mrow = a.mask + (long)bt * H;
    const int s_in = (H + 1) / 2, s_out = H / 2;
    const cf* sp = a.sens + ((long)b * a.C + coil) * HW;
    cf acc[kDcAcc];
#pragma unroll
    for (int q = 0; q < kDcAcc; ++q) acc[q] = mk(0.f, 0.f);
    for (int pass = 0; pass < 2; ++pass) {
        const cf* src = (pass ? a.g : a.m) + (long)bt * HW;
        const cf* oth = (pass ? a.m : a.g) + (long)bt * HW;
        __syncthreads();
        for (int e = tid; e < H * LINES; e += nt) {
            const int gg = e / LINES, l = e - gg * LINES, col = w0c + l;
            cf v = mk(0.f, 0.f);
            if (col < a.W) v = cmul(src[(long)gg * a.W + col], sp[(long)gg * a.W + col]);
            int n = gg + s_in; if (n >= H) n -= H;
            t0[n * LP + l] = v;
        }
        __syncthreads();
        cf* res = run_lines<F200, 1, LINES>(t0, t1, H, tw);
        cf* other = res == t0 ? t1 : t0;
        for (int e = tid; e < H * LINES; e += nt) {
            const int i = e / LINES, l = e - i * LINES;
            int k = i - s_out; if (k < 0) k += H;
            int n = i + s_in; if (n >= H) n -= H;
            other[n * LP + l] = cscale(res[res_pos<F200>(k) * LP + l], mrow[i] ? w1 : w0);
        }
        __syncthreads();
        cf* res2 = run_lines<F200, -1, LINES>(other, res, H, tw);
#pragma unroll
        for (int q = 0; q < kDcAcc; ++q) {
            const int e = tid + q * nt;
            if (e >= H * LINES) break;
            const int i = e / LINES, l = e - i * LINES, col = w0c + l;
            if (col >= a.W) continue;
            int k = i - s_out; if (k < 0) k += H;
            const cf u = cmulc(res2[res_pos<F200>(k) * LP + l], oth[(long)i * a.W + col]);    // pass 0: T(S m) conj(g); pass 1: T(S g) conj(m)
            acc[q].x += u.x; acc[q].y += u.y;
        }
    }
#pragma unroll
    for (int q = 0; q < kDcAcc; ++q) {
        const int e = tid + q * nt;
        if (e >= H * LINES) break;
        const int i = e / LINES, l = e - i * LINES, col = w0c + l;
        if (col >= a.W) continue;
        a.part[((long)bt * a.C + coil) * HW + (long)i * a.W + col] = acc[q];
    }
}

// ------------------------------------------------------------------ host side
static size_t lds_bytes(bool f200, int n, int lines) {
    const size_t tile = (size_t)n * (lines + 1) * sizeof(cf);
    return (f200 ? tile : 2 * tile) + (size_t)n * sizeof(cf);
}

static int check_n(int n, const char* what) {
    CINE_REQUIRE(n >= 1, CINE_EINVAL, "%s: length %d < 1", what, n);
    CINE_REQUIRE(n <= kMaxGenericN || (n <= kMaxSmoothN && MixedRadix::smooth(n)), CINE_EUNSUPPORTED,
                 "%s: FFT length %d unsupported (2^a 3^b 5^c up to %d via the mixed-radix engine, any length up to %d via the direct one)",
                 what, n, kMaxSmoothN, kMaxGenericN);
    return CINE_OK;
}

// lengths above 400 need more than the 64 KB of dynamic LDS a kernel gets by default: raise the limit once per (kernel, device)
// and keep the result, so that a refused request fails every later launch with its own message.  Keyed by the kernel's ADDRESS: the
// instantiations of one kernel template share a function-pointer type.
static int allow_lds(const void* kern, size_t lds, const char* what) {
    if (lds <= 64 * 1024) return CINE_OK;
    static std::mutex mu;
    static std::map<std::pair<const void*, int>, hipError_t> done;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) dev = 0;
    hipError_t st;
    {
        std::lock_guard<std::mutex> g(mu);
        auto it = done.find({kern, dev});
        if (it == done.end())
            it = done.emplace(std::make_pair(kern, dev), hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)).first;
        st = it->second;
    }
    CINE_REQUIRE(st == hipSuccess, CINE_EHIP, "%s: hipFuncSetAttribute(MaxDynamicSharedMemorySize): %s", what, hipGetErrorString(st));
    return CINE_OK;
}
template <typename K> static int allow_lds(K kern, size_t lds, const char* what) { return allow_lds(reinterpret_cast<const void*>(kern), lds, what); }

template <int POST, bool INV_AFTER = false, bool PREMASK = false>
static int launch_col(const ColArgs& a, long nimg, bool inverse, hipStream_t st) {
    if (nimg == 0) return CINE_OK;
    const bool f200 = a.H == 200;
    const int lines = f200 ? kFL : kLinesGen;
    dim3 grid(ceil_div(a.W, lines), (unsigned)nimg);
    CINE_REQUIRE(nimg <= 65535, CINE_EUNSUPPORTED, "column pass: %ld images > 65535", nimg);
    ProfScope prof(F_FFT_COL, st);
    if (f200) {
        const size_t lds = (size_t)200 * kLP200c * sizeof(cf);
        if (inverse) hipLaunchKernelGGL((col200_kernel<-1, POST, INV_AFTER, PREMASK>), grid, dim3(kFT), lds, st, a);
        else hipLaunchKernelGGL((col200_kernel<1, POST, INV_AFTER, PREMASK>), grid, dim3(kFT), lds, st, a);
    } else {
        static_assert(!INV_AFTER || true, "");
        const size_t lds = lds_bytes(false, a.H, lines);
        if (inverse) {
            if (int e = allow_lds(col_pass_kernel<false, -1, POST, kLinesGen>, lds, "col_pass_kernel")) return e;
            hipLaunchKernelGGL((col_pass_kernel<false, -1, POST, kLinesGen>), grid, dim3(kThreadsGen), lds, st, a);
        } else {
            if (int e = allow_lds(col_pass_kernel<false, 1, POST, kLinesGen>, lds, "col_pass_kernel")) return e;
            hipLaunchKernelGGL((col_pass_kernel<false, 1, POST, kLinesGen>), grid, dim3(kThreadsGen), lds, st, a);
        }
    }
    return check_launch("col_pass_kernel");
}

template <int PRE, int POST>
static int launch_row(RowArgs a, dim3 grid, bool inverse, hipStream_t st) {
    if (grid.x == 0 || grid.y == 0) return CINE_OK;
    const bool f200 = a.W == 200;
    const int lines = f200 ? kLines200 : kLinesGen;
    const size_t lds = lds_bytes(f200, a.W, lines);
    ProfScope prof(F_FFT_ROW, st);
    if (f200 && PRE == PRE_SMUL && POST == RPOST_NONE && !inverse) {
        hipLaunchKernelGGL(row200_expand_kernel, grid, dim3(kFT), (size_t)200 * kLP200r * sizeof(cf), st, a);
    } else if (f200 && PRE == PRE_NONE && POST != RPOST_NONE && inverse) {
        hipLaunchKernelGGL((row200_reduce_kernel<POST>), grid, dim3(kFT), (size_t)200 * kLP200r * sizeof(cf), st, a);
    } else if (f200) {
        if (inverse) hipLaunchKernelGGL((row_pass_kernel<true, -1, PRE, POST, kLines200>), grid, dim3(kThreads200), lds, st, a);
        else hipLaunchKernelGGL((row_pass_kernel<true, 1, PRE, POST, kLines200>), grid, dim3(kThreads200), lds, st, a);
    } else {
        if (inverse) {
            if (int e = allow_lds(row_pass_kernel<false, -1, PRE, POST, kLinesGen>, lds, "row_pass_kernel")) return e;
            hipLaunchKernelGGL((row_pass_kernel<false, -1, PRE, POST, kLinesGen>), grid, dim3(kThreadsGen), lds, st, a);
        } else {
            if (int e = allow_lds(row_pass_kernel<false, 1, PRE, POST, kLinesGen>, lds, "row_pass_kernel")) return e;
            hipLaunchKernelGGL((row_pass_kernel<false, 1, PRE, POST, kLinesGen>), grid, dim3(kThreadsGen), lds, st, a);
        }
    }
    return check_launch("row_pass_kernel");
}

// rows per workgroup / coils per chunk for the coil-mode row pass
static void coil_tiling(int C, int W, int& rpw, int& cc) {
    const bool f200 = W == 200;
    const int lines = f200 ? kFL : kLinesGen;
    const int nt = f200 ? kFT : kThreadsGen;
    cc = C < lines ? C : lines;
    rpw = lines / cc;
    while (rpw > 1 && (long)rpw * W > (long)kMaxOut * nt) --rpw;
}

int plain_rows(const cf* in, cf* out, long nlines, int n, bool inverse, int s_in, int s_out, hipStream_t st) {
    RowArgs r{};
    r.in = in; r.out = out; r.nlines = nlines; r.W = n; r.s_in = s_in; r.s_out = s_out;
    const int lines = n == 200 ? kLines200 : kLinesGen;
    const long blocks = ceil_div(nlines, (long)lines);
    CINE_REQUIRE(blocks <= 0x7fffffffL, CINE_EUNSUPPORTED, "row pass: too many lines");
    return launch_row<PRE_NONE, RPOST_NONE>(r, dim3((unsigned)blocks, 1), inverse, st);
}

}  // namespace cine

using namespace cine;

extern "C" int cine_fft2c(const float* in, float* out, int nimg, int h, int w, int inverse, void* stream) {
    CINE_REQUIRE(in && out, CINE_EINVAL, "cine_fft2c: null pointer");
    CINE_REQUIRE(nimg >= 0 && h > 0 && w > 0, CINE_EINVAL, "cine_fft2c: bad sizes nimg=%d h=%d w=%d", nimg, h, w);
    if (int e = check_n(h, "cine_fft2c(h)")) return e;
    if (int e = check_n(w, "cine_fft2c(w)")) return e;
    hipStream_t st = as_stream(stream);
    // the 65535 grid.y limit: split the image batch
    for (long i0 = 0; i0 < nimg; i0 += 32768) {
        const long ni = (nimg - i0) < 32768 ? (nimg - i0) : 32768;
        ColArgs c{};
        c.in = reinterpret_cast<const cf*>(in) + i0 * h * w;
        c.out = reinterpret_cast<cf*>(out) + i0 * h * w;
        c.H = h; c.W = w; c.s_in = (h + 1) / 2; c.s_out = h / 2; c.coils = 1;
        if (int e = launch_col<POST_NONE>(c, ni, inverse != 0, st)) return e;
    }
    return plain_rows(reinterpret_cast<cf*>(out), reinterpret_cast<cf*>(out), (long)nimg * h, w, inverse != 0,
                      (w + 1) / 2, w / 2, st);
}

extern "C" int cine_fft1c(const float* in, float* out, long nlines, int n, int inverse, int variant, void* stream) {
    CINE_REQUIRE(in && out, CINE_EINVAL, "cine_fft1c: null pointer");
    CINE_REQUIRE(nlines >= 0 && n > 0, CINE_EINVAL, "cine_fft1c: bad sizes");
    CINE_REQUIRE(variant == 0 || variant == 1, CINE_EINVAL, "cine_fft1c: variant %d", variant);
    if (int e = check_n(n, "cine_fft1c")) return e;
    // variant 1, forward = ifftshift(fft(fftshift(x))) (xpdnet.py:466): pre-roll n/2, post-roll (n+1)/2.
    // variant 1, inverse = fftshift(ifft(ifftshift(x))) (xpdnet.py:500) has the fftc.py shift order.
    const bool swapped = variant == 1 && !inverse;
    const int s_in = swapped ? n / 2 : (n + 1) / 2;
    const int s_out = swapped ? (n + 1) / 2 : n / 2;
    return plain_rows(reinterpret_cast<const cf*>(in), reinterpret_cast<cf*>(out), nlines, n, inverse != 0, s_in, s_out,
                      as_stream(stream));
}

// hybrid space = image along h, k-space along w: what a centered column IFFT of k-space gives.
extern "C" int cine_kspace_to_hybrid(const float* k, float* hyb, long nimg, int h, int w, void* stream) {
    CINE_REQUIRE(k && hyb, CINE_EINVAL, "cine_kspace_to_hybrid: null pointer");
    CINE_REQUIRE(nimg >= 0 && h > 0 && w > 0, CINE_EINVAL, "cine_kspace_to_hybrid: bad sizes");
    if (int e = check_n(h, "cine_kspace_to_hybrid(h)")) return e;
    hipStream_t st = as_stream(stream);
    for (long i0 = 0; i0 < nimg; i0 += 32768) {
        const long ni = (nimg - i0) < 32768 ? (nimg - i0) : 32768;
        ColArgs ca{};
        ca.in = reinterpret_cast<const cf*>(k) + i0 * h * w;
        ca.out = reinterpret_cast<cf*>(hyb) + i0 * h * w;
        ca.H = h; ca.W = w; ca.s_in = (h + 1) / 2; ca.s_out = h / 2; ca.coils = 1;
        if (int e = launch_col<POST_NONE>(ca, ni, true, st)) return e;
    }
    return CINE_OK;
}

extern "C" int cine_masked_kspace_to_hybrid(const float* k, const uint8_t* mask, float* hyb, int bt, int c, int h, int w,
                                            void* stream) {
    CINE_REQUIRE(k && mask && hyb, CINE_EINVAL, "cine_masked_kspace_to_hybrid: null pointer");
    CINE_REQUIRE(bt > 0 && c > 0 && c <= 32768 && h > 0 && w > 0, CINE_EINVAL, "cine_masked_kspace_to_hybrid: bad sizes");
    if (int e = check_n(h, "cine_masked_kspace_to_hybrid(h)")) return e;
    hipStream_t st = as_stream(stream);
    const long nimg = (long)bt * c, step = 32768 / c * c;
    for (long i0 = 0; i0 < nimg; i0 += step) {
        const long ni = (nimg - i0) < step ? (nimg - i0) : step;
        ColArgs ca{};
        ca.in = reinterpret_cast<const cf*>(k) + i0 * h * w;
        ca.out = reinterpret_cast<cf*>(hyb) + i0 * h * w;
        ca.H = h; ca.W = w; ca.s_in = (h + 1) / 2; ca.s_out = h / 2; ca.coils = c;
        ca.premask = mask + (i0 / c) * h;
        if (int e = launch_col<POST_NONE, false, true>(ca, ni, true, st)) return e;
    }
    return CINE_OK;
}

extern "C" size_t cine_image_dc_ws_bytes(int b, int t, int c, int h, int w) {
    if (b <= 0 || t <= 0 || c <= 0 || h != 200 || w <= 0 || c <= kDcCS) return 0;
    return (size_t)ceil_div(c, kDcCS) * b * t * h * w * sizeof(cf);
}

static int image_dc_impl(const float* img, const float* sens, const float* zf, const uint8_t* mask,
                         const float* lambda_dev, int lam_beta, float w_sampled, float w_unsampled, float beta,
                         float* out, int b, int t, int c, int h, int w, int magnitude,
                         void* ws, size_t ws_bytes, void* stream, float* pd_part = nullptr, float* pd_wg = nullptr, const float* sens_tiled = nullptr);

extern "C" int cine_image_dc(const float* img, const float* sens, const float* zf, const uint8_t* mask,
                             const float* lambda_dev, float w_sampled, float w_unsampled, float beta,
                             float* out, int b, int t, int c, int h, int w, int magnitude,
                             void* ws, size_t ws_bytes, void* stream) {
    return image_dc_impl(img, sens, zf, mask, lambda_dev, 0, w_sampled, w_unsampled, beta, out, b, t, c, h, w, magnitude, ws, ws_bytes, stream);
}

// CineNet's H operator (models/cinenet.py:121-133) for a row mask: A^H M A img + softplus(*lambda_dev) img, one kernel chain
extern "C" int cine_normal_op(const float* img, const float* sens, const uint8_t* mask, const float* lambda_dev,
                              float* out, int b, int t, int c, int h, int w, void* ws, size_t ws_bytes, void* stream) {
    CINE_REQUIRE(lambda_dev, CINE_EINVAL, "cine_normal_op: null lambda");
    return image_dc_impl(img, sens, img, mask, lambda_dev, 1, 1.f, 0.f, 0.f, out, b, t, c, h, w, 0, ws, ws_bytes, stream);
}
// The same two operators reading the sensitivities from their column-tile-major copy (cine_sens_tile_pack; h == 200 only, else ignored):
// imgdc200_kernel's loads of the maps drop from 128 KB to 40 KB of cache lines per workgroup (38 -> 34 us per launch at cfg 4 / cfg 2).
// The maps are constant over a forward pass: one pack serves its 6 (cfg 2) to 42 (cfg 4) operator applications.
extern "C" size_t cine_sens_tile_floats(int b, int c, int h, int w) {
    return (b <= 0 || c <= 0 || h != 200 || w <= 0) ? 0 : (size_t)b * c * ceil_div(w, kDcCW) * 200 * kDcCW * 2;
}
namespace cine {
__global__ __launch_bounds__(256) void sens_tile_pack_kernel(const cf* __restrict__ s, cf* __restrict__ o, int W, int ntx, long planes) {
    const long total = planes * ntx * 200L * kDcCW;                     // o[plane][tile][row][k] = s[plane][row][5 tile + k] (0 past the last column)
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
        const int k = (int)(e % kDcCW);
        long r = e / kDcCW;
        const int row = (int)(r % 200); r /= 200;
        const int tile = (int)(r % ntx);
        const long plane = r / ntx;
        const int col = tile * kDcCW + k;
        o[e] = col < W ? s[(plane * 200 + row) * W + col] : mk(0.f, 0.f);
    }
}
}  // namespace cine
extern "C" int cine_sens_tile_pack(const float* sens, float* tiled, int b, int c, int h, int w, void* stream) {
    CINE_REQUIRE(sens && tiled, CINE_EINVAL, "cine_sens_tile_pack: null pointer");
    CINE_REQUIRE(b > 0 && c > 0 && h == 200 && w > 0, CINE_EUNSUPPORTED, "cine_sens_tile_pack: h must be 200 (the only engine that reads the tiled maps)");
    hipStream_t st = as_stream(stream);
    ProfScope prof(F_PACK, st);
    const long total = (long)b * c * ceil_div(w, kDcCW) * 200 * kDcCW;
    hipLaunchKernelGGL(sens_tile_pack_kernel, dim3((unsigned)std::min<long>(ceil_div(total, 256L), 4096)), dim3(256), 0, st,
                       reinterpret_cast<const cf*>(sens), reinterpret_cast<cf*>(tiled), w, ceil_div(w, kDcCW), (long)b * c);
    return check_launch("sens_tile_pack_kernel");
}
extern "C" int cine_image_dc_t(const float* img, const float* sens, const float* sens_tiled, const float* zf, const uint8_t* mask,
                               const float* lambda_dev, float w_sampled, float w_unsampled, float beta,
                               float* out, int b, int t, int c, int h, int w, int magnitude,
                               void* ws, size_t ws_bytes, void* stream) {
    return image_dc_impl(img, sens, zf, mask, lambda_dev, 0, w_sampled, w_unsampled, beta, out, b, t, c, h, w, magnitude, ws, ws_bytes, stream,
                         nullptr, nullptr, sens_tiled);
}
extern "C" int cine_normal_op_t(const float* img, const float* sens, const float* sens_tiled, const uint8_t* mask, const float* lambda_dev,
                                float* out, int b, int t, int c, int h, int w, void* ws, size_t ws_bytes, void* stream) {
    CINE_REQUIRE(lambda_dev, CINE_EINVAL, "cine_normal_op_t: null lambda");
    return image_dc_impl(img, sens, img, mask, lambda_dev, 1, 1.f, 0.f, 0.f, out, b, t, c, h, w, 0, ws, ws_bytes, stream, nullptr, nullptr, sens_tiled);
}
// cine_normal_op that also leaves the 256 partial sums of <img, out> in pd_part (device, 256 floats): the p.d of the conjugate-gradient
// step that follows (cinenet.py:155-159), computed where out is produced instead of by a separate pass over both vectors.  Returns
// CINE_EUNSUPPORTED for shapes whose operator does not end in the partial-sum kernel (h != 200 or <= 5 coils): use cine_cg_step then.
extern "C" int cine_normal_op_pd(const float* img, const float* sens, const uint8_t* mask, const float* lambda_dev,
                                 float* out, float* pd_part, int b, int t, int c, int h, int w, void* ws, size_t ws_bytes, void* stream) {
    CINE_REQUIRE(lambda_dev && pd_part, CINE_EINVAL, "cine_normal_op_pd: null pointer");
    CINE_REQUIRE(cine_image_dc_ws_bytes(b, t, c, h, w) > 0, CINE_EUNSUPPORTED, "cine_normal_op_pd: this shape has no partial-sum kernel");
    return image_dc_impl(img, sens, img, mask, lambda_dev, 1, 1.f, 0.f, 0.f, out, b, t, c, h, w, 0, ws, ws_bytes, stream, pd_part);
}

// One conjugate-gradient iteration of cinenet.py:153-169 for a row mask in THREE launches: imgdc200_kernel leaves the coil groups'
// partial sums of A^H M A p and one partial sum of <p, H p> per workgroup; cg_update_fused_kernel adds the groups and the regulariser term
// (the arithmetic of imgdc_sum_kernel: d is bit-identical), alpha, x += alpha p, r -= alpha d, partial sums of r.r; cg_direction_kernel
// beta and p.  H p itself is never written.  p.d is added up per workgroup here instead of per 256-thread stripe of the vector, so alpha
// differs from cine_normal_op_pd + cine_cg_step_pd in the last bits (deterministic all the same).
namespace cine {
int launch_cg_update_fused(float* x, float* r, float* p, const cf* partial, int nz, long part_stride, const float* lam, long ncf,
                           const float* pd_wg, int npd, const float* rr_old, float* rr_new, float* rr_part, float* pd_out, hipStream_t st);   // pack_kernels.hip
}
static long cg_fused_blocks(int b, int t, int c, int h, int w) {
    if (b <= 0 || t <= 0 || c <= kDcCS || h != 200 || w <= 0) return 0;
    return 8L * ceil_div(ceil_div(w, kDcCW) * ceil_div(c, kDcCS), 8) * b * t;
}
extern "C" size_t cine_cg_fused_ws_bytes(int b, int t, int c, int h, int w) {
    const long nb = cg_fused_blocks(b, t, c, h, w);
    return nb ? (size_t)(nb + 256) * sizeof(float) : 0;           // per-workgroup p.d partials + 256 r.r partials
}
extern "C" int cine_normal_op_cg_fused_t(float* x, float* r, float* p, const float* sens, const float* sens_tiled, const uint8_t* mask,
                                         const float* lambda_dev, const float* rr_old_dev, float* rr_new_dev, float* pd_out_dev,
                                         int b, int t, int c, int h, int w, void* ws_dc, size_t ws_dc_bytes, void* ws_cg, size_t ws_cg_bytes, void* stream);
extern "C" int cine_normal_op_cg_fused(float* x, float* r, float* p, const float* sens, const uint8_t* mask, const float* lambda_dev,
                                       const float* rr_old_dev, float* rr_new_dev, float* pd_out_dev, int b, int t, int c, int h, int w,
                                       void* ws_dc, size_t ws_dc_bytes, void* ws_cg, size_t ws_cg_bytes, void* stream) {
    return cine_normal_op_cg_fused_t(x, r, p, sens, nullptr, mask, lambda_dev, rr_old_dev, rr_new_dev, pd_out_dev, b, t, c, h, w,
                                     ws_dc, ws_dc_bytes, ws_cg, ws_cg_bytes, stream);
}
extern "C" int cine_normal_op_cg_fused_t(float* x, float* r, float* p, const float* sens, const float* sens_tiled, const uint8_t* mask,
                                         const float* lambda_dev, const float* rr_old_dev, float* rr_new_dev, float* pd_out_dev,
                                         int b, int t, int c, int h, int w, void* ws_dc, size_t ws_dc_bytes, void* ws_cg, size_t ws_cg_bytes, void* stream) {
    CINE_REQUIRE(x && r && p && sens && mask && lambda_dev && rr_old_dev && rr_new_dev && ws_dc && ws_cg, CINE_EINVAL, "cine_normal_op_cg_fused: null pointer");
    CINE_REQUIRE(rr_old_dev != rr_new_dev, CINE_EINVAL, "cine_normal_op_cg_fused: rr_old and rr_new must be different scalars");
    const long nb = cg_fused_blocks(b, t, c, h, w);
    CINE_REQUIRE(nb > 0 && nb <= 0x7fffffffL, CINE_EUNSUPPORTED, "cine_normal_op_cg_fused: needs h == 200 and more than %d coils", kDcCS);
    CINE_REQUIRE(ws_cg_bytes >= cine_cg_fused_ws_bytes(b, t, c, h, w), CINE_EWORKSPACE, "cine_normal_op_cg_fused: workspace too small");
    float* pd_wg = reinterpret_cast<float*>(ws_cg);
    if (int e = image_dc_impl(p, sens, p, mask, lambda_dev, 1, 1.f, 0.f, 0.f, nullptr, b, t, c, h, w, 0, ws_dc, ws_dc_bytes, stream, nullptr, pd_wg, sens_tiled)) return e;
    const long ncf = (long)b * t * h * w;
    return launch_cg_update_fused(x, r, p, reinterpret_cast<const cf*>(ws_dc), ceil_div(c, kDcCS), ncf, lambda_dev, ncf, pd_wg, (int)nb,
                                  rr_old_dev, rr_new_dev, pd_wg + nb, pd_out_dev, as_stream(stream));
}

// ---- the whole conjugate-gradient solve of CineNet's DC block (cinenet.py:136-171) for a row mask, TWO launches per iteration:
//   set-up    imgdc200_kernel (H x0, coil-group sums) -> cg_init_kernel: r = b - H x0, {p_old = 0, r} pairs, partial sums of r.r
//   iteration imgdc200_kernel<true>: beta = r.r / r.r_old from the two partial-sum arrays, p = r + beta p_old formed ON LOAD (one 16-byte
//             {p_old, r} element where the plain kernel reads an 8-byte one), coil-group sums of A^H M A p, p.Hp per workgroup, p written
//             by coil group 0  ->  cg_update2_kernel: alpha, x += alpha p, r -= alpha (sum_z partial_z + v p), new {p, r} pairs, r.r partials
// The direction update of round 4's three-launch form (cg_direction_kernel) needs the global r.r, i.e. a kernel boundary behind the update;
// here that boundary is the one in front of the next operator anyway.  No direction is computed behind the last iteration (p is not used
// again), and the set-up is 2 launches instead of 7 (operator, coil sums, b - Hx, two copies, dot, dot): 14 launches per solve of 6
// iterations instead of 25.  Same arithmetic as the three-launch form except for the summation order of the first r.r.
namespace cine {
int launch_cg_init(const float* x, const float* rhs, int rhs_ref, const cf* partial, int nz, long part_stride, const float* lam, long ncf,
                   float4* pr, float* rr_part, hipStream_t st);                                                    // pack_kernels.hip
int launch_cg_update2(float* x, float4* pr, const cf* p, const cf* partial, int nz, long part_stride, const float* lam, long ncf,
                      const float* pd_wg, int npd, const float* rr_prev, float* rr_cur, int last, hipStream_t st,
                      float* rr_rec = nullptr, float* pd_rec = nullptr, float* rr_final = nullptr);
}
extern "C" size_t cine_conj_grad_ws_bytes(int b, int t, int c, int h, int w) {
    const long nb = cg_fused_blocks(b, t, c, h, w);
    if (!nb) return 0;
    const size_t ncf = (size_t)b * t * h * w;
    return cine_image_dc_ws_bytes(b, t, c, h, w) + (size_t)(nb + 512) * sizeof(float) + 256 + ncf * (sizeof(float4) + sizeof(cf));
}
static int conj_grad_impl(float* x, const float* rhs, int rhs_is_ref, const float* sens, const float* sens_tiled, const uint8_t* mask,
                          const float* lambda_dev, int iters, int b, int t, int c, int h, int w, void* ws, size_t ws_bytes, void* stream,
                          float* p_rec, float* rr_rec, float* pd_rec);
extern "C" int cine_conj_grad(float* x, const float* rhs, int rhs_is_ref, const float* sens, const float* sens_tiled, const uint8_t* mask,
                              const float* lambda_dev, int iters, int b, int t, int c, int h, int w, void* ws, size_t ws_bytes, void* stream) {
    return conj_grad_impl(x, rhs, rhs_is_ref, sens, sens_tiled, mask, lambda_dev, iters, b, t, c, h, w, ws, ws_bytes, stream, nullptr, nullptr, nullptr);
}
// cine_conj_grad that RECORDS what the adjoint recurrence of the iteration needs (training, cine_hip/autograd.py ConjGradFn: the reference detaches
// its step sizes, cinenet.py:159-169, so the iteration it differentiates is linear with these constants): p_rec (iters, b, t, 1, h, w, 2) receives
// every direction p_k (the operator kernel writes it there instead of into its scratch), rr_rec (iters + 1) the r_k . r_k and pd_rec (iters) the p_k . H p_k.
extern "C" int cine_conj_grad_rec(float* x, const float* rhs, int rhs_is_ref, const float* sens, const float* sens_tiled, const uint8_t* mask,
                                  const float* lambda_dev, int iters, int b, int t, int c, int h, int w, void* ws, size_t ws_bytes,
                                  float* p_rec, float* rr_rec, float* pd_rec, void* stream) {
    CINE_REQUIRE(p_rec && rr_rec && pd_rec && iters >= 1, CINE_EINVAL, "cine_conj_grad_rec: null record buffer or no iteration");
    return conj_grad_impl(x, rhs, rhs_is_ref, sens, sens_tiled, mask, lambda_dev, iters, b, t, c, h, w, ws, ws_bytes, stream, p_rec, rr_rec, pd_rec);
}
static int conj_grad_impl(float* x, const float* rhs, int rhs_is_ref, const float* sens, const float* sens_tiled, const uint8_t* mask,
                          const float* lambda_dev, int iters, int b, int t, int c, int h, int w, void* ws, size_t ws_bytes, void* stream,
                          float* p_rec, float* rr_rec, float* pd_rec) {
    CINE_REQUIRE(x && rhs && sens && mask && lambda_dev && ws, CINE_EINVAL, "cine_conj_grad: null pointer");
    CINE_REQUIRE(iters >= 0 && x != rhs, CINE_EINVAL, "cine_conj_grad: bad arguments");
    const long nb = cg_fused_blocks(b, t, c, h, w);
    CINE_REQUIRE(nb > 0 && nb <= 0x7fffffffL, CINE_EUNSUPPORTED, "cine_conj_grad: needs h == 200 and more than %d coils", kDcCS);
    CINE_REQUIRE((long)b * t <= 65535, CINE_EUNSUPPORTED, "cine_conj_grad: b*t > 65535");
    CINE_REQUIRE(ws_bytes >= cine_conj_grad_ws_bytes(b, t, c, h, w), CINE_EWORKSPACE, "cine_conj_grad: workspace too small");
    const long ncf = (long)b * t * h * w;
    const int nz = ceil_div(c, kDcCS);
    char* wp = reinterpret_cast<char*>(ws);
    cf* partial = reinterpret_cast<cf*>(wp); wp += cine_image_dc_ws_bytes(b, t, c, h, w);
    float* pd_wg = reinterpret_cast<float*>(wp); wp += (size_t)nb * sizeof(float);
    float* rr[2] = {reinterpret_cast<float*>(wp), reinterpret_cast<float*>(wp) + 256}; wp += 512 * sizeof(float);
    wp = reinterpret_cast<char*>((reinterpret_cast<uintptr_t>(wp) + 255) & ~uintptr_t(255));
    float4* pr = reinterpret_cast<float4*>(wp); wp += (size_t)ncf * sizeof(float4);
    cf* pbuf = reinterpret_cast<cf*>(wp);
    hipStream_t st = as_stream(stream);
    // H x0: the plain operator kernel; asking for its per-workgroup dot partials keeps the coil-group sums for cg_init_kernel
    if (int e = image_dc_impl(x, sens, x, mask, lambda_dev, 1, 1.f, 0.f, 0.f, nullptr, b, t, c, h, w, 0, partial, cine_image_dc_ws_bytes(b, t, c, h, w),
                              stream, nullptr, pd_wg, sens_tiled)) return e;
    if (int e = launch_cg_init(x, rhs, rhs_is_ref ? 1 : 0, partial, nz, ncf, lambda_dev, ncf, pr, rr[0], st)) return e;
    ImgDcArgs a{};
    a.sens = reinterpret_cast<const cf*>(sens); a.sens_t = reinterpret_cast<const cf*>(sens_tiled); a.mask = mask;
    a.lam = lambda_dev; a.lam_beta = 1; a.w1 = 1.f; a.w0 = 0.f; a.beta = 0.f;
    a.T = t; a.C = c; a.H = h; a.W = w; a.partial = partial; a.part_stride = ncf; a.BT = b * t; a.ntx = ceil_div(w, kDcCW); a.nz = nz;
    a.pd_wg = pd_wg; a.cg_pr = pr; a.cg_p_out = pbuf;
    for (int k = 0; k < iters; ++k) {
        // beta_k = r.r after update k-1 / r.r before it; k = 0: both are the set-up's sums and p_old = 0, i.e. p = r exactly
        a.cg_num = k == 0 ? rr[0] : rr[k & 1];
        a.cg_den = k == 0 ? rr[0] : rr[(k - 1) & 1];
        cf* pk = p_rec ? reinterpret_cast<cf*>(p_rec) + (long)k * ncf : pbuf;          // training: every direction is kept
        a.cg_p_out = pk;
        {
            ProfScope prof(F_FFT_COL, st);
            hipLaunchKernelGGL(imgdc200_kernel<true>, dim3((unsigned)nb), dim3(kDcT), (size_t)200 * kDcL * sizeof(cf), st, a);
            if (int e = check_launch("imgdc200_kernel<cg>")) return e;
        }
        if (int e = launch_cg_update2(x, pr, pk, partial, nz, ncf, lambda_dev, ncf, pd_wg, (int)nb, rr[k & 1], rr[(k + 1) & 1], k + 1 == iters, st,
                                      rr_rec ? rr_rec + k : nullptr, pd_rec ? pd_rec + k : nullptr, (rr_rec && k + 1 == iters) ? rr_rec + iters : nullptr)) return e;
    }
    return CINE_OK;
}

static int image_dc_impl(const float* img, const float* sens, const float* zf, const uint8_t* mask,
                         const float* lambda_dev, int lam_beta, float w_sampled, float w_unsampled, float beta,
                         float* out, int b, int t, int c, int h, int w, int magnitude,
                         void* ws, size_t ws_bytes, void* stream, float* pd_part, float* pd_wg, const float* sens_tiled) {
    CINE_REQUIRE(img && sens && mask && (out || pd_wg), CINE_EINVAL, "cine_image_dc: null pointer");
    CINE_REQUIRE(b > 0 && t > 0 && c > 0 && h > 0 && w > 0, CINE_EINVAL, "cine_image_dc: bad sizes");
    CINE_REQUIRE((long)b * t <= 65535, CINE_EUNSUPPORTED, "cine_image_dc: b*t > 65535");
    CINE_REQUIRE(img != out, CINE_EINVAL, "cine_image_dc: out must not alias img (workgroups read neighbouring columns' rows)");
    if (int e = check_n(h, "cine_image_dc(h)")) return e;
    const size_t need = cine_image_dc_ws_bytes(b, t, c, h, w);
    CINE_REQUIRE(need == 0 || (ws && ws_bytes >= need), CINE_EWORKSPACE, "cine_image_dc: workspace %zu < %zu", ws_bytes, need);
    ImgDcArgs a{};
    a.img = reinterpret_cast<const cf*>(img); a.sens = reinterpret_cast<const cf*>(sens);
    a.zf = reinterpret_cast<const cf*>(zf); a.mask = mask; a.lam = lambda_dev; a.lam_beta = lam_beta;
    a.w1 = w_sampled; a.w0 = w_unsampled; a.beta = beta;
    a.out = magnitude ? nullptr : reinterpret_cast<cf*>(out); a.out_abs = magnitude ? out : nullptr;
    a.T = t; a.C = c; a.H = h; a.W = w;
    hipStream_t st = as_stream(stream);
    ProfScope prof(F_FFT_COL, st);
    if (h == 200) {
        const int nz = ceil_div(c, kDcCS);
        CINE_REQUIRE(nz <= 65535, CINE_EUNSUPPORTED, "cine_image_dc: %d coils", c);
        a.partial = reinterpret_cast<cf*>(ws); a.part_stride = (long)b * t * h * w;
        a.BT = b * t; a.ntx = ceil_div(w, kDcCW); a.nz = nz;
        a.sens_t = reinterpret_cast<const cf*>(sens_tiled);
        const long nblk = 8L * ceil_div(a.ntx * nz, 8) * a.BT;
        CINE_REQUIRE(nblk <= 0x7fffffffL, CINE_EUNSUPPORTED, "cine_image_dc: grid too large");
        CINE_REQUIRE(!pd_wg || nz > 1, CINE_EUNSUPPORTED, "cine_image_dc: per-workgroup dot partials need more than one coil group");
        a.pd_wg = pd_wg;
        hipLaunchKernelGGL(imgdc200_kernel<false>, dim3((unsigned)nblk), dim3(kDcT), (size_t)200 * kDcL * sizeof(cf), st, a);
        if (int e = check_launch("imgdc200_kernel")) return e;
        if (nz > 1 && !pd_wg) {           // (with pd_wg the caller's next kernel adds the coil groups itself: cine_normal_op_cg_fused)
            const long n = a.part_stride;
            a.pd_part = pd_part;        // 256 workgroups when the p.d partial sums ride along (what cg_update_kernel adds up)
            hipLaunchKernelGGL(imgdc_sum_kernel, dim3(pd_part ? 256u : (unsigned)std::min<long>(ceil_div(n, 256L), 2048)), dim3(256), 0, st, a, nz, n);
        }
    } else {
        const size_t lds = lds_bytes(false, h, kLinesGen);
        if (int e = allow_lds(imgdc_generic_kernel, lds, "imgdc_generic_kernel")) return e;
        hipLaunchKernelGGL(imgdc_generic_kernel, dim3(ceil_div(w, kLinesGen), b * t), dim3(kThreadsGen), lds, st, a);
    }
    return check_launch("imgdc_kernel");
}

extern "C" int cine_hybrid_reduce(const float* hyb, const float* sens, float* out,
                                  int b, int t, int c, int h, int w, int magnitude, void* stream) {
    CINE_REQUIRE(hyb && sens && out, CINE_EINVAL, "cine_hybrid_reduce: null pointer");
    CINE_REQUIRE(b > 0 && t > 0 && c > 0 && h > 0 && w > 0, CINE_EINVAL, "cine_hybrid_reduce: bad sizes");
    if (int e = check_n(w, "cine_hybrid_reduce(w)")) return e;
    RowArgs r{};
    r.in = reinterpret_cast<const cf*>(hyb);
    r.out = reinterpret_cast<cf*>(out); r.out_abs = out;
    r.W = w; r.s_in = (w + 1) / 2; r.s_out = w / 2;
    r.sens = reinterpret_cast<const cf*>(sens);
    r.T = t; r.C = c; r.H = h;
    coil_tiling(c, w, r.rpw, r.cc);
    CINE_REQUIRE((long)b * t <= 65535, CINE_EUNSUPPORTED, "cine_hybrid_reduce: b*t > 65535");
    dim3 grid(ceil_div(h, r.rpw), b * t);
    return magnitude ? launch_row<PRE_NONE, RPOST_REDUCE_ABS>(r, grid, true, as_stream(stream))
                     : launch_row<PRE_NONE, RPOST_REDUCE>(r, grid, true, as_stream(stream));
}

// Zero-filled reconstruction (traintest_scripts/run_inference.py:64-67): rss_complex(ifft2c(k, norm=None) * sqrt(h w), dim=coil)
// = root-sum-of-squares over the coils of the ORTHO inverse transform.  tmp: scratch of k's size (may alias k: destroys it).
extern "C" int cine_zero_filled_rss(const float* k, float* out, float* tmp, int b, int t, int c, int h, int w, void* stream) {
    CINE_REQUIRE(k && out && tmp, CINE_EINVAL, "cine_zero_filled_rss: null pointer");
    CINE_REQUIRE(b > 0 && t > 0 && c > 0 && h > 0 && w > 0, CINE_EINVAL, "cine_zero_filled_rss: bad sizes");
    if (int e = check_n(w, "cine_zero_filled_rss(w)")) return e;
    if (int e = cine_kspace_to_hybrid(k, tmp, (long)b * t * c, h, w, stream)) return e;
    RowArgs r{};
    r.in = reinterpret_cast<const cf*>(tmp);
    r.out = nullptr; r.out_abs = out;
    r.W = w; r.s_in = (w + 1) / 2; r.s_out = w / 2;
    r.sens = nullptr;
    r.T = t; r.C = c; r.H = h;
    coil_tiling(c, w, r.rpw, r.cc);
    CINE_REQUIRE((long)b * t <= 65535, CINE_EUNSUPPORTED, "cine_zero_filled_rss: b*t > 65535");
    return launch_row<PRE_NONE, RPOST_RSS>(r, dim3(ceil_div(h, r.rpw), b * t), true, as_stream(stream));
}

extern "C" int cine_sens_reduce(const float* k, const float* sens, float* out, float* tmp,
                                int b, int t, int c, int h, int w, int magnitude, void* stream) {
    CINE_REQUIRE(k && sens && out && tmp, CINE_EINVAL, "cine_sens_reduce: null pointer");
    CINE_REQUIRE(b > 0 && t > 0 && c > 0 && h > 0 && w > 0, CINE_EINVAL, "cine_sens_reduce: bad sizes");
    if (int e = cine_kspace_to_hybrid(k, tmp, (long)b * t * c, h, w, stream)) return e;
    return cine_hybrid_reduce(tmp, sens, out, b, t, c, h, w, magnitude, stream);
}

// shared by cine_sens_expand_dc (to_hybrid = false) and cine_expand_dc_hybrid (to_hybrid = true)
static int expand_dc(const float* img, const float* sens, const float* kref, const uint8_t* mask,
                     const float* lambda_dev, float* out, int b, int t, int c, int h, int w,
                     int hard_mask, bool to_hybrid, void* stream, const char* what) {
    CINE_REQUIRE(img && sens && out, CINE_EINVAL, "%s: null pointer", what);
    CINE_REQUIRE(b > 0 && t > 0 && c > 0 && h > 0 && w > 0 && c <= 32768, CINE_EINVAL, "%s: bad sizes", what);
    CINE_REQUIRE(!hard_mask || mask, CINE_EINVAL, "%s: hard_mask needs mask", what);
    CINE_REQUIRE(hard_mask >= 0 && hard_mask <= 2 && (hard_mask != 2 || kref), CINE_EINVAL, "%s: hard_mask %d", what, hard_mask);
    CINE_REQUIRE(!kref || hard_mask || (mask && lambda_dev), CINE_EINVAL, "%s: soft DC needs mask and lambda_dev", what);
    if (int e = check_n(h, what)) return e;
    if (int e = check_n(w, what)) return e;
    hipStream_t st = as_stream(stream);
    RowArgs r{};
    r.out = reinterpret_cast<cf*>(out);
    r.W = w; r.s_in = (w + 1) / 2; r.s_out = w / 2;
    r.sens = reinterpret_cast<const cf*>(sens);
    r.img = reinterpret_cast<const cf*>(img);
    r.T = t; r.C = c; r.H = h;
    coil_tiling(c, w, r.rpw, r.cc);
    CINE_REQUIRE((long)b * t <= 65535, CINE_EUNSUPPORTED, "%s: b*t > 65535", what);
    if (int e = launch_row<PRE_SMUL, RPOST_NONE>(r, dim3(ceil_div(h, r.rpw), b * t), false, st)) return e;
    const long nimg = (long)b * t * c;
    const long step = 32768 / c * c;
    for (long i0 = 0; i0 < nimg; i0 += step) {
        const long ni = (nimg - i0) < step ? (nimg - i0) : step;
        ColArgs ca{};
        ca.in = reinterpret_cast<const cf*>(out) + i0 * h * w;
        ca.out = reinterpret_cast<cf*>(out) + i0 * h * w;
        ca.H = h; ca.W = w; ca.s_in = (h + 1) / 2; ca.s_out = h / 2; ca.coils = c;
        ca.kref = kref ? reinterpret_cast<const cf*>(kref) + i0 * h * w : nullptr;
        ca.mask = mask ? mask + (i0 / c) * h : nullptr;
        ca.lam = lambda_dev;
        int e;
        if (to_hybrid && h == 200) {
            // forward column FFT -> DC -> inverse column FFT in one kernel
            if (hard_mask == 2) e = launch_col<POST_RESID, true>(ca, ni, false, st);
            else if (hard_mask) e = launch_col<POST_HARD, true>(ca, ni, false, st);
            else if (kref) e = launch_col<POST_DC, true>(ca, ni, false, st);
            else e = launch_col<POST_NONE, true>(ca, ni, false, st);
            if (e) return e;
            continue;
        }
        if (hard_mask == 2) e = launch_col<POST_RESID>(ca, ni, false, st);
        else if (hard_mask) e = launch_col<POST_HARD>(ca, ni, false, st);
        else if (kref) e = launch_col<POST_DC>(ca, ni, false, st);
        else e = launch_col<POST_NONE>(ca, ni, false, st);
        if (e) return e;
        if (to_hybrid) {
            ColArgs ci = ca; ci.kref = nullptr; ci.mask = nullptr;
            if ((e = launch_col<POST_NONE>(ci, ni, true, st))) return e;
        }
    }
    return CINE_OK;
}

extern "C" int cine_sens_expand_dc(const float* img, const float* sens, const float* kref, const uint8_t* mask,
                                   const float* lambda_dev, float* out, int b, int t, int c, int h, int w,
                                   int hard_mask, void* stream) {
    return expand_dc(img, sens, kref, mask, lambda_dev, out, b, t, c, h, w, hard_mask, false, stream, "cine_sens_expand_dc");
}

extern "C" int cine_expand_dc_hybrid(const float* img, const float* sens, const float* kref, const uint8_t* mask,
                                     const float* lambda_dev, float* hyb, int b, int t, int c, int h, int w,
                                     int hard_mask, void* stream) {
    return expand_dc(img, sens, kref, mask, lambda_dev, hyb, b, t, c, h, w, hard_mask, true, stream, "cine_expand_dc_hybrid");
}

// Gradient of cine_image_dc's output with respect to the sensitivity maps, per frame: part (b, t, c, h, w) (see imgdc_sgrad_kernel).
// Weights as cine_image_dc (lambda_dev != NULL: soft DC).  Sum over the frames with cine_coil_accum(NULL, part, ...).
extern "C" int cine_image_dc_sens_grad(const float* img, const float* gout, const float* sens, const uint8_t* mask,
                                       const float* lambda_dev, float w_sampled, float w_unsampled,
                                       float* part, int b, int t, int c, int h, int w, void* stream) {
    CINE_REQUIRE(img && gout && sens && mask && part, CINE_EINVAL, "cine_image_dc_sens_grad: null pointer");
    CINE_REQUIRE(b > 0 && t > 0 && c > 0 && c <= 65535 && h > 0 && w > 0 && (long)b * t <= 65535, CINE_EINVAL, "cine_image_dc_sens_grad: bad sizes");
    if (int e = check_n(h, "cine_image_dc_sens_grad(h)")) return e;
    DcGradArgs a{};
    a.m = reinterpret_cast<const cf*>(img); a.g = reinterpret_cast<const cf*>(gout); a.sens = reinterpret_cast<const cf*>(sens);
    a.mask = mask; a.lam = lambda_dev; a.w1 = w_sampled; a.w0 = w_unsampled; a.part = reinterpret_cast<cf*>(part);
    a.T = t; a.C = c; a.H = h; a.W = w;
    hipStream_t st = as_stream(stream);
    ProfScope prof(F_FFT_COL, st);
    const dim3 grid(ceil_div(w, kLinesGen), c, b * t);
    const size_t lds = (size_t)(2 * h * (kLinesGen + 1) + h) * sizeof(cf);
    if (h == 200) hipLaunchKernelGGL(imgdc_sgrad_kernel<true>, grid, dim3(kThreadsGen), lds, st, a);
    else {
        if (int e = allow_lds(imgdc_sgrad_kernel<false>, lds, "imgdc_sgrad_kernel")) return e;
        hipLaunchKernelGGL(imgdc_sgrad_kernel<false>, grid, dim3(kThreadsGen), lds, st, a);
    }
    return check_launch("imgdc_sgrad_kernel");
}
