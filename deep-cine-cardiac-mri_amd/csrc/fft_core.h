// fft_core.h -- in-LDS line-FFT engines shared by every centered-FFT kernel.
//
// Compiles for the device (hipcc) and for the host (g++, used by
// tests/host/fft_host_check.cpp to validate the butterflies and index maps
// against a direct DFT without a GPU).
//
// A "tile" is LINES independent complex lines of N points kept in LDS as
//     tile[p * LP + line]            (line fastest, LP >= LINES)
// Engines transform every line in place along p.  Work is expressed as
// "items" so a kernel runs   for (i = tid; i < items; i += nthreads) stage(i)
// with a workgroup barrier between stages; the host check runs the same
// stage functions serially.
//
// Engine<200> is a two-stage Cooley-Tukey 10 x 20 (both radices done as
// twiddle-free prime-factor 2x5 / 4x5 butterflies in registers):
//     n = 20 n1 + n2 ,  k = k1 + 10 k2
//   NP flavour (natural in, permuted out): r10 over n1 (stride 20), twiddle
//     W200^(n2 k1), r20 over n2 (contiguous)  ->  X[k] sits at pos 20 k1 + k2
//   PN flavour (permuted in, natural out): r20 over k2, twiddle, r10 over k1
//     ->  x[n] sits at pos n.
// A forward NP followed by an inverse PN needs no reordering in between, which
// is what the fused FFT -> data-consistency -> IFFT column kernel uses.
// The ortho scale 1/sqrt(N) is folded into the inter-stage twiddle.
#pragma once

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define CINE_HD __host__ __device__ __forceinline__
typedef float2 cf;
#else
#include <cmath>
#define CINE_HD inline
struct cf { float x, y; };
#endif

namespace cine {

CINE_HD cf mk(float x, float y) { cf r; r.x = x; r.y = y; return r; }
CINE_HD cf cadd(cf a, cf b) { return mk(a.x + b.x, a.y + b.y); }
CINE_HD cf csub(cf a, cf b) { return mk(a.x - b.x, a.y - b.y); }
CINE_HD cf cmul(cf a, cf b) { return mk(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x); }
CINE_HD cf cmulc(cf a, cf b) { return mk(a.x * b.x + a.y * b.y, a.y * b.x - a.x * b.y); }  // a * conj(b)
CINE_HD cf cscale(cf a, float s) { return mk(a.x * s, a.y * s); }
// multiply by -i (DIR=+1, forward e^{-i..}) or +i (DIR=-1, inverse)
template <int DIR> CINE_HD cf mul_mi(cf a) { return DIR > 0 ? mk(a.y, -a.x) : mk(-a.y, a.x); }

// ---- radix-2/4/5 butterflies, in place on register arrays, natural order ----
template <int DIR> CINE_HD void dft2(cf& a, cf& b) { cf t = csub(a, b); a = cadd(a, b); b = t; }

template <int DIR> CINE_HD void dft3(cf& a0, cf& a1, cf& a2) {
    const float s3 = 0.86602540378443865f;    // sin(2pi/3)
    cf t = cadd(a1, a2), d = mul_mi<DIR>(cscale(csub(a1, a2), s3));
    cf m = mk(a0.x - 0.5f * t.x, a0.y - 0.5f * t.y);
    a0 = cadd(a0, t); a1 = cadd(m, d); a2 = csub(m, d);
}

template <int DIR> CINE_HD void dft4(cf& a0, cf& a1, cf& a2, cf& a3) {
    cf s02 = cadd(a0, a2), d02 = csub(a0, a2);
    cf s13 = cadd(a1, a3), d13 = mul_mi<DIR>(csub(a1, a3));
    a0 = cadd(s02, s13); a2 = csub(s02, s13);
    a1 = cadd(d02, d13); a3 = csub(d02, d13);
}

template <int DIR> CINE_HD void dft5(cf& a0, cf& a1, cf& a2, cf& a3, cf& a4) {
    const float c1 = 0.30901699437494742f;    // cos(2pi/5)
    const float c2 = -0.80901699437494742f;   // cos(4pi/5)
    const float s1 = 0.95105651629515357f;    // sin(2pi/5)
    const float s2 = 0.58778525229247313f;    // sin(4pi/5)
    cf t1 = cadd(a1, a4), t2 = cadd(a2, a3);
    cf d1 = csub(a1, a4), d2 = csub(a2, a3);
    cf m1 = mk(a0.x + c1 * t1.x + c2 * t2.x, a0.y + c1 * t1.y + c2 * t2.y);
    cf m2 = mk(a0.x + c2 * t1.x + c1 * t2.x, a0.y + c2 * t1.y + c1 * t2.y);
    // forward: X1 = m1 - i (s1 d1 + s2 d2), X2 = m2 - i (s2 d1 - s1 d2)
    cf u1 = mul_mi<DIR>(mk(s1 * d1.x + s2 * d2.x, s1 * d1.y + s2 * d2.y));
    cf u2 = mul_mi<DIR>(mk(s2 * d1.x - s1 * d2.x, s2 * d1.y - s1 * d2.y));
    a0 = cadd(a0, cadd(t1, t2));
    a1 = cadd(m1, u1); a4 = csub(m1, u1);
    a2 = cadd(m2, u2); a3 = csub(m2, u2);
}

// 10-point DFT, prime-factor 2 x 5: n = (5 n1 + 2 n2) % 10, k = (5 k1 + 6 k2) % 10.
// v[] natural order in, natural order out.
template <int DIR> CINE_HD void dft10(cf (&v)[10]) {
    cf a[2][5];
#pragma unroll
    for (int n1 = 0; n1 < 2; ++n1)
#pragma unroll
        for (int n2 = 0; n2 < 5; ++n2) a[n1][n2] = v[(5 * n1 + 2 * n2) % 10];
#pragma unroll
    for (int n1 = 0; n1 < 2; ++n1) dft5<DIR>(a[n1][0], a[n1][1], a[n1][2], a[n1][3], a[n1][4]);
#pragma unroll
    for (int k2 = 0; k2 < 5; ++k2) dft2<DIR>(a[0][k2], a[1][k2]);
#pragma unroll
    for (int k1 = 0; k1 < 2; ++k1)
#pragma unroll
        for (int k2 = 0; k2 < 5; ++k2) v[(5 * k1 + 6 * k2) % 10] = a[k1][k2];
}

// 20-point DFT, prime-factor 4 x 5: n = (5 n1 + 4 n2) % 20, k = (5 k1 + 16 k2) % 20.
template <int DIR> CINE_HD void dft20(cf (&v)[20]) {
    cf a[4][5];
#pragma unroll
    for (int n1 = 0; n1 < 4; ++n1)
#pragma unroll
        for (int n2 = 0; n2 < 5; ++n2) a[n1][n2] = v[(5 * n1 + 4 * n2) % 20];
#pragma unroll
    for (int n1 = 0; n1 < 4; ++n1) dft5<DIR>(a[n1][0], a[n1][1], a[n1][2], a[n1][3], a[n1][4]);
#pragma unroll
    for (int k2 = 0; k2 < 5; ++k2) dft4<DIR>(a[0][k2], a[1][k2], a[2][k2], a[3][k2]);
#pragma unroll
    for (int k1 = 0; k1 < 4; ++k1)
#pragma unroll
        for (int k2 = 0; k2 < 5; ++k2) v[(5 * k1 + 16 * k2) % 20] = a[k1][k2];
}

// ------------------------------------------------------------------ N = 200
// tw[j] = exp(-2 pi i j / 200) / sqrt(200), j in [0, 200): forward table; the
// inverse uses its conjugate.
struct Fft200 {
    static constexpr int N = 200;
    static constexpr int R1 = 10, R2 = 20;
    // item counts per stage for `lines` lines
    CINE_HD static int items_r10(int lines) { return lines * R2; }   // one per (line, n2|k2-col)
    CINE_HD static int items_r20(int lines) { return lines * R1; }   // one per (line, k1)
    // position of frequency (or, for PN input, of the k-th input) in the tile
    CINE_HD static int pos_of(int k) { return 20 * (k % 10) + k / 10; }

    // Register-level radix-10 for column c (= n2 in the NP flavour, = output column in PN):
    // TW_BEFORE multiplies input j by W200^(c j), TW_AFTER multiplies output j by W200^(c j);
    // either way the ortho scale 1/sqrt(200) rides along.  tw = forward table (conjugated for DIR < 0).
    template <int DIR, bool TW_BEFORE, bool TW_AFTER>
    CINE_HD static void r10_regs(cf (&v)[10], int c, const cf* tw) {
        if (TW_BEFORE) {
#pragma unroll
            for (int j = 1; j < 10; ++j) {
                cf w = tw[(c * j) % 200];
                v[j] = DIR > 0 ? cmul(v[j], w) : cmulc(v[j], w);
            }
            v[0] = cscale(v[0], 0.070710678118654752f);
        }
        dft10<DIR>(v);
        if (TW_AFTER) {
#pragma unroll
            for (int j = 1; j < 10; ++j) {
                cf w = tw[(c * j) % 200];
                v[j] = DIR > 0 ? cmul(v[j], w) : cmulc(v[j], w);
            }
            v[0] = cscale(v[0], 0.070710678118654752f);
        }
    }

    // Strided radix-10 over positions {20 j + c}, j = 0..9, for column c, in place on the tile.
    template <int DIR, bool TW_BEFORE, bool TW_AFTER>
    CINE_HD static void stage_r10(cf* tile, int LP, int item, int lines, const cf* tw) {
        const int line = item % lines, c = item / lines;
        cf v[10];
#pragma unroll
        for (int j = 0; j < 10; ++j) v[j] = tile[(20 * j + c) * LP + line];
        r10_regs<DIR, TW_BEFORE, TW_AFTER>(v, c, tw);
#pragma unroll
        for (int j = 0; j < 10; ++j) tile[(20 * j + c) * LP + line] = v[j];
    }

    // Contiguous radix-20 over positions {20 g + j}, j = 0..19, for group g.
    template <int DIR>
    CINE_HD static void stage_r20(cf* tile, int LP, int item, int lines) {
        const int line = item % lines, g = item / lines;
        cf v[20];
#pragma unroll
        for (int j = 0; j < 20; ++j) v[j] = tile[(20 * g + j) * LP + line];
        dft20<DIR>(v);
#pragma unroll
        for (int j = 0; j < 20; ++j) tile[(20 * g + j) * LP + line] = v[j];
    }
};

// ------------------------------------------------------------ generic N
// Direct O(N^2) DFT for any other length (test shapes, odd temporal lengths).
// Out of place: src tile -> dst tile, natural order both sides.
// tw[j] = exp(-2 pi i j / N) / sqrt(N).
struct DirectDft {
    CINE_HD static int items(int lines, int n) { return lines * n; }
    template <int DIR>
    CINE_HD static void stage(const cf* src, cf* dst, int LP, int item, int lines, int n, const cf* tw) {
        const int line = item % lines, k = item / lines;
        float ax = 0.f, ay = 0.f;
        int idx = 0;
        for (int j = 0; j < n; ++j) {
            cf x = src[j * LP + line];
            cf w = tw[idx];
            cf p = DIR > 0 ? cmul(x, w) : cmulc(x, w);
            ax += p.x; ay += p.y;
            idx += k; if (idx >= n) idx -= n;
        }
        dst[k * LP + line] = mk(ax, ay);
    }
};

// ------------------------------------------------------------------ N = 2^a 3^b 5^c: Stockham autosort, radices 4 / 2 / 3 / 5
// tw[j] = exp(-2 pi i j / n), j in [0, n), UNSCALED; the ortho scale 1 / sqrt(n) is applied by the last stage.  Stage s with radix R
// and Ns = product of the radices before it: butterfly j in [0, n / R) reads src[j + m n / R], m < R, multiplies by
// w_n^(m (j mod Ns) n / (Ns R)), transforms, and writes dst[(j / Ns) Ns R + (j mod Ns) + m Ns] -- natural order in, natural order out
// after the last stage, the two tiles ping-pong.  Lengths with another prime factor keep the direct engine.
struct MixedRadix {
    static constexpr int kMaxStages = 10;
    CINE_HD static bool smooth(int n) {
        if (n < 2) return false;
        while (n % 2 == 0) n /= 2;
        while (n % 3 == 0) n /= 3;
        while (n % 5 == 0) n /= 5;
        return n == 1;
    }
    // radix of stage s (4s first, then one 2, then 3s, then 5s); returns the stage count
    CINE_HD static int plan(int n, int (&radix)[kMaxStages]) {
        int ns = 0;
        while (n % 4 == 0) { radix[ns++] = 4; n /= 4; }
        if (n % 2 == 0) { radix[ns++] = 2; n /= 2; }
        while (n % 3 == 0) { radix[ns++] = 3; n /= 3; }
        while (n % 5 == 0) { radix[ns++] = 5; n /= 5; }
        return ns;
    }
    CINE_HD static int items(int lines, int n, int R) { return lines * (n / R); }
    template <int DIR, int R>
    CINE_HD static void stage_r(const cf* src, cf* dst, int LP, int item, int lines, int n, int Ns, const cf* tw, float scale) {
        const int line = item % lines, j = item / lines;
        const int m = n / R, k = j % Ns;
        const int step = k * (n / (Ns * R));                // twiddle index of input 1; input q uses q * step (< n)
        cf v[R];
#pragma unroll
        for (int q = 0; q < R; ++q) {
            cf x = src[(j + q * m) * LP + line];
            if (q > 0 && step > 0) { const cf w = tw[q * step]; x = DIR > 0 ? cmul(x, w) : cmulc(x, w); }
            v[q] = x;
        }
        if constexpr (R == 4) dft4<DIR>(v[0], v[1], v[2], v[3]);
        else if constexpr (R == 2) dft2<DIR>(v[0], v[1]);
        else if constexpr (R == 3) dft3<DIR>(v[0], v[1], v[2]);
        else dft5<DIR>(v[0], v[1], v[2], v[3], v[4]);
        const int o = (j / Ns) * Ns * R + k;
#pragma unroll
        for (int q = 0; q < R; ++q) dst[(o + q * Ns) * LP + line] = cscale(v[q], scale);
    }
    template <int DIR>
    CINE_HD static void stage(const cf* src, cf* dst, int LP, int item, int lines, int n, int R, int Ns, const cf* tw, float scale) {
        switch (R) {
            case 4: stage_r<DIR, 4>(src, dst, LP, item, lines, n, Ns, tw, scale); break;
            case 2: stage_r<DIR, 2>(src, dst, LP, item, lines, n, Ns, tw, scale); break;
            case 3: stage_r<DIR, 3>(src, dst, LP, item, lines, n, Ns, tw, scale); break;
            default: stage_r<DIR, 5>(src, dst, LP, item, lines, n, Ns, tw, scale); break;
        }
    }
};

}  // namespace cine
