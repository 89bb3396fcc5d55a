// Host check of csrc/fft_core.h: runs the same stage functions the HIP kernels
// run, serially, and compares with a double-precision direct DFT.
// Build: g++ -O2 -std=c++17 -I<csrc> fft_host_check.cpp -o fft_host_check
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <complex>
#include "fft_core.h"
using namespace cine;
typedef std::complex<double> cd;

static double check(const std::vector<cd>& ref, const std::vector<cf>& got, const char* what) {
    double emax = 0, rmax = 0;
    for (size_t i = 0; i < ref.size(); ++i) {
        emax = std::max(emax, std::abs(ref[i] - cd(got[i].x, got[i].y)));
        rmax = std::max(rmax, std::abs(ref[i]));
    }
    std::printf("%-28s rel_err %.3e\n", what, emax / rmax);
    return emax / rmax;
}

int main() {
    const int N = 200, LINES = 5, LP = 7;
    std::vector<cf> tw(N);
    for (int j = 0; j < N; ++j) {
        double a = -2.0 * M_PI * j / N, s = 1.0 / std::sqrt((double)N);
        tw[j] = mk((float)(std::cos(a) * s), (float)(std::sin(a) * s));
    }
    std::vector<cf> x(N * LP);
    srand(3);
    for (auto& v : x) v = mk(rand() / (float)RAND_MAX - .5f, rand() / (float)RAND_MAX - .5f);
    double worst = 0;
    for (int dir = 1; dir >= -1; dir -= 2) {
        // reference: ortho DFT per line
        std::vector<cd> ref(N * LINES);
        for (int l = 0; l < LINES; ++l)
            for (int k = 0; k < N; ++k) {
                cd acc = 0;
                for (int n = 0; n < N; ++n)
                    acc += cd(x[n * LP + l].x, x[n * LP + l].y) * std::polar(1.0, -dir * 2.0 * M_PI * ((n * k) % N) / N);
                ref[l * N + k] = acc / std::sqrt((double)N);
            }
        // NP flavour
        std::vector<cf> t = x;
        for (int i = 0; i < Fft200::items_r10(LINES); ++i)
            dir > 0 ? Fft200::stage_r10<1, false, true>(t.data(), LP, i, LINES, tw.data())
                    : Fft200::stage_r10<-1, false, true>(t.data(), LP, i, LINES, tw.data());
        for (int i = 0; i < Fft200::items_r20(LINES); ++i)
            dir > 0 ? Fft200::stage_r20<1>(t.data(), LP, i, LINES) : Fft200::stage_r20<-1>(t.data(), LP, i, LINES);
        std::vector<cf> got(N * LINES);
        for (int l = 0; l < LINES; ++l)
            for (int k = 0; k < N; ++k) got[l * N + k] = t[Fft200::pos_of(k) * LP + l];
        worst = std::max(worst, check(ref, got, dir > 0 ? "fft200 NP forward" : "fft200 NP inverse"));
        // PN flavour: place input permuted, expect natural output
        std::vector<cf> u(N * LP);
        for (int l = 0; l < LINES; ++l)
            for (int n = 0; n < N; ++n) u[Fft200::pos_of(n) * LP + l] = x[n * LP + l];
        for (int i = 0; i < Fft200::items_r20(LINES); ++i)
            dir > 0 ? Fft200::stage_r20<1>(u.data(), LP, i, LINES) : Fft200::stage_r20<-1>(u.data(), LP, i, LINES);
        for (int i = 0; i < Fft200::items_r10(LINES); ++i)
            dir > 0 ? Fft200::stage_r10<1, true, false>(u.data(), LP, i, LINES, tw.data())
                    : Fft200::stage_r10<-1, true, false>(u.data(), LP, i, LINES, tw.data());
        for (int l = 0; l < LINES; ++l)
            for (int k = 0; k < N; ++k) got[l * N + k] = u[k * LP + l];
        worst = std::max(worst, check(ref, got, dir > 0 ? "fft200 PN forward" : "fft200 PN inverse"));
        // direct engine
        std::vector<cf> d(N * LP);
        for (int i = 0; i < DirectDft::items(LINES, N); ++i)
            dir > 0 ? DirectDft::stage<1>(x.data(), d.data(), LP, i, LINES, N, tw.data())
                    : DirectDft::stage<-1>(x.data(), d.data(), LP, i, LINES, N, tw.data());
        for (int l = 0; l < LINES; ++l)
            for (int k = 0; k < N; ++k) got[l * N + k] = d[k * LP + l];
        worst = std::max(worst, check(ref, got, dir > 0 ? "direct forward" : "direct inverse"));
    }
    // forward NP then inverse PN round trip without reordering
    {
        std::vector<cf> t = x;
        for (int i = 0; i < Fft200::items_r10(LINES); ++i) Fft200::stage_r10<1, false, true>(t.data(), LP, i, LINES, tw.data());
        for (int i = 0; i < Fft200::items_r20(LINES); ++i) Fft200::stage_r20<1>(t.data(), LP, i, LINES);
        for (int i = 0; i < Fft200::items_r20(LINES); ++i) Fft200::stage_r20<-1>(t.data(), LP, i, LINES);
        for (int i = 0; i < Fft200::items_r10(LINES); ++i) Fft200::stage_r10<-1, true, false>(t.data(), LP, i, LINES, tw.data());
        std::vector<cd> ref(N * LINES); std::vector<cf> got(N * LINES);
        for (int l = 0; l < LINES; ++l)
            for (int n = 0; n < N; ++n) { ref[l * N + n] = cd(x[n * LP + l].x, x[n * LP + l].y); got[l * N + n] = t[n * LP + l]; }
        worst = std::max(worst, check(ref, got, "NP fwd -> PN inv round trip"));
    }
    // mixed-radix engine (N = 2^a 3^b 5^c) against the double-precision DFT, both directions, every supported kind of length
    for (int n : {2, 4, 6, 8, 12, 15, 16, 30, 45, 96, 120, 192, 240, 256, 320, 360, 384, 400, 450, 512}) {
        if (!MixedRadix::smooth(n)) { std::printf("n = %d not smooth?\n", n); return 1; }
        std::vector<cf> twn(n);
        for (int j = 0; j < n; ++j) { double a = -2.0 * M_PI * j / n; twn[j] = mk((float)std::cos(a), (float)std::sin(a)); }
        std::vector<cf> a(n * LP), b(n * LP);
        for (auto& v : a) v = mk(rand() / (float)RAND_MAX - .5f, rand() / (float)RAND_MAX - .5f);
        const std::vector<cf> x0 = a;
        for (int dir = 1; dir >= -1; dir -= 2) {
            a = x0;
            int radix[MixedRadix::kMaxStages];
            const int ns = MixedRadix::plan(n, radix);
            cf* src = a.data(); cf* dst = b.data();
            int Ns = 1;
            for (int s2 = 0; s2 < ns; ++s2) {
                const float sc = s2 == ns - 1 ? (float)(1.0 / std::sqrt((double)n)) : 1.f;
                for (int i = 0; i < MixedRadix::items(LINES, n, radix[s2]); ++i)
                    dir > 0 ? MixedRadix::stage<1>(src, dst, LP, i, LINES, n, radix[s2], Ns, twn.data(), sc)
                            : MixedRadix::stage<-1>(src, dst, LP, i, LINES, n, radix[s2], Ns, twn.data(), sc);
                Ns *= radix[s2];
                std::swap(src, dst);
            }
            std::vector<cd> ref(n * LINES); std::vector<cf> got(n * LINES);
            for (int l = 0; l < LINES; ++l)
                for (int k = 0; k < n; ++k) {
                    cd acc = 0;
                    for (int q = 0; q < n; ++q)
                        acc += cd(x0[q * LP + l].x, x0[q * LP + l].y) * std::polar(1.0, -dir * 2.0 * M_PI * (((long)q * k) % n) / n);
                    ref[l * n + k] = acc / std::sqrt((double)n);
                    got[l * n + k] = src[k * LP + l];
                }
            char what[64]; std::snprintf(what, sizeof what, "mixed radix n=%d %s", n, dir > 0 ? "fwd" : "inv");
            worst = std::max(worst, check(ref, got, what));
        }
    }
    std::printf("%s\n", worst < 2e-6 ? "OK" : "FAIL");
    return worst < 2e-6 ? 0 : 1;
}
