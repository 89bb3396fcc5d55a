// Diagnostic: phase shares of one workgroup of the lean plane kernel (conv_plane.hip): s_memtime stamps of lane 0.
// tools/build_plane_stamps.sh; run: tools/tmp/plane_stamps.bin CIN COUT H W [N] [MODE: 1 norm (default), 2 pooled, 0 plain]
#define CINE_STAMPS 1
#include "conv_plane.hip"
#include <vector>
#include <algorithm>
namespace cine { void set_wgrad_plane(int) {} }      // (grad_kernels.hip is not linked into this tool)
extern "C" int cine_conv_stat_partials(int, int, int, int);
extern "C" size_t cine_conv3x3_packed_floats(int, int);
extern "C" int cine_pack_conv3x3(const float*, float*, int, int, void*);
extern "C" int cine_instnorm_partials(const float*, float*, long, long, void*);
extern "C" int cine_conv3x3_in(const float*, const float*, int, int, int, int, int, const float*, const float*, int, int, int, int, int,
                               const float*, const float*, int, float*, float*, int, int, int, int, float, float, void*);
int main(int argc, char** argv) {
    const int cin = argc > 1 ? atoi(argv[1]) : 16, cout = argc > 2 ? atoi(argv[2]) : 16;
    const int h = argc > 3 ? atoi(argv[3]) : 208, w = argc > 4 ? atoi(argv[4]) : 16, n = argc > 5 ? atoi(argv[5]) : 400;
    const int mode = argc > 6 ? atoi(argv[6]) : 1;
    const int sh = mode == 2 ? 2 * h : h, sw = mode == 2 ? 2 * w : w;
    const size_t xe = (size_t)n * cin * sh * sw, ye = (size_t)n * cout * h * w;
    float *x, *y, *wt, *wp, *px, *py;
    hipMalloc(&x, xe * 4); hipMalloc(&y, ye * 4); hipMalloc(&wt, (size_t)cout * cin * 9 * 4);
    hipMalloc(&wp, cine_conv3x3_packed_floats(cout, cin) * 4);
    const int np = cine_conv_stat_partials(cout, h, w, 0);
    hipMalloc(&px, (size_t)n * cin * 3 * 4); hipMalloc(&py, (size_t)n * cout * np * 3 * 4);
    std::vector<float> hx(xe); for (auto& v : hx) v = rand() / (float)RAND_MAX - .5f;
    hipMemcpy(x, hx.data(), xe * 4, hipMemcpyHostToDevice);
    hipMemcpy(wt, hx.data(), (size_t)cout * cin * 9 * 4, hipMemcpyHostToDevice);
    cine_pack_conv3x3(wt, wp, cout, cin, nullptr);
    cine_instnorm_partials(x, px, (long)n * cin, (long)sh * sw, nullptr);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int it = 0; it < 3; ++it) {
        hipEventRecord(e0);
        int rc = cine_conv3x3_in(x, mode ? px : nullptr, mode ? 1 : 0, cin, mode, sh, sw, nullptr, nullptr, 0, 0, 0, 0, 0, wp, nullptr, 0, y, py, n, cout, h, w, 1e-5f, 0.2f, nullptr);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("rc=%d  launch %d: %.1f us  (%.1f TFLOP/s)\n", rc, it, ms * 1e3, 2.0 * n * h * w * 9.0 * cin * cout / (ms * 1e-3) / 1e12);
    }
    std::vector<unsigned long long> st(1 << 20);
    hipMemcpyFromSymbol(st.data(), HIP_SYMBOL(g_cine_stamps), st.size() * 8);
    double acc[9] = {0}; int cnt = 0;
    for (int b = 0; b < 65536; ++b) {
        const unsigned long long* s = &st[b * 16];
        if (!s[0] || !s[8] || s[8] < s[0] || s[8] - s[0] > 10000000ull) continue;
        bool mono = true; for (int i = 1; i <= 8; ++i) mono &= s[i] >= s[i - 1];
        if (!mono) continue;
        for (int i = 1; i <= 8; ++i) acc[i] += (double)(s[i] - s[i - 1]);
        ++cnt;
    }
    const char* nm[] = {"", "prologue (stats, issue(0), zero fill)", "first barrier", "commit chunk 0 (weights+input)", "barrier after commit", "issue(1) + MFMA sweep chunk 0", "remaining chunks", "stores", "statistics"};
    double tot = 0; for (int i = 1; i <= 8; ++i) tot += acc[i];
    if (!cnt) { printf("no stamps (did the lean kernel take the launch?)\n"); return 1; }
    for (int i = 1; i <= 8; ++i) printf("%-40s %9.0f cycles  %5.1f %%\n", nm[i], acc[i] / cnt, 100 * acc[i] / tot);
    printf("workgroup lifetime %.0f cycles over %d sampled workgroups\n", tot / cnt, cnt);
    {   // wall-clock residency per CU
        unsigned long long t0 = ~0ull, t1 = 0; double life = 0; int m = 0;
        for (int b = 0; b < 65536; ++b) {
            const unsigned long long* s = &st[b * 16];
            if (!s[9] || !s[10] || s[10] < s[9]) continue;
            t0 = std::min(t0, s[9]); t1 = std::max(t1, s[10]); life += (double)(s[10] - s[9]); ++m;
        }
        if (m) printf("kernel span %.1f us (100 MHz clock), mean workgroup life %.1f us, mean resident workgroups per CU %.2f\n", (t1 - t0) / 100.0, life / m / 100.0, life / (double)(t1 - t0) / 256.0);
    }
    return 0;
}
