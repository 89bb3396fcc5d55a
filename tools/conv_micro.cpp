// Microbenchmark of one conv3x3 layer through the C ABI (links libcine_hip.so; no kernels compiled here).
//   conv_micro cin cout h w [n=400] [mode=1] [reps=20] [c1=0 (second concat source, mode 1)]
// hipcc tools/conv_micro.cpp -I include -L deep-cine-cardiac-mri_amd/cine_hip -lcine_hip -Wl,-rpath,'$ORIGIN/../deep-cine-cardiac-mri_amd/cine_hip' -o tools/conv_micro.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "cine_hip.h"
int main(int argc, char** argv) {
    const int cin = argc > 1 ? atoi(argv[1]) : 16, cout = argc > 2 ? atoi(argv[2]) : 16;
    const int h = argc > 3 ? atoi(argv[3]) : 208, w = argc > 4 ? atoi(argv[4]) : 16;
    const int n = argc > 5 ? atoi(argv[5]) : 400, mode = argc > 6 ? atoi(argv[6]) : 1, reps = argc > 7 ? atoi(argv[7]) : 20;
    const int c1 = argc > 8 ? atoi(argv[8]) : 0;
    const int sh = mode == 2 ? 2 * h : h, sw = mode == 2 ? 2 * w : w;
    const size_t xe = (size_t)n * cin * sh * sw, x1e = (size_t)n * c1 * h * w, ye = (size_t)n * cout * h * w;
    float *x, *x1 = nullptr, *y, *wt, *wp, *px, *px1 = nullptr, *py;
    hipMalloc(&x, xe * 4); hipMalloc(&y, ye * 4); hipMalloc(&wt, (size_t)cout * (cin + c1) * 9 * 4);
    const size_t pf = cine_conv3x3_packed_floats(cout, cin + c1); hipMalloc(&wp, pf * 4);
    const int np = cine_conv_stat_partials(cout, h, w, 0);
    hipMalloc(&px, (size_t)n * cin * 3 * 4); hipMalloc(&py, (size_t)n * cout * np * 3 * 4);
    std::vector<float> hx(xe > x1e ? xe : x1e); for (auto& v : hx) v = rand() / (float)RAND_MAX - .5f;
    hipMemcpy(x, hx.data(), xe * 4, hipMemcpyHostToDevice);
    hipMemcpy(wt, hx.data(), (size_t)cout * (cin + c1) * 9 * 4, hipMemcpyHostToDevice);
    cine_pack_conv3x3(wt, wp, cout, cin + c1, nullptr);
    cine_instnorm_partials(x, px, (long)n * cin, (long)sh * sw, nullptr);
    if (c1) {
        hipMalloc(&x1, x1e * 4); hipMalloc(&px1, (size_t)n * c1 * 3 * 4);
        hipMemcpy(x1, hx.data(), x1e * 4, hipMemcpyHostToDevice);
        cine_instnorm_partials(x1, px1, (long)n * c1, (long)h * w, nullptr);
    }
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float best = 1e9, sum = 0;
    for (int it = 0; it < reps + 3; ++it) {
        hipEventRecord(e0);
        int rc = cine_conv3x3_in(x, px, 1, cin, mode, sh, sw, x1, px1, 1, c1, 1, h, w, wp, nullptr, 0, y, py, n, cout, h, w, 1e-5f, 0.2f, nullptr);
        hipEventRecord(e1); hipEventSynchronize(e1);
        if (rc) { printf("rc=%d %s\n", rc, cine_last_error()); return 1; }
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (it >= 3) { sum += ms; if (ms < best) best = ms; }
    }
    const double fl = 2.0 * n * h * w * 9.0 * (cin + c1) * cout;
    printf("conv %d(+%d)->%d %dx%d n=%d mode %d: mean %.1f us (%.1f TF)  best %.1f us (%.1f TF)\n", cin, c1, cout, h, w, n, mode,
           sum / reps * 1e3, fl / (sum / reps * 1e-3) / 1e12, best * 1e3, fl / (best * 1e-3) / 1e12);
    return 0;
}
