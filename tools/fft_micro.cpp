// Microbenchmark of the FFT + data-consistency passes of one cfg-2 cascade through the C ABI
// (links libcine_hip.so; LD_LIBRARY_PATH picks the build under test).   fft_micro [reps=30]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <vector>
#include "cine_hip.h"
int main(int argc, char** argv) {
    const int reps = argc > 1 ? atoi(argv[1]) : 30;
    const int t = 15, c = 15, h = 200, w = 200;
    const size_t ke = (size_t)t * c * h * w * 2, ie = (size_t)t * h * w * 2, se = (size_t)c * h * w * 2;
    float *k, *hyb, *sens, *img, *lam; uint8_t* mask;
    hipMalloc(&k, ke * 4); hipMalloc(&hyb, ke * 4); hipMalloc(&sens, se * 4); hipMalloc(&img, ie * 4); hipMalloc(&lam, 4);
    hipMalloc(&mask, t * h);
    std::vector<float> hx(ke); for (auto& v : hx) v = rand() / (float)RAND_MAX - .5f;
    hipMemcpy(k, hx.data(), ke * 4, hipMemcpyHostToDevice); hipMemcpy(sens, hx.data(), se * 4, hipMemcpyHostToDevice);
    std::vector<uint8_t> hm(t * h); for (auto& v : hm) v = (rand() % 4) == 0;
    hipMemcpy(mask, hm.data(), t * h, hipMemcpyHostToDevice);
    const float l = 0.54f; hipMemcpy(lam, &l, 4, hipMemcpyHostToDevice);
    const int nf = cine_profile_families();
    std::vector<double> ms(nf); std::vector<long> cnt(nf);
    auto run = [&](const char* name, auto fn, double mb) {
        for (int i = 0; i < 3; ++i) fn();
        hipDeviceSynchronize();
        cine_profile_begin();
        for (int i = 0; i < reps; ++i) fn();
        cine_profile_end(ms.data(), cnt.data(), nf);
        printf("%-22s", name);
        double tot = 0;
        for (int i = 0; i < nf; ++i) if (cnt[i]) { printf("  %s %.1f us", cine_profile_family_name(i), ms[i] / reps * 1e3); tot += ms[i] / reps * 1e3; }
        printf("   | total %.1f us, %.2f TB/s of %.0f MB\n", tot, mb / tot, mb);
    };
    const double K = ke * 4 / 1e6, I = ie * 4 / 1e6, S = se * 4 / 1e6;
    run("kspace_to_hybrid", [&] { cine_kspace_to_hybrid(k, hyb, (long)t * c, h, w, nullptr); }, 2 * K);
    run("hybrid_reduce", [&] { cine_hybrid_reduce(hyb, sens, img, 1, t, c, h, w, 0, nullptr); }, K + S + I);
    run("expand_dc_hybrid", [&] { cine_expand_dc_hybrid(img, sens, k, mask, lam, hyb, 1, t, c, h, w, 0, nullptr); }, I + S + K + K + K + K / 4);
    return 0;
}
