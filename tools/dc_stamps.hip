// Diagnostic: where does one workgroup of imgdc200_kernel spend its cycles?
// hipcc -O3 --offload-arch=gfx950 -DCINE_STAMPS -I deep-cine-cardiac-mri_amd/csrc tools/dc_stamps.hip \
//       deep-cine-cardiac-mri_amd/csrc/api.cpp -x hip -o tools/dc_stamps.bin
#define CINE_STAMPS 1
#include "fft_kernels.hip"
#include <vector>
#include <algorithm>
int main() {
    const int t = 15, c = 15, h = 200, w = 200;
    float *img, *sens, *zf, *out, *lam; uint8_t* mask; void* ws;
    const size_t ie = (size_t)t * h * w * 2, se = (size_t)c * h * w * 2;
    hipMalloc(&img, ie * 4); hipMalloc(&zf, ie * 4); hipMalloc(&out, ie * 4); hipMalloc(&sens, se * 4); hipMalloc(&lam, 4); hipMalloc(&mask, t * h);
    const size_t wsb = cine_image_dc_ws_bytes(1, t, c, h, w); hipMalloc(&ws, wsb);
    std::vector<float> hk(se); for (auto& v : hk) v = rand() / (float)RAND_MAX - .5f;
    hipMemcpy(img, hk.data(), ie * 4, hipMemcpyHostToDevice); hipMemcpy(zf, hk.data(), ie * 4, hipMemcpyHostToDevice);
    hipMemcpy(sens, hk.data(), se * 4, hipMemcpyHostToDevice);
    std::vector<uint8_t> hm(t * h); for (auto& v : hm) v = rand() % 4 == 0;
    hipMemcpy(mask, hm.data(), t * h, hipMemcpyHostToDevice);
    float l = 0.54f; hipMemcpy(lam, &l, 4, hipMemcpyHostToDevice);
    for (int it = 0; it < 3; ++it) cine_image_dc(img, sens, zf, mask, lam, 0, 0, 0, out, 1, t, c, h, w, 0, ws, wsb, nullptr);
    hipDeviceSynchronize();
    std::vector<unsigned long long> st(1 << 20);
    hipMemcpyFromSymbol(st.data(), HIP_SYMBOL(g_cine_stamps), st.size() * 8);
    const int nwg = 40 * 15 * 3;
    double acc[10] = {0}; std::vector<double> life;
    unsigned long long tmin = ~0ull, tmax = 0;
    for (int b = 0; b < nwg; ++b) {
        const unsigned long long* s = &st[b * 16];
        for (int i = 1; i < 10; ++i) acc[i] += (double)(s[i] - s[i - 1]);
        life.push_back((double)(s[9] - s[0]));
        if (s[0]) tmin = std::min(tmin, s[0]);
        tmax = std::max(tmax, s[9]);
    }
    const char* names[] = {"", "P1 loads + r10 (round 0)", "P1 loads + r10 (round 1)", "barrier 1", "P2 r20 pair", "barrier 2 wait", "P3 r10 inv (0)", "P3 r10 inv (1)",
                           "barrier 3", "P4 sum + store"};
    double tot = 0;
    for (int i = 1; i < 10; ++i) { printf("%-28s %9.0f cycles avg\n", names[i], acc[i] / nwg); tot += acc[i]; }
    std::sort(life.begin(), life.end());
    printf("workgroup lifetime median %.0f, p90 %.0f; kernel span %llu (s_memtime ticks = 100 MHz)\n", life[nwg / 2], life[nwg * 9 / 10], tmax - tmin);
    printf("=> average concurrent workgroups %.1f (%.2f per CU)\n", tot / (double)(tmax - tmin), tot / (double)(tmax - tmin) / 256);
    return 0;
}
