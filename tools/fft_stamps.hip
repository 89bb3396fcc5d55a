// Diagnostic: where does one workgroup of the fused column kernel spend its cycles?
// hipcc -O3 --offload-arch=gfx950 -DCINE_STAMPS -I deep-cine-cardiac-mri_amd/csrc tools/fft_stamps.hip \
//       deep-cine-cardiac-mri_amd/csrc/api.cpp -x hip -o /tmp/fft_stamps
#define CINE_STAMPS 1
#include "fft_kernels.hip"
#include <vector>
#include <algorithm>
int main() {
    const int t = 15, c = 15, h = 200, w = 200;
    const size_t n = (size_t)t * c * h * w * 2;
    float *k, *hyb, *img, *sens, *lam; uint8_t* mask;
    hipMalloc(&k, n * 4); hipMalloc(&hyb, n * 4); hipMalloc(&img, (size_t)t * h * w * 8); hipMalloc(&sens, (size_t)c * h * w * 8);
    hipMalloc(&lam, 4); hipMalloc(&mask, t * h);
    std::vector<float> hk(n); for (auto& v : hk) v = rand() / (float)RAND_MAX - .5f;
    hipMemcpy(k, hk.data(), n * 4, hipMemcpyHostToDevice); hipMemcpy(hyb, hk.data(), n * 4, hipMemcpyHostToDevice);
    hipMemcpy(img, hk.data(), (size_t)t * h * w * 8, hipMemcpyHostToDevice); hipMemcpy(sens, hk.data(), (size_t)c * h * w * 8, hipMemcpyHostToDevice);
    std::vector<uint8_t> hm(t * h); for (auto& v : hm) v = rand() % 4 == 0;
    hipMemcpy(mask, hm.data(), t * h, hipMemcpyHostToDevice);
    float l = 0.54f; hipMemcpy(lam, &l, 4, hipMemcpyHostToDevice);
    for (int it = 0; it < 3; ++it) cine_expand_dc_hybrid(img, sens, k, mask, lam, hyb, 1, t, c, h, w, 0, nullptr);
    hipDeviceSynchronize();
    std::vector<unsigned long long> st(1 << 20);
    hipMemcpyFromSymbol(st.data(), HIP_SYMBOL(g_cine_stamps), st.size() * 8);
    const int nwg = 13 * 225;
    double acc[10] = {0}; std::vector<double> life;
    unsigned long long tmin = ~0ull, tmax = 0;
    for (int b = 0; b < nwg; ++b) {
        const unsigned long long* s = &st[b * 16];
        for (int i = 1; i < 10; ++i) acc[i] += (double)(s[i] - s[i - 1]);
        life.push_back((double)(s[9] - s[0]));
        if (s[0]) tmin = std::min(tmin, s[0]);
        tmax = std::max(tmax, s[9]);
    }
    const char* names[] = {"", "loads+r10 (round 0)", "loads+r10 (round 1)", "barrier 1", "LDS read + r20", "DC blend", "r20 inv + LDS write",
                           "barrier 2", "LDS read + r10 inv + stores (0)", "LDS read + r10 inv + stores (1)"};
    for (int i = 1; i < 10; ++i) printf("%-36s %9.0f cycles avg\n", names[i], acc[i] / nwg);
    std::sort(life.begin(), life.end());
    printf("workgroup lifetime median %.0f cycles, p90 %.0f; kernel span %llu cycles (s_memtime ticks)\n", life[nwg / 2], life[nwg * 9 / 10], tmax - tmin);
    { hipFuncAttributes fa; hipFuncGetAttributes(&fa, (const void*)cine::col200_kernel<1, 1, true>);
      printf("col200_kernel<1,1,true>: %d VGPRs, %zu B static LDS, max threads %d\n", fa.numRegs, fa.sharedSizeBytes, fa.maxThreadsPerBlock); }
    printf("=> average concurrent workgroups %.1f (%.2f per CU)\n", (acc[1]+acc[2]+acc[3]+acc[4]+acc[5]+acc[6]+acc[7]+acc[8]+acc[9]) / (double)(tmax - tmin), (acc[1]+acc[2]+acc[3]+acc[4]+acc[5]+acc[6]+acc[7]+acc[8]+acc[9]) / (double)(tmax - tmin) / 256);
    return 0;
}
