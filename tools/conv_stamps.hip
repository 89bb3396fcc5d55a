// Diagnostic: phase shares of one conv3x3 workgroup (layer 16->16 on 400 planes of 208x16, cfg-2 level 0).
#define CINE_STAMPS 1
// #define CINE_FAST_BUILD 1   (16-wide instantiations only: quicker to build)
#include "conv_kernels.hip"
#include <vector>
#include <algorithm>
int main(int argc, char** argv) {
    const int n = argc > 6 ? atoi(argv[6]) : 400, cin = argc > 1 ? atoi(argv[1]) : 16, cout = argc > 2 ? atoi(argv[2]) : 16;
    const int h = argc > 3 ? atoi(argv[3]) : 208, w = argc > 4 ? atoi(argv[4]) : 16;
    const bool tconv = argc > 5 && atoi(argv[5]) == 2;     // 5th argument 2: transpose conv k2 s2 instead of conv3x3
    const size_t xe = (size_t)n * cin * h * w, ye = (size_t)n * cout * h * w * (tconv ? 4 : 1);
    float *x, *y, *wt, *wp, *px, *py;
    hipMalloc(&x, xe * 4); hipMalloc(&y, ye * 4); hipMalloc(&wt, (size_t)cout * cin * 9 * 4);
    const size_t pf = tconv ? cine_tconv2x2_packed_floats(cin, cout) : cine_conv3x3_packed_floats(cout, cin); hipMalloc(&wp, pf * 4);
    const int np = cine_conv_stat_partials(cout, h, w, tconv ? 1 : 0);
    hipMalloc(&px, (size_t)n * cin * 3 * 4); hipMalloc(&py, (size_t)n * cout * np * 3 * 4);
    std::vector<float> hx(xe); for (auto& v : hx) v = rand() / (float)RAND_MAX - .5f;
    hipMemcpy(x, hx.data(), xe * 4, hipMemcpyHostToDevice);
    hipMemcpy(wt, hx.data(), (size_t)cout * cin * 9 * 4, hipMemcpyHostToDevice);
    if (tconv) cine_pack_tconv2x2(wt, wp, cin, cout, nullptr); else cine_pack_conv3x3(wt, wp, cout, cin, nullptr);
    cine_instnorm_partials(x, px, (long)n * cin, (long)h * w, nullptr);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int it = 0; it < 3; ++it) {
        hipEventRecord(e0);
        int rc = tconv ? cine_tconv2x2_in(x, px, 1, 1, wp, nullptr, n, y, py, n, cin, cout, h, w, 1e-5f, 0.2f, nullptr)
                       : cine_conv3x3_in(x, px, 1, cin, 1, h, w, nullptr, nullptr, 0, 0, 0, 0, 0, wp, nullptr, 0, y, py, n, cout, h, w, 1e-5f, 0.2f, nullptr);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("rc=%d  launch %d: %.1f us  (%.1f TFLOP/s)\n", rc, it, ms * 1e3, 2.0 * n * h * w * 9.0 * cin * cout / (ms * 1e-3) / 1e12);
    }
    {
        auto kern = cine::conv_mfma_kernel<8, 1, 1, 4, 13, 16, 9>;
        using C = cine::ConvCfg<8, 1, 1, 4, 13, 16, 9>;
        hipFuncAttributes fa; hipFuncGetAttributes(&fa, (const void*)kern);
        for (size_t lds : {C::lds_bytes(16), (size_t)30000, (size_t)20000}) {
            int occ = -1; hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, kern, 256, lds);
            printf("config A: numRegs %d, static LDS %zu, dyn LDS %zu -> API blocks/CU %d\n", fa.numRegs, fa.sharedSizeBytes, lds, occ);
        }
    }
    std::vector<unsigned long long> st(1 << 20);
    hipMemcpyFromSymbol(st.data(), HIP_SYMBOL(g_cine_stamps), st.size() * 8);
    const char* names[] = {"", "prologue (stats merge + zero fill)", "(pre-loop)", "stage chunk 0 (loads+transform+LDS)", "barrier", "MFMA sweep chunk 0", "remaining chunks", "epilogue stats", "store"};
    double acc[9] = {0}; int cnt = 0;
    for (int b = 0; b < 65536; ++b) {
        const unsigned long long* s = &st[b * 16];
        if (!s[0] || !s[8] || s[8] < s[0] || s[8] - s[0] > 10000000ull) continue;
        bool mono = true; for (int i = 1; i <= 8; ++i) mono &= s[i] >= s[i - 1];
        if (!mono) continue;
        // slots: 0 start,1 after prologue,2 after first sync,3 after staging,4 after sync,5 after sweep0,6 loop end,7 after stats,8 end
        for (int i = 1; i <= 8; ++i) acc[i] += (double)(s[i] - s[i - 1]);
        ++cnt;
    }
    const char* nm[] = {"", "prologue (stats merge + zero fill)", "first barrier", "stage chunk 0 (weights+input)", "barrier after staging", "MFMA sweep chunk 0", "remaining chunks (stage+sweep)", "epilogue statistics", "store"};
    double tot = 0; for (int i = 1; i <= 8; ++i) tot += acc[i];
    for (int i = 1; i <= 8; ++i) printf("%-36s %9.0f cycles  %5.1f %%\n", nm[i], acc[i] / cnt, 100 * acc[i] / tot);
    printf("workgroup lifetime %.0f cycles over %d sampled workgroups\n", tot / cnt, cnt);
    {   // inside the prologue: 0 -> 11 (thread / tile decode, statistics loads issued), 11 -> 12 (first chunk's loads issued),
        // 12 -> 13 (statistics merged, table written), 13 -> 1 (halo columns zeroed)
        double sub[4] = {0, 0, 0, 0}; int c2 = 0;
        for (int b = 0; b < 65536; ++b) {
            const unsigned long long* s = &st[b * 16];
            if (!s[0] || !s[1] || !s[11] || !s[12] || !s[13] || s[1] < s[0] || s[1] - s[0] > 10000000ull) continue;
            sub[0] += (double)(s[11] - s[0]); sub[1] += (double)(s[12] - s[11]); sub[2] += (double)(s[13] - s[12]); sub[3] += (double)(s[1] - s[13]); ++c2;
        }
        if (c2) printf("prologue split: decode+stat loads %.0f | issue(0) %.0f | merge+table %.0f | halo zero %.0f cycles\n", sub[0] / c2, sub[1] / c2, sub[2] / c2, sub[3] / c2);
    }
    {   // wall-clock residency: per (xcc, se, cu) count how many workgroups overlap on average
        struct Iv { unsigned long long a, b; unsigned long long id; };
        std::vector<Iv> iv;
        unsigned long long t0 = ~0ull, t1 = 0; double life = 0;
        for (int b = 0; b < 65536; ++b) {
            const unsigned long long* s = &st[b * 16];
            if (!s[9] || !s[10] || s[10] < s[9]) continue;
            // HW_ID: cu_id bits 8-11, sh_id bit 12, se_id bits 13-15 (gfx9 layout); xcc from XCC_ID
            const unsigned hw = (unsigned)s[15], xcc = (unsigned)(s[15] >> 32) & 0xf;
            const unsigned long long id = ((unsigned long long)xcc << 16) | ((hw >> 8) & 0xff);
            iv.push_back({s[9], s[10], id});
            t0 = std::min(t0, s[9]); t1 = std::max(t1, s[10]); life += (double)(s[10] - s[9]);
        }
        std::vector<unsigned long long> ids; for (auto& v : iv) ids.push_back(v.id);
        std::sort(ids.begin(), ids.end()); ids.erase(std::unique(ids.begin(), ids.end()), ids.end());
        printf("wall clock: kernel span %.1f us, mean workgroup lifetime %.1f us, %zu workgroups on %zu distinct CU ids\n",
               (t1 - t0) * 0.01, life / iv.size() * 0.01, iv.size(), ids.size());
        printf("=> average resident workgroups %.1f (%.2f per CU id)\n", life / (double)(t1 - t0), life / (double)(t1 - t0) / ids.size());
        // start-time histogram: how many workgroups started within the first N us
        for (double us : {1.0, 5.0, 10.0, 20.0, 40.0, 80.0}) {
            int c = 0; for (auto& v : iv) c += (v.a - t0) * 0.01 <= us;
            printf("   started within %5.1f us: %d\n", us, c);
        }
    }
    return 0;
}
