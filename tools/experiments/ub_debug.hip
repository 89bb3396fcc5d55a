// Diagnostic: run unet_bottom_kernel on one plane with random inputs / weights, dump the LDS activation buffer after a layer.
// hipcc -O3 --offload-arch=gfx950 -DCINE_UB_DEBUG -I deep-cine-cardiac-mri_amd/csrc tools/ub_debug.hip deep-cine-cardiac-mri_amd/csrc/api.cpp -x hip -o tools/ub_debug.bin
// Writes raw float files into gpurun_out/ub/: inputs, unpacked weights and the dump; tools/ub_debug_check.py recomputes on the CPU.
#define CINE_UB_DEBUG 1
#include "unet_bottom.hip"
#include <vector>
#include <cstdio>
#include <string>
using namespace cine;
static std::vector<float> rnd(size_t n, float s, unsigned seed) { std::vector<float> v(n); srand(seed); for (auto& x : v) x = s * (rand() / (float)RAND_MAX - 0.5f); return v; }
static void save(const char* name, const std::vector<float>& v) { std::string p = std::string("gpurun_out/ub/") + name; FILE* f = fopen(p.c_str(), "wb"); fwrite(v.data(), 4, v.size(), f); fclose(f); }
// pack like pack_weights_kernel (conv_kernels.hip): conv3x3 [chunk][tap][ck=8][rowsp], tconv [chunk][1][ck=16][rowsp=4*cout]
static std::vector<float> pack3(const std::vector<float>& w, int cout, int cin) {
    const int ck = 8, nch = (cin + ck - 1) / ck, rowsp = (cout + 15) / 16 * 16; std::vector<float> p((size_t)nch * 9 * ck * rowsp, 0.f);
    for (int ch = 0; ch < nch; ++ch) for (int tap = 0; tap < 9; ++tap) for (int k = 0; k < ck; ++k) for (int m = 0; m < cout; ++m) {
        const int ci = ch * ck + k; if (ci < cin) p[(((size_t)ch * 9 + tap) * ck + k) * rowsp + m] = w[((size_t)m * cin + ci) * 9 + tap]; }
    return p;
}
static std::vector<float> packt(const std::vector<float>& w, int cin, int cout) {   // w (cin, cout, 2, 2)
    const int ck = 16, nch = cin / ck, rows = 4 * cout; std::vector<float> p((size_t)nch * ck * rows, 0.f);
    for (int ch = 0; ch < nch; ++ch) for (int k = 0; k < ck; ++k) for (int m = 0; m < rows; ++m) {
        const int ci = ch * ck + k, b = m & 1, co = (m >> 1) % cout, a_ = (m >> 1) / cout;
        p[((size_t)ch * ck + k) * rows + m] = w[((size_t)ci * cout + co) * 4 + 2 * a_ + b]; }
    return p;
}
int main(int argc, char** argv) {
    const int stop = argc > 1 ? atoi(argv[1]) : 0;
    system("mkdir -p gpurun_out/ub");
    auto x1 = rnd(32 * 104 * 8, 2.f, 1);
    std::vector<float> px1(32 * 3);
    for (int c = 0; c < 32; ++c) {   // exact stats of x1 as one record
        double s = 0, q = 0; for (int i = 0; i < 832; ++i) s += x1[c * 832 + i]; const double m = s / 832;
        for (int i = 0; i < 832; ++i) q += (x1[c * 832 + i] - m) * (x1[c * 832 + i] - m);
        px1[3 * c] = 832; px1[3 * c + 1] = (float)m; px1[3 * c + 2] = (float)q; }
    const int cin[7] = {32, 64, 64, 128, 128, 128, 64}, cout[7] = {64, 64, 128, 128, 64, 64, 64};
    std::vector<std::vector<float>> w(7), wp(7);
    for (int l = 0; l < 7; ++l) {
        if (l == 4) { w[l] = rnd((size_t)128 * 64 * 4, 0.2f, 10 + l); wp[l] = packt(w[l], 128, 64); }
        else { w[l] = rnd((size_t)cout[l] * cin[l] * 9, 0.2f, 10 + l); wp[l] = pack3(w[l], cout[l], cin[l]); }
        char nm[32]; snprintf(nm, sizeof nm, "w%d.bin", l); save(nm, w[l]);
    }
    save("x1.bin", x1);
    BottomArgs a{};
    float *dx1, *dpx1, *dw[7], *dskip, *dy, *dpy, *ddbg;
    hipMalloc(&dx1, x1.size() * 4); hipMemcpy(dx1, x1.data(), x1.size() * 4, hipMemcpyHostToDevice);
    hipMalloc(&dpx1, px1.size() * 4); hipMemcpy(dpx1, px1.data(), px1.size() * 4, hipMemcpyHostToDevice);
    for (int l = 0; l < 7; ++l) { hipMalloc(&dw[l], wp[l].size() * 4); hipMemcpy(dw[l], wp[l].data(), wp[l].size() * 4, hipMemcpyHostToDevice); a.w[l][0] = a.w[l][1] = dw[l]; }
    hipMalloc(&dskip, 64 * 208 * 4); hipMalloc(&dy, 64 * 208 * 4); hipMalloc(&dpy, 64 * 3 * 4); hipMalloc(&ddbg, ub::BUF_FLOATS * 4);
    hipMemset(ddbg, 0, ub::BUF_FLOATS * 4); hipMemset(dy, 0, 64 * 208 * 4);
    a.x1 = dx1; a.px1 = dpx1; a.np1 = 1; a.set_split = 1; a.skip2 = dskip; a.y = dy; a.py = dpy; a.eps = 1e-5f; a.slope = 0.2f;
    a.dbg = ddbg; a.dbg_stop = stop;
    int e = launch_unet_bottom(a, 1, nullptr);
    hipError_t he = hipDeviceSynchronize();
    printf("launch rc %d, sync %s, LDS %d floats, PS2 %d PS3 %d\n", e, hipGetErrorString(he), ub::LDS_FLOATS, ub::PS2, ub::PS3);
    std::vector<float> d(ub::BUF_FLOATS), y(64 * 208), py(64 * 3), sk(64 * 208);
    hipMemcpy(d.data(), ddbg, d.size() * 4, hipMemcpyDeviceToHost); hipMemcpy(y.data(), dy, y.size() * 4, hipMemcpyDeviceToHost);
    hipMemcpy(py.data(), dpy, py.size() * 4, hipMemcpyDeviceToHost); hipMemcpy(sk.data(), dskip, sk.size() * 4, hipMemcpyDeviceToHost);
    save("dump.bin", d); save("y.bin", y); save("py.bin", py); save("skip.bin", sk);
    return 0;
}
