// api.cpp -- version / error string of the C ABI (include/cine_hip.h).
#include <atomic>
#include <cstdarg>
#include <cstdio>
#include <mutex>
#include <vector>
#include "common.h"

namespace cine {
static thread_local char g_err[512] = "";
void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

thread_local int g_alone_on_chip = 0;      // conv_cfg.h: AloneScope (set by entry points whose arguments say one slice runs alone)

static std::atomic<long> g_diag[D_COUNT];
void diag_count(int which) { if (which >= 0 && which < D_COUNT) g_diag[which].fetch_add(1, std::memory_order_relaxed); }

// ---- optional launch profiler
static std::atomic<int> g_prof_on{0};
struct ProfRec { int fam; hipEvent_t e0, e1; };
static std::mutex g_prof_mu;
static std::vector<ProfRec> g_prof;

ProfScope::ProfScope(int family, hipStream_t stream) : fam(family), st(stream), on(false) {
    if (!g_prof_on.load(std::memory_order_relaxed)) return;
    if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) return;
    on = hipEventRecord(e0, st) == hipSuccess;
}
ProfScope::~ProfScope() {
    if (!on) return;
    (void)hipEventRecord(e1, st);
    std::lock_guard<std::mutex> lk(g_prof_mu);
    g_prof.push_back(ProfRec{fam, e0, e1});
}
}  // namespace cine

extern "C" {
int cine_profile_begin(void) {
    std::lock_guard<std::mutex> lk(cine::g_prof_mu);
    cine::g_prof.clear();
    cine::g_prof_on.store(1);
    return CINE_OK;
}
int cine_profile_end(double* ms, long* launches, int nfam) {
    cine::g_prof_on.store(0);
    std::lock_guard<std::mutex> lk(cine::g_prof_mu);
    for (int i = 0; i < nfam; ++i) { if (ms) ms[i] = 0; if (launches) launches[i] = 0; }
    for (auto& r : cine::g_prof) {
        float t = 0.f;
        (void)hipEventSynchronize(r.e1);
        (void)hipEventElapsedTime(&t, r.e0, r.e1);
        if (r.fam < nfam) { if (ms) ms[r.fam] += t; if (launches) launches[r.fam] += 1; }
        (void)hipEventDestroy(r.e0); (void)hipEventDestroy(r.e1);
    }
    cine::g_prof.clear();
    return CINE_OK;
}
int cine_profile_families(void) { return cine::F_COUNT; }
const char* cine_profile_family_name(int i) {
    static const char* names[] = {"fft_col_pass", "fft_row_pass", "conv3x3_mfma", "instnorm_stats", "tconv2x2",
                                  "conv1x1_bias", "pack_unpack", "misc"};
    return (i >= 0 && i < cine::F_COUNT) ? names[i] : "";
}
long cine_diag_counter(int which, int reset) {
    if (which < 0 || which >= cine::D_COUNT) return -1;
    return reset ? cine::g_diag[which].exchange(0) : cine::g_diag[which].load();
}
int cine_version(void) { return 3; }       // 3: U-Net passes as concurrent branches, C time sweeps, diagnostic counters (round 6); 2: LeakyReLU slope per call (round 5)
const char* cine_last_error(void) { return cine::g_err; }
const char* cine_build_arch(void) { return "gfx950"; }
int cine_pad16(int n) { return ((n - 1) | 15) + 1; }
}
