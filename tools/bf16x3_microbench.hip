// tools/bf16x3_microbench.hip -- what would split-bf16 MFMA buy the regulariser's convolutions on gfx950, and what would it cost?
//
// The product path multiplies in exact fp32 (v_mfma_f32_16x16x4_f32, 32 cycles, 157 TFLOP/s dense).  The alternative: split every fp32 operand
// into bf16 pieces (x = hi + lo [+ lo2]) and form the product from bf16 MFMAs (v_mfma_f32_16x16x32_bf16, 16 cycles, 2.5 PFLOP/s dense):
//   bf16x3: hi.hi + hi.lo + lo.hi            (2-way split, 3 MFMAs per 32-deep k block, ~2^-16 per product)
//   bf16x6: + lo.lo + hi.lo2 + lo2.hi        (3-way split, 6 MFMAs,                       ~2^-23: fp32-like)
// This program measures, with the operand traffic the conv kernels have (a wave = 64 output rows x 16 positions, weights AND activations read
// from LDS for every k block, one wave per SIMD and two):
//   1. the inner-loop rate of each form, whole chip, as EFFECTIVE fp32 TFLOP/s (2 M N K of the product being formed);
//   2. the error of each form against float64 on conv-like data (K = 288 = 32 channels x 9 taps);
//   3. the staging cost of the split: fp32 from HBM -> {hi, lo[, lo2]} in LDS against fp32 -> act() -> LDS.
// build: hipcc -O3 --offload-arch=gfx950 tools/bf16x3_microbench.hip -o tools/tmp/bf16x3_microbench     run: tools/tmp/bf16x3_microbench
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); std::exit(1); } } while (0)

constexpr int K = 288;            // 32 input channels x 9 taps
constexpr int ROWS = 64;          // output rows of a wave (4 row tiles)
constexpr int NPOS = 16;          // positions of a wave's tile
constexpr int KB = K / 32;        // 32-deep k blocks of the bf16 forms
// fp32 images (ds_read_b32 is served per half-wave: lanes (q, kk = 0 | 1), then kk = 2 | 3): A[row][k], row stride == 2 (mod 32): bank 2 q + kk; B[k][pos], stride 16: bank 16 kk + q
constexpr int CK32 = 72, CK16 = 96;                    // k values of one staged chunk: 8 channels x 9 taps (fp32), 3 blocks of 32 (bf16): the loops re-read one chunk's image
constexpr int AS32 = CK32 + 2, BS32 = NPOS;            // dwords
// bf16 images: one 16-byte piece (8 consecutive k) per (row | pos, k block, kk); row stride in pieces chosen == 5 (mod 16) so that 16 rows x 16 B tile the 64 banks
constexpr int PS16 = (CK16 / 32) * 4 + 1;              // pieces per row: 13 (13 x 4 dwords = 52: 16 rows tile the 64 banks)

__device__ __forceinline__ unsigned short bf16_bits(float x) { __bf16 b = (__bf16)x; return *reinterpret_cast<unsigned short*>(&b); }
__device__ __forceinline__ float bf16_val(unsigned short u) { unsigned v = (unsigned)u << 16; return *reinterpret_cast<float*>(&v); }

// ---- 1. inner-loop rates ---------------------------------------------------------------------------------------------------------------
// FORM 0: fp32 MFMA; 1: bf16x1 (no split: the raw bf16 rate with this operand traffic); 3: bf16x3; 6: bf16x6.  REG: operands stay in registers
// (no LDS reads in the loop: the MFMA-only ceiling of the form).
template <int FORM, bool REG>
__global__ __launch_bounds__(256, 2) void rate_kernel(float* __restrict__ out, int iters) {
    extern __shared__ __align__(16) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, q = lane & 15, kk = lane >> 4;
    f32x4 acc[4];
#pragma unroll
    for (int ct = 0; ct < 4; ++ct) acc[ct] = (f32x4){0.f, 0.f, 0.f, 0.f};
    if constexpr (FORM == 0) {
        float* A = reinterpret_cast<float*>(smem);                   // [4 waves share][ROWS][AS32]
        float* B = A + ROWS * AS32 + wave * (CK32 * BS32);           // per wave [CK32][BS32]  (positions differ per wave in a conv)
        for (int e = tid; e < ROWS * AS32; e += 256) A[e] = 1e-3f * (e % 97);
        for (int e = tid; e < 4 * CK32 * BS32; e += 256) (A + ROWS * AS32)[e] = 1e-3f * (e % 89);
        __syncthreads();
        float a_r[4] = {A[q * AS32 + kk], A[(16 + q) * AS32 + kk], A[(32 + q) * AS32 + kk], A[(48 + q) * AS32 + kk]}, b_r = B[kk * BS32 + q];
        for (int it = 0; it < iters * (K / CK32); ++it) {
            asm volatile("" ::: "memory");                           // a new chunk: the LDS image is re-read (nothing hoisted out of the loop)
#pragma unroll
            for (int s = 0; s < CK32 / 4; ++s) {
                if constexpr (!REG) {
                    b_r = B[(4 * s + kk) * BS32 + q];
#pragma unroll
                    for (int ct = 0; ct < 4; ++ct) a_r[ct] = A[(16 * ct + q) * AS32 + 4 * s + kk];
                }
#pragma unroll
                for (int ct = 0; ct < 4; ++ct) acc[ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(a_r[ct], b_r, acc[ct], 0, 0, 0);
            }
        }
    } else {
        constexpr int NP = FORM == 6 ? 3 : (FORM == 3 ? 2 : 1);      // bf16 pieces per operand
        uint4* A = reinterpret_cast<uint4*>(smem);                   // [NP][ROWS][PS16] pieces
        uint4* B = A + NP * ROWS * PS16 + wave * (NP * NPOS * PS16); // per wave [NP][NPOS][PS16]
        for (int e = tid; e < NP * ROWS * PS16 + 4 * NP * NPOS * PS16; e += 256) A[e] = make_uint4(0x3c003c00u + e, 0x3c003c00u, 0x3c003c00u, 0x3c003c00u);
        __syncthreads();
        bf16x8 a_r[NP][4], b_r[NP];
#pragma unroll
        for (int p = 0; p < NP; ++p) {
            b_r[p] = *reinterpret_cast<bf16x8*>(&B[(p * NPOS + q) * PS16 + kk]);
#pragma unroll
            for (int ct = 0; ct < 4; ++ct) a_r[p][ct] = *reinterpret_cast<bf16x8*>(&A[(p * ROWS + 16 * ct + q) * PS16 + kk]);
        }
        for (int it = 0; it < iters * (K / CK16); ++it) {
            asm volatile("" ::: "memory");
#pragma unroll
            for (int s = 0; s < CK16 / 32; ++s) {
                if constexpr (!REG) {
#pragma unroll
                    for (int p = 0; p < NP; ++p) {
                        b_r[p] = *reinterpret_cast<bf16x8*>(&B[(p * NPOS + q) * PS16 + 4 * s + kk]);
#pragma unroll
                        for (int ct = 0; ct < 4; ++ct) a_r[p][ct] = *reinterpret_cast<bf16x8*>(&A[(p * ROWS + 16 * ct + q) * PS16 + 4 * s + kk]);
                    }
                }
#pragma unroll
                for (int ct = 0; ct < 4; ++ct) {
                    acc[ct] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a_r[0][ct], b_r[0], acc[ct], 0, 0, 0);
                    if constexpr (NP >= 2) {
                        acc[ct] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a_r[0][ct], b_r[1], acc[ct], 0, 0, 0);
                        acc[ct] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a_r[1][ct], b_r[0], acc[ct], 0, 0, 0);
                    }
                    if constexpr (NP >= 3) {
                        acc[ct] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a_r[1][ct], b_r[1], acc[ct], 0, 0, 0);
                        acc[ct] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a_r[0][ct], b_r[2], acc[ct], 0, 0, 0);
                        acc[ct] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a_r[2][ct], b_r[0], acc[ct], 0, 0, 0);
                    }
                }
            }
        }
    }
    float s = 0.f;
#pragma unroll
    for (int ct = 0; ct < 4; ++ct) s += acc[ct][0] + acc[ct][1] + acc[ct][2] + acc[ct][3];
    out[(size_t)blockIdx.x * 256 + tid] = s;
}

template <int FORM, bool REG>
static double run_rate(float* out, int wgs, int iters, size_t lds) {
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(rate_kernel<FORM, REG>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL((rate_kernel<FORM, REG>), dim3(wgs), dim3(256), lds, 0, out, 4);       // warm-up
    CK(hipDeviceSynchronize());
    float best = 1e30f;
    for (int r = 0; r < 3; ++r) {
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL((rate_kernel<FORM, REG>), dim3(wgs), dim3(256), lds, 0, out, iters);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); best = std::min(best, ms);
    }
    const double flop = 2.0 * ROWS * NPOS * K * (double)iters * 4 /*waves*/ * wgs;
    return flop / (best * 1e-3) / 1e12;
}

// ---- 2. accuracy -----------------------------------------------------------------------------------------------------------------------
// one wave: D[64][16] = A[64][K] B[K][16] in each form; operands split on the device exactly as a staging phase would
__global__ __launch_bounds__(64) void accuracy_kernel(const float* __restrict__ A, const float* __restrict__ B, float* __restrict__ D /* [4 forms][64][16] */) {
    const int lane = threadIdx.x, q = lane & 15, kk = lane >> 4;
    auto split = [](float x, __bf16 (&p)[3]) {
        p[0] = (__bf16)x; float r = x - (float)p[0];
        p[1] = (__bf16)r; r -= (float)p[1];
        p[2] = (__bf16)r;
    };
    for (int form = 0; form < 4; ++form) {            // 0 fp32, 1 bf16x1, 2 bf16x3, 3 bf16x6
        for (int ct = 0; ct < 4; ++ct) {
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
            if (form == 0) {
                for (int s = 0; s < K / 4; ++s)
                    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(A[(16 * ct + q) * K + 4 * s + kk], B[(4 * s + kk) * NPOS + q], acc, 0, 0, 0);
            } else {
                for (int s = 0; s < KB; ++s) {
                    bf16x8 a[3], b[3];
                    for (int j = 0; j < 8; ++j) {
                        __bf16 pa[3], pb[3];
                        split(A[(16 * ct + q) * K + 32 * s + 8 * kk + j], pa); split(B[(32 * s + 8 * kk + j) * NPOS + q], pb);
                        for (int p = 0; p < 3; ++p) { a[p][j] = pa[p]; b[p][j] = pb[p]; }
                    }
                    // smallest terms first
                    if (form == 3) {
                        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[2], b[0], acc, 0, 0, 0);
                        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[0], b[2], acc, 0, 0, 0);
                        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[1], b[1], acc, 0, 0, 0);
                    }
                    if (form >= 2) {
                        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[1], b[0], acc, 0, 0, 0);
                        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[0], b[1], acc, 0, 0, 0);
                    }
                    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[0], b[0], acc, 0, 0, 0);
                }
            }
            // lane (q, kk) holds D[row = 16 ct + 4 kk + j][col = q]
            for (int j = 0; j < 4; ++j) D[(form * ROWS + 16 * ct + 4 * kk + j) * NPOS + q] = acc[j];
        }
    }
}

// ---- 3. staging cost -------------------------------------------------------------------------------------------------------------------
// a workgroup stages chunks of 8 channels x 256 positions (what a plane-conv chunk holds) from HBM into LDS: MODE 0 fp32 act(), 2 / 3: act() then
// the 2- / 3-way bf16 split written as separate planes.  The LDS image is read back once (checksum) so that nothing is optimised away.
template <int MODE>
__global__ __launch_bounds__(256, 2) void stage_kernel(const float* __restrict__ x, float* __restrict__ out, int chunks) {
    __shared__ __align__(16) unsigned short lds16[3][8 * 256 + 64];
    __shared__ __align__(16) float lds32[8 * 256 + 64];
    const int tid = threadIdx.x;
    const float* p = x + (size_t)blockIdx.x * chunks * 2048;
    float chk = 0.f;
    for (int c = 0; c < chunks; ++c) {
        float4 v[2];
        v[0] = reinterpret_cast<const float4*>(p + (size_t)c * 2048)[tid]; v[1] = reinterpret_cast<const float4*>(p + (size_t)c * 2048)[256 + tid];
        float e[8] = {v[0].x, v[0].y, v[0].z, v[0].w, v[1].x, v[1].y, v[1].z, v[1].w};
#pragma unroll
        for (int j = 0; j < 8; ++j) { const float t = fmaf(e[j], 1.25f, -0.1f); e[j] = fmaxf(t, 0.2f * t); }        // InstanceNorm + LeakyReLU
        if constexpr (MODE == 0) {
            reinterpret_cast<float4*>(lds32)[tid] = make_float4(e[0], e[1], e[2], e[3]);
            reinterpret_cast<float4*>(lds32)[256 + tid] = make_float4(e[4], e[5], e[6], e[7]);
        } else {
            unsigned short h[3][8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                float r = e[j];
#pragma unroll
                for (int pz = 0; pz < MODE; ++pz) { h[pz][j] = bf16_bits(r); r -= bf16_val(h[pz][j]); }
            }
#pragma unroll
            for (int pz = 0; pz < MODE; ++pz)
                reinterpret_cast<uint4*>(lds16[pz])[tid] = make_uint4(h[pz][0] | (h[pz][1] << 16), h[pz][2] | (h[pz][3] << 16), h[pz][4] | (h[pz][5] << 16), h[pz][6] | (h[pz][7] << 16));
        }
        __syncthreads();
        if constexpr (MODE == 0) chk += lds32[(tid * 7 + c) & 2047];
        else chk += (float)lds16[MODE - 1][(tid * 7 + c) & 2047];
        __syncthreads();
    }
    out[(size_t)blockIdx.x * 256 + tid] = chk;
}
template <int MODE>
static double run_stage(const float* x, float* out, int wgs, int chunks) {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL((stage_kernel<MODE>), dim3(wgs), dim3(256), 0, 0, x, out, chunks);
    CK(hipDeviceSynchronize());
    float best = 1e30f;
    for (int r = 0; r < 3; ++r) {
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL((stage_kernel<MODE>), dim3(wgs), dim3(256), 0, 0, x, out, chunks);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); best = std::min(best, ms);
    }
    return (double)wgs * chunks * 2048 / (best * 1e-3) / 1e9;      // G elements / s
}

int main() {
    float* out; CK(hipMalloc(&out, (size_t)4096 * 256 * sizeof(float)));
    std::printf("bf16x3 micro-benchmark, gfx950 (K = %d, a wave = %d rows x %d positions, operands re-read from LDS for every k block)\n", K, ROWS, NPOS);
    const size_t lds32 = (size_t)(ROWS * AS32 + 4 * CK32 * BS32) * 4;
    auto lds16 = [](int np) { return (size_t)(np * ROWS * PS16 + 4 * np * NPOS * PS16) * 16; };
    for (int per_cu = 1; per_cu <= 2; ++per_cu) {
        const int wgs = 256 * per_cu, it = 400;
        std::printf("-- %d workgroup(s) of 4 waves per CU (%d wave(s) per SIMD): effective fp32 TFLOP/s, whole chip\n", per_cu, per_cu);
        std::printf("   form            operands from LDS   operands in registers (MFMA-only ceiling)\n");
        std::printf("   fp32 16x16x4    %8.1f            %8.1f\n", run_rate<0, false>(out, wgs, it, lds32), run_rate<0, true>(out, wgs, it, lds32));
        std::printf("   bf16 x1         %8.1f            %8.1f     (no split: not a candidate, the raw rate)\n", run_rate<1, false>(out, wgs, it, lds16(1)), run_rate<1, true>(out, wgs, it, lds16(1)));
        std::printf("   bf16 x3         %8.1f            %8.1f\n", run_rate<3, false>(out, wgs, it, lds16(2)), run_rate<3, true>(out, wgs, it, lds16(2)));
        std::printf("   bf16 x6         %8.1f            %8.1f\n", run_rate<6, false>(out, wgs, it, lds16(3)), run_rate<6, true>(out, wgs, it, lds16(3)));
    }
    // accuracy on conv-like data: weights ~ N(0, 1) / sqrt(K) (the scale InstanceNorm-ed networks train to), activations ~ LeakyReLU(N(0, 1))
    std::vector<float> A(ROWS * K), B(K * NPOS);
    unsigned s = 12345;
    auto rnd = [&]() { double u = 0; for (int i = 0; i < 12; ++i) { s = s * 1664525u + 1013904223u; u += (s >> 8) / 16777216.0; } return u - 6.0; };
    for (auto& v : A) v = (float)(rnd() / std::sqrt((double)K));
    for (auto& v : B) { const double t = rnd(); v = (float)(t > 0 ? t : 0.2 * t); }
    float *dA, *dB, *dD;
    CK(hipMalloc(&dA, A.size() * 4)); CK(hipMalloc(&dB, B.size() * 4)); CK(hipMalloc(&dD, 4 * ROWS * NPOS * 4));
    CK(hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(accuracy_kernel, dim3(1), dim3(64), 0, 0, dA, dB, dD);
    std::vector<float> D(4 * ROWS * NPOS);
    CK(hipMemcpy(D.data(), dD, D.size() * 4, hipMemcpyDeviceToHost));
    std::printf("-- error of one K = %d product sum against float64 (relative to the RMS of the exact result)\n", K);
    const char* names[4] = {"fp32 16x16x4", "bf16 x1     ", "bf16 x3     ", "bf16 x6     "};
    for (int f = 0; f < 4; ++f) {
        double se = 0, sr = 0, mx = 0;
        for (int r = 0; r < ROWS; ++r)
            for (int c = 0; c < NPOS; ++c) {
                double ref = 0;
                for (int k = 0; k < K; ++k) ref += (double)A[r * K + k] * (double)B[k * NPOS + c];
                const double d = D[(f * ROWS + r) * NPOS + c] - ref;
                se += d * d; sr += ref * ref; mx = std::max(mx, std::fabs(d));
            }
        std::printf("   %s  rms %.2e   max %.2e\n", names[f], std::sqrt(se / sr), mx / std::sqrt(sr / (ROWS * NPOS)));
    }
    // staging
    const int wgs = 1024, chunks = 64;
    float* x; CK(hipMalloc(&x, (size_t)wgs * chunks * 2048 * 4)); CK(hipMemset(x, 0, (size_t)wgs * chunks * 2048 * 4));
    std::printf("-- staging 8-channel x 256-position chunks HBM -> act() -> LDS, whole chip, G elements / s (HBM-bound at ~1 000 - 1 500)\n");
    std::printf("   fp32 image %.0f   2-way bf16 split %.0f   3-way bf16 split %.0f\n", run_stage<0>(x, out, wgs, chunks), run_stage<2>(x, out, wgs, chunks), run_stage<3>(x, out, wgs, chunks));
    return 0;
}
