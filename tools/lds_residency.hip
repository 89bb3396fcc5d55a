// How many workgroups of a given dynamic-LDS size are co-resident per CU on this device?
// Each block spins for ~20 us; total time / 20 us = rounds = ceil(blocks_per_cu_launched / resident).
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void spin(unsigned long long ticks, int* sink) {
    extern __shared__ int lds[];
    lds[threadIdx.x] = threadIdx.x;
    unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    while (__builtin_amdgcn_s_memrealtime() - t0 < ticks) { __builtin_amdgcn_s_sleep(8); }
    if (lds[threadIdx.x] == -1) *sink = 1;
}
int main() {
    int* sink; hipMalloc(&sink, 4);
    hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
    printf("CUs %d  sharedMemPerBlock %zu  sharedMemPerMultiprocessor %zu maxSharedOptin %zu\n", p.multiProcessorCount,
           p.sharedMemPerBlock, p.sharedMemPerMultiprocessor, p.sharedMemPerBlockOptin);
    const unsigned long long ticks = 2000;   // 100 MHz realtime -> 20 us
    for (int set_attr = 0; set_attr < 2; ++set_attr) {
        if (set_attr) hipFuncSetAttribute((const void*)spin, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        for (int threads : {64, 256, 320}) {
            for (int kb : {1, 12, 26, 37, 52, 64, 80}) {
                if (!set_attr && kb > 64) continue;
                int occ = -1;
                hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, spin, threads, kb * 1024);
                hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
                const int per_cu = 16;
                hipLaunchKernelGGL(spin, dim3(p.multiProcessorCount * per_cu), dim3(threads), kb * 1024, 0, ticks, sink);
                hipDeviceSynchronize();
                hipEventRecord(e0);
                hipLaunchKernelGGL(spin, dim3(p.multiProcessorCount * per_cu), dim3(threads), kb * 1024, 0, ticks, sink);
                hipEventRecord(e1); hipEventSynchronize(e1);
                float ms; hipEventElapsedTime(&ms, e0, e1);
                printf("attr=%d threads=%3d lds=%2dKB  api_occ=%2d  time=%7.1f us  -> rounds %.2f -> resident/CU ~ %.1f\n", set_attr, threads, kb,
                       occ, ms * 1e3, ms * 1e3 / 20.0, per_cu / (ms * 1e3 / 20.0));
            }
        }
    }
    return 0;
}
