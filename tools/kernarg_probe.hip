// Probe: how large may a by-value kernel argument be on this runtime?
#include <hip/hip_runtime.h>
#include <cstdio>
template <int N> struct Big { int v[N]; };
template <int N> __global__ void k(Big<N> b, int* out) { if (threadIdx.x == 0) out[0] = b.v[N - 1]; }
template <int N> void probe(int* d) {
    Big<N> b; for (int i = 0; i < N; ++i) b.v[i] = i;
    hipLaunchKernelGGL(k<N>, dim3(1), dim3(64), 0, 0, b, d);
    hipError_t e = hipDeviceSynchronize(); hipError_t e2 = hipGetLastError();
    int h = -1; hipMemcpy(&h, d, 4, hipMemcpyDeviceToHost);
    printf("%6d bytes: sync=%s last=%s value=%d (want %d)\n", N * 4, hipGetErrorString(e), hipGetErrorString(e2), h, N - 1);
}
int main() { int* d; hipMalloc(&d, 4); probe<512>(d); probe<1000>(d); probe<1500>(d); probe<2000>(d); probe<4000>(d); probe<8000>(d); return 0; }
