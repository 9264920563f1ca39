// probe (round 6): does a kernel argument block above 4 KB launch on this runtime?  k_addb_alf would carry ALF's class x transpose coefficient table (2.8 KB) in its arguments.
#include <hip/hip_runtime.h>
#include <stdio.h>
template <int N> struct Big { unsigned v[N]; };
template <int N> __global__ void k(Big<N> a, unsigned *out) { out[threadIdx.x] = a.v[(threadIdx.x * 37) % N] + a.v[N - 1]; }
template <int N> void run()
{
    Big<N> a; for (int i = 0; i < N; i++) a.v[i] = i * 3 + 1;
    unsigned *d, h[64]; hipMalloc(&d, 256);
    hipLaunchKernelGGL(k<N>, dim3(1), dim3(64), 0, 0, a, d);
    hipError_t e = hipGetLastError(); hipError_t e2 = hipDeviceSynchronize();
    hipMemcpy(h, d, 256, hipMemcpyDeviceToHost);
    bool ok = e == hipSuccess && e2 == hipSuccess;
    for (int t = 0; t < 64 && ok; t++) ok = h[t] == (unsigned)(((t * 37) % N) * 3 + 1 + (N - 1) * 3 + 1);
    printf("%5zu-byte argument block: launch %s (%s), result %s\n", sizeof(a) + 8, hipGetErrorName(e), hipGetErrorName(e2), ok ? "correct" : "WRONG");
}
int main() { run<512>(); run<1000>(); run<1100>(); run<1536>(); run<2048>(); return 0; }
