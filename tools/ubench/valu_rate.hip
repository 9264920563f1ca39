// micro-benchmark: issue rate of v_dot2c_i32_i16, v_mad_i32_i24, v_mad_i64_i32, v_perm_b32, v_alignbit on gfx950
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef short v2s __attribute__((ext_vector_type(2)));
#define N_IT 4096
template <int OP> __global__ void k(int *out, int a0, int b0)
{
    int a = a0 + threadIdx.x, b = b0;
    int acc[8] = {0, 1, 2, 3, 4, 5, 6, 7};
    long long acc64[8] = {0, 1, 2, 3, 4, 5, 6, 7};
    for (int i = 0; i < N_IT; i++) {
#pragma unroll
        for (int j = 0; j < 8; j++) {
            if (OP == 0) acc[j] = __builtin_amdgcn_sdot2(__builtin_bit_cast(v2s, a), __builtin_bit_cast(v2s, b + j), acc[j], false);
            if (OP == 1) acc[j] = __mul24(acc[j], b + j) + a;
            if (OP == 2) acc64[j] = (long long)(int)acc64[j] * (b + j) + acc64[j];
            if (OP == 3) acc[j] = __builtin_amdgcn_perm(acc[j], a, b + j);
            if (OP == 4) acc[j] = __builtin_amdgcn_alignbit(acc[j], a, 16);
            if (OP == 5) acc[j] = acc[j] * (b + j) + a;
            if (OP == 6) acc[j] = max(min(acc[j] + a, b + j), -b);
            if (OP == 7) acc[j] = (acc[j] >> (b & 7)) + a;
        }
    }
    int s = 0;
    for (int j = 0; j < 8; j++) s += acc[j] + (int)acc64[j];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int OP> void run(const char *name)
{
    int *d; hipMalloc(&d, 256 * 1024 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<OP>, dim3(1024), dim3(256), 0, 0, d, 3, 5);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<OP>, dim3(1024), dim3(256), 0, 0, d, 3, 5);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double ops = 1024.0 * 256 * N_IT * 8;
    printf("%-18s %8.3f ms  %7.2f Tlane-op/s  (%.2f cycles/wave-instr/SIMD at 2.4GHz)\n", name, ms, ops / ms / 1e9,
           (ms * 1e-3 * 2.4e9) / (ops / 64 / 1024));
    hipFree(d);
}
int main() { run<0>("v_dot2c_i32_i16"); run<1>("v_mad_i32_i24"); run<2>("v_mad_i64_i32"); run<3>("v_perm_b32"); run<4>("v_alignbit+add"); run<5>("mul_lo+add"); run<6>("add+min+max"); run<7>("ashr+add"); return 0; }
