// micro-benchmark (round 6): do the "fast" VALU instructions of gfx950 (v_add_u32, v_and_b32, v_mov_b32, 16-bit VOP2: ~1.15 ns per wave-instruction in valu_rate2) share
// the issue port of the "slow" ones (v_perm_b32, v_pk_*, v_dot2*, v_mad_*: ~1.9 ns)?  N slow + M fast per iteration, in one wave (interleaved) or in different waves.
#include <hip/hip_runtime.h>
#include <stdio.h>
#define N_IT 2048
#define S(r) asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(r) : "v"(a), "v"(b));
#define F(r) asm volatile("v_add_u32 %0, %0, %1" : "+v"(r) : "v"(a));
template <int MODE> __global__ void k(int *out, int a0, int b0)
{
    int a = a0 + threadIdx.x, b = b0;
    int r0 = 0, r1 = 1, r2 = 2, r3 = 3, r4 = 4, r5 = 5, r6 = 6, r7 = 7, f0 = 0, f1 = 1, f2 = 2, f3 = 3, f4 = 4, f5 = 5, f6 = 6, f7 = 7;
    const int wave = threadIdx.x >> 6;
    for (int i = 0; i < N_IT; i++) {
        if (MODE == 0) { S(r0) S(r1) S(r2) S(r3) S(r4) S(r5) S(r6) S(r7) }                                         // 8 slow
        if (MODE == 1) { F(f0) F(f1) F(f2) F(f3) F(f4) F(f5) F(f6) F(f7) }                                         // 8 fast
        if (MODE == 2) { S(r0) F(f0) S(r1) F(f1) S(r2) F(f2) S(r3) F(f3) S(r4) F(f4) S(r5) F(f5) S(r6) F(f6) S(r7) F(f7) }      // 8 slow + 8 fast interleaved in every wave
        if (MODE == 3) { if (wave & 1) { S(r0) S(r1) S(r2) S(r3) S(r4) S(r5) S(r6) S(r7) } else { F(f0) F(f1) F(f2) F(f3) F(f4) F(f5) F(f6) F(f7) } }   // half the waves slow, half fast
        if (MODE == 4) { S(r0) S(r1) S(r2) S(r3) S(r4) S(r5) S(r6) S(r7) F(f0) F(f1) F(f2) F(f3) F(f4) F(f5) F(f6) F(f7) }      // 8 slow then 8 fast
        if (MODE == 5) { S(r0) F(f0) F(f1) S(r1) F(f2) F(f3) S(r2) F(f4) F(f5) S(r3) F(f6) F(f7) S(r4) F(f0) F(f1) S(r5) F(f2) F(f3) S(r6) F(f4) F(f5) S(r7) F(f6) F(f7) }   // 8 slow + 16 fast
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = r0 + r1 + r2 + r3 + r4 + r5 + r6 + r7 + f0 + f1 + f2 + f3 + f4 + f5 + f6 + f7;
}
template <int MODE> void run(const char *name, int wg)
{
    int *d; hipMalloc(&d, 8192 * 256 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<MODE>, dim3(wg), dim3(256), 0, 0, d, 3, 5);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(wg), dim3(256), 0, 0, d, 3, 5);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("%-48s %4d workgroups  %8.3f ms\n", name, wg, ms);
    hipFree(d);
}
int main()
{
    for (int wg : {1024, 2048}) {
        run<0>("8 slow", wg); run<1>("8 fast", wg); run<2>("8 slow + 8 fast interleaved", wg); run<3>("odd waves 8 slow, even waves 8 fast", wg);
        run<4>("8 slow then 8 fast", wg); run<5>("8 slow + 16 fast interleaved", wg);
    }
    return 0;
}
