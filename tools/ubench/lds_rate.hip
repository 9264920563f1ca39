// micro-benchmark (round 6): cost of LDS wave-instructions by width on gfx950 - sub-dword stores in particular (k_addb_alf lost 34 us to sixteen ds_write_b16 per lane).
// Every lane touches its own dword (conflict-free), 4 waves per SIMD resident, 1024 workgroups; ns per wave-instruction and CU.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#define N_IT 512
template <int OP> __global__ void k(int *out, int a0)
{
    __shared__ __attribute__((aligned(16))) uint32_t lds[256 * 4 + 64];      // 4 KB: a lane's 16 bytes at 16 * lane
    const uint32_t base = threadIdx.x * 4;            // byte offset of the lane's dword (b128: 16 * lane)
    uint32_t v = a0 + threadIdx.x, r0 = 0, r1 = 0, r2 = 0, r3 = 0;
    lds[threadIdx.x] = v; lds[256 + threadIdx.x] = v; lds[512 + threadIdx.x] = v; lds[768 + threadIdx.x] = v;
    __syncthreads();
    for (int i = 0; i < N_IT; i++) {
#pragma unroll
        for (int j = 0; j < 8; j++) {
            if (OP == 0) asm volatile("ds_write_b32 %0, %1 offset:%2" :: "v"(base), "v"(v), "n"(0) : "memory");
            if (OP == 1) asm volatile("ds_write_b16 %0, %1 offset:%2" :: "v"(base), "v"(v), "n"(0) : "memory");
            if (OP == 2) asm volatile("ds_write_b8 %0, %1 offset:%2" :: "v"(base), "v"(v), "n"(0) : "memory");
            if (OP == 3) asm volatile("ds_write_b64 %0, %1 offset:%2" :: "v"(base * 2), "v"((uint64_t)v), "n"(0) : "memory");
            if (OP == 4) { typedef uint32_t u4 __attribute__((ext_vector_type(4))); const u4 q = { v, v, v, v }; asm volatile("ds_write_b128 %0, %1 offset:%2" :: "v"(base * 4), "v"(q), "n"(0) : "memory"); }
            if (OP == 5) asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(r0) : "v"(base), "n"(0) : "memory");
            if (OP == 6) asm volatile("ds_read_u16 %0, %1 offset:%2" : "=v"(r0) : "v"(base), "n"(0) : "memory");
            if (OP == 7) asm volatile("ds_read_u8 %0, %1 offset:%2" : "=v"(r0) : "v"(base), "n"(0) : "memory");
            if (OP == 8) { uint64_t q; asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(q) : "v"(base * 2), "n"(0) : "memory"); r1 = (uint32_t)q; }
            if (OP == 9) { typedef uint32_t u4 __attribute__((ext_vector_type(4))); u4 q; asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(q) : "v"(base * 4), "n"(0) : "memory"); r2 = q.x; }
            if (OP == 10) asm volatile("ds_write_b16_d16_hi %0, %1 offset:%2" :: "v"(base), "v"(v), "n"(0) : "memory");
            if (OP == 11) asm volatile("ds_read_u8 %0, %1 offset:%2" : "=v"(r0) : "v"(threadIdx.x), "n"(0) : "memory");          // consecutive BYTES: four lanes per dword
            if (OP == 12) asm volatile("ds_write_b8 %0, %1 offset:%2" :: "v"(threadIdx.x), "v"(v), "n"(0) : "memory");
            if (OP == 13) asm volatile("ds_write_b16 %0, %1 offset:%2" :: "v"(threadIdx.x * 2), "v"(v), "n"(0) : "memory");      // consecutive halves: two lanes per dword
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = r0 + r1 + r2 + r3 + lds[threadIdx.x];
}
template <int OP> void run(const char *name)
{
    int *d; hipMalloc(&d, 1024 * 256 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<OP>, dim3(1024), dim3(256), 0, 0, d, 3);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<OP>, dim3(1024), dim3(256), 0, 0, d, 3);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double per_cu = 1024.0 * 4 * N_IT * 8 / 256;          // wave-instructions per CU
    printf("%-34s %8.3f ms   %6.2f ns per wave-instruction and CU\n", name, ms, ms * 1e6 / per_cu);
    hipFree(d);
}
int main()
{
    run<0>("ds_write_b32"); run<1>("ds_write_b16"); run<2>("ds_write_b8"); run<10>("ds_write_b16_d16_hi"); run<3>("ds_write_b64"); run<4>("ds_write_b128");
    run<13>("ds_write_b16, consecutive halves"); run<12>("ds_write_b8, consecutive bytes");
    run<5>("ds_read_b32"); run<6>("ds_read_u16"); run<7>("ds_read_u8"); run<11>("ds_read_u8, consecutive bytes"); run<8>("ds_read_b64"); run<9>("ds_read_b128");
    return 0;
}
