// copy_bw.hip - which plain device copy reaches what on this part (the yardstick of bench.py's roofline.measured_copy_bw_gbps).
// build: hipcc --offload-arch=gfx950 -O3 -o copy_bw copy_bw.hip        run: ./copy_bw [MiB]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

// U independent 16-byte loads per lane before the first store; NT: non-temporal loads and stores; the grid covers the buffer exactly (no grid-stride loop) when G == 0
template <int U, bool NT>
__global__ __launch_bounds__(256) void k(const uint4 *__restrict__ src, uint4 *__restrict__ dst, size_t n)
{
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride * U) {
        uint4 v[U];
#pragma unroll
        for (int u = 0; u < U; u++)
            if (i + u * stride < n) {
                if (NT) { const unsigned long long *p = (const unsigned long long *)(src + i + u * stride);
                          const unsigned long long a = __builtin_nontemporal_load(p), b = __builtin_nontemporal_load(p + 1);
                          v[u] = make_uint4((unsigned)a, (unsigned)(a >> 32), (unsigned)b, (unsigned)(b >> 32)); }
                else v[u] = src[i + u * stride];
            }
#pragma unroll
        for (int u = 0; u < U; u++)
            if (i + u * stride < n) {
                if (NT) { unsigned long long *p = (unsigned long long *)(dst + i + u * stride);
                          __builtin_nontemporal_store(((unsigned long long)v[u].y << 32) | v[u].x, p);
                          __builtin_nontemporal_store(((unsigned long long)v[u].w << 32) | v[u].z, p + 1); }
                else dst[i + u * stride] = v[u];
            }
    }
}
// a workgroup owns a contiguous chunk (1 KB per wave and round) instead of a grid-strided one
template <int U>
__global__ __launch_bounds__(256) void kc(const uint4 *__restrict__ src, uint4 *__restrict__ dst, size_t n)
{
    const size_t per = (n + gridDim.x - 1) / gridDim.x, b0 = per * blockIdx.x, b1 = b0 + per < n ? b0 + per : n;
    for (size_t i = b0 + threadIdx.x; i < b1; i += 256 * U) {
        uint4 v[U];
#pragma unroll
        for (int u = 0; u < U; u++) if (i + u * 256 < b1) v[u] = src[i + u * 256];
#pragma unroll
        for (int u = 0; u < U; u++) if (i + u * 256 < b1) dst[i + u * 256] = v[u];
    }
}
// read-only and write-only halves, to see which side bounds the copy
__global__ __launch_bounds__(256) void kr(const uint4 *__restrict__ src, uint4 *__restrict__ dst, size_t n)
{
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    uint4 acc = make_uint4(0, 0, 0, 0);
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) { const uint4 v = src[i]; acc.x ^= v.x; acc.y ^= v.y; acc.z ^= v.z; acc.w ^= v.w; }
    if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678u) dst[0] = acc;
}
__global__ __launch_bounds__(256) void kw(const uint4 *__restrict__ src, uint4 *__restrict__ dst, size_t n)
{
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) dst[i] = make_uint4((unsigned)i, 1, 2, 3);
}

typedef void (*kern_t)(const uint4 *, uint4 *, size_t);
static void run(const char *name, kern_t f, int grid, const uint4 *s, uint4 *d, size_t n, double bytes_per_elem)
{
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    for (int i = 0; i < 3; i++) hipLaunchKernelGGL(f, dim3(grid), dim3(256), 0, 0, s, d, n);
    hipEventRecord(a, 0);
    const int reps = 10;
    for (int i = 0; i < reps; i++) hipLaunchKernelGGL(f, dim3(grid), dim3(256), 0, 0, s, d, n);
    hipEventRecord(b, 0); hipEventSynchronize(b);
    float ms = 0; hipEventElapsedTime(&ms, a, b);
    printf("%-28s grid %6d  %8.1f GB/s\n", name, grid, bytes_per_elem * n * reps / (ms * 1e-3) / 1e9);
    fflush(stdout);
}

int main(int argc, char **argv)
{
    const size_t mib = argc > 1 ? atoi(argv[1]) : 1024, bytes = mib << 20, n = bytes / 16;
    uint4 *s, *d;
    if (hipMalloc(&s, bytes) != hipSuccess || hipMalloc(&d, bytes) != hipSuccess) { printf("alloc failed\n"); return 1; }
    hipMemset(s, 1, bytes); hipMemset(d, 2, bytes);
    printf("buffer %zu MiB (read + write counted)\n", mib);
    const int grids[] = { 1024, 2048, 4096, 8192, 16384, 65536 };
    for (int g : grids) {
        run("grid-stride U1", k<1, false>, g, s, d, n, 32);
        run("grid-stride U2", k<2, false>, g, s, d, n, 32);
        run("grid-stride U4", k<4, false>, g, s, d, n, 32);
        run("grid-stride U4 nontemporal", k<4, true>, g, s, d, n, 32);
        run("grid-stride U1 nontemporal", k<1, true>, g, s, d, n, 32);
        run("chunk U4", kc<4>, g, s, d, n, 32);
        run("chunk U8", kc<8>, g, s, d, n, 32);
    }
    run("exact grid U1", k<1, false>, (int)(n / 256), s, d, n, 32);
    run("exact grid U4", k<4, false>, (int)(n / 1024), s, d, n, 32);
    run("exact grid U4 nontemporal", k<4, true>, (int)(n / 1024), s, d, n, 32);
    run("read only", kr, 4096, s, d, n, 16);
    run("write only", kw, 4096, s, d, n, 16);
    hipError_t e = hipDeviceSynchronize();
    printf("%s\n", hipGetErrorString(e));
    return 0;
}
