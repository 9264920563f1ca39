// split_pat.hip - the load pattern of k_inter's per-lane path (tiles of 8x8 / 4x4 CUs with their own vectors) against a cooperative one, no arithmetic: what the
// vector-memory pipeline (TA / TCP / TD) charges for each.  One wave per 32x32 luma tile of a 7680x4320 picture, every 8x8 block with its own pseudo-random vector.
//   per-lane : lane = 4x4 SCU, 11 rows x (16 + 8 bytes) of its own 11x11 window                     (k_inter today: 22 load instructions per list)
//   block    : the 4 lanes of an 8x8 block share its 15x15 window: 15 rows x 2 chunks of 16 bytes = 30 chunks, 8 per lane (8 load instructions)
//   block16  : the 16 lanes of a 16x16 block share its 23x23 window: 23 rows x 3 chunks = 69 chunks, 5 per lane
// build: hipcc --offload-arch=gfx950 -O3 -o split_pat split_pat.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#define PW 7680
#define PH 4320
#define STRIDE 8064
__device__ __forceinline__ uint32_t hash(uint32_t x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }
__device__ __forceinline__ void mvof(int bx, int by, int sigma, int &mx, int &my) { const uint32_t h = hash((uint32_t)(by * 4096 + bx)); mx = (int)(h % (2 * sigma + 1)) - sigma; my = (int)((h >> 12) % (2 * sigma + 1)) - sigma; }

template <int MODE>
__global__ __launch_bounds__(256) void k(const int16_t *ref, int16_t *dst, int sigma)
{
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int tiles_x = PW / 32, tile = blockIdx.x * 4 + wave;
    const int tx = tile % tiles_x, ty = tile / tiles_x;
    if (ty >= PH / 32) return;
    uint4 acc = make_uint4(0, 0, 0, 0);
    if (MODE == 0) {
        const int sx = tx * 8 + (lane & 7), sy = ty * 8 + (lane >> 3);
        int mx, my; mvof(sx >> 1, sy >> 1, sigma, mx, my);
        const int16_t *p = ref + (size_t)(sy * 4 + my - 3) * STRIDE + sx * 4 + mx - 3;
        uint4 a[11]; uint2 b[11];
#pragma unroll
        for (int r = 0; r < 11; r++) { a[r] = *(const uint4 *)(p + (size_t)r * STRIDE); b[r] = *(const uint2 *)(p + (size_t)r * STRIDE + 8); }
#pragma unroll
        for (int r = 0; r < 11; r++) { acc.x ^= a[r].x ^ b[r].x; acc.y ^= a[r].y ^ b[r].y; acc.z ^= a[r].z; acc.w ^= a[r].w; }
    } else if (MODE == 1) {
        const int blk = lane >> 2, q = lane & 3;                  // 16 blocks of 8x8 in the tile, 4 lanes each
        const int bx = tx * 4 + (blk & 3), by = ty * 4 + (blk >> 2);
        int mx, my; mvof(bx, by, sigma, mx, my);
        const int16_t *p = ref + (size_t)(by * 8 + my - 3) * STRIDE + bx * 8 + mx - 3;
        uint4 a[8];
#pragma unroll
        for (int i = 0; i < 8; i++) { const int c = q + 4 * i, row = c >> 1, k2 = c & 1; a[i] = row < 15 ? *(const uint4 *)(p + (size_t)row * STRIDE + 8 * k2) : make_uint4(0, 0, 0, 0); }
#pragma unroll
        for (int i = 0; i < 8; i++) { acc.x ^= a[i].x; acc.y ^= a[i].y; acc.z ^= a[i].z; acc.w ^= a[i].w; }
    } else {
        const int blk = lane >> 4, q = lane & 15;                 // 4 blocks of 16x16, 16 lanes each
        const int bx = tx * 2 + (blk & 1), by = ty * 2 + (blk >> 1);
        int mx, my; mvof(bx * 2, by * 2, sigma, mx, my);
        const int16_t *p = ref + (size_t)(by * 16 + my - 3) * STRIDE + bx * 16 + mx - 3;
        uint4 a[5];
#pragma unroll
        for (int i = 0; i < 5; i++) { const int c = q + 16 * i, row = c / 3, k3 = c - row * 3; a[i] = row < 23 ? *(const uint4 *)(p + (size_t)row * STRIDE + 8 * k3) : make_uint4(0, 0, 0, 0); }
#pragma unroll
        for (int i = 0; i < 5; i++) { acc.x ^= a[i].x; acc.y ^= a[i].y; acc.z ^= a[i].z; acc.w ^= a[i].w; }
    }
    if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x1234567u) dst[0] = 1;
}
template <int MODE> static void run(const char *name, const int16_t *ref, int16_t *dst, int sigma)
{
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    const int grid = (PW / 32) * (PH / 32) / 4;
    for (int i = 0; i < 2; i++) hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(256), 0, 0, ref, dst, sigma);
    hipEventRecord(a, 0);
    for (int i = 0; i < 8; i++) hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(256), 0, 0, ref, dst, sigma);
    hipEventRecord(b, 0); hipEventSynchronize(b);
    float ms = 0; hipEventElapsedTime(&ms, a, b);
    printf("%-9s sigma %2d : %7.1f us per 8K luma plane\n", name, sigma, ms * 1e3 / 8); fflush(stdout);
}
int main()
{
    int16_t *ref, *dst;
    const size_t bytes = (size_t)STRIDE * (PH + 288) * 2 + 4096;
    hipMalloc(&ref, bytes); hipMalloc(&dst, 4096); hipMemset(ref, 1, bytes);
    const int16_t *r = ref + (size_t)144 * STRIDE + 192;
    for (int sigma : { 0, 16 }) { run<0>("per-lane", r, dst, sigma); run<1>("block 8", r, dst, sigma); run<2>("block 16", r, dst, sigma); }
    printf("%s\n", hipGetErrorString(hipDeviceSynchronize()));
    return 0;
}
