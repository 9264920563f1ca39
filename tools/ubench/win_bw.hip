// win_bw.hip - what the memory system gives k_inter's access pattern with no arithmetic at all: every wave reads a (TW+7) x (TH+7) sample window of two
// "reference pictures" at its tile position plus a pseudo-random vector, and writes its TW x TH tile of a third picture.  Tile shapes from 32x32 (k_inter's wave
// tile) to 256x4 vary the length of the contiguous runs per row; the picture is 7680 x 6480 16-bit samples (the bytes of an 8K 4:2:0 picture).
// build: hipcc --offload-arch=gfx950 -O3 -o win_bw win_bw.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <stdint.h>

#define PW 7680
#define PH 6480
#define STRIDE 8064            // samples: 7680 + 192 + 192
#define MARGIN 192
#define PADROWS 144

__device__ __forceinline__ uint32_t hash(uint32_t x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }

// one wave per TW x TH tile, 4 waves (2 x 2 tiles) per workgroup; REFS reference windows; WR: write the tile; sigma: vector range +- sigma samples (uniform)
template <int TW, int TH, int REFS, bool WR, int AX = 2>
__global__ __launch_bounds__(256) void k_win(const int16_t *r0, const int16_t *r1, int16_t *dst, int tiles_x, int tiles_y, int sigma, int strip, int xcd_map, int aln = 0)
{
    constexpr int AY = 4 / AX;                              // the workgroup's four waves as AX x AY tiles
    const int wg_x = tiles_x / AX, wg_y = tiles_y / AY, n_wg = wg_x * wg_y;
    int idx = blockIdx.x;
    if (xcd_map) idx = (blockIdx.x & 7) * (gridDim.x >> 3) + (blockIdx.x >> 3);
    if (idx >= n_wg) return;
    const int per_strip = strip * wg_y, st = idx / per_strip, ks = idx - st * per_strip;
    const int sw = min(strip, wg_x - st * strip);
    const int gy = ks / sw, gx = st * strip + (ks - gy * sw);
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int tx = gx * AX + (wave % AX), ty = gy * AY + (wave / AX);
    const int x0 = tx * TW, y0 = ty * TH;
    constexpr int WW = TW + 7, WH = TH + 7, CH = (WW * 2 + 15 + 14) / 16;      // 16-byte chunks per window row (unaligned start)
    uint4 acc = make_uint4(0, 0, 0, 0);
#pragma unroll
    for (int r = 0; r < REFS; r++) {
        const uint32_t h = hash((uint32_t)(ty * tiles_x + tx) * 2 + r);
        const int mx = sigma ? (int)(h % (2 * sigma + 1)) - sigma : 0, my = sigma ? (int)((h >> 12) % (2 * sigma + 1)) - sigma : 0;
        // aln: the window's chunks start at the 16-byte boundary below its first sample (one more chunk per row would cover it: same count here, the last one clipped)
        const int16_t *p = (r ? r1 : r0) + (size_t)(y0 + my - 3) * STRIDE + (aln ? ((x0 + mx - 3) & ~7) : x0 + mx - 3);
#pragma unroll
        for (int i = lane; i < WH * CH; i += 64) {
            const int row = i / CH, c = i - row * CH;
            const uint4 v = *(const uint4 *)(p + (size_t)row * STRIDE + 8 * c);
            acc.x ^= v.x; acc.y ^= v.y; acc.z ^= v.z; acc.w ^= v.w;
        }
    }
    if (WR) {
        constexpr int OC = TW / 4;                           // 8-byte chunks per tile row: a lane writes 4 samples of a row (k_inter: 4 rows x 8 bytes per lane)
        for (int i = lane; i < TH * OC; i += 64) {
            const int row = i / OC, c = i - row * OC;
            *(uint2 *)(dst + (size_t)(y0 + row) * STRIDE + x0 + 4 * c) = make_uint2(acc.x + i, acc.y);
        }
    } else if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345u) dst[0] = 1;
}

template <int TW, int TH, int REFS, bool WR, int AX = 2>
static void run(const char *name, const int16_t *r0, const int16_t *r1, int16_t *dst, int sigma, int strip_px, int xcd, int aln = 0)
{
    const int tiles_x = PW / TW, tiles_y = PH / TH, n_wg = (tiles_x / AX) * (tiles_y / (4 / AX));
    const int strip = strip_px / (AX * TW) > 0 ? strip_px / (AX * TW) : 1;
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    const int grid = ((n_wg + 7) >> 3) << 3;
    for (int i = 0; i < 2; i++) hipLaunchKernelGGL((k_win<TW, TH, REFS, WR, AX>), dim3(grid), dim3(256), 0, 0, r0, r1, dst, tiles_x, tiles_y, sigma, strip, xcd, aln);
    hipEventRecord(a, 0);
    const int reps = 8;
    for (int i = 0; i < reps; i++) hipLaunchKernelGGL((k_win<TW, TH, REFS, WR, AX>), dim3(grid), dim3(256), 0, 0, r0, r1, dst, tiles_x, tiles_y, sigma, strip, xcd, aln);
    hipEventRecord(b, 0); hipEventSynchronize(b);
    float ms = 0; hipEventElapsedTime(&ms, a, b);
    const double us = ms * 1e3 / reps, bytes = (double)PW * PH * 2 * (REFS + (WR ? 1 : 0));
    if (aln) name = "rw aligned";
    printf("%-10s wg %dx%d tile %3dx%-3d refs %d write %d sigma %2d strip %4dpx xcd %d : %7.1f us  %7.1f GB/s (compulsory bytes)\n", name, AX, 4 / AX, TW, TH, REFS, (int)WR, sigma, strip_px, xcd, us, bytes / us / 1e3);
    fflush(stdout);
}


// k_inter's wave tile (32x32) with SEQ horizontally adjacent tiles per WAVE, one after the other (grid / SEQ workgroups).  OVERLAP = false: the loads of tile i + 1 are
// issued behind the stores of tile i (what a plain loop in k_inter would do); true: all windows are requested first (what a 64-wide wave tile does to the memory system).
template <int SEQ, bool OVERLAP>
__global__ __launch_bounds__(256) void k_seq(const int16_t *r0, const int16_t *r1, int16_t *dst, int tiles_x, int tiles_y, int sigma, int strip, int xcd_map)
{
    constexpr int TW = 32, TH = 32;
    const int wg_x = tiles_x / (2 * SEQ), wg_y = tiles_y / 2, n_wg = wg_x * wg_y;
    int idx = blockIdx.x;
    if (xcd_map) idx = (blockIdx.x & 7) * (gridDim.x >> 3) + (blockIdx.x >> 3);
    if (idx >= n_wg) return;
    const int per_strip = strip * wg_y, st = idx / per_strip, ks = idx - st * per_strip;
    const int sw = min(strip, wg_x - st * strip);
    const int gy = ks / sw, gx = st * strip + (ks - gy * sw);
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int ty = gy * 2 + (wave >> 1);
    constexpr int WW = TW + 7, WH = TH + 7, CH = (WW * 2 + 15 + 14) / 16, NL = (WH * CH + 63) / 64;
    uint4 v[OVERLAP ? SEQ : 1][2][NL];
    uint4 acc[SEQ];
#pragma unroll
    for (int q = 0; q < SEQ; q++) {
        const int tx = (gx * 2 + (wave & 1)) * SEQ + q, x0 = tx * TW, y0 = ty * TH;
#pragma unroll
        for (int r = 0; r < 2; r++) {
            const uint32_t h = hash((uint32_t)(ty * tiles_x + tx) * 2 + r);
            const int mx = sigma ? (int)(h % (2 * sigma + 1)) - sigma : 0, my = sigma ? (int)((h >> 12) % (2 * sigma + 1)) - sigma : 0;
            const int16_t *p = (r ? r1 : r0) + (size_t)(y0 + my - 3) * STRIDE + x0 + mx - 3;
#pragma unroll
            for (int k = 0; k < NL; k++) {
                const int i = lane + 64 * k, row = min(i / CH, WH - 1), c = i - (i / CH) * CH;
                v[OVERLAP ? q : 0][r][k] = *(const uint4 *)(p + (size_t)row * STRIDE + 8 * c);
            }
        }
        if (!OVERLAP) {
            uint4 a = make_uint4(0, 0, 0, 0);
#pragma unroll
            for (int r = 0; r < 2; r++)
#pragma unroll
                for (int k = 0; k < NL; k++) { a.x ^= v[0][r][k].x; a.y ^= v[0][r][k].y; a.z ^= v[0][r][k].z; a.w ^= v[0][r][k].w; }
            acc[q] = a;
            for (int i = lane; i < TH * (TW / 4); i += 64) { const int row = i / (TW / 4), c = i - row * (TW / 4); *(uint2 *)(dst + (size_t)(y0 + row) * STRIDE + x0 + 4 * c) = make_uint2(a.x + i, a.y); }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
    }
    if (OVERLAP) {
#pragma unroll
        for (int q = 0; q < SEQ; q++) {
            const int tx = (gx * 2 + (wave & 1)) * SEQ + q, x0 = tx * TW, y0 = ty * TH;
            uint4 a = make_uint4(0, 0, 0, 0);
#pragma unroll
            for (int r = 0; r < 2; r++)
#pragma unroll
                for (int k = 0; k < NL; k++) { a.x ^= v[q][r][k].x; a.y ^= v[q][r][k].y; a.z ^= v[q][r][k].z; a.w ^= v[q][r][k].w; }
            for (int i = lane; i < TH * (TW / 4); i += 64) { const int row = i / (TW / 4), c = i - row * (TW / 4); *(uint2 *)(dst + (size_t)(y0 + row) * STRIDE + x0 + 4 * c) = make_uint2(a.x + i, a.y); }
        }
    }
    (void)acc;
}
template <int SEQ, bool OVERLAP>
static void run_seq(const int16_t *r0, const int16_t *r1, int16_t *dst, int sigma, int strip_px)
{
    const int tiles_x = PW / 32, tiles_y = PH / 32, n_wg = (tiles_x / (2 * SEQ)) * (tiles_y / 2);
    const int strip = strip_px / (64 * SEQ) > 0 ? strip_px / (64 * SEQ) : 1;
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    const int grid = ((n_wg + 7) >> 3) << 3;
    for (int i = 0; i < 2; i++) hipLaunchKernelGGL((k_seq<SEQ, OVERLAP>), dim3(grid), dim3(256), 0, 0, r0, r1, dst, tiles_x, tiles_y, sigma, strip, 1);
    hipEventRecord(a, 0);
    const int reps = 8;
    for (int i = 0; i < reps; i++) hipLaunchKernelGGL((k_seq<SEQ, OVERLAP>), dim3(grid), dim3(256), 0, 0, r0, r1, dst, tiles_x, tiles_y, sigma, strip, 1);
    hipEventRecord(b, 0); hipEventSynchronize(b);
    float ms = 0; hipEventElapsedTime(&ms, a, b);
    const double us = ms * 1e3 / reps, bytes = (double)PW * PH * 2 * 3;
    printf("seq        32x32 tiles, %d per wave one after the other, windows %s: %7.1f us  %7.1f GB/s\n", SEQ, OVERLAP ? "all requested first" : "tile by tile", us, bytes / us / 1e3);
    fflush(stdout);
}

int main(int argc, char **argv)
{
    const bool all = argc > 1;
    const size_t elems = (size_t)STRIDE * (PH + 2 * PADROWS);
    int16_t *buf[3];
    for (int i = 0; i < 3; i++) { if (hipMalloc(&buf[i], elems * 2 + 4096) != hipSuccess) { printf("alloc\n"); return 1; } hipMemset(buf[i], i, elems * 2); }
    const int16_t *r0 = buf[0] + (size_t)PADROWS * STRIDE + MARGIN, *r1 = buf[1] + (size_t)PADROWS * STRIDE + MARGIN;
    int16_t *d = buf[2] + (size_t)PADROWS * STRIDE + MARGIN;
    if (all) {
    for (int sigma : { 0, 16 }) {
        run<32, 32, 2, true>("rw", r0, r1, d, sigma, 1024, 1);
        run<64, 16, 2, true>("rw", r0, r1, d, sigma, 1024, 1);
        run<128, 8, 2, true>("rw", r0, r1, d, sigma, 1024, 1);
        run<256, 4, 2, true>("rw", r0, r1, d, sigma, 1024, 1);
        run<64, 32, 2, true>("rw", r0, r1, d, sigma, 1024, 1);
        run<64, 64, 2, true>("rw", r0, r1, d, sigma, 1024, 1);
        run<128, 32, 2, true>("rw", r0, r1, d, sigma, 1024, 1);
    }
    run<32, 32, 2, false>("read", r0, r1, d, 16, 1024, 1);
    run<128, 8, 2, false>("read", r0, r1, d, 16, 1024, 1);
    run<32, 32, 1, true>("rw1", r0, r1, d, 16, 1024, 1);
    run<32, 32, 1, false>("read1", r0, r1, d, 16, 1024, 1);
    run<32, 32, 2, true>("rw", r0, r1, d, 16, 1024, 0);
    run<32, 32, 2, true>("rw", r0, r1, d, 16, 512, 1);
    run<32, 32, 2, true>("rw", r0, r1, d, 16, 7680, 1);
    run<32, 32, 2, true>("rw", r0, r1, d, 16, 256, 1);
    }
    run<32, 32, 2, true, 2>("rw", r0, r1, d, 16, 1024, 1, 1);
    run<64, 32, 2, true, 2>("rw", r0, r1, d, 16, 1024, 1, 1);
    run<32, 32, 2, false, 2>("read", r0, r1, d, 16, 1024, 1, 0);
    run<32, 32, 2, false, 2>("read", r0, r1, d, 16, 1024, 1, 1);
    // the workgroup's shape with k_inter's 32x32 wave tile, and the wider wave tiles
    run<32, 32, 2, true, 2>("rw", r0, r1, d, 16, 1024, 1);
    run<32, 32, 2, true, 4>("rw", r0, r1, d, 16, 1024, 1);
    run<32, 32, 2, true, 1>("rw", r0, r1, d, 16, 1024, 1);
    run<64, 32, 2, true, 2>("rw", r0, r1, d, 16, 1024, 1);
    run<64, 32, 2, true, 4>("rw", r0, r1, d, 16, 1024, 1);
    run<64, 32, 2, true, 1>("rw", r0, r1, d, 16, 1024, 1);
    run<64, 16, 2, true, 4>("rw", r0, r1, d, 16, 1024, 1);
    run<64, 16, 2, true, 1>("rw", r0, r1, d, 16, 1024, 1);
    run<128, 32, 2, true, 2>("rw", r0, r1, d, 16, 2048, 1);
    run<32, 32, 2, true, 4>("rw", r0, r1, d, 16, 2048, 1);
    run<32, 32, 2, true, 4>("rw", r0, r1, d, 16, 7680, 1);
    run_seq<1, false>(r0, r1, d, 16, 1024);
    run_seq<2, false>(r0, r1, d, 16, 1024);
    run_seq<2, true>(r0, r1, d, 16, 1024);
    run_seq<4, false>(r0, r1, d, 16, 1024);
    run_seq<4, true>(r0, r1, d, 16, 1024);
    printf("%s\n", hipGetErrorString(hipDeviceSynchronize()));
    return 0;
}
