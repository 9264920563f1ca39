// md5_probe.hip - how fast is ONE MD5 chain on the device?  The picture signature (xevd_md5_imgb, src_base/xevd_util.c:985-1002) is one MD5 per plane over the plane's
// 16-bit samples: a serial chain of 64-byte blocks (66 MB of luma at 8K = 1.04 M blocks), which one lane has to walk alone - the other 16 383 wave slots cannot help.
// Prints the rate of one lane (the message words come from memory through a 64-lane coalesced load + readlane, the best a wave can do for one chain), and checks the
// digest against the host's.   hipcc --offload-arch=gfx950 -O3 md5_probe.hip -o md5_probe && ./md5_probe
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <vector>

static const uint32_t KH[64] = {
    0xd76aa478,0xe8c7b756,0x242070db,0xc1bdceee,0xf57c0faf,0x4787c62a,0xa8304613,0xfd469501,0x698098d8,0x8b44f7af,0xffff5bb1,0x895cd7be,0x6b901122,0xfd987193,0xa679438e,0x49b40821,
    0xf61e2562,0xc040b340,0x265e5a51,0xe9b6c7aa,0xd62f105d,0x02441453,0xd8a1e681,0xe7d3fbc8,0x21e1cde6,0xc33707d6,0xf4d50d87,0x455a14ed,0xa9e3e905,0xfcefa3f8,0x676f02d9,0x8d2a4c8a,
    0xfffa3942,0x8771f681,0x6d9d6122,0xfde5380c,0xa4beea44,0x4bdecfa9,0xf6bb4b60,0xbebfbc70,0x289b7ec6,0xeaa127fa,0xd4ef3085,0x04881d05,0xd9d4d039,0xe6db99e5,0x1fa27cf8,0xc4ac5665,
    0xf4292244,0x432aff97,0xab9423a7,0xfc93a039,0x655b59c3,0x8f0ccc92,0xffeff47d,0x85845dd1,0x6fa87e4f,0xfe2ce6e0,0xa3014314,0x4e0811a1,0xf7537e82,0xbd3af235,0x2ad7d2bb,0xeb86d391 };
static const int SH[64] = { 7,12,17,22,7,12,17,22,7,12,17,22,7,12,17,22, 5,9,14,20,5,9,14,20,5,9,14,20,5,9,14,20, 4,11,16,23,4,11,16,23,4,11,16,23,4,11,16,23, 6,10,15,21,6,10,15,21,6,10,15,21,6,10,15,21 };
__constant__ uint32_t dK[64];

#define ROTL(x, s) (((x) << (s)) | ((x) >> (32 - (s))))
template <class GET> __host__ __device__ inline void md5_block(uint32_t h[4], GET m, const uint32_t *K)
{
    uint32_t a = h[0], b = h[1], c = h[2], d = h[3];
#pragma unroll
    for (int i = 0; i < 64; i++) {
        uint32_t f; int g;
        if (i < 16) { f = d ^ (b & (c ^ d)); g = i; } else if (i < 32) { f = c ^ (d & (b ^ c)); g = (5 * i + 1) & 15; } else if (i < 48) { f = b ^ c ^ d; g = (3 * i + 5) & 15; } else { f = c ^ (b | ~d); g = (7 * i) & 15; }
        constexpr int S[64] = { 7,12,17,22,7,12,17,22,7,12,17,22,7,12,17,22, 5,9,14,20,5,9,14,20,5,9,14,20,5,9,14,20, 4,11,16,23,4,11,16,23,4,11,16,23,4,11,16,23, 6,10,15,21,6,10,15,21,6,10,15,21,6,10,15,21 };
        const uint32_t t = a + f + K[i] + m(g);
        a = d; d = c; c = b; b = b + ROTL(t, S[i]);
    }
    h[0] += a; h[1] += b; h[2] += c; h[3] += d;
}
// one wave, one chain: every step the wave loads 4 blocks (64 lanes x 4 bytes = 256 B), then walks them one after the other on scalar-broadcast words
__global__ void k_md5(const uint32_t *msg, size_t n_blocks, uint32_t *out)
{
    uint32_t h[4] = { 0x67452301, 0xefcdab89, 0x98badcfe, 0x10325476 };
    const int lane = threadIdx.x;
    for (size_t b0 = 0; b0 < n_blocks; b0 += 4) {
        const uint32_t w = msg[b0 * 16 + lane];
#pragma unroll
        for (int k = 0; k < 4; k++)
            md5_block(h, [&](int g) { return (uint32_t)__builtin_amdgcn_readlane((int)w, k * 16 + g); }, dK);
    }
    if (lane == 0) for (int i = 0; i < 4; i++) out[i] = h[i];
}
int main()
{
    const size_t n_blocks = 1 << 16;                        // 4 MB
    std::vector<uint32_t> m(n_blocks * 16);
    for (size_t i = 0; i < m.size(); i++) m[i] = (uint32_t)(i * 2654435761u);
    uint32_t hh[4] = { 0x67452301, 0xefcdab89, 0x98badcfe, 0x10325476 };
    auto t0 = std::chrono::steady_clock::now();
    for (size_t b = 0; b < n_blocks; b++) { const uint32_t *p = &m[b * 16]; md5_block(hh, [&](int g) { return p[g]; }, KH); }
    const double host_s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    uint32_t *dm, *dout;
    hipMalloc(&dm, m.size() * 4); hipMalloc(&dout, 16);
    hipMemcpy(dm, m.data(), m.size() * 4, hipMemcpyHostToDevice);
    hipMemcpyToSymbol(HIP_SYMBOL(dK), KH, sizeof(KH));
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k_md5, dim3(1), dim3(64), 0, 0, dm, n_blocks, dout);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k_md5, dim3(1), dim3(64), 0, 0, dm, n_blocks, dout);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    uint32_t o[4]; hipMemcpy(o, dout, 16, hipMemcpyDeviceToHost);
    const double mb = n_blocks * 64 / 1e6;
    printf("one MD5 chain over %.1f MB (compression function only, no padding block): device %.1f MB/s (%.2f ms), host core %.1f MB/s; states %s\n", mb, mb / (ms * 1e-3), ms, mb / host_s,
           memcmp(o, hh, 16) == 0 ? "equal" : "DIFFER");
    printf("an 8K 4:2:0 picture is three chains over 66.4 + 16.6 + 16.6 MB: the luma chain alone would take %.0f ms on the device, %.0f ms on one host core\n", 66.4 / (mb / (ms * 1e-3)) * 1e3, 66.4 / (mb / host_s) * 1e3);
    return 0;
}
