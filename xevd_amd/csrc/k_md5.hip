// k_md5.hip - the picture signature on the device: one MD5 per plane over the plane's 16-bit samples, row by row (xevd_md5_imgb, src_base/xevd_util.c:985-1002;
// xevd_md5_update / _finish :905-983; checked against the SEI by xevd_picbuf_check_signature, :1557-1572, on the DRA-mapped copy when the PPS names a DRA parameter
// set, src_main/xevdm.c:3256-3287).
//
// An MD5 is ONE serial chain of 64-byte blocks per plane (1.04 M blocks for the luma plane of an 8K picture): nothing of a GPU's width applies to it, and a lone
// wave issues one instruction every ~4 cycles, so the rate of a chain is instructions per step x 4 cycles.  What the device can do is (a) keep a step at five
// instructions - the round function in one v_bitop3_b32, the message word and the round constant added off the chain (v_add3_u32 with a literal), v_alignbit_b32
// as the rotation - and (b) walk the THREE chains of a picture in the three lanes of one wave, at the price of one.  Measured (tools/md5_rate.py): 76 - 87 MB/s on the
// luma chain (tools/ubench/md5_probe.hip's scalar-broadcast form: 54 MB/s) - 48 ms for a 1080p picture, 217 ms at 4K, 0.87 s at 8K, where one host thread hashes the
// three planes in 6.8 / 28 / 114 ms.  It costs the host nothing and the device three lanes, but it does not make `-s` decoding faster than host threads do; DESIGN 7
// says when it pays.
//
// The message of a plane is the plane's rows without padding - exactly the bytes k_output packs at the coding depth (16-bit samples also for 8-bit pictures): the
// caller packs first (launch_output raw16) and this kernel reads three contiguous messages.
#include "xgpu_internal.h"

struct Md5Args { const uint8_t *msg; unsigned long long off[3], len[3]; uint32_t *digest; };      // digest[3][4]: A, B, C, D of every plane (the 16 bytes, little endian)

struct __attribute__((packed, aligned(2))) Md5Quad { uint32_t a, b, c, d; };      // a 16-byte piece of the message at its 2-byte aligned address

__device__ __forceinline__ uint32_t md5_rotl(uint32_t x, int s) { return __builtin_amdgcn_alignbit(x, x, 32 - s); }

// the compression function (RFC 1321) on the lane's chain; m[16] in registers, every index and constant a literal after unrolling
__device__ __forceinline__ void md5_block(uint32_t h[4], const uint32_t m[16])
{
    constexpr uint32_t K[64] = {
        0xd76aa478,0xe8c7b756,0x242070db,0xc1bdceee,0xf57c0faf,0x4787c62a,0xa8304613,0xfd469501,0x698098d8,0x8b44f7af,0xffff5bb1,0x895cd7be,0x6b901122,0xfd987193,0xa679438e,0x49b40821,
        0xf61e2562,0xc040b340,0x265e5a51,0xe9b6c7aa,0xd62f105d,0x02441453,0xd8a1e681,0xe7d3fbc8,0x21e1cde6,0xc33707d6,0xf4d50d87,0x455a14ed,0xa9e3e905,0xfcefa3f8,0x676f02d9,0x8d2a4c8a,
        0xfffa3942,0x8771f681,0x6d9d6122,0xfde5380c,0xa4beea44,0x4bdecfa9,0xf6bb4b60,0xbebfbc70,0x289b7ec6,0xeaa127fa,0xd4ef3085,0x04881d05,0xd9d4d039,0xe6db99e5,0x1fa27cf8,0xc4ac5665,
        0xf4292244,0x432aff97,0xab9423a7,0xfc93a039,0x655b59c3,0x8f0ccc92,0xffeff47d,0x85845dd1,0x6fa87e4f,0xfe2ce6e0,0xa3014314,0x4e0811a1,0xf7537e82,0xbd3af235,0x2ad7d2bb,0xeb86d391 };
    constexpr int S[64] = { 7,12,17,22,7,12,17,22,7,12,17,22,7,12,17,22, 5,9,14,20,5,9,14,20,5,9,14,20,5,9,14,20, 4,11,16,23,4,11,16,23,4,11,16,23,4,11,16,23, 6,10,15,21,6,10,15,21,6,10,15,21,6,10,15,21 };
    uint32_t a = h[0], b = h[1], c = h[2], d = h[3];
#pragma unroll
    for (int i = 0; i < 64; i++) {
        uint32_t f;
        int g;
        if (i < 16)      { f = d ^ (b & (c ^ d)); g = i; }
        else if (i < 32) { f = c ^ (d & (b ^ c)); g = (5 * i + 1) & 15; }
        else if (i < 48) { f = b ^ c ^ d;         g = (3 * i + 5) & 15; }
        else             { f = c ^ (b | ~d);      g = (7 * i) & 15; }
        const uint32_t t = (a + m[g] + K[i]) + f;               // (the bracket does not wait for the chain: a is four steps old)
        a = d; d = c; c = b; b = b + md5_rotl(t, S[i]);
    }
    h[0] += a; h[1] += b; h[2] += c; h[3] += d;
}

__global__ __launch_bounds__(64) void k_md5_planes(const Md5Args p)
{
    const int lane = threadIdx.x;
    if (lane >= 3) return;                                       // three chains, one per lane: the wave's instruction stream advances all of them together
    const uint8_t *msg = p.msg + p.off[lane];
    const unsigned long long len = p.len[lane], n_full = len >> 6;
    uint32_t h[4] = { 0x67452301u, 0xefcdab89u, 0x98badcfeu, 0x10325476u };
    uint32_t m[16], nx[16];
    auto fetch = [&](unsigned long long b, uint32_t w[16]) {
        const Md5Quad *q = (const Md5Quad *)(msg + (b << 6));
#pragma unroll
        for (int k = 0; k < 4; k++) { const Md5Quad v = q[k]; w[4 * k] = v.a; w[4 * k + 1] = v.b; w[4 * k + 2] = v.c; w[4 * k + 3] = v.d; }
    };
    if (n_full) fetch(0, nx);
    for (unsigned long long b = 0; b < n_full; b++) {             // (the chroma lanes leave the loop after a quarter of the luma lane's blocks)
#pragma unroll
        for (int k = 0; k < 16; k++) m[k] = nx[k];
        if (b + 1 < n_full) fetch(b + 1, nx);                     // the next block travels while this one is hashed
        md5_block(h, m);
    }
    // the tail (xevd_md5_finish): the remaining bytes - an even number, the samples are 16 bit -, 0x80, zeros up to 56 mod 64, the length in bits as 64 bits
    const int rem = (int)(len & 63);
    uint32_t w[32];
#pragma unroll
    for (int k = 0; k < 32; k++) w[k] = 0;
    const uint8_t *t = msg + (n_full << 6);
    for (int k = 0; k < rem; k += 2) {
        const uint32_t s = *(const uint16_t *)(t + k);
#pragma unroll
        for (int j = 0; j < 16; j++) if (j == (k >> 2)) w[j] |= s << ((k & 2) * 8);
    }
#pragma unroll
    for (int j = 0; j < 16; j++) if (j == (rem >> 2)) w[j] |= 0x80u << ((rem & 3) * 8);
    const bool two = rem >= 56;
    const unsigned long long bits = len << 3;
#pragma unroll
    for (int j = 0; j < 2; j++) {
        const uint32_t v = j ? (uint32_t)(bits >> 32) : (uint32_t)bits;
        if (two) w[30 + j] = v; else w[14 + j] = v;
    }
    md5_block(h, w);
    if (two) md5_block(h, w + 16);
#pragma unroll
    for (int k = 0; k < 4; k++) p.digest[lane * 4 + k] = h[k];
}

// planes packed back to back at d_msg (launch_output raw16): w x h, then two of (w / 2) x (h / 2), two bytes per sample
void launch_md5(xgpu_ctx *c, hipStream_t s, const uint8_t *d_msg, int w, int h, uint32_t *d_digest)
{
    Md5Args p;
    p.msg = d_msg; p.digest = d_digest;
    const unsigned long long ly = (unsigned long long)w * h * 2, lc = (unsigned long long)(w >> 1) * (h >> 1) * 2;
    p.off[0] = 0; p.off[1] = ly; p.off[2] = ly + lc;
    p.len[0] = ly; p.len[1] = lc; p.len[2] = lc;
    hipLaunchKernelGGL(k_md5_planes, dim3(1), dim3(64), 0, s, p);
}
