// k_pad.hip - border replication of a decoded picture into its padding (xevd_picbuf_expand -> picbuf_expand,
// src_base/xevd_util.c:365-427), and the plain copy kernel used as the measured bandwidth roofline.
//
// One workgroup per padded row and plane.  Rows inside the active area only write their left/right margins;
// rows above/below copy the nearest active row including its margins (same result as the reference's
// row-then-full-stride-copy order).  The device margins (192/128 left, >=144/72 right, 144/72 rows) are at
// least the reference's 144/72 everywhere, so every sample an MV can legally address (xevd_mv_clip) is defined.
#include "xgpu_internal.h"

struct PadArgs { int16_t *a[3]; int s[3], w[3], h[3], ml[3], rows_pad[3]; int row_start[4]; };

__global__ __launch_bounds__(256) void k_pad(const PadArgs p)
{
    const int gr = blockIdx.x;
    const int c = gr < p.row_start[1] ? 0 : (gr < p.row_start[2] ? 1 : 2);
    const int r = gr - p.row_start[c] - p.rows_pad[c];            // row relative to the active area
    const int w = p.w[c], h = p.h[c], s = p.s[c], ml = p.ml[c];
    const int rs = min(max(r, 0), h - 1);
    const int16_t *src = p.a[c] + (size_t)rs * s;
    int16_t *dst = p.a[c] + (size_t)r * s;
    const int mr = s - ml - w;
    const int16_t lv = src[0], rv = src[w - 1];
    const uint32_t l2 = (uint16_t)lv * 0x10001u, r2 = (uint16_t)rv * 0x10001u;
    // margins (even-sized and 4-byte aligned: luma w % 8 == 0, chroma w % 4 == 0, ml and s multiples of 64)
    for (int i = threadIdx.x; i < ml / 2; i += 256) ((uint32_t *)(dst - ml))[i] = l2;
    for (int i = threadIdx.x; i < mr / 2; i += 256) ((uint32_t *)(dst + w))[i] = r2;
    if (r != rs)
        for (int i = threadIdx.x; i < w / 4; i += 256) ((uint2 *)dst)[i] = ((const uint2 *)src)[i];   // w % 4 == 0 (chroma of w % 8 == 0)
}

void launch_pad(xgpu_ctx *c, const DevPic &pic)
{
    PadArgs p;
    int16_t *pl[3] = { pic.y, pic.u, pic.v };
    int rows = 0;
    for (int i = 0; i < 3; i++) {
        p.a[i] = pl[i];
        p.s[i] = i ? pic.s_c : pic.s_l;
        p.w[i] = i ? c->sp.width >> 1 : c->sp.width;
        p.h[i] = i ? c->sp.height >> 1 : c->sp.height;
        p.ml[i] = i ? XGPU_MARGIN_C : XGPU_MARGIN_L;
        p.rows_pad[i] = i ? XGPU_PAD_C : XGPU_PAD_L;
        p.row_start[i] = rows;
        rows += p.h[i] + 2 * p.rows_pad[i];
    }
    p.row_start[3] = rows;
    hipLaunchKernelGGL(k_pad, dim3(rows), dim3(256), 0, c->stream, p);
}

// The bandwidth yardstick of bench.py (roofline.measured_copy_bw_gbps): a float4 copy, ONE 16-byte element per lane, as many workgroups as the buffer has 4 KB
// pieces.  tools/ubench/copy_bw.hip (round 4, 1 GiB): this form 6.18 TB/s - the guide's 6.29 TB/s figure for the part; a grid-stride loop over 2048 workgroups
// (the yardstick of rounds 1-3) 4.8 TB/s, over 1024 workgroups 5.6, four loads per lane before the first store 4.3-5.3, non-temporal accesses 4.2-5.8;
// read-only 6.39 TB/s, write-only 3.87 TB/s.
__global__ __launch_bounds__(256) void k_copy(const uint4 *__restrict__ src, uint4 *__restrict__ dst, size_t n)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) dst[i] = src[i];
}
void launch_copy_bw(xgpu_ctx *c, const void *src, void *dst, size_t bytes)
{
    hipLaunchKernelGGL(k_copy, dim3((unsigned)((bytes / 16 + 255) / 256)), dim3(256), 0, c->stream, (const uint4 *)src, (uint4 *)dst, bytes / 16);
}
