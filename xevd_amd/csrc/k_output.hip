// k_output.hip - the output side of a decoded picture: conformance-window crop, bit-depth conversion and tight packing on the
// device, so that one contiguous (and, for 8-bit output, half-sized) buffer crosses PCIe instead of three padded 16-bit planes.
//
// Replaces xevd_pull's hand-off (src_base/xevd.c:2042-2071, crop fields :2058-2069) followed by the application's
// imgb_cpy_codec_to_out (app/xevd_app_util.h:656-700) with its conversions:
//   to 8 bit           : (v + (1 << (shift-1))) >> shift, clipped to [0,255], one byte per sample      (:465-494)
//   to a lower depth   : the same rounding shift, clipped to [0, 2^dst - 1], 16 bit                    (:519-552)
//   to a higher depth  : v << shift                                                                    (:496-517)
//   same depth         : copy
// Optional DRA post-filter first (xevd_apply_filter, src_main/xevdm.c:3305-3349; sample processing xevdm_dra.c:272-355): Cb / Cr
// scaled around 512 by a factor looked up with the unmapped co-located luma sample, luma through its inverse table.
// Output layout = what imgb_write puts in the .yuv file: Y, U, V planes back to back, rows without padding.
//
// One workgroup per output row and plane (like k_pad); a thread converts 4 neighbouring samples per step: 8-byte reads at the
// 2-byte-aligned crop position (gfx950 runs in unaligned-access mode), 4- or 8-byte writes.  HBM-bound, no reuse.
#include "xgpu_internal.h"

struct __attribute__((packed, aligned(2))) S16x4u { int16_t a, b, c, d; };
struct __attribute__((packed, aligned(1))) U8x4u  { uint8_t a, b, c, d; };
struct __attribute__((packed, aligned(2))) U16x4u { uint16_t a, b, c, d; };

struct OutArgs {
    const int16_t *src[3];          // first sample of the cropped area of every plane
    int      s[3], w[3], h[3];
    int      row_start[4];
    size_t   dst_off[3];            // byte offset of every plane in dst
    uint8_t *dst;
    int      shift;                 // > 0: rounding right shift, < 0: left shift by -shift, 0: copy
    int      out8;                  // one byte per sample
    int      maxv;
    const int32_t *dra;             // [3][1024] luma / Cb / Cr inverse tables, or NULL
};

__device__ __forceinline__ int conv1(int v, int shift, int maxv, int out8)
{
    if (out8) return min(max((v + (shift ? 1 << (shift - 1) : 0)) >> shift, 0), 255);            // signed samples (:464-494)
    if (shift > 0) return min(((int)(uint16_t)v + (1 << (shift - 1))) >> shift, maxv);            // unsigned samples (:519-552)
    return shift < 0 ? (int)(uint16_t)(v << -shift) : v;
}

// v: the plane's sample; luma: the unmapped luma sample at (2y, 2x) for a chroma plane
__device__ __forceinline__ int dra1(const int32_t *lut, int c, int v, int luma)
{
    if (c == 0) return (int)(int16_t)lut[min(max(v, 0), 1023)];
    const int sv = v - 512;
    int off = (abs(sv) * lut[c * 1024 + min(max(luma, 0), 1023)] + (1 << 8)) >> 9;
    if (sv < 0) off = -off;
    return (int)(int16_t)(512 + off);
}

__global__ __launch_bounds__(256) void k_output(const OutArgs p)
{
    const int gr = blockIdx.x;
    const int c = gr < p.row_start[1] ? 0 : (gr < p.row_start[2] ? 1 : 2);
    const int r = gr - p.row_start[c], w = p.w[c];
    const int16_t *src = p.src[c] + (size_t)r * p.s[c];
    uint8_t *dst = p.dst + p.dst_off[c] + (size_t)r * w * (p.out8 ? 1 : 2);
    const int shift = p.shift, maxv = p.maxv;
    const int w4 = w >> 2;
    const int32_t *dra = p.dra;
    const int16_t *lrow = p.src[0] + (size_t)(2 * r) * p.s[0];     // the luma row a chroma row's DRA scale comes from
    for (int i = threadIdx.x; i < w4; i += 256) {
        S16x4u v = ((const S16x4u *)src)[i];
        if (dra) {
            if (c == 0) { v.a = (int16_t)dra1(dra, 0, v.a, 0); v.b = (int16_t)dra1(dra, 0, v.b, 0); v.c = (int16_t)dra1(dra, 0, v.c, 0); v.d = (int16_t)dra1(dra, 0, v.d, 0); }
            else {
                const int16_t *l = lrow + 8 * i;
                v.a = (int16_t)dra1(dra, c, v.a, l[0]); v.b = (int16_t)dra1(dra, c, v.b, l[2]); v.c = (int16_t)dra1(dra, c, v.c, l[4]); v.d = (int16_t)dra1(dra, c, v.d, l[6]);
            }
        }
        const int a = conv1(v.a, shift, maxv, p.out8), b = conv1(v.b, shift, maxv, p.out8), cc = conv1(v.c, shift, maxv, p.out8), d = conv1(v.d, shift, maxv, p.out8);
        if (p.out8) { U8x4u o = { (uint8_t)a, (uint8_t)b, (uint8_t)cc, (uint8_t)d }; ((U8x4u *)dst)[i] = o; }
        else        { U16x4u o = { (uint16_t)a, (uint16_t)b, (uint16_t)cc, (uint16_t)d }; ((U16x4u *)dst)[i] = o; }
    }
    for (int i = (w4 << 2) + threadIdx.x; i < w; i += 256) {        // widths that are not a multiple of 4 (cropped chroma)
        int sv = src[i];
        if (dra) sv = dra1(dra, c, sv, c ? lrow[2 * i] : 0);
        const int a = conv1(sv, shift, maxv, p.out8);
        if (p.out8) dst[i] = (uint8_t)a; else ((uint16_t *)dst)[i] = (uint16_t)a;
    }
}

// raw16: the samples as they are, two bytes each whatever the coding depth (the bytes the picture signature is made of: xevd_md5_imgb hashes w x 2 bytes per row)
void launch_output(xgpu_ctx *c, const DevPic &pic, const int32_t *d_dra, int out_bd, int crop_l, int crop_r, int crop_t, int crop_b, uint8_t *d_dst, bool raw16)
{
    if (raw16) out_bd = c->sp.bit_depth_luma;
    OutArgs p;
    const int16_t *pl[3] = { pic.y, pic.u, pic.v };
    const int bps = (out_bd == 8 && !raw16) ? 1 : 2;
    int rows = 0;
    size_t off = 0;
    for (int i = 0; i < 3; i++) {
        const int sh = i ? 1 : 0;
        p.s[i] = i ? pic.s_c : pic.s_l;
        p.w[i] = (c->sp.width - crop_l - crop_r) >> sh;
        p.h[i] = (c->sp.height - crop_t - crop_b) >> sh;
        p.src[i] = pl[i] + (size_t)(crop_t >> sh) * p.s[i] + (crop_l >> sh);
        p.row_start[i] = rows;
        rows += p.h[i];
        p.dst_off[i] = off;
        off += (size_t)p.w[i] * p.h[i] * bps;
    }
    p.row_start[3] = rows;
    p.dst = d_dst;
    const int src_bd = c->sp.bit_depth_luma;
    p.shift = src_bd - out_bd;
    p.out8 = out_bd == 8 && !raw16;
    p.maxv = (1 << out_bd) - 1;
    p.dra = d_dra;
    hipLaunchKernelGGL(k_output, dim3(rows), dim3(256), 0, c->stream, p);
}
