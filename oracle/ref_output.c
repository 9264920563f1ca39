/*
 * ref_output.c - TEST INFRASTRUCTURE ONLY: a harness around the reference APPLICATION's output conversion, compiled against the
 * reference sources where they lie (oracle/Makefile.ref -> oracle/_ref/libref_output.so; nothing is copied).
 * It wraps tightly packed 16-bit planes in XEVD_IMGBs and runs the real imgb_cpy_codec_to_out (app/xevd_app_util.h:665-708),
 * which is what xevd_app does with every pulled picture before imgb_write.  Checks xgpu_pic_output and orc_output_convert.
 */
#include <xevd.h>
#include "xevd_app_util.h"

/* src: 3 tight s16 planes (w x h, w/2 x h/2 twice); dst: tight planes, bytes when dst_bd == 8, else 16 bit */
int refh_output_convert(const short *y, const short *u, const short *v, int w, int h, int src_bd, int dst_bd, void *dst)
{
    XEVD_IMGB s, d;
    int i;
    unsigned char *o = (unsigned char *)dst;
    const int bps = dst_bd == 8 ? 1 : 2;
    memset(&s, 0, sizeof(s)); memset(&d, 0, sizeof(d));
    s.cs = XEVD_CS_SET(XEVD_CF_YCBCR420, src_bd, 0); d.cs = XEVD_CS_SET(XEVD_CF_YCBCR420, dst_bd, 0);
    s.np = d.np = 3;
    s.a[0] = (void *)y; s.a[1] = (void *)u; s.a[2] = (void *)v;
    for (i = 0; i < 3; i++) {
        s.w[i] = d.w[i] = i ? w >> 1 : w; s.h[i] = d.h[i] = i ? h >> 1 : h;
        s.s[i] = s.w[i] * 2; d.s[i] = d.w[i] * bps;
        d.a[i] = o; o += d.s[i] * d.h[i];
    }
    imgb_cpy_codec_to_out(&d, &s);
    return 0;
}
