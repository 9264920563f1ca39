/*
 * ref_decode.c - OUR driver around the REAL reference decoder's public API (inc/xevd.h:369-373: xevd_create / xevd_decode /
 * xevd_pull / xevd_delete), built into the executable oracle/_ref/ref_decode against libxevdb_ref.so (Baseline) by
 * oracle/Makefile.ref - a process per decode: the reference keeps process-global tables and corrupts the heap when several
 * decoders of different picture sizes live in one process.  Test infrastructure:
 * decodes a length-prefixed .evc byte string to 16-bit planar pictures in output order - the frame-level oracle of
 * tests/test_stream.py and the `-m N` CPU baseline of bench.py's stream mode.  The calling sequence is the one of the
 * reference's own sample application (app/xevd_app.c:455-600): one NAL unit per xevd_decode, xevd_pull after every VCL NAL
 * unit, then pull until the decoder reports that nothing is left.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include "xevd.h"

#ifdef REFD_HIP
/* ref_binding.c: the MI355X backend installed behind the reference decoder's function-table slots */
int refb_install(void *id);
int refb_failed(void *id);
void refb_uninstall(void *id);
#define REFD_DELETE(id) do { refb_uninstall(id); xevd_delete(id); } while (0)
#else
#define REFD_DELETE(id) xevd_delete(id)
#endif

/* out: [max_pics][h*w + 2*(h/2)*(w/2)] s16, Y then U then V per picture; returns the number of pictures or a negative error */
int refd_decode(const uint8_t *bytes, size_t size, int threads, int16_t *out, int max_pics, int w, int h)
{
    XEVD_CDSC cdsc;
    XEVD id;
    XEVD_BITB bitb;
    XEVD_STAT stat;
    XEVD_IMGB *imgb;
    size_t pos = 0;
    int n = 0, ret, bumping = 0, idle = 0;
    const size_t pic_elems = (size_t)w * h * 3 / 2;

    memset(&cdsc, 0, sizeof(cdsc));
    cdsc.threads = threads;
    id = xevd_create(&cdsc, &ret);
    if (!id) return -1000 + ret;
#ifdef REFD_HIP
    if (refb_install(id)) { xevd_delete(id); return -1001; }
#endif
    {   /* picture-signature SEIs are VERIFIED (app/xevd_app.c:177-182): a stream that carries our MD5s makes the reference check them */
        int on = 1, sz = sizeof(int);
        (void)xevd_config(id, XEVD_CFG_SET_USE_PIC_SIGNATURE, &on, &sz);
    }
    for (;;) {
        memset(&stat, 0, sizeof(stat));
        stat.fnum = -1;
        if (!bumping) {
            if (pos + 4 > size) { bumping = 1; continue; }
            const size_t len = ((size_t)bytes[pos] << 24) | ((size_t)bytes[pos + 1] << 16) | ((size_t)bytes[pos + 2] << 8) | bytes[pos + 3];
            if (pos + 4 + len > size) { REFD_DELETE(id); return -2; }
            memset(&bitb, 0, sizeof(bitb));
            bitb.addr = (void *)(bytes + pos + 4);
            bitb.ssize = (int)len;
            pos += 4 + len;
            ret = xevd_decode(id, &bitb, &stat);
            if (XEVD_FAILED(ret)) { REFD_DELETE(id); return -3000 + ret; }
        }
        if (stat.fnum < 0 && !bumping) continue;
        imgb = NULL;
        ret = xevd_pull(id, &imgb);
        if (ret == XEVD_ERR_UNEXPECTED) break;                 /* bumping completed */
        if (XEVD_FAILED(ret)) { REFD_DELETE(id); return -4000 + ret; }
        /* a stream that ends inside a sub-GOP leaves pictures the reference never outputs (it waits for the missing POC for ever) */
        if (bumping && !imgb && ++idle > 64) break;
        if (imgb) {
            idle = 0;
            if (n < max_pics) {
                int16_t *dst = out + (size_t)n * pic_elems;
                int c, r;
                for (c = 0; c < 3; c++) {
                    const int pw = c ? w / 2 : w, ph = c ? h / 2 : h;
                    for (r = 0; r < ph; r++)
                        memcpy(dst + (size_t)r * pw, (const uint8_t *)imgb->a[c] + (size_t)r * imgb->s[c], sizeof(int16_t) * pw);
                    dst += (size_t)pw * ph;
                }
            }
            n++;
            imgb->release(imgb);
        }
    }
#ifdef REFD_HIP
    if (refb_failed(id)) n = -5000;
#endif
    REFD_DELETE(id);
    return n;
}

#ifdef REFD_MAIN
/* ref_decode <in.evc> <out.raw> <w> <h> <threads> [repeat]: one decoder per process (the reference keeps process-global tables);
   prints "<pictures> <seconds of the last repeat>" */
#include <stdio.h>
#include <time.h>
int main(int argc, char **argv)
{
    if (argc < 6) return 2;
    FILE *f = fopen(argv[1], "rb");
    if (!f) return 3;
    fseek(f, 0, SEEK_END);
    const long size = ftell(f);
    fseek(f, 0, SEEK_SET);
    uint8_t *buf = (uint8_t *)malloc((size_t)size);
    if (fread(buf, 1, (size_t)size, f) != (size_t)size) return 4;
    fclose(f);
    const int w = atoi(argv[3]), h = atoi(argv[4]), threads = atoi(argv[5]), repeat = argc > 6 ? atoi(argv[6]) : 1;
    const int max_pics = 4096;
    const size_t elems = (size_t)w * h * 3 / 2;
    /* output of the first 64 pictures only (tests); timing runs decode everything */
    int16_t *out = (int16_t *)malloc(sizeof(int16_t) * elems * 64);
    int n = 0, r;
    double secs = 0;
    for (r = 0; r < repeat; r++) {
        struct timespec t0, t1;
        clock_gettime(CLOCK_MONOTONIC, &t0);
        n = refd_decode(buf, (size_t)size, threads, out, 64, w, h);
        clock_gettime(CLOCK_MONOTONIC, &t1);
        secs = (double)(t1.tv_sec - t0.tv_sec) + 1e-9 * (double)(t1.tv_nsec - t0.tv_nsec);
        if (n < 0) { fprintf(stderr, "refd_decode -> %d\n", n); return 5; }
    }
    (void)max_pics;
    if (strcmp(argv[2], "-") != 0) {
        f = fopen(argv[2], "wb");
        if (!f) return 6;
        fwrite(out, sizeof(int16_t), elems * (size_t)(n < 64 ? n : 64), f);
        fclose(f);
    }
    fprintf(stderr, "%d %.6f\n", n, secs);
    return 0;
}
#endif
