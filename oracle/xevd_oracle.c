/*
 * xevd_oracle.c - plain-C CPU restatement of the reference's per-CU reconstruction path.
 *
 * TEST INFRASTRUCTURE ONLY (see xevd_oracle.h).  Parity status: PINNED against the reference built in
 * oracle/_ref (tests/test_oracle_vs_ref.py) and against tests/golden/*.npz.
 *
 * Written from the reference's behaviour, not copied from it: FIRs are generic tap loops, the inverse
 * transform is an exact integer matrix product with tables generated from the closed form, the deblocking
 * driver walks the CU list.  Every function names the reference lines it must agree with bit for bit.
 */
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include "xevd_oracle.h"

#define CLIP3(lo, hi, v) ((v) < (lo) ? (lo) : ((v) > (hi) ? (hi) : (v)))
#define MAX_CU 128
#define ORC_PI 3.14159265358979323846

/* ------------------------------------------------------------------------------------------------
 * interpolation filter tables.  Baseline: src_base/xevd_mc.c:80-134 (only phases 0,4,8,12 / 0,4,..,28 are
 * populated).  Main (sps_admvp_flag): src_main/xevdm_mc.c:121-175, 16 luma / 32 chroma phases.
 * ---------------------------------------------------------------------------------------------- */
static const int16_t k_luma_base[4][8] = {          /* phase = row*4 */
    { 0, 0,   0, 64,  0,   0, 0, 0 },
    { 0, 1,  -5, 52, 20,  -5, 1, 0 },
    { 0, 2, -10, 40, 40, -10, 2, 0 },
    { 0, 1,  -5, 20, 52,  -5, 1, 0 },
};
static const int16_t k_chroma_base[8][4] = {        /* phase = row*4 */
    {  0, 64,  0,  0 }, { -2, 58, 10, -2 }, { -4, 52, 20, -4 }, { -6, 46, 30, -6 },
    { -8, 40, 40, -8 }, { -6, 30, 46, -6 }, { -4, 20, 52, -4 }, { -2, 10, 58, -2 },
};
static const int16_t k_luma_main[16][8] = {
    {  0, 0,   0, 64,  0,   0,  0,  0 }, {  0, 1,  -3, 63,  4,  -2,  1,  0 },
    { -1, 2,  -5, 62,  8,  -3,  1,  0 }, { -1, 3,  -8, 60, 13,  -4,  1,  0 },
    { -1, 4, -10, 58, 17,  -5,  1,  0 }, { -1, 4, -11, 52, 26,  -8,  3, -1 },
    { -1, 3,  -9, 47, 31, -10,  4, -1 }, { -1, 4, -11, 45, 34, -10,  4, -1 },
    { -1, 4, -11, 40, 40, -11,  4, -1 }, { -1, 4, -10, 34, 45, -11,  4, -1 },
    { -1, 4, -10, 31, 47,  -9,  3, -1 }, { -1, 3,  -8, 26, 52, -11,  4, -1 },
    {  0, 1,  -5, 17, 58, -10,  4, -1 }, {  0, 1,  -4, 13, 60,  -8,  3, -1 },
    {  0, 1,  -3,  8, 62,  -5,  2, -1 }, {  0, 1,  -2,  4, 63,  -3,  1,  0 },
};
static const int16_t k_chroma_main[32][4] = {
    {  0, 64,  0,  0 }, { -1, 63,  2,  0 }, { -2, 62,  4,  0 }, { -2, 60,  7, -1 },
    { -2, 58, 10, -2 }, { -3, 57, 12, -2 }, { -4, 56, 14, -2 }, { -4, 55, 15, -2 },
    { -4, 54, 16, -2 }, { -5, 53, 18, -2 }, { -6, 52, 20, -2 }, { -6, 49, 24, -3 },
    { -6, 46, 28, -4 }, { -5, 44, 29, -4 }, { -4, 42, 30, -4 }, { -4, 39, 33, -4 },
    { -4, 36, 36, -4 }, { -4, 33, 39, -4 }, { -4, 30, 42, -4 }, { -4, 29, 44, -5 },
    { -4, 28, 46, -6 }, { -3, 24, 49, -6 }, { -2, 20, 52, -6 }, { -2, 18, 53, -5 },
    { -2, 16, 54, -4 }, { -2, 15, 55, -4 }, { -2, 14, 56, -4 }, { -2, 12, 57, -3 },
    { -2, 10, 58, -2 }, { -1,  7, 60, -2 }, {  0,  4, 62, -2 }, {  0,  2, 63, -1 },
};

static void luma_taps(int phase, int admvp, int16_t t[8])
{
    int k;
    if (admvp) { for (k = 0; k < 8; k++) t[k] = k_luma_main[phase][k]; return; }
    /* baseline table rows other than 0,4,8,12 are all-zero (xevd_mc.c:83-97) */
    for (k = 0; k < 8; k++) t[k] = (phase & 3) ? 0 : k_luma_base[phase >> 2][k];
}
static void chroma_taps(int phase, int admvp, int16_t t[4])
{
    int k;
    if (admvp) { for (k = 0; k < 4; k++) t[k] = k_chroma_main[phase][k]; return; }
    for (k = 0; k < 4; k++) t[k] = (phase & 3) ? 0 : k_chroma_base[phase >> 2][k];
}

/*
 * Generic separable FIR with the reference's four rounding regimes (ntap = 8 luma / 4 chroma):
 *   00: copy                                              xevd_mc.c:169-188 / 290-309
 *   n0: (sum) >> 6, no rounding offset, clip              :190-212 / 311-333   (MAC_SFT_N0 6, MAC_ADD_N0 0, xevd_mc.h:34-38)
 *   0n: same, vertical                                    :215-237 / 335-358
 *   nn: stage1 (sum) >> min(4,bd-8) stored as s16 (wraps), stage2 (sum + 2^(s2-1)) >> s2, s2 = max(8,20-bd), clip
 *                                                         :240-284 / 360-408
 * prec = fractional bits of gmv (4 luma, 5 chroma).
 */
static void fir_block(const int16_t *ref, int gmv_x, int gmv_y, int s_ref, int s_pred, int16_t *pred, int w, int h,
                      int bd, int has_dx, int has_dy, int ntap, int prec, const int16_t *tx, const int16_t *ty)
{
    const int half = ntap / 2 - 1;              /* 3 luma, 1 chroma */
    const int ix = gmv_x >> prec, iy = gmv_y >> prec;
    const int maxv = (1 << bd) - 1;
    int i, j, k;

    if (!has_dx && !has_dy) {
        const int16_t *r = ref + iy * s_ref + ix;
        for (i = 0; i < h; i++) for (j = 0; j < w; j++) pred[i * s_pred + j] = r[i * s_ref + j];
    } else if (has_dx && !has_dy) {
        const int16_t *r = ref + iy * s_ref + ix - half;
        for (i = 0; i < h; i++) for (j = 0; j < w; j++) {
            int32_t s = 0;
            for (k = 0; k < ntap; k++) s += tx[k] * r[i * s_ref + j + k];
            s >>= 6;
            pred[i * s_pred + j] = (int16_t)CLIP3(0, maxv, s);
        }
    } else if (!has_dx && has_dy) {
        const int16_t *r = ref + (iy - half) * s_ref + ix;
        for (i = 0; i < h; i++) for (j = 0; j < w; j++) {
            int32_t s = 0;
            for (k = 0; k < ntap; k++) s += ty[k] * r[(i + k) * s_ref + j];
            s >>= 6;
            pred[i * s_pred + j] = (int16_t)CLIP3(0, maxv, s);
        }
    } else {
        const int16_t *r = ref + (iy - half) * s_ref + ix - half;
        const int shift1 = bd - 8 < 4 ? bd - 8 : 4;
        const int shift2 = 20 - bd > 8 ? 20 - bd : 8;
        const int32_t off2 = 1 << (shift2 - 1);
        int16_t *tmp = (int16_t *)malloc(sizeof(int16_t) * (size_t)(h + ntap - 1) * w);
        for (i = 0; i < h + ntap - 1; i++) for (j = 0; j < w; j++) {
            int32_t s = 0;
            for (k = 0; k < ntap; k++) s += tx[k] * r[i * s_ref + j + k];
            tmp[i * w + j] = (int16_t)(s >> shift1);          /* s16 store: wraps like the reference's buffer */
        }
        for (i = 0; i < h; i++) for (j = 0; j < w; j++) {
            int32_t s = 0;
            for (k = 0; k < ntap; k++) s += ty[k] * tmp[(i + k) * w + j];
            s = (s + off2) >> shift2;
            pred[i * s_pred + j] = (int16_t)CLIP3(0, maxv, s);
        }
        free(tmp);
    }
}

void orc_mc_l(const int16_t *ref, int gmv_x, int gmv_y, int s_ref, int s_pred, int16_t *pred, int w, int h,
              int bit_depth, int has_dx, int has_dy, int admvp)
{
    int16_t tx[8], ty[8];
    luma_taps(gmv_x & 15, admvp, tx);
    luma_taps(gmv_y & 15, admvp, ty);
    fir_block(ref, gmv_x, gmv_y, s_ref, s_pred, pred, w, h, bit_depth, has_dx, has_dy, 8, 4, tx, ty);
}

void orc_mc_c(const int16_t *ref, int gmv_x, int gmv_y, int s_ref, int s_pred, int16_t *pred, int w, int h,
              int bit_depth, int has_dx, int has_dy, int admvp)
{
    int16_t tx[4], ty[4];
    chroma_taps(gmv_x & 31, admvp, tx);
    chroma_taps(gmv_y & 31, admvp, ty);
    fir_block(ref, gmv_x, gmv_y, s_ref, s_pred, pred, w, h, bit_depth, has_dx, has_dy, 4, 5, tx, ty);
}

/* xevd_mv_clip, src_base/xevd_mc.c:435-467: block start clipped to [-128, pic-1+128] in quarter-pel units */
static void mv_clip(int x, int y, int pic_w, int pic_h, int w, int h, const int8_t refi[2], const int16_t mv[2][2], int16_t mv_t[2][2])
{
    const int min_x = -(MAX_CU << 2), min_y = -(MAX_CU << 2);
    const int max_x = (pic_w - 1 + MAX_CU) << 2, max_y = (pic_h - 1 + MAX_CU) << 2;
    int l;
    x <<= 2; y <<= 2; w <<= 2; h <<= 2;
    for (l = 0; l < 2; l++) {
        mv_t[l][0] = mv[l][0]; mv_t[l][1] = mv[l][1];
        if (refi[l] < 0) continue;
        if (x + mv[l][0] < min_x) mv_t[l][0] = (int16_t)(min_x - x);
        if (y + mv[l][1] < min_y) mv_t[l][1] = (int16_t)(min_y - y);
        if (x + mv[l][0] + w - 4 > max_x) mv_t[l][0] = (int16_t)(max_x - x - w + 4);
        if (y + mv[l][1] + h - 4 > max_y) mv_t[l][1] = (int16_t)(max_y - y - h + 4);
    }
}

int orc_mc_cu(const xgpu_seq_params *sp, const orc_frame *fr, int x, int y, int w, int h,
              const int8_t refi[2], const int16_t mv[2][2], int16_t *pred0[3], int16_t *pred1[3])
{
    int16_t mv_t[2][2];
    int16_t **dst[2] = { pred0, pred1 };
    int bidx = 0, l, i;
    const int wc = w >> 1, hc = h >> 1;      /* 4:2:0 */

    mv_clip(x, y, sp->width, sp->height, w, h, refi, mv, mv_t);

    for (l = 0; l < 2; l++) {
        const orc_pic *rp;
        int gx, gy, ldx, ldy, cdx, cdy;
        if (refi[l] < 0) continue;
        if (l == 1 && refi[0] >= 0) {
            /* identical-motion early-out, xevd_mc.c:512-519 (POC equality of the two reference pictures and
               equality of the CLIPPED vectors) */
            if (fr->refp[refi[0]][0].poc == fr->refp[refi[1]][1].poc &&
                mv_t[0][0] == mv_t[1][0] && mv_t[0][1] == mv_t[1][1]) break;
        }
        rp = &fr->refp[refi[l]][l];
        gx = ((x << 2) + mv_t[l][0]) << 2;      /* 1/16-pel luma units, xevd_mc.c:498-502 */
        gy = ((y << 2) + mv_t[l][1]) << 2;
        /* filter variant from the UNCLIPPED vector (xevd_mc.h:61-74): any of the low 4 (luma) / 5 (chroma)
           bits of mv<<2 set */
        ldx = ((mv[l][0] << 2) & 15) != 0;  ldy = ((mv[l][1] << 2) & 15) != 0;
        cdx = ((mv[l][0] << 2) & 31) != 0;  cdy = ((mv[l][1] << 2) & 31) != 0;
        orc_mc_l(rp->y, gx, gy, rp->s_l, w, dst[bidx][0], w, h, sp->bit_depth_luma, ldx, ldy, sp->tool_admvp);
        orc_mc_c(rp->u, gx, gy, rp->s_c, wc, dst[bidx][1], wc, hc, sp->bit_depth_chroma, cdx, cdy, sp->tool_admvp);
        orc_mc_c(rp->v, gx, gy, rp->s_c, wc, dst[bidx][2], wc, hc, sp->bit_depth_chroma, cdx, cdy, sp->tool_admvp);
        bidx++;
    }
    if (bidx == 2) {
        /* xevd_average_16b_no_clip, xevd_mc.c:145-167: average of the two already-clipped predictions */
        for (i = 0; i < w * h; i++)   pred0[0][i] = (int16_t)((pred0[0][i] + pred1[0][i] + 1) >> 1);
        for (i = 0; i < wc * hc; i++) pred0[1][i] = (int16_t)((pred0[1][i] + pred1[1][i] + 1) >> 1);
        for (i = 0; i < wc * hc; i++) pred0[2][i] = (int16_t)((pred0[2][i] + pred1[2][i] + 1) >> 1);
    }
    return bidx;
}

/* ------------------------------------------------------------------------------------------------
 * inverse transform.  xevd_tbl_tm{2..64} (src_base/xevd_tbl.c:89-243) equal
 * round(64*sqrt(2)*cos((2n+1)k*pi/2N)) with row 0 = 64 (checked entry-by-entry against the reference's
 * exported tables in tests/test_oracle_vs_ref.py); the partial butterflies of xevd_itdq.c:48-461 are an
 * evaluation order of the plain products dst[n] = sum_k tm[k][n]*src[k] and exact in integers.
 * ---------------------------------------------------------------------------------------------- */
static int8_t g_tm[7][64 * 64];
static int g_tm_ready = 0;
static void tm_init(void)
{
    int l, k, n;
    if (g_tm_ready) return;
    for (l = 1; l <= 6; l++) {
        const int N = 1 << l;
        for (k = 0; k < N; k++) for (n = 0; n < N; n++) {
            double v = k == 0 ? 64.0 : 64.0 * sqrt(2.0) * cos((2 * n + 1) * k * ORC_PI / (2.0 * N));
            g_tm[l][k * N + n] = (int8_t)(v >= 0 ? floor(v + 0.5) : -floor(-v + 0.5));
        }
    }
    g_tm_ready = 1;
}
const int8_t *orc_tm(int log2n) { tm_init(); return g_tm[log2n]; }

/* first stage: columns, s16 -> s32, no shift, output transposed (dst[j*N+n]); xevd_itx_pb*b step 0 */
void orc_itx_pass0(const int16_t *src, int32_t *dst, int log2n, int line)
{
    const int N = 1 << log2n; const int8_t *tm = orc_tm(log2n);
    int j, n, k;
    for (j = 0; j < line; j++) for (n = 0; n < N; n++) {
        int64_t s = 0;
        for (k = 0; k < N; k++) s += (int64_t)tm[k * N + n] * src[k * line + j];
        if (s > 2147483647LL) s = 2147483647LL;             /* ITX_CLIP_32, xevd_itdq.c:41-45 */
        if (s < -2147483647LL - 1) s = -2147483647LL - 1;
        dst[j * N + n] = (int32_t)s;
    }
}
/* second stage: s32 -> s16, (sum + 2^(shift-1)) >> shift, clip to s16; xevd_itx_pb*b step 1 */
void orc_itx_pass1(const int32_t *src, int16_t *dst, int log2n, int line, int shift)
{
    const int N = 1 << log2n; const int8_t *tm = orc_tm(log2n);
    const int64_t add = shift == 0 ? 0 : (int64_t)1 << (shift - 1);
    int j, n, k;
    for (j = 0; j < line; j++) for (n = 0; n < N; n++) {
        int64_t s = 0;
        for (k = 0; k < N; k++) s += (int64_t)tm[k * N + n] * src[k * line + j];
        s = (s + add) >> shift;
        dst[j * N + n] = (int16_t)CLIP3(-32768, 32767, s);
    }
}
/* IQT stage (src_main/xevdm_itdq.c:423-706): s16 -> s16 with rounding shift and clip after every stage, 32-bit sums */
static void itx_iqt(const int16_t *src, int16_t *dst, int log2n, int line, int shift)
{
    const int N = 1 << log2n; const int8_t *tm = orc_tm(log2n);
    const int32_t add = shift == 0 ? 0 : 1 << (shift - 1);
    int j, n, k;
    for (j = 0; j < line; j++) for (n = 0; n < N; n++) {
        int32_t s = 0;
        for (k = 0; k < N; k++) s += tm[k * N + n] * src[k * line + j];
        s = (s + add) >> shift;
        dst[j * N + n] = (int16_t)CLIP3(-32768, 32767, s);
    }
}

void orc_itdq(int16_t *coef, int log2w, int log2h, int qp, int bit_depth, int iqt)
{
    /* scale tables xevd_tbl_dq_scale / _b, src_base/xevd_tbl.c:255-256; selection xevdm_itdq.c:848-855, xevd_itdq.c:594 */
    static const int scale_main[6] = { 40, 45, 51, 57, 64, 72 };
    static const int scale_base[6] = { 40, 45, 51, 57, 64, 71 };
    const int n = 1 << (log2w + log2h);
    const int scale = (iqt ? scale_main : scale_base)[qp % 6] << (qp / 6);
    const int odd = (log2w + log2h) & 1;
    /* xevd_itdq.c:511-515: tr_shift = 15 - bd - log2_size; shift = 20 - 14 - tr_shift (+8 non-square) */
    const int tr_shift = 15 - bit_depth - ((log2w + log2h) >> 1);
    const int shift = 20 - 14 - tr_shift + (odd ? 8 : 0);
    const int64_t offset = shift == 0 ? 0 : (int64_t)1 << (shift - 1);
    const int64_t mul = (int64_t)scale * (odd ? 181 : 1);
    int i;
    for (i = 0; i < n; i++) {                                   /* xevd_dquant, xevd_itdq.c:480-492 */
        int64_t lev = (coef[i] * mul + offset) >> shift;
        coef[i] = (int16_t)CLIP3(-32768, 32767, lev);
    }
    if (iqt) {                                                  /* xevdm_itrans, xevdm_itdq.c:708-716 */
        int16_t *t = (int16_t *)malloc(sizeof(int16_t) * n);
        itx_iqt(coef, t, log2h, 1 << log2w, 7);
        itx_iqt(t, coef, log2w, 1 << log2h, 12 - (bit_depth - 8));
        free(t);
    } else {                                                    /* xevd_itrans, xevd_itdq.c:473-478 */
        int32_t *t = (int32_t *)malloc(sizeof(int32_t) * n);
        orc_itx_pass0(coef, t, log2h, 1 << log2w);
        orc_itx_pass1(t, coef, log2w, 1 << log2h, 7 + 12 - (bit_depth - 8));
        free(t);
    }
}

/* ------------------------------------------------------------------------------------------------
 * ATS: DST-VII / DCT-VIII.  Matrices built like xevdm_init_multi_tbl (src_main/xevdm_itdq.c:81-119):
 *   DCT8[k][n] = (s16)(64*sqrt(N) * cos(pi(k+.5)(n+.5)/(N+.5)) * sqrt(2/(N+.5)) +- .5), DST7 with sin(pi(k+.5)(n+1)/(N+.5)),
 * in double precision.  type index: DCT8 = 0, DST7 = 1 (enum in xevdm_def.h).  Checked against the reference's
 * tables in tests/test_oracle_vs_ref.py.
 * ---------------------------------------------------------------------------------------------- */
static int16_t g_ats[2][6][32 * 32];
static int g_ats_ready = 0;
const int16_t *orc_ats_tm(int type, int log2n)
{
    if (!g_ats_ready) {
        int l, k, n;
        for (l = 1; l <= 5; l++) {
            const int N = 1 << l;
            const double s = sqrt((double)N) * 64;
            for (k = 0; k < N; k++) for (n = 0; n < N; n++) {
                double v = cos(ORC_PI * (k + 0.5) * (n + 0.5) / (N + 0.5)) * sqrt(2.0 / (N + 0.5));
                g_ats[0][l][k * N + n] = (int16_t)(s * v + (v > 0 ? 0.5 : -0.5));
                v = sin(ORC_PI * (k + 0.5) * (n + 1) / (N + 0.5)) * sqrt(2.0 / (N + 0.5));
                g_ats[1][l][k * N + n] = (int16_t)(s * v + (v > 0 ? 0.5 : -0.5));
            }
        }
        g_ats_ready = 1;
    }
    return g_ats[type][log2n];
}
/* one ATS stage: out[j*N + n] = clip16((sum_k tm[k][n]*src[k*line+j] + rnd) >> shift).  The 4-point kernels of the
   reference use a factorised form built from three entries of the first matrix row (xevdm_itdq.c:163-190, 284-312):
   the products below are that form written as a 4x4 matrix. */
static void ats_stage(const int16_t *src, int16_t *dst, int type, int log2n, int line, int shift)
{
    const int N = 1 << log2n;
    const int16_t *tm = orc_ats_tm(type, log2n);
    int16_t e4[16];
    int j, n, k;
    if (N == 4) {
        const int a = tm[0], b = tm[1], c = tm[2], d = tm[3];
        if (type == 1) {   /* DST7_B4 */
            const int16_t m[16] = { (int16_t)a, (int16_t)b, (int16_t)c, (int16_t)(b + a),   (int16_t)c, (int16_t)c, 0, (int16_t)-c,
                                    (int16_t)(a + b), (int16_t)-a, (int16_t)-c, (int16_t)b,  (int16_t)b, (int16_t)-(b + a), (int16_t)c, (int16_t)-a };
            memcpy(e4, m, sizeof(m));
        } else {           /* DCT8_B4 */
            const int16_t m[16] = { (int16_t)(d + c), (int16_t)b, (int16_t)c, (int16_t)d,   (int16_t)b, 0, (int16_t)-b, (int16_t)-b,
                                    (int16_t)c, (int16_t)-b, (int16_t)-d, (int16_t)(d + c),  (int16_t)d, (int16_t)-b, (int16_t)(d + c), (int16_t)-c };
            memcpy(e4, m, sizeof(m));
        }
        tm = e4;           /* e4[k*4+n]: coefficient k -> output n */
    }
    for (j = 0; j < line; j++) for (n = 0; n < N; n++) {
        int32_t s = 0;
        for (k = 0; k < N; k++) s += tm[k * N + n] * src[k * line + j];
        s = (s + (1 << (shift - 1))) >> shift;
        dst[j * N + n] = (int16_t)CLIP3(-32768, 32767, s);
    }
}

void orc_itdq_ats(int16_t *coef, int log2w, int log2h, int qp, int bit_depth, int iqt, int tr_v, int tr_h)
{
    static const int scale_main[6] = { 40, 45, 51, 57, 64, 72 };
    static const int scale_base[6] = { 40, 45, 51, 57, 64, 71 };
    const int n = 1 << (log2w + log2h);
    const int scale = (iqt ? scale_main : scale_base)[qp % 6] << (qp / 6);
    const int odd = (log2w + log2h) & 1;
    const int shift = 20 - 14 - (15 - bit_depth - ((log2w + log2h) >> 1)) + (odd ? 8 : 0);
    const int64_t offset = shift == 0 ? 0 : (int64_t)1 << (shift - 1);
    const int64_t mul = (int64_t)scale * (odd ? 181 : 1);
    int16_t *t = (int16_t *)malloc(sizeof(int16_t) * n);
    int i;
    for (i = 0; i < n; i++) {
        int64_t lev = (coef[i] * mul + offset) >> shift;
        coef[i] = (int16_t)CLIP3(-32768, 32767, lev);
    }
    /* xevdm_it_MxN_ats_intra, xevdm_itdq.c:406-421: shift_1st = 7, shift_2nd = 6 + 15 - 1 - bit_depth; type index DST7 = 1 - tr, i.e.
       xevd_tbl_tr_subset_intra = { DST7, DCT8 } (xevdm_tbl.c:51) maps tr 0 -> DST7, 1 -> DCT8 */
    ats_stage(coef, t, tr_v ? 0 : 1, log2h, 1 << log2w, 7);
    ats_stage(t, coef, tr_h ? 0 : 1, log2w, 1 << log2h, 20 - bit_depth);
    free(t);
}

void orc_recon(const int16_t *coef, const int16_t *pred, int is_coef, int cuw, int cuh, int s_rec, int16_t *rec, int bit_depth)
{
    const int maxv = (1 << bit_depth) - 1;
    int i, j;
    for (i = 0; i < cuh; i++) for (j = 0; j < cuw; j++) {
        /* the sum is formed in 16 bits and wraps (s16 t0, xevd_recon.c:39,60) */
        int16_t t = is_coef ? (int16_t)(coef[i * cuw + j] + pred[i * cuw + j]) : pred[i * cuw + j];
        rec[i * s_rec + j] = (int16_t)CLIP3(0, maxv, t);
    }
}

/* ------------------------------------------------------------------------------------------------
 * baseline deblocking, one 4-sample (luma) / 2-sample (chroma) edge segment.  src_base/xevd_df.c:96-289.
 * is_ver: filter ACROSS a vertical edge (samples along a row, step 1), else across a horizontal edge.
 * ---------------------------------------------------------------------------------------------- */
static void dbk_line(int16_t *p, int step, int st, int maxv, int luma)
{
    int16_t A = p[-2 * step], B = p[-step], C = p[0], D = p[step];
    int16_t d = (int16_t)((A - (B << 2) + (C << 2) - D) / 8);     /* C division: toward zero */
    int16_t abs_d = (int16_t)(d < 0 ? -d : d);
    int16_t t16 = (int16_t)((abs_d - st) << 1); if (t16 < 0) t16 = 0;
    int16_t clip = (int16_t)(abs_d - t16);     if (clip < 0) clip = 0;
    int16_t d1 = (int16_t)(d < 0 ? -clip : clip);
    B = (int16_t)(B + d1); C = (int16_t)(C - d1);
    if (luma) {
        int16_t c2 = (int16_t)(clip >> 1);
        int16_t d2 = (int16_t)CLIP3(-c2, c2, (A - D) / 4);
        A = (int16_t)(A - d2); D = (int16_t)(D + d2);
        p[-2 * step] = (int16_t)CLIP3(0, maxv, A);
        p[step]      = (int16_t)CLIP3(0, maxv, D);
    }
    p[-step] = (int16_t)CLIP3(0, maxv, B);
    p[0]     = (int16_t)CLIP3(0, maxv, C);
}
void orc_dbk_luma(int16_t *buf, int st, int stride, int bit_depth, int is_ver)
{
    int i; const int maxv = (1 << bit_depth) - 1;
    for (i = 0; i < 4; i++) dbk_line(is_ver ? buf + i * stride : buf + i, is_ver ? 1 : stride, st, maxv, 1);
}
void orc_dbk_chroma(int16_t *u, int16_t *v, int st_u, int st_v, int stride, int bit_depth, int is_ver)
{
    int i; const int maxv = (1 << bit_depth) - 1;
    for (i = 0; i < 2; i++) {
        if (st_u) dbk_line(is_ver ? u + i * stride : u + i, is_ver ? 1 : stride, st_u, maxv, 0);
        if (st_v) dbk_line(is_ver ? v + i * stride : v + i, is_ver ? 1 : stride, st_v, maxv, 0);
    }
}

/* ------------------------------------------------------------------------------------------------
 * picture level
 * ---------------------------------------------------------------------------------------------- */
#define MCU_IF(m)   (((m) >> 15) & 1)
#define MCU_QP(m)   (((m) >> 16) & 0x7F)
#define MCU_CBFL(m) (((m) >> 24) & 1)
#define MCU_IBC(m)  (((m) >> 26) & 1)      /* src_main/xevdm_def.h:325-329 */
#define MCU_COD(m)  (((m) >> 31) & 1)

/* ATS-inter TU of a CU: size (xevdm_get_tu_size, src_main/xevdm_util.c:3585-3608) and offset (get_tu_pos_offset :3610-3634) in
   luma samples; info = idx | pos<<4, idx 1/3 vertical split half/quarter width, 2/4 horizontal split */
static void ats_inter_tu(int info, int w, int h, int *tx, int *ty, int *tw, int *th)
{
    const int idx = info & 15, pos = (info >> 4) & 15;
    *tx = 0; *ty = 0; *tw = w; *th = h;
    if (idx == 0) return;
    if (idx == 2 || idx == 4) { *th = idx == 4 ? h / 4 : h / 2; *ty = pos ? h - *th : 0; }
    else { *tw = idx == 3 ? w / 4 : w / 2; *tx = pos ? w - *tw : 0; }
}

/* xevd_set_dec_info (src_base/xevd_util.c:1574-1660): every SCU of the CU receives intra flag (bit 15), QP
   (bits 16-22, core->qp = qp_y - 6*(bd-8)), skip flag (23), luma cbf (24), COD (31), refi and mv. */
#define CU_PLANES(b, i) ((b)->tree ? ((b)->tree[i] == 1 ? 1 : (b)->tree[i] == 2 ? 2 : 3) : 3)
static void set_dec_info(const xgpu_seq_params *sp, const xgpu_cu_batch *b, int i, orc_maps *m)
{
    const int xs = b->x[i] >> 2, ys = b->y[i] >> 2, ws = (1 << b->log2w[i]) >> 2, hs = (1 << b->log2h[i]) >> 2;
    const int intra = b->pred_mode[i] == XGPU_MODE_INTRA;
    const uint32_t qp = (uint32_t)(b->qp[i * 3] - 6 * (sp->bit_depth_luma - 8)) & 0x7F;
    uint32_t v = (qp << 16) | ((uint32_t)intra << 15) | (1u << 31);
    int r, c;
    if (b->pred_mode[i] == XGPU_MODE_SKIP) v |= 1u << 23;
    if (b->pred_mode[i] == XGPU_MODE_IBC) v |= 1u << 26;      /* xevdm_set_dec_info, xevdm_util.c:4289-4296 */
    /* the luma cbf flag of the map is is_coef_sub[Y_C][0] (xevd_util.c:1615): for a CU above 64 only its first 64x64 sub-block counts */
    if ((b->cbf[i] & 1) && (!(b->log2w[i] > 6 || b->log2h[i] > 6) || !b->cbf_sub || (b->cbf_sub[i] & 1))) v |= 1u << 24;
    const int ai = (!intra && b->pred_mode[i] != XGPU_MODE_IBC && b->ats_inter) ? b->ats_inter[i] : 0;
    int tx = 0, ty = 0, tw = 0, th = 0;
    if (ai) ats_inter_tu(ai, ws * 4, hs * 4, &tx, &ty, &tw, &th);
    for (r = 0; r < hs; r++) for (c = 0; c < ws; c++) {
        const int k = (ys + r) * m->w_scu + xs + c;
        m->map_scu[k] = v;
        if (ai) {      /* xevdm_set_cu_cbf_flags (xevdm_util.c:3670-3712): luma cbf only on the coded part */
            m->map_scu[k] &= ~(1u << 24);
            if ((b->cbf[i] & 1) && c * 4 >= tx && c * 4 < tx + tw && r * 4 >= ty && r * 4 < ty + th) m->map_scu[k] |= 1u << 24;
        }
        if (m->map_ats) m->map_ats[k] = (uint8_t)ai;
        if (intra) {
            m->map_refi[k * 2] = m->map_refi[k * 2 + 1] = -1;
            memset(&m->map_mv[k * 4], 0, 4 * sizeof(int16_t));
        } else if (b->pred_mode[i] == XGPU_MODE_IBC) {      /* refi -1 / -1, the block vector in list 0 (xevdm.c:1098-1110); the filters never look at it */
            m->map_refi[k * 2] = m->map_refi[k * 2 + 1] = -1;
            memset(&m->map_mv[k * 4], 0, 4 * sizeof(int16_t));
            memcpy(&m->map_mv[k * 4], &b->mv[i * 4], 2 * sizeof(int16_t));
        } else {
            m->map_refi[k * 2] = b->refi[i * 2]; m->map_refi[k * 2 + 1] = b->refi[i * 2 + 1];
            memcpy(&m->map_mv[k * 4], &b->mv[i * 4], 4 * sizeof(int16_t));
        }
    }
}

/* ------------------------------------------------------------------------------------------------
 * DMVR (Main, sps->tool_dmvr): decoder-side refinement of the two vectors of a merge-mode bi-predicted CU whose references lie at equal
 * distances on either side of the picture (xevdm_mc, src_main/xevdm_mc.c:1860-2038; processDMVR :1647-1829).  Per 16x16 sub-block: bilinear
 * pre-interpolation of both lists two samples wider than the block, up to two rounds of a 5-point SAD search with mirrored offsets
 * (xevd_DMVR_refine :1293-1339), a parametric sub-sample step from the last round's cross of costs (xevd_SubPelErrorSrfc :1373-1427),
 * then the real 8 / 4-tap interpolation at the refined sixteenth-sample vector out of a window fetched at the STARTING vector and
 * replicate-padded by 2 / 1 samples (prefetch_for_mc :1481-1544, final_paddedMC_forDMVR :1548-1644).
 * ---------------------------------------------------------------------------------------------- */
static int dmvr_clip_one(int x, int y, int pic_w, int pic_h, int w, int h, const int16_t mv[2], int16_t mv_t[2])      /* :939-980 */
{
    const int min_c = -(MAX_CU << 2), max_x = (pic_w - 1 + MAX_CU) << 2, max_y = (pic_h - 1 + MAX_CU) << 2;
    int clip = 0;
    x <<= 2; y <<= 2; w <<= 2; h <<= 2;
    mv_t[0] = mv[0]; mv_t[1] = mv[1];
    if (x + mv[0] < min_c) { clip = 1; mv_t[0] = (int16_t)(min_c - x); }
    if (y + mv[1] < min_c) { clip = 1; mv_t[1] = (int16_t)(min_c - y); }
    if (x + mv[0] + w - 4 > max_x) { clip = 1; mv_t[0] = (int16_t)(max_x - x - w + 4); }
    if (y + mv[1] + h - 4 > max_y) { clip = 1; mv_t[1] = (int16_t)(max_y - y - h + 4); }
    return clip;
}
static int dmvr_cost(int w, int h, const int16_t *a, const int16_t *b, int s)      /* xevd_DMVR_cost :1270-1291 */
{
    int sad = 0, i, j;
    for (i = 0; i < h; i++) for (j = 0; j < w; j++) sad += abs(a[i * s + j] - b[i * s + j]);
    return sad;
}
static int dmvr_div_q7(int64_t n, int64_t d)      /* div_for_maxq7 :1341-1372: three bits of n / d */
{
    int sign = 0, q = 0;
    if (n < 0) { sign = 1; n = -n; }
    d <<= 3;
    if (n >= d) { n -= d; q++; }
    q <<= 1; d >>= 1;
    if (n >= d) { n -= d; q++; }
    q <<= 1;
    if (n >= (d >> 1)) q++;
    return sign ? -q : q;
}
/* refined[k][list][x/y]: the vectors of sub-block k (raster order inside the CU) in QUARTER samples, what the decoder stores for temporal
   prediction (dmvr_mv, :1783-1797).  pred0 / pred1: the two predictions, not yet averaged. */
static void dmvr_process(const xgpu_seq_params *sp, const orc_frame *fr, int x, int y, int w, int h, const int8_t refi[2], const int16_t mv[2][2],
                         int16_t *pred0[3], int16_t *pred1[3], int16_t (*refined)[2][2])
{
    enum { IT = 2, BOTTOM = 0, TOP, RIGHT, LEFT, DIAG, CENTER = 8 };
    const int stride = w + 2 * IT, dx = w < 16 ? w : 16, dy = h < 16 ? h : 16, bd = sp->bit_depth_luma;
    int16_t start[2][2], *bl[2], **dst[2] = { pred0, pred1 };
    int l, sx, sy, num = 0;
    mv_clip(x, y, sp->width, sp->height, w, h, refi, mv, start);
    for (l = 0; l < 2; l++) {
        /* xevdm_bl_mc_l (:358-486): (w + 4) x (h + 4) samples from two samples up-left of the starting position, 2-tap { 64 - 4p, 4p } in the
           regimes of the long filters */
        const orc_pic *rp = &fr->refp[refi[l]][l];
        const int gx = ((x << 2) + start[l][0] - (IT << 2)) << 2, gy = ((y << 2) + start[l][1] - (IT << 2)) << 2;
        const int16_t tx[2] = { (int16_t)(64 - 4 * (gx & 15)), (int16_t)(4 * (gx & 15)) }, ty[2] = { (int16_t)(64 - 4 * (gy & 15)), (int16_t)(4 * (gy & 15)) };
        bl[l] = (int16_t *)malloc(sizeof(int16_t) * (size_t)stride * (h + 2 * IT));
        fir_block(rp->y, gx, gy, rp->s_l, stride, bl[l], w + 2 * IT, h + 2 * IT, bd, (gx & 15) != 0, (gy & 15) != 0, 2, 4, tx, ty);
    }
    for (sy = 0; sy < h; sy += dy) for (sx = 0; sx < w; sx += dx, num++) {
        const int16_t *c0 = bl[0] + (IT + sy) * stride + IT + sx, *c1 = bl[1] + (IT + sy) * stride + IT + sx;
        int tot[2] = { 0, 0 }, not_zero = 1, min_cost = 0, cost[9], i, k;
        int32_t ref16[2][2];
        for (k = 0; k < 9; k++) cost[k] = 0x7FFFFFFF;
        for (i = 0; i < IT; i++) {
            const int16_t *a0 = c0 + tot[0] + tot[1] * stride, *a1 = c1 - (tot[0] + tot[1] * stride);
            int ox[5] = { 0, 0, 1, -1, 0 }, oy[5] = { 1, -1, 0, 0, 0 }, d[2] = { 0, 0 }, idx;
            for (k = 0; k < 9; k++) cost[k] = 0x7FFFFFFF;
            if (i == 0) min_cost = dmvr_cost(dx, dy, a0, a1, stride);
            if ((i > 0 && min_cost == 0) || (i == 0 && min_cost < dx * dy)) { not_zero = 0; break; }
            cost[CENTER] = min_cost;
            for (idx = BOTTOM; idx <= DIAG; idx++) {      /* xevd_DMVR_refine: below, above, right, left, then the diagonal between the better two */
                const int c = dmvr_cost(dx, dy, a0 + ox[idx] + oy[idx] * stride, a1 - ox[idx] - oy[idx] * stride, stride);
                cost[idx] = c;
                if (idx == LEFT) { ox[DIAG] = cost[RIGHT] <= cost[LEFT] ? 1 : -1; oy[DIAG] = cost[BOTTOM] <= cost[TOP] ? 1 : -1; }
                if (c < min_cost) { min_cost = c; d[0] = ox[idx]; d[1] = oy[idx]; }
            }
            if (d[0] == 0 && d[1] == 0) break;
            tot[0] += d[0]; tot[1] += d[1];
        }
        tot[0] <<= 4; tot[1] <<= 4;
        if (not_zero && min_cost == cost[CENTER]) {      /* the centre of the last round won: parametric error surface through its cross */
            const int sb[5] = { cost[CENTER], cost[LEFT], cost[TOP], cost[RIGHT], cost[BOTTOM] };
            int a;
            for (a = 0; a < 2; a++) {
                const int64_t nu = (int64_t)((sb[1 + a] - sb[3 + a]) << 4), de = (int64_t)(sb[1 + a] + sb[3 + a] - (sb[0] << 1));
                if (de != 0) tot[a] += (sb[1 + a] != sb[0] && sb[3 + a] != sb[0]) ? dmvr_div_q7(nu, de) : (sb[1 + a] == sb[0] ? -8 : 8);
            }
        }
        for (l = 0; l < 2; l++) {
            ref16[l][0] = (start[l][0] << 2) + (l ? -tot[0] : tot[0]);
            ref16[l][1] = (start[l][1] << 2) + (l ? -tot[1] : tot[1]);
            refined[num][l][0] = (int16_t)(ref16[l][0] >> 2); refined[num][l][1] = (int16_t)(ref16[l][1] >> 2);
        }
        /* final prediction of the sub-block: the long filters at the refined vector, reading the window of the starting vector */
        for (l = 0; l < 2; l++) {
            const orc_pic *rp = &fr->refp[refi[l]][l];
            const int px = x + sx, py = y + sy;
            const int16_t tq[2] = { (int16_t)(ref16[l][0] >> 2), (int16_t)(ref16[l][1] >> 2) };
            int16_t mvc[2];
            const int clip = dmvr_clip_one(px, py, sp->width, sp->height, dx, dy, tq, mvc);
            const int gx = (px << 4) + (clip ? mvc[0] << 2 : ref16[l][0]), gy = (py << 4) + (clip ? mvc[1] << 2 : ref16[l][1]);
            const int wx = ((((px << 2) + start[l][0]) << 2) >> 4) - 3, wy = ((((py << 2) + start[l][1]) << 2) >> 4) - 3;      /* window origin (luma) */
            const int dlx = (clip ? mvc[0] >> 2 : ref16[l][0] >> 4) - (start[l][0] >> 2), dly = (clip ? mvc[1] >> 2 : ref16[l][1] >> 4) - (start[l][1] >> 2);
            const int dcx = (clip ? mvc[0] >> 3 : ref16[l][0] >> 5) - (start[l][0] >> 3), dcy = (clip ? mvc[1] >> 3 : ref16[l][1] >> 5) - (start[l][1] >> 3);
            int16_t buf[(16 + 7 + 4) * (16 + 7 + 4)], tx[8], ty[8];
            int r, c, comp;
            /* luma: (dx + 7) x (dy + 7) window, 2 samples of replicate padding = clamped indexing */
            for (r = 0; r < dy + 11; r++) for (c = 0; c < dx + 11; c++) {
                const int rr = r - 2 < 0 ? 0 : (r - 2 > dy + 6 ? dy + 6 : r - 2), cc = c - 2 < 0 ? 0 : (c - 2 > dx + 6 ? dx + 6 : c - 2);
                buf[r * (dx + 11) + c] = rp->y[(wy + rr) * rp->s_l + wx + cc];
            }
            luma_taps(gx & 15, sp->tool_admvp, tx); luma_taps(gy & 15, sp->tool_admvp, ty);
            fir_block(buf + (2 + 3 + dly) * (dx + 11) + 2 + 3 + dlx, gx & 15, gy & 15, dx + 11, w, dst[l][0] + sy * w + sx, dx, dy, bd,
                      (gx & 15) != 0, (gy & 15) != 0, 8, 4, tx, ty);
            /* chroma: (dx/2 + 3) x (dy/2 + 3) window at the starting vector's chroma position, 1 sample of padding */
            for (comp = 1; comp < 3; comp++) {
                const int16_t *plane = comp == 1 ? rp->u : rp->v;
                const int cx0 = ((((px << 2) + start[l][0]) << 2) >> 5) - 1, cy0 = ((((py << 2) + start[l][1]) << 2) >> 5) - 1, cw = dx >> 1, ch = dy >> 1;
                int16_t tcx[4], tcy[4];
                for (r = 0; r < ch + 5; r++) for (c = 0; c < cw + 5; c++) {
                    const int rr = r - 1 < 0 ? 0 : (r - 1 > ch + 2 ? ch + 2 : r - 1), cc = c - 1 < 0 ? 0 : (c - 1 > cw + 2 ? cw + 2 : c - 1);
                    buf[r * (cw + 5) + c] = plane[(cy0 + rr) * rp->s_c + cx0 + cc];
                }
                chroma_taps(gx & 31, sp->tool_admvp, tcx); chroma_taps(gy & 31, sp->tool_admvp, tcy);
                fir_block(buf + (1 + 1 + dcy) * (cw + 5) + 1 + 1 + dcx, gx & 31, gy & 31, cw + 5, w >> 1, dst[l][comp] + (sy >> 1) * (w >> 1) + (sx >> 1), cw, ch,
                          sp->bit_depth_chroma, (gx & 31) != 0, (gy & 31) != 0, 4, 5, tcx, tcy);
            }
        }
    }
    free(bl[0]); free(bl[1]);
}

/* xevdm_mc with apply_DMVR (:1860-2038): the conditions that are left when the CU's merge mode allows the refinement; 1 = refined (both predictions
   made and averaged into pred0), 0 = the caller predicts the CU the ordinary way */
int orc_dmvr_cu(const xgpu_seq_params *sp, const orc_frame *fr, int x, int y, int w, int h, const int8_t refi[2], const int16_t mv[2][2],
                int16_t *pred0[3], int16_t *pred1[3], int16_t (*refined)[2][2])
{
    int16_t mv_t[2][2];
    int i;
    if (refi[0] < 0 || refi[1] < 0 || w < 8 || h < 8) return 0;
    {
        const int poc_c = fr->cur.poc, poc0 = fr->refp[refi[0]][0].poc, poc1 = fr->refp[refi[1]][1].poc;
        if (!((poc_c - poc0) * (poc_c - poc1) < 0 && abs(poc_c - poc0) == abs(poc_c - poc1))) return 0;
        mv_clip(x, y, sp->width, sp->height, w, h, refi, mv, mv_t);
        if (poc0 == poc1 && mv_t[0][0] == mv_t[1][0] && mv_t[0][1] == mv_t[1][1]) return 0;
    }
    dmvr_process(sp, fr, x, y, w, h, refi, mv, pred0, pred1, refined);
    for (i = 0; i < w * h; i++) pred0[0][i] = (int16_t)((pred0[0][i] + pred1[0][i] + 1) >> 1);
    for (i = 0; i < (w >> 1) * (h >> 1); i++) {
        pred0[1][i] = (int16_t)((pred0[1][i] + pred1[1][i] + 1) >> 1);
        pred0[2][i] = (int16_t)((pred0[2][i] + pred1[2][i] + 1) >> 1);
    }
    return 1;
}

/* ------------------------------------------------------------------------------------------------
 * Intra prediction, Baseline modes (also what the Main profile runs with sps->tool_eipd = 0, src_main/xevdm.c:1368-1381).
 * ---------------------------------------------------------------------------------------------- */
/* xevd_get_nbr_b (src_base/xevd_ipred.c:33-94): neighbour samples of one component.  `up` and `left` point at index 0 and
   are valid from -1 to cw+ch-1.  A 4x4-luma unit (2x2 chroma) of the row above / column to the left is taken from the picture
   when its SCU is inside the picture and already reconstructed (COD, bit 31 of map_scu) - and intra-coded under
   constrained_intra_pred -, else it is the mid value of the LUMA bit depth (xevd.c:455-473 passes it for all components). */
#define TILE_SAME(m, a, b) (!(m)->map_tidx || (m)->map_tidx[a] == (m)->map_tidx[b])      /* map_tidx[curr] == map_tidx[neighbour] in all of the reference's availability tests */
/* ctx->map_tidx (set_tile_info, src_main/xevdm.c:2300-2330) from the batch's tile grid; NULL for one tile.  The caller frees it. */
static uint8_t *tile_map(const xgpu_tile_grid *g, int w_scu, int h_scu)
{
    uint8_t *t;
    int i, j, y;
    if (!g || g->n_cols * g->n_rows <= 1) return NULL;
    t = (uint8_t *)calloc((size_t)w_scu * h_scu, 1);
    for (j = 0; j < g->n_rows; j++) for (i = 0; i < g->n_cols; i++)
        for (y = g->row_bd[j] * 16; y < g->row_bd[j + 1] * 16 && y < h_scu; y++) {
            const int x0 = g->col_bd[i] * 16, x1 = g->col_bd[i + 1] * 16 < w_scu ? g->col_bd[i + 1] * 16 : w_scu;
            memset(t + (size_t)y * w_scu + x0, j * g->n_cols + i, (size_t)(x1 - x0));
        }
    return t;
}
static void intra_neighbours(const xgpu_seq_params *sp, const orc_maps *m, const int16_t *src, int s, int x_scu, int y_scu,
                             int cw, int ch, int unit, int constrained, int16_t *up, int16_t *left)
{
    const int scuw = cw / unit, scuh = ch / unit, ws = m->w_scu, scup = x_scu + y_scu * ws;
    const int16_t mid = (int16_t)(1 << (sp->bit_depth_luma - 1));
    int i, j;
#define NB_OK(k) (MCU_COD(m->map_scu[k]) && (!constrained || MCU_IF(m->map_scu[k])) && TILE_SAME(m, scup, k))
    up[-1] = (x_scu > 0 && y_scu > 0 && NB_OK(scup - ws - 1)) ? src[-s - 1] : mid;      /* AVAIL_UP_LE, xevd_util.c:722-725 */
    for (i = 0; i < scuw + scuh; i++) {
        const int ok = y_scu > 0 && x_scu + i < ws && NB_OK(scup - ws + i);
        for (j = 0; j < unit; j++) up[i * unit + j] = ok ? src[-s + i * unit + j] : mid;
    }
    for (i = 0; i < scuh + scuw; i++) {
        const int ok = x_scu > 0 && y_scu + i < m->h_scu && NB_OK(scup - 1 + i * ws);
        for (j = 0; j < unit; j++) left[i * unit + j] = ok ? src[(i * unit + j) * s - 1] : mid;
    }
    left[-1] = up[-1];
#undef NB_OK
}

/* xevd_ipred_b / xevd_ipred_uv_b (xevd_ipred.c:625-676): IPD_DC_B 0, HOR 1, VER 2, UL 3, UR 4 for luma and chroma alike */
static void intra_predict(const int16_t *left, const int16_t *up, int16_t *dst, int mode, int w, int h)
{
    int i, j;
    switch (mode) {
    case 2: for (i = 0; i < h; i++) for (j = 0; j < w; j++) dst[i * w + j] = up[j]; break;                 /* :112-124 */
    case 1: for (i = 0; i < h; i++) for (j = 0; j < w; j++) dst[i * w + j] = left[i]; break;               /* :96-108 */
    case 0: {                                                                                              /* ipred_dc_b :149-164 */
        int dc = 0, lw = 0;
        while ((1 << lw) < w) lw++;
        for (i = 0; i < h; i++) dc += left[i];
        for (j = 0; j < w; j++) dc += up[j];
        dc = (dc + w) >> (lw + 1);                   /* the width alone sets rounding and shift, also for non-square blocks */
        for (i = 0; i < w * h; i++) dst[i] = (int16_t)dc;
        break; }
    case 3:                                                                                                /* ipred_ul :587-609 */
        for (i = 0; i < h; i++) for (j = 0; j < w; j++)
            dst[i * w + j] = i > j ? left[i - j - 1] : (i == j ? up[-1] : up[j - i - 1]);
        break;
    case 4:                                                                                                /* ipred_ur :611-622 */
        for (i = 0; i < h; i++) for (j = 0; j < w; j++) dst[i * w + j] = (int16_t)((up[i + j + 1] + left[i + j + 1]) >> 1);
        break;
    default: break;
    }
}

/* ------------------------------------------------------------------------------------------------
 * Intra prediction with sps->tool_eipd (Main): 33 luma modes (DC 0, planar 1, bilinear 2, 30 angular with VER 12 / HOR 24) and
 * 5 chroma modes (DM 0, BI 1, DC 2, HOR 3, VER 4); avail_lr says which of the left / right columns next to the block are reconstructed (the right one only with SUCO).
 * ---------------------------------------------------------------------------------------------- */
/* xevdm_get_nbr (src_main/xevdm_ipred.c:39-148): as intra_neighbours, but an unavailable unit REPEATS the sample before it
   (towards the corner) instead of the mid value, and an unavailable corner takes up[0]. */
static void intra_neighbours_eipd(const xgpu_seq_params *sp, const orc_maps *m, const int16_t *src, int s, int x_scu, int y_scu,
                                  int cw, int ch, int unit, int constrained, int16_t *up, int16_t *left, int16_t *right)
{
    const int scuw = cw / unit, scuh = ch / unit, ws = m->w_scu, scup = x_scu + y_scu * ws;
    int i, j;
#define NB_OK(k) (MCU_COD(m->map_scu[k]) && (!constrained || MCU_IF(m->map_scu[k])) && TILE_SAME(m, scup, k))
    const int ul_ok = x_scu > 0 && y_scu > 0 && NB_OK(scup - ws - 1);
    up[-1] = ul_ok ? src[-s - 1] : (int16_t)(1 << (sp->bit_depth_luma - 1));      /* the mid value only feeds the repetition below */
    for (i = 0; i < scuw + scuh; i++) {
        const int ok = y_scu > 0 && x_scu + i < ws && NB_OK(scup - ws + i);
        for (j = 0; j < unit; j++) up[i * unit + j] = ok ? src[-s + i * unit + j] : up[i * unit - 1];
    }
    if (!ul_ok) up[-1] = up[0];                      /* :80-99: the part left of up[0] repeats up[0] when its unit is unavailable */
    left[-1] = up[-1];
    for (i = 0; i < scuh + scuw; i++) {
        const int ok = x_scu > 0 && y_scu + i < m->h_scu && NB_OK(scup - 1 + i * ws);
        for (j = 0; j < unit; j++) left[i * unit + j] = ok ? src[(i * unit + j) * s - 1] : left[i * unit - 1];
    }
    /* the column right of the block (:123-147): decoded before the block only where SUCO runs a split right to left; starts from the sample above it */
    right[-1] = up[cw];
    for (i = 0; i < scuh + scuw; i++) {
        const int ok = x_scu + scuw < ws && y_scu + i < m->h_scu && NB_OK(scup + scuw + i * ws);
        for (j = 0; j < unit; j++) right[i * unit + j] = ok ? src[(i * unit + j) * s + cw] : right[i * unit - 1];
    }
#undef NB_OK
}
/* xevd_check_nev_avail (src_base/xevd_util.c:1156-1174): bit 0 - the SCU left of the block's first one is reconstructed, bit 1 - the one right of its first row */
static int avail_lr_of(const orc_maps *m, int xs, int ys, int scuw)
{
    const int ws = m->w_scu, scup = xs + ys * ws;
    int lr = 0;
    if (xs > 0 && MCU_COD(m->map_scu[scup - 1]) && TILE_SAME(m, scup, scup - 1)) lr |= 1;
    if (xs + scuw < ws && MCU_COD(m->map_scu[scup + scuw]) && TILE_SAME(m, scup, scup + scuw)) lr |= 2;
    return lr;
}

static int ilog2(int v) { int l = 0; while ((1 << l) < v) l++; return l; }
static const int k_inv_size[8] = { 2048, 1365, 819, 455, 241, 124, 63, 32 };      /* 4096 / (2^k + 1), xevd_ipred.c:122 */

/* slopes of the angular modes in 1/1024: { dx per dy, dy per dx } (xevd_tbl_ipred_dxdy, xevd_tbl.c:294-304) */
static void ang_slopes(int mode, int *dx, int *dy)
{
    static const int t[9] = { 2816, 2048, 1408, 1024, 744, 512, 372, 256, 128 };       /* modes 3..11 dx; mirrored for the rest */
    static const int u[11] = { 128, 256, 372, 512, 744, 1024, 1408, 2048, 2816, 4096, 8192 };
    if (mode >= 3 && mode <= 11)       { *dx = t[mode - 3]; *dy = u[mode - 1]; }
    else if (mode >= 13 && mode <= 23) { *dx = u[mode - 13]; *dy = u[23 - mode]; }
    else if (mode >= 25 && mode <= 32) { *dx = u[35 - mode]; *dy = u[mode - 25]; }
    else                               { *dx = 0; *dy = 0; }
}

/* ipred_ang_val (xevd_ipred.c:377-569): a 4-tap interpolation { 32-o, 64-o, 32+o, o } / 128 between reference positions clamped to [-1, w+h-1] on the row
   above, the left column or - where it is reconstructed (lr & 2) - the right column */
static int ang_sample(const int16_t *left, const int16_t *up, const int16_t *right, int lr, int mode, int i, int j, int w, int h, int maxv)
{
    int dx, dy, o, p, dir;
    const int16_t *ref;
    const int ri = (lr & 2) != 0;
    ang_slopes(mode, &dx, &dy);
#define POS(mt, d, q, off) do { (q) = ((d) * (mt)) >> 10; (off) = (((d) * (mt)) >> 5) - ((q) << 5); } while (0)
    if (mode < 12) {                       /* from the row above, leaning right - past the block's right edge from the right column */
        int t;
        POS(dx, j + 1, t, o);
        if (ri && i >= w - t) { POS(dy, w - i, t, o); p = j - t; ref = right; dir = -1; }
        else { p = i + t; ref = up; dir = 1; }
    } else if (mode > 24) {                /* from the left column, leaning down; with a right column: from above or from the right instead */
        int t;
        if (ri) {
            POS(dy, w - i, t, o);
            if (j < t) { POS(dx, w - i, t, o); p = i + t; ref = up; dir = 1; }
            else { p = j - t; ref = right; dir = -1; }
        } else { POS(dy, i + 1, t, o); p = j + t; ref = left; dir = 1; }
    } else {                               /* between vertical and horizontal: from above or from the left, leaning back (only a right column: from it, leaning down) */
        int ty, t;
        POS(dy, i + 1, ty, o);
        if (j < ty) { POS(dx, j + 1, t, o); p = i - t; ref = up; dir = -1; }
        else if (lr == 2) { POS(dy, w - i, t, o); p = j + t; ref = right; dir = 1; }
        else { p = j - ty; ref = left; dir = -1; }
    }
#undef POS
    {
        const int hi = w + h - 1;
#define CL(v) ((v) < -1 ? -1 : ((v) > hi ? hi : (v)))
        const int v = (int16_t)((ref[CL(p - dir)] * (32 - o) + ref[CL(p)] * (64 - o) + ref[CL(p + dir)] * (32 + o) + ref[CL(p + 2 * dir)] * o + 64) >> 7);
#undef CL
        return v < 0 ? 0 : (v > maxv ? maxv : v);
    }
}

/* xevdm_ipred / xevdm_ipred_uv (src_main/xevdm_ipred.c:241-305).  `mode` is a LUMA mode number; lr = avail_lr (bit 0 left, bit 1 right column reconstructed). */
static void intra_predict_eipd(const int16_t *left, const int16_t *up, const int16_t *right, int lr, int16_t *dst, int mode, int w, int h, int bit_depth)
{
    const int lw = ilog2(w), lh = ilog2(h), maxv = (1 << bit_depth) - 1;
    int i, j;
    if (mode == 12) { for (j = 0; j < h; j++) for (i = 0; i < w; i++) dst[j * w + i] = up[i]; return; }            /* xevd_ipred_vert */
    if (mode == 24) {                                                                                           /* xevdm_ipred_hor :153-196 */
        for (j = 0; j < h; j++) for (i = 0; i < w; i++)
            dst[j * w + i] = lr == 3 ? (int16_t)(((left[j] * (w - i) + right[j] * (i + 1) + (w >> 1)) * k_inv_size[lw]) >> 12) : (lr == 2 ? right[j] : left[j]);
        return;
    }
    if (mode == 0) {                                                                                            /* xevdm_ipred_dc :198-229, xevd_get_dc */
        int dc = 0, hh = h, lhh = lh;
        if (lr != 2) for (j = 0; j < h; j++) dc += left[j];
        if (lr & 2) for (j = 0; j < h; j++) dc += right[j];
        for (i = 0; i < w; i++) dc += up[i];
        if (lr == 3) { hh = h << 1; lhh = lh + 1; }                       /* both columns: the mean over w + 2h samples */
        dc = ((dc + ((w + hh) >> 1)) * k_inv_size[lw > lhh ? lw - lhh : lhh - lw]) >> ((lw < lhh ? lw : lhh) + 12);
        for (i = 0; i < w * h; i++) dst[i] = (int16_t)dc;
        return;
    }
    if (mode == 1) {                                                                                            /* xevd_ipred_plane :163-249 */
        static const int mult[6] = { 13, 17, 5, 11, 23, 47 }, shift[6] = { 7, 10, 11, 15, 19, 23 };
        const int w2 = w >> 1, h2 = h >> 1, iw = lw < 2 ? 0 : lw - 2, ih = lh < 2 ? 0 : lh - 2;
        int ch_ = 0, cv = 0, a, b, c, base;
        if (lr & 2) {                                /* a right column: the mirror image - gradients towards the left, anchored at the bottom-right and top-left samples */
            for (i = 1; i <= w2; i++) ch_ += i * (up[w2 - i] - up[w2 + i]);
            for (j = 1; j <= h2; j++) cv += j * (right[h2 - 1 + j] - right[h2 - 1 - j]);
            a = (right[h - 1] + up[0]) << 4;
        } else {
            for (i = 1; i <= w2; i++) ch_ += i * (up[w2 - 1 + i] - up[w2 - 1 - i]);
            for (j = 1; j <= h2; j++) cv += j * (left[h2 - 1 + j] - left[h2 - 1 - j]);
            a = (left[h - 1] + up[w - 1]) << 4;
        }
        b = ((ch_ << 5) * mult[iw] + (1 << (shift[iw] - 1))) >> shift[iw];
        c = ((cv << 5) * mult[ih] + (1 << (shift[ih] - 1))) >> shift[ih];
        base = a - (h2 - 1) * c - (w2 - 1) * b + 16;
        for (j = 0; j < h; j++) for (i = 0; i < w; i++) {
            const int v = (base + j * c + ((lr & 2) ? w - 1 - i : i) * b) >> 5;
            dst[j * w + i] = (int16_t)(v < 0 ? 0 : (v > maxv ? maxv : v));
        }
        return;
    }
    if (mode == 2) {                                                                                            /* xevd_ipred_bi :251-369 */
        static const int wc_tbl[6] = { -1, 341, 205, 114, 60, 31 };
        if (lr == 3) {                               /* both columns: rows interpolated between them, then towards the bottom row of that from the row above */
            for (j = 0; j < h; j++) for (i = 0; i < w; i++) {
                const int row = ((left[j] * (w - i) + right[j] * (i + 1) + (w >> 1)) * k_inv_size[lw]) >> 12;
                const int bot = ((left[h - 1] * (w - i) + right[h - 1] * (i + 1) + (w >> 1)) * k_inv_size[lw]) >> 12;
                const int t = (up[i] * (h - 1 - j) + bot * (j + 1) + (h >> 1)) >> lh;
                dst[j * w + i] = (int16_t)((row + t + 1) >> 1);
            }
            return;
        }
        {
            /* one column (the right one: the mirror image): corner values a (beyond the row above) and b (below the column) */
            const int16_t *col = lr == 2 ? right : left;
            const int a = lr == 2 ? up[-1] : up[w], b = col[h], ms = lw < lh ? lw : lh;
            const int c = w == h ? (a + b + 1) >> 1 : (((a << lw) + (b << lh)) * wc_tbl[lw > lh ? lw - lh : lh - lw] + (1 << (ms + 9))) >> (ms + 10);
            const int wt = (c << 1) - a - b;
            for (j = 0; j < h; j++) for (i = 0; i < w; i++) {
                const int k = lr == 2 ? w - 1 - i : i;          /* distance from the column */
                const int px = (col[j] << lw) + (k + 1) * (a - col[j]);
                const int py = (up[i] << lh) + (j + 1) * (b - up[i]);
                const int v = ((px << lh) + (py << lw) + k * j * wt + (1 << (lw + lh))) >> (lw + lh + 1);
                dst[j * w + i] = (int16_t)(v < 0 ? 0 : (v > maxv ? maxv : v));
            }
        }
        return;
    }
    for (j = 0; j < h; j++) for (i = 0; i < w; i++) dst[j * w + i] = (int16_t)ang_sample(left, up, right, lr, mode, i, j, w, h, maxv);
}
/* chroma mode -> the luma-numbered predictor to run (xevdm_ipred_uv :267-305; DM follows the luma mode) */
static int eipd_chroma_mode(int ipm_c, int ipm_l)
{
    static const int direct[5] = { -1, 2, 0, 24, 12 };      /* BI_C 1, DC_C 2, HOR_C 3, VER_C 4 */
    return ipm_c == 0 ? ipm_l : direct[ipm_c];
}

/* ------------------------------------------------------------------------------------------------
 * Affine motion compensation (Main, sps->tool_affine): xevdm_affine_mc, src_main/xevdm_mc.c:2606-2685.
 * Control-point vectors are quarter-pel; the model is kept at 2 + 7 fractional bits.
 * ---------------------------------------------------------------------------------------------- */
#define AFF_BIT 7                 /* MAX_CU_LOG2 */
#define AFF_MAX_CU 128
static int aff_round(int v, int shift) { return (v + (1 << (shift - 1)) - (v >= 0)) >> shift; }     /* xevdm_mv_rounding_s32, xevdm_util.c:1857-1868 */
static int aff_clip18(int v) { return CLIP3(-(1 << 17), (1 << 17) - 1, v); }

/* the four model deltas: xevdm_mc.c:2294-2306 (== xevdm_util.c:1891-1903) */
static void aff_deltas(const int16_t mv[3][2], int lw, int lh, int vn, int dh[2], int dv[2])
{
    dh[0] = ((mv[1][0] - mv[0][0]) * (1 << AFF_BIT)) >> lw;
    dh[1] = ((mv[1][1] - mv[0][1]) * (1 << AFF_BIT)) >> lw;
    if (vn == 3) {
        dv[0] = ((mv[2][0] - mv[0][0]) * (1 << AFF_BIT)) >> lh;
        dv[1] = ((mv[2][1] - mv[0][1]) * (1 << AFF_BIT)) >> lh;
    } else { dv[0] = -dh[1]; dv[1] = dh[0]; }
}

/* xevdm_check_eif_applicability_uni, xevdm_util.c:2073-2097 (bounding box of a 4x4 sub-block :2041-2060, fetched lines :2062-2071) */
static int aff_eif_applicable(const int dh[2], const int dv[2], int *mem_band)
{
    const int P = 2 + AFF_BIT, one = 1 << P;
    int cx[4], cy[4], k, mx, nx, my, ny;
    cx[0] = 0; cx[1] = 5 * (dh[0] + one); cx[2] = 5 * dv[0]; cx[3] = cx[1] + cx[2];
    cy[0] = 0; cy[1] = 5 * dh[1]; cy[2] = 5 * (dv[1] + one); cy[3] = cy[1] + cy[2];
    mx = nx = my = ny = 0;
    for (k = 1; k < 4; k++) { if (cx[k] > mx) mx = cx[k]; if (cx[k] < nx) nx = cx[k]; if (cy[k] > my) my = cy[k]; if (cy[k] < ny) ny = cy[k]; }
    *mem_band = (((mx - nx + one - 1) >> P) + 2) * (((my - ny + one - 1) >> P) + 2) <= 72;
    if (dv[1] < -one) return 0;
    if (((dv[1] > 0 ? dv[1] : 0) + abs(dh[1])) * 5 > (1 << P)) return 0;
    return 1;
}

/* xevdm_derive_affine_subblock_size_bi, xevdm_util.c:1870-1945 */
static void aff_subblock(const int16_t mv[2][3][2], const int8_t refi[2], int lw, int lh, int vn, int *sub_w, int *sub_h, int *mem_band)
{
    static const int lut[4] = { 32, 16, 8, 8 };
    const int cuw = 1 << lw, cuh = 1 << lh;
    int l, apply = 1, mb = 1;
    *sub_w = cuw; *sub_h = cuh;
    for (l = 0; l < 2; l++) {
        int dh[2], dv[2], wx, wy, w, h;
        if (refi[l] < 0) continue;
        aff_deltas(mv[l], lw, lh, vn, dh, dv);
        wx = abs(dh[0]) > abs(dh[1]) ? abs(dh[0]) : abs(dh[1]);
        wy = abs(dv[0]) > abs(dv[1]) ? abs(dv[0]) : abs(dv[1]);
        w = wx > 4 ? 4 : (wx == 0 ? cuw : lut[wx - 1]);
        h = wy > 4 ? 4 : (wy == 0 ? cuh : lut[wy - 1]);
        if (w < *sub_w) *sub_w = w;
        if (h < *sub_h) *sub_h = h;
    }
    for (l = 0; l < 2 && apply; l++) {      /* xevdm_check_eif_applicability_bi, :2099-2125: stops at the first list that fails */
        int dh[2], dv[2], m;
        if (refi[l] < 0) continue;
        aff_deltas(mv[l], lw, lh, vn, dh, dv);
        if (!aff_eif_applicable(dh, dv, &m)) apply = 0;
        mb &= m;
    }
    if (!apply) { if (*sub_w < 8) *sub_w = 8; if (*sub_h < 8) *sub_h = 8; }
    if (mem_band) *mem_band = mb;
}

/* eif_derive_mv_clip_range, xevdm_mc.c:2108-2150: 1/32-sample units */
static void aff_eif_range(int x, int y, int lw, int lh, const int dh[2], const int dv[2], const int mv_scale[2], int pic_w, int pic_h,
                          int range_clip, int max_mv[2], int min_mv[2])
{
    static const int spread[5] = { 128, 256, 544, 1120, 2272 };
    const int cuw = 1 << lw, cuh = 1 << lh;
    const int max_pic[2] = { (pic_w + AFF_MAX_CU - x - cuw - 1) * 32, (pic_h + AFF_MAX_CU - y - cuh - 1) * 32 };
    const int min_pic[2] = { (-x - AFF_MAX_CU) * 32, (-y - AFF_MAX_CU) * 32 };
    int c;
    for (c = 0; c < 2; c++) {
        if (!range_clip) { max_mv[c] = max_pic[c]; min_mv[c] = min_pic[c]; }
        else {
            const int centre = aff_round(mv_scale[c] + dh[c] * (cuw >> 1) + dv[c] * (cuh >> 1), 4);
            const int sp = spread[(c == 0 ? lw : lh) - 3];
            min_mv[c] = centre - sp; max_mv[c] = centre + sp;
            if (min_mv[c] < min_pic[c]) { min_mv[c] = min_pic[c]; max_mv[c] = max_pic[c] < min_pic[c] + 2 * sp ? max_pic[c] : min_pic[c] + 2 * sp; }
            else if (max_mv[c] > max_pic[c]) { max_mv[c] = max_pic[c]; min_mv[c] = min_pic[c] > max_pic[c] - 2 * sp ? min_pic[c] : max_pic[c] - 2 * sp; }
        }
        max_mv[c] = aff_clip18(max_mv[c]); min_mv[c] = aff_clip18(min_mv[c]);
    }
}

/* xevdm_eif_mc, xevdm_mc.c:2543-2604: per-sample bilinear fetch at the model's vector (1/32 sample, clamped to the range -
   xevdm_eif_bilinear_clip :2456-2499; the no-clip variant is the same when no vector leaves the range), then the 3-tap
   [-1 10 -1] enhancement filter in both directions (xevdm_eif_filter :2428-2454).  Intermediates are `pel` (s16). */
static void aff_eif(int bw, int bh, int x, int y, const int mv_scale[2], const int dh[2], const int dv[2], const int max_mv_[2], const int min_mv_[2],
                    const int16_t *ref, int s_ref, int16_t *dst, int s_dst, int bd, int chroma)
{
    int mv0[2] = { mv_scale[0], mv_scale[1] }, mx[2] = { max_mv_[0], max_mv_[1] }, mn[2] = { min_mv_[0], min_mv_[1] };
    const int shift1 = bd - 8 < 4 ? bd - 8 : 4, shift2 = 20 - bd > 8 ? 20 - bd : 8, off2 = 1 << (shift2 - 1);
    const int sh2 = bd + 5 - 16 > 0 ? bd + 5 - 16 : 0, sh3 = 6 - sh2;
    const int of2 = sh2 ? 1 << (sh2 - 1) : 0, of3 = 1 << (sh3 - 1);      /* sh2 = 0: the reference's 1 << -1 drops out of the s16 store */
    const int ts = bw + 2;
    int16_t *tmp = (int16_t *)malloc(sizeof(int16_t) * (size_t)ts * (bh + 2));
    int px, py;
    if (chroma) { mv0[0] >>= 1; mv0[1] >>= 1; mx[0] >>= 1; mx[1] >>= 1; mn[0] >>= 1; mn[1] >>= 1; x >>= 1; y >>= 1; }
    ref += y * s_ref + x;
    for (py = -1; py <= bh; py++) for (px = -1; px <= bw; px++) {
        int vx = (mv0[0] + px * dh[0] + py * dv[0]) >> 4, vy = (mv0[1] + px * dh[1] + py * dv[1]) >> 4;
        const int16_t *r;
        int fx, fy, s1, s2;
        vx = vx < mn[0] ? mn[0] : (vx > mx[0] ? mx[0] : vx);
        vy = vy < mn[1] ? mn[1] : (vy > mx[1] ? mx[1] : vy);
        r = ref + (py + (vy >> 5)) * s_ref + px + (vx >> 5);
        fx = vx & 31; fy = vy & 31;
        s1 = (int16_t)(((64 - 2 * fx) * r[0] + 2 * fx * r[1]) >> shift1);
        s2 = (int16_t)(((64 - 2 * fx) * r[s_ref] + 2 * fx * r[s_ref + 1]) >> shift1);
        tmp[(py + 1) * ts + px + 1] = (int16_t)(((64 - 2 * fy) * s1 + 2 * fy * s2 + off2) >> shift2);
    }
    for (py = 0; py < bh + 2; py++) {       /* in place, left to right: every read still sees the unfiltered neighbours */
        int16_t *t = tmp + py * ts;
        for (px = 0; px < bw; px++) t[px] = (int16_t)((-t[px] + t[px + 1] * 10 - t[px + 2] + of2) >> sh2);
    }
    for (py = 0; py < bh; py++) for (px = 0; px < bw; px++) {
        const int16_t *t = tmp + (py + 1) * ts + px;
        const int16_t res = (int16_t)((-t[-ts] + t[0] * 10 - t[ts] + of3) >> sh3);
        dst[py * s_dst + px] = (int16_t)CLIP3(0, (1 << bd) - 1, res);
    }
    free(tmp);
}

/* one list: xevdm_affine_mc_lc, xevdm_mc.c:2259-2391 */
static void aff_mc_list(const xgpu_seq_params *sp, const orc_pic *rp, int x, int y, int lw, int lh, const int16_t mv[3][2], int vn,
                        int sub_w, int sub_h, int mem_band, int16_t *pred[3])
{
    const int cuw = 1 << lw, cuh = 1 << lh, wc = cuw >> 1;
    const int mv_scale[2] = { mv[0][0] * (1 << AFF_BIT), mv[0][1] * (1 << AFF_BIT) };
    int dh[2], dv[2], w, h;
    aff_deltas(mv, lw, lh, vn, dh, dv);
    if (sub_w < 8 || sub_h < 8) {
        int mx[2], mn[2];
        aff_eif_range(x, y, lw, lh, dh, dv, mv_scale, sp->width, sp->height, !mem_band, mx, mn);
        aff_eif(cuw, cuh, x, y, mv_scale, dh, dv, mx, mn, rp->y, rp->s_l, pred[0], cuw, sp->bit_depth_luma, 0);
        aff_eif(cuw >> 1, cuh >> 1, x, y, mv_scale, dh, dv, mx, mn, rp->u, rp->s_c, pred[1], wc, sp->bit_depth_chroma, 1);
        aff_eif(cuw >> 1, cuh >> 1, x, y, mv_scale, dh, dv, mx, mn, rp->v, rp->s_c, pred[2], wc, sp->bit_depth_chroma, 1);
        return;
    }
    {
        /* sub-block translation.  As the reference has it (:2352-2353) the sub-block's position does not enter the vector: every
           sub-block of the CU moves with the vector at the centre of the FIRST one. */
        const int hor_max = (sp->width + AFF_MAX_CU - x - cuw) * 16, ver_max = (sp->height + AFF_MAX_CU - y - cuh) * 16;
        const int hor_min = (-AFF_MAX_CU - x) * 16, ver_min = (-AFF_MAX_CU - y) * 16;
        const int ox = aff_clip18(aff_round(mv_scale[0] + dh[0] * (sub_w >> 1) + dv[0] * (sub_h >> 1), 5));
        const int oy = aff_clip18(aff_round(mv_scale[1] + dh[1] * (sub_w >> 1) + dv[1] * (sub_h >> 1), 5));
        const int cx = ox < hor_min ? hor_min : (ox > hor_max ? hor_max : ox), cy = oy < ver_min ? ver_min : (oy > ver_max ? ver_max : oy);
        for (h = 0; h < cuh; h += sub_h) for (w = 0; w < cuw; w += sub_w) {
            const int gx = (x + w) * 16 + cx, gy = (y + h) * 16 + cy;
            orc_mc_l(rp->y, gx, gy, rp->s_l, cuw, pred[0] + h * cuw + w, sub_w, sub_h, sp->bit_depth_luma, (ox & 15) != 0, (oy & 15) != 0, sp->tool_admvp);
            orc_mc_c(rp->u, gx, gy, rp->s_c, wc, pred[1] + (h >> 1) * wc + (w >> 1), sub_w >> 1, sub_h >> 1, sp->bit_depth_chroma, (ox & 31) != 0, (oy & 31) != 0, sp->tool_admvp);
            orc_mc_c(rp->v, gx, gy, rp->s_c, wc, pred[2] + (h >> 1) * wc + (w >> 1), sub_w >> 1, sub_h >> 1, sp->bit_depth_chroma, (ox & 31) != 0, (oy & 31) != 0, sp->tool_admvp);
        }
    }
}

int orc_affine_mc_cu(const xgpu_seq_params *sp, const orc_frame *fr, int x, int y, int lw, int lh, const int8_t refi[2],
                     const int16_t mv[2][3][2], int vn, int16_t *pred0[3], int16_t *pred1[3])
{
    int16_t **dst[2] = { pred0, pred1 };
    const int w = 1 << lw, h = 1 << lh;
    int sub_w, sub_h, mem_band, l, bidx = 0, i;
    aff_subblock(mv, refi, lw, lh, vn, &sub_w, &sub_h, &mem_band);
    for (l = 0; l < 2; l++) {
        if (refi[l] < 0) continue;
        aff_mc_list(sp, &fr->refp[refi[l]][l], x, y, lw, lh, mv[l], vn, sub_w, sub_h, mem_band, dst[bidx]);
        bidx++;
    }
    if (bidx == 2) {      /* :2649-2683; no identical-motion shortcut here */
        for (i = 0; i < w * h; i++) pred0[0][i] = (int16_t)((pred0[0][i] + pred1[0][i] + 1) >> 1);
        for (i = 0; i < (w >> 1) * (h >> 1); i++) {
            pred0[1][i] = (int16_t)((pred0[1][i] + pred1[1][i] + 1) >> 1);
            pred0[2][i] = (int16_t)((pred0[2][i] + pred1[2][i] + 1) >> 1);
        }
    }
    return bidx;
}

/* xevdm_set_affine_mvf, xevdm_util.c:4095-4190: the vectors the in-loop filters (and later pictures) see - one per sub-block,
   the control points themselves at the CU's corners */
static void affine_set_mvf(const xgpu_cu_batch *b, int i, orc_maps *m)
{
    const int lw = b->log2w[i], lh = b->log2h[i], w_cu = (1 << lw) >> 2, h_cu = (1 << lh) >> 2, vn = b->affine[i];
    const int16_t (*mv)[3][2] = (const int16_t (*)[3][2])&b->affine_mv[i * 12];
    const int scup = (b->y[i] >> 2) * m->w_scu + (b->x[i] >> 2);
    int sub_w, sub_h, l, h, w, yy, xx;
    aff_subblock(mv, &b->refi[i * 2], lw, lh, vn, &sub_w, &sub_h, NULL);
    for (l = 0; l < 2; l++) {
        int dh[2], dv[2];
        if (b->refi[i * 2 + l] < 0) continue;
        aff_deltas(mv[l], lw, lh, vn, dh, dv);
        for (h = 0; h < h_cu; h += sub_h >> 2) for (w = 0; w < w_cu; w += sub_w >> 2) {
            int vx, vy;
            if (w == 0 && h == 0) { vx = mv[l][0][0]; vy = mv[l][0][1]; }
            else if (w + (sub_w >> 2) == w_cu && h == 0) { vx = mv[l][1][0]; vy = mv[l][1][1]; }
            else if (w == 0 && h + (sub_h >> 2) == h_cu && vn == 3) { vx = mv[l][2][0]; vy = mv[l][2][1]; }
            else {
                const int px = (w << 2) + (sub_w >> 1), py = (h << 2) + (sub_h >> 1);
                vx = aff_clip18(aff_round(mv[l][0][0] * (1 << AFF_BIT) + dh[0] * px + dv[0] * py, 5)) >> 2;
                vy = aff_clip18(aff_round(mv[l][0][1] * (1 << AFF_BIT) + dh[1] * px + dv[1] * py, 5)) >> 2;
            }
            for (yy = h; yy < h + (sub_h >> 2); yy++) for (xx = w; xx < w + (sub_w >> 2); xx++) {
                m->map_mv[(scup + yy * m->w_scu + xx) * 4 + l * 2 + 0] = (int16_t)vx;
                m->map_mv[(scup + yy * m->w_scu + xx) * 4 + l * 2 + 1] = (int16_t)vy;
            }
        }
    }
}


/* ------------------------------------------------------------------------------------------------
 * HTDF, the Hadamard transform domain filter (Main, sps->tool_htdf): xevdm_htdf, src_main/xevdm_recon.c:153-385, called per CU right after
 * its reconstruction (xevdm.c:1381-1392).  Luma only.
 * ---------------------------------------------------------------------------------------------- */
static const uint8_t k_htdf_thr_log2[5] = { 6, 7, 7, 8, 8 };
static const uint8_t k_htdf_tbl[5][16] = {
    { 0, 0, 2,  6, 10, 14, 19, 23, 28, 32,  36,  41,  45,  49,  53,  57 },
    { 0, 0, 5, 12, 20, 29, 38, 47, 56, 65,  73,  82,  90,  98, 107, 115 },
    { 0, 0, 1,  4,  9, 16, 24, 32, 41, 50,  59,  68,  77,  86,  94, 103 },
    { 0, 0, 3,  9, 19, 32, 47, 64, 81, 99, 117, 135, 154, 179, 205, 230 },
    { 0, 0, 0,  2,  6, 11, 18, 27, 38, 51,  64,  96, 128, 160, 192, 224 },
};
/* read_table (:176-189): small coefficients go through the table, large ones pass */
static int htdf_lut(int z, const uint8_t *tbl, int thr, int shift, int rnd)
{
    const int a = z < 0 ? -z : z;
    const int v = a < thr ? tbl[((a + rnd) & thr) >> shift] : a;      /* (a + rnd) & thr: the reference's index mask, kept as it is */
    return z < 0 ? -v : v;
}
/* xevd_get_avail_intra, src_base/xevd_util.c:689-745: bit 0 up, 1 left, 3 right, 5 up-left, 6 up-right, 7 low-left, 8 low-right */
static int avail_intra(const orc_maps *m, int xs, int ys, int scuw, int scuh)
{
    const int k = ys * m->w_scu + xs;
    int av = 0;
#define AV_OK(n) (MCU_COD(m->map_scu[n]) && TILE_SAME(m, k, n))
    if (xs > 0 && AV_OK(k - 1)) {
        av |= 1 << 1;
        if (ys + scuh + scuw - 1 < m->h_scu && AV_OK(k + m->w_scu * (scuw + scuh) - m->w_scu - 1)) av |= 1 << 7;
    }
    if (ys > 0) {
        if (TILE_SAME(m, k, k - m->w_scu)) av |= 1 << 0;
        if (xs > 0 && AV_OK(k - m->w_scu - 1)) av |= 1 << 5;
        if (xs + scuw < m->w_scu && AV_OK(k - m->w_scu + scuw)) av |= 1 << 6;
    }
    if (xs + scuw < m->w_scu && AV_OK(k + scuw)) {
        av |= 1 << 3;
        if (ys + scuh + scuw - 1 < m->h_scu && AV_OK(k + m->w_scu * (scuw + scuh - 1) + scuw)) av |= 1 << 8;
    }
#undef AV_OK
    return av;
}
static void orc_htdf(int16_t *rec, int s, int w, int h, int qp, int intra, const orc_maps *m, int xs, int ys, int constrained, int bd)
{
    const int we = w + 2, he = h + 2, k = ys * m->w_scu + xs;
    const int mn = w < h ? w : h, mxs = w > h ? w : h;
    int16_t *tb, *acc;
    int av, i, r, c, idx, thr_log2, shift, rnd, thr;
    /* xevdm_htdf_skip_condition (:270-297) */
    if (qp <= 17 || w * h < 64 || mxs >= 128) return;
    if (!intra) { if (mn >= 32) return; }
    else if (w == h && mn >= 32) qp -= 8;
    av = avail_intra(m, xs, ys, w >> 2, h >> 2);
    tb = (int16_t *)malloc(sizeof(int16_t) * we * he);
    acc = (int16_t *)calloc((size_t)we * he, sizeof(int16_t));
    for (i = 0; i < h; i++) memcpy(tb + (i + 1) * we + 1, rec + i * s, sizeof(int16_t) * w);
    /* one sample of border: the neighbour's reconstruction where it exists (and is intra under constrained intra prediction), else the CU's own edge */
    for (i = 0; i < h; i++) {
        tb[(i + 1) * we] = ((av >> 1) & 1) && (!constrained || MCU_IF(m->map_scu[k - 1 + (i >> 2) * m->w_scu])) ? rec[i * s - 1] : rec[i * s];
        tb[(i + 1) * we + we - 1] = ((av >> 3) & 1) && (!constrained || MCU_IF(m->map_scu[k + (w >> 2) + (i >> 2) * m->w_scu])) ? rec[i * s + w] : rec[i * s + w - 1];
    }
    for (i = 0; i < w; i++) {
        tb[i + 1] = (av & 1) && (!constrained || MCU_IF(m->map_scu[k - m->w_scu + (i >> 2)])) ? rec[i - s] : rec[i];
        tb[(he - 1) * we + i + 1] = rec[(h - 1) * s + i];
    }
    tb[0] = ((av >> 5) & 1) ? rec[-1 - s] : rec[0];
    tb[we - 1] = ((av >> 6) & 1) ? rec[w - s] : rec[w - 1];
    tb[we * (he - 1)] = ((av >> 7) & 1) ? rec[-1 + h * s] : rec[(h - 1) * s];
    tb[we - 1 + we * (he - 1)] = ((av >> 8) & 1) ? rec[w + h * s] : rec[w - 1 + (h - 1) * s];
    /* filter_block_luma (:252-268) + xevdm_htdf_filter_block (:201-250): every 2x2 window, Hadamard, table on the three AC terms, back,
       accumulated into the four samples; a sample is final once its fourth window has gone by */
    idx = (qp - 20 + 4) >> 3; idx = idx < 0 ? 0 : (idx > 4 ? 4 : idx);
    thr_log2 = k_htdf_thr_log2[idx]; shift = thr_log2 - 4; rnd = (1 << shift) >> 1; thr = (1 << thr_log2) - (1 << shift);
    for (r = 0; r < he - 1; r++) for (c = 0; c < we - 1; c++) {
        int16_t *in = tb + r * we + c, *out = acc + r * we + c;
        const int x0 = in[0], x1 = in[1], x2 = in[we], x3 = in[we + 1];
        const int y0 = x0 + x2, y1 = x1 + x3, y2 = x0 - x2, y3 = x1 - x3;
        const int z0 = y0 + y1, z1 = htdf_lut(y0 - y1, k_htdf_tbl[idx], thr, shift, rnd), z2 = htdf_lut(y2 + y3, k_htdf_tbl[idx], thr, shift, rnd),
                  z3 = htdf_lut(y2 - y3, k_htdf_tbl[idx], thr, shift, rnd);
        const int i0 = z0 + z2, i1 = z1 + z3, i2 = z0 - z2, i3 = z1 - z3;
        out[0] = (int16_t)(out[0] + ((i0 + i1) >> 2));
        out[1] = (int16_t)(out[1] + ((i0 - i1) >> 2));
        out[we] = (int16_t)(out[we] + ((i2 + i3) >> 2));
        out[we + 1] = (int16_t)(out[we + 1] + ((i2 - i3) >> 2));
        in[0] = (int16_t)CLIP3(0, (1 << bd) - 1, (out[0] + 2) >> 2);
    }
    for (i = 0; i < h; i++) memcpy(rec + i * s, tb + (i + 1) * we + 1, sizeof(int16_t) * w);
    free(tb); free(acc);
}


int orc_recon_batch(const xgpu_seq_params *sp, const orc_frame *fr, const xgpu_cu_batch *b, orc_maps *maps, int16_t *resid_out)
{
    return orc_recon_batch_ex(sp, fr, b, maps, resid_out, NULL);
}

/* dmvr_mv_out (or NULL): for every CU of the batch that carries the DMVR flag, has two references and is at least 8x8 - in batch order, its 16x16
   sub-blocks in raster order - the vectors [list][x/y] in quarter samples the decoder keeps for temporal prediction: refined where the
   refinement ran, the CU's own where its conditions failed (xgpu_batch_dmvr_mvs returns the same array) */
int orc_recon_batch_ex(const xgpu_seq_params *sp, const orc_frame *fr, const xgpu_cu_batch *b, orc_maps *maps, int16_t *resid_out, int16_t *dmvr_mv_out)
{
    int16_t *pred[2][3], *res;
    int i, c, l;
    size_t dmvr_n = 0;
    int16_t refined[64][2][2];
    int dmvr_done = 0;
    uint8_t *tmap = maps ? tile_map(b->tiles, maps->w_scu, maps->h_scu) : NULL;
    if (maps) maps->map_tidx = tmap;
    for (l = 0; l < 2; l++) for (c = 0; c < 3; c++) pred[l][c] = (int16_t *)malloc(sizeof(int16_t) * MAX_CU * MAX_CU);
    res = (int16_t *)malloc(sizeof(int16_t) * MAX_CU * MAX_CU);

    for (i = 0; i < b->n_cu; i++) {
        const int x = b->x[i], y = b->y[i], lw = b->log2w[i], lh = b->log2h[i], w = 1 << lw, h = 1 << lh;
        size_t off = b->coef_off[i];
        const int inter = b->pred_mode[i] != XGPU_MODE_INTRA;
        /* local dual tree: a luma-only (intra / IBC) or chroma-only (intra) CU - prediction, residual, HTDF and the map update per plane it has (xevd_check_luma /
           xevd_check_chroma in xevd_recon_unit, src_main/xevdm.c:1230-1405; a chroma-only CU leaves the maps alone, xevdm_set_dec_info xevdm_util.c:4241) */
        const int planes = CU_PLANES(b, i);
#define HAS_PLANE(c) ((planes >> ((c) ? 1 : 0)) & 1)
        if (b->pred_mode[i] == XGPU_MODE_IBC) {
            /* xevdm_IBC_mc, xevdm_mc.c:2040-2106: a copy out of the CURRENT picture (reconstructed, not yet filtered) at the whole-sample block
               vector mv[0]; chroma at the halved vector */
            const int bx = b->mv[i * 4], by = b->mv[i * 4 + 1];
            int r;
            for (r = 0; r < h; r++) memcpy(pred[0][0] + r * w, fr->cur.y + (y + by + r) * fr->cur.s_l + x + bx, sizeof(int16_t) * w);
            for (r = 0; r < h >> 1; r++) {
                memcpy(pred[0][1] + r * (w >> 1), fr->cur.u + ((y >> 1) + (by >> 1) + r) * fr->cur.s_c + (x >> 1) + (bx >> 1), sizeof(int16_t) * (w >> 1));
                memcpy(pred[0][2] + r * (w >> 1), fr->cur.v + ((y >> 1) + (by >> 1) + r) * fr->cur.s_c + (x >> 1) + (bx >> 1), sizeof(int16_t) * (w >> 1));
            }
        } else if (inter && b->affine && b->affine[i])
            orc_affine_mc_cu(sp, fr, x, y, lw, lh, &b->refi[i * 2], (const int16_t (*)[3][2])&b->affine_mv[i * 12], b->affine[i], pred[0], pred[1]);
        else if (inter) {
            int done = 0;
            dmvr_done = 0;
            if (b->dmvr && b->dmvr[i] && b->refi[i * 2] >= 0 && b->refi[i * 2 + 1] >= 0 && w >= 8 && h >= 8) {
                const int nsub = (w > 16 ? w / 16 : 1) * (h > 16 ? h / 16 : 1);
                int k;
                done = orc_dmvr_cu(sp, fr, x, y, w, h, &b->refi[i * 2], (const int16_t (*)[2])&b->mv[i * 4], pred[0], pred[1], refined);
                for (k = 0; k < nsub && dmvr_mv_out; k++)
                    memcpy(dmvr_mv_out + (dmvr_n + (size_t)k) * 4, done ? &refined[k][0][0] : &b->mv[i * 4], 4 * sizeof(int16_t));
                dmvr_n += (size_t)nsub;
                dmvr_done = done;
            }
            if (!done) orc_mc_cu(sp, fr, x, y, w, h, &b->refi[i * 2], (const int16_t (*)[2])&b->mv[i * 4], pred[0], pred[1]);
        }
        else if (maps) {      /* xevd_recon_unit's intra branch, xevd.c:731-741 (availability needs the SCU map) */
            int16_t nb_up[2 * MAX_CU + 8], nb_le[2 * MAX_CU + 8], nb_ri[2 * MAX_CU + 8];
            const int lr = avail_lr_of(maps, x >> 2, y >> 2, w >> 2);
            for (c = 0; c < 3; c++) {
                const int cw = c ? w >> 1 : w, ch = c ? h >> 1 : h, s = c ? fr->cur.s_c : fr->cur.s_l;
                const int16_t *plane = c == 0 ? fr->cur.y : (c == 1 ? fr->cur.u : fr->cur.v);
                if (!HAS_PLANE(c)) continue;
                if (sp->tool_eipd) {
                    const int ml = b->ipm ? b->ipm[i * 2] : 0, mc = b->ipm ? b->ipm[i * 2 + 1] : 0;
                    intra_neighbours_eipd(sp, maps, plane + (c ? (y >> 1) * s + (x >> 1) : y * s + x), s, x >> 2, y >> 2, cw, ch, c ? 2 : 4,
                                          b->constrained_intra_pred, nb_up + 4, nb_le + 4, nb_ri + 4);
                    intra_predict_eipd(nb_le + 4, nb_up + 4, nb_ri + 4, lr, pred[0][c], c ? eipd_chroma_mode(mc, ml) : ml, cw, ch,
                                       c ? sp->bit_depth_chroma : sp->bit_depth_luma);
                    continue;
                }
                intra_neighbours(sp, maps, plane + (c ? (y >> 1) * s + (x >> 1) : y * s + x), s, x >> 2, y >> 2, cw, ch, c ? 2 : 4,
                                 b->constrained_intra_pred, nb_up + 4, nb_le + 4);
                intra_predict(nb_le + 4, nb_up + 4, pred[0][c], b->ipm ? b->ipm[i * 2 + (c ? 1 : 0)] : 0, cw, ch);
            }
        }
        const int ai = (inter && b->pred_mode[i] != XGPU_MODE_IBC && b->ats_inter) ? b->ats_inter[i] : 0;      /* no ATS for IBC, xevdm.c:602 */
        for (c = 0; c < 3 && ai; c++) {
            /* ATS-inter (xevdm_sub_block_itdq :808-816, xevdm_recon :62-112): one TU of half/quarter size per component, luma
               with DST-VII/DCT-VIII when the CU is at most 32x32 (xevdm_get_ats_inter_trs, xevdm_util.c:3636-3668), residual
               added only inside the TU */
            const int cw = c ? w >> 1 : w, ch = c ? h >> 1 : h;
            const int coded = (b->cbf[i] >> c) & 1;
            int16_t *plane = c == 0 ? fr->cur.y : (c == 1 ? fr->cur.u : fr->cur.v);
            const int s = c ? fr->cur.s_c : fr->cur.s_l;
            int16_t *rec = plane + (c ? (y >> 1) * s + (x >> 1) : y * s + x);
            int tx, ty, tw, th, r, q;
            ats_inter_tu(ai, cw, ch, &tx, &ty, &tw, &th);
            if (coded) {
                int l2w = 0, l2h = 0;
                while ((1 << l2w) < tw) l2w++;
                while ((1 << l2h) < th) l2h++;
                memcpy(res, b->coef + off, sizeof(int16_t) * tw * th);
                if (c == 0 && lw <= 5 && lh <= 5) {
                    const int idx = ai & 15, pos = (ai >> 4) & 15, hor = idx == 2 || idx == 4;
                    const int t_h = hor ? 0 : (pos == 0), t_v = hor ? (pos == 0) : 0;      /* 1 = DCT-VIII, 0 = DST-VII */
                    orc_itdq_ats(res, l2w, l2h, b->qp[i * 3], sp->bit_depth_luma, sp->tool_iqt, t_v, t_h);
                } else
                    orc_itdq(res, l2w, l2h, b->qp[i * 3 + c], sp->bit_depth_luma, sp->tool_iqt);
                if (resid_out) memcpy(resid_out + off, res, sizeof(int16_t) * tw * th);
                off += (size_t)tw * th;
            }
            for (r = 0; r < ch; r++) for (q = 0; q < cw; q++) {
                const int in = coded && q >= tx && q < tx + tw && r >= ty && r < ty + th;
                const int16_t t = (int16_t)((in ? res[(r - ty) * tw + (q - tx)] : 0) + pred[0][c][r * cw + q]);
                rec[r * s + q] = (int16_t)CLIP3(0, (1 << sp->bit_depth_luma) - 1, t);
            }
        }
        for (c = 0; c < 3 && !ai; c++) {
            const int cw = c ? w >> 1 : w, ch = c ? h >> 1 : h, clw = c ? lw - 1 : lw, clh = c ? lh - 1 : lh;
            const int coded = HAS_PLANE(c) && ((b->cbf[i] >> c) & 1);
            int16_t *plane = c == 0 ? fr->cur.y : (c == 1 ? fr->cur.u : fr->cur.v);
            const int s = c ? fr->cur.s_c : fr->cur.s_l;
            if (coded) {
                /* xevd_sub_block_itdq (xevd_itdq.c:544-621): the LUMA bit depth is used for all components
                   (xevd.c:441-442); TBs are at most 64x64: a larger CU is transformed as 64x64 (chroma 32x32)
                   sub-blocks sb = (j<<1)|i that are copied out of and back into the CU-strided block */
                const int tlw = clw > (c ? 5 : 6) ? (c ? 5 : 6) : clw, tlh = clh > (c ? 5 : 6) ? (c ? 5 : 6) : clh;
                const int nsx = 1 << (clw - tlw), nsy = 1 << (clh - tlh), tw = 1 << tlw, th = 1 << tlh;
                int si, sj, r;
                memcpy(res, b->coef + off, sizeof(int16_t) * cw * ch);
                for (sj = 0; sj < nsy; sj++) for (si = 0; si < nsx; si++) {
                    const int sb = (sj << 1) | si;
                    int16_t *blk = res + sj * th * cw + si * tw;
                    if (nsx * nsy > 1 && b->cbf_sub && !((b->cbf_sub[i] >> (4 * c + sb)) & 1)) continue;
                    if (nsx * nsy == 1) {
                        const int a = b->ats ? b->ats[i] : 0;
                        if (c == 0 && (a & 1) && !inter)      /* ats_intra_cu: luma TB of an intra CU, xevdm_itdq.c:820-829 */
                            orc_itdq_ats(res, tlw, tlh, b->qp[i * 3 + c], sp->bit_depth_luma, sp->tool_iqt, (a >> 1) & 1, (a >> 2) & 1);
                        else
                            orc_itdq(res, tlw, tlh, b->qp[i * 3 + c], sp->bit_depth_luma, sp->tool_iqt);
                        continue;
                    }
                    {
                        int16_t *tmp = (int16_t *)malloc(sizeof(int16_t) * tw * th);
                        for (r = 0; r < th; r++) memcpy(tmp + r * tw, blk + r * cw, sizeof(int16_t) * tw);
                        orc_itdq(tmp, tlw, tlh, b->qp[i * 3 + c], sp->bit_depth_luma, sp->tool_iqt);
                        for (r = 0; r < th; r++) memcpy(blk + r * cw, tmp + r * tw, sizeof(int16_t) * tw);
                        free(tmp);
                    }
                }
                if (resid_out) memcpy(resid_out + off, res, sizeof(int16_t) * cw * ch);
                off += (size_t)cw * ch;
            }
            if ((inter || maps) && HAS_PLANE(c))      /* xevd_recon_yuv passes the luma bit depth for chroma too, xevd_recon.c:75-90 */
                orc_recon(res, pred[0][c], coded, cw, ch, s, plane + (c ? (y >> 1) * s + (x >> 1) : y * s + x), sp->bit_depth_luma);
        }
        if (maps && b->htdf_slice_qp && b->pred_mode[i] != XGPU_MODE_IBC && ((b->cbf[i] & 1) || !inter) && (planes & 1))      /* xevdm.c:1381-1392 */
            orc_htdf(fr->cur.y + y * fr->cur.s_l + x, fr->cur.s_l, w, h, b->htdf_slice_qp, !inter, maps, x >> 2, y >> 2, !inter && b->constrained_intra_pred,
                     sp->bit_depth_luma);
        if (maps && (planes & 1)) set_dec_info(sp, b, i, maps);
#undef HAS_PLANE
        if (maps && inter && dmvr_done && !sp->tool_addb) {
            /* The SCU map after a refined CU.  xevdm_set_dec_info keeps both sets of vectors (map_mv refined, map_unrefined_mv not, xevdm_util.c:
               4327-4338); the ADDB filter is handed the unrefined ones (xevdm.c:2009-2041) - but the Main library's copy of the BASELINE filter reads
               ctx->map_mv itself (xevdm_df.c:118,209): with sps->tool_addb off the deblocking filter sees the REFINED vectors.  This map exists for
               the filters only, so it holds what the filter of the sequence reads. */
            const int dx = w < 16 ? w : 16, dy = h < 16 ? h : 16;
            int sx, sy, u, v, k = 0;
            for (sy = 0; sy < h; sy += dy) for (sx = 0; sx < w; sx += dx, k++)
                for (v = 0; v < dy >> 2; v++) for (u = 0; u < dx >> 2; u++)
                    memcpy(&maps->map_mv[((size_t)(((y + sy) >> 2) + v) * maps->w_scu + ((x + sx) >> 2) + u) * 4], &refined[k][0][0], 4 * sizeof(int16_t));
        }
        if (maps && inter && b->affine && b->affine[i]) affine_set_mvf(b, i, maps);
    }
    for (l = 0; l < 2; l++) for (c = 0; c < 3; c++) free(pred[l][c]);
    free(res);
    if (maps) maps->map_tidx = NULL;
    free(tmap);
    return 0;
}

/* xevd_tbl_df_st, src_base/xevd_tbl.c:306-324 (filter strength by edge class and QP) */
static const uint8_t k_df_st[4][52] = {
    { 0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,1,1,1,1,1,1,1,1,1,2,2,2,2,2,3,3,3,4,4,4,5,5,6,6,7,8,9,10,11,12,12,12,12,12 },
    { 0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,1,1,1,1,1,2,2,2,3,3,3,4,4,5,5,6,7,8, 9,10,11,11,11,11,11 },
    { 0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,1,1,1,2,2,2,3,3,4,4,5,6,7, 8, 9,10,10,10,10,10 },
    { 0 },
};
/* xevd_tbl_qp_chroma_adjust_base, src_base/xevd_tbl.c:345-354 (Baseline default chroma QP mapping) */
/* xevd_tbl_qp_chroma_adjust_main, xevd_tbl.c:334-342: the default mapping when sps->tool_iqt is on (src_main/xevdm.c:471-479) */
static const int8_t k_chroma_qp_main[58] = { 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16, 17, 18, 19, 20, 21, 22, 23, 24, 25, 26, 27, 28, 29,
    29, 30, 31, 32, 33, 34, 35, 36, 37, 37, 38, 39, 40, 40, 41, 42, 43, 44, 45, 46, 47, 48, 49, 50, 51, 52, 53, 54 };
static const int8_t k_chroma_qp_base[58] = {
     0,  1,  2,  3,  4,  5,  6,  7,  8,  9, 10, 11, 12, 13, 14, 15, 16, 17, 18, 19,
    20, 21, 22, 23, 24, 25, 26, 27, 28, 29, 29, 29, 30, 31, 32, 32, 33, 33, 34, 34,
    35, 35, 36, 36, 36, 37, 37, 37, 38, 38, 39, 39, 40, 40, 40, 41, 41, 41 };
const int8_t *orc_default_chroma_qp_table(void) { return k_chroma_qp_base; }

/* get_tbl_qp_to_st, src_base/xevd_df.c:34-94: edge class 0 intra, 1 luma cbf, 2 motion differs, 3 none */
static int edge_class(const orc_maps *m, int k0, int k1)
{
    const uint32_t m0 = m->map_scu[k0], m1 = m->map_scu[k1];
    const int8_t *r0 = &m->map_refi[k0 * 2], *r1 = &m->map_refi[k1 * 2];
    int mv0[2][2], mv1[2][2], l, d;
    if (MCU_IF(m0) || MCU_IF(m1)) return 0;
    if (MCU_CBFL(m0) || MCU_CBFL(m1)) return 1;
    if (MCU_IBC(m0) || MCU_IBC(m1)) return 2;      /* the Main library's copy, xevdm_df.c:52-55 */
    for (l = 0; l < 2; l++) for (d = 0; d < 2; d++) {
        mv0[l][d] = r0[l] >= 0 ? m->map_mv[k0 * 4 + l * 2 + d] : 0;
        mv1[l][d] = r1[l] >= 0 ? m->map_mv[k1 * 4 + l * 2 + d] : 0;
    }
    if (r0[0] == r1[0] && r0[1] == r1[1])
        return (abs(mv0[0][0] - mv1[0][0]) >= 4 || abs(mv0[0][1] - mv1[0][1]) >= 4 ||
                abs(mv0[1][0] - mv1[1][0]) >= 4 || abs(mv0[1][1] - mv1[1][1]) >= 4) ? 2 : 3;
    if (r0[0] == r1[1] && r0[1] == r1[0])
        return (abs(mv0[0][0] - mv1[1][0]) >= 4 || abs(mv0[0][1] - mv1[1][1]) >= 4 ||
                abs(mv0[1][0] - mv1[0][0]) >= 4 || abs(mv0[1][1] - mv1[0][1]) >= 4) ? 2 : 3;
    return 2;
}

/* xevd_qp_chroma_dynamic[c][clip(-6*(bdc-8), 57, qp)] (xevd_df.c:362-365).  A caller-supplied table starts at
   qp = -6*(bdc-8); the default is the Baseline static table with the identity extension below 0 that
   xevd_set_chroma_qp_tbl_loc builds (xevd_tbl.c:364-372).  The strength-table index is clamped to 0..51: the
   reference reads out of bounds there (xevd_tbl_df_st rows are 52 long), conformant streams never do. */
static int chroma_qp(const xgpu_seq_params *sp, int c, int qp)
{
    const int boff = 6 * (sp->bit_depth_chroma - 8);
    int v;
    qp = CLIP3(-boff, 57, qp);
    if (sp->chroma_qp_table[c]) v = sp->chroma_qp_table[c][qp + boff];
    else v = qp < 0 ? qp : (sp->tool_iqt ? k_chroma_qp_main[qp] : k_chroma_qp_base[qp]);
    return CLIP3(0, 51, v);
}

/* one 4-sample luma edge segment + its chroma, between SCU kq (right/below, supplies the QP) and SCU kp.
   xevd_df.c:343-371 (hor) / :442-476 (ver) */
/* which planes the CU whose edge is being filtered has (local dual tree, xevd_check_luma_fn / xevd_check_chroma_fn in xevdm_df.c:155-160, 916-920): bit 0 luma, bit 1 chroma */
static int g_dbk_planes = 3;
static void dbk_segment(const xgpu_seq_params *sp, const orc_frame *fr, const orc_maps *m, int kq, int kp,
                        int x_pel, int y_pel, int is_ver)
{
    const int cls = edge_class(m, kq, kp);
    const int qp = MCU_QP(m->map_scu[kq]);
    const int bdl = sp->bit_depth_luma, bdc = sp->bit_depth_chroma;
    const int st = k_df_st[cls][qp] << (bdl - 8);
    int st_u, st_v;
    if (st && (g_dbk_planes & 1)) orc_dbk_luma(fr->cur.y + y_pel * fr->cur.s_l + x_pel, st, fr->cur.s_l, bdl, is_ver);
    st_u = k_df_st[cls][chroma_qp(sp, 0, qp + fr->qp_u_offset)] << (bdc - 8);
    st_v = k_df_st[cls][chroma_qp(sp, 1, qp + fr->qp_v_offset)] << (bdc - 8);
    if ((st_u || st_v) && (g_dbk_planes & 2)) {
        const int off = (y_pel >> 1) * fr->cur.s_c + (x_pel >> 1);
        orc_dbk_chroma(fr->cur.u + off, fr->cur.v + off, st_u, st_v, fr->cur.s_c, bdc, is_ver);
    }
}

int orc_deblock_baseline(const xgpu_seq_params *sp, const orc_frame *fr, const xgpu_cu_batch *b, orc_maps *m)
{
    const int ws = m->w_scu;
    int i, r, c, k;
    /* edges on a tile border are left alone unless pps.loop_filter_across_tiles_enabled_flag (no_boundary, src_main/xevdm_df.c:142, 233, 274) */
    uint8_t *tmap = (b->tiles && !b->tiles->loop_filter_across_tiles) ? tile_map(b->tiles, m->w_scu, m->h_scu) : NULL;
#define TB_OK(p, q) (!tmap || tmap[p] == tmap[q])
    /* pass 1: vertical edges (xevd.c:1190-1210 with is_hor_edge=0 -> xevd_deblock_cu_ver, xevd_df.c:385-546).
       COD is cleared, CUs are visited in decode order; the left edge is filtered when the left neighbour is
       already visited, the right edge when the right neighbour is (never the case in quad-tree z-order). */
    for (k = 0; k < ws * m->h_scu; k++) m->map_scu[k] &= 0x7FFFFFFFu;
    for (i = 0; i < b->n_cu; i++) {
        /* a CU wider than 64 is filtered as two halves, each like a CU of its own (deblock_tree, src_main/xevdm.c:2017-2037) */
        const int cw = 1 << b->log2w[i], y = b->y[i], w = cw > 64 ? 64 : cw, h = 1 << b->log2h[i];
        int x;
        g_dbk_planes = CU_PLANES(b, i);
        for (x = b->x[i]; x < b->x[i] + cw; x += 64) {
            const int t = (x >> 2) + (y >> 2) * ws;
            if (x > 0 && MCU_COD(m->map_scu[t - 1]) && TB_OK(t, t - 1))
                for (r = 0; r < h >> 2; r++) dbk_segment(sp, fr, m, t + r * ws, t + r * ws - 1, x, y + 4 * r, 1);
            if (x + w < sp->width && MCU_COD(m->map_scu[t + (w >> 2)]) && TB_OK(t, t + (w >> 2)))
                for (r = 0; r < h >> 2; r++) dbk_segment(sp, fr, m, t + r * ws + (w >> 2), t + r * ws + (w >> 2) - 1, x + w, y + 4 * r, 1);
            for (r = 0; r < h >> 2; r++) for (c = 0; c < w >> 2; c++) m->map_scu[t + r * ws + c] |= 1u << 31;
        }
    }
    /* pass 2: horizontal edges (xevd_deblock_cu_hor, xevd_df.c:291-383): top edge of every CU (or 64-row half) below row 0 */
    for (i = 0; i < b->n_cu; i++) {
        const int x = b->x[i], w = 1 << b->log2w[i], ch = 1 << b->log2h[i];
        int y;
        g_dbk_planes = CU_PLANES(b, i);
        for (y = b->y[i]; y < b->y[i] + ch; y += 64) {
            const int t = (x >> 2) + (y >> 2) * ws;
            if (y > 0 && TB_OK(t, t - ws))
                for (c = 0; c < w >> 2; c++) dbk_segment(sp, fr, m, t + c, t + c - ws, x + 4 * c, y, 0);
        }
    }
    free(tmap);
    return 0;
}

/* ------------------------------------------------------------------------------------------------
 * ADDB deblocking (Main profile, sps->tool_addb).  src_main/xevdm_df.c:361-1135.
 * Tables ALPHA_TABLE / BETA_TABLE / CLIP_TAB: src_main/xevdm_tbl.c:377-379 (constants of the EVC specification).
 * ---------------------------------------------------------------------------------------------- */
static const uint8_t k_addb_alpha[52] = { 0,0,0,0,0,0,0,0,0,0,0,0, 0,0,0,0,4,4,5,6, 7,8,9,10,12,13,15,17, 20,22,25,28,32,36,40,45,
    50,56,63,71,80,90,101,113, 127,144,162,182,203,226,255,255 };
static const uint8_t k_addb_beta[52] = { 0,0,0,0,0,0,0,0,0,0,0,0, 0,0,0,0,2,2,2,3, 3,3,3,4,4,4,6,6, 7,7,8,8,9,9,10,10,
    11,11,12,12,13,13,14,14, 15,15,16,16,17,17,18,18 };
static const uint8_t k_addb_clip[52][5] = {
    {0,0,0,0,0},{0,0,0,0,0},{0,0,0,0,0},{0,0,0,0,0},{0,0,0,0,0},{0,0,0,0,0},{0,0,0,0,0},{0,0,0,0,0},
    {0,0,0,0,0},{0,0,0,0,0},{0,0,0,0,0},{0,0,0,0,0},{0,0,0,0,0},{0,0,0,0,0},{0,0,0,0,0},{0,0,0,0,0},
    {0,0,0,0,0},{0,0,0,1,1},{0,0,0,1,1},{0,0,0,1,1},{0,0,0,1,1},{0,0,1,1,1},{0,0,1,1,1},{0,1,1,1,1},
    {0,1,1,1,1},{0,1,1,1,1},{0,1,1,1,1},{0,1,1,2,2},{0,1,1,2,2},{0,1,1,2,2},{0,1,1,2,2},{0,1,2,3,3},
    {0,1,2,3,3},{0,2,2,3,3},{0,2,2,4,4},{0,2,3,4,4},{0,2,3,4,4},{0,3,3,5,5},{0,3,4,6,6},{0,3,4,6,6},
    {0,4,5,7,7},{0,4,5,8,8},{0,4,6,9,9},{0,5,7,10,10},{0,6,8,11,11},{0,6,8,13,13},{0,7,10,14,14},{0,8,11,16,16},
    {0,9,12,18,18},{0,10,13,20,20},{0,11,15,23,23},{0,13,17,25,25} };

/* get_bs, xevdm_df.c:361-513.  k0 = current (right/below) SCU, k1 = neighbour; (x0,y0),(x1,y1) their sample positions.
   Reference pictures are compared by identity (XEVD_PIC pointers) - here by their luma plane pointer. */
static int addb_bs(const xgpu_seq_params *sp, const orc_frame *fr, const orc_maps *m, int k0, int x0, int y0, int k1, int x1, int y1)
{
    const uint32_t m0 = m->map_scu[k0], m1 = m->map_scu[k1];
    const int intra = MCU_IF(m0) || MCU_IF(m1);
    const int lg = sp->log2_ctu;
    const int8_t *r0 = &m->map_refi[k0 * 2], *r1 = &m->map_refi[k1 * 2];
    const int16_t *p0[2], *p1[2];
    int mv0[2][2], mv1[2][2], l, d;
    if (intra && ((x0 >> lg) != (x1 >> lg) || (y0 >> lg) != (y1 >> lg))) return 4;
    if (intra) return 3;
    if (MCU_IBC(m0) || MCU_IBC(m1)) return 3;      /* xevdm_df.c:411-414 */
    if (MCU_CBFL(m0) || MCU_CBFL(m1) || (m->map_ats && (m->map_ats[k0] || m->map_ats[k1]))) return 2;     /* ats_present, xevdm_df.c:415 */
    for (l = 0; l < 2; l++) {
        p0[l] = r0[l] >= 0 ? fr->refp[r0[l]][l].y : NULL;
        p1[l] = r1[l] >= 0 ? fr->refp[r1[l]][l].y : NULL;
        for (d = 0; d < 2; d++) {
            mv0[l][d] = r0[l] >= 0 ? m->map_mv[k0 * 4 + l * 2 + d] : 0;
            mv1[l][d] = r1[l] >= 0 ? m->map_mv[k1 * 4 + l * 2 + d] : 0;
        }
    }
#define MVSAME(a, b) (abs((a)[0] - (b)[0]) < 4 && abs((a)[1] - (b)[1]) < 4)
    if ((p0[0] == p1[0] && p0[1] == p1[1]) || (p0[0] == p1[1] && p0[1] == p1[0])) {
        if (p0[0] == p0[1])
            return (MVSAME(mv0[0], mv1[0]) && MVSAME(mv0[1], mv1[1]) && MVSAME(mv0[0], mv1[1]) && MVSAME(mv0[1], mv1[0])) ? 0 : 1;
        if (p0[0] == p1[0] && p0[1] == p1[1])
            return (MVSAME(mv0[0], mv1[0]) && MVSAME(mv0[1], mv1[1])) ? 0 : 1;
        return (MVSAME(mv0[0], mv1[1]) && MVSAME(mv0[1], mv1[0])) ? 0 : 1;
    }
    return 1;
#undef MVSAME
}

/* deblock_scu_line_luma, xevdm_df.c:584-709.  p[i] = buf[-(i+1)*step], q[i] = buf[i*step] */
static void addb_line_luma(int16_t *buf, int step, int bs, int alpha, int beta, int c1, int bd)
{
    const int maxv = (1 << bd) - 1;
    int16_t p[4], q[4], po[4], qo[4];
    int i, ap, aq;
    for (i = 0; i < 4; i++) { q[i] = buf[i * step]; p[i] = buf[-(i + 1) * step]; po[i] = p[i]; qo[i] = q[i]; }
    if (!(bs && abs(p[0] - q[0]) < alpha && abs(p[1] - p[0]) < beta && abs(q[1] - q[0]) < beta)) return;
    ap = abs(p[0] - p[2]) < beta;
    aq = abs(q[0] - q[2]) < beta;
    if (bs == 4) {
        const int strong = abs(p[0] - q[0]) < ((alpha >> 2) + 2);
        if (ap && strong) {
            po[0] = (int16_t)((p[2] + 2 * (p[1] + p[0] + q[0]) + q[1] + 4) >> 3);
            po[1] = (int16_t)((p[2] + p[1] + p[0] + q[0] + 2) >> 2);
            po[2] = (int16_t)((2 * p[3] + 3 * p[2] + p[1] + p[0] + q[0] + 4) >> 3);
        } else po[0] = (int16_t)((2 * p[1] + p[0] + q[1] + 2) >> 2);
        if (aq && strong) {
            qo[0] = (int16_t)((q[2] + 2 * (q[1] + q[0] + p[0]) + p[1] + 4) >> 3);
            qo[1] = (int16_t)((q[2] + q[1] + q[0] + p[0] + 2) >> 2);
            qo[2] = (int16_t)((2 * q[3] + 3 * q[2] + q[1] + q[0] + p[0] + 4) >> 3);
        } else qo[0] = (int16_t)((2 * q[1] + q[0] + p[1] + 2) >> 2);
    } else {
        const int c0 = (uint8_t)(c1 + ((ap + aq) << (bd - 9 > 0 ? bd - 9 : 0)));
        const int d0 = CLIP3(-c0, c0, (4 * (q[0] - p[0]) + p[1] - q[1] + 4) >> 3);
        po[0] = (int16_t)CLIP3(0, maxv, p[0] + d0);
        qo[0] = (int16_t)CLIP3(0, maxv, q[0] - d0);
        if (ap) po[1] = (int16_t)(p[1] + CLIP3(-c1, c1, (((p[2] + p[0] + q[0]) * 3) - 8 * p[1] - q[1]) >> 4));
        if (aq) qo[1] = (int16_t)(q[1] + CLIP3(-c1, c1, (((q[2] + q[0] + p[0]) * 3) - 8 * q[1] - p[1]) >> 4));
    }
    for (i = 0; i < 4; i++) { buf[i * step] = (int16_t)CLIP3(0, maxv, qo[i]); buf[-(i + 1) * step] = (int16_t)CLIP3(0, maxv, po[i]); }
}
/* deblock_scu_line_chroma, xevdm_df.c:710-781 */
static void addb_line_chroma(int16_t *buf, int step, int bs, int alpha, int beta, int c0, int bd)
{
    const int maxv = (1 << bd) - 1;
    const int16_t p0 = buf[-step], p1 = buf[-2 * step], q0 = buf[0], q1 = buf[step];
    int po = p0, qo = q0;
    if (!(bs && abs(p0 - q0) < alpha && abs(p1 - p0) < beta && abs(q1 - q0) < beta)) return;
    if (bs == 4) {
        po = (2 * p1 + p0 + q1 + 2) >> 2;
        qo = (2 * q1 + q0 + p1 + 2) >> 2;
    } else {
        const int d0 = CLIP3(-c0, c0, (4 * (q0 - p0) + p1 - q1 + 4) >> 3);
        po = CLIP3(0, maxv, p0 + d0);
        qo = CLIP3(0, maxv, q0 - d0);
    }
    buf[-step] = (int16_t)CLIP3(0, maxv, po); buf[0] = (int16_t)CLIP3(0, maxv, qo);
    buf[-2 * step] = (int16_t)CLIP3(0, maxv, p1); buf[step] = (int16_t)CLIP3(0, maxv, q1);
}

/* one 4-sample segment of an 8x8-grid edge: deblock_addb_cu_hor (xevdm_df.c:893-944) / deblock_addb_cu_ver_yuv (:947-1034).
   get_index() takes its arguments as u8 (xevdm_df.c:356-359): offsets and negative chroma QPs wrap like there. */
static int addb_index(int qp, int offset) { return CLIP3(0, 51, (int)(uint8_t)qp + (int)(uint8_t)offset); }
static void addb_segment(const xgpu_seq_params *sp, const orc_frame *fr, const orc_maps *m, int kq, int kp,
                         int x_pel, int y_pel, int is_ver, int alpha_off, int beta_off)
{
    const int bdl = sp->bit_depth_luma, bdc = sp->bit_depth_chroma, scale = bdl - 8;
    const int bs = is_ver ? addb_bs(sp, fr, m, kq, x_pel, y_pel, kp, x_pel - 1, y_pel) : addb_bs(sp, fr, m, kq, x_pel, y_pel, kp, x_pel, y_pel - 1);
    const int qp = (MCU_QP(m->map_scu[kq]) + MCU_QP(m->map_scu[kp]) + 1) >> 1;
    int ia = addb_index(qp, alpha_off), ib = addb_index(qp, beta_off);
    int alpha = k_addb_alpha[ia] << scale, beta = (uint8_t)(k_addb_beta[ib] << scale);
    int c1 = (uint8_t)(k_addb_clip[ia][bs] << (bdl - 9 > 0 ? bdl - 9 : 0));
    int i, c;
    int16_t *y = fr->cur.y + y_pel * fr->cur.s_l + x_pel;
    for (i = 0; i < 4 && (g_dbk_planes & 1); i++)
        addb_line_luma(is_ver ? y + i * fr->cur.s_l : y + i, is_ver ? 1 : fr->cur.s_l, bs, alpha, beta, c1, bdl);
    for (c = 0; c < 2 && (g_dbk_planes & 2); c++) {
        int16_t *pl = (c ? fr->cur.v : fr->cur.u) + (y_pel >> 1) * fr->cur.s_c + (x_pel >> 1);
        const int boff = 6 * (bdc - 8);
        int q = CLIP3(-boff, 57, qp + (c ? fr->qp_v_offset : fr->qp_u_offset));
        int qc = sp->chroma_qp_table[c] ? sp->chroma_qp_table[c][q + boff] : (q < 0 ? q : (sp->tool_iqt ? k_chroma_qp_main[q] : k_chroma_qp_base[q]));
        int c0;
        ia = addb_index(qc, alpha_off); ib = addb_index(qc, beta_off);
        alpha = k_addb_alpha[ia] << scale;                    /* luma bit depth scales chroma too, xevdm_df.c:926-927 */
        beta = (uint8_t)(k_addb_beta[ib] << scale);
        c0 = (uint8_t)((k_addb_clip[ia][bs] + 1) << (bdc - 9 > 0 ? bdc - 9 : 0));
        for (i = 0; i < 2; i++)
            addb_line_chroma(is_ver ? pl + i * fr->cur.s_c : pl + i, is_ver ? 1 : fr->cur.s_c, bs, alpha, beta, c0, bdc);
    }
}

int orc_deblock_addb(const xgpu_seq_params *sp, const orc_frame *fr, const xgpu_cu_batch *b, orc_maps *m, int alpha_off, int beta_off)
{
    const int ws = m->w_scu;
    int i, r, c, k;
    uint8_t *tmap = (b->tiles && !b->tiles->loop_filter_across_tiles) ? tile_map(b->tiles, m->w_scu, m->h_scu) : NULL;      /* xevdm_df.c:877, 1088, 1106 */
    /* vertical edges on the 8x8 luma grid (deblock_addb_cu_ver, xevdm_df.c:1036-1135), then horizontal (:835-945) */
    for (k = 0; k < ws * m->h_scu; k++) m->map_scu[k] &= 0x7FFFFFFFu;
    /* a CU wider (taller) than 64 is passed to the CU filter as two 64-sample halves (deblock_tree, xevdm.c:1989-2037),
       which makes its inner 64-sample boundary an edge */
    for (i = 0; i < b->n_cu; i++) {
        const int cx = b->x[i], y = b->y[i], cw = 1 << b->log2w[i], h = 1 << b->log2h[i];
        int hx;
        g_dbk_planes = CU_PLANES(b, i);
        for (hx = 0; hx < cw; hx += 64) {
            const int x = cx + hx, w = cw > 64 ? 64 : cw;
            const int t = (x >> 2) + (y >> 2) * ws;
            if ((x & 7) == 0 && x > 0 && MCU_COD(m->map_scu[t - 1]) && TB_OK(t, t - 1))
                for (r = 0; r < h >> 2; r++) addb_segment(sp, fr, m, t + r * ws, t + r * ws - 1, x, y + 4 * r, 1, alpha_off, beta_off);
            if (((x + w) & 7) == 0 && x + w < sp->width && MCU_COD(m->map_scu[t + (w >> 2)]) && TB_OK(t, t + (w >> 2)))
                for (r = 0; r < h >> 2; r++) addb_segment(sp, fr, m, t + r * ws + (w >> 2), t + r * ws + (w >> 2) - 1, x + w, y + 4 * r, 1, alpha_off, beta_off);
            for (r = 0; r < h >> 2; r++) for (c = 0; c < w >> 2; c++) m->map_scu[t + r * ws + c] |= 1u << 31;
        }
    }
    for (i = 0; i < b->n_cu; i++) {
        const int x = b->x[i], cy = b->y[i], w = 1 << b->log2w[i], ch = 1 << b->log2h[i];
        int hy;
        g_dbk_planes = CU_PLANES(b, i);
        for (hy = 0; hy < ch; hy += 64) {
            const int y = cy + hy;
            const int t = (x >> 2) + (y >> 2) * ws;
            if ((y & 7) == 0 && y > 0 && TB_OK(t, t - ws))
                for (c = 0; c < w >> 2; c++) addb_segment(sp, fr, m, t + c, t + c - ws, x + 4 * c, y, 0, alpha_off, beta_off);
        }
    }
    g_dbk_planes = 3;
    free(tmap);
    return 0;
#undef TB_OK
}

/* ------------------------------------------------------------------------------------------------
 * Adaptive loop filter.  src_main/xevdm_alf.c.
 * The reference filters CTU by CTU from a copy of the deblocked picture that is replicate-extended by 3 samples
 * (alf_copy_and_extend_tile :805-842).  Every CTU gets a private (W+6)x(H+6) window (alf_process_tile :1000-1052):
 *   - its own rows take their left/right halo from the copy when that side is "available", else by mirroring
 *     around the CTU edge without repeating the edge sample;
 *   - the 3 rows above/below are copied WHOLE (halo columns included) from the copy when available - so next to a
 *     picture's left/right border those corner samples are the copy's replicate extension, not mirrored ones -
 *     else they mirror the window's own rows (which already carry their halos).
 * Availability = not on the tile (picture) border; with pps.loop_filter_across_tiles_enabled_flag the test is
 * made against pic_width-1 / pic_height-1 (:990-999), which makes the right and bottom picture borders count as
 * available (they then read the replicate extension).
 * Several tiles: every tile has its OWN replicate-extended copy (alf_process_tile copies w_tile x h_tile samples to a
 * private place of the temporary picture, :934-967), so "available" at a tile border inside the picture means the
 * replicated edge of the CTU's tile, never the neighbouring tile's samples - the clamp below is to the tile [tx0,tx1) x [ty0,ty1).
 * ---------------------------------------------------------------------------------------------- */
static int16_t *alf_ctu_window(const int16_t *a, int s, int tx0, int tx1, int ty0, int ty1, int x0, int y0, int cw, int ch,
                               int aL, int aR, int aT, int aB, int *ws_out)
{
    const int m = 3, ws = cw + 2 * m;
    int16_t *buf = (int16_t *)malloc(sizeof(int16_t) * (size_t)ws * (ch + 2 * m));
    int16_t *o = buf + m * ws + m;
    int r, c;
#define PIC(yy, xx) a[CLIP3(ty0, ty1 - 1, (yy)) * s + CLIP3(tx0, tx1 - 1, (xx))]        /* the tile's replicate-extended copy */
    for (r = 0; r < ch; r++) for (c = -m; c < cw + m; c++) {
        int xx = x0 + c;
        if (c < 0 && !aL) xx = x0 - c;
        if (c >= cw && !aR) xx = x0 + cw - 1 - (c - cw + 1);
        o[r * ws + c] = PIC(y0 + r, xx);
    }
    for (r = 1; r <= m; r++) for (c = -m; c < cw + m; c++) {
        o[-r * ws + c] = aT ? PIC(y0 - r, x0 + c) : o[r * ws + c];
        o[(ch - 1 + r) * ws + c] = aB ? PIC(y0 + ch - 1 + r, x0 + c) : o[(ch - 1 - r) * ws + c];
    }
#undef PIC
    *ws_out = ws;
    return o;
}

/* alf_derive_classification_blk, xevdm_alf.c:38-208: class (0..24) and transpose index (0..3) of the 4x4 block at (bx,by) */
static void alf_classify(const int16_t *o, int es, int bx, int by, int bd, int *cls, int *tr)
{
    static const int th[16] = { 0, 1, 2, 2, 2, 2, 2, 3, 3, 3, 3, 3, 3, 3, 3, 4 };
    static const int trans_tbl[8] = { 0, 1, 0, 2, 2, 3, 1, 3 };
    int sv = 0, sh = 0, sd0 = 0, sd1 = 0, r, c;
    int hv1, hv0, d1, d0, dir_hv, dir_d, hvd1, hvd0, main_dir, sec_dir, act, ci, strength = 0;
    for (r = by - 2; r < by + 6; r++) for (c = bx - 2; c < bx + 6; c++) {
        const int16_t *p = o + r * es + c;
        const int16_t p2 = (int16_t)(p[0] << 1);
        sv  += abs(p2 - p[-es] - p[es]);
        sh  += abs(p2 - p[1] - p[-1]);
        sd0 += abs(p2 - p[-es - 1] - p[es + 1]);
        sd1 += abs(p2 - p[es - 1] - p[-es + 1]);
    }
    act = (int16_t)CLIP3(0, 15, (sv + sh) >> (bd - 2));
    ci = th[act];
    if (sv > sh) { hv1 = sv; hv0 = sh; dir_hv = 1; } else { hv1 = sh; hv0 = sv; dir_hv = 3; }
    if (sd0 > sd1) { d1 = sd0; d0 = sd1; dir_d = 0; } else { d1 = sd1; d0 = sd0; dir_d = 2; }
    if (d1 * hv0 > hv1 * d0) { hvd1 = d1; hvd0 = d0; main_dir = dir_d; sec_dir = dir_hv; }
    else { hvd1 = hv1; hvd0 = hv0; main_dir = dir_hv; sec_dir = dir_d; }
    if (hvd1 > 2 * hvd0) strength = 1;
    if (hvd1 * 2 > 9 * hvd0) strength = 2;
    if (strength) ci += (((main_dir & 1) << 1) + strength) * 5;
    *cls = ci; *tr = trans_tbl[main_dir * 2 + (sec_dir >> 1)];
}

int orc_alf(const xgpu_seq_params *sp, const orc_pic *pic, const xgpu_alf_params *ap)
{
    /* coefficient order per transpose index, xevdm_alf.c:268-273 */
    static const int l[4][13] = {
        { 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12 }, { 9, 4, 10, 8, 1, 5, 11, 7, 3, 0, 2, 6, 12 },
        { 0, 3, 2, 1, 8, 7, 6, 5, 4, 9, 10, 11, 12 }, { 9, 8, 10, 4, 3, 7, 11, 5, 1, 0, 2, 6, 12 } };
    const int w = sp->width, h = sp->height, ctu = 1 << sp->log2_ctu, w_ctu = (w + ctu - 1) / ctu;
    const int maxv = (1 << sp->bit_depth_luma) - 1;          /* one bit depth for all clip ranges, xevd_alf_init :431-437 */
    int16_t *copy[3];
    int c, x0, y0, i;
    if (!ap->enable[0] && !ap->enable[1] && !ap->enable[2]) return 0;
    for (c = 0; c < 3; c++) {                                 /* the pre-filter copy every CTU reads from */
        const int pw = c ? w >> 1 : w, ph = c ? h >> 1 : h, s = c ? pic->s_c : pic->s_l;
        const int16_t *pl = c == 0 ? pic->y : (c == 1 ? pic->u : pic->v);
        int y;
        copy[c] = (int16_t *)malloc(sizeof(int16_t) * (size_t)pw * ph);
        for (y = 0; y < ph; y++) memcpy(copy[c] + (size_t)y * pw, pl + (size_t)y * s, sizeof(int16_t) * pw);
    }
    for (y0 = 0; y0 < h; y0 += ctu) for (x0 = 0; x0 < w; x0 += ctu) {
        const int cw = x0 + ctu > w ? w - x0 : ctu, ch = y0 + ctu > h ? h - y0 : ctu;
        /* the CTU's tile in luma samples (one tile: the picture) */
        int tx0 = 0, tx1 = w, ty0 = 0, ty1 = h, aL, aR, aT, aB;
        if (ap->tiles) {
            const xgpu_tile_grid *g = ap->tiles;
            int t;
            for (t = 0; t < g->n_cols; t++) if ((x0 >> sp->log2_ctu) >= g->col_bd[t] && (x0 >> sp->log2_ctu) < g->col_bd[t + 1]) { tx0 = g->col_bd[t] << sp->log2_ctu; tx1 = g->col_bd[t + 1] << sp->log2_ctu; }
            for (t = 0; t < g->n_rows; t++) if ((y0 >> sp->log2_ctu) >= g->row_bd[t] && (y0 >> sp->log2_ctu) < g->row_bd[t + 1]) { ty0 = g->row_bd[t] << sp->log2_ctu; ty1 = g->row_bd[t + 1] << sp->log2_ctu; }
            if (tx1 > w) tx1 = w;
            if (ty1 > h) ty1 = h;
        }
        /* tile_boundary_check against the tile, or - across tiles - against (0, pic_w - 1, 0, pic_h - 1) (:990-999) */
        aL = ap->across_tiles ? x0 != 0 : x0 != tx0; aT = ap->across_tiles ? y0 != 0 : y0 != ty0;
        aR = ap->across_tiles ? 1 : (x0 + cw != tx1); aB = ap->across_tiles ? 1 : (y0 + ch != ty1);
        const int ctu_idx = (y0 >> sp->log2_ctu) * w_ctu + (x0 >> sp->log2_ctu);
        int ws, x, y;
        if (ap->enable[0] && (!ap->ctb_flag || ap->ctb_flag[ctu_idx])) {
            int16_t *o = alf_ctu_window(copy[0], w, tx0, tx1, ty0, ty1, x0, y0, cw, ch, aL, aR, aT, aB, &ws);
            for (y = 0; y < ch; y += 4) for (x = 0; x < cw; x += 4) {
                int cls, tr, ii, jj;
                int16_t f[13];
                alf_classify(o, ws, x, y, sp->bit_depth_luma, &cls, &tr);
                for (i = 0; i < 13; i++) f[i] = ap->luma_coef[cls * 13 + l[tr][i]];
                for (ii = 0; ii < 4; ii++) for (jj = 0; jj < 4; jj++) {             /* alf_filter_blk_7, :210-337 */
                    const int16_t *p = o + (y + ii) * ws + x + jj;
                    int sum = f[0] * (p[3 * ws] + p[-3 * ws])
                            + f[1] * (p[2 * ws + 1] + p[-2 * ws - 1]) + f[2] * (p[2 * ws] + p[-2 * ws]) + f[3] * (p[2 * ws - 1] + p[-2 * ws + 1])
                            + f[4] * (p[ws + 2] + p[-ws - 2]) + f[5] * (p[ws + 1] + p[-ws - 1]) + f[6] * (p[ws] + p[-ws])
                            + f[7] * (p[ws - 1] + p[-ws + 1]) + f[8] * (p[ws - 2] + p[-ws + 2])
                            + f[9] * (p[3] + p[-3]) + f[10] * (p[2] + p[-2]) + f[11] * (p[1] + p[-1]) + f[12] * p[0];
                    sum = (sum + 256) >> 9;
                    pic->y[(y0 + y + ii) * pic->s_l + x0 + x + jj] = (int16_t)CLIP3(0, maxv, sum);
                }
            }
            free(o - 3 * ws - 3);
        }
        for (c = 1; c < 3; c++) {
            int16_t *pl = c == 1 ? pic->u : pic->v;
            const int16_t *f = ap->chroma_coef;
            int16_t *o;
            if (!ap->enable[c]) continue;
            o = alf_ctu_window(copy[c], w >> 1, tx0 >> 1, tx1 >> 1, ty0 >> 1, ty1 >> 1, x0 >> 1, y0 >> 1, cw >> 1, ch >> 1, aL, aR, aT, aB, &ws);
            for (y = 0; y < ch >> 1; y++) for (x = 0; x < cw >> 1; x++) {          /* alf_filter_blk_5, :339-429 */
                const int16_t *p = o + y * ws + x;
                int sum = f[0] * (p[2 * ws] + p[-2 * ws]) + f[1] * (p[ws + 1] + p[-ws - 1]) + f[2] * (p[ws] + p[-ws]) + f[3] * (p[ws - 1] + p[-ws + 1])
                        + f[4] * (p[2] + p[-2]) + f[5] * (p[1] + p[-1]) + f[6] * p[0];
                sum = (sum + 256) >> 9;
                pl[((y0 >> 1) + y) * pic->s_c + (x0 >> 1) + x] = (int16_t)CLIP3(0, maxv, sum);
            }
            free(o - 3 * ws - 3);
        }
    }
    for (c = 0; c < 3; c++) free(copy[c]);
    return 0;
}

void orc_pad(const xgpu_seq_params *sp, const orc_pic *p)
{
    int c, i, j;
    for (c = 0; c < 3; c++) {
        int16_t *a = c == 0 ? p->y : (c == 1 ? p->u : p->v);
        const int s = c ? p->s_c : p->s_l, w = c ? sp->width >> 1 : sp->width, h = c ? sp->height >> 1 : sp->height;
        const int e = c ? XGPU_PAD_C : XGPU_PAD_L;
        for (i = 0; i < h; i++) for (j = 0; j < e; j++) { a[i * s - e + j] = a[i * s]; a[i * s + w + j] = a[i * s + w - 1]; }
        for (i = 0; i < e; i++) {
            memcpy(a - e - (i + 1) * s, a - e, sizeof(int16_t) * s);
            memcpy(a - e + (h + i) * s, a - e + (h - 1) * s, sizeof(int16_t) * s);
        }
    }
}

/* app/xevd_app_util.h:665-708: dst 8 bit -> imgb_conv_shift_right_8b (:464-494); same depth -> plane copy; lower ->
   imgb_conv_shift_right (:519-552); higher -> imgb_conv_shift_left (:495-517) */
void orc_output_convert(const int16_t *src, int stride, int w, int h, int src_bd, int dst_bd, void *dst)
{
    int i, j;
    uint8_t *d8 = (uint8_t *)dst;
    uint16_t *d16 = (uint16_t *)dst;
    for (i = 0; i < h; i++) {
        const int16_t *s = src + (size_t)i * stride;
        for (j = 0; j < w; j++) {
            if (dst_bd == 8) {
                const int shift = src_bd - 8, add = shift ? 1 << (shift - 1) : 0;
                int t = (s[j] + add) >> shift;
                d8[(size_t)i * w + j] = (uint8_t)(t < 0 ? 0 : (t > 255 ? 255 : t));
            } else if (dst_bd == src_bd) {
                d16[(size_t)i * w + j] = (uint16_t)s[j];
            } else if (dst_bd < src_bd) {
                const int shift = src_bd - dst_bd, add = 1 << (shift - 1), maxv = (1 << dst_bd) - 1;
                int t = ((uint16_t)s[j] + add) >> shift;
                d16[(size_t)i * w + j] = (uint16_t)(t < 0 ? 0 : (t > maxv ? maxv : t));
            } else {
                d16[(size_t)i * w + j] = (uint16_t)((uint16_t)s[j] << (dst_bd - src_bd));
            }
        }
    }
}

/* src_main/xevdm_dra.c:301-355 (chroma: |v - 512| * scale(luma) + 256 >> 9, sign restored, + 512; the luma index is clamped at 0
   only) then :272-299 (luma: table look-up).  Results are stored as shorts without clipping, as the reference does. */
void orc_dra_apply(int16_t *y, int16_t *u, int16_t *v, int w, int h, const int32_t *luma_inv, const int32_t *cb_inv, const int32_t *cr_inv)
{
    int c, j, k;
    for (c = 1; c < 3; c++) {
        int16_t *pl = c == 1 ? u : v;
        const int32_t *lut = c == 1 ? cb_inv : cr_inv;
        for (j = 0; j < h / 2; j++)
            for (k = 0; k < w / 2; k++) {
                int ref = y[(size_t)(2 * j) * w + 2 * k];
                int sv = pl[(size_t)j * (w / 2) + k] - 512, off = sv < 0 ? -sv : sv;
                if (ref < 0) ref = 0;
                off = (off * lut[ref] + (1 << 8)) >> 9;
                if (sv < 0) off = -off;
                pl[(size_t)j * (w / 2) + k] = (int16_t)(512 + off);
            }
    }
    for (j = 0; j < h; j++)
        for (k = 0; k < w; k++) y[(size_t)j * w + k] = (int16_t)luma_inv[y[(size_t)j * w + k]];
}
