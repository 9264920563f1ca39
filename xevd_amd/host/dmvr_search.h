// dmvr_search.h - the SEARCH half of decoder-side motion vector refinement on the host (plain C++, no device code).
//
// Why the front end needs it: with sps->tool_dmvr the reference decoder refines the two vectors of a merge-mode bi-predicted CU while it reconstructs the
// picture CU by CU, and the REFINED vectors become decoder state inside the same picture - xevdm_set_dec_info copies them into ctx->map_mv
// (src_main/xevdm_util.c:4327-4338) and ends with core->mv = map_mv[first SCU] (:4384-4387), which is what the history buffer receives (tool_hmvp,
// src_main/xevdm.c:1335-1342) and what the merge list of a later MMVD CU is built from (tool_mmvd, xevdm_util.c:246-247: map_mv, no unrefined map).
// The syntax of the next CU therefore depends on reference SAMPLES.  The backend (k_dmvr.hip) refines whole pictures at once, after parsing; for
// streams that switch DMVR on together with HMVP or MMVD - what Main-profile encoders do - the parser runs the refinement search itself, on host
// copies of the two reference pictures' luma planes (xhost_parser_set_ref_luma), and the backend repeats it for the prediction (both are bit-exact
// restatements of processDMVR, so they agree).
//
// What is restated here: the conditions of xevdm_mc (src_main/xevdm_mc.c:1895-1911), mv_clip (:939-980), the bilinear pre-interpolation of both
// lists two samples wider than the CU (xevdm_bl_mc_l, :358-486), and per 16x16 sub-block (processDMVR :1647-1829) up to two rounds of the 5-point SAD
// search with the mirrored offset in list 1 (xevd_DMVR_refine :1293-1339, xevd_DMVR_cost :1270-1291) and the parametric sub-sample step
// (xevd_SubPelErrorSrfc :1373-1427, div_for_maxq7 :1341-1372).  Not here: the final 8-tap prediction - that is the backend's.
#pragma once
#include <stdint.h>
#include <stdlib.h>
#include <vector>

struct DmvrRefPlane { const int16_t *y; int stride; int poc; };      // y = sample (0, 0) of a plane with at least 144 samples of replicated border

// xevdm_mc's apply_DMVR for a CU whose merge mode allows the refinement: two references at equal POC distances on either side of the picture, at least 8x8
static inline bool dmvr_search_applies(int cur_poc, int poc0, int poc1, int w, int h)
{
    return (cur_poc - poc0) * (cur_poc - poc1) < 0 && abs(cur_poc - poc0) == abs(cur_poc - poc1) && w >= 8 && h >= 8;
}

// refined[k][list][x / y]: quarter-sample vectors of sub-block k (16x16 or the CU if smaller; raster order inside the CU) = mcore->dmvr_mv
static inline void dmvr_search_cu(int pic_w, int pic_h, int bd, int x, int y, int w, int h, const int16_t mv[2][2], const DmvrRefPlane ref[2],
                                  int16_t (*refined)[2][2], std::vector<int16_t> &scratch)
{
    enum { IT = 2, BOTTOM = 0, TOP, RIGHT, LEFT, DIAG, CENTER = 8 };
    const int stride = w + 2 * IT, dx = w < 16 ? w : 16, dy = h < 16 ? h : 16, maxv = (1 << bd) - 1;
    // mv_clip: the block may reach 128 samples out of the picture
    int16_t start[2][2];
    {
        const int min_c = -(128 << 2), max_x = (pic_w - 1 + 128) << 2, max_y = (pic_h - 1 + 128) << 2, qx = x << 2, qy = y << 2, qw = w << 2, qh = h << 2;
        for (int l = 0; l < 2; l++) {
            start[l][0] = mv[l][0]; start[l][1] = mv[l][1];
            if (qx + mv[l][0] < min_c) start[l][0] = (int16_t)(min_c - qx);
            if (qy + mv[l][1] < min_c) start[l][1] = (int16_t)(min_c - qy);
            if (qx + mv[l][0] + qw - 4 > max_x) start[l][0] = (int16_t)(max_x - qx - qw + 4);
            if (qy + mv[l][1] + qh - 4 > max_y) start[l][1] = (int16_t)(max_y - qy - qh + 4);
        }
    }
    // bilinear windows: (w + 4) x (h + 4) from two samples up-left of the starting position, taps { 64 - 4p, 4p } at the sixteenth-sample phase p, in the
    // rounding regimes of the long filters (copy / one direction: >> 6 and clip / both: >> shift1 into s16, then + offset >> shift2 and clip)
    const size_t plane = (size_t)stride * (size_t)(h + 2 * IT);
    scratch.resize(2 * plane + (size_t)stride * (size_t)(h + 2 * IT + 1));
    int16_t *bl[2] = { scratch.data(), scratch.data() + plane }, *tmp = scratch.data() + 2 * plane;
    const int shift1 = bd - 8 < 4 ? bd - 8 : 4, shift2 = 20 - bd > 8 ? 20 - bd : 8, off2 = 1 << (shift2 - 1);
    for (int l = 0; l < 2; l++) {
        const int gx = ((x << 2) + start[l][0] - (IT << 2)) << 2, gy = ((y << 2) + start[l][1] - (IT << 2)) << 2;
        const int px = gx & 15, py = gy & 15, ww = w + 2 * IT, hh = h + 2 * IT, s = ref[l].stride;
        const int tx0 = 64 - 4 * px, tx1 = 4 * px, ty0 = 64 - 4 * py, ty1 = 4 * py;
        const int16_t *r = ref[l].y + (gy >> 4) * s + (gx >> 4);
        int16_t *d = bl[l];
        if (!px && !py) {
            for (int i = 0; i < hh; i++) for (int j = 0; j < ww; j++) d[i * stride + j] = r[i * s + j];
        } else if (px && !py) {
            for (int i = 0; i < hh; i++) for (int j = 0; j < ww; j++) {
                const int v = (tx0 * r[i * s + j] + tx1 * r[i * s + j + 1]) >> 6;
                d[i * stride + j] = (int16_t)(v < 0 ? 0 : v > maxv ? maxv : v);
            }
        } else if (!px && py) {
            for (int i = 0; i < hh; i++) for (int j = 0; j < ww; j++) {
                const int v = (ty0 * r[i * s + j] + ty1 * r[(i + 1) * s + j]) >> 6;
                d[i * stride + j] = (int16_t)(v < 0 ? 0 : v > maxv ? maxv : v);
            }
        } else {
            for (int i = 0; i < hh + 1; i++) for (int j = 0; j < ww; j++) tmp[i * stride + j] = (int16_t)((tx0 * r[i * s + j] + tx1 * r[i * s + j + 1]) >> shift1);
            for (int i = 0; i < hh; i++) for (int j = 0; j < ww; j++) {
                const int v = (ty0 * tmp[i * stride + j] + ty1 * tmp[(i + 1) * stride + j] + off2) >> shift2;
                d[i * stride + j] = (int16_t)(v < 0 ? 0 : v > maxv ? maxv : v);
            }
        }
    }
    auto cost = [&](const int16_t *a, const int16_t *b) {
        int sad = 0;
        for (int i = 0; i < dy; i++) for (int j = 0; j < dx; j++) sad += abs(a[i * stride + j] - b[i * stride + j]);
        return sad;
    };
    auto div_q7 = [](long long n, long long d) {       // three bits of n / d
        int sign = 0, q = 0;
        if (n < 0) { sign = 1; n = -n; }
        d <<= 3;
        if (n >= d) { n -= d; q++; }
        q <<= 1; d >>= 1;
        if (n >= d) { n -= d; q++; }
        q <<= 1;
        if (n >= (d >> 1)) q++;
        return sign ? -q : q;
    };
    int num = 0;
    for (int sy = 0; sy < h; sy += dy) for (int sx = 0; sx < w; sx += dx, num++) {
        const int16_t *c0 = bl[0] + (IT + sy) * stride + IT + sx, *c1 = bl[1] + (IT + sy) * stride + IT + sx;
        int tot[2] = { 0, 0 }, not_zero = 1, min_cost = 0, cst[9];
        for (int k = 0; k < 9; k++) cst[k] = 0x7FFFFFFF;
        for (int i = 0; i < IT; i++) {
            const int16_t *a0 = c0 + tot[0] + tot[1] * stride, *a1 = c1 - (tot[0] + tot[1] * stride);
            int ox[5] = { 0, 0, 1, -1, 0 }, oy[5] = { 1, -1, 0, 0, 0 }, d[2] = { 0, 0 };
            for (int k = 0; k < 9; k++) cst[k] = 0x7FFFFFFF;
            if (i == 0) min_cost = cost(a0, a1);
            if ((i > 0 && min_cost == 0) || (i == 0 && min_cost < dx * dy)) { not_zero = 0; break; }
            cst[CENTER] = min_cost;
            for (int idx = BOTTOM; idx <= DIAG; idx++) {      // below, above, right, left, then the diagonal between the better two
                const int c = cost(a0 + ox[idx] + oy[idx] * stride, a1 - ox[idx] - oy[idx] * stride);
                cst[idx] = c;
                if (idx == LEFT) { ox[DIAG] = cst[RIGHT] <= cst[LEFT] ? 1 : -1; oy[DIAG] = cst[BOTTOM] <= cst[TOP] ? 1 : -1; }
                if (c < min_cost) { min_cost = c; d[0] = ox[idx]; d[1] = oy[idx]; }
            }
            if (d[0] == 0 && d[1] == 0) break;
            tot[0] += d[0]; tot[1] += d[1];
        }
        tot[0] <<= 4; tot[1] <<= 4;
        if (not_zero && min_cost == cst[CENTER]) {      // the centre of the last round won: parametric error surface through its cross
            const int sb[5] = { cst[CENTER], cst[LEFT], cst[TOP], cst[RIGHT], cst[BOTTOM] };
            for (int a = 0; a < 2; a++) {
                const long long nu = (long long)((sb[1 + a] - sb[3 + a]) << 4), de = (long long)(sb[1 + a] + sb[3 + a] - (sb[0] << 1));
                if (de != 0) tot[a] += (sb[1 + a] != sb[0] && sb[3 + a] != sb[0]) ? div_q7(nu, de) : (sb[1 + a] == sb[0] ? -8 : 8);
            }
        }
        for (int l = 0; l < 2; l++) {
            const int r0 = (start[l][0] << 2) + (l ? -tot[0] : tot[0]), r1 = (start[l][1] << 2) + (l ? -tot[1] : tot[1]);
            refined[num][l][0] = (int16_t)(r0 >> 2); refined[num][l][1] = (int16_t)(r1 >> 2);
        }
    }
}
