// intra_pred.h - what the intra kernels share (k_intra.hip: one wave per CU, global done flags; and, until round 4, a per-CTU formulation):
// the neighbour-array layout in LDS, the Baseline and EIPD predictors as functions of (column, row) over those arrays, packed reconstruction, coherent accesses.
#pragma once
#include "xgpu_internal.h"

#define NB_MAX 264      // up to 128 + 128 + 1 neighbour samples per side (luma of a 128x128 CU)

__device__ __forceinline__ uint32_t pack2i(int lo, int hi) { return ((uint32_t)lo & 0xFFFFu) | ((uint32_t)hi << 16); }
__device__ __forceinline__ int clip3i(int lo, int hi, int v) { return min(max(v, lo), hi); }
// rec = clip(0, max, (s16)(res + pred)) on packed pairs: the 16-bit sum wraps (xevd_recon.c:39,60)
// (packed 16-bit instructions - v_pk_add_u16, v_pk_max_i16, v_pk_min_i16: three for the pair; the halves taken apart, sign-extended, clipped and packed again were ten,
//  twelve pairs per SCU on the path of every link of a dependency chain)
__device__ __forceinline__ uint32_t recon2i(uint32_t pred, uint32_t res, int maxv)
{
    typedef short v2s_ __attribute__((ext_vector_type(2)));
    typedef unsigned short v2u_ __attribute__((ext_vector_type(2)));
    // (the sum as UNSIGNED halves: pred + res may pass 32767 and has to wrap like the reference's (s16) cast - signed vector overflow would be undefined)
    const v2s_ sum = __builtin_bit_cast(v2s_, __builtin_bit_cast(v2u_, pred) + __builtin_bit_cast(v2u_, res));
    const v2s_ r = __builtin_elementwise_min(__builtin_elementwise_max(sum, (v2s_)(0)), (v2s_)((short)maxv));
    return __builtin_bit_cast(uint32_t, r);
}

// Neighbour samples of one component in LDS, laid out along the block's "diagonal axis":
//   nb[NB_C0 - 1 - k] = left[k],  nb[NB_C0] = up[-1],  nb[NB_C0 + 1 + k] = up[k]   (k = 0 .. w+h-1)
//   nb[NB_UR + k] = (up[k] + left[k]) >> 1  (only filled for IPD_UR_B),  nb[NB_DC] = the DC value (only for IPD_DC_B).
// Every Baseline predictor (IPD_DC_B 0, HOR 1, VER 2, UL 3, UR 4; src_base/xevd_ipred.c:96-164, 587-622) of a 4x4 (2x2) block
// then reads at most 7 (3) CONSECUTIVE entries: HOR left[i], VER up[j], UL nb[NB_C0 + j - i], UR nb[NB_UR + i + j + 1].
#define NB_DC 0
#define NB_C0 261            // odd: up[k] pairs (k even) are 4-byte aligned
#define NB_UR 520
#define NB_LEN (NB_UR + 256)

// N consecutive neighbour entries for the block at (lx, ly): v[k]; the sample at row r, column q is v[sel(mode, r, q)]
template <int N, int C0 = NB_C0, int UR = NB_UR>
__device__ __forceinline__ void nb_fetch(const int16_t *nb, int mode, int lx, int ly, int v[N])
{
    int base = NB_DC, step = 0;
    if (mode == 1) { base = C0 - 1 - ly; step = -1; }
    if (mode == 2) { base = C0 + 1 + lx; step = 1; }
    if (mode == 3) { base = C0 + lx - ly - (N >> 1); step = 1; }
    if (mode == 4) { base = UR + lx + ly + 1; step = 1; }
#pragma unroll
    for (int k = 0; k < N; k++) v[k] = (uint16_t)nb[base + step * k];
}
__device__ __forceinline__ int nb_sel(int mode, int r, int q, int half)      // compile-time r, q; mode is wave-uniform
{
    return mode == 1 ? r : mode == 2 ? q : mode == 3 ? half + q - r : mode == 4 ? r + q : 0;
}

// Samples that other workgroups of the SAME launch wrote (or will read) move with agent-scope relaxed atomics, i.e. plain
// loads/stores with the sc1 bit: coherent across the XCD L2s without the bulk write-back / invalidate an agent-scope fence costs.
__device__ __forceinline__ uint32_t ld_coherent(const int16_t *p)      // p 4-byte aligned
{
    return __hip_atomic_load((uint32_t *)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void st_coherent(int16_t *p, uint32_t v)
{
    __hip_atomic_store((uint32_t *)p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void st_coherent2(int16_t *p, uint32_t lo, uint32_t hi)      // p 8-byte aligned
{
    __hip_atomic_store((uint64_t *)p, (uint64_t)lo | ((uint64_t)hi << 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// ---------------------------------------------------------------------------------------------------------
// sps->tool_eipd (Main): 33 luma modes - DC 0, planar 1, bilinear 2, angular 3..32 with VER 12 / HOR 24 - and chroma DM / BI / DC /
// HOR / VER (xevdm_ipred / xevdm_ipred_uv, src_main/xevdm_ipred.c:241-305; predictors src_base/xevd_ipred.c:110-585).  The neighbour arrays follow xevdm_get_nbr (xevdm_ipred.c:39-148): an unavailable
// unit REPEATS the sample before it (towards the corner; the mid value only where nothing precedes), an unavailable corner takes
// up[0].  Every predictor is a function of (i, j) and the arrays, so a lane evaluates its 4x4 (2x2) samples independently.
// ---------------------------------------------------------------------------------------------------------
struct EipdPlan { int mode, p0, p1, p2, lr; };  // wave-uniform: DC p0 = value; planar p0 = base, p1 = b, p2 = c; bilinear p0 = a, p1 = b, p2 = wt; angular p0 = dx, p1 = dy
                                                // lr = avail_lr (xevd_check_nev_avail): bit 0 the column left of the block is reconstructed, bit 1 the one to its right (SUCO)
// The right column (sps_suco_flag: a split coded right to left leaves a block with its RIGHT neighbours reconstructed; xevdm_get_nbr :123-147) takes the place of the
// Baseline-only arrays: right[p] = A[NB_UR + 1 + p], p = -1 .. w+h-1, right[-1] = up[w].  It is staged, and the LR_01 / LR_11 forms of the predictors run, only
// for blocks with lr & 2; everything else takes the paths below unchanged.

// sum over the 64 lanes, the same value in every lane: two quad permutes and two mirrors in DPP leave every lane of a row of 16 with the row's sum, the four row
// sums meet on the scalar unit (six rounds of __shfl_xor were six LDS-crossbar round trips on the critical path of every DC / planar CU of a dependency chain)
__device__ __forceinline__ int row_sum16(int v)      // the sum over a row of 16 lanes, in every lane of the row
{
    v += __builtin_amdgcn_update_dpp(0, v, 0xB1, 0xF, 0xF, true);      // quad_perm [1, 0, 3, 2]
    v += __builtin_amdgcn_update_dpp(0, v, 0x4E, 0xF, 0xF, true);      // quad_perm [2, 3, 0, 1]
    v += __builtin_amdgcn_update_dpp(0, v, 0x141, 0xF, 0xF, true);     // row_half_mirror
    v += __builtin_amdgcn_update_dpp(0, v, 0x140, 0xF, 0xF, true);     // row_mirror
    return v;
}
__device__ __forceinline__ int wave_sum(int v)
{
    v += __builtin_amdgcn_update_dpp(0, v, 0xB1, 0xF, 0xF, true);      // quad_perm [1, 0, 3, 2]
    v += __builtin_amdgcn_update_dpp(0, v, 0x4E, 0xF, 0xF, true);      // quad_perm [2, 3, 0, 1]
    v += __builtin_amdgcn_update_dpp(0, v, 0x141, 0xF, 0xF, true);     // row_half_mirror
    v += __builtin_amdgcn_update_dpp(0, v, 0x140, 0xF, 0xF, true);     // row_mirror
    return __builtin_amdgcn_readlane(v, 0) + __builtin_amdgcn_readlane(v, 16) + __builtin_amdgcn_readlane(v, 32) + __builtin_amdgcn_readlane(v, 48);
}
__device__ __forceinline__ void ang_slopes(int mode, int &dx, int &dy)      // xevd_tbl_ipred_dxdy (xevd_tbl.c:294-304): slopes in 1/1024
{
    const int u[11] = { 128, 256, 372, 512, 744, 1024, 1408, 2048, 2816, 4096, 8192 };
    int a = 0, b = 0;
    if (mode >= 3 && mode <= 11)       { a = 11 - mode; b = mode - 1; }
    else if (mode >= 13 && mode <= 23) { a = mode - 13; b = 23 - mode; }
    else if (mode >= 25 && mode <= 32) { a = 35 - mode; b = mode - 25; }
    dx = 0; dy = 0;
#pragma unroll
    for (int k = 0; k < 11; k++) { if (k == a) dx = u[k]; if (k == b) dy = u[k]; }
}
// mode: a LUMA mode number (chroma modes are mapped by the caller); A = the component's neighbour array; all lanes take part
// ... with a reconstructed right column (lr 2: only that one, lr 3: both): xevdm_ipred_dc / xevdm_ipred_hor (xevdm_ipred.c:153-229), xevd_ipred_plane / xevd_ipred_bi (xevd_ipred.c:
// 163-369), first branches
__device__ __forceinline__ EipdPlan eipd_plan_lr(const int16_t *A, int mode, int w, int h, int lw, int lh, int t, int lr)
{
    EipdPlan k = { mode, 0, 0, 0, lr };
    const int16_t *up = A + NB_C0 + 1, *ri = A + NB_UR + 1;
    const int inv[8] = { 2048, 1365, 819, 455, 241, 124, 63, 32 };
    int inv_w = 0;
#pragma unroll
    for (int q = 0; q < 8; q++) if (q == lw) inv_w = inv[q];
    if (mode == 0) {
        int acc = 0;
        for (int e = t; e < w + h; e += 64) acc += e < h ? ((lr & 1) ? A[NB_C0 - 1 - e] : 0) + ri[e] : up[e - h];
        const int lhh = lr == 3 ? lh + 1 : lh, hh = lr == 3 ? h << 1 : h, asp = lw > lhh ? lw - lhh : lhh - lw;
        int m = 0;
#pragma unroll
        for (int q = 0; q < 8; q++) if (q == asp) m = inv[q];
        k.p0 = ((wave_sum(acc) + ((w + hh) >> 1)) * m) >> (min(lw, lhh) + 12);
    } else if (mode == 1) {                            // the mirror image: gradients towards the left, anchored at the bottom-right and top-left samples
        const int mult[6] = { 13, 17, 5, 11, 23, 47 }, shift[6] = { 7, 10, 11, 15, 19, 23 };
        const int w2 = w >> 1, h2 = h >> 1, iw = max(lw - 2, 0), ih = max(lh - 2, 0);
        int ch = 0, cv = 0;
        for (int x = 1 + t; x <= w2; x += 64) ch += x * (up[w2 - x] - up[w2 + x]);
        for (int y = 1 + t; y <= h2; y += 64) cv += y * (ri[h2 - 1 + y] - ri[h2 - 1 - y]);
        ch = wave_sum(ch); cv = wave_sum(cv);
        int mh = 0, sh = 0, mv = 0, sv = 0;
#pragma unroll
        for (int q = 0; q < 6; q++) { if (q == iw) { mh = mult[q]; sh = shift[q]; } if (q == ih) { mv = mult[q]; sv = shift[q]; } }
        const int a = (ri[h - 1] + up[0]) << 4;
        k.p1 = ((ch << 5) * mh + (1 << (sh - 1))) >> sh;
        k.p2 = ((cv << 5) * mv + (1 << (sv - 1))) >> sv;
        k.p0 = a - (h2 - 1) * k.p2 - (w2 - 1) * k.p1 + 16;
    } else if (mode == 2 && lr == 2) {
        const int wc_tbl[6] = { -1, 341, 205, 114, 60, 31 };
        const int a = up[-1], b = ri[h], ms = min(lw, lh), asp = lw > lh ? lw - lh : lh - lw;
        int wc = 0;
#pragma unroll
        for (int q = 0; q < 6; q++) if (q == asp) wc = wc_tbl[q];
        const int c = w == h ? (a + b + 1) >> 1 : (((a << lw) + (b << lh)) * wc + (1 << (ms + 9))) >> (ms + 10);
        k.p0 = a; k.p1 = b; k.p2 = (c << 1) - a - b;
    } else if (mode == 2 || mode == 24) {
        k.p0 = inv_w;                                  // both columns: rows interpolated between them with 4096 / (w + 1)
    } else if (mode != 12) {
        ang_slopes(mode, k.p0, k.p1);
    }
    return k;
}
// ... and the sample at column i, row j of such a block (ipred_ang_val's right-column branches, xevd_ipred.c:377-569)
__device__ __forceinline__ int eipd_sample_lr(const int16_t *A, const EipdPlan &k, int i, int j, int w, int h, int lw, int lh, int maxv)
{
    const int mode = k.mode, lr = k.lr;
    const int16_t *up = A + NB_C0 + 1, *ri = A + NB_UR + 1;
    auto le = [&](int p) -> int { return A[NB_C0 - 1 - p]; };
    if (mode == 12) return up[i];
    if (mode == 24) return lr == 3 ? ((le(j) * (w - i) + ri[j] * (i + 1) + (w >> 1)) * k.p0) >> 12 : ri[j];
    if (mode == 0) return k.p0;
    if (mode == 1) return clip3i(0, maxv, (k.p0 + j * k.p2 + (w - 1 - i) * k.p1) >> 5);
    if (mode == 2) {
        if (lr == 3) {
            const int row = ((le(j) * (w - i) + ri[j] * (i + 1) + (w >> 1)) * k.p0) >> 12, bot = ((le(h - 1) * (w - i) + ri[h - 1] * (i + 1) + (w >> 1)) * k.p0) >> 12;
            return (row + ((up[i] * (h - 1 - j) + bot * (j + 1) + (h >> 1)) >> lh) + 1) >> 1;
        }
        const int d = w - 1 - i, col = ri[j], u = up[i];
        const int px = (col << lw) + (d + 1) * (k.p0 - col), py = (u << lh) + (j + 1) * (k.p1 - u);
        return clip3i(0, maxv, ((px << lh) + (py << lw) + d * j * k.p2 + (1 << (lw + lh))) >> (lw + lh + 1));
    }
    int p, o, dir, src;                                // src 0 up, 1 left, 2 right
    auto pos = [&](int mt, int d, int &q) { q = (d * mt) >> 10; o = ((d * mt) >> 5) & 31; };
    if (mode < 12) {
        int tq;
        pos(k.p0, j + 1, tq);
        if (i >= w - tq) { pos(k.p1, w - i, tq); p = j - tq; src = 2; dir = -1; }
        else { p = i + tq; src = 0; dir = 1; }
    } else if (mode > 24) {
        int tq;
        pos(k.p1, w - i, tq);
        if (j < tq) { pos(k.p0, w - i, tq); p = i + tq; src = 0; dir = 1; }
        else { p = j - tq; src = 2; dir = -1; }
    } else {
        int ty, tq;
        pos(k.p1, i + 1, ty);
        if (j < ty) { pos(k.p0, j + 1, tq); p = i - tq; src = 0; dir = -1; }
        else if (lr == 2) { pos(k.p1, w - i, tq); p = j + tq; src = 2; dir = 1; }
        else { p = j - ty; src = 1; dir = -1; }
    }
    const int hi = w + h - 1;
    auto ref = [&](int q) -> int { q = clip3i(-1, hi, q); return src == 0 ? up[q] : src == 1 ? le(q) : ri[q]; };
    return clip3i(0, maxv, (int)(int16_t)((ref(p - dir) * (32 - o) + ref(p) * (64 - o) + ref(p + dir) * (32 + o) + ref(p + 2 * dir) * o + 64) >> 7));
}
__device__ __forceinline__ EipdPlan eipd_plan(const int16_t *A, int mode, int w, int h, int lw, int lh, int t, int lr = 0)
{
    if (lr & 2) return eipd_plan_lr(A, mode, w, h, lw, lh, t, lr);
    EipdPlan k = { mode, 0, 0, 0, 0 };
    const int16_t *up = A + NB_C0 + 1;
    if (mode == 0) {                                   // xevdm_ipred_dc + xevd_get_dc (xevd_ipred.c:124-144): 4096 / (2^k + 1) scaling of non-square sums
        const int inv[8] = { 2048, 1365, 819, 455, 241, 124, 63, 32 };
        int acc = 0;
        for (int e = t; e < w + h; e += 64) acc += e < h ? A[NB_C0 - 1 - e] : up[e - h];
        const int asp = lw > lh ? lw - lh : lh - lw;
        int m = 0;
#pragma unroll
        for (int q = 0; q < 8; q++) if (q == asp) m = inv[q];
        k.p0 = ((wave_sum(acc) + ((w + h) >> 1)) * m) >> (min(lw, lh) + 12);
    } else if (mode == 1) {                            // xevd_ipred_plane (:163-249), left-neighbour branch
        const int mult[6] = { 13, 17, 5, 11, 23, 47 }, shift[6] = { 7, 10, 11, 15, 19, 23 };
        const int w2 = w >> 1, h2 = h >> 1, iw = max(lw - 2, 0), ih = max(lh - 2, 0);
        int ch = 0, cv = 0;
        for (int x = 1 + t; x <= w2; x += 64) ch += x * (up[w2 - 1 + x] - up[w2 - 1 - x]);
        for (int y = 1 + t; y <= h2; y += 64) cv += y * (A[NB_C0 - 1 - (h2 - 1 + y)] - A[NB_C0 - 1 - (h2 - 1 - y)]);
        ch = wave_sum(ch); cv = wave_sum(cv);
        int mh = 0, sh = 0, mv = 0, sv = 0;
#pragma unroll
        for (int q = 0; q < 6; q++) { if (q == iw) { mh = mult[q]; sh = shift[q]; } if (q == ih) { mv = mult[q]; sv = shift[q]; } }
        const int a = (A[NB_C0 - 1 - (h - 1)] + up[w - 1]) << 4;
        k.p1 = ((ch << 5) * mh + (1 << (sh - 1))) >> sh;
        k.p2 = ((cv << 5) * mv + (1 << (sv - 1))) >> sv;
        k.p0 = a - (h2 - 1) * k.p2 - (w2 - 1) * k.p1 + 16;
    } else if (mode == 2) {                            // xevd_ipred_bi (:251-369), left-neighbour branch
        const int wc_tbl[6] = { -1, 341, 205, 114, 60, 31 };
        const int a = up[w], b = A[NB_C0 - 1 - h], ms = min(lw, lh), asp = lw > lh ? lw - lh : lh - lw;
        int wc = 0;
#pragma unroll
        for (int q = 0; q < 6; q++) if (q == asp) wc = wc_tbl[q];
        const int c = w == h ? (a + b + 1) >> 1 : (((a << lw) + (b << lh)) * wc + (1 << (ms + 9))) >> (ms + 10);
        k.p0 = a; k.p1 = b; k.p2 = (c << 1) - a - b;
    } else if (mode != 12 && mode != 24) {
        ang_slopes(mode, k.p0, k.p1);
    }
    return k;
}
// the predicted sample at column i, row j (ipred_ang_val :377-569 for the angular modes: 4 taps { 32-o, 64-o, 32+o, o } / 128 between
// reference positions clamped to [-1, w+h-1]; up[p] = A[NB_C0 + 1 + p], left[p] = A[NB_C0 - 1 - p], both with p = -1 at the corner)
__device__ __forceinline__ int eipd_sample(const int16_t *A, const EipdPlan &k, int i, int j, int w, int h, int lw, int lh, int maxv)
{
    const int mode = k.mode;
    if (mode == 12) return A[NB_C0 + 1 + i];
    if (mode == 24) return A[NB_C0 - 1 - j];
    if (mode == 0) return k.p0;
    if (mode == 1) return clip3i(0, maxv, (k.p0 + j * k.p2 + i * k.p1) >> 5);
    if (mode == 2) {
        const int le = A[NB_C0 - 1 - j], u = A[NB_C0 + 1 + i];
        const int px = (le << lw) + (i + 1) * (k.p0 - le), py = (u << lh) + (j + 1) * (k.p1 - u);
        return clip3i(0, maxv, ((px << lh) + (py << lw) + i * j * k.p2 + (1 << (lw + lh))) >> (lw + lh + 1));
    }
    int p, o, sgn, dir;                                // reference = A[NB_C0 + sgn * (1 + position)]
    if (mode < 12)      { const int tt = (j + 1) * k.p0; p = i + (tt >> 10); o = (tt >> 5) & 31; sgn = 1; dir = 1; }
    else if (mode > 24) { const int tt = (i + 1) * k.p1; p = j + (tt >> 10); o = (tt >> 5) & 31; sgn = -1; dir = 1; }
    else {
        const int ty = (i + 1) * k.p1;
        if (j < (ty >> 10)) { const int tx = (j + 1) * k.p0; p = i - (tx >> 10); o = (tx >> 5) & 31; sgn = 1; }
        else                { p = j - (ty >> 10); o = (ty >> 5) & 31; sgn = -1; }
        dir = -1;
    }
    const int hi = w + h - 1;
    const int r0 = A[NB_C0 + sgn * (1 + clip3i(-1, hi, p - dir))], r1 = A[NB_C0 + sgn * (1 + clip3i(-1, hi, p))];
    const int r2 = A[NB_C0 + sgn * (1 + clip3i(-1, hi, p + dir))], r3 = A[NB_C0 + sgn * (1 + clip3i(-1, hi, p + 2 * dir))];
    return clip3i(0, maxv, (r0 * (32 - o) + r1 * (64 - o) + r2 * (32 + o) + r3 * o + 64) >> 7);
}

// N consecutive samples of row j from column i0: eipd_sample for each of them with the mode looked at once (eipd_sample's chain of uniform branches per sample
// was most of a prediction step: a lone wave pays every taken branch in full), and for the vertical angular modes (3 .. 11) one offset / one filter phase per row
// and N + 3 reference reads instead of 4 N.  Same arithmetic, sample for sample.
template <int N>
__device__ __forceinline__ void eipd_row(const int16_t *A, const EipdPlan &k, int i0, int j, int w, int h, int lw, int lh, int maxv, int out[N])
{
    const int mode = k.mode, hi = w + h - 1;
    if (k.lr & 2) {                                    // (wave-uniform) a block with a reconstructed right column
#pragma unroll 1
        for (int q = 0; q < N; q++) {                  // (one copy of the sample function per row form: this path is rare, the kernel's size is not)
            const int v = eipd_sample_lr(A, k, i0 + q, j, w, h, lw, lh, maxv);
#pragma unroll
            for (int m = 0; m < N; m++) if (m == q) out[m] = v;
        }
        return;
    }
    if (mode == 12) {
#pragma unroll
        for (int q = 0; q < N; q++) out[q] = A[NB_C0 + 1 + i0 + q];
    } else if (mode == 24) {
        const int v = A[NB_C0 - 1 - j];
#pragma unroll
        for (int q = 0; q < N; q++) out[q] = v;
    } else if (mode == 0) {
#pragma unroll
        for (int q = 0; q < N; q++) out[q] = k.p0;
    } else if (mode == 1) {
        const int base = k.p0 + j * k.p2 + i0 * k.p1;
#pragma unroll
        for (int q = 0; q < N; q++) out[q] = clip3i(0, maxv, (base + q * k.p1) >> 5);
    } else if (mode == 2) {
        const int le = A[NB_C0 - 1 - j];
#pragma unroll
        for (int q = 0; q < N; q++) {
            const int i = i0 + q, u = A[NB_C0 + 1 + i];
            const int px = (le << lw) + (i + 1) * (k.p0 - le), py = (u << lh) + (j + 1) * (k.p1 - u);
            out[q] = clip3i(0, maxv, ((px << lh) + (py << lw) + i * j * k.p2 + (1 << (lw + lh))) >> (lw + lh + 1));
        }
    } else if (mode < 12) {
        const int tt = (j + 1) * k.p0, o = (tt >> 5) & 31, p0 = i0 + (tt >> 10) - 1;
        int r[N + 3];
#pragma unroll
        for (int m = 0; m < N + 3; m++) r[m] = A[NB_C0 + 1 + clip3i(-1, hi, p0 + m)];
#pragma unroll
        for (int q = 0; q < N; q++) out[q] = clip3i(0, maxv, (r[q] * (32 - o) + r[q + 1] * (64 - o) + r[q + 2] * (32 + o) + r[q + 3] * o + 64) >> 7);
    } else if (mode > 24) {
#pragma unroll
        for (int q = 0; q < N; q++) {
            const int tt = (i0 + q + 1) * k.p1, p = j + (tt >> 10), o = (tt >> 5) & 31;
            const int r0 = A[NB_C0 - 1 - clip3i(-1, hi, p - 1)], r1 = A[NB_C0 - 1 - clip3i(-1, hi, p)], r2 = A[NB_C0 - 1 - clip3i(-1, hi, p + 1)], r3 = A[NB_C0 - 1 - clip3i(-1, hi, p + 2)];
            out[q] = clip3i(0, maxv, (r0 * (32 - o) + r1 * (64 - o) + r2 * (32 + o) + r3 * o + 64) >> 7);
        }
    } else {
        const int tx = (j + 1) * k.p0;
#pragma unroll
        for (int q = 0; q < N; q++) {
            const int i = i0 + q, ty = (i + 1) * k.p1;
            const bool from_up = j < (ty >> 10);
            const int p = from_up ? i - (tx >> 10) : j - (ty >> 10), o = ((from_up ? tx : ty) >> 5) & 31, sgn = from_up ? 1 : -1;
            const int r0 = A[NB_C0 + sgn * (1 + clip3i(-1, hi, p + 1))], r1 = A[NB_C0 + sgn * (1 + clip3i(-1, hi, p))];
            const int r2 = A[NB_C0 + sgn * (1 + clip3i(-1, hi, p - 1))], r3 = A[NB_C0 + sgn * (1 + clip3i(-1, hi, p - 2))];
            out[q] = clip3i(0, maxv, (r0 * (32 - o) + r1 * (64 - o) + r2 * (32 + o) + r3 * o + 64) >> 7);
        }
    }
}

__device__ __forceinline__ void wave_lds_sync()      // LDS traffic of one wave is processed in order: only the compiler needs the fence
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
__device__ __forceinline__ uint32_t uni(uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); }
