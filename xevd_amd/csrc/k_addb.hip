// k_addb.hip - ADDB deblocking (Main profile, sps->tool_addb), vertical-edge pass and horizontal-edge pass.
//
// Replaces xevdm_deblock -> deblock_tree -> xevdm_deblock_cu_ver / _hor -> deblock_addb_cu_* -> deblock_scu_line_*
// (src_main/xevdm.c:1935-2103, src_main/xevdm_df.c:361-1135).  Semantics reproduced: only CU edges on the 8x8 luma
// grid; boundary strength 0..4 from intra / CTU crossing / luma cbf / reference PICTURES + MVs (get_bs); QP =
// average of both sides; alpha/beta/clip tables indexed through get_index()'s u8 arguments; luma strong (bS 4)
// and normal filters over 3 samples per side, chroma over 1; all vertical edges before all horizontal ones.
//
// MI355X mapping - out of place (SRC -> DST) and order-free: grid edges are 8 samples apart and touch 3 samples per side, so edges never interact and
// the 8-sample windows centred on the grid lines tile the picture in both directions: one LANE per 4-sample edge SEGMENT owns the SCU on either side
// (P = left / above, Q = right / below), filters once and writes both halves - every sample is read once and written once, no filter is evaluated
// twice.  Both edge directions run in ONE kernel (k_addb_fused below; round 2 had a kernel per direction: two reads and two writes of the picture).
// All loads are issued before any decision; decisions are lane-local integer tests, tables in LDS.
#include "xgpu_internal.h"

struct __attribute__((packed, aligned(8))) U32x4a8 { uint32_t a, b, c, d; };
struct __attribute__((packed, aligned(4))) U32x2a4 { uint32_t a, b; };

__device__ __forceinline__ int clip3a(int lo, int hi, int v) { return min(max(v, lo), hi); }

// ALPHA_TABLE / BETA_TABLE / CLIP_TAB (src_main/xevdm_tbl.c:377-379) - tables of the EVC specification
__constant__ uint8_t k_alpha[52] = { 0,0,0,0,0,0,0,0,0,0,0,0, 0,0,0,0,4,4,5,6, 7,8,9,10,12,13,15,17, 20,22,25,28,32,36,40,45,
    50,56,63,71,80,90,101,113, 127,144,162,182,203,226,255,255 };
__constant__ uint8_t k_beta[52] = { 0,0,0,0,0,0,0,0,0,0,0,0, 0,0,0,0,2,2,2,3, 3,3,3,4,4,4,6,6, 7,7,8,8,9,9,10,10,
    11,11,12,12,13,13,14,14, 15,15,16,16,17,17,18,18 };
__constant__ uint8_t k_clip[52][5] = {
    {0,0,0,0,0},{0,0,0,0,0},{0,0,0,0,0},{0,0,0,0,0},{0,0,0,0,0},{0,0,0,0,0},{0,0,0,0,0},{0,0,0,0,0},
    {0,0,0,0,0},{0,0,0,0,0},{0,0,0,0,0},{0,0,0,0,0},{0,0,0,0,0},{0,0,0,0,0},{0,0,0,0,0},{0,0,0,0,0},
    {0,0,0,0,0},{0,0,0,1,1},{0,0,0,1,1},{0,0,0,1,1},{0,0,0,1,1},{0,0,1,1,1},{0,0,1,1,1},{0,1,1,1,1},
    {0,1,1,1,1},{0,1,1,1,1},{0,1,1,1,1},{0,1,1,2,2},{0,1,1,2,2},{0,1,1,2,2},{0,1,1,2,2},{0,1,2,3,3},
    {0,1,2,3,3},{0,2,2,3,3},{0,2,2,4,4},{0,2,3,4,4},{0,2,3,4,4},{0,3,3,5,5},{0,3,4,6,6},{0,3,4,6,6},
    {0,4,5,7,7},{0,4,5,8,8},{0,4,6,9,9},{0,5,7,10,10},{0,6,8,11,11},{0,6,8,13,13},{0,7,10,14,14},{0,8,11,16,16},
    {0,9,12,18,18},{0,10,13,20,20},{0,11,15,23,23},{0,13,17,25,25} };

// get_bs, xevdm_df.c:361-513.  q = record of the right/below SCU, p = left/above; cross_ctu: the edge lies on a CTU boundary.
__device__ __forceinline__ int addb_bs(const uint4 q, const uint4 p, bool cross_ctu, const uint8_t *pic_id)
{
    const bool intra = ((q.x | p.x) >> 15) & 1;
    if (intra) return cross_ctu ? 4 : 3;
    if (((q.x | p.x) >> 26) & 1) return 3;                 // IBC on either side (xevdm_df.c:411-414)
    if ((((q.x | p.x) >> 24) & 1) || ((q.y | p.y) >> 16)) return 2;      // luma cbf, or ATS-inter on either side (ats_present, xevdm_df.c:415)
    const int q0 = (int8_t)(q.y & 0xFF), q1 = (int8_t)((q.y >> 8) & 0xFF), p0 = (int8_t)(p.y & 0xFF), p1 = (int8_t)((p.y >> 8) & 0xFF);
    // reference pictures by identity (XEVD_PIC pointers in the reference): device picture slot, 255 = none
    const int Q0 = q0 >= 0 ? pic_id[q0 * 2] : 255, Q1 = q1 >= 0 ? pic_id[q1 * 2 + 1] : 255;
    const int P0 = p0 >= 0 ? pic_id[p0 * 2] : 255, P1 = p1 >= 0 ? pic_id[p1 * 2 + 1] : 255;
    const int qm[2][2] = { { q0 >= 0 ? (int16_t)(q.z & 0xFFFF) : 0, q0 >= 0 ? (int16_t)(q.z >> 16) : 0 },
                           { q1 >= 0 ? (int16_t)(q.w & 0xFFFF) : 0, q1 >= 0 ? (int16_t)(q.w >> 16) : 0 } };
    const int pm[2][2] = { { p0 >= 0 ? (int16_t)(p.z & 0xFFFF) : 0, p0 >= 0 ? (int16_t)(p.z >> 16) : 0 },
                           { p1 >= 0 ? (int16_t)(p.w & 0xFFFF) : 0, p1 >= 0 ? (int16_t)(p.w >> 16) : 0 } };
#define MVSAME(a, b) (abs((a)[0] - (b)[0]) < 4 && abs((a)[1] - (b)[1]) < 4)
    if ((Q0 == P0 && Q1 == P1) || (Q0 == P1 && Q1 == P0)) {
        if (Q0 == Q1) return (MVSAME(qm[0], pm[0]) && MVSAME(qm[1], pm[1]) && MVSAME(qm[0], pm[1]) && MVSAME(qm[1], pm[0])) ? 0 : 1;
        if (Q0 == P0 && Q1 == P1) return (MVSAME(qm[0], pm[0]) && MVSAME(qm[1], pm[1])) ? 0 : 1;
        return (MVSAME(qm[0], pm[1]) && MVSAME(qm[1], pm[0])) ? 0 : 1;
    }
    return 1;
#undef MVSAME
}

// deblock_scu_line_luma, xevdm_df.c:584-709.  s[0..3] = p3 p2 p1 p0, s[4..7] = q0 q1 q2 q3; in place.
__device__ __forceinline__ void addb_line_luma(int s[8], int bs, int alpha, int beta, int c1, int bd, int maxv)
{
    const int p0 = s[3], p1 = s[2], p2 = s[1], p3 = s[0], q0 = s[4], q1 = s[5], q2 = s[6], q3 = s[7];
    if (!(bs && abs(p0 - q0) < alpha && abs(p1 - p0) < beta && abs(q1 - q0) < beta)) return;
    const int ap = abs(p0 - p2) < beta, aq = abs(q0 - q2) < beta;
    int po0 = p0, po1 = p1, po2 = p2, qo0 = q0, qo1 = q1, qo2 = q2;
    if (bs == 4) {
        const bool strong = abs(p0 - q0) < ((alpha >> 2) + 2);
        if (ap && strong) {
            po0 = (p2 + 2 * (p1 + p0 + q0) + q1 + 4) >> 3;
            po1 = (p2 + p1 + p0 + q0 + 2) >> 2;
            po2 = (2 * p3 + 3 * p2 + p1 + p0 + q0 + 4) >> 3;
        } else po0 = (2 * p1 + p0 + q1 + 2) >> 2;
        if (aq && strong) {
            qo0 = (q2 + 2 * (q1 + q0 + p0) + p1 + 4) >> 3;
            qo1 = (q2 + q1 + q0 + p0 + 2) >> 2;
            qo2 = (2 * q3 + 3 * q2 + q1 + q0 + p0 + 4) >> 3;
        } else qo0 = (2 * q1 + q0 + p1 + 2) >> 2;
    } else {
        const int c0 = (c1 + ((ap + aq) << max(0, bd - 9))) & 0xFF;          // u8 c0, xevdm_df.c:650
        const int d0 = clip3a(-c0, c0, (4 * (q0 - p0) + p1 - q1 + 4) >> 3);
        po0 = clip3a(0, maxv, p0 + d0);
        qo0 = clip3a(0, maxv, q0 - d0);
        if (ap) po1 = p1 + clip3a(-c1, c1, (((p2 + p0 + q0) * 3) - 8 * p1 - q1) >> 4);
        if (aq) qo1 = q1 + clip3a(-c1, c1, (((q2 + q0 + p0) * 3) - 8 * q1 - p1) >> 4);
    }
    s[3] = clip3a(0, maxv, (int)(int16_t)po0); s[2] = clip3a(0, maxv, (int)(int16_t)po1); s[1] = clip3a(0, maxv, (int)(int16_t)po2);
    s[4] = clip3a(0, maxv, (int)(int16_t)qo0); s[5] = clip3a(0, maxv, (int)(int16_t)qo1); s[6] = clip3a(0, maxv, (int)(int16_t)qo2);
}
// deblock_scu_line_chroma, xevdm_df.c:710-781.  s = p1 p0 q0 q1; only p0, q0 change.
__device__ __forceinline__ void addb_line_chroma(int s[4], int bs, int alpha, int beta, int c0, int maxv)
{
    const int p1 = s[0], p0 = s[1], q0 = s[2], q1 = s[3];
    if (!(bs && abs(p0 - q0) < alpha && abs(p1 - p0) < beta && abs(q1 - q0) < beta)) return;
    if (bs == 4) {
        s[1] = clip3a(0, maxv, (2 * p1 + p0 + q1 + 2) >> 2);
        s[2] = clip3a(0, maxv, (2 * q1 + q0 + p1 + 2) >> 2);
    } else {
        const int d0 = clip3a(-c0, c0, (4 * (q0 - p0) + p1 - q1 + 4) >> 3);
        s[1] = clip3a(0, maxv, p0 + d0);
        s[2] = clip3a(0, maxv, q0 - d0);
    }
}

__device__ __forceinline__ int addb_index(int qp, int offset) { return clip3a(0, 51, (qp & 0xFF) + (offset & 0xFF)); }   // u8 arguments

// The decisions and filters of one 4-sample edge segment (deblock_addb_cu_hor :893-944 / deblock_addb_cu_ver_yuv :947-1034): rq / rp = the SCU records
// after / before the grid line, eq = the Q side's SCU position along the filtered axis, L / Cc = the windows (filtered in place).
template <int DIR>
__device__ __forceinline__ void addb_edge(const AddbArgs &a, const uint4 rq, const uint4 rp, int eq, int L[4][8], int Cc[2][2][4],
                                          const uint8_t *s_alpha, const uint8_t *s_beta, const uint8_t *s_clip, const int8_t *s_cqp, const uint8_t *s_pic)
{
    const uint32_t eflag = DIR == 0 ? SCU_EDGE_L : SCU_EDGE_T;
    const uint32_t nflag = DIR == 0 ? SCU_NOCH_L : SCU_NOCH_T;      // a luma CU's edge inside the chroma block of a local dual tree: luma only (xevdm_df.c:916-920, 986-997)
    const int maxl = (1 << a.bd_l) - 1, maxc = (1 << a.bd_c) - 1;
    // an edge on a tile border stays as it is unless the PPS filters across tiles (no_boundary, src_main/xevdm_df.c:877, 1088, 1106)
    const int ctu_sh = a.log2_ctu - 2;
    const bool tile_edge = (eq & ((1 << ctu_sh) - 1)) == 0 && (DIR == 0 ? a.no_filter.col_start((eq >> ctu_sh) & 255) : a.no_filter.row_start((eq >> ctu_sh) & 255));
    if (!(rq.x & eflag) || tile_edge) return;
    const int epos = eq << 2;
    const bool cross = (epos & ((1 << a.log2_ctu) - 1)) == 0;
    const int bs = addb_bs(rq, rp, cross, s_pic);
    const int qp = (((rq.x >> 16) & 0x7F) + ((rp.x >> 16) & 0x7F) + 1) >> 1;
    const int scale = a.bd_l - 8;
    {
        const int ia = addb_index(qp, a.alpha_off), ib = addb_index(qp, a.beta_off);
        const int alpha = s_alpha[ia] << scale, beta = (s_beta[ib] << scale) & 0xFF;
        const int c1 = (s_clip[ia * 5 + bs] << max(0, a.bd_l - 9)) & 0xFF;
#pragma unroll
        for (int r = 0; r < 4; r++) addb_line_luma(L[r], bs, alpha, beta, c1, a.bd_l, maxl);
    }
    const int boff = 6 * (a.bd_c - 8);
    if (!(rq.x & nflag))
#pragma unroll
    for (int pl = 0; pl < 2; pl++) {
        const int q = clip3a(-boff, 57, qp + (pl ? a.qp_v_off : a.qp_u_off));
        const int qc = s_cqp[pl * 96 + q + boff];
        const int ia = addb_index(qc, a.alpha_off), ib = addb_index(qc, a.beta_off);
        const int alpha = s_alpha[ia] << scale, beta = (s_beta[ib] << scale) & 0xFF;      // luma depth scales chroma too (:926-927)
        const int c0 = ((s_clip[ia * 5 + bs] + 1) << max(0, a.bd_c - 9)) & 0xFF;
#pragma unroll
        for (int r = 0; r < 2; r++) addb_line_chroma(Cc[pl][r], bs, alpha, beta, c0, maxc);
    }
}

// ---------------------------------------------------------------------------------------------------------
// k_addb_fused - both edge directions in ONE kernel: one read and one write of the picture instead of two of each.
//
// Grid edges are 8 samples apart and a filter touches at most 4 samples on either side (3 written), so the 8-sample windows centred on the grid lines
// tile the picture in BOTH directions.  A workgroup therefore owns the tile [X0 - 4, X0 + 252) x [Y0 - 4, Y0 + 12) - shifted by half a window against
// the 8x8 grid - which holds 32 x 4 complete vertical-edge windows (8 samples x 4 rows) and, at the same time, 64 x 2 complete horizontal-edge windows
// (4 samples x 8 rows): no halo in either direction, nothing is read or filtered twice.
//   phase V: lane = one vertical-edge segment: SCU records + its luma / chroma windows from HBM (a wave's loads are two 512-byte runs per
//            row), filter, windows -> LDS, the two SCU records -> LDS (the horizontal phase needs the same 64 x 8 records);
//   barrier;
//   phase H: lane = one horizontal-edge segment: windows and records from LDS, filter, stores to DST (512-byte runs per row).
// All vertical edges of the picture before all horizontal ones (xevdm_deblock, src_main/xevdm.c:2048-2103) holds per sample: a horizontal filter reads
// only its own window, which phase V of the same workgroup has completed.
// ---------------------------------------------------------------------------------------------------------
#define AF_LS 264                 // LDS luma row stride in samples: 256 + 8 (rows 4 apart land 16 banks apart; rows stay 16-byte aligned)
#define AF_CS 136                 // chroma: 128 + 8
template <int SR>                 // SCU rows of the tile: 4 = 128 threads, 256 x 16 samples (8 = 256 threads, 256 x 32 samples measured the same)
__global__ __launch_bounds__(32 * SR) void k_addb_fused(const AddbArgs a, const int16_t *__restrict__ sy_, const int16_t *__restrict__ su_,
                                                    const int16_t *__restrict__ sv_, int16_t *__restrict__ dy_, int16_t *__restrict__ du_,
                                                    int16_t *__restrict__ dv_, int tiles_x)
{
    __shared__ uint8_t s_alpha[52], s_beta[52], s_clip[52 * 5], s_pic[XGPU_MAX_REFS * 2];
    __shared__ int8_t s_cqp[2 * 96];
    constexpr int NT = 32 * SR;
    __shared__ __attribute__((aligned(16))) int16_t s_y[4 * SR * AF_LS];
    __shared__ __attribute__((aligned(16))) int16_t s_c[2][2 * SR * AF_CS];
    __shared__ uint4 s_map[SR][64];
    const int t = threadIdx.x;
    for (int i = t; i < 52; i += NT) { s_alpha[i] = k_alpha[i]; s_beta[i] = k_beta[i]; }
    for (int i = t; i < 260; i += NT) s_clip[i] = ((const uint8_t *)k_clip)[i];
    for (int i = t; i < XGPU_MAX_REFS * 2; i += NT) s_pic[i] = a.pic_id[i];
    for (int i = t; i < 192; i += NT) s_cqp[i] = a.chroma_qp[i];
    const int tx = blockIdx.x % tiles_x, ty = blockIdx.x / tiles_x;
    const int n_ex = (a.w_scu >> 1) + 1, n_ey = (a.h_scu >> 1) + 1;     // grid lines incl. the picture borders (which only copy their inner half)
#define PK2(lo, hi) ((uint32_t)(uint16_t)(lo) | ((uint32_t)(uint16_t)(hi) << 16))

    // ---------------------------------------------------------------- phase V: vertical edges
    {
        const int wx = t & 31, sr = t >> 5;                  // window along x, SCU row of the tile
        const int ex = (tx << 5) + wx;                       // grid line x = 8 * ex
        const int srow = ty * SR - 1 + sr;                   // SCU row in the picture
        const int sxq = ex << 1;                             // the Q-side SCU column
        const bool ok = ex < n_ex && srow >= 0 && srow < a.h_scu;
        const bool has_p = ok && ex > 0, has_q = ok && sxq < a.w_scu;
        const uint4 *maps = (const uint4 *)a.maps;
        uint4 rq = make_uint4(0, 0, 0, 0), rp = rq;
        int L[4][8], Cc[2][2][4];
        if (ok) {
            const int kq = srow * a.w_scu + sxq;
            if (has_q) rq = maps[kq];
            if (has_p) rp = maps[kq - 1];
            const int x = sxq << 2, y = srow << 2, cy = srow << 1;
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const U32x4a8 v = *(const U32x4a8 *)(sy_ + (y + r) * a.s_l + x - 4);
                L[r][0] = (int16_t)(v.a & 0xFFFF); L[r][1] = (int16_t)(v.a >> 16); L[r][2] = (int16_t)(v.b & 0xFFFF); L[r][3] = (int16_t)(v.b >> 16);
                L[r][4] = (int16_t)(v.c & 0xFFFF); L[r][5] = (int16_t)(v.c >> 16); L[r][6] = (int16_t)(v.d & 0xFFFF); L[r][7] = (int16_t)(v.d >> 16);
            }
#pragma unroll
            for (int pl = 0; pl < 2; pl++)
#pragma unroll
                for (int r = 0; r < 2; r++) {
                    const U32x2a4 v = *(const U32x2a4 *)((pl ? sv_ : su_) + (cy + r) * a.s_c + (x >> 1) - 2);
                    Cc[pl][r][0] = (int16_t)(v.a & 0xFFFF); Cc[pl][r][1] = (int16_t)(v.a >> 16);
                    Cc[pl][r][2] = (int16_t)(v.b & 0xFFFF); Cc[pl][r][3] = (int16_t)(v.b >> 16);
                }
        } else {
#pragma unroll
            for (int r = 0; r < 4; r++)
#pragma unroll
                for (int k = 0; k < 8; k++) L[r][k] = 0;
#pragma unroll
            for (int pl = 0; pl < 2; pl++)
#pragma unroll
                for (int r = 0; r < 2; r++)
#pragma unroll
                    for (int k = 0; k < 4; k++) Cc[pl][r][k] = 0;
        }
        __syncthreads();                                     // the tables (the loads above are in flight across it)
        if (has_p && has_q) addb_edge<0>(a, rq, rp, sxq, L, Cc, s_alpha, s_beta, s_clip, s_cqp, s_pic);
        s_map[sr][2 * wx] = rp; s_map[sr][2 * wx + 1] = rq;
#pragma unroll
        for (int r = 0; r < 4; r++)
            *(uint4 *)(s_y + (4 * sr + r) * AF_LS + 8 * wx) = make_uint4(PK2(L[r][0], L[r][1]), PK2(L[r][2], L[r][3]), PK2(L[r][4], L[r][5]), PK2(L[r][6], L[r][7]));
#pragma unroll
        for (int pl = 0; pl < 2; pl++)
#pragma unroll
            for (int r = 0; r < 2; r++)
                *(uint2 *)(s_c[pl] + (2 * sr + r) * AF_CS + 4 * wx) = make_uint2(PK2(Cc[pl][r][0], Cc[pl][r][1]), PK2(Cc[pl][r][2], Cc[pl][r][3]));
    }
    __syncthreads();

    // ---------------------------------------------------------------- phase H: horizontal edges
    {
        const int sx = t & 63, g = t >> 6;                   // SCU column of the tile, grid line of the tile
        const int scol = (tx << 6) - 1 + sx;                 // SCU column in the picture
        const int ey = ty * (SR / 2) + g;                    // grid line y = 8 * ey
        if (scol < 0 || scol >= a.w_scu || ey >= n_ey) return;
        const int syq = ey << 1;                             // the Q-side SCU row
        const bool has_p = ey > 0, has_q = syq < a.h_scu;
        const uint4 rq = s_map[2 * g + 1][sx], rp = s_map[2 * g][sx];
        int L[4][8], Cc[2][2][4];
#pragma unroll
        for (int r = 0; r < 8; r++) {
            const uint2 v = *(const uint2 *)(s_y + (8 * g + r) * AF_LS + 4 * sx);
            L[0][r] = (int16_t)(v.x & 0xFFFF); L[1][r] = (int16_t)(v.x >> 16); L[2][r] = (int16_t)(v.y & 0xFFFF); L[3][r] = (int16_t)(v.y >> 16);
        }
#pragma unroll
        for (int pl = 0; pl < 2; pl++)
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const uint32_t v = *(const uint32_t *)(s_c[pl] + (4 * g + r) * AF_CS + 2 * sx);
                Cc[pl][0][r] = (int16_t)(v & 0xFFFF); Cc[pl][1][r] = (int16_t)(v >> 16);
            }
        if (has_p && has_q) addb_edge<1>(a, rq, rp, syq, L, Cc, s_alpha, s_beta, s_clip, s_cqp, s_pic);
        const int x = scol << 2, y = syq << 2, cx = scol << 1, cy = syq << 1;
#pragma unroll
        for (int r = 0; r < 8; r++) {
            if (r < 4 ? !has_p : !has_q) continue;
            *(uint2 *)(dy_ + (y - 4 + r) * a.s_l + x) = make_uint2(PK2(L[0][r], L[1][r]), PK2(L[2][r], L[3][r]));
        }
#pragma unroll
        for (int pl = 0; pl < 2; pl++)
#pragma unroll
            for (int r = 0; r < 4; r++) {
                if (r < 2 ? !has_p : !has_q) continue;
                *(uint32_t *)((pl ? dv_ : du_) + (cy - 2 + r) * a.s_c + cx) = PK2(Cc[pl][0][r], Cc[pl][1][r]);
            }
    }
#undef PK2
}

void launch_addb_fused(xgpu_ctx *c, const AddbArgs &a, const DevPic &src, const DevPic &dst)
{
    const int n_ex = (a.w_scu >> 1) + 1, n_ey = (a.h_scu >> 1) + 1;
    // tile height: 4 SCU rows (128 threads, 17 KB of LDS: nine workgroups per CU in different phases of load - filter - store).  Measured against 8 rows (256
    // threads, 34 KB): 64.9 / 65.0 us at 8K, 25.0 / 26.4 us at 4K - the kernel is bound by neither VALU issue (an edge-compacted variant with 12.7 M instead of
    // 21.7 M VALU instructions took the same 66 us) nor workgroup granularity; 233 MB in 65 us is 75 % of what a plain copy kernel reaches on this part.
    const int tiles_x = (n_ex + 31) >> 5;
    hipLaunchKernelGGL(k_addb_fused<4>, dim3(tiles_x * ((n_ey + 1) >> 1)), dim3(128), 0, c->stream, a, src.y, src.u, src.v, dst.y, dst.u, dst.v, tiles_x);
}

