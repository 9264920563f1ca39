// addb_filter.h - the arithmetic of the ADDB deblocking filter shared by k_addb.hip (the filter alone) and k_alf.hip (the filter fused in front of ALF): the
// specification's tables, boundary strength (get_bs, src_main/xevdm_df.c:361-513), the luma / chroma line filters (:584-781) and the decisions of one 4-sample
// edge segment (deblock_addb_cu_hor :893-944 / deblock_addb_cu_ver_yuv :947-1034).
#pragma once
#include "xgpu_internal.h"

struct __attribute__((packed, aligned(8))) U32x4a8 { uint32_t a, b, c, d; };
struct __attribute__((packed, aligned(4))) U32x2a4 { uint32_t a, b; };

__device__ __forceinline__ int clip3a(int lo, int hi, int v) { return min(max(v, lo), hi); }

// ALPHA_TABLE / BETA_TABLE / CLIP_TAB (src_main/xevdm_tbl.c:377-379) - tables of the EVC specification
#define ADDB_ALPHA_INIT { 0,0,0,0,0,0,0,0,0,0,0,0, 0,0,0,0,4,4,5,6, 7,8,9,10,12,13,15,17, 20,22,25,28,32,36,40,45, \
    50,56,63,71,80,90,101,113, 127,144,162,182,203,226,255,255 }
#define ADDB_BETA_INIT { 0,0,0,0,0,0,0,0,0,0,0,0, 0,0,0,0,2,2,2,3, 3,3,3,4,4,4,6,6, 7,7,8,8,9,9,10,10, \
    11,11,12,12,13,13,14,14, 15,15,16,16,17,17,18,18 }
#define ADDB_CLIP_INIT { \
    {0,0,0,0,0},{0,0,0,0,0},{0,0,0,0,0},{0,0,0,0,0},{0,0,0,0,0},{0,0,0,0,0},{0,0,0,0,0},{0,0,0,0,0}, \
    {0,0,0,0,0},{0,0,0,0,0},{0,0,0,0,0},{0,0,0,0,0},{0,0,0,0,0},{0,0,0,0,0},{0,0,0,0,0},{0,0,0,0,0}, \
    {0,0,0,0,0},{0,0,0,1,1},{0,0,0,1,1},{0,0,0,1,1},{0,0,0,1,1},{0,0,1,1,1},{0,0,1,1,1},{0,1,1,1,1}, \
    {0,1,1,1,1},{0,1,1,1,1},{0,1,1,1,1},{0,1,1,2,2},{0,1,1,2,2},{0,1,1,2,2},{0,1,1,2,2},{0,1,2,3,3}, \
    {0,1,2,3,3},{0,2,2,3,3},{0,2,2,4,4},{0,2,3,4,4},{0,2,3,4,4},{0,3,3,5,5},{0,3,4,6,6},{0,3,4,6,6}, \
    {0,4,5,7,7},{0,4,5,8,8},{0,4,6,9,9},{0,5,7,10,10},{0,6,8,11,11},{0,6,8,13,13},{0,7,10,14,14},{0,8,11,16,16}, \
    {0,9,12,18,18},{0,10,13,20,20},{0,11,15,23,23},{0,13,17,25,25} }
static __constant__ uint8_t k_alpha[52] = ADDB_ALPHA_INIT;
static __constant__ uint8_t k_beta[52] = ADDB_BETA_INIT;
static __constant__ uint8_t k_clip[52][5] = ADDB_CLIP_INIT;
// the same tables on the host: xgpu_deblock lays them out, with the picture's reference identities and chroma QP mapping, as the block k_addb_alf copies into LDS
// dword by dword (AddbArgs.lds_tables; round 6: one load and one store per thread instead of six byte copies from five places)
static const uint8_t h_addb_alpha[52] = ADDB_ALPHA_INIT;
static const uint8_t h_addb_beta[52] = ADDB_BETA_INIT;
static const uint8_t h_addb_clip[52][5] = ADDB_CLIP_INIT;

// |a - b| < 4 in both components of two packed (x, y) vectors.  The reference subtracts in int (abs(a - b) of two s16 never wraps); here the packed difference
// SATURATES (v_pk_sub_i16 ... clamp), so a difference beyond s16 stays large instead of wrapping into (-4, 4), and the test is one mask: |d| < 4 <=> no bit above bit 1
__device__ __forceinline__ bool mv_same(uint32_t a, uint32_t b)
{
    uint32_t d, e, m;
    asm("v_pk_sub_i16 %0, %1, %2 clamp" : "=v"(d) : "v"(a), "v"(b));
    asm("v_pk_sub_i16 %0, %1, %2 clamp" : "=v"(e) : "v"(b), "v"(a));
    asm("v_pk_max_i16 %0, %1, %2" : "=v"(m) : "v"(d), "v"(e));
    return (m & 0xFFFCFFFCu) == 0;
}
// get_bs, xevdm_df.c:361-513.  q = record of the right/below SCU, p = left/above; cross_ctu: the edge lies on a CU boundary that is a CTU boundary.
// (Round 6: the vector comparisons on packed pairs - four v_pk_* per pair of vectors instead of two subtractions, two negations, two maxima and two compares per
//  component; the kernel this runs in, k_addb_alf, is bound by instruction issue and every one of its 324 edge segments per tile computes a strength.)
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
    // the vector of an unused list counts as (0, 0) (xevdm_df.c:437-452)
    const uint32_t qm0 = q0 >= 0 ? q.z : 0u, qm1 = q1 >= 0 ? q.w : 0u, pm0 = p0 >= 0 ? p.z : 0u, pm1 = p1 >= 0 ? p.w : 0u;
    const bool straight = Q0 == P0 && Q1 == P1, crossed = Q0 == P1 && Q1 == P0;
    if (!straight && !crossed) return 1;
    const bool s_ok = mv_same(qm0, pm0) && mv_same(qm1, pm1), c_ok = mv_same(qm0, pm1) && mv_same(qm1, pm0);
    if (Q0 == Q1) return (s_ok && c_ok) ? 0 : 1;
    if (straight) return s_ok ? 0 : 1;
    return c_ok ? 0 : 1;
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
// decision half: the boundary strength of the segment, 0 = nothing to filter (not a CU / transform boundary, a tile border the PPS keeps, or strength 0)
template <int DIR>
__device__ __forceinline__ int addb_edge_strength(const AddbArgs &a, const TileMask &no_filter, const uint4 rq, const uint4 rp, int eq, const uint8_t *s_pic)
{
    const uint32_t eflag = DIR == 0 ? SCU_EDGE_L : SCU_EDGE_T;
    // an edge on a tile border stays as it is unless the PPS filters across tiles (no_boundary, src_main/xevdm_df.c:877, 1088, 1106)
    const int ctu_sh = a.log2_ctu - 2;
    const bool tile_edge = (eq & ((1 << ctu_sh) - 1)) == 0 && (DIR == 0 ? no_filter.col_start((eq >> ctu_sh) & 255) : no_filter.row_start((eq >> ctu_sh) & 255));
    if (!(rq.x & eflag) || tile_edge) return 0;
    const int epos = eq << 2;
    const bool cross = (epos & ((1 << a.log2_ctu) - 1)) == 0;
    return addb_bs(rq, rp, cross, s_pic);
}

// ... with the direction per lane (k_alf.hip's interior tiles decide the vertical and the horizontal segments of a tile in one pass of all lanes)
__device__ __forceinline__ int addb_edge_strength_rt(const AddbArgs &a, const TileMask &no_filter, const uint4 rq, const uint4 rp, int eq, bool hor, const uint8_t *s_pic)
{
    const uint32_t eflag = hor ? SCU_EDGE_T : SCU_EDGE_L;
    const int ctu_sh = a.log2_ctu - 2, ctu_i = (eq >> ctu_sh) & 255;
    const uint32_t starts = hor ? no_filter.hb[ctu_i >> 5] : no_filter.vb[ctu_i >> 5];
    const bool tile_edge = (eq & ((1 << ctu_sh) - 1)) == 0 && ((starts >> (ctu_i & 31)) & 1);
    if (!(rq.x & eflag) || tile_edge) return 0;
    const bool cross = ((eq << 2) & ((1 << a.log2_ctu) - 1)) == 0;
    return addb_bs(rq, rp, cross, s_pic);
}

// filter half: bs > 0
template <int DIR>
__device__ __forceinline__ void addb_edge_filter(const AddbArgs &a, const uint4 rq, const uint4 rp, int bs, int L[4][8], int Cc[2][2][4],
                                                 const uint8_t *s_alpha, const uint8_t *s_beta, const uint8_t *s_clip, const int8_t *s_cqp)
{
    const uint32_t nflag = DIR == 0 ? SCU_NOCH_L : SCU_NOCH_T;      // a luma CU's edge inside the chroma block of a local dual tree: luma only (xevdm_df.c:916-920, 986-997)
    const int maxl = (1 << a.bd_l) - 1, maxc = (1 << a.bd_c) - 1;
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

template <int DIR>
__device__ __forceinline__ void addb_edge(const AddbArgs &a, const uint4 rq, const uint4 rp, int eq, int L[4][8], int Cc[2][2][4],
                                          const uint8_t *s_alpha, const uint8_t *s_beta, const uint8_t *s_clip, const int8_t *s_cqp, const uint8_t *s_pic)
{
    const int bs = addb_edge_strength<DIR>(a, a.no_filter, rq, rp, eq, s_pic);
    if (bs) addb_edge_filter<DIR>(a, rq, rp, bs, L, Cc, s_alpha, s_beta, s_clip, s_cqp);
}

// ---------------------------------------------------------------------------------------------------------------------------------------------------
// Two lines of an edge segment at once (round 4): the same filters on PACKED pairs - the low half of every dword belongs to one line of the segment, the high half
// to the next - with v_pk_* arithmetic; every condition is a 0 / 0xFFFF mask per half and every "if" a v_bfi select between the variants (a wave whose lanes take
// different branches executes all of them in the scalar form too).  k_addb_alf is bound by VALU issue and the scalar line filters were a third of its
// instructions.  Exact for bit depths up to 10: the largest intermediate, 3 (p2 + p0 + q0) - 8 p1 - q1, stays inside s16 (9 x 1023); deeper pictures take the
// scalar form.  s[] as above: p3 p2 p1 p0 q0 q1 q2 q3.
typedef short pk_s __attribute__((ext_vector_type(2)));
__device__ __forceinline__ pk_s pks(uint32_t x) { return __builtin_bit_cast(pk_s, x); }
__device__ __forceinline__ uint32_t pku(pk_s x) { return __builtin_bit_cast(uint32_t, x); }
__device__ __forceinline__ uint32_t pk_rep(int v) { return ((uint32_t)v & 0xFFFFu) * 0x00010001u; }
__device__ __forceinline__ pk_s pk_abs(pk_s x) { return __builtin_elementwise_max(x, (pk_s){0, 0} - x); }
__device__ __forceinline__ uint32_t pk_lt(pk_s x, pk_s thr) { return pku((x - thr) >> (pk_s){15, 15}); }      // 0xFFFF where x < thr (no overflow at these ranges)
__device__ __forceinline__ uint32_t pk_sel(uint32_t m, uint32_t a, uint32_t b) { return (a & m) | (b & ~m); }  // -> v_bfi_b32
__device__ __forceinline__ pk_s pk_clip(pk_s lo, pk_s hi, pk_s v) { return __builtin_elementwise_min(__builtin_elementwise_max(v, lo), hi); }

__device__ __forceinline__ void addb_line_luma_pk(uint32_t s[8], int bs, int alpha, int beta, int c1, int bd, int maxv)
{
    const pk_s p3 = pks(s[0]), p2 = pks(s[1]), p1 = pks(s[2]), p0 = pks(s[3]), q0 = pks(s[4]), q1 = pks(s[5]), q2 = pks(s[6]), q3 = pks(s[7]);
    const pk_s A = pks(pk_rep(alpha)), B = pks(pk_rep(beta)), Z = { 0, 0 }, MX = pks(pk_rep(maxv));
    const pk_s dpq = pk_abs(p0 - q0);
    const uint32_t on = pk_lt(dpq, A) & pk_lt(pk_abs(p1 - p0), B) & pk_lt(pk_abs(q1 - q0), B);      // (bs != 0: the caller's)
    if (!on) return;
    const uint32_t ap = pk_lt(pk_abs(p0 - p2), B), aq = pk_lt(pk_abs(q0 - q2), B);
    pk_s po0, po1 = p1, po2 = p2, qo0, qo1 = q1, qo2 = q2;
    if (bs == 4) {
        const uint32_t strong = pk_lt(dpq, pks(pk_rep((alpha >> 2) + 2)));
        const pk_s two = { 2, 2 }, four = { 4, 4 }, sh2 = { 2, 2 }, sh3 = { 3, 3 };
        const pk_s mid = p0 + q0, pw0 = (2 * p1 + p0 + q1 + two) >> sh2, qw0 = (2 * q1 + q0 + p1 + two) >> sh2;
        const pk_s ps0 = (p2 + 2 * (p1 + mid) + q1 + four) >> sh3, ps1 = (p2 + p1 + mid + two) >> sh2, ps2 = (2 * p3 + 3 * p2 + p1 + mid + four) >> sh3;
        const pk_s qs0 = (q2 + 2 * (q1 + mid) + p1 + four) >> sh3, qs1 = (q2 + q1 + mid + two) >> sh2, qs2 = (2 * q3 + 3 * q2 + q1 + mid + four) >> sh3;
        const uint32_t mp = ap & strong, mq = aq & strong;
        po0 = pks(pk_sel(mp, pku(ps0), pku(pw0))); po1 = pks(pk_sel(mp, pku(ps1), pku(p1))); po2 = pks(pk_sel(mp, pku(ps2), pku(p2)));
        qo0 = pks(pk_sel(mq, pku(qs0), pku(qw0))); qo1 = pks(pk_sel(mq, pku(qs1), pku(q1))); qo2 = pks(pk_sel(mq, pku(qs2), pku(q2)));
    } else {
        const int sh = max(0, bd - 9);
        const pk_s C1 = pks(pk_rep(c1));
        const pk_s c0 = pks((pku(C1) + ((((ap & 0x00010001u) + (aq & 0x00010001u))) << sh)) & 0x00FF00FFu);          // u8 c0, xevdm_df.c:650
        const pk_s d0 = pk_clip(Z - c0, c0, (4 * (q0 - p0) + p1 - q1 + (pk_s){4, 4}) >> (pk_s){3, 3});
        po0 = pk_clip(Z, MX, p0 + d0);
        qo0 = pk_clip(Z, MX, q0 - d0);
        const pk_s dp = pk_clip(Z - C1, C1, ((p2 + p0 + q0) * (pk_s){3, 3} - 8 * p1 - q1) >> (pk_s){4, 4});
        const pk_s dq = pk_clip(Z - C1, C1, ((q2 + q0 + p0) * (pk_s){3, 3} - 8 * q1 - p1) >> (pk_s){4, 4});
        po1 = pks(pk_sel(ap, pku(p1 + dp), pku(p1)));
        qo1 = pks(pk_sel(aq, pku(q1 + dq), pku(q1)));
    }
    s[3] = pk_sel(on, pku(pk_clip(Z, MX, po0)), s[3]); s[2] = pk_sel(on, pku(pk_clip(Z, MX, po1)), s[2]); s[1] = pk_sel(on, pku(pk_clip(Z, MX, po2)), s[1]);
    s[4] = pk_sel(on, pku(pk_clip(Z, MX, qo0)), s[4]); s[5] = pk_sel(on, pku(pk_clip(Z, MX, qo1)), s[5]); s[6] = pk_sel(on, pku(pk_clip(Z, MX, qo2)), s[6]);
}
// chroma, two lines: s = p1 p0 q0 q1
__device__ __forceinline__ void addb_line_chroma_pk(uint32_t s[4], int bs, int alpha, int beta, int c0v, int maxv)
{
    const pk_s p1 = pks(s[0]), p0 = pks(s[1]), q0 = pks(s[2]), q1 = pks(s[3]);
    const pk_s A = pks(pk_rep(alpha)), B = pks(pk_rep(beta)), Z = { 0, 0 }, MX = pks(pk_rep(maxv));
    const uint32_t on = pk_lt(pk_abs(p0 - q0), A) & pk_lt(pk_abs(p1 - p0), B) & pk_lt(pk_abs(q1 - q0), B);
    if (!on) return;
    pk_s po, qo;
    if (bs == 4) {
        po = pk_clip(Z, MX, (2 * p1 + p0 + q1 + (pk_s){2, 2}) >> (pk_s){2, 2});
        qo = pk_clip(Z, MX, (2 * q1 + q0 + p1 + (pk_s){2, 2}) >> (pk_s){2, 2});
    } else {
        const pk_s C0 = pks(pk_rep(c0v));
        const pk_s d0 = pk_clip(Z - C0, C0, (4 * (q0 - p0) + p1 - q1 + (pk_s){4, 4}) >> (pk_s){3, 3});
        po = pk_clip(Z, MX, p0 + d0);
        qo = pk_clip(Z, MX, q0 - d0);
    }
    s[1] = pk_sel(on, pku(po), s[1]); s[2] = pk_sel(on, pku(qo), s[2]);
}
// The parameters of a segment's luma filter and of one chroma plane's filter (the front halves of addb_edge_filter above), for callers that give the line pairs and
// the planes of a segment to different waves (k_alf.hip, round 6).  chroma: false = a luma-only edge (local dual tree), nothing to filter in the planes
__device__ __forceinline__ void addb_luma_params(const AddbArgs &a, uint32_t rqx, uint32_t rpx, int bs, const uint8_t *s_alpha, const uint8_t *s_beta, const uint8_t *s_clip,
                                                 int &alpha, int &beta, int &c1)
{
    const int qp = (((rqx >> 16) & 0x7F) + ((rpx >> 16) & 0x7F) + 1) >> 1, scale = a.bd_l - 8;
    const int ia = addb_index(qp, a.alpha_off), ib = addb_index(qp, a.beta_off);
    alpha = s_alpha[ia] << scale; beta = (s_beta[ib] << scale) & 0xFF;
    c1 = (s_clip[ia * 5 + bs] << max(0, a.bd_l - 9)) & 0xFF;
}
template <int DIR>
__device__ __forceinline__ bool addb_chroma_params(const AddbArgs &a, uint32_t rqx, uint32_t rpx, int bs, int pl, const uint8_t *s_alpha, const uint8_t *s_beta, const uint8_t *s_clip,
                                                   const int8_t *s_cqp, int &alpha, int &beta, int &c0)
{
    if (rqx & (DIR == 0 ? SCU_NOCH_L : SCU_NOCH_T)) return false;
    const int qp = (((rqx >> 16) & 0x7F) + ((rpx >> 16) & 0x7F) + 1) >> 1, scale = a.bd_l - 8, boff = 6 * (a.bd_c - 8);
    const int q = clip3a(-boff, 57, qp + (pl ? a.qp_v_off : a.qp_u_off));
    const int qc = s_cqp[pl * 96 + q + boff];
    const int ia = addb_index(qc, a.alpha_off), ib = addb_index(qc, a.beta_off);
    alpha = s_alpha[ia] << scale; beta = (s_beta[ib] << scale) & 0xFF;      // luma depth scales chroma too (:926-927)
    c0 = ((s_clip[ia * 5 + bs] + 1) << max(0, a.bd_c - 9)) & 0xFF;
    return true;
}

// filter half of an edge segment on packed line pairs: LP[0] = lines 0 / 1, LP[1] = lines 2 / 3 of the luma segment (8 positions across the edge each),
// CP[plane] = the two chroma lines (4 positions).  bs > 0, bit depths <= 10.
template <int DIR>
__device__ __forceinline__ void addb_edge_filter_pk(const AddbArgs &a, const uint4 rq, const uint4 rp, int bs, uint32_t LP[2][8], uint32_t CP[2][4],
                                                    const uint8_t *s_alpha, const uint8_t *s_beta, const uint8_t *s_clip, const int8_t *s_cqp)
{
    const uint32_t nflag = DIR == 0 ? SCU_NOCH_L : SCU_NOCH_T;
    const int maxl = (1 << a.bd_l) - 1, maxc = (1 << a.bd_c) - 1;
    const int qp = (((rq.x >> 16) & 0x7F) + ((rp.x >> 16) & 0x7F) + 1) >> 1;
    const int scale = a.bd_l - 8;
    {
        const int ia = addb_index(qp, a.alpha_off), ib = addb_index(qp, a.beta_off);
        const int alpha = s_alpha[ia] << scale, beta = (s_beta[ib] << scale) & 0xFF;
        const int c1 = (s_clip[ia * 5 + bs] << max(0, a.bd_l - 9)) & 0xFF;
        addb_line_luma_pk(LP[0], bs, alpha, beta, c1, a.bd_l, maxl);
        addb_line_luma_pk(LP[1], bs, alpha, beta, c1, a.bd_l, maxl);
    }
    const int boff = 6 * (a.bd_c - 8);
    if (!(rq.x & nflag))
#pragma unroll
    for (int pl = 0; pl < 2; pl++) {
        const int q = clip3a(-boff, 57, qp + (pl ? a.qp_v_off : a.qp_u_off));
        const int qc = s_cqp[pl * 96 + q + boff];
        const int ia = addb_index(qc, a.alpha_off), ib = addb_index(qc, a.beta_off);
        const int alpha = s_alpha[ia] << scale, beta = (s_beta[ib] << scale) & 0xFF;
        const int c0 = ((s_clip[ia * 5 + bs] + 1) << max(0, a.bd_c - 9)) & 0xFF;
        addb_line_chroma_pk(CP[pl], bs, alpha, beta, c0, maxc);
    }
}
