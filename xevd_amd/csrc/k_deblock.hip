// k_deblock.hip - baseline EVC deblocking filter, vertical-edge pass and horizontal-edge pass.
//
// Replaces xevd_deblock -> deblock_tree -> xevd_deblock_cu_ver / _hor -> deblock_scu_* (src_base/xevd.c:1057-1243,
// src_base/xevd_df.c:34-546).  Semantics reproduced: every CU left/top edge inside the picture is filtered once
// in 4-sample segments; edge class from intra / luma cbf / reference + MV comparison (get_tbl_qp_to_st); QP of
// the right/below block; strength = xevd_tbl_df_st[class][qp] << (bd-8); all vertical edges of the picture
// before all horizontal edges; along one line of CHROMA samples edges that are 2 samples apart are applied in
// increasing coordinate order, edge k+1 reading the sample edge k wrote.
//
// MI355X mapping - order-free, out of place:
//   * each pass reads picture SRC and writes picture DST (recon -> scratch -> DPB picture); every sample is
//     written exactly once, so nothing depends on workgroup order and a pass moves 2 B in + 2 B out per sample,
//     exactly the reference's in-place traffic;
//   * one LANE per 4-sample edge segment (the left / top edge of an SCU, the grid line one past the picture included) owns what that
//     edge can change - luma [e-2, e+2), chroma [e/2-1, e/2+1) along the filtered axis; the windows tile the picture: no write
//     conflicts, each filter evaluated once (the first version gave a lane its SCU and made it evaluate both edges touching it);
//   * the chroma order dependence (edge k needs the C' sample of edge k-1 when that edge is active) is resolved
//     per lane by walking back to the head of the dependency chain and recomputing it forward in registers from
//     the ORIGINAL samples - chains are as long as a run of 4-wide CUs (typically 1-3), and every lane stays
//     independent, which is what lets the whole picture run as one flat grid;
//   * the per-edge decisions are pure lane-local integer tests on one 16-byte SCU record per side; the
//     strength tables (component x class x QP, built on the host from xevd_tbl_df_st and the chroma QP mapping)
//     are staged in LDS.
#include "xgpu_internal.h"

// The vertical pass resolves its chroma chains with the DPP control wave_shr:1, which exists on the 64-lane wave of GFX9 / CDNA only (gfx10+ has no wave_shr and runs
// 32-lane waves): this file is written for gfx950 and refuses to build device code for anything else instead of filtering wrongly there.
#if defined(__HIP_DEVICE_COMPILE__) && !defined(__GFX9__)
#error "k_deblock.hip: the chroma chain exchange uses DPP wave_shr:1 (GFX9 / CDNA, wave64) - build with --offload-arch=gfx950"
#endif

struct __attribute__((packed, aligned(4))) U32x4a4 { uint32_t a, b, c, d; };   // 16-byte load at 4-byte alignment
struct __attribute__((packed, aligned(4))) U32x2a4 { uint32_t a, b; };         // 8 bytes at 4-byte alignment
struct __attribute__((packed, aligned(2))) U32x1a2 { uint32_t a; };            // 4 bytes at 2-byte alignment (gfx950: unaligned-access mode)

__device__ __forceinline__ int clip3i(int lo, int hi, int v) { return min(max(v, lo), hi); }

// get_tbl_qp_to_st, xevd_df.c:34-94.  q = record of the right/below SCU, p = left/above.
__device__ __forceinline__ int edge_class(const uint4 q, const uint4 p)
{
    if (((q.x | p.x) >> 15) & 1) return 0;
    if (((q.x | p.x) >> 24) & 1) return 1;
    if (((q.x | p.x) >> 26) & 1) return 2;                 // IBC on either side: the Main library's copy of the filter (xevdm_df.c:52-55)
    const int q0 = (int8_t)(q.y & 0xFF), q1 = (int8_t)((q.y >> 8) & 0xFF), p0 = (int8_t)(p.y & 0xFF), p1 = (int8_t)((p.y >> 8) & 0xFF);
    const int qm[2][2] = { { q0 >= 0 ? (int16_t)(q.z & 0xFFFF) : 0, q0 >= 0 ? (int16_t)(q.z >> 16) : 0 },
                           { q1 >= 0 ? (int16_t)(q.w & 0xFFFF) : 0, q1 >= 0 ? (int16_t)(q.w >> 16) : 0 } };
    const int pm[2][2] = { { p0 >= 0 ? (int16_t)(p.z & 0xFFFF) : 0, p0 >= 0 ? (int16_t)(p.z >> 16) : 0 },
                           { p1 >= 0 ? (int16_t)(p.w & 0xFFFF) : 0, p1 >= 0 ? (int16_t)(p.w >> 16) : 0 } };
    if (q0 == p0 && q1 == p1)
        return (abs(qm[0][0] - pm[0][0]) >= 4 || abs(qm[0][1] - pm[0][1]) >= 4 || abs(qm[1][0] - pm[1][0]) >= 4 || abs(qm[1][1] - pm[1][1]) >= 4) ? 2 : 3;
    if (q0 == p1 && q1 == p0)
        return (abs(qm[0][0] - pm[1][0]) >= 4 || abs(qm[0][1] - pm[1][1]) >= 4 || abs(qm[1][0] - pm[0][0]) >= 4 || abs(qm[1][1] - pm[0][1]) >= 4) ? 2 : 3;
    return 2;
}

// one line of deblock_scu_* (xevd_df.c:96-135): all arithmetic in s16 like the reference, '/' toward zero.
struct Line4 { int A, B, C, D; };
__device__ __forceinline__ Line4 filt_luma(Line4 s, int st, int maxv)
{
    const int d = (int16_t)((s.A - (s.B << 2) + (s.C << 2) - s.D) / 8);
    const int ad = abs(d);
    const int t16 = max(0, (ad - st) << 1);
    int clip = max(0, ad - t16);
    const int d1 = d < 0 ? -clip : clip;
    clip >>= 1;
    const int d2 = clip3i(-clip, clip, (s.A - s.D) / 4);
    Line4 o;
    o.A = clip3i(0, maxv, s.A - d2);
    o.B = clip3i(0, maxv, s.B + d1);
    o.C = clip3i(0, maxv, s.C - d1);
    o.D = clip3i(0, maxv, s.D + d2);
    return o;
}
// chroma: only B and C move (xevd_df.c:137-195)
__device__ __forceinline__ void filt_chroma(int A, int B, int C, int D, int st, int maxv, int &Bo, int &Co)
{
    const int d = (int16_t)((A - (B << 2) + (C << 2) - D) / 8);
    const int ad = abs(d);
    const int t16 = max(0, (ad - st) << 1);
    const int clip = max(0, ad - t16);
    const int d1 = d < 0 ? -clip : clip;
    Bo = clip3i(0, maxv, B + d1);
    Co = clip3i(0, maxv, C - d1);
}

// DIR 0: vertical edges (filter along x), DIR 1: horizontal edges (filter along y).
// One lane per 4-sample edge segment - the left / top edge of SCU (sx, sy), grid lines one past the picture included.  The lane owns what its
// edge can change: luma [e-2, e+2) and chroma [e/2-1, e/2+1) along the filtered axis, 4 (2) lines across; these windows tile the picture, so
// every sample is written exactly once, each filter is evaluated once, and the window is a single 8-byte (4-byte) load per line.
// ORD: the picture has CUs decoded after their right-hand neighbours (sps_suco_flag) - vertical chroma edges in the order the reference reaches them
template <int DIR, bool ORD>
__global__ __launch_bounds__(256) void k_dbk(const DbkArgs a, const int16_t *__restrict__ sy_, const int16_t *__restrict__ su_,
                                             const int16_t *__restrict__ sv_, int16_t *__restrict__ dy_, int16_t *__restrict__ du_,
                                             int16_t *__restrict__ dv_)
{
    __shared__ uint8_t s_st[3 * 4 * 64];
    for (int i = threadIdx.x; i < 3 * 4 * 64 / 4; i += 256) ((uint32_t *)s_st)[i] = ((const uint32_t *)a.st)[i];
    __syncthreads();

    // vertical edges: a wave = 64 neighbouring segments of one SCU row (512-byte runs of a picture row), 4 such rows per workgroup;
    // horizontal edges keep a 16 x 16 lane tile (measured: the wide mapping is slower there)
    const int n_ex = DIR == 0 ? a.w_scu + 1 : a.w_scu, n_ey = DIR == 0 ? a.h_scu : a.h_scu + 1;
    const int LW = DIR == 0 ? 6 : 4, LH = 8 - LW;
    const int tiles_x = (n_ex + (1 << LW) - 1) >> LW;
    const int tx = blockIdx.x % tiles_x, ty = blockIdx.x / tiles_x;
    const int sx = (tx << LW) + (threadIdx.x & ((1 << LW) - 1)), sy = (ty << LH) + (threadIdx.x >> LW);
    const bool valid = sx < n_ex && sy < n_ey;               // (no early exit: the chroma chains are resolved between the threads of the workgroup, with barriers)
    const int step = DIR == 0 ? 1 : a.w_scu;                 // SCU-map step along the filtering axis
    const int pos = DIR == 0 ? sx : sy, npos = DIR == 0 ? a.w_scu : a.h_scu;
    const uint32_t eflag = DIR == 0 ? SCU_EDGE_L : SCU_EDGE_T;
    const uint32_t nflag = DIR == 0 ? SCU_NOCH_L : SCU_NOCH_T;      // a luma CU's edge inside the chroma block of a local dual tree: luma only (xevdm_df.c:155-160)
    const uint4 *maps = (const uint4 *)a.maps;
    const bool has_p = valid && pos > 0, has_q = valid && pos < npos, in_range = has_p && has_q;
    const int k0 = in_range ? sy * a.w_scu + sx : 0;
    const int maxl = (1 << a.bd_l) - 1, maxc = (1 << a.bd_c) - 1;

    // ---- all loads first: the two SCU records (+ the one before: first link of a chroma chain), the luma and chroma windows ----
    const uint4 rq = maps[k0], rp = maps[in_range ? k0 - step : 0], rpp = maps[in_range && pos > 1 ? k0 - 2 * step : 0];
    const int x = valid ? sx << 2 : 0, y = valid ? sy << 2 : 0, cx = x >> 1, cy = y >> 1;      // (threads outside the edge grid read at the origin and store nothing)
    int L[4][4];           // luma [line][A B C D]
    int Cw[2][2][4];       // chroma [plane][line][A B C D]
    if (DIR == 0) {
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const U32x2a4 q = *(const U32x2a4 *)(sy_ + (y + r) * a.s_l + x - 2);
            const uint32_t d0 = q.a, d1 = q.b;
            L[r][0] = (int16_t)(d0 & 0xFFFF); L[r][1] = (int16_t)(d0 >> 16); L[r][2] = (int16_t)(d1 & 0xFFFF); L[r][3] = (int16_t)(d1 >> 16);
        }
#pragma unroll
        for (int pl = 0; pl < 2; pl++)
#pragma unroll
            for (int ln = 0; ln < 2; ln++) {
                const U32x2a4 q = *(const U32x2a4 *)((pl ? sv_ : su_) + (cy + ln) * a.s_c + cx - 2);
                const uint32_t d0 = q.a, d1 = q.b;
                Cw[pl][ln][0] = (int16_t)(d0 & 0xFFFF); Cw[pl][ln][1] = (int16_t)(d0 >> 16); Cw[pl][ln][2] = (int16_t)(d1 & 0xFFFF); Cw[pl][ln][3] = (int16_t)(d1 >> 16);
            }
    } else {
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const uint2 v = *(const uint2 *)(sy_ + (y - 2 + r) * a.s_l + x);
            L[0][r] = (int16_t)(v.x & 0xFFFF); L[1][r] = (int16_t)(v.x >> 16); L[2][r] = (int16_t)(v.y & 0xFFFF); L[3][r] = (int16_t)(v.y >> 16);
        }
#pragma unroll
        for (int pl = 0; pl < 2; pl++)
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const uint32_t v = *(const uint32_t *)((pl ? sv_ : su_) + (cy - 2 + r) * a.s_c + cx);
                Cw[pl][0][r] = (int16_t)(v & 0xFFFF); Cw[pl][1][r] = (int16_t)(v >> 16);
            }
    }

    int st[3] = {0, 0, 0};
    // an edge on a tile border stays as it is unless the PPS filters across tiles (no_boundary, src_main/xevdm_df.c:142, 233, 274)
    auto on_tile_border = [&](int e) -> bool {
        return (e & ((1 << a.ctu_sh) - 1)) == 0 && (DIR == 0 ? a.no_filter.col_start((e >> a.ctu_sh) & 255) : a.no_filter.row_start((e >> a.ctu_sh) & 255));
    };
    if (in_range && (rq.x & eflag) && !on_tile_border(pos)) {
        const int cls = edge_class(rq, rp), qp = (rq.x >> 16) & 0x7F;
#pragma unroll
        for (int c = 0; c < 3; c++) st[c] = s_st[(c * 4 + cls) * 64 + (qp & 63)];
        if (rq.x & nflag) st[1] = st[2] = 0;
    }

    // ------------------------------------------------ luma -----------------------------------------------
    if (st[0]) {
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const Line4 o = filt_luma({ L[r][0], L[r][1], L[r][2], L[r][3] }, st[0], maxl);
            L[r][0] = o.A; L[r][1] = o.B; L[r][2] = o.C; L[r][3] = o.D;
        }
    }
#define PK2(lo, hi) ((uint32_t)(uint16_t)(lo) | ((uint32_t)(uint16_t)(hi) << 16))
    if (DIR == 0) {
#pragma unroll
        for (int r = 0; r < 4; r++) {
            int16_t *d = dy_ + (y + r) * a.s_l + x;
            if (in_range) { U32x2a4 w = { PK2(L[r][0], L[r][1]), PK2(L[r][2], L[r][3]) }; *(U32x2a4 *)(d - 2) = w; }
            else if (has_p) *(uint32_t *)(d - 2) = PK2(L[r][0], L[r][1]);
            else if (has_q) *(uint32_t *)d = PK2(L[r][2], L[r][3]);
        }
    } else {
#pragma unroll
        for (int r = 0; r < 4; r++) {
            if (r < 2 ? !has_p : !has_q) continue;
            *(uint2 *)(dy_ + (y - 2 + r) * a.s_l + x) = make_uint2(PK2(L[0][r], L[1][r]), PK2(L[2][r], L[3][r]));
        }
    }

    // ------------------------------------------------ chroma ---------------------------------------------
    // Edges 2 chroma samples apart are applied in increasing coordinate order (xevd_df.c:238-289): this edge's A is the C' of the previous
    // edge when that one is active.  The lane walks back to the head of the chain of consecutive active edges and recomputes it forward
    // from the ORIGINAL samples (chains are as long as a run of 4-wide CUs; every lane stays independent).
    // (Round 3: a variant that loaded the records of three more edges with the first loads, walked over them in registers and fetched a chain's samples in one batch
    //  was bit-exact and SLOWER - 30.6 / 27.7 us instead of 28 / 25 at 1080p, 90 / 81 instead of 83 / 71 at 8K: chains are rare enough that the dependent loads below
    //  are seldom executed, and the 30 extra registers cost more than they saved.)
    // sps_suco_flag (Main library, xevdm_df.c:186-330): a vertical CU edge is reached with the LATER of its two CUs - as the left edge of the right CU or as the right
    // edge of the left one, a CU filtering its left edge first - so a split coded right to left applies its edges right to left.  before(k): edge k - 1 is applied
    // before edge k, from the places in decoding order (SCU_RANK) of the SCUs k - 2, k - 1, k; the CTUs themselves follow each other left to right.  An edge then
    // reads what its EARLIER neighbours wrote: the chain to the left (below), and with SUCO possibly one to the right.
    const int alongc = DIR == 0 ? 1 : a.s_c, acrossc = DIR == 0 ? a.s_c : 1;
    auto before = [&](uint32_t ma, uint32_t mb, uint32_t mc, int k) -> bool {
        if (DIR != 0 || !ORD) return true;
        const int cm = (1 << a.ctu_sh) - 1;
        int da = ((SCU_RANK_OF(ma) - SCU_RANK_OF(mb)) << 21) >> 21, dc = ((SCU_RANK_OF(mc) - SCU_RANK_OF(mb)) << 21) >> 21;
        if (((k - 1) & cm) == 0) da = -4096;                 // SCU k - 2 lies in the CTU before
        if ((k & cm) == 0) dc = 4096;                        // SCU k starts the next CTU
        return max(da, 0) <= max(dc, 0);
    };
    // Round 4, pictures in decoding order left to right (ORD false): the chains are resolved BETWEEN the threads that own the edges instead of by every thread for
    // itself.  Whether an edge reads the C' of the edge before it follows from the three records the thread has loaded anyway; an edge that does waits until the thread
    // before it (the lane to the left for vertical edges, 16 threads up for horizontal ones) has published its C' in LDS, filters, publishes its own - rounds = the longest
    // chain inside the tile, no memory access.  Only a thread in the tile's first column / row whose edge is linked walks back through memory like before, with one
    // walk for both planes and one load round per edge.  (Before: every thread of a chain of L edges did 6 L - 2 dependent memory round trips, and the longest chain of
    // the picture was 18 of a pass's 28 us at 1080p.)
    bool coop_done = false;
    if (!ORD) {
        __shared__ int s_cp[2][2][256];                      // C' of [plane][line], per thread
        __shared__ uint8_t s_ok[2][256];                     // [plane]: published
        const int tid = threadIdx.x, pred = DIR == 0 ? tid - 1 : tid - 16;
        const bool has_pred = DIR == 0 ? (tid & 63) != 0 : (tid >> 4) != 0;
        const bool pe = in_range && pos - 1 > 0 && (rp.x & eflag) && !(rp.x & nflag) && !on_tile_border(pos - 1);
        int sp[2] = { 0, 0 };                                // strength of the edge before this one
        if (pe) {
            const int cls = edge_class(rp, rpp), qp = (rp.x >> 16) & 0x7F;
            sp[0] = s_st[(1 * 4 + cls) * 64 + (qp & 63)]; sp[1] = s_st[(2 * 4 + cls) * 64 + (qp & 63)];
        }
        bool pend[2];
        int rc[2][2] = { { 0, 0 }, { 0, 0 } }, rok[2] = { 0, 0 };      // vertical edges: what the thread has published stays in registers (the lane to the right fetches it with a wave shift)
        auto finish = [&](int pl, int a0, int a1) {          // this edge's two lines with A' = a0 / a1; publishes the C' values
#pragma unroll
            for (int ln = 0; ln < 2; ln++) {
                int *w = Cw[pl][ln];
                if (st[1 + pl]) { int Bo, Co; filt_chroma(ln ? a1 : a0, w[1], w[2], w[3], st[1 + pl], maxc, Bo, Co); w[1] = Bo; w[2] = Co; }
                if (DIR == 0) rc[pl][ln] = w[2]; else s_cp[pl][ln][tid] = w[2];
            }
            if (DIR == 0) rok[pl] = 1; else s_ok[pl][tid] = 1;
        };
        const bool l0 = st[1] && sp[0], l1 = st[2] && sp[1];
        pend[0] = l0 && has_pred; pend[1] = l1 && has_pred;
        if (DIR != 0) { s_ok[0][tid] = 0; s_ok[1][tid] = 0; }
        if (!l0) finish(0, Cw[0][0][0], Cw[0][1][0]);
        if (!l1) finish(1, Cw[1][0][0], Cw[1][1][0]);
        if ((l0 || l1) && !has_pred) {
            // the chain enters the tile: walk back through memory - one walk for both planes, then forward with one round of loads per edge
            int head[2] = { l0 ? pos - 1 : pos, l1 ? pos - 1 : pos };
            {
                uint4 cur = rpp;                             // record of SCU e - 1 while edge e (between SCU e - 1 and e) is looked at
                bool on0 = l0, on1 = l1;
                for (int e = pos - 2; e > 0 && (on0 || on1); e--) {
                    // edge e is the left / top edge of SCU e: record maps[.. e], its neighbour maps[.. e - 1]
                    const uint4 re = cur, rn = maps[k0 - (pos - e + 1) * step];
                    if (!(re.x & eflag) || (re.x & nflag) || on_tile_border(e)) break;
                    const int cls = edge_class(re, rn), qp = (re.x >> 16) & 0x7F;
                    on0 = on0 && s_st[(1 * 4 + cls) * 64 + (qp & 63)] != 0; on1 = on1 && s_st[(2 * 4 + cls) * 64 + (qp & 63)] != 0;
                    if (on0) head[0] = e;
                    if (on1) head[1] = e;
                    cur = rn;
                }
            }
            int prevC[2][2] = { { 0, 0 }, { 0, 0 } };
            for (int e = min(head[0], head[1]); e < pos; e++) {
                const int rel = (e - pos) * 2, ke = k0 + (e - pos) * step;
                const uint4 q = maps[ke], pp = maps[ke - step];
                int wv[2][2][4];
#pragma unroll
                for (int pl = 0; pl < 2; pl++)
#pragma unroll
                    for (int ln = 0; ln < 2; ln++) {
                        const int16_t *p = (pl ? sv_ : su_) + cy * a.s_c + cx + ln * acrossc;
#pragma unroll
                        for (int i = 0; i < 4; i++) wv[pl][ln][i] = p[(rel - 2 + i) * alongc];
                    }
                const int cls = edge_class(q, pp), qp = (q.x >> 16) & 0x7F;
#pragma unroll
                for (int pl = 0; pl < 2; pl++) {
                    if (e < head[pl]) continue;
                    const int ste = s_st[((1 + pl) * 4 + cls) * 64 + (qp & 63)];
#pragma unroll
                    for (int ln = 0; ln < 2; ln++) {
                        int Bo, Co;
                        filt_chroma(e == head[pl] ? wv[pl][ln][0] : prevC[pl][ln], wv[pl][ln][1], wv[pl][ln][2], wv[pl][ln][3], ste, maxc, Bo, Co);
                        prevC[pl][ln] = Co;
                    }
                }
            }
            if (l0) finish(0, prevC[0][0], prevC[0][1]);
            if (l1) finish(1, prevC[1][0], prevC[1][1]);
        }
        // A round: every waiting thread looks at the flag and the C' values of the thread before it; who finds them filters and publishes its own.  Vertical edges: a chain
        // stays inside its row = one wave - the values stay in registers and move one lane to the right per round (DPP wave_shr:1; through LDS a round was two LDS round
        // trips, and a run of 4x4 CUs is 16 - 32 rounds).  Horizontal edges: a chain crosses the workgroup's waves - LDS, a barrier before the looks, one between looks and
        // publications (one barrier with fenced flag-then-value reads measured slower: 20.3 against 17.5 us).
        for (;;) {
            const bool waiting = pend[0] || pend[1];
            if (DIR == 0) { if (__ballot(waiting) == 0) break; }
            else if (!__syncthreads_or(waiting)) break;
            int ok[2] = { 0, 0 }, c[2][2] = { { 0, 0 }, { 0, 0 } };
            if (DIR == 0) {
#pragma unroll
                for (int pl = 0; pl < 2; pl++) {
                    ok[pl] = __builtin_amdgcn_update_dpp(0, rok[pl], 0x138, 0xF, 0xF, false);          // wave_shr:1 - lane i receives lane i - 1's value
                    c[pl][0] = __builtin_amdgcn_update_dpp(0, rc[pl][0], 0x138, 0xF, 0xF, false);
                    c[pl][1] = __builtin_amdgcn_update_dpp(0, rc[pl][1], 0x138, 0xF, 0xF, false);
                }
            } else {
#pragma unroll
                for (int pl = 0; pl < 2; pl++)
                    if (pend[pl]) { ok[pl] = s_ok[pl][pred]; c[pl][0] = s_cp[pl][0][pred]; c[pl][1] = s_cp[pl][1][pred]; }
                __syncthreads();                             // everybody has looked before anybody publishes
            }
#pragma unroll
            for (int pl = 0; pl < 2; pl++)
                if (pend[pl] && ok[pl]) { finish(pl, c[pl][0], c[pl][1]); pend[pl] = false; }
        }
        coop_done = true;
    }
#pragma unroll
    for (int pl = 0; pl < 2; pl++) {
        const int stt = coop_done ? 0 : st[1 + pl];          // (resolved above: only the stores are left)
        int head = pos, tail = pos;
        if (stt) {
            int kk = k0;
            uint4 cur = rp, nxt = rq;                        // records of SCU head-1, head
            while (head - 1 > 0) {
                if (!(cur.x & eflag) || (cur.x & nflag) || on_tile_border(head - 1)) break;
                const uint4 prv = kk == k0 ? rpp : maps[kk - 2 * step];
                const int cls = edge_class(cur, prv), qp = (cur.x >> 16) & 0x7F;
                if (s_st[((1 + pl) * 4 + cls) * 64 + (qp & 63)] == 0) break;
                if (!before(prv.x, cur.x, nxt.x, head)) break;      // that edge comes later: this one reads the original sample
                head--; kk -= step; nxt = cur; cur = prv;
            }
            // a chain to the RIGHT starts only where the CU to the left of this edge comes after the CU to its right (never without SUCO)
            if (DIR == 0 && ORD && (pos & ((1 << a.ctu_sh) - 1)) != 0 && (((SCU_RANK_OF(rp.x) - SCU_RANK_OF(rq.x)) << 21) >> 21) > 0) {
                uint4 lft = rp, mid = rq;                    // records of SCU tail-1, tail
                while (tail + 1 < npos) {
                    const uint4 rgt = maps[k0 + (tail + 1 - pos)];      // SCU tail+1: its left edge is edge tail+1
                    if (!(rgt.x & eflag) || (rgt.x & nflag) || on_tile_border(tail + 1)) break;
                    const int cls = edge_class(rgt, mid), qp = (rgt.x >> 16) & 0x7F;
                    if (s_st[((1 + pl) * 4 + cls) * 64 + (qp & 63)] == 0) break;
                    if (before(lft.x, mid.x, rgt.x, tail + 1)) break;  // edge tail is applied before edge tail+1
                    tail++; lft = mid; mid = rgt;
                }
            }
        }
#pragma unroll
        for (int ln = 0; ln < 2; ln++) {
            int *s = Cw[pl][ln];
            if (!stt) continue;
            if (tail > pos) {                                // the edges to the right that come first, from the far end: each hands its B' to the one before it
                const int16_t *p = (pl ? sv_ : su_) + cy * a.s_c + cx + ln * acrossc;
                int prevB = 0;
                for (int e = tail; e > pos; e--) {
                    const int rel = (e - pos) * 2;
                    const int ke = k0 + (e - pos) * step;
                    const uint4 q = maps[ke], pp = maps[ke - step];
                    const int cls = edge_class(q, pp), qp = (q.x >> 16) & 0x7F;
                    const int ste = s_st[((1 + pl) * 4 + cls) * 64 + (qp & 63)];
                    const int D = (e == tail) ? p[(rel + 1) * alongc] : prevB;
                    int Bo, Co;
                    filt_chroma(p[(rel - 2) * alongc], p[(rel - 1) * alongc], p[rel * alongc], D, ste, maxc, Bo, Co);
                    prevB = Bo;
                }
                s[3] = prevB;
            }
            int a_in = s[0];
            if (head < pos) {
                const int16_t *p = (pl ? sv_ : su_) + cy * a.s_c + cx + ln * acrossc;
                int prevC = 0;
                for (int e = head; e < pos; e++) {
                    const int rel = (e - pos) * 2;                   // chroma offset of edge e relative to this edge
                    const int ke = k0 + (e - pos) * step;
                    const uint4 q = maps[ke], pp = maps[ke - step];
                    const int cls = edge_class(q, pp), qp = (q.x >> 16) & 0x7F;
                    const int ste = s_st[((1 + pl) * 4 + cls) * 64 + (qp & 63)];
                    const int A = (e == head) ? p[(rel - 2) * alongc] : prevC;
                    int Bo, Co;
                    filt_chroma(A, p[(rel - 1) * alongc], p[rel * alongc], p[(rel + 1) * alongc], ste, maxc, Bo, Co);
                    prevC = Co;
                }
                a_in = prevC;
            }
            int Bo, Co;
            filt_chroma(a_in, s[1], s[2], s[3], stt, maxc, Bo, Co);
            s[1] = Bo; s[2] = Co;
        }
        int16_t *dst = (pl ? dv_ : du_) + cy * a.s_c + cx;
        if (DIR == 0) {      // line = row: B' at cx - 1, C' at cx
#pragma unroll
            for (int ln = 0; ln < 2; ln++) {
                if (in_range) { U32x1a2 w = { PK2(Cw[pl][ln][1], Cw[pl][ln][2]) }; *(U32x1a2 *)(dst + ln * a.s_c - 1) = w; }
                else if (has_p) dst[ln * a.s_c - 1] = (int16_t)Cw[pl][ln][1];
                else if (has_q) dst[ln * a.s_c] = (int16_t)Cw[pl][ln][2];
            }
        } else {             // line = column: row cy - 1 holds B' of both columns, row cy their C'
            if (has_p) *(uint32_t *)(dst - a.s_c) = PK2(Cw[pl][0][1], Cw[pl][1][1]);
            if (has_q) *(uint32_t *)dst = PK2(Cw[pl][0][2], Cw[pl][1][2]);
        }
    }
#undef PK2
}

void launch_dbk(xgpu_ctx *c, const DbkArgs &a, int dir, const DevPic &src, const DevPic &dst, bool order_rl)
{
    const int n_ex = dir == 0 ? a.w_scu + 1 : a.w_scu, n_ey = dir == 0 ? a.h_scu : a.h_scu + 1;
    const int tiles = dir == 0 ? ((n_ex + 63) >> 6) * ((n_ey + 3) >> 2) : ((n_ex + 15) >> 4) * ((n_ey + 15) >> 4);
    if (dir == 0 && order_rl)
        hipLaunchKernelGGL((k_dbk<0, true>), dim3(tiles), dim3(256), 0, c->stream, a, src.y, src.u, src.v, dst.y, dst.u, dst.v);
    else if (dir == 0)
        hipLaunchKernelGGL((k_dbk<0, false>), dim3(tiles), dim3(256), 0, c->stream, a, src.y, src.u, src.v, dst.y, dst.u, dst.v);
    else
        hipLaunchKernelGGL((k_dbk<1, false>), dim3(tiles), dim3(256), 0, c->stream, a, src.y, src.u, src.v, dst.y, dst.u, dst.v);
}

// ---------------------------------------------------------------------------------------------------------
// fine-grained shims: one edge segment through the kernels' own line filters with the call shape of the reference's
// XEVD_DBK / XEVD_DBK_CH table entries (src_base/xevd_def.h:363-364; deblock_scu_hor / _ver[_chroma], xevd_df.c:96-289).
// `buf` = the first sample on the far side of the edge; hor: the edge is horizontal (the filter runs down a column).
// ---------------------------------------------------------------------------------------------------------
__global__ void k_test_dbk(int16_t *buf, int16_t *buf_v, int st, int st_v, int stride, int bd, int hor, int chroma)
{
    const int i = threadIdx.x, maxv = (1 << bd) - 1;
    const int step = hor ? stride : 1, line = hor ? 1 : stride;              // along the filter / from line to line
    if (!chroma) {
        if (i >= 4) return;
        int16_t *p = buf + i * line;
        const Line4 o = filt_luma(Line4{ p[-2 * step], p[-step], p[0], p[step] }, st, maxv);
        p[-2 * step] = (int16_t)o.A; p[-step] = (int16_t)o.B; p[0] = (int16_t)o.C; p[step] = (int16_t)o.D;
    } else {
        if (i >= 4) return;                                                   // lanes 0,1: Cb; 2,3: Cr
        int16_t *p = (i < 2 ? buf : buf_v) + (i & 1) * line;
        const int s = i < 2 ? st : st_v;
        if (!s) return;
        int B, C;
        filt_chroma(p[-2 * step], p[-step], p[0], p[step], s, maxv, B, C);
        p[-step] = (int16_t)B; p[0] = (int16_t)C;
    }
}
void launch_test_dbk(xgpu_ctx *c, int16_t *buf, int16_t *buf_v, int st, int st_v, int stride, int bd, int hor, int chroma)
{
    hipLaunchKernelGGL(k_test_dbk, dim3(1), dim3(64), 0, c->stream, buf, buf_v, st, st_v, stride, bd, hor, chroma);
}

