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
//   * one LANE per 4x4 SCU produces all 16 luma + 2x(2x2) chroma samples of its SCU.  The samples of an SCU are
//     touched by two edges: its own left/top edge (C,D side) and the next SCU's edge (A,B side), so the lane
//     evaluates both filters and keeps its half - twice the trivial ALU work in exchange for zero write
//     conflicts and fully coalesced 128-byte row segments per 16 lanes;
//   * the chroma order dependence (edge k needs the C' sample of edge k-1 when that edge is active) is resolved
//     per lane by walking back to the head of the dependency chain and recomputing it forward in registers from
//     the ORIGINAL samples - chains are as long as a run of 4-wide CUs (typically 1-3), and every lane stays
//     independent, which is what lets the whole picture run as one flat grid;
//   * the per-edge decisions are pure lane-local integer tests on one 16-byte SCU record per side; the
//     strength tables (component x class x QP, built on the host from xevd_tbl_df_st and the chroma QP mapping)
//     are staged in LDS.
#include "xgpu_internal.h"

struct __attribute__((packed, aligned(4))) U32x4a4 { uint32_t a, b, c, d; };   // 16-byte load at 4-byte alignment

__device__ __forceinline__ int clip3i(int lo, int hi, int v) { return min(max(v, lo), hi); }

// get_tbl_qp_to_st, xevd_df.c:34-94.  q = record of the right/below SCU, p = left/above.
__device__ __forceinline__ int edge_class(const uint4 q, const uint4 p)
{
    if (((q.x | p.x) >> 15) & 1) return 0;
    if (((q.x | p.x) >> 24) & 1) return 1;
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

// DIR 0: vertical edges (filter along x), DIR 1: horizontal edges (filter along y)
template <int DIR>
__global__ __launch_bounds__(256) void k_dbk(const DbkArgs a, const int16_t *__restrict__ sy_, const int16_t *__restrict__ su_,
                                             const int16_t *__restrict__ sv_, int16_t *__restrict__ dy_, int16_t *__restrict__ du_,
                                             int16_t *__restrict__ dv_)
{
    __shared__ uint8_t s_st[3 * 4 * 64];
    for (int i = threadIdx.x; i < 3 * 4 * 64 / 4; i += 256) ((uint32_t *)s_st)[i] = ((const uint32_t *)a.st)[i];
    __syncthreads();

    // vertical edges: a wave = 64 neighbouring SCUs of one SCU row (512-byte runs of a picture row), 4 such rows per workgroup - 8 % faster at
    // 8K than a 16 x 16 lane tile; horizontal edges (8 rows of 8 bytes per lane) measured the other way round and keep the square tile
    const int LW = DIR == 0 ? 6 : 4, LH = 8 - LW;
    const int tiles_x = (a.w_scu + (1 << LW) - 1) >> LW;
    const int tx = blockIdx.x % tiles_x, ty = blockIdx.x / tiles_x;
    const int sx = (tx << LW) + (threadIdx.x & ((1 << LW) - 1)), sy = (ty << LH) + (threadIdx.x >> LW);
    if (sx >= a.w_scu || sy >= a.h_scu) return;
    const int step = DIR == 0 ? 1 : a.w_scu;                 // SCU-map step along the filtering axis
    const int pos = DIR == 0 ? sx : sy;                      // coordinate along the filtering axis
    const int npos = DIR == 0 ? a.w_scu : a.h_scu;
    const uint32_t eflag = DIR == 0 ? SCU_EDGE_L : SCU_EDGE_T;
    const uint4 *maps = (const uint4 *)a.maps;
    const int k0 = sy * a.w_scu + sx;
    const int maxl = (1 << a.bd_l) - 1, maxc = (1 << a.bd_c) - 1;

    // ---- everything this lane can need is requested up front (one memory round trip): the three SCU records,
    //      the luma window and the chroma windows of both planes; decisions come afterwards ----
    const uint4 rq = maps[k0];
    const uint4 rp = maps[pos > 0 ? k0 - step : k0];
    const uint4 rn = maps[pos + 1 < npos ? k0 + step : k0];
    const uint4 rpp = maps[pos > 1 ? k0 - 2 * step : k0];    // two back: the first link of a chroma dependency chain (below) without a second round trip
    const int x = sx << 2, y = sy << 2, cx = sx << 1, cy = sy << 1;
    U32x4a4 lrow[4];       // DIR 0: rows y..y+3, samples x-2..x+5
    uint2   lcol[8];       // DIR 1: rows y-2..y+5, samples x..x+3
    int cs[2][2][6];       // chroma [plane][line][A B C D | C' D' of the next edge]: 6 samples along the filtering axis from -2
    if (DIR == 0) {
#pragma unroll
        for (int r = 0; r < 4; r++) lrow[r] = *(const U32x4a4 *)(sy_ + (y + r) * a.s_l + x - 2);
#pragma unroll
        for (int pl = 0; pl < 2; pl++)
#pragma unroll
            for (int ln = 0; ln < 2; ln++) {
                const U32x4a4 v = *(const U32x4a4 *)((pl ? sv_ : su_) + (cy + ln) * a.s_c + cx - 2);
                cs[pl][ln][0] = (int16_t)(v.a & 0xFFFF); cs[pl][ln][1] = (int16_t)(v.a >> 16);
                cs[pl][ln][2] = (int16_t)(v.b & 0xFFFF); cs[pl][ln][3] = (int16_t)(v.b >> 16);
                cs[pl][ln][4] = (int16_t)(v.c & 0xFFFF); cs[pl][ln][5] = (int16_t)(v.c >> 16);
            }
    } else {
#pragma unroll
        for (int r = 0; r < 8; r++) lcol[r] = *(const uint2 *)(sy_ + (y - 2 + r) * a.s_l + x);
#pragma unroll
        for (int pl = 0; pl < 2; pl++)
#pragma unroll
            for (int r = 0; r < 6; r++) {
                const uint32_t v = *(const uint32_t *)((pl ? sv_ : su_) + (cy - 2 + r) * a.s_c + cx);
                cs[pl][0][r] = (int16_t)(v & 0xFFFF); cs[pl][1][r] = (int16_t)(v >> 16);
            }
    }

    // edge on this SCU's left/top side, and the edge on the far side (belongs to the next SCU)
    int st_this[3] = {0, 0, 0}, st_next[3] = {0, 0, 0};
    if (pos > 0 && (rq.x & eflag)) {
        const int cls = edge_class(rq, rp), qp = (rq.x >> 16) & 0x7F;
#pragma unroll
        for (int c = 0; c < 3; c++) st_this[c] = s_st[(c * 4 + cls) * 64 + (qp & 63)];
    }
    if (pos + 1 < npos && (rn.x & eflag)) {
        const int cls = edge_class(rn, rq), qp = (rn.x >> 16) & 0x7F;
#pragma unroll
        for (int c = 0; c < 3; c++) st_next[c] = s_st[(c * 4 + cls) * 64 + (qp & 63)];
    }

    // ------------------------------------------------ luma -----------------------------------------------
    if (DIR == 0) {
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const U32x4a4 v = lrow[r];
            int s[8] = { (int16_t)(v.a & 0xFFFF), (int16_t)(v.a >> 16), (int16_t)(v.b & 0xFFFF), (int16_t)(v.b >> 16),
                         (int16_t)(v.c & 0xFFFF), (int16_t)(v.c >> 16), (int16_t)(v.d & 0xFFFF), (int16_t)(v.d >> 16) };
            if (st_this[0]) { const Line4 o = filt_luma({ s[0], s[1], s[2], s[3] }, st_this[0], maxl); s[2] = o.C; s[3] = o.D; }
            if (st_next[0]) { const Line4 o = filt_luma({ s[4], s[5], s[6], s[7] }, st_next[0], maxl); s[4] = o.A; s[5] = o.B; }
            uint2 w;
            w.x = (uint32_t)(uint16_t)s[2] | ((uint32_t)(uint16_t)s[3] << 16);
            w.y = (uint32_t)(uint16_t)s[4] | ((uint32_t)(uint16_t)s[5] << 16);
            *(uint2 *)(dy_ + (y + r) * a.s_l + x) = w;
        }
    } else {
#pragma unroll
        for (int c = 0; c < 4; c++) {
            int s[8];
#pragma unroll
            for (int r = 0; r < 8; r++) {
                const uint32_t d = (c < 2) ? lcol[r].x : lcol[r].y;
                s[r] = (c & 1) ? (int16_t)(d >> 16) : (int16_t)(d & 0xFFFF);
            }
            if (st_this[0]) { const Line4 o = filt_luma({ s[0], s[1], s[2], s[3] }, st_this[0], maxl); s[2] = o.C; s[3] = o.D; }
            if (st_next[0]) { const Line4 o = filt_luma({ s[4], s[5], s[6], s[7] }, st_next[0], maxl); s[4] = o.A; s[5] = o.B; }
#pragma unroll
            for (int r = 2; r < 6; r++) {
                uint32_t &d = (c < 2) ? lcol[r].x : lcol[r].y;
                d = (c & 1) ? ((d & 0xFFFFu) | ((uint32_t)(uint16_t)s[r] << 16)) : ((d & 0xFFFF0000u) | (uint32_t)(uint16_t)s[r]);
            }
        }
#pragma unroll
        for (int r = 2; r < 6; r++) *(uint2 *)(dy_ + (y - 2 + r) * a.s_l + x) = lcol[r];
    }

    // ------------------------------------------------ chroma ---------------------------------------------
    // The SCU owns chroma samples (cx..cx+1, cy..cy+1).  Along the filtering axis, sample 0 is C' of this SCU's
    // edge and sample 1 is B' of the next SCU's edge; both need the final value of the sample two positions
    // before the edge (A), i.e. the C' of the previous edge when that edge is active.
    const int alongc = DIR == 0 ? 1 : a.s_c, acrossc = DIR == 0 ? a.s_c : 1;
    int outc[2][2][2];     // [plane][line][sample along the axis]
#pragma unroll
    for (int pl = 0; pl < 2; pl++) {
        const int stt = st_this[1 + pl], stn = st_next[1 + pl];
        // head of the chain of consecutive active edges that ends at this SCU's edge (xevd_df.c:238-289 order
        // dependence); identical for both lines of the SCU.  Rare (runs of 4-wide / 4-tall CUs), so the extra
        // records and samples are fetched inside the loop.
        int head = pos;
        if (stt) {
            int kk = k0;
            uint4 cur = rp;                                  // record of SCU head-1
            while (head - 1 > 0) {
                if (!(cur.x & eflag)) break;
                const uint4 prv = kk == k0 ? rpp : maps[kk - 2 * step];
                const int cls = edge_class(cur, prv), qp = (cur.x >> 16) & 0x7F;
                if (s_st[((1 + pl) * 4 + cls) * 64 + (qp & 63)] == 0) break;
                head--; kk -= step; cur = prv;
            }
        }
#pragma unroll
        for (int ln = 0; ln < 2; ln++) {
            const int *s = cs[pl][ln];
            int o0 = s[2], o1 = s[3], a_in = s[0];
            if (stt) {
                if (head < pos) {
                    // recompute the chain forward from its head using ORIGINAL samples
                    const int16_t *p = (pl ? sv_ : su_) + cy * a.s_c + cx + ln * acrossc;
                    int prevC = 0;
                    for (int e = head; e < pos; e++) {
                        const int rel = (e - pos) * 2;                   // chroma offset of edge e relative to this edge
                        const int ke = k0 + (e - pos) * step;
                        const uint4 q = maps[ke], pp = maps[ke - step];
                        const int cls = edge_class(q, pp), qp = (q.x >> 16) & 0x7F;
                        const int st = s_st[((1 + pl) * 4 + cls) * 64 + (qp & 63)];
                        const int A = (e == head) ? p[(rel - 2) * alongc] : prevC;
                        int Bo, Co;
                        filt_chroma(A, p[(rel - 1) * alongc], p[rel * alongc], p[(rel + 1) * alongc], st, maxc, Bo, Co);
                        prevC = Co;
                    }
                    a_in = prevC;
                }
                int Bo, Co;
                filt_chroma(a_in, s[1], s[2], s[3], stt, maxc, Bo, Co);
                o0 = Co;
            }
            if (stn) {
                int Bo, Co;
                filt_chroma(o0, s[3], s[4], s[5], stn, maxc, Bo, Co);
                o1 = Bo;
            }
            outc[pl][ln][0] = o0; outc[pl][ln][1] = o1;
        }
    }
#pragma unroll
    for (int pl = 0; pl < 2; pl++) {
        int16_t *dst = (pl ? dv_ : du_) + cy * a.s_c + cx;
        if (DIR == 0) {      // line = row, samples along x: one dword per row
            *(uint32_t *)dst = (uint32_t)(uint16_t)outc[pl][0][0] | ((uint32_t)(uint16_t)outc[pl][0][1] << 16);
            *(uint32_t *)(dst + a.s_c) = (uint32_t)(uint16_t)outc[pl][1][0] | ((uint32_t)(uint16_t)outc[pl][1][1] << 16);
        } else {             // line = column, samples along y: row cy holds sample 0 of both columns
            *(uint32_t *)dst = (uint32_t)(uint16_t)outc[pl][0][0] | ((uint32_t)(uint16_t)outc[pl][1][0] << 16);
            *(uint32_t *)(dst + a.s_c) = (uint32_t)(uint16_t)outc[pl][0][1] | ((uint32_t)(uint16_t)outc[pl][1][1] << 16);
        }
    }
}

void launch_dbk(xgpu_ctx *c, const DbkArgs &a, int dir, const DevPic &src, const DevPic &dst)
{
    const int tiles = dir == 0 ? ((a.w_scu + 63) >> 6) * ((a.h_scu + 3) >> 2) : ((a.w_scu + 15) >> 4) * ((a.h_scu + 15) >> 4);
    if (dir == 0)
        hipLaunchKernelGGL(k_dbk<0>, dim3(tiles), dim3(256), 0, c->stream, a, src.y, src.u, src.v, dst.y, dst.u, dst.v);
    else
        hipLaunchKernelGGL(k_dbk<1>, dim3(tiles), dim3(256), 0, c->stream, a, src.y, src.u, src.v, dst.y, dst.u, dst.v);
}
