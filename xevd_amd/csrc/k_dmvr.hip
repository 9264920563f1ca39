// k_dmvr.hip - decoder-side motion vector refinement (Main, sps->tool_dmvr) of merge-mode bi-predicted CUs: search, refined prediction,
// bi-prediction average, residual add + clip, and the refined vectors for the host's temporal motion prediction.
//
// Replaces xevdm_mc's apply_DMVR branch (src_main/xevdm_mc.c:1860-2038) with processDMVR (:1647-1829): per 16x16 sub-block, bilinear
// pre-interpolation of both lists two samples wider than the block (xevdm_bl_mc_l :358-486), up to two rounds of a 5-point SAD search with
// mirrored offsets (xevd_DMVR_refine :1293-1339, xevd_DMVR_cost :1270-1291), a parametric sub-sample step through the cross of costs of the
// last round (xevd_SubPelErrorSrfc :1373-1427, div_for_maxq7 :1341-1372), then the 8 / 4-tap interpolation at the refined sixteenth-sample
// vector out of a window fetched at the STARTING vector and replicate-padded by 2 / 1 samples (prefetch_for_mc :1481-1544,
// final_paddedMC_forDMVR :1548-1644; xevd_mc_dmvr_l_* :224-355, _c_* :490-625).  The SCU map keeps the unrefined vectors (what ADDB reads,
// xevdm.c:2009-2041): k_inter writes it and leaves the samples of these CUs to this kernel - both evaluate dmvr_applies().
//
// MI355X mapping: one WAVE per sub-block (the unit the reference refines independently), four per workgroup, no workgroup barrier.  The two
// bilinear blocks live in the wave's LDS; a SAD is 4 absolute differences per lane and a 6-step cross-lane sum, so the whole search - at most
// 11 SADs, wave-uniform control flow - is a few hundred instructions.  The refined prediction reads its windows once into LDS; the padding of
// the reference's scratch buffer is clamped indexing into that window.  Instruction counts do not matter here (merge-mode bi-predicted CUs
// with symmetric references are a fraction of a picture); latency does: every global access of a sub-block is issued in one sweep per phase.
#include "xgpu_internal.h"
#include "mc_filters.h"

#define DM_BL   20            // bilinear block: (16 + 4)^2
#define DM_RS   24            // row stride of the staged windows: three 16-byte chunks (the 21 / 23 samples a 16-wide sub-block needs per row)
#define DM_CS   16            // row stride of a chroma window: two chunks (11 samples)
#define DM_WIN  (23 * DM_RS + 2 * 11 * DM_CS)      // luma + two chroma windows of the final prediction; the two 21-row search windows (2 x 21 x 24) use the same space
#define DM_STAGE (23 * 40 + 2 * 11 * 24)      // the padded windows of dmvr_predict_packed (luma 23 rows x 40, chroma 2 x 11 x 24); the search windows (2 x 21 x 24) and the scalar form's fit inside
#define DM_TMP   (29 * 16)                   // intermediate rows of the horizontal pass: window rows + 3 of padding above / below

__device__ __forceinline__ void dm_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
// sum over the 64 lanes, the same value in every lane: two quad permutes and two mirrors (DPP, no LDS traffic) leave every lane of a row of 16 with the row's sum,
// the four row sums are added on the scalar unit
__device__ __forceinline__ int wave_sum(int v)
{
    v += __builtin_amdgcn_update_dpp(0, v, 0xB1, 0xF, 0xF, true);      // quad_perm [1, 0, 3, 2]
    v += __builtin_amdgcn_update_dpp(0, v, 0x4E, 0xF, 0xF, true);      // quad_perm [2, 3, 0, 1]
    v += __builtin_amdgcn_update_dpp(0, v, 0x141, 0xF, 0xF, true);     // row_half_mirror
    v += __builtin_amdgcn_update_dpp(0, v, 0x140, 0xF, 0xF, true);     // row_mirror
    return __builtin_amdgcn_readlane(v, 0) + __builtin_amdgcn_readlane(v, 16) + __builtin_amdgcn_readlane(v, 32) + __builtin_amdgcn_readlane(v, 48);
}
__device__ __forceinline__ int dm_div_q7(long long n, long long d)      // div_for_maxq7: three bits of n / d
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
__device__ __forceinline__ int clampi(int v, int lo, int hi) { return min(max(v, lo), hi); }

struct DmvrCu {        // the wave-uniform state of one sub-block
    int cu_x, cu_y, cw, chh, cbf, px, py, ats;
    uint32_t coef_off;
    int st[2][2], mvu[2][2];
    const int16_t *ry[2], *ru[2], *rv[2];      // the two references' planes (fields, not RefEntry copies: those went through scratch)
    int rpoc[2];
};

__device__ __forceinline__ void dmvr_unpack(const DmvrArgs &a, const uint4 r0, const uint4 r1, int isx, int isy, DmvrCu &u)
{
    u.cu_x = r0.x & 0xFFFF; u.cu_y = r0.x >> 16; u.cw = 1 << (r0.y & 0xFF); u.chh = 1 << ((r0.y >> 8) & 0xFF); u.cbf = r0.y >> 24;
    const int refi[2] = { (int)(int8_t)(r0.z & 0xFF), (int)(int8_t)((r0.z >> 8) & 0xFF) };
    u.coef_off = r0.w; u.ats = (int)((r1.w >> 8) & 0xFF);
    u.mvu[0][0] = (int)(int16_t)(r1.x & 0xFFFF); u.mvu[0][1] = (int)(int16_t)(r1.x >> 16); u.mvu[1][0] = (int)(int16_t)(r1.y & 0xFFFF); u.mvu[1][1] = (int)(int16_t)(r1.y >> 16);
    u.px = u.cu_x + isx * 4; u.py = u.cu_y + isy * 4;
    // starting vectors: the CU's, clipped like xevd_mv_clip (mv_clip, xevdm_mc.c:1831-1858)
    {
        const int min_c = -(128 << 2), max_x = (a.pic_w - 1 + 128) << 2, max_y = (a.pic_h - 1 + 128) << 2;
#pragma unroll
        for (int l = 0; l < 2; l++) {
            int mx = u.mvu[l][0], my = u.mvu[l][1];
            if ((u.cu_x << 2) + u.mvu[l][0] < min_c) mx = min_c - (u.cu_x << 2);
            if ((u.cu_y << 2) + u.mvu[l][1] < min_c) my = min_c - (u.cu_y << 2);
            if ((u.cu_x << 2) + u.mvu[l][0] + (u.cw << 2) - 4 > max_x) mx = max_x - (u.cu_x << 2) - (u.cw << 2) + 4;
            if ((u.cu_y << 2) + u.mvu[l][1] + (u.chh << 2) - 4 > max_y) my = max_y - (u.cu_y << 2) - (u.chh << 2) + 4;
            u.st[l][0] = mx; u.st[l][1] = my;
        }
    }
#pragma unroll
    for (int l = 0; l < 2; l++) { const RefEntry &e = a.refp[refi[l]][l]; u.ry[l] = e.y; u.ru[l] = e.u; u.rv[l] = e.v; u.rpoc[l] = e.poc; }
}

// The refined prediction, sample by sample with clamped window indices (any offset between the refined and the starting vector): the fallback of
// dmvr_predict_packed below.
template <int DX, int DY>
__device__ __forceinline__ void dmvr_predict_scalar(const DmvrArgs &a, const DmvrCu &u, const int r16[2][2], int16_t *W, int16_t *T, int lane)
{
    constexpr int CH = DX == 16 ? 3 : 2;
    const int bd = a.bd_l, maxl = (1 << a.bd_l) - 1, maxc = (1 << a.bd_c) - 1;
    const int sh1 = min(4, bd - 8), sh2 = max(8, 20 - bd), off2 = 1 << (sh2 - 1);
    // ---- the refined prediction: lane = 4 luma samples of a row (lanes below DX*DY/4) and one chroma sample per plane (lanes below DX*DY/4) ----
    constexpr int NL4 = (DX * DY) >> 2, NC = (DX >> 1) * (DY >> 1), WCW = (DX >> 1) + 3, WCH = (DY >> 1) + 3;
    const int lr = (lane * 4) / DX, lc = (lane * 4) - lr * DX;                    // luma row / first column inside the sub-block
    const int cr = lane / (DX >> 1), cc = lane - cr * (DX >> 1);                  // chroma row / column
    int accl[4] = { 0, 0, 0, 0 }, accu = 0, accv = 0;
#pragma unroll
    for (int l = 0; l < 2; l++) {
        // clip of the refined vector at the sub-block (mv_clip_only_one_ref_dmvr :939-980)
        int tq[2] = { (int)(int16_t)(r16[l][0] >> 2), (int)(int16_t)(r16[l][1] >> 2) }, mc[2] = { tq[0], tq[1] }, clip = 0;
        {
            const int min_c = -(128 << 2), max_x = (a.pic_w - 1 + 128) << 2, max_y = (a.pic_h - 1 + 128) << 2;
            if ((u.px << 2) + tq[0] < min_c) { clip = 1; mc[0] = min_c - (u.px << 2); }
            if ((u.py << 2) + tq[1] < min_c) { clip = 1; mc[1] = min_c - (u.py << 2); }
            if ((u.px << 2) + tq[0] + (DX << 2) - 4 > max_x) { clip = 1; mc[0] = max_x - (u.px << 2) - (DX << 2) + 4; }
            if ((u.py << 2) + tq[1] + (DY << 2) - 4 > max_y) { clip = 1; mc[1] = max_y - (u.py << 2) - (DY << 2) + 4; }
            mc[0] = (int)(int16_t)mc[0]; mc[1] = (int)(int16_t)mc[1];
        }
        const int gx = (u.px << 4) + (clip ? mc[0] << 2 : r16[l][0]), gy = (u.py << 4) + (clip ? mc[1] << 2 : r16[l][1]);
        const int dlx = (clip ? mc[0] >> 2 : r16[l][0] >> 4) - (u.st[l][0] >> 2), dly = (clip ? mc[1] >> 2 : r16[l][1] >> 4) - (u.st[l][1] >> 2);
        const int dcx = (clip ? mc[0] >> 3 : r16[l][0] >> 5) - (u.st[l][0] >> 3), dcy = (clip ? mc[1] >> 3 : r16[l][1] >> 5) - (u.st[l][1] >> 3);
        // windows at the STARTING vector: luma (DX + 7) x (DY + 7) from 3 samples up-left, chroma (DX/2 + 3) x (DY/2 + 3) from 1 sample up-left
        const int q16x = ((u.px << 2) + u.st[l][0]) << 2, q16y = ((u.py << 2) + u.st[l][1]) << 2;
        {
            const gs16 sy_ = (gs16)u.ry[l] + ((q16y >> 4) - 3) * a.s_l + (q16x >> 4) - 3;
            for (int i = lane; i < (DY + 7) * CH; i += 64) { const int r = i / CH, k = i - r * CH; *(uint4 *)(W + r * DM_RS + 8 * k) = gload16(sy_ + r * a.s_l + 8 * k); }
            const int co = ((q16y >> 5) - 1) * a.s_c + (q16x >> 5) - 1;
            const gs16 su_ = (gs16)u.ru[l] + co, sv_ = (gs16)u.rv[l] + co;
            for (int i = lane; i < 2 * WCH * 2; i += 64) {      // two planes x rows x two chunks
                const int pl = i >= WCH * 2, j = i - pl * WCH * 2, r = j >> 1, k = j & 1;
                *(uint4 *)(W + 23 * DM_RS + pl * 11 * DM_CS + r * DM_CS + 8 * k) = gload16((pl ? sv_ : su_) + r * a.s_c + 8 * k);
            }
        }
        dm_sync();
        // luma: horizontal pass over every window row (the rows of the padding are copies of window rows), then vertical
        {
            const int fx = gx & 15, fy = gy & 15;
            int th[8], tv[8];
#pragma unroll
            for (int k = 0; k < 8; k++) { th[k] = (int)(int16_t)(k_luma_taps[a.admvp][fx][k >> 1] >> ((k & 1) * 16)); tv[k] = (int)(int16_t)(k_luma_taps[a.admvp][fy][k >> 1] >> ((k & 1) * 16)); }
            for (int i = lane; i < (DY + 7) * DX; i += 64) {
                const int r = i / DX, j = i - r * DX;
                int v;
                if (!fx) v = W[r * DM_RS + clampi(3 + dlx + j, 0, DX + 6)];
                else {
                    int sacc = 0;
#pragma unroll
                    for (int k = 0; k < 8; k++) sacc += th[k] * (int)W[r * DM_RS + clampi(dlx + j + k, 0, DX + 6)];
                    v = fy ? (int)(int16_t)(sacc >> sh1) : clampi(sacc >> 6, 0, maxl);
                }
                T[r * 16 + j] = (int16_t)v;
            }
            dm_sync();
            if (lane < NL4) {
#pragma unroll
                for (int e = 0; e < 4; e++) {
                    int v;
                    if (!fy) v = T[clampi(3 + dly + lr, 0, DY + 6) * 16 + lc + e];
                    else {
                        int sacc = 0;
#pragma unroll
                        for (int k = 0; k < 8; k++) sacc += tv[k] * (int)T[clampi(dly + lr + k, 0, DY + 6) * 16 + lc + e];
                        v = fx ? clampi((sacc + off2) >> sh2, 0, maxl) : clampi(sacc >> 6, 0, maxl);
                    }
                    accl[e] = l ? (accl[e] + v + 1) >> 1 : v;
                }
            }
            dm_sync();
        }
        // chroma, both planes: the same with the 4-tap tables at the thirty-second-sample phase
        {
            const int fx = gx & 31, fy = gy & 31;
            constexpr int CWD = DX >> 1;
            int th[4], tv[4];
#pragma unroll
            for (int k = 0; k < 4; k++) { th[k] = (int)(int16_t)(k_chroma_taps[a.admvp][fx][k >> 1] >> ((k & 1) * 16)); tv[k] = (int)(int16_t)(k_chroma_taps[a.admvp][fy][k >> 1] >> ((k & 1) * 16)); }
            const int shc1 = min(4, a.bd_c - 8), shc2 = max(8, 20 - a.bd_c), offc2 = 1 << (shc2 - 1);
            for (int i = lane; i < 2 * WCH * CWD; i += 64) {
                const int pl = i >= WCH * CWD, q = i - pl * WCH * CWD, r = q / CWD, j = q - r * CWD;
                const int16_t *Wc = W + 23 * DM_RS + pl * 11 * DM_CS;
                int v;
                if (!fx) v = Wc[r * DM_CS + clampi(1 + dcx + j, 0, WCW - 1)];
                else {
                    int sacc = 0;
#pragma unroll
                    for (int k = 0; k < 4; k++) sacc += th[k] * (int)Wc[r * DM_CS + clampi(dcx + j + k, 0, WCW - 1)];
                    v = fy ? (int)(int16_t)(sacc >> shc1) : clampi(sacc >> 6, 0, maxc);
                }
                T[pl * 11 * 8 + r * 8 + j] = (int16_t)v;
            }
            dm_sync();
            if (lane < NC) {
#pragma unroll
                for (int pl = 0; pl < 2; pl++) {
                    const int16_t *Tc = T + pl * 11 * 8;
                    int v;
                    if (!fy) v = Tc[clampi(1 + dcy + cr, 0, WCH - 1) * 8 + cc];
                    else {
                        int sacc = 0;
#pragma unroll
                        for (int k = 0; k < 4; k++) sacc += tv[k] * (int)Tc[clampi(dcy + cr + k, 0, WCH - 1) * 8 + cc];
                        v = fx ? clampi((sacc + offc2) >> shc2, 0, maxc) : clampi(sacc >> 6, 0, maxc);
                    }
                    if (pl) accv = l ? (accv + v + 1) >> 1 : v; else accu = l ? (accu + v + 1) >> 1 : v;
                }
            }
            dm_sync();
        }
    }

    // ---- residual add + clip (xevd_recon.c:35-71; the LUMA bit depth clips all three components) and the stores ----
    int tu_x = 0, tu_y = 0, tu_w = u.cw, tu_h = u.chh;
    if (u.ats) {      // ATS-inter: the coded TU is one half / quarter of the CU at its start or end (xevdm_util.c:3585-3634)
        const int idx = u.ats & 15, pos = u.ats >> 4;
        if (idx == 2 || idx == 4) { tu_h = u.chh >> (idx == 4 ? 2 : 1); tu_y = pos ? u.chh - tu_h : 0; }
        else                      { tu_w = u.cw >> (idx == 3 ? 2 : 1);  tu_x = pos ? u.cw - tu_w : 0; }
    }
    const int cwc = tu_w >> 1, cbf = u.cbf;
    const uint32_t off_u = u.coef_off + ((cbf & 1) ? tu_w * tu_h : 0), off_v = off_u + ((cbf & 2) ? cwc * (tu_h >> 1) : 0);
    if (lane < NL4) {
        const int x = u.px + lc, y = u.py + lr, lx = x - u.cu_x - tu_x, ly = y - u.cu_y - tu_y;
        int o[4];
#pragma unroll
        for (int e = 0; e < 4; e++) {
            o[e] = accl[e];
            if ((cbf & 1) && (uint32_t)(lx + e) < (uint32_t)tu_w && (uint32_t)ly < (uint32_t)tu_h)
                o[e] = clampi((int)(int16_t)(a.resid[u.coef_off + ly * tu_w + lx + e] + o[e]), 0, maxl);
        }
        *(uint2 *)(a.cur_y + y * a.s_l + x) = make_uint2((uint32_t)(uint16_t)o[0] | ((uint32_t)(uint16_t)o[1] << 16), (uint32_t)(uint16_t)o[2] | ((uint32_t)(uint16_t)o[3] << 16));
    }
    if (lane < NC) {
        const int xc = (u.px >> 1) + cc, yc = (u.py >> 1) + cr, lxc = xc - ((u.cu_x + tu_x) >> 1), lyc = yc - ((u.cu_y + tu_y) >> 1);
        const bool in_tu = (uint32_t)lxc < (uint32_t)cwc && (uint32_t)lyc < (uint32_t)(tu_h >> 1);
        int uu = accu, vv = accv;
        if ((cbf & 2) && in_tu) uu = clampi((int)(int16_t)(a.resid[off_u + lyc * cwc + lxc] + uu), 0, maxl);
        if ((cbf & 4) && in_tu) vv = clampi((int)(int16_t)(a.resid[off_v + lyc * cwc + lxc] + vv), 0, maxl);
        a.cur_u[yc * a.s_c + xc] = (int16_t)uu;
        a.cur_v[yc * a.s_c + xc] = (int16_t)vv;
    }
}

// ---------------------------------------------------------------------------------------------------------
// The refined prediction on packed sample pairs (round 3; the scalar form above took 1200 of the kernel's ~2300 instructions per sub-block).
// The windows fetched at the STARTING vector go into LDS with room for the padding the reference adds around them (2 luma / 1 chroma samples, plus the
// rounding of the sub-sample split: the refined vector's whole-sample part lies at most 3 luma / 2 chroma samples from the starting vector's): window sample s
// sits at column 8 + s of its row, the 3 (2) columns on either side are filled with the edge samples, rows are clamped when they are read.  With that no index
// needs a clamp: a lane filters FOUR neighbouring outputs from 6 (4) aligned dwords with v_dot2_i32_i16 on the packed tap pairs of mc_filters.h - the even and
// odd outputs of a dword run swap roles with the parity of the offset -, the vertical pass pairs rows with v_perm like k_inter's tile path.  Same rounding
// regimes as the scalar form (xevd_mc_dmvr_l_00 / n0 / 0n / nn, src_main/xevdm_mc.c:224-355; chroma :490-625).
// ---------------------------------------------------------------------------------------------------------
#define DM_LWS 40            // padded luma window row stride in samples: [8 | three 8-sample chunks | 8]
#define DM_CWS 24            // padded chroma window row: [8 | two chunks]
template <int DX, int DY>
__device__ __forceinline__ void dmvr_predict_packed(const DmvrArgs &a, const DmvrCu &u, const int r16[2][2], const int dl[2][2], const int dc[2][2], const int gxy[2][2],
                                                    int16_t *W, int16_t *T, int lane)
{
    constexpr int CH = DX == 16 ? 3 : 2;
    constexpr int NL4 = (DX * DY) >> 2, WCW = (DX >> 1) + 3, WCH = (DY >> 1) + 3, G = DX >> 2, GC = DX >> 3, RL = DY + 13, RC = WCH + 4;
    constexpr int NC4 = 2 * (DY >> 1) * GC;                          // lanes of the chroma vertical pass: plane x row x group of four outputs
    static_assert((DY + 7) * DM_LWS + 2 * WCH * DM_CWS <= DM_STAGE && RL * 16 <= DM_TMP && 2 * RC * 8 <= DM_TMP, "LDS budget of a wave");
    const int bd = a.bd_l, maxl = (1 << a.bd_l) - 1, maxc = (1 << a.bd_c) - 1;
    const int sh1 = min(4, bd - 8), sh2 = max(8, 20 - bd), off2 = 1 << (sh2 - 1);
    const int shc1 = min(4, a.bd_c - 8), shc2 = max(8, 20 - a.bd_c), offc2 = 1 << (shc2 - 1);
    int16_t *Wc = W + (DY + 7) * DM_LWS;                             // the two chroma windows behind the luma one
    const int lr = (lane * 4) / DX, lc = (lane * 4) - lr * DX;       // luma: row / first column of the lane's four outputs (lanes below NL4)
    const int cpl = lane / ((DY >> 1) * GC), crem = lane - cpl * ((DY >> 1) * GC), cr = crem / GC, cg = crem - cr * GC;      // chroma: plane, row, group (lanes below NC4)
    uint32_t pl[2] = { 0, 0 }, pc[2] = { 0, 0 };
#pragma unroll
    for (int l = 0; l < 2; l++) {
        // ---- windows at the starting vector -> LDS, then the padding columns ----
        const int q16x = ((u.px << 2) + u.st[l][0]) << 2, q16y = ((u.py << 2) + u.st[l][1]) << 2;
        {
            const gs16 sy_ = (gs16)u.ry[l] + ((q16y >> 4) - 3) * a.s_l + (q16x >> 4) - 3;
            // all loads of the list first, then the LDS stores (a loop of load + store waits for every round on its own: three memory round trips instead of one)
            constexpr int NLW = (DY + 7) * CH, NCW = 2 * WCH * 2;      // 16-byte chunks of the luma window (at most 69: two per lane) and of the two chroma windows (at most 44)
            static_assert(NLW <= 128 && NCW <= 64, "one or two chunks per lane");
            const int co = ((q16y >> 5) - 1) * a.s_c + (q16x >> 5) - 1;
            const gs16 su_ = (gs16)u.ru[l] + co, sv_ = (gs16)u.rv[l] + co;
            const int r0 = lane / CH, k0 = lane - r0 * CH, r1 = (lane + 64) / CH, k1 = (lane + 64) - r1 * CH;
            const int cp = lane >= WCH * 2, cj = lane - cp * WCH * 2, cr_ = cj >> 1, ck = cj & 1;
            uint4 v0 = make_uint4(0, 0, 0, 0), v1 = v0, vc = v0;
            if (lane < NLW) v0 = gload16(sy_ + r0 * a.s_l + 8 * k0);
            if (lane + 64 < NLW) v1 = gload16(sy_ + r1 * a.s_l + 8 * k1);
            if (lane < NCW) vc = gload16((cp ? sv_ : su_) + cr_ * a.s_c + 8 * ck);
            if (lane < NLW) *(uint4 *)(W + r0 * DM_LWS + 8 + 8 * k0) = v0;
            if (lane + 64 < NLW) *(uint4 *)(W + r1 * DM_LWS + 8 + 8 * k1) = v1;
            if (lane < NCW) *(uint4 *)(Wc + cp * WCH * DM_CWS + cr_ * DM_CWS + 8 + 8 * ck) = vc;
        }
        dm_sync();
        for (int i = lane; i < 2 * (DY + 7) + 4 * WCH; i += 64) {
            if (i < 2 * (DY + 7)) {
                int16_t *row = W + (i >> 1) * DM_LWS;
                if (i & 1) { const int16_t v = row[8 + DX + 6]; row[8 + DX + 7] = v; row[8 + DX + 8] = v; row[8 + DX + 9] = v; }
                else       { const int16_t v = row[8]; row[5] = v; row[6] = v; row[7] = v; }
            } else {
                const int j = i - 2 * (DY + 7);
                int16_t *row = Wc + (j >> 1) * DM_CWS;                 // (the two planes' rows follow each other)
                if (j & 1) { const int16_t v = row[8 + WCW - 1]; row[8 + WCW] = v; row[8 + WCW + 1] = v; }
                else       { const int16_t v = row[8]; row[6] = v; row[7] = v; }
            }
        }
        dm_sync();
        // ---- luma: horizontal pass over the window rows and the three rows of padding above / below (copies of the edge rows), then vertical ----
        {
            const int fx = gxy[l][0] & 15, fy = gxy[l][1] & 15, dlx = dl[l][0], dly = dl[l][1];
            const uint32_t ch0 = k_luma_taps[a.admvp][fx][0], ch1 = k_luma_taps[a.admvp][fx][1], ch2 = k_luma_taps[a.admvp][fx][2], ch3 = k_luma_taps[a.admvp][fx][3];
            for (int i = lane; i < RL * G; i += 64) {
                const int rr = i / G, g = i - rr * G, src = clampi(rr - 3, 0, DY + 6);
                uint2 o;
                if (fx) {
                    const int base = 8 + dlx + 4 * g;                 // column of the first tap of output 4 g
                    const uint32_t *p = (const uint32_t *)(W + src * DM_LWS + (base & ~1));
                    const uint32_t D0 = p[0], D1 = p[1], D2 = p[2], D3 = p[3], D4 = p[4], D5 = p[5];
                    const uint32_t Q0 = hi_lo(D1, D0), Q1 = hi_lo(D2, D1), Q2 = hi_lo(D3, D2), Q3 = hi_lo(D4, D3), Q4 = hi_lo(D5, D4);
                    int t[4];
                    if (base & 1) {                                   // wave-uniform (the parity of dlx)
                        t[0] = dot2(ch3, Q3, dot2(ch2, Q2, dot2(ch1, Q1, dot2z(ch0, Q0))));
                        t[1] = dot2(ch3, D4, dot2(ch2, D3, dot2(ch1, D2, dot2z(ch0, D1))));
                        t[2] = dot2(ch3, Q4, dot2(ch2, Q3, dot2(ch1, Q2, dot2z(ch0, Q1))));
                        t[3] = dot2(ch3, D5, dot2(ch2, D4, dot2(ch1, D3, dot2z(ch0, D2))));
                    } else {
                        t[0] = dot2(ch3, D3, dot2(ch2, D2, dot2(ch1, D1, dot2z(ch0, D0))));
                        t[1] = dot2(ch3, Q3, dot2(ch2, Q2, dot2(ch1, Q1, dot2z(ch0, Q0))));
                        t[2] = dot2(ch3, D4, dot2(ch2, D3, dot2(ch1, D2, dot2z(ch0, D1))));
                        t[3] = dot2(ch3, Q4, dot2(ch2, Q3, dot2(ch1, Q2, dot2z(ch0, Q1))));
                    }
#pragma unroll
                    for (int q = 0; q < 4; q++) t[q] = fy ? t[q] >> sh1 : clampi(t[q] >> 6, 0, maxl);
                    o = make_uint2(pack2(t[0], t[1]), pack2(t[2], t[3]));
                } else {
                    const int c0 = 8 + 3 + dlx + 4 * g;
                    const uint32_t *p = (const uint32_t *)(W + src * DM_LWS + (c0 & ~1));
                    const uint32_t D0 = p[0], D1 = p[1], D2 = p[2];
                    o = (c0 & 1) ? make_uint2(hi_lo(D1, D0), hi_lo(D2, D1)) : make_uint2(D0, D1);
                }
                *(uint2 *)(T + rr * 16 + 4 * g) = o;
            }
            dm_sync();
            if (lane < NL4) {
                const int16_t *tb = T + (3 + dly + lr) * 16 + lc;     // the first of the eight rows under the taps (the row itself: + 3)
                uint32_t v0, v1;
                if (fy) {
                    uint2 q[8];
#pragma unroll
                    for (int k = 0; k < 8; k++) q[k] = *(const uint2 *)(tb + k * 16);
                    int acc[4];
#pragma unroll
                    for (int m = 0; m < 4; m++) {
                        const uint32_t cv = k_luma_taps[a.admvp][fy][m];
                        const uint32_t p0 = __builtin_amdgcn_perm(q[2 * m + 1].x, q[2 * m].x, 0x05040100u), p1 = __builtin_amdgcn_perm(q[2 * m + 1].x, q[2 * m].x, 0x07060302u);
                        const uint32_t p2 = __builtin_amdgcn_perm(q[2 * m + 1].y, q[2 * m].y, 0x05040100u), p3 = __builtin_amdgcn_perm(q[2 * m + 1].y, q[2 * m].y, 0x07060302u);
                        if (m == 0) { acc[0] = dot2z(cv, p0); acc[1] = dot2z(cv, p1); acc[2] = dot2z(cv, p2); acc[3] = dot2z(cv, p3); }
                        else { acc[0] = dot2(cv, p0, acc[0]); acc[1] = dot2(cv, p1, acc[1]); acc[2] = dot2(cv, p2, acc[2]); acc[3] = dot2(cv, p3, acc[3]); }
                    }
#pragma unroll
                    for (int e = 0; e < 4; e++) acc[e] = fx ? clampi((acc[e] + off2) >> sh2, 0, maxl) : clampi(acc[e] >> 6, 0, maxl);
                    v0 = pack2(acc[0], acc[1]); v1 = pack2(acc[2], acc[3]);
                } else {
                    const uint2 q = *(const uint2 *)(tb + 3 * 16);
                    v0 = q.x; v1 = q.y;
                }
                pl[0] = l ? avg2(pl[0], v0) : v0; pl[1] = l ? avg2(pl[1], v1) : v1;
            }
            dm_sync();
        }
        // ---- chroma, both planes: the same with the 4-tap tables at the thirty-second-sample phase ----
        {
            const int fx = gxy[l][0] & 31, fy = gxy[l][1] & 31, dcx = dc[l][0], dcy = dc[l][1];
            const uint32_t c0 = k_chroma_taps[a.admvp][fx][0], c1 = k_chroma_taps[a.admvp][fx][1];
            for (int i = lane; i < 2 * RC * GC; i += 64) {
                const int p = i >= RC * GC, j = i - p * RC * GC, rr = j / GC, g = j - rr * GC, src = clampi(rr - 2, 0, WCH - 1);
                const int16_t *row = Wc + p * WCH * DM_CWS + src * DM_CWS;
                uint2 o;
                if (fx) {
                    const int base = 8 + dcx + 4 * g;
                    const uint32_t *pp = (const uint32_t *)(row + (base & ~1));
                    const uint32_t D0 = pp[0], D1 = pp[1], D2 = pp[2], D3 = pp[3];
                    const uint32_t Q0 = hi_lo(D1, D0), Q1 = hi_lo(D2, D1), Q2 = hi_lo(D3, D2);
                    int t[4];
                    if (base & 1) { t[0] = dot2(c1, Q1, dot2z(c0, Q0)); t[1] = dot2(c1, D2, dot2z(c0, D1)); t[2] = dot2(c1, Q2, dot2z(c0, Q1)); t[3] = dot2(c1, D3, dot2z(c0, D2)); }
                    else          { t[0] = dot2(c1, D1, dot2z(c0, D0)); t[1] = dot2(c1, Q1, dot2z(c0, Q0)); t[2] = dot2(c1, D2, dot2z(c0, D1)); t[3] = dot2(c1, Q2, dot2z(c0, Q1)); }
#pragma unroll
                    for (int q = 0; q < 4; q++) t[q] = fy ? t[q] >> shc1 : clampi(t[q] >> 6, 0, maxc);
                    o = make_uint2(pack2(t[0], t[1]), pack2(t[2], t[3]));
                } else {
                    const int cc0 = 8 + 1 + dcx + 4 * g;
                    const uint32_t *pp = (const uint32_t *)(row + (cc0 & ~1));
                    const uint32_t D0 = pp[0], D1 = pp[1], D2 = pp[2];
                    o = (cc0 & 1) ? make_uint2(hi_lo(D1, D0), hi_lo(D2, D1)) : make_uint2(D0, D1);
                }
                *(uint2 *)(T + p * RC * 8 + rr * 8 + 4 * g) = o;
            }
            dm_sync();
            if (lane < NC4) {
                const int16_t *tb = T + cpl * RC * 8 + (2 + dcy + cr) * 8 + 4 * cg;     // the first of the four rows under the taps (the row itself: + 1)
                uint32_t v0, v1;
                if (fy) {
                    uint2 q[4];
#pragma unroll
                    for (int k = 0; k < 4; k++) q[k] = *(const uint2 *)(tb + k * 8);
                    int acc[4];
#pragma unroll
                    for (int m = 0; m < 2; m++) {
                        const uint32_t cv = k_chroma_taps[a.admvp][fy][m];
                        const uint32_t p0 = __builtin_amdgcn_perm(q[2 * m + 1].x, q[2 * m].x, 0x05040100u), p1 = __builtin_amdgcn_perm(q[2 * m + 1].x, q[2 * m].x, 0x07060302u);
                        const uint32_t p2 = __builtin_amdgcn_perm(q[2 * m + 1].y, q[2 * m].y, 0x05040100u), p3 = __builtin_amdgcn_perm(q[2 * m + 1].y, q[2 * m].y, 0x07060302u);
                        if (m == 0) { acc[0] = dot2z(cv, p0); acc[1] = dot2z(cv, p1); acc[2] = dot2z(cv, p2); acc[3] = dot2z(cv, p3); }
                        else { acc[0] = dot2(cv, p0, acc[0]); acc[1] = dot2(cv, p1, acc[1]); acc[2] = dot2(cv, p2, acc[2]); acc[3] = dot2(cv, p3, acc[3]); }
                    }
#pragma unroll
                    for (int e = 0; e < 4; e++) acc[e] = fx ? clampi((acc[e] + offc2) >> shc2, 0, maxc) : clampi(acc[e] >> 6, 0, maxc);
                    v0 = pack2(acc[0], acc[1]); v1 = pack2(acc[2], acc[3]);
                } else {
                    const uint2 q = *(const uint2 *)(tb + 8);
                    v0 = q.x; v1 = q.y;
                }
                pc[0] = l ? avg2(pc[0], v0) : v0; pc[1] = l ? avg2(pc[1], v1) : v1;
            }
            dm_sync();
        }
    }

    // ---- residual add + clip (xevd_recon.c:35-71; the LUMA bit depth clips all three components) and the stores ----
    int tu_x = 0, tu_y = 0, tu_w = u.cw, tu_h = u.chh;
    if (u.ats) {      // ATS-inter: the coded TU is one half / quarter of the CU at its start or end (xevdm_util.c:3585-3634)
        const int idx = u.ats & 15, pos = u.ats >> 4;
        if (idx == 2 || idx == 4) { tu_h = u.chh >> (idx == 4 ? 2 : 1); tu_y = pos ? u.chh - tu_h : 0; }
        else                      { tu_w = u.cw >> (idx == 3 ? 2 : 1);  tu_x = pos ? u.cw - tu_w : 0; }
    }
    const int cwc = tu_w >> 1, cbf = u.cbf;
    const uint32_t off_u = u.coef_off + ((cbf & 1) ? tu_w * tu_h : 0), off_v = off_u + ((cbf & 2) ? cwc * (tu_h >> 1) : 0);
    if (lane < NL4) {
        const int x = u.px + lc, y = u.py + lr, lx = x - u.cu_x - tu_x, ly = y - u.cu_y - tu_y;      // TU borders are multiples of four samples: the lane's four are inside or outside together
        if ((cbf & 1) && (uint32_t)lx < (uint32_t)tu_w && (uint32_t)ly < (uint32_t)tu_h) {
            const uint2 r = *(const uint2 *)(a.resid + u.coef_off + ly * tu_w + lx);
            pl[0] = recon2(pl[0], r.x, maxl); pl[1] = recon2(pl[1], r.y, maxl);
        }
        *(uint2 *)(a.cur_y + y * a.s_l + x) = make_uint2(pl[0], pl[1]);
    }
    if (lane < NC4) {
        const int xc = (u.px >> 1) + 4 * cg, yc = (u.py >> 1) + cr, lyc = yc - ((u.cu_y + tu_y) >> 1);
        const uint32_t off = cpl ? off_v : off_u;
        const bool coded = (cbf & (cpl ? 4 : 2)) != 0 && (uint32_t)lyc < (uint32_t)(tu_h >> 1);
        int16_t *dst = (cpl ? a.cur_v : a.cur_u) + yc * a.s_c + xc;
#pragma unroll
        for (int d = 0; d < 2; d++) {                                 // chroma TU borders are multiples of two samples
            const int lxc = xc + 2 * d - ((u.cu_x + tu_x) >> 1);
            if (coded && (uint32_t)lxc < (uint32_t)cwc) pc[d] = recon2(pc[d], *(const uint32_t *)(a.resid + off + lyc * cwc + lxc), maxl);
            *(uint32_t *)(dst + 2 * d) = pc[d];
        }
    }
}

// One sub-block of DX x DY luma samples (8 or 16 each way: compile-time, so that every index split below is a shift or a multiplication by a constant).
// Every global read of a phase is ONE sweep of 16-byte loads into the wave's LDS (rows of three / two chunks); the filters then run out of LDS.
template <int DX, int DY>
__device__ __forceinline__ void dmvr_block(const DmvrArgs &a, const uint4 r0, const uint4 r1, int isx, int isy, int16_t *BL, int16_t *W, int16_t *T, int16_t *out_mv, int lane)
{
    DmvrCu u;
    dmvr_unpack(a, r0, r1, isx, isy, u);
    constexpr int CH = DX == 16 ? 3 : 2;                       // 16-byte chunks per staged luma row
    const int bd = a.bd_l, maxl = (1 << a.bd_l) - 1;
    const int sh1 = min(4, bd - 8), sh2 = max(8, 20 - bd), off2 = 1 << (sh2 - 1);

    // ---- search windows: (DY + 5) rows x (DX + 5) samples of both lists at the (clipped) starting vector, then the bilinear blocks ----
    int bfx[2], bfy[2];
    uint4 sw[2] = { make_uint4(0, 0, 0, 0), make_uint4(0, 0, 0, 0) };
#pragma unroll
    for (int l = 0; l < 2; l++) {
        const int gx = ((u.cu_x << 2) + u.st[l][0] - 8) << 2, gy = ((u.cu_y << 2) + u.st[l][1] - 8) << 2;      // sixteenth samples of the CU block's corner
        bfx[l] = gx & 15; bfy[l] = gy & 15;
        const gs16 src = (gs16)u.ry[l] + ((gy >> 4) + (u.py - u.cu_y)) * a.s_l + (gx >> 4) + (u.px - u.cu_x);
        static_assert((DY + 5) * CH <= 64, "one chunk of a search window per lane");
        if (lane < (DY + 5) * CH) sw[l] = gload16(src + (lane / CH) * a.s_l + 8 * (lane % CH));      // both lists' loads go out before either is stored
    }
#pragma unroll
    for (int l = 0; l < 2; l++)
        if (lane < (DY + 5) * CH) *(uint4 *)(W + l * 21 * DM_RS + (lane / CH) * DM_RS + 8 * (lane % CH)) = sw[l];
    dm_sync();
#pragma unroll
    for (int l = 0; l < 2; l++) {
        // four neighbouring outputs per lane from three dwords of two window rows; taps { 64 - 4 f, 4 f } as one packed pair per direction
        const int fx = bfx[l], fy = bfy[l];
        const uint32_t cx = pack2(64 - 4 * fx, 4 * fx), cy = pack2(64 - 4 * fy, 4 * fy);
        const int16_t *R = W + l * 21 * DM_RS;
        constexpr int GB = (DX + 4) >> 2;
        for (int i = lane; i < (DY + 4) * GB; i += 64) {
            const int r = i / GB, c = (i - r * GB) << 2;
            const uint32_t *r0 = (const uint32_t *)(R + r * DM_RS + c), *r1 = (const uint32_t *)(R + (r + 1) * DM_RS + c);
            const uint32_t a0 = r0[0], a1 = r0[1], a2 = r0[2], b0 = r1[0], b1 = r1[1], b2 = r1[2];
            const uint32_t PA[4] = { a0, hi_lo(a1, a0), a1, hi_lo(a2, a1) }, PB[4] = { b0, hi_lo(b1, b0), b1, hi_lo(b2, b1) };      // (s_c+e, s_c+e+1) of both rows
            int v[4];
#pragma unroll
            for (int e = 0; e < 4; e++) {
                if (!fx && !fy) v[e] = (int)(PA[e] & 0xFFFFu);
                else if (fx && !fy) v[e] = clampi(dot2z(cx, PA[e]) >> 6, 0, maxl);
                else if (!fx) v[e] = clampi(dot2z(cy, __builtin_amdgcn_perm(PB[e], PA[e], 0x05040100u)) >> 6, 0, maxl);      // (A, C): the sample and the one below it
                else v[e] = clampi((dot2a(cy, pack2(dot2z(cx, PA[e]) >> sh1, dot2z(cx, PB[e]) >> sh1), off2)) >> sh2, 0, maxl);
            }
            *(uint2 *)(BL + l * DM_BL * DM_BL + r * DM_BL + c) = make_uint2(pack2(v[0], v[1]), pack2(v[2], v[3]));
        }
    }
    dm_sync();

    // ---- the search: list 0 at +offset against list 1 at -offset ----
    // a lane takes four neighbouring samples of a row (lanes below DX * DY / 4): two or three aligned dwords per list - the offsets of both lists have the parity of
    // ox -, |a - b| of a sample pair + accumulator is one v_sad_u16
    const int sl_r = (lane * 4) / DX, sl_c = (lane * 4) - sl_r * DX;
    auto cost_at = [&](int ox, int oy) -> int {
        int s = 0;
        if (lane < ((DX * DY) >> 2)) {
            const int16_t *p0 = BL + (2 + oy + sl_r) * DM_BL + 2 + ox + sl_c, *p1 = BL + DM_BL * DM_BL + (2 - oy + sl_r) * DM_BL + 2 - ox + sl_c;
            uint32_t A0, A1, B0, B1;
            if (ox & 1) {
                const uint32_t *q0 = (const uint32_t *)(p0 - 1), *q1 = (const uint32_t *)(p1 - 1);
                const uint32_t a0 = q0[0], a1 = q0[1], a2 = q0[2], b0 = q1[0], b1 = q1[1], b2 = q1[2];
                A0 = hi_lo(a1, a0); A1 = hi_lo(a2, a1); B0 = hi_lo(b1, b0); B1 = hi_lo(b2, b1);
            } else {
                const uint32_t *q0 = (const uint32_t *)p0, *q1 = (const uint32_t *)p1;
                A0 = q0[0]; A1 = q0[1]; B0 = q1[0]; B1 = q1[1];
            }
            s = (int)__builtin_amdgcn_sad_u16(A0, B0, __builtin_amdgcn_sad_u16(A1, B1, 0u));
        }
        return wave_sum(s);
    };
    int tot[2] = { 0, 0 }, not_zero = 1, min_cost = 0;
    int cB = 0x7FFFFFFF, cT = 0x7FFFFFFF, cR = 0x7FFFFFFF, cL = 0x7FFFFFFF, cC = 0x7FFFFFFF;      // bottom, top, right, left, centre of the last round
    for (int i = 0; i < 2; i++) {
        int d[2] = { 0, 0 };
        if (i == 0) min_cost = cost_at(0, 0);
        if ((i > 0 && min_cost == 0) || (i == 0 && min_cost < DX * DY)) { not_zero = 0; break; }
        cC = min_cost;
        // the four SADs of the cross are independent of each other: evaluated together (their loads and reductions interleave), compared in the reference's order
        cB = cost_at(tot[0], tot[1] + 1); cT = cost_at(tot[0], tot[1] - 1); cR = cost_at(tot[0] + 1, tot[1]); cL = cost_at(tot[0] - 1, tot[1]);
        if (cB < min_cost) { min_cost = cB; d[0] = 0; d[1] = 1; }
        if (cT < min_cost) { min_cost = cT; d[0] = 0; d[1] = -1; }
        if (cR < min_cost) { min_cost = cR; d[0] = 1; d[1] = 0; }
        if (cL < min_cost) { min_cost = cL; d[0] = -1; d[1] = 0; }
        const int dgx = cR <= cL ? 1 : -1, dgy = cB <= cT ? 1 : -1;
        const int cD = cost_at(tot[0] + dgx, tot[1] + dgy); if (cD < min_cost) { min_cost = cD; d[0] = dgx; d[1] = dgy; }
        if (d[0] == 0 && d[1] == 0) break;
        tot[0] += d[0]; tot[1] += d[1];
        if (i == 0) { cB = cT = cR = cL = cC = 0x7FFFFFFF; }      // the costs of a round are only meaningful around ITS centre
    }
    tot[0] <<= 4; tot[1] <<= 4;
    if (not_zero && min_cost == cC) {
        const int sb[5] = { cC, cL, cT, cR, cB };
#pragma unroll
        for (int ax = 0; ax < 2; ax++) {
            const long long nu = (long long)((sb[1 + ax] - sb[3 + ax]) << 4), de = (long long)(sb[1 + ax] + sb[3 + ax] - (sb[0] << 1));
            if (de != 0) tot[ax] += (sb[1 + ax] != sb[0] && sb[3 + ax] != sb[0]) ? dm_div_q7(nu, de) : (sb[1 + ax] == sb[0] ? -8 : 8);
        }
    }
    int r16[2][2];
#pragma unroll
    for (int l = 0; l < 2; l++) { r16[l][0] = (u.st[l][0] << 2) + (l ? -tot[0] : tot[0]); r16[l][1] = (u.st[l][1] << 2) + (l ? -tot[1] : tot[1]); }
    if (lane < 4) out_mv[lane] = (int16_t)((lane == 0 ? r16[0][0] : lane == 1 ? r16[0][1] : lane == 2 ? r16[1][0] : r16[1][1]) >> 2);      // selects: an array indexed by the lane would live in scratch
    // The deblocking filter's view of a refined CU: ADDB is handed the UNREFINED vectors (map_unrefined_mv, xevdm.c:2009-2041 - what k_inter wrote),
    // the Main library's copy of the baseline filter reads ctx->map_mv, which holds the refined ones (xevdm_df.c:118,209; xevdm_util.c:4327-4332)
    if (a.refined_to_map && lane < (DX >> 2) * (DY >> 2)) {
        const int uu = lane % (DX >> 2), vv = lane / (DX >> 2);
        ScuRec *m = a.maps + ((u.py >> 2) + vv) * a.w_scu + (u.px >> 2) + uu;
        *(uint2 *)&m->mv[0][0] = make_uint2((uint32_t)(uint16_t)(r16[0][0] >> 2) | ((uint32_t)(uint16_t)(r16[0][1] >> 2) << 16),
                                            (uint32_t)(uint16_t)(r16[1][0] >> 2) | ((uint32_t)(uint16_t)(r16[1][1] >> 2) << 16));
    }

    // ---- the refined prediction: whole-sample offset of the refined vector from the starting vector's window (per list), then the packed form when the
    //      offsets lie inside the padding it provides - always, except for vectors clipped at the picture's far border ----
    int dl[2][2], dc[2][2], gxy[2][2];
    bool packed = true;
#pragma unroll
    for (int l = 0; l < 2; l++) {
        // clip of the refined vector at the sub-block (mv_clip_only_one_ref_dmvr :939-980)
        int tq[2] = { (int)(int16_t)(r16[l][0] >> 2), (int)(int16_t)(r16[l][1] >> 2) }, mc[2] = { tq[0], tq[1] }, clip = 0;
        const int min_c = -(128 << 2), max_x = (a.pic_w - 1 + 128) << 2, max_y = (a.pic_h - 1 + 128) << 2;
        if ((u.px << 2) + tq[0] < min_c) { clip = 1; mc[0] = min_c - (u.px << 2); }
        if ((u.py << 2) + tq[1] < min_c) { clip = 1; mc[1] = min_c - (u.py << 2); }
        if ((u.px << 2) + tq[0] + (DX << 2) - 4 > max_x) { clip = 1; mc[0] = max_x - (u.px << 2) - (DX << 2) + 4; }
        if ((u.py << 2) + tq[1] + (DY << 2) - 4 > max_y) { clip = 1; mc[1] = max_y - (u.py << 2) - (DY << 2) + 4; }
        mc[0] = (int)(int16_t)mc[0]; mc[1] = (int)(int16_t)mc[1];
        gxy[l][0] = (u.px << 4) + (clip ? mc[0] << 2 : r16[l][0]); gxy[l][1] = (u.py << 4) + (clip ? mc[1] << 2 : r16[l][1]);
        dl[l][0] = (clip ? mc[0] >> 2 : r16[l][0] >> 4) - (u.st[l][0] >> 2); dl[l][1] = (clip ? mc[1] >> 2 : r16[l][1] >> 4) - (u.st[l][1] >> 2);
        dc[l][0] = (clip ? mc[0] >> 3 : r16[l][0] >> 5) - (u.st[l][0] >> 3); dc[l][1] = (clip ? mc[1] >> 3 : r16[l][1] >> 5) - (u.st[l][1] >> 3);
        packed = packed && abs(dl[l][0]) <= 3 && abs(dl[l][1]) <= 3 && abs(dc[l][0]) <= 2 && abs(dc[l][1]) <= 2;
    }
    if (packed) dmvr_predict_packed<DX, DY>(a, u, r16, dl, dc, gxy, W, T, lane);
    else        dmvr_predict_scalar<DX, DY>(a, u, r16, W, T, lane);
}

__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(8, 8))) void k_dmvr(const DmvrArgs a)      // (64 VGPRs: eight waves per SIMD - the LDS footprint allows as many)
{
    __shared__ __attribute__((aligned(16))) int16_t s_bl[4][2 * DM_BL * DM_BL];
    __shared__ __attribute__((aligned(16))) int16_t s_win[4][DM_STAGE];
    const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int item = __builtin_amdgcn_readfirstlane(blockIdx.x * 4 + wv);
    if (item >= a.n_items) return;
    const DmvrItem it = a.items[item];
    uint4 r0 = ((const uint4 *)&a.cus[it.cu])[0], r1 = ((const uint4 *)&a.cus[it.cu])[1];
    // one item per wave: everything below is wave-uniform - in scalar registers (and the reference table is indexed in the kernel arguments, not in a scratch copy)
    r0.x = __builtin_amdgcn_readfirstlane(r0.x); r0.y = __builtin_amdgcn_readfirstlane(r0.y); r0.z = __builtin_amdgcn_readfirstlane(r0.z); r0.w = __builtin_amdgcn_readfirstlane(r0.w);
    r1.x = __builtin_amdgcn_readfirstlane(r1.x); r1.y = __builtin_amdgcn_readfirstlane(r1.y); r1.w = __builtin_amdgcn_readfirstlane(r1.w);
    const int isx = __builtin_amdgcn_readfirstlane((int)it.sx), isy = __builtin_amdgcn_readfirstlane((int)it.sy);
    const int refi0 = (int)(int8_t)(r0.z & 0xFF), refi1 = (int)(int8_t)((r0.z >> 8) & 0xFF), cw = 1 << (r0.y & 0xFF), chh = 1 << ((r0.y >> 8) & 0xFF);
    int16_t *out_mv = a.out_mv + (size_t)item * 4;
    if (!dmvr_applies(a.cur_poc, a.refp[refi0][0].poc, a.refp[refi1][1].poc)) {
        // the CU was predicted by k_inter; the vector kept for temporal prediction is its own
        if (lane < 4) out_mv[lane] = (int16_t)((lane & 2 ? r1.y : r1.x) >> ((lane & 1) * 16));
        return;
    }
    int16_t *BL = s_bl[wv], *W = s_win[wv], *T = s_bl[wv];      // the intermediate rows of the prediction lie over the bilinear blocks (the search is over by then)
    static_assert(DM_TMP <= 2 * DM_BL * DM_BL, "the intermediate rows fit into the bilinear blocks");
    const bool w16 = cw >= 16, h16 = chh >= 16;
    if (w16) { if (h16) dmvr_block<16, 16>(a, r0, r1, isx, isy, BL, W, T, out_mv, lane); else dmvr_block<16, 8>(a, r0, r1, isx, isy, BL, W, T, out_mv, lane); }
    else     { if (h16) dmvr_block<8, 16>(a, r0, r1, isx, isy, BL, W, T, out_mv, lane);  else dmvr_block<8, 8>(a, r0, r1, isx, isy, BL, W, T, out_mv, lane); }
}

void launch_dmvr(xgpu_ctx *c, const DmvrArgs &a, hipStream_t st)
{
    if (a.n_items > 0) hipLaunchKernelGGL(k_dmvr, dim3((a.n_items + 3) / 4), dim3(256), 0, st, a);
}
