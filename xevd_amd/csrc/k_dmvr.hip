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
#define DM_WL   23            // luma window: (16 + 7)^2
#define DM_WC   11            // chroma window: (8 + 3)^2 per plane

__device__ __forceinline__ void dm_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
__device__ __forceinline__ int wave_sum(int v)
{
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m, 64);
    return v;
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

__global__ __launch_bounds__(256) void k_dmvr(const DmvrArgs a)
{
    __shared__ int16_t s_bl[4][2][DM_BL * DM_BL];
    __shared__ int16_t s_win[4][DM_WL * DM_WL + 2 * DM_WC * DM_WC];
    __shared__ int16_t s_tmp[4][DM_WL * 16];
    const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int item = blockIdx.x * 4 + wv;
    if (item >= a.n_items) return;
    const DmvrItem it = a.items[item];
    const uint4 r0 = ((const uint4 *)&a.cus[it.cu])[0], r1 = ((const uint4 *)&a.cus[it.cu])[1];
    const int cu_x = r0.x & 0xFFFF, cu_y = r0.x >> 16, cw = 1 << (r0.y & 0xFF), chh = 1 << ((r0.y >> 8) & 0xFF), cbf = r0.y >> 24;
    const int refi[2] = { (int)(int8_t)(r0.z & 0xFF), (int)(int8_t)((r0.z >> 8) & 0xFF) };
    const uint32_t coef_off = r0.w;
    const int mvu[2][2] = { { (int)(int16_t)(r1.x & 0xFFFF), (int)(int16_t)(r1.x >> 16) }, { (int)(int16_t)(r1.y & 0xFFFF), (int)(int16_t)(r1.y >> 16) } };
    const int dx = min(cw, 16), dy = min(chh, 16), px = cu_x + it.sx * 4, py = cu_y + it.sy * 4;
    // starting vectors: the CU's, clipped like xevd_mv_clip (mv_clip, xevdm_mc.c:1831-1858)
    int st[2][2];
    {
        const int min_c = -(128 << 2), max_x = (a.pic_w - 1 + 128) << 2, max_y = (a.pic_h - 1 + 128) << 2;
#pragma unroll
        for (int l = 0; l < 2; l++) {
            int mx = mvu[l][0], my = mvu[l][1];
            if ((cu_x << 2) + mvu[l][0] < min_c) mx = min_c - (cu_x << 2);
            if ((cu_y << 2) + mvu[l][1] < min_c) my = min_c - (cu_y << 2);
            if ((cu_x << 2) + mvu[l][0] + (cw << 2) - 4 > max_x) mx = max_x - (cu_x << 2) - (cw << 2) + 4;
            if ((cu_y << 2) + mvu[l][1] + (chh << 2) - 4 > max_y) my = max_y - (cu_y << 2) - (chh << 2) + 4;
            st[l][0] = mx; st[l][1] = my;
        }
    }
    const RefEntry re[2] = { a.refp[refi[0]][0], a.refp[refi[1]][1] };
    int16_t *out_mv = a.out_mv + (size_t)item * 4;
    if (!dmvr_applies(a.cur_poc, re[0].poc, re[1].poc)) {
        // the CU was predicted by k_inter; the vector kept for temporal prediction is its own
        if (lane < 4) out_mv[lane] = (int16_t)mvu[lane >> 1][lane & 1];
        return;
    }
    const int bd = a.bd_l, maxl = (1 << a.bd_l) - 1, maxc = (1 << a.bd_c) - 1;
    const int sh1 = min(4, bd - 8), sh2 = max(8, 20 - bd), off2 = 1 << (sh2 - 1);

    // ---- bilinear blocks of the sub-block + 2 samples around it, both lists ----
#pragma unroll
    for (int l = 0; l < 2; l++) {
        const int gx = ((cu_x << 2) + st[l][0] - 8) << 2, gy = ((cu_y << 2) + st[l][1] - 8) << 2;      // sixteenth samples of the CU block's corner
        const int fx = gx & 15, fy = gy & 15, c0 = 64 - 4 * fx, c1 = 4 * fx, d0 = 64 - 4 * fy, d1 = 4 * fy;
        const gs16 src = (gs16)re[l].y + ((gy >> 4) + it.sy * 4) * a.s_l + (gx >> 4) + it.sx * 4;
        for (int i = lane; i < (dy + 4) * (dx + 4); i += 64) {
            const int r = i / (dx + 4), c = i - r * (dx + 4);
            const gs16 p = src + r * a.s_l + c;
            const int A = p[0], B = p[1], Cc = p[a.s_l], D = p[a.s_l + 1];
            int v;
            if (!fx && !fy) v = A;
            else if (fx && !fy) v = clampi((c0 * A + c1 * B) >> 6, 0, maxl);
            else if (!fx) v = clampi((d0 * A + d1 * Cc) >> 6, 0, maxl);
            else {
                const int t0 = (int)(int16_t)((c0 * A + c1 * B) >> sh1), t1 = (int)(int16_t)((c0 * Cc + c1 * D) >> sh1);
                v = clampi((d0 * t0 + d1 * t1 + off2) >> sh2, 0, maxl);
            }
            s_bl[wv][l][r * DM_BL + c] = (int16_t)v;
        }
    }
    dm_sync();

    // ---- the search: list 0 at +offset against list 1 at -offset ----
    auto cost_at = [&](int ox, int oy) -> int {
        int s = 0;
        for (int i = lane; i < dx * dy; i += 64) {
            const int r = i / dx, c = i - r * dx;
            s += abs((int)s_bl[wv][0][(2 + oy + r) * DM_BL + 2 + ox + c] - (int)s_bl[wv][1][(2 - oy + r) * DM_BL + 2 - ox + c]);
        }
        return wave_sum(s);
    };
    enum { BOTTOM = 0, TOP, RIGHT, LEFT, DIAG, CENTER = 8 };
    int tot[2] = { 0, 0 }, not_zero = 1, min_cost = 0, cost[9];
    for (int k = 0; k < 9; k++) cost[k] = 0x7FFFFFFF;
    for (int i = 0; i < 2; i++) {
        int ox[5] = { 0, 0, 1, -1, 0 }, oy[5] = { 1, -1, 0, 0, 0 }, d[2] = { 0, 0 };
        for (int k = 0; k < 9; k++) cost[k] = 0x7FFFFFFF;
        if (i == 0) min_cost = cost_at(0, 0);
        if ((i > 0 && min_cost == 0) || (i == 0 && min_cost < dx * dy)) { not_zero = 0; break; }
        cost[CENTER] = min_cost;
        for (int idx = BOTTOM; idx <= DIAG; idx++) {
            const int c = cost_at(tot[0] + ox[idx], tot[1] + oy[idx]);
            cost[idx] = c;
            if (idx == LEFT) { ox[DIAG] = cost[RIGHT] <= cost[LEFT] ? 1 : -1; oy[DIAG] = cost[BOTTOM] <= cost[TOP] ? 1 : -1; }
            if (c < min_cost) { min_cost = c; d[0] = ox[idx]; d[1] = oy[idx]; }
        }
        if (d[0] == 0 && d[1] == 0) break;
        tot[0] += d[0]; tot[1] += d[1];
    }
    tot[0] <<= 4; tot[1] <<= 4;
    if (not_zero && min_cost == cost[CENTER]) {
        const int sb[5] = { cost[CENTER], cost[LEFT], cost[TOP], cost[RIGHT], cost[BOTTOM] };
#pragma unroll
        for (int ax = 0; ax < 2; ax++) {
            const long long nu = (long long)((sb[1 + ax] - sb[3 + ax]) << 4), de = (long long)(sb[1 + ax] + sb[3 + ax] - (sb[0] << 1));
            if (de != 0) tot[ax] += (sb[1 + ax] != sb[0] && sb[3 + ax] != sb[0]) ? dm_div_q7(nu, de) : (sb[1 + ax] == sb[0] ? -8 : 8);
        }
    }
    int r16[2][2];
#pragma unroll
    for (int l = 0; l < 2; l++) { r16[l][0] = (st[l][0] << 2) + (l ? -tot[0] : tot[0]); r16[l][1] = (st[l][1] << 2) + (l ? -tot[1] : tot[1]); }
    if (lane < 4) out_mv[lane] = (int16_t)(r16[lane >> 1][lane & 1] >> 2);
    // The deblocking filter's view of a refined CU: ADDB is handed the UNREFINED vectors (map_unrefined_mv, xevdm.c:2009-2041 - what k_inter wrote),
    // the Main library's copy of the baseline filter reads ctx->map_mv, which holds the refined ones (xevdm_df.c:118,209; xevdm_util.c:4327-4332)
    if (a.refined_to_map && lane < (dx >> 2) * (dy >> 2)) {
        const int u = lane % (dx >> 2), v = lane / (dx >> 2);
        ScuRec *m = a.maps + ((py >> 2) + v) * a.w_scu + (px >> 2) + u;
        *(uint2 *)&m->mv[0][0] = make_uint2((uint32_t)(uint16_t)(r16[0][0] >> 2) | ((uint32_t)(uint16_t)(r16[0][1] >> 2) << 16),
                                            (uint32_t)(uint16_t)(r16[1][0] >> 2) | ((uint32_t)(uint16_t)(r16[1][1] >> 2) << 16));
    }

    // ---- the refined prediction: lane = 4 luma samples of a row (lanes below dx*dy/4) and one chroma sample per plane (lanes below dx*dy/4) ----
    const int nl4 = (dx * dy) >> 2, nc = (dx >> 1) * (dy >> 1);
    const int lr = (lane * 4) / dx, lc = (lane * 4) - lr * dx;                    // luma row / first column inside the sub-block
    const int cr = lane / (dx >> 1), cc = lane - cr * (dx >> 1);                  // chroma row / column
    int accl[4] = { 0, 0, 0, 0 }, accu = 0, accv = 0;
    int16_t *W = s_win[wv], *T = s_tmp[wv];
#pragma unroll
    for (int l = 0; l < 2; l++) {
        // clip of the refined vector at the sub-block (mv_clip_only_one_ref_dmvr :939-980)
        int tq[2] = { (int)(int16_t)(r16[l][0] >> 2), (int)(int16_t)(r16[l][1] >> 2) }, mc[2] = { tq[0], tq[1] }, clip = 0;
        {
            const int min_c = -(128 << 2), max_x = (a.pic_w - 1 + 128) << 2, max_y = (a.pic_h - 1 + 128) << 2;
            if ((px << 2) + tq[0] < min_c) { clip = 1; mc[0] = min_c - (px << 2); }
            if ((py << 2) + tq[1] < min_c) { clip = 1; mc[1] = min_c - (py << 2); }
            if ((px << 2) + tq[0] + (dx << 2) - 4 > max_x) { clip = 1; mc[0] = max_x - (px << 2) - (dx << 2) + 4; }
            if ((py << 2) + tq[1] + (dy << 2) - 4 > max_y) { clip = 1; mc[1] = max_y - (py << 2) - (dy << 2) + 4; }
            mc[0] = (int)(int16_t)mc[0]; mc[1] = (int)(int16_t)mc[1];
        }
        const int gx = (px << 4) + (clip ? mc[0] << 2 : r16[l][0]), gy = (py << 4) + (clip ? mc[1] << 2 : r16[l][1]);
        const int dlx = (clip ? mc[0] >> 2 : r16[l][0] >> 4) - (st[l][0] >> 2), dly = (clip ? mc[1] >> 2 : r16[l][1] >> 4) - (st[l][1] >> 2);
        const int dcx = (clip ? mc[0] >> 3 : r16[l][0] >> 5) - (st[l][0] >> 3), dcy = (clip ? mc[1] >> 3 : r16[l][1] >> 5) - (st[l][1] >> 3);
        // windows at the STARTING vector: luma (dx + 7) x (dy + 7) from 3 samples up-left, chroma (dx/2 + 3) x (dy/2 + 3) from 1 sample up-left
        const int q16x = ((px << 2) + st[l][0]) << 2, q16y = ((py << 2) + st[l][1]) << 2;
        {
            const gs16 sy_ = (gs16)re[l].y + ((q16y >> 4) - 3) * a.s_l + (q16x >> 4) - 3;
            for (int i = lane; i < (dy + 7) * (dx + 7); i += 64) { const int r = i / (dx + 7), c = i - r * (dx + 7); W[r * DM_WL + c] = sy_[r * a.s_l + c]; }
            const int co = ((q16y >> 5) - 1) * a.s_c + (q16x >> 5) - 1, wcw = (dx >> 1) + 3, wch = (dy >> 1) + 3;
            const gs16 su_ = (gs16)re[l].u + co, sv_ = (gs16)re[l].v + co;
            for (int i = lane; i < 2 * wch * wcw; i += 64) {
                const int pl = i >= wch * wcw, j = i - pl * wch * wcw, r = j / wcw, c = j - r * wcw;
                W[DM_WL * DM_WL + pl * DM_WC * DM_WC + r * DM_WC + c] = (pl ? sv_ : su_)[r * a.s_c + c];
            }
        }
        dm_sync();
        // luma: horizontal pass over every window row (the rows of the padding are copies of window rows), then vertical
        {
            const int fx = gx & 15, fy = gy & 15;
            const uint32_t *th = k_luma_taps[a.admvp][fx], *tv = k_luma_taps[a.admvp][fy];
            for (int i = lane; i < (dy + 7) * dx; i += 64) {
                const int r = i / dx, j = i - r * dx;
                int v;
                if (!fx) v = W[r * DM_WL + clampi(3 + dlx + j, 0, dx + 6)];
                else {
                    int s = 0;
#pragma unroll
                    for (int k = 0; k < 8; k++) s += (int)(int16_t)(th[k >> 1] >> ((k & 1) * 16)) * (int)W[r * DM_WL + clampi(dlx + j + k, 0, dx + 6)];
                    v = fy ? (int)(int16_t)(s >> sh1) : clampi(s >> 6, 0, maxl);
                }
                T[r * 16 + j] = (int16_t)v;
            }
            dm_sync();
            if (lane < nl4) {
#pragma unroll
                for (int e = 0; e < 4; e++) {
                    int v;
                    if (!fy) v = T[clampi(3 + dly + lr, 0, dy + 6) * 16 + lc + e];
                    else {
                        int s = 0;
#pragma unroll
                        for (int k = 0; k < 8; k++) s += (int)(int16_t)(tv[k >> 1] >> ((k & 1) * 16)) * (int)T[clampi(dly + lr + k, 0, dy + 6) * 16 + lc + e];
                        v = fx ? clampi((s + off2) >> sh2, 0, maxl) : clampi(s >> 6, 0, maxl);
                    }
                    accl[e] = l ? (accl[e] + v + 1) >> 1 : v;
                }
            }
            dm_sync();
        }
        // chroma, both planes: the same with the 4-tap tables at the thirty-second-sample phase
        {
            const int fx = gx & 31, fy = gy & 31, wcw = (dx >> 1) + 3, wch = (dy >> 1) + 3, cwd = dx >> 1;
            const uint32_t *th = k_chroma_taps[a.admvp][fx], *tv = k_chroma_taps[a.admvp][fy];
            const int shc1 = min(4, a.bd_c - 8), shc2 = max(8, 20 - a.bd_c), offc2 = 1 << (shc2 - 1);
            for (int i = lane; i < 2 * wch * cwd; i += 64) {
                const int pl = i >= wch * cwd, q = i - pl * wch * cwd, r = q / cwd, j = q - r * cwd;
                const int16_t *Wc = W + DM_WL * DM_WL + pl * DM_WC * DM_WC;
                int v;
                if (!fx) v = Wc[r * DM_WC + clampi(1 + dcx + j, 0, wcw - 1)];
                else {
                    int s = 0;
#pragma unroll
                    for (int k = 0; k < 4; k++) s += (int)(int16_t)(th[k >> 1] >> ((k & 1) * 16)) * (int)Wc[r * DM_WC + clampi(dcx + j + k, 0, wcw - 1)];
                    v = fy ? (int)(int16_t)(s >> shc1) : clampi(s >> 6, 0, maxc);
                }
                T[pl * DM_WC * 8 + r * 8 + j] = (int16_t)v;
            }
            dm_sync();
            if (lane < nc) {
#pragma unroll
                for (int pl = 0; pl < 2; pl++) {
                    const int16_t *Tc = T + pl * DM_WC * 8;
                    int v;
                    if (!fy) v = Tc[clampi(1 + dcy + cr, 0, wch - 1) * 8 + cc];
                    else {
                        int s = 0;
#pragma unroll
                        for (int k = 0; k < 4; k++) s += (int)(int16_t)(tv[k >> 1] >> ((k & 1) * 16)) * (int)Tc[clampi(dcy + cr + k, 0, wch - 1) * 8 + cc];
                        v = fx ? clampi((s + offc2) >> shc2, 0, maxc) : clampi(s >> 6, 0, maxc);
                    }
                    if (pl) accv = l ? (accv + v + 1) >> 1 : v; else accu = l ? (accu + v + 1) >> 1 : v;
                }
            }
            dm_sync();
        }
    }

    // ---- residual add + clip (xevd_recon.c:35-71; the LUMA bit depth clips all three components) and the stores ----
    const int ai = (int)((r1.w >> 8) & 0xFF);
    int tu_x = 0, tu_y = 0, tu_w = cw, tu_h = chh;
    if (ai) {      // ATS-inter: the coded TU is one half / quarter of the CU at its start or end (xevdm_util.c:3585-3634)
        const int idx = ai & 15, pos = ai >> 4;
        if (idx == 2 || idx == 4) { tu_h = chh >> (idx == 4 ? 2 : 1); tu_y = pos ? chh - tu_h : 0; }
        else                      { tu_w = cw >> (idx == 3 ? 2 : 1);  tu_x = pos ? cw - tu_w : 0; }
    }
    const int cwc = tu_w >> 1;
    const uint32_t off_u = coef_off + ((cbf & 1) ? tu_w * tu_h : 0), off_v = off_u + ((cbf & 2) ? cwc * (tu_h >> 1) : 0);
    if (lane < nl4) {
        const int x = px + lc, y = py + lr, lx = x - cu_x - tu_x, ly = y - cu_y - tu_y;
        int o[4];
#pragma unroll
        for (int e = 0; e < 4; e++) {
            o[e] = accl[e];
            if ((cbf & 1) && (uint32_t)(lx + e) < (uint32_t)tu_w && (uint32_t)ly < (uint32_t)tu_h)
                o[e] = clampi((int)(int16_t)(a.resid[coef_off + ly * tu_w + lx + e] + o[e]), 0, maxl);
        }
        *(uint2 *)(a.cur_y + y * a.s_l + x) = make_uint2((uint32_t)(uint16_t)o[0] | ((uint32_t)(uint16_t)o[1] << 16), (uint32_t)(uint16_t)o[2] | ((uint32_t)(uint16_t)o[3] << 16));
    }
    if (lane < nc) {
        const int xc = (px >> 1) + cc, yc = (py >> 1) + cr, lxc = xc - ((cu_x + tu_x) >> 1), lyc = yc - ((cu_y + tu_y) >> 1);
        const bool in_tu = (uint32_t)lxc < (uint32_t)cwc && (uint32_t)lyc < (uint32_t)(tu_h >> 1);
        int u = accu, v = accv;
        if ((cbf & 2) && in_tu) u = clampi((int)(int16_t)(a.resid[off_u + lyc * cwc + lxc] + u), 0, maxl);
        if ((cbf & 4) && in_tu) v = clampi((int)(int16_t)(a.resid[off_v + lyc * cwc + lxc] + v), 0, maxl);
        a.cur_u[yc * a.s_c + xc] = (int16_t)u;
        a.cur_v[yc * a.s_c + xc] = (int16_t)v;
    }
}

void launch_dmvr(xgpu_ctx *c, const DmvrArgs &a)
{
    if (a.n_items > 0) hipLaunchKernelGGL(k_dmvr, dim3((a.n_items + 3) / 4), dim3(256), 0, c->stream, a);
}
