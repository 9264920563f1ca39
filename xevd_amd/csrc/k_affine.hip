// k_affine.hip - affine motion compensation (Main, sps->tool_affine) + residual add + the sub-block vectors of the SCU map.
//
// Replaces, for the affine CUs of a picture, xevdm_affine_mc -> xevd_recon_yuv -> xevdm_set_affine_mvf
// (src_main/xevdm_mc.c:2606-2685 with :2259-2391, :2108-2150, :2393-2604; src_main/xevdm_util.c:1870-2125, :4095-4190).
//
// MI355X mapping: affine CUs are a minority of a picture, so they do not ride in k_inter's per-SCU lanes (k_inter writes their map records
// and leaves the samples alone); the host cuts them into tiles of at most 32x32 luma samples and one 256-thread workgroup predicts one tile:
//   * EIF (a sub-block would be smaller than 8 samples): every sample has its own vector - mv0 + x*dHor + y*dVer at 9 fractional bits,
//     reduced to 1/32 sample and clamped - and is a bilinear fetch of 4 reference samples (a gather: one lane per sample of the
//     (w+2) x (h+2) window, straight from L2/HBM), followed by the [-1 10 -1] enhancement filter along rows and columns through LDS.  The
//     vector of a sample only depends on its position in the CU, so tiles are independent and a tile's one-sample halo is simply computed again.
//   * otherwise: sub-block translation with the 8-tap / 4-tap filters of k_inter (mc_filters.h), one lane per 4x4 SCU.  As the reference has
//     it (:2352-2353 - the sub-block's offset does not enter the vector) every sub-block moves with the vector at the centre of the FIRST one.
//   * the model (deltas, sub-block size, EIF decision and clamp range) is wave-uniform scalar arithmetic recomputed by every workgroup from
//     the control points - cheaper than another host-built table.
// Integer arithmetic and intermediate widths follow the reference exactly (its intermediates are s16 `pel`).
#include "mc_filters.h"

#define AFF_BIT 7                    // MAX_CU_LOG2: the model keeps 2 + 7 fractional bits
#define AFF_TILE 32

struct AffModel { int dh[2], dv[2]; };

__device__ __forceinline__ int aff_round(int v, int shift) { return (v + (1 << (shift - 1)) - (v >= 0)) >> shift; }     // xevdm_mv_rounding_s32
__device__ __forceinline__ int aff_clip18(int v) { return clip3(-(1 << 17), (1 << 17) - 1, v); }

__device__ __forceinline__ AffModel aff_model(const int16_t *mv, int lw, int lh, int vn)          // mv[vertex][x/y]
{
    AffModel m;
    m.dh[0] = ((mv[2] - mv[0]) * (1 << AFF_BIT)) >> lw;
    m.dh[1] = ((mv[3] - mv[1]) * (1 << AFF_BIT)) >> lw;
    if (vn == 3) {
        m.dv[0] = ((mv[4] - mv[0]) * (1 << AFF_BIT)) >> lh;
        m.dv[1] = ((mv[5] - mv[1]) * (1 << AFF_BIT)) >> lh;
    } else { m.dv[0] = -m.dh[1]; m.dv[1] = m.dh[0]; }
    return m;
}

// xevdm_check_eif_applicability_uni (xevdm_util.c:2073-2097): bounding box of a 4x4 sub-block's fetch, fetched-lines limit
__device__ __forceinline__ bool aff_eif_applicable(const AffModel &m, bool &mem_band)
{
    const int P = 2 + AFF_BIT, one = 1 << P;
    const int x1 = 5 * (m.dh[0] + one), x2 = 5 * m.dv[0], x3 = x1 + x2;
    const int y1 = 5 * m.dh[1], y2 = 5 * (m.dv[1] + one), y3 = y1 + y2;
    const int mx = max(max(0, x1), max(x2, x3)), nx = min(min(0, x1), min(x2, x3));
    const int my = max(max(0, y1), max(y2, y3)), ny = min(min(0, y1), min(y2, y3));
    mem_band = (((mx - nx + one - 1) >> P) + 2) * (((my - ny + one - 1) >> P) + 2) <= 72;
    if (m.dv[1] < -one) return false;
    return (max(0, m.dv[1]) + abs(m.dh[1])) * 5 <= one;
}

// xevdm_derive_affine_subblock_size_bi (xevdm_util.c:1870-1945)
__device__ __forceinline__ void aff_subblock(const AffModel m[2], const bool use[2], int lw, int lh, int &sub_w, int &sub_h, bool &mem_band)
{
    sub_w = 1 << lw; sub_h = 1 << lh;
    bool apply = true;
    mem_band = true;
#pragma unroll
    for (int l = 0; l < 2; l++) {
        if (!use[l]) continue;
        const int wx = max(abs(m[l].dh[0]), abs(m[l].dh[1])), wy = max(abs(m[l].dv[0]), abs(m[l].dv[1]));
        const int w = wx > 4 ? 4 : (wx == 0 ? 1 << lw : (wx == 1 ? 32 : (wx == 2 ? 16 : 8)));
        const int h = wy > 4 ? 4 : (wy == 0 ? 1 << lh : (wy == 1 ? 32 : (wy == 2 ? 16 : 8)));
        sub_w = min(sub_w, w); sub_h = min(sub_h, h);
    }
#pragma unroll
    for (int l = 0; l < 2; l++) {
        if (!use[l] || !apply) continue;             // the reference stops at the first list that fails
        bool mb;
        if (!aff_eif_applicable(m[l], mb)) apply = false;
        mem_band = mem_band && mb;
    }
    if (!apply) { sub_w = max(sub_w, 8); sub_h = max(sub_h, 8); }
}

// eif_derive_mv_clip_range (xevdm_mc.c:2108-2150), 1/32 sample
__device__ __forceinline__ void aff_eif_range(int x, int y, int lw, int lh, const AffModel &m, const int mv_scale[2], int pic_w, int pic_h,
                                              bool range_clip, int mx[2], int mn[2])
{
    const int cuw = 1 << lw, cuh = 1 << lh;
    const int max_pic[2] = { (pic_w + 128 - x - cuw - 1) * 32, (pic_h + 128 - y - cuh - 1) * 32 };
    const int min_pic[2] = { (-x - 128) * 32, (-y - 128) * 32 };
#pragma unroll
    for (int c = 0; c < 2; c++) {
        if (!range_clip) { mx[c] = max_pic[c]; mn[c] = min_pic[c]; }
        else {
            const int centre = aff_round(mv_scale[c] + m.dh[c] * (cuw >> 1) + m.dv[c] * (cuh >> 1), 4);
            const int l = (c == 0 ? lw : lh) - 3;
            const int sp = l == 0 ? 128 : l == 1 ? 256 : l == 2 ? 544 : l == 3 ? 1120 : 2272;      // g_aff_mvDevBB2_125
            mn[c] = centre - sp; mx[c] = centre + sp;
            if (mn[c] < min_pic[c]) { mn[c] = min_pic[c]; mx[c] = min(max_pic[c], min_pic[c] + 2 * sp); }
            else if (mx[c] > max_pic[c]) { mx[c] = max_pic[c]; mn[c] = max(min_pic[c], max_pic[c] - 2 * sp); }
        }
        mx[c] = aff_clip18(mx[c]); mn[c] = aff_clip18(mn[c]);
    }
}

__global__ __launch_bounds__(256) void k_affine(const AffineArgs a)
{
    __shared__ int16_t s_bl[(AFF_TILE + 2) * (AFF_TILE + 2)];          // bilinear samples of the tile + halo
    __shared__ int16_t s_h[(AFF_TILE + 2) * AFF_TILE];                 // after the horizontal enhancement pass
    __shared__ int16_t s_pred[3][AFF_TILE * AFF_TILE];                 // prediction of the tile (chroma uses a quarter)
    __shared__ uint4   s_ltap[17];
    __shared__ uint2   s_ctap[33];
    const int t = threadIdx.x;
    const AffItem it = a.items[blockIdx.x];
    const uint4 r0 = ((const uint4 *)&a.cus[it.cu])[0], r1 = ((const uint4 *)&a.cus[it.cu])[1];
    const int cu_x = r0.x & 0xFFFF, cu_y = r0.x >> 16;
    const int lw = r0.y & 0xFF, lh = (r0.y >> 8) & 0xFF, cbf = r0.y >> 24;
    const int refis[2] = { (int)(int8_t)(r0.z & 0xFF), (int)(int8_t)((r0.z >> 8) & 0xFF) };
    const uint32_t coef_off = r0.w;
    const int ai = (int)((r1.w >> 8) & 0xFF), vn = (int)((r1.w >> 16) & 0xFF);
    const int cuw = 1 << lw, cuh = 1 << lh;
    const int tw = min(AFF_TILE, cuw), th = min(AFF_TILE, cuh), tx = it.tx, ty = it.ty;
    const int maxl = (1 << a.bd_l) - 1, maxc = (1 << a.bd_c) - 1;

    if (t < 17) s_ltap[t] = *(const uint4 *)k_luma_taps[a.admvp][t];
    else if (t >= 64 && t < 64 + 33) s_ctap[t - 64] = *(const uint2 *)k_chroma_taps[a.admvp][t - 64];

    int16_t cp[2][6];
    {
        const uint32_t *q = (const uint32_t *)(a.cpmv + (size_t)it.aff * 12);
#pragma unroll
        for (int k = 0; k < 6; k++) { const uint32_t v = q[k]; cp[k / 3][(k % 3) * 2] = (int16_t)(v & 0xFFFF); cp[k / 3][(k % 3) * 2 + 1] = (int16_t)(v >> 16); }
    }
    const bool use[2] = { refis[0] >= 0, refis[1] >= 0 };
    AffModel md[2];
    md[0] = aff_model(cp[0], lw, lh, vn); md[1] = aff_model(cp[1], lw, lh, vn);
    int sub_w, sub_h; bool mem_band;
    aff_subblock(md, use, lw, lh, sub_w, sub_h, mem_band);
    const bool eif = sub_w < 8 || sub_h < 8;
    __syncthreads();

    int nl = 0;
    for (int l = 0; l < 2; l++) {
        if (!use[l]) continue;
        const RefEntry &re = a.refp[refis[l]][l];
        const AffModel &m = md[l];
        const int mv_scale[2] = { cp[l][0] * (1 << AFF_BIT), cp[l][1] * (1 << AFF_BIT) };
        if (eif) {
            int mx[2], mn[2];
            aff_eif_range(cu_x, cu_y, lw, lh, m, mv_scale, a.pic_w, a.pic_h, !mem_band, mx, mn);
            for (int comp = 0; comp < 3; comp++) {
                const int cs = comp ? 1 : 0;                                     // 4:2:0
                const int bw = tw >> cs, bh = th >> cs, ox = tx >> cs, oy = ty >> cs, bd = comp ? a.bd_c : a.bd_l;
                const int16_t *ref = (comp == 0 ? re.y : comp == 1 ? re.u : re.v) + (cu_y >> cs) * (comp ? a.s_c : a.s_l) + (cu_x >> cs);
                const int s_ref = comp ? a.s_c : a.s_l;
                const int m0x = mv_scale[0] >> cs, m0y = mv_scale[1] >> cs;
                const int hx = mx[0] >> cs, hy = mx[1] >> cs, lx_ = mn[0] >> cs, ly_ = mn[1] >> cs;
                const int shift1 = min(4, bd - 8), shift2 = max(8, 20 - bd), off2 = 1 << (shift2 - 1);
                const int sh2 = max(bd + 5 - 16, 0), sh3 = 6 - sh2, of2 = sh2 ? 1 << (sh2 - 1) : 0, of3 = 1 << (sh3 - 1);
                const int ts = bw + 2;
                // bilinear fetch: xevdm_eif_bilinear_clip (:2456-2499); positions are relative to the CU
                for (int i = t; i < ts * (bh + 2); i += 256) {
                    const int py = i / ts - 1 + oy, px = i % ts - 1 + ox;
                    int vx = (m0x + px * m.dh[0] + py * m.dv[0]) >> 4, vy = (m0y + px * m.dh[1] + py * m.dv[1]) >> 4;
                    vx = min(hx, max(lx_, vx)); vy = min(hy, max(ly_, vy));
                    const int16_t *r = ref + (py + (vy >> 5)) * s_ref + px + (vx >> 5);
                    const int fx = vx & 31, fy = vy & 31;
                    const int s1 = (int)(int16_t)(((64 - 2 * fx) * r[0] + 2 * fx * r[1]) >> shift1);
                    const int s2 = (int)(int16_t)(((64 - 2 * fx) * r[s_ref] + 2 * fx * r[s_ref + 1]) >> shift1);
                    s_bl[i] = (int16_t)(((64 - 2 * fy) * s1 + 2 * fy * s2 + off2) >> shift2);
                }
                __syncthreads();
                // enhancement filter, rows then columns (xevdm_eif_filter :2428-2454)
                for (int i = t; i < bw * (bh + 2); i += 256) {
                    const int row = i / bw, col = i % bw;
                    const int16_t *q = s_bl + row * ts + col;
                    s_h[i] = (int16_t)((-q[0] + q[1] * 10 - q[2] + of2) >> sh2);
                }
                __syncthreads();
                for (int i = t; i < bw * bh; i += 256) {
                    const int row = i / bw, col = i % bw;
                    const int16_t *q = s_h + row * bw + col;
                    const int res = (int)(int16_t)((-q[0] + q[bw] * 10 - q[2 * bw] + of3) >> sh3);
                    const int v = clip3(0, (1 << bd) - 1, res);
                    s_pred[comp][i] = (int16_t)(nl ? (s_pred[comp][i] + v + 1) >> 1 : v);
                }
                __syncthreads();
            }
        } else {
            // one vector for the whole CU (see the header); filter variant from the unclipped vector like xevd_mc_l / xevd_mc_c (xevd_mc.h:66-74)
            const int hor_max = (a.pic_w + 128 - cu_x - cuw) * 16, ver_max = (a.pic_h + 128 - cu_y - cuh) * 16;
            const int hor_min = (-128 - cu_x) * 16, ver_min = (-128 - cu_y) * 16;
            const int ox = aff_clip18(aff_round(mv_scale[0] + m.dh[0] * (sub_w >> 1) + m.dv[0] * (sub_h >> 1), 5));
            const int oy = aff_clip18(aff_round(mv_scale[1] + m.dh[1] * (sub_w >> 1) + m.dv[1] * (sub_h >> 1), 5));
            const int cx = min(hor_max, max(hor_min, ox)), cy = min(ver_max, max(ver_min, oy));
            const int nsx = tw >> 2;
            if (t < nsx * (th >> 2)) {
                const int bx = (t % nsx) << 2, by = (t / nsx) << 2;                // inside the tile
                const int gx = (cu_x + tx + bx) * 16 + cx, gy = (cu_y + ty + by) * 16 + cy;
                const int ldx = (ox & 15) != 0, ldy = (oy & 15) != 0, cdx = (ox & 31) != 0, cdy = (oy & 31) != 0;
                uint32_t ch[4], cv[4], o[8], c2h[2], c2v[2], ou[2], ov[2];
                const uint4 lth = s_ltap[ldx ? (gx & 15) : 16], ltv = s_ltap[ldy ? (gy & 15) : 16];
                ch[0] = lth.x; ch[1] = lth.y; ch[2] = lth.z; ch[3] = lth.w; cv[0] = ltv.x; cv[1] = ltv.y; cv[2] = ltv.z; cv[3] = ltv.w;
                const int16_t *p = re.y + ((gy >> 4) - 3) * a.s_l + (gx >> 4) - 3;
                const Regime rg = regime(ldx, ldy, a.bd_l);
                if (ldx) { if (ldy) mc_luma_4x4<true, true>(p, a.s_l, ch, cv, rg, maxl, o); else mc_luma_4x4<true, false>(p, a.s_l, ch, cv, rg, maxl, o); }
                else     { if (ldy) mc_luma_4x4<false, true>(p, a.s_l, ch, cv, rg, maxl, o); else mc_luma_4x4<false, false>(p, a.s_l, ch, cv, rg, maxl, o); }
                const uint2 cth = s_ctap[cdx ? (gx & 31) : 32], ctv = s_ctap[cdy ? (gy & 31) : 32];
                c2h[0] = cth.x; c2h[1] = cth.y; c2v[0] = ctv.x; c2v[1] = ctv.y;
                const int off = ((gy >> 5) - 1) * a.s_c + (gx >> 5) - 1;
                const Regime rc = regime(cdx, cdy, a.bd_c);
#define MC_C(H, V) do { mc_chroma_2x2<H, V>(re.u + off, a.s_c, c2h, c2v, rc, maxc, ou); mc_chroma_2x2<H, V>(re.v + off, a.s_c, c2h, c2v, rc, maxc, ov); } while (0)
                if (cdx) { if (cdy) MC_C(true, true); else MC_C(true, false); }
                else     { if (cdy) MC_C(false, true); else MC_C(false, false); }
#undef MC_C
                uint32_t *dl = (uint32_t *)(s_pred[0] + by * tw + bx);
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    uint32_t *d = dl + r * (tw >> 1);
                    d[0] = nl ? avg2(d[0], o[r * 2]) : o[r * 2];
                    d[1] = nl ? avg2(d[1], o[r * 2 + 1]) : o[r * 2 + 1];
                }
                uint32_t *du = (uint32_t *)(s_pred[1] + (by >> 1) * (tw >> 1) + (bx >> 1)), *dv = (uint32_t *)(s_pred[2] + (by >> 1) * (tw >> 1) + (bx >> 1));
#pragma unroll
                for (int r = 0; r < 2; r++) {
                    du[r * (tw >> 2)] = nl ? avg2(du[r * (tw >> 2)], ou[r]) : ou[r];
                    dv[r * (tw >> 2)] = nl ? avg2(dv[r * (tw >> 2)], ov[r]) : ov[r];
                }
            }
            __syncthreads();
        }
        nl++;
    }
    if (nl == 0) return;

    // ---- residual add + clip + store (xevd_recon.c:35-92; ATS-inter: the coded TU is one end of the CU, k_inter.hip) ----
    int tu_x = 0, tu_y = 0, tu_w = cuw, tu_h = cuh;
    if (ai) {
        const int idx = ai & 15, pos = ai >> 4;
        if (idx == 2 || idx == 4) { tu_h = cuh >> (idx == 4 ? 2 : 1); tu_y = pos ? cuh - tu_h : 0; }
        else                      { tu_w = cuw >> (idx == 3 ? 2 : 1); tu_x = pos ? cuw - tu_w : 0; }
    }
    uint32_t off = coef_off;
    for (int comp = 0; comp < 3; comp++) {
        const int cs = comp ? 1 : 0;
        const int bw = tw >> cs, bh = th >> cs, cw_tu = tu_w >> cs, ch_tu = tu_h >> cs;
        const bool coded = (cbf >> comp) & 1;
        int16_t *plane = comp == 0 ? a.cur_y : comp == 1 ? a.cur_u : a.cur_v;
        const int s = comp ? a.s_c : a.s_l;
        int16_t *dst = plane + ((cu_y + ty) >> cs) * s + ((cu_x + tx) >> cs);
        for (int i = t; i < (bw >> 1) * bh; i += 256) {                       // two samples per lane: dword stores
            const int row = i / (bw >> 1), col = (i % (bw >> 1)) << 1;
            uint32_t pr = *(const uint32_t *)(s_pred[comp] + row * bw + col);
            if (coded) {
                const int lx = ((tx - tu_x) >> cs) + col, ly = ((ty - tu_y) >> cs) + row;      // inside the TU?
                if ((uint32_t)lx < (uint32_t)cw_tu && (uint32_t)ly < (uint32_t)ch_tu)
                    pr = recon2(pr, *(const uint32_t *)(a.resid + off + ly * cw_tu + lx), maxl);
            }
            *(uint32_t *)(dst + row * s + col) = pr;
        }
        if (coded) off += cw_tu * ch_tu;
    }

    // ---- the vectors of the SCU map: xevdm_set_affine_mvf (xevdm_util.c:4095-4190) ----
    if (t < (tw >> 2) * (th >> 2)) {
        const int sxc = (tx >> 2) + t % (tw >> 2), syc = (ty >> 2) + t / (tw >> 2);         // SCU position inside the CU
        const int sws = sub_w >> 2, shs = sub_h >> 2, w_cu = cuw >> 2, h_cu = cuh >> 2;
        const int w0 = sxc - sxc % sws, h0 = syc - syc % shs;
        ScuRec *rec = &a.maps[((cu_y >> 2) + syc) * a.w_scu + (cu_x >> 2) + sxc];
#pragma unroll
        for (int l = 0; l < 2; l++) {
            if (!use[l]) continue;
            int vx, vy;
            if (w0 == 0 && h0 == 0) { vx = cp[l][0]; vy = cp[l][1]; }
            else if (w0 + sws == w_cu && h0 == 0) { vx = cp[l][2]; vy = cp[l][3]; }
            else if (w0 == 0 && h0 + shs == h_cu && vn == 3) { vx = cp[l][4]; vy = cp[l][5]; }
            else {
                const int px = (w0 << 2) + (sub_w >> 1), py = (h0 << 2) + (sub_h >> 1);
                vx = aff_clip18(aff_round(cp[l][0] * (1 << AFF_BIT) + md[l].dh[0] * px + md[l].dv[0] * py, 5)) >> 2;
                vy = aff_clip18(aff_round(cp[l][1] * (1 << AFF_BIT) + md[l].dh[1] * px + md[l].dv[1] * py, 5)) >> 2;
            }
            *(uint32_t *)rec->mv[l] = (uint32_t)(uint16_t)(int16_t)vx | ((uint32_t)(uint16_t)(int16_t)vy << 16);
        }
    }
}

void launch_affine(xgpu_ctx *c, const AffineArgs &a)
{
    if (a.n_items) hipLaunchKernelGGL(k_affine, dim3(a.n_items), dim3(256), 0, c->stream, a);
}
