// k_affine.hip - affine motion compensation (Main, sps->tool_affine) + residual add + the sub-block vectors of the SCU map.
//
// Replaces, for the affine CUs of a picture, xevdm_affine_mc -> xevd_recon_yuv -> xevdm_set_affine_mvf
// (src_main/xevdm_mc.c:2606-2685 with :2259-2391, :2108-2150, :2393-2604; src_main/xevdm_util.c:1870-2125, :4095-4190).
//
// MI355X mapping: affine CUs are a minority of a picture, so they do not ride in k_inter's per-SCU lanes (k_inter writes their map records
// and leaves the samples alone); the host derives each CU's branch (affine_model.h, the same code the kernels run) and cuts the CU into tiles for one of two kernels:
//   * k_affine_eif, one WAVE per tile of at most 16x16 luma samples (four independent tiles per workgroup, wave-private LDS, no workgroup
//     barrier): EIF (a sub-block would be smaller than 8 samples): every sample has its own vector - mv0 + x*dHor + y*dVer at 9 fractional bits,
//     reduced to 1/32 sample and clamped - and is a bilinear fetch of 4 reference samples (a gather: one lane per sample of the
//     (w+2) x (h+2) window, straight from L2/HBM), followed by the [-1 10 -1] enhancement filter along rows and columns through LDS.  The
//     vector of a sample only depends on its position in the CU, so tiles are independent and a tile's one-sample halo is simply computed again.
//   * k_affine_sub, one wave per tile of at most 32x32 luma samples: sub-block translation with the 8-tap / 4-tap filters of k_inter
//     (mc_filters.h), one lane per 4x4 SCU, prediction kept in registers.  As the reference has
//     it (:2352-2353 - the sub-block's offset does not enter the vector) every sub-block moves with the vector at the centre of the FIRST one.
//   * the model (deltas, sub-block size, EIF decision and clamp range) is wave-uniform scalar arithmetic recomputed by every workgroup from
//     the control points - cheaper than another host-built table.
// Integer arithmetic and intermediate widths follow the reference exactly (its intermediates are s16 `pel`).
#include "mc_filters.h"
#include <type_traits>

#include "affine_model.h"

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

#define EIF_TILE 16                  // luma samples per tile side of the EIF kernel
#define SUB_TILE 32                  // ... of the translation kernel (8 x 8 SCUs = one wave)

__device__ __forceinline__ void aff_wave_sync()      // LDS traffic of one wave is processed in order: only the compiler needs the fence
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// what both kernels need of an item: the CU, its control points, the model
struct AffCu {
    int cu_x, cu_y, lw, lh, cbf, refi[2], ai, vn, tx, ty;
    uint32_t coef_off;
    int16_t cp[2][6];
    AffModel md[2];
    bool use[2], mem_band;
    int sub_w, sub_h;
};
__device__ __forceinline__ void aff_load(const AffineArgs &a, int item, AffCu &k)
{
    const AffItem it = a.items[item];
    const uint4 r0 = ((const uint4 *)&a.cus[it.cu])[0], r1 = ((const uint4 *)&a.cus[it.cu])[1];
    k.cu_x = r0.x & 0xFFFF; k.cu_y = r0.x >> 16; k.lw = r0.y & 0xFF; k.lh = (r0.y >> 8) & 0xFF; k.cbf = r0.y >> 24;
    k.refi[0] = (int)(int8_t)(r0.z & 0xFF); k.refi[1] = (int)(int8_t)((r0.z >> 8) & 0xFF);
    k.coef_off = r0.w; k.ai = (int)((r1.w >> 8) & 0xFF); k.vn = (int)((r1.w >> 16) & 0xFF); k.tx = it.tx; k.ty = it.ty;
    const uint32_t *q = (const uint32_t *)(a.cpmv + (size_t)it.aff * 12);
#pragma unroll
    for (int i = 0; i < 6; i++) { const uint32_t v = q[i]; k.cp[i / 3][(i % 3) * 2] = (int16_t)(v & 0xFFFF); k.cp[i / 3][(i % 3) * 2 + 1] = (int16_t)(v >> 16); }
    k.use[0] = k.refi[0] >= 0; k.use[1] = k.refi[1] >= 0;
    k.md[0] = aff_model(k.cp[0], k.lw, k.lh, k.vn); k.md[1] = aff_model(k.cp[1], k.lw, k.lh, k.vn);
    aff_subblock(k.md, k.use, k.lw, k.lh, k.sub_w, k.sub_h, k.mem_band);
}

// ATS-inter: the coded TU is one end of the CU (xevdm_get_tu_size / get_tu_pos_offset, k_inter.hip)
__device__ __forceinline__ void aff_tu(const AffCu &k, int &tu_x, int &tu_y, int &tu_w, int &tu_h)
{
    const int cuw = 1 << k.lw, cuh = 1 << k.lh;
    tu_x = 0; tu_y = 0; tu_w = cuw; tu_h = cuh;
    if (k.ai) {
        const int idx = k.ai & 15, pos = k.ai >> 4;
        if (idx == 2 || idx == 4) { tu_h = cuh >> (idx == 4 ? 2 : 1); tu_y = pos ? cuh - tu_h : 0; }
        else                      { tu_w = cuw >> (idx == 3 ? 2 : 1); tu_x = pos ? cuw - tu_w : 0; }
    }
}

// the vectors of the SCU map: xevdm_set_affine_mvf (xevdm_util.c:4095-4190); (sxc, syc) = SCU position inside the CU
__device__ __forceinline__ void aff_store_mvf(const AffineArgs &a, const AffCu &k, int sxc, int syc)
{
    const int sws = k.sub_w >> 2, shs = k.sub_h >> 2, w_cu = (1 << k.lw) >> 2, h_cu = (1 << k.lh) >> 2;
    const int w0 = sxc & ~(sws - 1), h0 = syc & ~(shs - 1);
    ScuRec *rec = &a.maps[((k.cu_y >> 2) + syc) * a.w_scu + (k.cu_x >> 2) + sxc];
#pragma unroll
    for (int l = 0; l < 2; l++) {
        if (!k.use[l]) continue;
        int vx, vy;
        if (w0 == 0 && h0 == 0) { vx = k.cp[l][0]; vy = k.cp[l][1]; }
        else if (w0 + sws == w_cu && h0 == 0) { vx = k.cp[l][2]; vy = k.cp[l][3]; }
        else if (w0 == 0 && h0 + shs == h_cu && k.vn == 3) { vx = k.cp[l][4]; vy = k.cp[l][5]; }
        else {
            const int px = (w0 << 2) + (k.sub_w >> 1), py = (h0 << 2) + (k.sub_h >> 1);
            vx = aff_clip18(aff_round(k.cp[l][0] * (1 << AFF_BIT) + k.md[l].dh[0] * px + k.md[l].dv[0] * py, 5)) >> 2;
            vy = aff_clip18(aff_round(k.cp[l][1] * (1 << AFF_BIT) + k.md[l].dh[1] * px + k.md[l].dv[1] * py, 5)) >> 2;
        }
        *(uint32_t *)rec->mv[l] = (uint32_t)(uint16_t)(int16_t)vx | ((uint32_t)(uint16_t)(int16_t)vy << 16);
    }
}

// ---------------------------------------------------------------------------------------------------------
// EIF tiles.  All gathers of a tile - both lists, three components - are issued in one phase (independent loads, one wait), then one
// horizontal and one vertical enhancement phase.
// ---------------------------------------------------------------------------------------------------------
#define EIF_WIN  ((EIF_TILE + 2) * (EIF_TILE + 2) + 2 * (EIF_TILE / 2 + 2) * (EIF_TILE / 2 + 2))       // window samples of one list: 324 + 2 * 100
#define EIF_HROW ((EIF_TILE + 2) * EIF_TILE + 2 * (EIF_TILE / 2 + 2) * (EIF_TILE / 2))
__global__ __launch_bounds__(256) void k_affine_eif(const AffineArgs a)
{
    __shared__ int16_t s_bl_[4][2][EIF_WIN];             // per wave, per list: bilinear samples of the three windows
    __shared__ int16_t s_h_[4][2][EIF_HROW];             // after the horizontal enhancement pass
    const int t = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int item = blockIdx.x * 4 + wv;
    if (item >= a.n_eif) return;
    AffCu k;
    aff_load(a, item, k);
    const int cuh = 1 << k.lh;
    const int ltw = min(4, k.lw), tw = 1 << ltw, th = min(EIF_TILE, cuh), tx = k.tx, ty = k.ty;      // tile: 8 or 16 samples a side
    const int maxl = (1 << a.bd_l) - 1;

    // ---- phase 1: bilinear fetch, xevdm_eif_bilinear_clip (:2456-2499); positions are relative to the CU ----
#pragma unroll
    for (int l = 0; l < 2; l++) {
        if (!k.use[l]) continue;
        const RefEntry &re = a.refp[__builtin_amdgcn_readfirstlane(k.refi[l])][l];      // (wave-uniform: scalar loads of the kernel arguments instead of a vector load per plane)
        const AffModel &m = k.md[l];
        const int mv_scale[2] = { k.cp[l][0] * (1 << AFF_BIT), k.cp[l][1] * (1 << AFF_BIT) };
        int mx[2], mn[2];
        aff_eif_range(k.cu_x, k.cu_y, k.lw, k.lh, m, mv_scale, a.pic_w, a.pic_h, !k.mem_band, mx, mn);
        // Every gather of the list - three components: six + two + two rounds of 64 lanes - goes out before the first sample is computed (a loop of gather +
        // arithmetic + LDS store waited for its four loads in every round: ten memory round trips per list, now one).  The two samples of a row are one dword
        // load at the 2-byte-aligned address.
        constexpr int RL = ((EIF_TILE + 2) * (EIF_TILE + 2) + 63) / 64, RC = ((EIF_TILE / 2 + 2) * (EIF_TILE / 2 + 2) + 63) / 64;
        uint32_t top[RL + 2 * RC], bot[RL + 2 * RC];
        int frac[RL + 2 * RC];
#pragma unroll
        for (int pass = 0; pass < 2; pass++) {
            int base = 0;
#pragma unroll
            for (int comp = 0; comp < 3; comp++) {
                const int cs = comp ? 1 : 0;                                     // 4:2:0
                const int bw = tw >> cs, bh = th >> cs, ox = tx >> cs, oy = ty >> cs, bd = comp ? a.bd_c : a.bd_l;
                const int s_ref = comp ? a.s_c : a.s_l;
                const int16_t *ref = (comp == 0 ? re.y : comp == 1 ? re.u : re.v) + (k.cu_y >> cs) * s_ref + (k.cu_x >> cs);
                const int m0x = mv_scale[0] >> cs, m0y = mv_scale[1] >> cs;
                const int hx = mx[0] >> cs, hy = mx[1] >> cs, lx_ = mn[0] >> cs, ly_ = mn[1] >> cs;
                const int shift1 = min(4, bd - 8), shift2 = max(8, 20 - bd), off2 = 1 << (shift2 - 1);
                const int ts = bw + 2, inv = (65536 + ts - 1) / ts;              // i / ts == (i * inv) >> 16 for the few hundred i of a window
                int16_t *dst = s_bl_[wv][l] + base;
#pragma unroll
                for (int it = 0; it < RL; it++) {
                    if (it >= (comp ? RC : RL)) continue;
                    const int q = (comp == 0 ? 0 : comp == 1 ? RL : RL + RC) + it, i = t + 64 * it;
                    if (pass == 0) {
                        top[q] = bot[q] = 0; frac[q] = 0;
                        if (i < ts * (bh + 2)) {
                            const int row = (i * inv) >> 16, col = i - row * ts;
                            const int py = row - 1 + oy, px = col - 1 + ox;
                            int vx = (m0x + px * m.dh[0] + py * m.dv[0]) >> 4, vy = (m0y + px * m.dh[1] + py * m.dv[1]) >> 4;
                            vx = min(hx, max(lx_, vx)); vy = min(hy, max(ly_, vy));
                            const gs16 r = (gs16)ref + (py + (vy >> 5)) * s_ref + px + (vx >> 5);
                            top[q] = gload4(r); bot[q] = gload4(r + s_ref);
                            frac[q] = (vx & 31) | ((vy & 31) << 8);
                        }
                    } else if (i < ts * (bh + 2)) {
                        const int fx = frac[q] & 31, fy = frac[q] >> 8;
                        const int s1 = (int)(int16_t)(((64 - 2 * fx) * (int)(int16_t)(top[q] & 0xFFFF) + 2 * fx * (int)(int16_t)(top[q] >> 16)) >> shift1);
                        const int s2 = (int)(int16_t)(((64 - 2 * fx) * (int)(int16_t)(bot[q] & 0xFFFF) + 2 * fx * (int)(int16_t)(bot[q] >> 16)) >> shift1);
                        dst[i] = (int16_t)(((64 - 2 * fy) * s1 + 2 * fy * s2 + off2) >> shift2);
                    }
                }
                base += ts * (bh + 2);
            }
        }
    }
    aff_wave_sync();
    // ---- phase 2: enhancement filter along the rows (xevdm_eif_filter :2428-2454) ----
#pragma unroll
    for (int l = 0; l < 2; l++) {
        if (!k.use[l]) continue;
        int base = 0, hbase = 0;
#pragma unroll
        for (int comp = 0; comp < 3; comp++) {
            const int cs = comp ? 1 : 0, lbw = ltw - cs, bw = 1 << lbw, bh = th >> cs, bd = comp ? a.bd_c : a.bd_l, ts = bw + 2;
            const int sh2 = max(bd + 5 - 16, 0), of2 = sh2 ? 1 << (sh2 - 1) : 0;
            const int16_t *src = s_bl_[wv][l] + base;
            int16_t *dst = s_h_[wv][l] + hbase;
            for (int i = t; i < bw * (bh + 2); i += 64) {
                const int16_t *q = src + (i >> lbw) * ts + (i & (bw - 1));
                dst[i] = (int16_t)((-q[0] + q[1] * 10 - q[2] + of2) >> sh2);
            }
            base += ts * (bh + 2); hbase += bw * (bh + 2);
        }
    }
    aff_wave_sync();
    // ---- phase 3: along the columns, bi-prediction average, residual add (xevd_recon.c:35-92), store.  Two samples per lane ----
    int tu_x, tu_y, tu_w, tu_h;
    aff_tu(k, tu_x, tu_y, tu_w, tu_h);
    uint32_t off = k.coef_off;
    int hbase = 0;
#pragma unroll
    for (int comp = 0; comp < 3; comp++) {
        const int cs = comp ? 1 : 0, lbw = ltw - cs, bw = 1 << lbw, bh = th >> cs, bd = comp ? a.bd_c : a.bd_l, cw_tu = tu_w >> cs, ch_tu = tu_h >> cs;
        const int sh3 = 6 - max(bd + 5 - 16, 0), of3 = 1 << (sh3 - 1), maxv = (1 << bd) - 1;
        const bool coded = (k.cbf >> comp) & 1;
        int16_t *plane = comp == 0 ? a.cur_y : comp == 1 ? a.cur_u : a.cur_v;
        const int s = comp ? a.s_c : a.s_l;
        int16_t *dst = plane + ((k.cu_y + ty) >> cs) * s + ((k.cu_x + tx) >> cs);
        for (int i = t; i < (bw >> 1) * bh; i += 64) {
            const int row = i >> (lbw - 1), col = (i & ((bw >> 1) - 1)) << 1;
            int v[2] = { 0, 0 };
            int nl = 0;
#pragma unroll
            for (int l = 0; l < 2; l++) {
                if (!k.use[l]) continue;
                const int16_t *q = s_h_[wv][l] + hbase + row * bw + col;
#pragma unroll
                for (int e = 0; e < 2; e++) {
                    const int res = (int)(int16_t)((-q[e] + q[bw + e] * 10 - q[2 * bw + e] + of3) >> sh3);
                    const int p = clip3(0, maxv, res);
                    v[e] = nl ? (v[e] + p + 1) >> 1 : p;
                }
                nl++;
            }
            uint32_t pr = pack2(v[0], v[1]);
            if (coded) {
                const int lx = ((tx - tu_x) >> cs) + col, ly = ((ty - tu_y) >> cs) + row;      // inside the TU?
                if ((uint32_t)lx < (uint32_t)cw_tu && (uint32_t)ly < (uint32_t)ch_tu)
                    pr = recon2(pr, *(const uint32_t *)(a.resid + off + ly * cw_tu + lx), maxl);
            }
            *(uint32_t *)(dst + row * s + col) = pr;
        }
        if (coded) off += cw_tu * ch_tu;
        hbase += bw * (bh + 2);
    }
    if (t < (tw >> 2) * (th >> 2)) aff_store_mvf(a, k, (tx >> 2) + (t & ((tw >> 2) - 1)), (ty >> 2) + (t >> (ltw - 2)));
}

// ---------------------------------------------------------------------------------------------------------
// Sub-block translation tiles: one vector for the whole CU (see the header); one lane per 4x4 SCU like k_inter's per-lane path.
// ---------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_affine_sub(const AffineArgs a)
{
    __shared__ uint4   s_ltap[17];
    __shared__ uint2   s_ctap[33];
    if (threadIdx.x < 17) s_ltap[threadIdx.x] = *(const uint4 *)k_luma_taps[a.admvp][threadIdx.x];
    else if (threadIdx.x >= 64 && threadIdx.x < 64 + 33) s_ctap[threadIdx.x - 64] = *(const uint2 *)k_chroma_taps[a.admvp][threadIdx.x - 64];
    __syncthreads();
    const int t = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int item = blockIdx.x * 4 + wv;
    if (item >= a.n_sub) return;
    AffCu k;
    aff_load(a, a.n_eif + item, k);
    const int cuw = 1 << k.lw, cuh = 1 << k.lh;
    const int ltw = min(5, k.lw), tw = 1 << ltw, th = min(SUB_TILE, cuh);
    if (t >= (tw >> 2) * (th >> 2)) return;
    const int bx = k.tx + ((t & ((tw >> 2) - 1)) << 2), by = k.ty + ((t >> (ltw - 2)) << 2);      // inside the CU
    const int x = k.cu_x + bx, y = k.cu_y + by;
    const int maxl = (1 << a.bd_l) - 1, maxc = (1 << a.bd_c) - 1;
    const int hor_max = (a.pic_w + 128 - k.cu_x - cuw) * 16, ver_max = (a.pic_h + 128 - k.cu_y - cuh) * 16;
    const int hor_min = (-128 - k.cu_x) * 16, ver_min = (-128 - k.cu_y) * 16;
    uint32_t pl[8], pu[2], pv[2];
    int nl = 0;
#pragma unroll
    for (int l = 0; l < 2; l++) {
        if (!k.use[l]) continue;
        const RefEntry &re = a.refp[__builtin_amdgcn_readfirstlane(k.refi[l])][l];      // (wave-uniform: scalar loads of the kernel arguments instead of a vector load per plane)
        const AffModel &m = k.md[l];
        // filter variant from the unclipped vector like xevd_mc_l / xevd_mc_c (xevd_mc.h:66-74)
        const int ox = aff_clip18(aff_round(k.cp[l][0] * (1 << AFF_BIT) + m.dh[0] * (k.sub_w >> 1) + m.dv[0] * (k.sub_h >> 1), 5));
        const int oy = aff_clip18(aff_round(k.cp[l][1] * (1 << AFF_BIT) + m.dh[1] * (k.sub_w >> 1) + m.dv[1] * (k.sub_h >> 1), 5));
        const int cx = min(hor_max, max(hor_min, ox)), cy = min(ver_max, max(ver_min, oy));
        const int gx = x * 16 + cx, gy = y * 16 + cy;
        const int ldx = (ox & 15) != 0, ldy = (oy & 15) != 0, cdx = (ox & 31) != 0, cdy = (oy & 31) != 0;
        uint32_t ch[4], cv[4], o[8], c2h[2], c2v[2], ou[2], ov[2];
        const uint4 lth = s_ltap[ldx ? (gx & 15) : 16], ltv = s_ltap[ldy ? (gy & 15) : 16];
        ch[0] = lth.x; ch[1] = lth.y; ch[2] = lth.z; ch[3] = lth.w; cv[0] = ltv.x; cv[1] = ltv.y; cv[2] = ltv.z; cv[3] = ltv.w;
        const gs16 p = (gs16)re.y + ((gy >> 4) - 3) * a.s_l + (gx >> 4) - 3;
        const Regime rg = regime(ldx, ldy, a.bd_l);
        if (ldx) { if (ldy) mc_luma_4x4<true, true>(p, a.s_l, ch, cv, rg, maxl, o); else mc_luma_4x4<true, false>(p, a.s_l, ch, cv, rg, maxl, o); }
        else     { if (ldy) mc_luma_4x4<false, true>(p, a.s_l, ch, cv, rg, maxl, o); else mc_luma_4x4<false, false>(p, a.s_l, ch, cv, rg, maxl, o); }
        const uint2 cth = s_ctap[cdx ? (gx & 31) : 32], ctv = s_ctap[cdy ? (gy & 31) : 32];
        c2h[0] = cth.x; c2h[1] = cth.y; c2v[0] = ctv.x; c2v[1] = ctv.y;
        const int off = ((gy >> 5) - 1) * a.s_c + (gx >> 5) - 1;
        const Regime rc = regime(cdx, cdy, a.bd_c);
#define MC_C(H, V) do { mc_chroma_2x2<H, V>((gs16)re.u + off, a.s_c, c2h, c2v, rc, maxc, ou); mc_chroma_2x2<H, V>((gs16)re.v + off, a.s_c, c2h, c2v, rc, maxc, ov); } while (0)
        if (cdx) { if (cdy) MC_C(true, true); else MC_C(true, false); }
        else     { if (cdy) MC_C(false, true); else MC_C(false, false); }
#undef MC_C
#pragma unroll
        for (int i = 0; i < 8; i++) pl[i] = nl ? avg2(pl[i], o[i]) : o[i];
#pragma unroll
        for (int i = 0; i < 2; i++) { pu[i] = nl ? avg2(pu[i], ou[i]) : ou[i]; pv[i] = nl ? avg2(pv[i], ov[i]) : ov[i]; }
        nl++;
    }
    if (nl == 0) return;
    // ---- residual add + clip (xevd_recon.c:35-92) ----
    int tu_x, tu_y, tu_w, tu_h;
    aff_tu(k, tu_x, tu_y, tu_w, tu_h);
    const int lx = bx - tu_x, ly = by - tu_y;
    if ((uint32_t)lx < (uint32_t)tu_w && (uint32_t)ly < (uint32_t)tu_h) {
        uint32_t off = k.coef_off;
        const int cwc = tu_w >> 1;
        if (k.cbf & 1) {
            const int16_t *r = a.resid + off + ly * tu_w + lx;
#pragma unroll
            for (int i = 0; i < 4; i++) { const uint2 v = *(const uint2 *)(r + i * tu_w); pl[i * 2] = recon2(pl[i * 2], v.x, maxl); pl[i * 2 + 1] = recon2(pl[i * 2 + 1], v.y, maxl); }
            off += tu_w * tu_h;
        }
        if (k.cbf & 2) {
            const int16_t *r = a.resid + off + (ly >> 1) * cwc + (lx >> 1);
            pu[0] = recon2(pu[0], *(const uint32_t *)r, maxl); pu[1] = recon2(pu[1], *(const uint32_t *)(r + cwc), maxl);
            off += cwc * (tu_h >> 1);
        }
        if (k.cbf & 4) {
            const int16_t *r = a.resid + off + (ly >> 1) * cwc + (lx >> 1);
            pv[0] = recon2(pv[0], *(const uint32_t *)r, maxl); pv[1] = recon2(pv[1], *(const uint32_t *)(r + cwc), maxl);
        }
    }
    int16_t *dy = a.cur_y + y * a.s_l + x;
#pragma unroll
    for (int i = 0; i < 4; i++) *(uint2 *)(dy + i * a.s_l) = make_uint2(pl[i * 2], pl[i * 2 + 1]);
    const int coff = (y >> 1) * a.s_c + (x >> 1);
    *(uint32_t *)(a.cur_u + coff) = pu[0];
    *(uint32_t *)(a.cur_u + coff + a.s_c) = pu[1];
    *(uint32_t *)(a.cur_v + coff) = pv[0];
    *(uint32_t *)(a.cur_v + coff + a.s_c) = pv[1];
    aff_store_mvf(a, k, bx >> 2, by >> 2);
}

void launch_affine(xgpu_ctx *c, const AffineArgs &a, hipStream_t st)
{
    if (a.n_eif) hipLaunchKernelGGL(k_affine_eif, dim3((a.n_eif + 3) / 4), dim3(256), 0, st, a);
    if (a.n_sub) hipLaunchKernelGGL(k_affine_sub, dim3((a.n_sub + 3) / 4), dim3(256), 0, st, a);
}
