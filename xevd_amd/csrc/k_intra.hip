// k_intra.hip - intra prediction (Baseline modes) + residual add + clip for the intra CUs of one dependency level.
//
// Replaces the intra branch of xevd_recon_unit (src_base/xevd.c:731-741): xevd_get_avail_intra + xevd_get_nbr_b
// (src_base/xevd_ipred.c:33-94), xevd_ipred_b / xevd_ipred_uv_b (:96-164, 587-676) and xevd_recon_yuv.  The Main
// profile runs the same predictors when sps->tool_eipd = 0 (src_main/xevdm.c:1346-1381), on non-square CUs too.
//
// An intra CU predicts from reconstructed samples of CUs decoded before it, so intra CUs form a dependency graph
// on top of the inter CUs (which k_inter finishes first).  The batch builder (host) derives, per intra CU, which
// 4-sample units of the row above / the column to the left exist "already reconstructed" in decode order
// (= the reference's COD flags at that CU's turn) and the CU's level = 1 + max level of the intra CUs it reads;
// one launch of this kernel handles all CUs of one level - CUs of a level are independent by construction.
//
// MI355X mapping: one wavefront (= one 64-thread workgroup) per CU.  The neighbour arrays of the three components
// are staged once in LDS (unavailable units -> mid grey of the LUMA bit depth, xevd.c:455-473), DC sums are wave
// reductions, then every lane predicts whole 4x4 SCUs (+ their 2x2 chroma blocks) exactly like k_inter's lanes
// reconstruct theirs, so residual addressing and stores are shared idioms.  HBM-bound integer work: no MFMA.
#include "xgpu_internal.h"

#define NB_MAX 264      // up to 128 + 128 + 1 neighbour samples per side (luma of a 128x128 CU)

__device__ __forceinline__ uint32_t pack2i(int lo, int hi) { return ((uint32_t)lo & 0xFFFFu) | ((uint32_t)hi << 16); }
__device__ __forceinline__ int clip3i(int lo, int hi, int v) { return min(max(v, lo), hi); }
// rec = clip(0, max, (s16)(res + pred)) on packed pairs: the 16-bit sum wraps (xevd_recon.c:39,60)
__device__ __forceinline__ uint32_t recon2i(uint32_t pred, uint32_t res, int maxv)
{
    const int lo = (int)(int16_t)((pred & 0xFFFFu) + (res & 0xFFFFu));
    const int hi = (int)(int16_t)((pred >> 16) + (res >> 16));
    return pack2i(clip3i(0, maxv, lo), clip3i(0, maxv, hi));
}

// one predicted sample at (j, i) of a w-wide block; up/left are LDS arrays whose element [0] is index -1
__device__ __forceinline__ int ipred_sample(const int16_t *up, const int16_t *le, int mode, int dc, int i, int j)
{
    switch (mode) {
    case 0:  return dc;                                                              // IPD_DC_B
    case 1:  return le[1 + i];                                                       // IPD_HOR_B
    case 2:  return up[1 + j];                                                       // IPD_VER_B
    case 3:  return i > j ? le[i - j] : (i == j ? up[0] : up[j - i]);                // IPD_UL_B: le[i-j-1] / up[-1] / up[j-i-1]
    default: return (up[2 + i + j] + le[2 + i + j]) >> 1;                            // IPD_UR_B: index i+j+1
    }
}

__global__ __launch_bounds__(64) void k_intra(const IntraArgs a, int first)
{
    __shared__ int16_t s_up[3][NB_MAX], s_le[3][NB_MAX];
    const int t = threadIdx.x;
    const IntraRec ir = a.list[first + blockIdx.x];
    const uint4 r0 = ((const uint4 *)&a.cus[ir.cu])[0];
    const uint4 r1 = ((const uint4 *)&a.cus[ir.cu])[1];
    const int cu_x = r0.x & 0xFFFF, cu_y = r0.x >> 16;
    const int lw = r0.y & 0xFF, lh = (r0.y >> 8) & 0xFF, cbf = r0.y >> 24;
    const uint32_t coef_off = r0.w;
    const int cw = 1 << lw, chh = 1 << lh;
    const int mode_l = (r1.z >> 24) & 0xFF, mode_c = r1.w & 0xFF;                    // CuRec.ipm[0], ipm[1]
    const int mid = 1 << (a.bd_l - 1);
    const int maxv = (1 << a.bd_l) - 1;

    // ---- neighbour staging (xevd_get_nbr_b): element e of a side belongs to unit e / unit_size ----
#pragma unroll
    for (int c = 0; c < 3; c++) {
        const int16_t *plane = c == 0 ? a.cur_y : (c == 1 ? a.cur_u : a.cur_v);
        const int s = c ? a.s_c : a.s_l, sh = c ? 1 : 0, ush = c ? 1 : 2;
        const int16_t *org = plane + (cu_y >> sh) * s + (cu_x >> sh);
        const int n = (cw + chh) >> sh;
        for (int e = t; e < n; e += 64) {
            const int k = e >> ush;
            s_up[c][1 + e] = ((ir.up >> k) & 1) ? org[-s + e] : (int16_t)mid;
            s_le[c][1 + e] = ((ir.le >> k) & 1) ? org[e * s - 1] : (int16_t)mid;
        }
        if (t == 0) {
            const int16_t ul = (ir.flags & 1) ? org[-s - 1] : (int16_t)mid;
            s_up[c][0] = ul; s_le[c][0] = ul;
        }
    }
    __syncthreads();

    // ---- DC values (ipred_dc_b): (sum of h left + w up samples + w) >> (log2 w + 1) ----
    int dc[3] = { 0, 0, 0 };
#pragma unroll
    for (int c = 0; c < 3; c++) {
        const int w = c ? cw >> 1 : cw, h = c ? chh >> 1 : chh;
        int acc = 0;
        for (int e = t; e < w + h; e += 64) acc += e < h ? s_le[c][1 + e] : s_up[c][1 + e - h];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
        dc[c] = (acc + w) >> ((c ? lw - 1 : lw) + 1);
    }

    // ---- prediction + reconstruction, one 4x4 SCU per lane and step ----
    const int scuw = cw >> 2, nscu = scuw * (chh >> 2);
    const int cwc = cw >> 1;
    const uint32_t off_u = coef_off + ((cbf & 1) ? (uint32_t)(cw * chh) : 0u);
    const uint32_t off_v = off_u + ((cbf & 2) ? (uint32_t)(cwc * (chh >> 1)) : 0u);
    for (int sidx = t; sidx < nscu; sidx += 64) {
        const int lx = (sidx % scuw) << 2, ly = (sidx / scuw) << 2;
        const int x = cu_x + lx, y = cu_y + ly;
        int16_t *dy = a.cur_y + y * a.s_l + x;
#pragma unroll
        for (int r = 0; r < 4; r++) {
            int p[4];
#pragma unroll
            for (int q = 0; q < 4; q++) p[q] = ipred_sample(s_up[0], s_le[0], mode_l, dc[0], ly + r, lx + q);
            uint32_t v0 = pack2i(p[0], p[1]), v1 = pack2i(p[2], p[3]);
            if (cbf & 1) {
                const uint2 rs = *(const uint2 *)(a.resid + coef_off + (ly + r) * cw + lx);
                v0 = recon2i(v0, rs.x, maxv); v1 = recon2i(v1, rs.y, maxv);
            }
            *(uint2 *)(dy + r * a.s_l) = make_uint2(v0, v1);
        }
        const int coff = (y >> 1) * a.s_c + (x >> 1);
#pragma unroll
        for (int c = 1; c < 3; c++) {
            int16_t *d = (c == 1 ? a.cur_u : a.cur_v) + coff;
            const bool coded = (cbf >> c) & 1;
            const int16_t *rs = a.resid + (c == 1 ? off_u : off_v) + (ly >> 1) * cwc + (lx >> 1);
#pragma unroll
            for (int r = 0; r < 2; r++) {
                const int p0 = ipred_sample(s_up[c], s_le[c], mode_c, dc[c], (ly >> 1) + r, (lx >> 1));
                const int p1 = ipred_sample(s_up[c], s_le[c], mode_c, dc[c], (ly >> 1) + r, (lx >> 1) + 1);
                uint32_t v = pack2i(p0, p1);
                if (coded) v = recon2i(v, *(const uint32_t *)(rs + r * cwc), maxv);      // the luma depth clips chroma too (xevd_recon.c:75-90)
                *(uint32_t *)(d + r * a.s_c) = v;
            }
        }
    }
}

void launch_intra(xgpu_ctx *c, const IntraArgs &a, int first, int count)
{
    hipLaunchKernelGGL(k_intra, dim3(count), dim3(64), 0, c->stream, a, first);
}
