// k_intra_ctu.hip - the intra CUs of a picture, one workgroup per CTU with the CTU in LDS.  An OPTION (XEVD_HIP_INTRA_CTU=1), bit-exact, measured slower than
// k_intra.hip's data-flow launch on the pictures it was written for - kept with its numbers because it is the design the graph's depth suggests.
//
// k_intra.hip hands samples from CU to CU through global memory: a link of a dependency chain is a flag store becoming visible, a poll seeing it, the neighbour
// loads and the store acknowledgement - coherent memory round trips; an all-intra 1080p picture has 650 (Baseline) to 1 700 (EIPD) levels and takes 2.3 / 8.0 ms
// in which the machine does next to nothing.  Here the chain of a CTU stays inside one workgroup: the CTU's samples (plus the row above it, twice as wide - the
// above-right neighbours -, and the column to its left) live in an LDS tile, a CU reads its neighbour arrays from the tile and writes its reconstruction to the
// tile and to the picture; "done" is a per-SCU word in LDS (bit 0 luma / bit 1 chroma pending, so the luma-only and chroma-only CUs of a local dual tree wait
// for the right thing), polled by the waves of the workgroup.  Global flags (one per CTU, the launch's epoch as in k_intra.hip) only order CTUs: a workgroup
// draws its CTU from a ticket counter in raster order - every CTU it can wait for (left, above-left, above, above-right; only those the host found a dependency
// on) has a lower ticket, so it is running or done whatever the dispatcher does - loads the borders with coherent loads once those CTUs have published, and
// publishes its own flag after its stores have drained.
//
// Measured (1080p all-intra, tools/time_all_intra.py): Baseline 3.3 ms (data-flow launch 2.3), Main / EIPD 8.9 ms (8.0); with the waits between CTUs switched
// off (XEVD_HIP_INTRA_CTU=2, wrong pictures) a CTU of 26 CUs takes 60 / 135 us - 2.5 - 5 us per CU with no memory round trip in the chain.  Cycle stamps
// (-DCTU_PROFILE=<ticket>): staging 2 100 - 2 600 core clocks, plan 1 000 - 1 700, 4 400 per step of six angular samples (a 32x32 CU: four steps), release 290:
// what a link costs is the instruction latency of ONE wave walking ~10^3 dependent instructions, in either kernel; the round trips are the smaller part.  And
// a CTU that starts when the CTU above-right has finished gives up the CU-granular overlap between CTUs (w_ctu + 2 h_ctu steps of a whole CTU's chain
// are as many CU links as the graph's depth).  What would help both kernels: fewer dependent instructions per CU (several waves per large CU, predictors that
// share work along a row).
//
// Same arithmetic as k_intra.hip (intra_pred.h): xevd_get_nbr_b / xevdm_get_nbr availability from the host's unit masks, the Baseline predictors
// (src_base/xevd_ipred.c:96-164, 587-676) or the EIPD ones (src_main/xevdm_ipred.c:241-305), xevd_recon's clip.  No IBC and no HTDF nodes: batches with
// those keep the data-flow launch (the host decides, build_intra_plan).
#include "intra_pred.h"

#define CTU_WAVES   8
#define CTU_REC_CAP 256          // records of a CTU kept in LDS (a 64x64 CTU of 4x4 CUs); CUs beyond that read theirs from memory

struct CtuLds { int tl, tu, tv, nb, left, rec, misc, total; };      // byte offsets
__host__ __device__ inline CtuLds ctu_lds(int log2_ctu)
{
    const int ctu = 1 << log2_ctu, cc = ctu >> 1, SL = 2 * ctu + 8, SC = 2 * cc + 8, nsc = ctu >> 2;
    auto a16 = [](int v) { return (v + 15) & ~15; };
    CtuLds L;
    int o = 0;
    L.tl = o; o += a16((ctu + 1) * SL * 2);
    L.tu = o; o += a16((cc + 1) * SC * 2);
    L.tv = o; o += a16((cc + 1) * SC * 2);
    L.nb = o; o += a16(CTU_WAVES * 3 * NB_LEN * 2);
    L.left = o; o += a16(nsc * nsc * 4);
    L.rec = o; o += CTU_REC_CAP * (int)sizeof(IntraRec);
    L.misc = o; o += 16;
    L.total = o;
    return L;
}

template <bool EIPD>
__global__ __launch_bounds__(64 * CTU_WAVES) void k_intra_ctu(const IntraArgs a, const IntraCtu *ctus, int log2_ctu, int pic_w, int pic_h)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    constexpr int NT = 64 * CTU_WAVES;
    const int tid = threadIdx.x, t = tid & 63, wv = tid >> 6;
    const int ctu = 1 << log2_ctu, cc = ctu >> 1, SL = 2 * ctu + 8, SC = 2 * cc + 8, nsc = ctu >> 2;
    const CtuLds L = ctu_lds(log2_ctu);
    // tile coordinates: (x, y) relative to the CTU's origin, x from -1 (row -1: to 2 * ctu - 1), y from -1; sample (x, y) sits at [(y + 1) * S + x + 4]
    int16_t *const tl = (int16_t *)(smem + L.tl), *const tu = (int16_t *)(smem + L.tu), *const tv = (int16_t *)(smem + L.tv);
    int16_t (*const nb)[NB_LEN] = (int16_t (*)[NB_LEN])(smem + L.nb + wv * 3 * NB_LEN * 2);
    uint32_t *const s_left = (uint32_t *)(smem + L.left);
    uint4 *const s_rec = (uint4 *)(smem + L.rec);
    uint32_t *const s_misc = (uint32_t *)(smem + L.misc);
    const int mid = 1 << (a.bd_l - 1), maxv = (1 << a.bd_l) - 1;

    if (tid == 0) s_misc[0] = atomicAdd(&a.done[a.n_intra], 1u) - a.ticket_base;
    for (int i = tid; i < nsc * nsc; i += NT) s_left[i] = 0;
    __syncthreads();
    const uint32_t tk = uni(s_misc[0]);
    const IntraCtu *const ent = ctus + tk;
    const uint32_t first = uni(ent->first), count = uni(ent->count);
    const int x0 = (int)(uni(ent->xy) & 0xFFFF), y0 = (int)(uni(ent->xy) >> 16);

    // ---- the CTU's records and what the picture holds inside the CTU (the inter CUs of a mixed picture; plain loads: written by earlier launches) ----
    {
        const uint4 *list4 = (const uint4 *)(a.list + first);
        const int nrec4 = (int)min(count, (uint32_t)CTU_REC_CAP) * 3;
        for (int i = tid; i < nrec4; i += NT) s_rec[i] = list4[i];
        const int hw = ctu >> 1, lhw = log2_ctu - 1;
        for (int i0 = 0; i0 < ctu * hw; i0 += NT * 4) {
            uint32_t d[4];
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const int i = i0 + q * NT + tid, r = i >> lhw, x = (i & (hw - 1)) << 1;
                d[q] = 0;
                if (i < ctu * hw && y0 + r < pic_h && x0 + x < pic_w) d[q] = *(const uint32_t *)(a.cur_y + (y0 + r) * a.s_l + x0 + x);
            }
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const int i = i0 + q * NT + tid, r = i >> lhw, x = (i & (hw - 1)) << 1;
                if (i < ctu * hw) *(uint32_t *)&tl[(r + 1) * SL + x + 4] = d[q];
            }
        }
        const int hwc = cc >> 1, lhwc = log2_ctu - 2, per = cc * hwc;
        for (int i0 = 0; i0 < 2 * per; i0 += NT * 2) {
            uint32_t d[2];
#pragma unroll
            for (int q = 0; q < 2; q++) {
                const int i = i0 + q * NT + tid, p = i >= per ? 1 : 0, k = i - p * per, r = k >> lhwc, x = (k & (hwc - 1)) << 1;
                d[q] = 0;
                if (i < 2 * per && (y0 >> 1) + r < (pic_h >> 1) && (x0 >> 1) + x < (pic_w >> 1)) d[q] = *(const uint32_t *)((p ? a.cur_v : a.cur_u) + ((y0 >> 1) + r) * a.s_c + (x0 >> 1) + x);
            }
#pragma unroll
            for (int q = 0; q < 2; q++) {
                const int i = i0 + q * NT + tid, p = i >= per ? 1 : 0, k = i - p * per, r = k >> lhwc, x = (k & (hwc - 1)) << 1;
                if (i < 2 * per) *(uint32_t *)&(p ? tv : tu)[(r + 1) * SC + x + 4] = d[q];
            }
        }
    }
    __syncthreads();
    auto load_rec = [&](int idx, uint4 &q0, uint4 &q1, uint4 &q2) {
        if (idx < CTU_REC_CAP) { q0 = s_rec[idx * 3]; q1 = s_rec[idx * 3 + 1]; q2 = s_rec[idx * 3 + 2]; }
        else { const uint4 *rec = (const uint4 *)&a.list[first + idx]; q0 = rec[0]; q1 = rec[1]; q2 = rec[2]; }
    };
    // ---- pending bits of the SCUs the CTU's CUs cover (everything else - inter CUs, the space outside the picture - reads as done) ----
    for (int idx = wv; idx < (int)count; idx += CTU_WAVES) {
        uint4 q0, q1, q2;
        load_rec(idx, q0, q1, q2);
        const uint32_t g = uni(q2.x), m = uni(q2.y), fl = uni(q0.y);
        const int lx = (int)(g & 0xFFFF) - x0, ly = (int)(g >> 16) - y0, lw = m & 0xFF, lh = (m >> 8) & 0xFF;
        const int scuw = 1 << (lw - 2), nscu = scuw << (lh - 2);
        const uint32_t bits = (fl & 64u) ? 1u : (fl & 32u) ? 2u : 3u;
        for (int s = t; s < nscu; s += 64) atomicOr(&s_left[((ly >> 2) + (s >> (lw - 2))) * nsc + (lx >> 2) + (s & (scuw - 1))], bits);
    }
    // ---- the CTUs this one reads from have published; then their samples along the borders (coherent loads: other workgroups of this launch wrote them) ----
    if (wv == 0 && t < 4) {
        const uint32_t j = ent->nbr[t];
        if (j != 0xFFFFFFFFu) while (__hip_atomic_load(&a.done[j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != a.epoch) __builtin_amdgcn_s_sleep(1);
    }
    __syncthreads();
    asm volatile("" ::: "memory");
    {
        const int n_tl = ctu + 1, n_tc = cc + 1, ntask = n_tl + 2 * n_tc + ctu + 2 * cc;
        for (int k = tid; k < ntask; k += NT) {
            if (k < n_tl) {                                            // the row above, samples -2 .. 2 * ctu - 1 in pairs
                const int x = -2 + 2 * k;
                if (y0 > 0 && x0 + x >= 0 && x0 + x < pic_w) *(uint32_t *)&tl[x + 4] = ld_coherent(a.cur_y + (y0 - 1) * a.s_l + x0 + x);
            } else if (k < n_tl + 2 * n_tc) {
                const int kk = k - n_tl, p = kk >= n_tc ? 1 : 0, x = -2 + 2 * (kk - p * n_tc);
                if (y0 > 0 && (x0 >> 1) + x >= 0 && (x0 >> 1) + x < (pic_w >> 1))
                    *(uint32_t *)&(p ? tv : tu)[x + 4] = ld_coherent((p ? a.cur_v : a.cur_u) + ((y0 >> 1) - 1) * a.s_c + (x0 >> 1) + x);
            } else if (k < n_tl + 2 * n_tc + ctu) {                    // the column to the left: the high half of the dword that ends at the CTU's edge
                const int r = k - n_tl - 2 * n_tc;
                if (x0 > 0 && y0 + r < pic_h) tl[(r + 1) * SL + 3] = (int16_t)(ld_coherent(a.cur_y + (y0 + r) * a.s_l + x0 - 2) >> 16);
            } else {
                const int kk = k - n_tl - 2 * n_tc - ctu, p = kk >= cc ? 1 : 0, r = kk - p * cc;
                if (x0 > 0 && (y0 >> 1) + r < (pic_h >> 1)) (p ? tv : tu)[(r + 1) * SC + 3] = (int16_t)(ld_coherent((p ? a.cur_v : a.cur_u) + ((y0 >> 1) + r) * a.s_c + (x0 >> 1) - 2) >> 16);
            }
        }
    }
    __syncthreads();

    // ---- the CUs in decoding order, wave w takes CUs w, w + 8, ...: every CU a wave can wait for is in the hands of a wave that is not waiting for this one ----
#ifdef CTU_PROFILE
#define STAMP(k) do { if (tk == CTU_PROFILE && t == 0 && idx < 64) a.done[gridDim.x + 8 * idx + (k)] = (uint32_t)clock64(); } while (0)
    if (tk == CTU_PROFILE && tid == 0) { a.done[gridDim.x + 8 * 64] = (uint32_t)clock64(); a.done[gridDim.x + 8 * 64 + 1] = (uint32_t)wall_clock64(); }
#else
#define STAMP(k)
#endif
    for (int idx = wv; idx < (int)count; idx += CTU_WAVES) {
        STAMP(0);
        uint4 q0, q1, q2;
        load_rec(idx, q0, q1, q2);
        const uint32_t nflags = uni(q0.y);
        const uint32_t avail_ul = nflags & 1;
        const uint64_t avail_up = (uint64_t)uni(q0.z) | ((uint64_t)uni(q0.w) << 32), avail_le = (uint64_t)uni(q1.x) | ((uint64_t)uni(q1.y) << 32);
        const uint32_t g = uni(q2.x), m = uni(q2.y), ipm = uni(q2.z), coef_off = uni(q2.w);
        const int cu_x = g & 0xFFFF, cu_y = g >> 16, lx0 = cu_x - x0, ly0 = cu_y - y0;
        const int lw = m & 0xFF, lh = (m >> 8) & 0xFF, cbf = (m >> 16) & 0xFF;
        const int mode_l = ipm & 0xFF, mode_c = (ipm >> 8) & 0xFF;
        const int cw = 1 << lw, chh = 1 << lh, scuw = cw >> 2, nscu = scuw * (chh >> 2), cwc = cw >> 1, units = (cw + chh) >> 2;
        const uint32_t off_u = coef_off + ((cbf & 1) ? (uint32_t)(cw * chh) : 0u);
        const uint32_t off_v = off_u + ((cbf & 2) ? (uint32_t)(cwc * (chh >> 1)) : 0u);
        const uint32_t own = (nflags & 64u) ? 1u : (nflags & 32u) ? 2u : 3u;      // what this CU reconstructs = what it needs of its neighbours

        // residual: requested before the wait (plain loads, the residual pass ran in an earlier launch).  EIPD works in units (k_intra.hip), Baseline in SCUs
        const int nunit = nscu << 2, uhalf = nscu << 1;
        uint2 ul[4] = { { 0, 0 }, { 0, 0 }, { 0, 0 }, { 0, 0 } };
        uint32_t uc[4] = { 0, 0, 0, 0 };
        auto fetch_units = [&](int u0) {
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const int u = u0 + 64 * k, c = u >= uhalf ? 1 : 0;
                ul[k] = make_uint2(0, 0); uc[k] = 0;
                if (u < nunit) {
                    if (cbf & 1) ul[k] = *(const uint2 *)(a.resid + coef_off + 4 * u);
                    if ((cbf >> (1 + c)) & 1) uc[k] = *(const uint32_t *)(a.resid + (c ? off_v : off_u) + 2 * (u - (c ? uhalf : 0)));
                }
            }
        };
        uint2 rl[4] = { { 0, 0 }, { 0, 0 }, { 0, 0 }, { 0, 0 } };
        uint32_t rc[2][2] = { { 0, 0 }, { 0, 0 } };
        auto fetch_resid = [&](int lx, int ly) {
            if (cbf & 1)
#pragma unroll
                for (int r = 0; r < 4; r++) rl[r] = *(const uint2 *)(a.resid + coef_off + (ly + r) * cw + lx);
#pragma unroll
            for (int c = 1; c < 3; c++)
                if ((cbf >> c) & 1)
#pragma unroll
                    for (int r = 0; r < 2; r++) rc[c - 1][r] = *(const uint32_t *)(a.resid + (c == 1 ? off_u : off_v) + ((ly >> 1) + r) * cwc + (lx >> 1));
        };
        if (EIPD) fetch_units(t);
        else { const int s0 = min(t, nscu - 1); fetch_resid((s0 % scuw) << 2, (s0 / scuw) << 2); }      // (every lane, clamped: the residual is then consumed on every path - see below)

        // ---- wait for the neighbour SCUs inside the CTU that the predictors read (outside: the CTU-level wait has covered them) ----
        {
            int need_up = units, need_le = units;
            bool need_ul = true;
            if (!EIPD) {                                               // xevd_ipred.c:96-164, 587-622: DC 0, HOR 1, VER 2, UL 3, UR 4
                need_up = need_le = 0; need_ul = false;
#pragma unroll
                for (int k = 0; k < 2; k++) {
                    const int md = k ? mode_c : mode_l;
                    if (md == 0 || md == 3) { need_up = max(need_up, scuw); need_le = max(need_le, chh >> 2); need_ul |= md == 3; }
                    else if (md == 1) need_le = max(need_le, chh >> 2);
                    else if (md == 2) need_up = max(need_up, scuw);
                    else { need_up = units; need_le = units; }
                }
            }
            const int sx = lx0 >> 2, sy = ly0 >> 2;
            int i_up = -1, i_le = -1, i_ul = -1;
            if (t < units) {
                if (sy > 0 && t < need_up && ((avail_up >> t) & 1) && sx + t < nsc) i_up = (sy - 1) * nsc + sx + t;
                if (sx > 0 && t < need_le && ((avail_le >> t) & 1) && sy + t < nsc) i_le = (sy + t) * nsc + sx - 1;
            }
            if (t == 0 && need_ul && avail_ul && sx > 0 && sy > 0) i_ul = (sy - 1) * nsc + sx - 1;
            auto busy = [&](int i) -> bool { return i >= 0 && (__hip_atomic_load(&s_left[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) & own) != 0; };
            while (__builtin_amdgcn_ballot_w64(busy(i_up) || busy(i_le) || busy(i_ul)) != 0) __builtin_amdgcn_s_sleep(1);
            asm volatile("" ::: "memory");                            // (LDS is read in program order; a workgroup-scope fence would also wait for the wave's global stores)
        }

        STAMP(1);
        // ---- neighbour arrays from the tile ----
#pragma unroll
        for (int c = 0; c < 3; c++) {
            const int16_t *tile = c == 0 ? tl : (c == 1 ? tu : tv);
            const int S = c ? SC : SL, sh = c ? 1 : 0, ush = c ? 1 : 2, usz = c ? 2 : 4;
            const int ox = lx0 >> sh, oy = ly0 >> sh, n = (cw + chh) >> sh;
            auto T = [&](int x, int y) -> int { return (int)(uint16_t)tile[(y + 1) * S + x + 4]; };
            if (EIPD) {
                // xevdm_get_nbr (xevdm_ipred.c:39-148): an unavailable unit repeats the last sample of the nearest available unit before it (the corner value when
                // there is none)
                const int corner_pre = avail_ul ? T(ox - 1, oy - 1) : mid;
                const int corner = avail_ul ? corner_pre : ((avail_up & 1) ? T(ox, oy - 1) : mid);
                for (int e = t; e < n; e += 64) {
                    const int u = e >> ush;
                    const uint64_t below_up = avail_up & ((1ull << u) - 1), below_le = avail_le & ((1ull << u) - 1);
                    int vu = corner_pre, vl = corner;
                    if ((avail_up >> u) & 1) vu = T(ox + e, oy - 1);
                    else if (below_up)       vu = T(ox + (63 - __clzll((long long)below_up)) * usz + usz - 1, oy - 1);
                    if ((avail_le >> u) & 1) vl = T(ox - 1, oy + e);
                    else if (below_le)       vl = T(ox - 1, oy + (63 - __clzll((long long)below_le)) * usz + usz - 1);
                    nb[c][NB_C0 + 1 + e] = (int16_t)vu;
                    nb[c][NB_C0 - 1 - e] = (int16_t)vl;
                }
                if (t == 0) nb[c][NB_C0] = (int16_t)corner;
            } else {
                // xevd_get_nbr_b (xevd_ipred.c:47-92): unavailable -> mid grey of the luma depth
                for (int e = t; e < n; e += 64) {
                    const int u = e >> ush;
                    nb[c][NB_C0 + 1 + e] = (int16_t)(((avail_up >> u) & 1) ? T(ox + e, oy - 1) : mid);
                    nb[c][NB_C0 - 1 - e] = (int16_t)(((avail_le >> u) & 1) ? T(ox - 1, oy + e) : mid);
                }
                if (t == 0) nb[c][NB_C0] = (int16_t)(avail_ul ? T(ox - 1, oy - 1) : mid);
            }
        }
        wave_lds_sync();
        STAMP(2);

        EipdPlan plan[3];
        if (EIPD) {
            const int mc = mode_c == 0 ? mode_l : (mode_c == 1 ? 2 : mode_c == 2 ? 0 : mode_c == 3 ? 24 : 12);      // xevdm_ipred_uv :267-305
            plan[0] = eipd_plan(nb[0], mode_l, cw, chh, lw, lh, t);
            plan[1] = eipd_plan(nb[1], mc, cw >> 1, chh >> 1, lw - 1, lh - 1, t);
            plan[2] = eipd_plan(nb[2], mc, cw >> 1, chh >> 1, lw - 1, lh - 1, t);
        } else {
#pragma unroll
            for (int c = 0; c < 3; c++) {
                const int mode = c ? mode_c : mode_l;
                const int w = c ? cw >> 1 : cw, h = c ? chh >> 1 : chh;
                if (mode == 0) {                                       // ipred_dc_b: (sum of h left + w up samples + w) >> (log2 w + 1)
                    int acc = 0;
                    for (int e = t; e < w + h; e += 64) acc += e < h ? nb[c][NB_C0 - 1 - e] : nb[c][NB_C0 + 1 + e - h];
                    acc = wave_sum(acc);
                    if (t == 0) nb[c][NB_DC] = (int16_t)((acc + w) >> ((c ? lw - 1 : lw) + 1));
                } else if (mode == 4) {
                    for (int e = t; e < w + h; e += 64) nb[c][NB_UR + e] = (int16_t)((nb[c][NB_C0 + 1 + e] + nb[c][NB_C0 - 1 - e]) >> 1);
                }
            }
        }
        wave_lds_sync();
        STAMP(3);

        int16_t *const py = a.cur_y + cu_y * a.s_l + cu_x, *const ty = tl + (ly0 + 1) * SL + lx0 + 4;
        const int coff_g = (cu_y >> 1) * a.s_c + (cu_x >> 1), coff_t = ((ly0 >> 1) + 1) * SC + (lx0 >> 1) + 4;
        if (EIPD) {
            const int maxc = (1 << a.bd_c) - 1, lsw = lw - 2;
            int ub = 0;                                               // (a scalar loop counter: the trip counts below stay on the scalar unit)
            do {
                const int u0 = ub + t;
                // (no request for the next batch while this one computes, unlike k_intra.hip: a load that may still be in flight at the end of the CU makes the
                //  compiler wait for the memory counter before the pending bits clear - and the counter holds the CU's stores: the round trip this kernel avoids)
                if (ub) fetch_units(u0);
                uint2 cl[4] = { ul[0], ul[1], ul[2], ul[3] };
                uint32_t cq[4] = { uc[0], uc[1], uc[2], uc[3] };
                const int steps = min(4, (nunit - ub + 63) >> 6);
                int k = 0;
#pragma unroll 1
                do {                                                   // (at least one step: the wait for the batch's residual is not skipped on any path)
                    const int u = u0 + 64 * k;
                    if (u < nunit) {
                        const int c = u >= uhalf ? 1 : 0, v = u - (c ? uhalf : 0);
                        const int lx = (u & (scuw - 1)) << 2, ly = u >> lsw, cx = (v & (scuw - 1)) << 1, cy = v >> lsw;
                        const EipdPlan kc = { plan[1].mode, c ? plan[2].p0 : plan[1].p0, c ? plan[2].p1 : plan[1].p1, c ? plan[2].p2 : plan[1].p2 };
                        int pl[4], pc[2];
                        eipd_row<4>(nb[0], plan[0], lx, ly, cw, chh, lw, lh, maxv, pl);
                        eipd_row<2>(nb[1 + c], kc, cx, cy, cw >> 1, chh >> 1, lw - 1, lh - 1, maxc, pc);
                        // also without coefficients: the reference clips the prediction (xevd_recon.c:44-51); the luma depth clips chroma too (:75-90)
                        const uint32_t o0 = recon2i(pack2i(pl[0], pl[1]), (cbf & 1) ? cl[0].x : 0u, maxv), o1 = recon2i(pack2i(pl[2], pl[3]), (cbf & 1) ? cl[0].y : 0u, maxv);
                        const uint32_t o2 = recon2i(pack2i(pc[0], pc[1]), ((cbf >> (1 + c)) & 1) ? cq[0] : 0u, maxv);
                        // local dual tree: a chroma-only CU (flag 32) leaves luma alone, a luma-only one (64) chroma
                        if (!(nflags & 32u)) { *(uint2 *)(ty + ly * SL + lx) = make_uint2(o0, o1); st_coherent2(py + ly * a.s_l + lx, o0, o1); }
                        if (!(nflags & 64u)) { *(uint32_t *)((c ? tv : tu) + coff_t + cy * SC + cx) = o2; st_coherent((c ? a.cur_v : a.cur_u) + coff_g + cy * a.s_c + cx, o2); }
                    }
                    cl[0] = cl[1]; cl[1] = cl[2]; cl[2] = cl[3];
                    cq[0] = cq[1]; cq[1] = cq[2]; cq[2] = cq[3];
                } while (++k < steps);
            } while ((ub += 256) < nunit);
        } else {
            int sb = 0;
            do {                                                       // (do-while and every use of the residual before the first store: see the EIPD branch)
                const int sidx = sb + t, sclamp = min(sidx, nscu - 1);
                {
                    const int lx = (sclamp % scuw) << 2, ly = (sclamp / scuw) << 2;
                    if (sb) fetch_resid(lx, ly);
                    int pl[4][4], pc[2][2][2];
                    int vl[7], vc[2][3];
                    nb_fetch<7>(nb[0], mode_l, lx, ly, vl);
                    nb_fetch<3>(nb[1], mode_c, lx >> 1, ly >> 1, vc[0]);
                    nb_fetch<3>(nb[2], mode_c, lx >> 1, ly >> 1, vc[1]);
#pragma unroll
                    for (int md = 0; md < 5; md++) {                   // uniform: one of the five register shuffles runs
                        if (md == mode_l)
#pragma unroll
                            for (int r = 0; r < 4; r++)
#pragma unroll
                                for (int q = 0; q < 4; q++) pl[r][q] = vl[nb_sel(md, r, q, 3)];
                        if (md == mode_c)
#pragma unroll
                            for (int c = 0; c < 2; c++)
#pragma unroll
                                for (int r = 0; r < 2; r++)
#pragma unroll
                                    for (int q = 0; q < 2; q++) pc[c][r][q] = vc[c][nb_sel(md, r, q, 1)];
                    }
                    uint32_t ol[4][2], oc[2][2];
#pragma unroll
                    for (int r = 0; r < 4; r++) {
                        ol[r][0] = recon2i(pack2i(pl[r][0], pl[r][1]), (cbf & 1) ? rl[r].x : 0u, maxv);
                        ol[r][1] = recon2i(pack2i(pl[r][2], pl[r][3]), (cbf & 1) ? rl[r].y : 0u, maxv);
                    }
#pragma unroll
                    for (int c = 1; c < 3; c++)
#pragma unroll
                        for (int r = 0; r < 2; r++) oc[c - 1][r] = recon2i(pack2i(pc[c - 1][r][0], pc[c - 1][r][1]), ((cbf >> c) & 1) ? rc[c - 1][r] : 0u, maxv);
                    asm volatile("" : "+v"(ol[0][0]), "+v"(ol[0][1]), "+v"(ol[1][0]), "+v"(ol[1][1]), "+v"(ol[2][0]), "+v"(ol[2][1]), "+v"(ol[3][0]), "+v"(ol[3][1]),
                                      "+v"(oc[0][0]), "+v"(oc[0][1]), "+v"(oc[1][0]), "+v"(oc[1][1]) :: "memory");      // (keeps the chroma arithmetic - the last use of the residual - above the luma stores)
                    if (sidx < nscu && !(nflags & 32u))
#pragma unroll
                        for (int r = 0; r < 4; r++) {
                            *(uint2 *)(ty + (ly + r) * SL + lx) = make_uint2(ol[r][0], ol[r][1]);
                            st_coherent2(py + (ly + r) * a.s_l + lx, ol[r][0], ol[r][1]);
                        }
                    if (sidx < nscu && !(nflags & 64u))
#pragma unroll
                        for (int c = 1; c < 3; c++)
#pragma unroll
                            for (int r = 0; r < 2; r++) {
                                *(uint32_t *)((c == 1 ? tu : tv) + coff_t + ((ly >> 1) + r) * SC + (lx >> 1)) = oc[c - 1][r];
                                st_coherent((c == 1 ? a.cur_u : a.cur_v) + coff_g + ((ly >> 1) + r) * a.s_c + (lx >> 1), oc[c - 1][r]);
                            }
                }
            } while ((sb += 64) < nscu);
        }
        STAMP(4);
        // ---- done: the tile holds the CU (the LDS writes above are complete before the bits clear) ----
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");            // (not a workgroup-scope release fence: that waits for the acknowledgement of the global stores too - the round trip this kernel exists to avoid)
        for (int s = t; s < nscu; s += 64) atomicAnd(&s_left[((ly0 >> 2) + (s >> (lw - 2))) * nsc + (lx0 >> 2) + (s & (scuw - 1))], ~own);
        wave_lds_sync();                                               // the wave's next CU reuses the neighbour arrays
        STAMP(5);
    }
#ifdef CTU_PROFILE
    if (tk == CTU_PROFILE && tid == 0) { a.done[gridDim.x + 8 * 64 + 2] = (uint32_t)clock64(); a.done[gridDim.x + 8 * 64 + 3] = (uint32_t)wall_clock64(); a.done[gridDim.x + 8 * 64 + 4] = count; }
#endif
    // ---- publish the CTU: the sc1 stores of every wave have reached the coherence point once vmcnt drains ----
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) __hip_atomic_store(&a.done[tk], a.epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

int launch_intra_ctu(xgpu_ctx *c, const IntraArgs &a, const IntraCtu *ctus, int n_ctus)
{
    const int lc = c->sp.log2_ctu;
    const CtuLds L = ctu_lds(lc);
    const int e = c->sp.tool_eipd ? 1 : 0;
    if (!c->ctu_attr[e]) {
        const hipError_t r = e ? hipFuncSetAttribute((const void *)k_intra_ctu<true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)
                               : hipFuncSetAttribute((const void *)k_intra_ctu<false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (r != hipSuccess) return -1;
        c->ctu_attr[e] = 1;
    }
    const dim3 g((uint32_t)n_ctus), b(64 * CTU_WAVES);
    if (e) hipLaunchKernelGGL((k_intra_ctu<true>), g, b, (size_t)L.total, c->stream, a, ctus, lc, c->sp.width, c->sp.height);
    else   hipLaunchKernelGGL((k_intra_ctu<false>), g, b, (size_t)L.total, c->stream, a, ctus, lc, c->sp.width, c->sp.height);
#ifdef CTU_PROFILE
    static int shots = 0;
    if (++shots == 5) {
        hipStreamSynchronize(c->stream);
        uint32_t h[8 * 64 + 8];
        hipMemcpy(h, a.done + n_ctus, sizeof(h), hipMemcpyDeviceToHost);
        const uint32_t n = h[8 * 64 + 4];
        fprintf(stderr, "CTU %d: %u CUs, %u core clocks = %u ticks of 10 ns\n", CTU_PROFILE, n, h[8 * 64 + 2] - h[8 * 64], h[8 * 64 + 3] - h[8 * 64 + 1]);
        for (uint32_t i = 0; i < n && i < 64; i++)
            fprintf(stderr, "  cu %2u start %7u  wait %6u stage %5u plan %5u predict %5u release %5u\n", i, h[8 * i] - h[8 * 64], h[8 * i + 1] - h[8 * i], h[8 * i + 2] - h[8 * i + 1],
                    h[8 * i + 3] - h[8 * i + 2], h[8 * i + 4] - h[8 * i + 3], h[8 * i + 5] - h[8 * i + 4]);
    }
#endif
    return 0;
}
int intra_ctu_lds_bytes(int log2_ctu) { return ctu_lds(log2_ctu).total; }
