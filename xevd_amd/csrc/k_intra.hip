// k_intra.hip - the order-dependent part of a picture: intra prediction + residual add + clip of the intra CUs, intra block copy, HTDF.
//
// Replaces the intra branch of xevd_recon_unit (src_base/xevd.c:731-741): xevd_get_avail_intra + xevd_get_nbr_b
// (src_base/xevd_ipred.c:33-94), xevd_ipred_b / xevd_ipred_uv_b (:96-164, 587-676) and xevd_recon_yuv.  The Main
// profile runs the same predictors when sps->tool_eipd = 0 (src_main/xevdm.c:1346-1381), on non-square CUs too.
//
// An intra CU predicts from reconstructed samples of CUs decoded before it, so intra CUs form a dependency graph
// on top of the inter CUs (which k_inter finishes first).  The batch builder (host) derives, per intra CU, which
// 4-sample units of the row above / the column to the left exist "already reconstructed" in decode order
// (= the reference's COD flags at that CU's turn), the list of intra CUs it reads from, and sorts the CUs
// topologically (by level = 1 + max level of the intra CUs read).
//
// Two launches per picture: level 1 (no intra neighbour; the bulk of the intra CUs of a P/B picture) as a plain kernel,
// all deeper levels as ONE data-flow kernel (a launch per level would cost ~15 us each, and an all-intra picture has
// ~10^3 levels) - see k_intra<DEP> below.
//
// The same graph carries the other order-dependent Main tools as extra node kinds (template parameters, so pictures without them run the lean
// instantiations): EIPD's 33 luma / 5 chroma predictors (k_intra<., EIPD>), intra block copy (<., ., IBC>: xevdm_IBC_mc, src_main/xevdm_mc.c:2040-2106 -
// the node waits for the CUs under its source block and copies instead of predicting) and HTDF (<., ., ., HTDF>: xevdm_htdf, src_main/xevdm_recon.c:
// 153-385 - a filter stage after the prediction pass, and filter-only nodes for the inter CUs it applies to).
//
// MI355X mapping: one wavefront per CU, eight independent waves per workgroup.  The neighbour arrays of the three components
// are staged once in LDS (unavailable units -> mid grey of the LUMA bit depth, xevd.c:455-473), DC sums are wave
// reductions, then every lane predicts whole 4x4 SCUs (+ their 2x2 chroma blocks) exactly like k_inter's lanes
// reconstruct theirs, so residual addressing and stores are shared idioms (Baseline predictors; the EIPD instantiations work in
// units of one luma row of four + one chroma pair, see intra_body).  A CU of several 64-unit steps sits in the list once per step
// (parts, xgpu_builder.hip).  Integer work bound by the latency of dependent instructions and memory round trips: no MFMA.
#include "xgpu_internal.h"
#include "itdq_body.h"

#include <type_traits>
#include "intra_pred.h"

// waves (= CUs in flight) per workgroup = INTRA_CHUNK list positions per ticket.  Measured at 8K (chunk, waves): (8, 4) 81 us, (4, 4) 72, (2, 2) 83,
// (1, 1) 106, (8, 8) 68, (12, 12) 69, (16, 16) 70: one pass per workgroup (no CU waits behind another one of its workgroup) and few tickets
// (the counter is one contended atomic)
#define INTRA_WAVES 8


// DEP = false: the CUs of level 1 (no intra CU among their neighbours) - independent, plain accesses, static assignment.
// DEP = true : all deeper levels in ONE launch.  Workgroups draw chunks of list positions from a ticket counter, so every
//              lower position is already running or finished whatever order the dispatcher picks; a wave waits on the done
//              flags of the CUs it reads from.  Samples move with sc1 accesses (coherent across the XCD L2s), flags hold the
//              launch's epoch (no reset between pictures).  There are no workgroup barriers: a wave may wait for a flag that
//              another wave of its own workgroup sets.
// HTDF tables (xevdm_recon.c:163-174)
__constant__ uint8_t k_htdf_tbl[5][16] = {
    { 0, 0, 2,  6, 10, 14, 19, 23, 28, 32,  36,  41,  45,  49,  53,  57 },
    { 0, 0, 5, 12, 20, 29, 38, 47, 56, 65,  73,  82,  90,  98, 107, 115 },
    { 0, 0, 1,  4,  9, 16, 24, 32, 41, 50,  59,  68,  77,  86,  94, 103 },
    { 0, 0, 3,  9, 19, 32, 47, 64, 81, 99, 117, 135, 154, 179, 205, 230 },
    { 0, 0, 0,  2,  6, 11, 18, 27, 38, 51,  64,  96, 128, 160, 192, 224 },
};
#define HTDF_EXT (66 * 66)           // a filtered CU is at most 64 x 64 (xevdm_htdf_skip_condition): the block plus one sample of border

// per wave: the neighbour arrays of the prediction pass; the HTDF instantiation lays its 66 x 66 block over them afterwards (70 KB per 8-wave workgroup:
// two workgroups per CU)
template <bool HTDF> struct IntraLds { static constexpr int WAVE = (HTDF && HTDF_EXT > 3 * NB_LEN) ? HTDF_EXT : 3 * NB_LEN; };

// The work of one workgroup of WAVES waves; `block` = its position in the launch (static assignment, DEP = false).  s_nb = WAVES x IntraLds<HTDF>::WAVE samples,
// s_chunk = one dword.
#ifdef INTRA_PROFILE      // cycle stamps of 64 consecutive list positions of the data-flow launch (make EXTRA=-DINTRA_PROFILE=<first position>; printed by launch_intra)
// Measured with them (all-intra 1080p Main, 2.43 GHz): per CU ~4 400 clocks of staging (per component ~1 000 of address arithmetic and load issue, 400 - 1 000 until the loads are
// back, 150 of LDS stores), 1 100 - 1 900 of plan, 2 000 - 3 500 for a step of six samples, 270 - 800 until the stores are acknowledged: a link of a chain is ~9 000 clocks of one
// wave's instruction latency plus ~0.7 us of flag hand-over, not memory round trips.  Tried on top and measured slower: the three components' loads in one batch (bits instead of
// pointers kept alive: stage 5 000 - 7 000), 32-bit availability masks for CUs up to w + h = 128 (issue 1 000 -> 1 200).
__device__ uint32_t g_intra_prof[8 * 64];
#define ISTAMP(k) do { if (DEP && t == 0 && item >= INTRA_PROFILE && item < INTRA_PROFILE + 64) g_intra_prof[8 * (item - INTRA_PROFILE) + (k)] = (uint32_t)clock64(); } while (0)
#else
#define ISTAMP(k)
#endif
// EIPD: 0 Baseline modes, 1 sps->tool_eipd, 2 tool_eipd in a picture with CUs whose RIGHT neighbours are reconstructed first (sps_suco_flag: the right reference
// column and the LR_01 / LR_11 predictor forms; pictures without such CUs keep the smaller instantiation)
template <bool DEP, int EIPD, bool IBC, bool HTDF, int WAVES>
__device__ __forceinline__ void intra_body(const IntraArgs &a, uint32_t block, int16_t *s_nb_, uint32_t *s_chunk)
{
    constexpr int WAVE_LDS = IntraLds<HTDF>::WAVE;
    int16_t (*s_nb)[WAVE_LDS] = (int16_t (*)[WAVE_LDS])s_nb_;
    const int t = threadIdx.x & 63, wv = threadIdx.x >> 6;
    uint32_t chunk = block;
    if (DEP) {
        if (threadIdx.x == 0) *s_chunk = atomicAdd(&a.done[a.n_intra], 1u) - a.ticket_base;
        __syncthreads();                                             // the only workgroup barrier, before any wave can wait
        chunk = *s_chunk;
    }
    int16_t (*nb)[NB_LEN] = (int16_t (*)[NB_LEN])s_nb[wv];
    const int mid = 1 << (a.bd_l - 1);
    const int maxv = (1 << a.bd_l) - 1;

    // one CU per wave and workgroup pass: a ticket (DEP) covers WAVES list positions.  In the data-flow launch a wave then follows its CU's STRAND: the host links a CU
    // whose only unfinished dependency is one CU to that CU (IntraRec.cu = list position of the successor), and the wave that has just published the parent goes on
    // with the child - no flag store to become visible, no poll to see it: a link of a strand costs the store acknowledgement and the neighbour loads (two coherent
    // round trips) instead of four.  Strand members behind the head lie behind the launch's range of the list: no ticket reaches them.
    uint32_t item = a.first + chunk * WAVES + wv;
    if (item >= (uint32_t)(a.first + a.count)) return;
    for (;;) {
        const uint4 *rec = (const uint4 *)&a.list[item];
        const uint4 q0 = rec[0], q1 = rec[1], q2 = rec[2];
        const uint32_t avail_ul = uni(q0.y) & 1;
        // sps_suco_flag: a CU whose RIGHT neighbours are reconstructed first (flag bit 24; bit 23: so is its left side - the two bits are avail_lr) keeps the mask of the
        // right column's units in the upper half of `up` - such a CU lies inside a node of at most 64x64, its masks have at most 24 bits (xgpu_internal.h)
        const int lrf = (int)((uni(q0.y) >> 23) & 3), lr = EIPD == 2 ? lrf : 0;      // (the Baseline predictors have no right-hand form, xevd_ipred.c:95-164,587-622)
        uint32_t up_hi = q0.w;
        asm volatile("" : "+v"(up_hi));                             // (the word is wanted whatever the flags say: left to itself the compiler loads it behind a test of the flags - a second memory round trip in front of every CU of a chain)
        up_hi = uni(up_hi);
        const uint32_t avail_ri = (lrf & 2) ? up_hi : 0u;
        const uint64_t avail_up = (uint64_t)uni(q0.z) | ((lrf & 2) ? 0ull : (uint64_t)up_hi << 32);
        uint64_t avail_le = (uint64_t)uni(q1.x) | ((uint64_t)uni(q1.y) << 32);
        const uint64_t avail_le_raw = avail_le;
        // intra block copy (batches that have such CUs run the IBC instantiation): the record's `le` word carries the block vector, there are no
        // neighbour samples to stage (the masks are empty), and the SCU loop below copies instead of predicting
        const bool ibc_cu = IBC && ((uni(q0.y) >> 1) & 1);
        const int bvx = (int)(int16_t)(avail_le & 0xFFFF), bvy = (int)(int16_t)((avail_le >> 16) & 0xFFFF);
        if (ibc_cu) avail_le = 0;
        const uint32_t nflags = uni(q0.y);
        const bool htdf_cu = HTDF && ((nflags >> 2) & 1), htdf_only = HTDF && ((nflags >> 3) & 1);
        const uint32_t dep_first = uni(q1.z), dep_count = uni(q1.w);
        const uint32_t g = uni(q2.x), m = uni(q2.y), ipm = uni(q2.z), coef_off = uni(q2.w);
        const int cu_x = g & 0xFFFF, cu_y = g >> 16;
        const int lw = m & 0xFF, lh = (m >> 8) & 0xFF, cbf = (m >> 16) & 0xFF;
        const int mode_l = ipm & 0xFF, mode_c = (ipm >> 8) & 0xFF;
        const int cw = 1 << lw, chh = 1 << lh;
        const int scuw = cw >> 2, nscu = scuw * (chh >> 2);
        const int cwc = cw >> 1;
        const uint32_t off_u = coef_off + ((cbf & 1) ? (uint32_t)(cw * chh) : 0u);
        const uint32_t off_v = off_u + ((cbf & 2) ? (uint32_t)(cwc * (chh >> 1)) : 0u);

        // the residual of the lane's first SCU does not depend on any neighbour: fetch it before waiting
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
        // EIPD instantiations work in UNITS instead of SCUs: unit u = the row of four luma samples at (4 * (u % scuw), u / scuw) and one pair of chroma samples
        // (4:2:0: as many pairs as luma rows - U in the first half of the units, V in the second, row-major).  A lane of the SCU mapping evaluates 24 predictor
        // samples one after another - at the ~35 instructions of an angular sample 1.5 us in which a 4x4 CU keeps one lane busy and a 16x16 CU sixteen; a chain of
        // small CUs (an all-intra picture is one) pays that per link.  With units a CU up to 16x16 is one step of six samples per lane, and the loop body holds six
        // inlined predictors instead of 24.  The residual of a unit is contiguous (luma at 4 u, chroma at 2 v): up to four units per lane are requested at once.
        const int nunit = nscu << 2, uhalf = nscu << 1;
        // parts: a large CU sits in the list several times (the host's plan, xgpu_builder.hip: one entry per step of 64 units / SCUs, each with its own done flag); every part
        // stages the neighbours and derives the plan, and reconstructs the units [u_lo, u_hi) / the SCUs [s_lo, s_hi)
        const int part = (int)(m >> 24), nparts = max(1, (int)((ipm >> 16) & 0xFF));
        // (shifts and masks: widths, SCU counts and part counts are powers of two - left as divisions by values only known at run time they were three
        //  reciprocal sequences of ~30 dependent instructions each on the critical path of every link of a chain)
        const int lparts = 31 - __builtin_clz((unsigned)nparts), lscuw = lw - 2;
        const int u_lo = part * (nunit >> lparts), u_hi = u_lo + (nunit >> lparts), s_lo = part * (nscu >> lparts), s_hi = s_lo + (nscu >> lparts);
        uint2 ul[4] = { { 0, 0 }, { 0, 0 }, { 0, 0 }, { 0, 0 } };
        uint32_t uc[4] = { 0, 0, 0, 0 };
        auto fetch_units = [&](int u0) {
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const int u = u0 + 64 * k, c = u >= uhalf ? 1 : 0;
                ul[k] = make_uint2(0, 0); uc[k] = 0;
                if (u < u_hi) {
                    if (cbf & 1) ul[k] = *(const uint2 *)(a.resid + coef_off + 4 * u);
                    if ((cbf >> (1 + c)) & 1) uc[k] = *(const uint32_t *)(a.resid + (c ? off_v : off_u) + 2 * (u - (c ? uhalf : 0)));
                }
            }
        };
        if (EIPD) fetch_units(u_lo + t);
        else if (s_lo + t < s_hi) fetch_resid(((s_lo + t) & (scuw - 1)) << 2, ((s_lo + t) >> lscuw) << 2);

        ISTAMP(0);
        if (DEP) {      // wait until the intra CUs this one reads from have published their samples
            for (uint32_t d = t; d < dep_count; d += 64) {
                const uint32_t j = a.deps[dep_first + d];
                while (__hip_atomic_load(&a.done[j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != a.epoch) __builtin_amdgcn_s_sleep(4);      // (256 clocks between polls: 1 / 4 / 16 measured the same on the bench workloads, all-intra 1080p Main 7.36 / 7.22 / 7.20 ms)
            }
            wave_lds_sync();
            asm volatile("" ::: "memory");
        }

        ISTAMP(1);
        // ---- neighbour staging (xevd_get_nbr_b): sample e of a side belongs to unit e / unit_size.  All loads of the first round
        //      (covers CUs up to w + h = 128) are issued before any LDS store so that they overlap ----
        if (!htdf_only) {      // (a filter-only node - an inter CU that HTDF runs on - predicts nothing: no neighbour arrays, no plan; cycle stamps on the HTDF workload: 3 400 of a node's 12 500 clocks)
        if (EIPD) {
            // xevdm_get_nbr: an unavailable unit repeats the last sample of the nearest available unit before it (the mid value when
            // there is none above; the corner when there is none to the left); every element is one load at a computed position
            // every element is one load at a computed position: the dword that holds it is requested for all elements of a component first (up to four rounds of 64
            // lanes per side), the halves are picked afterwards - a select right behind a load would make every load its own memory round trip
            auto ldd = [&](const int16_t *p) -> uint32_t {
                const uintptr_t q = (uintptr_t)p & ~(uintptr_t)3;
                return DEP ? ld_coherent((const int16_t *)q) : *(const uint32_t *)q;
            };
            auto half = [&](const int16_t *p, uint32_t d) -> int { return (int)(((uintptr_t)p & 2) ? d >> 16 : d & 0xFFFFu); };
#pragma unroll
            for (int c = 0; c < 3; c++) {
                const int16_t *plane = c == 0 ? a.cur_y : (c == 1 ? a.cur_u : a.cur_v);
                const int s = c ? a.s_c : a.s_l, sh = c ? 1 : 0, ush = c ? 1 : 2, usz = c ? 2 : 4;
                const int16_t *org = plane + (cu_y >> sh) * s + (cu_x >> sh);
                const int n = (cw + chh) >> sh;
                const int16_t *pc0 = avail_ul ? org - s - 1 : nullptr, *pc1 = (!avail_ul && (avail_up & 1)) ? org - s : nullptr;
                const uint32_t dc0 = pc0 ? ldd(pc0) : 0u, dc1 = pc1 ? ldd(pc1) : 0u;
                const int16_t *pu[4], *ple[4];
                uint32_t du[4], dle[4];
#pragma unroll
                for (int it = 0; it < 4; it++) {
                    const int e = t + 64 * it, u = e >> ush;
                    pu[it] = ple[it] = nullptr; du[it] = dle[it] = 0;
                    if (e < n) {
                        const uint64_t below_up = avail_up & ((1ull << u) - 1), below_le = avail_le & ((1ull << u) - 1);
                        if ((avail_up >> u) & 1) pu[it] = org - s + e;
                        else if (below_up)       pu[it] = org - s + (63 - __clzll((long long)below_up)) * usz + usz - 1;
                        if ((avail_le >> u) & 1) ple[it] = org + e * s - 1;
                        else if (below_le)       ple[it] = org + ((63 - __clzll((long long)below_le)) * usz + usz - 1) * s - 1;
                        if (pu[it]) du[it] = ldd(pu[it]);
                        if (ple[it]) dle[it] = ldd(ple[it]);
                    }
                }
                uint32_t dc0_ = dc0, dc1_ = dc1;
                if (c == 0) ISTAMP(6);
                if (DEP) asm volatile("" : "+v"(dc0_), "+v"(dc1_), "+v"(du[0]), "+v"(du[1]), "+v"(du[2]), "+v"(du[3]), "+v"(dle[0]), "+v"(dle[1]), "+v"(dle[2]), "+v"(dle[3]));      // (keeps the selects out of the loads' branches)
                const int corner_pre = pc0 ? half(pc0, dc0_) : mid;
                const int corner = pc0 ? corner_pre : (pc1 ? half(pc1, dc1_) : mid);
#ifdef INTRA_PROFILE
                if (c == 0) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); ISTAMP(7); }
#endif
#pragma unroll
                for (int it = 0; it < 4; it++) {
                    const int e = t + 64 * it;
                    if (e < n) {
                        nb[c][NB_C0 + 1 + e] = (int16_t)(pu[it] ? half(pu[it], du[it]) : corner_pre);
                        nb[c][NB_C0 - 1 - e] = (int16_t)(ple[it] ? half(ple[it], dle[it]) : corner);
                    }
                }
                if (t == 0) nb[c][NB_C0] = (int16_t)corner;
                if (lr & 2) {
                    // the right column (xevdm_get_nbr :123-147): unit by unit from the picture where it is reconstructed, else the sample before it; it starts from up[w],
                    // the sample above it - itself a picture sample or a repetition.  (Not on the critical path of ordinary streams: plain code.)
                    const int wc_ = cw >> sh, ue = wc_ >> ush;
                    const uint64_t below_e = avail_up & ((1ull << ue) - 1);
                    const int16_t *pe = ((avail_up >> ue) & 1) ? org - s + wc_ : below_e ? org - s + (63 - __clzll((long long)below_e)) * usz + usz - 1 : nullptr;
                    const int first = pe ? half(pe, ldd(pe)) : corner_pre;
#pragma unroll
                    for (int it = 0; it < 2; it++) {
                        const int e = t + 64 * it, u = e >> ush;
                        if (e < n) {
                            const uint32_t below = avail_ri & ((1u << u) - 1u);
                            const int16_t *pr = ((avail_ri >> u) & 1) ? org + e * s + wc_ : below ? org + ((31 - __clz((int)below)) * usz + usz - 1) * s + wc_ : nullptr;
                            nb[c][NB_UR + 1 + e] = (int16_t)(pr ? half(pr, ldd(pr)) : first);
                        }
                    }
                    if (t == 0) nb[c][NB_UR] = (int16_t)first;
                }
            }
        } else {
        // (In the data-flow launch the loads are coherent dword loads whose wanted half is picked AFTER every load of the round is out: a shift or select right
        //  behind a load makes the compiler wait for it on the spot, and these loads are memory round trips - nine in a row was most of a dependency hop.)
        uint32_t v_up[3], v_le[3][2], v_ul[3];
        bool h_le[3][2], h_ul[3];
#pragma unroll
        for (int c = 0; c < 3; c++) {
            const int16_t *plane = c == 0 ? a.cur_y : (c == 1 ? a.cur_u : a.cur_v);
            const int s = c ? a.s_c : a.s_l, sh = c ? 1 : 0, ush = c ? 1 : 2;
            const int16_t *org = plane + (cu_y >> sh) * s + (cu_x >> sh);
            const int n = (cw + chh) >> sh;
            const int e = 2 * t;
            if (DEP) {
                // the data-flow launch: every position of the arrays lies inside the padded picture, so the loads are unconditional (a position beyond the side's
                // length repeats its last one) and availability is a select - twelve branches less on the path of every link of a chain: both intra launches 27.0 -> 25.6 us at 1080p,
                // 33.4 -> 32.5 at 4K, nothing at 8K (tools/archive/r5_w.sh)
                const int ec = min(e, n - 2), e0 = min(t, n - 1), e1 = min(t + 64, n - 1);
                const uint32_t ru = ld_coherent(org - s + ec), r0 = ld_coherent(org + e0 * s - 2), r1 = ld_coherent(org + e1 * s - 2), rc = ld_coherent(org - s - 2);
                v_up[c] = (e < n && ((avail_up >> (e >> ush)) & 1)) ? ru : (uint32_t)mid * 0x10001u;
                h_le[c][0] = t < n && ((avail_le >> (e0 >> ush)) & 1); v_le[c][0] = r0;
                h_le[c][1] = t + 64 < n && ((avail_le >> (e1 >> ush)) & 1); v_le[c][1] = r1;
                h_ul[c] = t == 0 && avail_ul; v_ul[c] = rc;
                continue;
            }
            v_up[c] = (uint32_t)mid * 0x10001u;
            if (e < n && ((avail_up >> (e >> ush)) & 1)) v_up[c] = DEP ? ld_coherent(org - s + e) : *(const uint32_t *)(org - s + e);
#pragma unroll
            for (int k = 0; k < 2; k++) {
                const int el = t + 64 * k;
                v_le[c][k] = 0;
                h_le[c][k] = el < n && ((avail_le >> (el >> ush)) & 1);
                if (h_le[c][k]) v_le[c][k] = DEP ? ld_coherent(org + el * s - 2) : ((uint32_t)(uint16_t)org[el * s - 1] << 16);      // the sample = the dword's high half
            }
            v_ul[c] = 0;
            h_ul[c] = t == 0 && avail_ul;
            if (h_ul[c]) v_ul[c] = DEP ? ld_coherent(org - s - 2) : ((uint32_t)(uint16_t)org[-s - 1] << 16);
        }
        // (opaque to the optimiser: without it the shifts below are sunk back into the branches of the loads - one wait per load again)
#pragma unroll
        for (int c = 0; c < 3; c++) if (DEP) asm volatile("" : "+v"(v_up[c]), "+v"(v_le[c][0]), "+v"(v_le[c][1]), "+v"(v_ul[c]));      // (the plain loads of the level-1 launch are the compiler's to schedule: 23.5 us without, 27.8 with the fence)
#pragma unroll
        for (int c = 0; c < 3; c++) {
            const int sh = c ? 1 : 0, n = (cw + chh) >> sh;
            if (2 * t < n) *(uint32_t *)&nb[c][NB_C0 + 1 + 2 * t] = v_up[c];
            if (t < n) nb[c][NB_C0 - 1 - t] = (int16_t)(h_le[c][0] ? v_le[c][0] >> 16 : (uint32_t)mid);
            if (t + 64 < n) nb[c][NB_C0 - 1 - (t + 64)] = (int16_t)(h_le[c][1] ? v_le[c][1] >> 16 : (uint32_t)mid);
            if (t == 0) nb[c][NB_C0] = (int16_t)(h_ul[c] ? v_ul[c] >> 16 : (uint32_t)mid);
        }
        if (cw + chh > 128) {                                        // the rest of the largest CUs (luma only can exceed one round)
            const int16_t *org = a.cur_y + cu_y * a.s_l + cu_x;
            const int n = cw + chh;
            for (int e = 2 * t + 128; e < n; e += 128)
                *(uint32_t *)&nb[0][NB_C0 + 1 + e] = ((avail_up >> (e >> 2)) & 1) ? (DEP ? ld_coherent(org - a.s_l + e) : *(const uint32_t *)(org - a.s_l + e))
                                                                            : ((uint32_t)mid * 0x10001u);
            for (int e = t + 128; e < n; e += 64)
                nb[0][NB_C0 - 1 - e] = ((avail_le >> (e >> 2)) & 1) ? (DEP ? (int16_t)(ld_coherent(org + e * a.s_l - 2) >> 16) : org[e * a.s_l - 1]) : (int16_t)mid;
        }
        }
        }
        wave_lds_sync();
        ISTAMP(2);

        EipdPlan plan[3];
        if (!htdf_only) {
        if (EIPD) {
            // chroma mode -> the luma-numbered predictor (xevdm_ipred_uv :267-305): DM follows the luma mode, then BI / DC / HOR / VER
            const int mc = mode_c == 0 ? mode_l : (mode_c == 1 ? 2 : mode_c == 2 ? 0 : mode_c == 3 ? 24 : 12);
            plan[0] = eipd_plan(nb[0], mode_l, cw, chh, lw, lh, t, lr);
            plan[1] = eipd_plan(nb[1], mc, cw >> 1, chh >> 1, lw - 1, lh - 1, t, lr);
            plan[2] = eipd_plan(nb[2], mc, cw >> 1, chh >> 1, lw - 1, lh - 1, t, lr);
        } else {
        // ---- DC values (ipred_dc_b): (sum of h left + w up samples + w) >> (log2 w + 1) ----
#pragma unroll
        for (int c = 0; c < 3; c++) {
            const int mode = c ? mode_c : mode_l;                    // wave-uniform
            const int w = c ? cw >> 1 : cw, h = c ? chh >> 1 : chh;
            if (mode == 0) {
                int acc = 0;
                for (int e = t; e < w + h; e += 64) acc += e < h ? nb[c][NB_C0 - 1 - e] : nb[c][NB_C0 + 1 + e - h];
                acc = wave_sum(acc);
                if (t == 0) nb[c][NB_DC] = (int16_t)((acc + w) >> ((c ? lw - 1 : lw) + 1));
            } else if (mode == 4) {
                for (int e = t; e < w + h; e += 64) nb[c][NB_UR + e] = (int16_t)((nb[c][NB_C0 + 1 + e] + nb[c][NB_C0 - 1 - e]) >> 1);
            }
        }
        }
        }
        wave_lds_sync();
        ISTAMP(3);
        // ---- prediction + reconstruction, one 4x4 SCU per lane and step ----
        if (EIPD) {
        const int maxc = (1 << a.bd_c) - 1, lsw = lw - 2;
        for (int ub = u_lo; ub < (htdf_only ? 0 : u_hi); ub += 256) {      // (a scalar loop counter: the trip counts below stay on the scalar unit)
            const int u0 = ub + t;
            uint2 cl[4] = { ul[0], ul[1], ul[2], ul[3] };
            uint32_t cc[4] = { uc[0], uc[1], uc[2], uc[3] };
            if (ub + 256 < u_hi) fetch_units(u0 + 256);
            const int steps = min(4, (u_hi - ub + 63) >> 6);
#pragma unroll 1
            for (int k = 0; k < steps; k++) {
                const int u = u0 + 64 * k;
                if (u < u_hi) {
                    const int c = u >= uhalf ? 1 : 0, v = u - (c ? uhalf : 0);
                    const int lx = (u & (scuw - 1)) << 2, ly = u >> lsw, cx = (v & (scuw - 1)) << 1, cy = v >> lsw;
                    int pl[4], pc[2];
                    if (IBC && ibc_cu) {
                        // xevdm_IBC_mc (xevdm_mc.c:2040-2106): the block at the whole-sample vector in the current picture, chroma at the halved vector
                        auto ldw = [&](const int16_t *p) -> uint32_t { return DEP ? ld_coherent(p) : *(const uint32_t *)p; };
                        const int sxl = cu_x + lx + bvx, ol_ = sxl & 1, sxc = (cu_x >> 1) + cx + (bvx >> 1), oc_ = sxc & 1;
                        const int16_t *src = a.cur_y + (cu_y + ly + bvy) * a.s_l + (sxl - ol_);
                        const int16_t *sc = (c ? a.cur_v : a.cur_u) + ((cu_y >> 1) + cy + (bvy >> 1)) * a.s_c + (sxc - oc_);
                        const uint32_t d0 = ldw(src), d1 = ldw(src + 2), d2 = ldw(src + 4), e0 = ldw(sc), e1 = ldw(sc + 2);      // (the third dword only serves an odd vector; it lies inside the padded picture)
                        const uint32_t o0 = ol_ ? (d0 >> 16) | (d1 << 16) : d0, o1 = ol_ ? (d1 >> 16) | (d2 << 16) : d1, oc0 = oc_ ? (e0 >> 16) | (e1 << 16) : e0;
                        pl[0] = (int)(o0 & 0xFFFF); pl[1] = (int)(o0 >> 16); pl[2] = (int)(o1 & 0xFFFF); pl[3] = (int)(o1 >> 16);
                        pc[0] = (int)(oc0 & 0xFFFF); pc[1] = (int)(oc0 >> 16);
                    } else {
                        const EipdPlan kc = { plan[1].mode, c ? plan[2].p0 : plan[1].p0, c ? plan[2].p1 : plan[1].p1, c ? plan[2].p2 : plan[1].p2, plan[1].lr };
                        eipd_row<4>(nb[0], plan[0], lx, ly, cw, chh, lw, lh, maxv, pl);
                        eipd_row<2>(nb[1 + c], kc, cx, cy, cw >> 1, chh >> 1, lw - 1, lh - 1, maxc, pc);
                    }
                    // also without coefficients: the reference clips the prediction (xevd_recon.c:44-51); the luma depth clips chroma too (:75-90)
                    const uint32_t o0 = recon2i(pack2i(pl[0], pl[1]), (cbf & 1) ? cl[0].x : 0u, maxv), o1 = recon2i(pack2i(pl[2], pl[3]), (cbf & 1) ? cl[0].y : 0u, maxv);
                    const uint32_t o2 = recon2i(pack2i(pc[0], pc[1]), ((cbf >> (1 + c)) & 1) ? cc[0] : 0u, maxv);
                    int16_t *dy = a.cur_y + (cu_y + ly) * a.s_l + cu_x + lx, *dc = (c ? a.cur_v : a.cur_u) + ((cu_y >> 1) + cy) * a.s_c + (cu_x >> 1) + cx;
                    // local dual tree: a chroma-only CU (flag 32) leaves luma alone, a luma-only one (64) chroma
                    if (!(nflags & 32u)) { if (DEP) st_coherent2(dy, o0, o1); else *(uint2 *)dy = make_uint2(o0, o1); }
                    if (!(nflags & 64u)) { if (DEP) st_coherent(dc, o2); else *(uint32_t *)dc = o2; }
                }
                cl[0] = cl[1]; cl[1] = cl[2]; cl[2] = cl[3];
                cc[0] = cc[1]; cc[1] = cc[2]; cc[2] = cc[3];
            }
        }
        } else
        for (int sidx = s_lo + t; sidx < (htdf_only ? 0 : s_hi); sidx += 64) {
            const int lx = (sidx & (scuw - 1)) << 2, ly = (sidx >> lscuw) << 2;
            const int x = cu_x + lx, y = cu_y + ly;
            // a CU above 32x32 takes several rounds: the residual of the NEXT round is requested before this round's arithmetic, so that a round does not
            // start with a memory round trip (the level-1 launch is as long as its 64x64 CUs take)
            const uint2 rl_cur[4] = { rl[0], rl[1], rl[2], rl[3] };
            const uint32_t rc_cur[2][2] = { { rc[0][0], rc[0][1] }, { rc[1][0], rc[1][1] } };
            if (sidx + 64 < s_hi) fetch_resid(((sidx + 64) & (scuw - 1)) << 2, ((sidx + 64) >> lscuw) << 2);
            int pl[4][4], pc[2][2][2];
            if (IBC && ibc_cu) {
                // xevdm_IBC_mc (xevdm_mc.c:2040-2106): the block at the whole-sample vector in the current picture, chroma at the halved vector.
                // Aligned dword loads (coherent ones inside the data-flow launch: the source may have been written by this launch) + a parity shift
                auto ldw = [&](const int16_t *p) -> uint32_t { return DEP ? ld_coherent(p) : *(const uint32_t *)p; };
                const int sxl = x + bvx, syl = y + bvy, ol_ = sxl & 1;
                const int16_t *src = a.cur_y + syl * a.s_l + (sxl - ol_);
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    const uint32_t d0 = ldw(src + r * a.s_l), d1 = ldw(src + r * a.s_l + 2), d2 = ldw(src + r * a.s_l + 4);      // (the third dword is only used for an odd vector; loading it unconditionally keeps the four rows' loads in flight together - it lies inside the padded picture)
                    const uint32_t o0 = ol_ ? (d0 >> 16) | (d1 << 16) : d0, o1 = ol_ ? (d1 >> 16) | (d2 << 16) : d1;
                    pl[r][0] = (int)(o0 & 0xFFFF); pl[r][1] = (int)(o0 >> 16); pl[r][2] = (int)(o1 & 0xFFFF); pl[r][3] = (int)(o1 >> 16);
                }
                const int sxc = (x >> 1) + (bvx >> 1), syc = (y >> 1) + (bvy >> 1), oc_ = sxc & 1;
#pragma unroll
                for (int c = 0; c < 2; c++) {
                    const int16_t *sc = (c == 0 ? a.cur_u : a.cur_v) + syc * a.s_c + (sxc - oc_);
#pragma unroll
                    for (int r = 0; r < 2; r++) {
                        const uint32_t d0 = ldw(sc + r * a.s_c), d1 = ldw(sc + r * a.s_c + 2);
                        const uint32_t o0 = oc_ ? (d0 >> 16) | (d1 << 16) : d0;
                        pc[c][r][0] = (int)(o0 & 0xFFFF); pc[c][r][1] = (int)(o0 >> 16);
                    }
                }
            } else {
            // all LDS reads of the SCU first (one wait), then the arithmetic, then the stores
            int vl[7], vc[2][3];
            nb_fetch<7>(nb[0], mode_l, lx, ly, vl);
            nb_fetch<3>(nb[1], mode_c, lx >> 1, ly >> 1, vc[0]);
            nb_fetch<3>(nb[2], mode_c, lx >> 1, ly >> 1, vc[1]);
#pragma unroll
            for (int md = 0; md < 5; md++) {                         // uniform: one of the five register shuffles runs
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
            }
            int16_t *dy = a.cur_y + y * a.s_l + x;
            uint32_t ol[4][2], oc[2][2];
#pragma unroll
            for (int r = 0; r < 4; r++) {
                ol[r][0] = pack2i(pl[r][0], pl[r][1]); ol[r][1] = pack2i(pl[r][2], pl[r][3]);
                // also without coefficients: the reference clips the prediction (xevd_recon.c:44-51) - the DC of a 4x8 / 8x4 block next to an unavailable side
                // (mid-grey neighbours, sum of 12 samples shifted by 3) leaves the sample range
                ol[r][0] = recon2i(ol[r][0], (cbf & 1) ? rl_cur[r].x : 0u, maxv); ol[r][1] = recon2i(ol[r][1], (cbf & 1) ? rl_cur[r].y : 0u, maxv);
            }
#pragma unroll
            for (int c = 1; c < 3; c++)
#pragma unroll
                for (int r = 0; r < 2; r++) {
                    oc[c - 1][r] = pack2i(pc[c - 1][r][0], pc[c - 1][r][1]);
                    oc[c - 1][r] = recon2i(oc[c - 1][r], ((cbf >> c) & 1) ? rc_cur[c - 1][r] : 0u, maxv);   // the luma depth clips chroma too (xevd_recon.c:75-90)
                }
            const int coff = (y >> 1) * a.s_c + (x >> 1);
            // local dual tree: a chroma-only CU (flag 32) leaves luma alone, a luma-only one (64) chroma - what was computed for the missing plane is dropped
            if (!(nflags & 32u))
#pragma unroll
            for (int r = 0; r < 4; r++) {
                if (DEP) st_coherent2(dy + r * a.s_l, ol[r][0], ol[r][1]); else *(uint2 *)(dy + r * a.s_l) = make_uint2(ol[r][0], ol[r][1]);
            }
            if (!(nflags & 64u))
#pragma unroll
            for (int c = 1; c < 3; c++) {
                int16_t *d = (c == 1 ? a.cur_u : a.cur_v) + coff;
#pragma unroll
                for (int r = 0; r < 2; r++)
                    if (DEP) st_coherent(d + r * a.s_c, oc[c - 1][r]); else *(uint32_t *)(d + r * a.s_c) = oc[c - 1][r];
            }
        }
        if (HTDF && htdf_cu) {
            // ---- HTDF (xevdm_htdf, xevdm_recon.c:153-385): the CU's luma block with one sample of border in LDS, then every sample from the four
            //      2x2 windows that hold it: Hadamard, table on the three AC terms, back, (sum of the four quarters + 2) >> 2 ----
            const uint32_t av = (nflags >> 8) & 0x1FF;
            const bool cmask = (nflags >> 4) & 1;                     // constrained intra prediction: border units only from intra neighbours
            const int tidx = (nflags >> 20) & 7;
            const int thr_log2 = tidx == 0 ? 6 : (tidx < 3 ? 7 : 8), shift = thr_log2 - 4, rnd = (1 << shift) >> 1, thr = (1 << thr_log2) - (1 << shift);
            int16_t *tb = s_nb[wv];
            wave_lds_sync();                                          // the prediction pass is done with the neighbour arrays
            const int we = cw + 2, he = chh + 2;
            int16_t *org = a.cur_y + cu_y * a.s_l + cu_x;
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");           // the wave's own stores of the prediction pass
            auto ldw = [&](const int16_t *p) -> uint32_t { return DEP ? ld_coherent(p) : *(const uint32_t *)p; };
            // Every global load of the stage is issued before the first LDS store that needs one: the loop this replaces waited for each round of 64 dwords before it
            // asked for the next one - a 32x32 block was eight memory round trips (~2 us each for the coherent loads of the data-flow launch), the border four more,
            // which is what a level of the HTDF workload cost (15 us; 26 levels at 8K).  A filtered CU is at most 64 samples wide and high: one lane per border sample.
            auto ldr = [&](const int16_t *p) -> uint32_t { return ldw((const int16_t *)((uintptr_t)p & ~(uintptr_t)3)); };      // the dword that holds the sample
            auto pick = [&](const int16_t *p, uint32_t d) -> int16_t { return (int16_t)(((uintptr_t)p & 2) ? d >> 16 : d & 0xFFFFu); };
            const int16_t *p_l = nullptr, *p_r = nullptr, *p_u = nullptr, *p_d = nullptr, *p_c = nullptr;
            uint32_t d_l = 0, d_r = 0, d_u = 0, d_d = 0, d_c = 0;
            if (t < chh) {
                const bool ok_l = ((av >> 1) & 1) && (!cmask || ((avail_le_raw >> (t >> 2)) & 1));
                const bool ok_r = ((av >> 3) & 1) && (!cmask || ((avail_ri >> (t >> 2)) & 1));
                p_l = org + t * a.s_l + (ok_l ? -1 : 0); p_r = org + t * a.s_l + (ok_r ? cw : cw - 1);
                d_l = ldr(p_l); d_r = ldr(p_r);
            }
            if (t < cw) {
                const bool ok_u = (av & 1) && (!cmask || ((avail_up >> (t >> 2)) & 1));
                p_u = org + t - (ok_u ? a.s_l : 0); p_d = org + (chh - 1) * a.s_l + t;
                d_u = ldr(p_u); d_d = ldr(p_d);
            }
            if (t < 4) {
                p_c = t == 0 ? (((av >> 5) & 1) ? org - 1 - a.s_l : org) : t == 1 ? (((av >> 6) & 1) ? org + cw - a.s_l : org + cw - 1)
                    : t == 2 ? (((av >> 7) & 1) ? org - 1 + chh * a.s_l : org + (chh - 1) * a.s_l) : (((av >> 8) & 1) ? org + cw + chh * a.s_l : org + cw - 1 + (chh - 1) * a.s_l);
                d_c = ldr(p_c);
            }
            const int n2 = (cw >> 1) * chh;
            ISTAMP(6);
            for (int i0 = 0; i0 < n2; i0 += 64 * 8) {
                uint32_t d[8];
#pragma unroll
                for (int q = 0; q < 8; q++) {
                    const int i = i0 + q * 64 + t;
                    d[q] = 0;
                    if (i < n2) d[q] = ldw(org + (i >> (lw - 1)) * a.s_l + ((i & ((cw >> 1) - 1)) << 1));
                }
#pragma unroll
                for (int q = 0; q < 8; q++) {
                    const int i = i0 + q * 64 + t, r = i >> (lw - 1), c = (i & ((cw >> 1) - 1)) << 1;
                    if (i < n2) { tb[(r + 1) * we + c + 1] = (int16_t)(d[q] & 0xFFFF); tb[(r + 1) * we + c + 2] = (int16_t)(d[q] >> 16); }
                }
            }
            asm volatile("" : "+v"(d_l), "+v"(d_r), "+v"(d_u), "+v"(d_d), "+v"(d_c));      // (keeps the picks out of the loads' branches)
            if (t < chh) { tb[(t + 1) * we] = pick(p_l, d_l); tb[(t + 1) * we + we - 1] = pick(p_r, d_r); }
            if (t < cw) { tb[t + 1] = pick(p_u, d_u); tb[(he - 1) * we + t + 1] = pick(p_d, d_d); }
            if (t < 4) tb[t == 0 ? 0 : t == 1 ? we - 1 : t == 2 ? we * (he - 1) : we - 1 + we * (he - 1)] = pick(p_c, d_c);
            wave_lds_sync();
            ISTAMP(7);
            // the 16-byte table in four scalar registers, an entry picked with two byte permutes: 18 look-ups per pair of samples were 18 LDS reads in the dependent chain
            // of a wave that mostly runs alone
            const uint32_t *tb32 = (const uint32_t *)k_htdf_tbl[tidx];
            const uint32_t tq0 = uni(tb32[0]), tq1 = uni(tb32[1]), tq2 = uni(tb32[2]), tq3 = uni(tb32[3]);
            auto lutf = [&](int z) -> int {                           // read_table (:176-189)
                const int ab = z < 0 ? -z : z;
                const uint32_t idx = (uint32_t)(((ab + rnd) & thr) >> shift), sel = (idx & 7u) | 0x0C0C0C00u;
                const int e = (int)((idx & 8u) ? __builtin_amdgcn_perm(tq3, tq2, sel) : __builtin_amdgcn_perm(tq1, tq0, sel));
                const int v = ab < thr ? e : ab;
                return z < 0 ? -v : v;
            };
            // a lane filters a tile of TR x TC samples from the (TR + 1) x (TC + 1) windows over the (TR + 2) x (TC + 2) samples around them (a sample lies in four windows):
            // 2 x 2 = 2.25 windows per sample (one pair per lane needed three), 4 x 4 = 1.56 - for the blocks of 1024 samples and more (32 x 32, 64 x 16 ...: four rounds of 2 x 2 tiles at ~390 instructions against one of 4 x 4 at ~1 100): their waves
            // run several rounds, and a level of the graph lasts as long as its largest node
            auto filter_tiles = [&](auto TRc, auto TCc) {
                constexpr int TR = decltype(TRc)::value, TC = decltype(TCc)::value, LR = TR == 4 ? 2 : 1, LC = TC == 4 ? 2 : 1;
                const int tiles_x = cw >> LC, ltx = lw - LC;
                for (int i = t; i < tiles_x * (chh >> LR); i += 64) {
                    const int r = (i >> ltx) << LR, c = (i & (tiles_x - 1)) << LC;          // output samples (r .. r + TR - 1, c .. c + TC - 1) = ext (r + 1 .., c + 1 ..)
                    int p[TR + 2][TC + 2];
#pragma unroll
                    for (int rr = 0; rr < TR + 2; rr++)
#pragma unroll
                        for (int cc = 0; cc < TC + 2; cc += 2) {
                            const uint32_t d = *(const uint32_t *)&tb[(r + rr) * we + c + cc];
                            p[rr][cc] = (int)(int16_t)(d & 0xFFFF); p[rr][cc + 1] = (int)(int16_t)(d >> 16);
                        }
                    int acc[TR][TC];
#pragma unroll
                    for (int dr = 0; dr < TR; dr++)
#pragma unroll
                        for (int dc = 0; dc < TC; dc++) acc[dr][dc] = 0;
#pragma unroll
                    for (int wr = 0; wr < TR + 1; wr++)
#pragma unroll
                        for (int wc = 0; wc < TC + 1; wc++) {              // window with its top-left sample at ext (r + wr, c + wc)
                            const int x0 = p[wr][wc], x1 = p[wr][wc + 1], x2 = p[wr + 1][wc], x3 = p[wr + 1][wc + 1];
                            const int y0 = x0 + x2, y1 = x1 + x3, y2 = x0 - x2, y3 = x1 - x3;
                            const int z0 = y0 + y1, z1 = lutf(y0 - y1), z2 = lutf(y2 + y3), z3 = lutf(y2 - y3);
                            const int i0_ = z0 + z2, i1_ = z1 + z3, i2_ = z0 - z2, i3_ = z1 - z3;
                            // the output sample (dr, dc) = ext (r + 1 + dr, c + 1 + dc) sits in this window at row 1 + dr - wr, column 1 + dc - wc
#pragma unroll
                            for (int row = 0; row < 2; row++)
#pragma unroll
                                for (int col = 0; col < 2; col++) {
                                    const int dr = wr - 1 + row, dc = wc - 1 + col;
                                    if (dr < 0 || dr >= TR || dc < 0 || dc >= TC) continue;
                                    const int v = row == 1 ? (col == 0 ? i2_ + i3_ : i2_ - i3_) : (col == 0 ? i0_ + i1_ : i0_ - i1_);
                                    acc[dr][dc] += v >> 2;
                                }
                        }
#pragma unroll
                    for (int dr = 0; dr < TR; dr++)
#pragma unroll
                        for (int dc = 0; dc < TC; dc += 2) {
                            const int o0 = clip3i(0, maxv, ((int)(int16_t)acc[dr][dc] + 2) >> 2), o1 = clip3i(0, maxv, ((int)(int16_t)acc[dr][dc + 1] + 2) >> 2);
                            if (DEP) st_coherent(org + (r + dr) * a.s_l + c + dc, pack2i(o0, o1)); else *(uint32_t *)(org + (r + dr) * a.s_l + c + dc) = pack2i(o0, o1);
                        }
                }
            };
            if (cw * chh >= 1024) filter_tiles(std::integral_constant<int, 4>{}, std::integral_constant<int, 4>{});
            else                 filter_tiles(std::integral_constant<int, 2>{}, std::integral_constant<int, 2>{});
        }
        ISTAMP(4);
        if (DEP) {      // publish: the wave's sc1 stores have reached the coherence point once vmcnt drains; then the done flag
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            ISTAMP(5);
            if (t == 0) __hip_atomic_store(&a.done[item], a.epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        wave_lds_sync();                                             // the next CU of this wave reuses the LDS arrays
        if (!DEP) break;
        item = uni(q0.x);                                            // the strand's next CU
        if (item == 0xFFFFFFFFu) break;
    }
}

// ---------------------------------------------------------------------------------------------------------
// Level 1, Baseline predictors, CUs of at most 16 SCUs (any shape from 4x4 to 16x16, 32x8, 64x4): SIXTEEN LANES per CU, four CUs per wave.
// With a wave per CU the level-1 launch of an 8K P/B picture was 15.7 thousand waves for CUs of which most use one to four lanes - 2.6 rounds of the machine's wave
// slots, each the length of a wave's three dependent memory round trips (record -> neighbours + residual -> stores).  The list is sorted by size inside a level, so
// these CUs are the END of the level's range (IntraArgs.n_small, counted by the host's plan): the launch gives the large CUs a wave each as before and the small ones a
// row of 16 lanes each.  Everything that is wave-uniform in intra_body is uniform per row here; the DC sum is a DPP row reduction; the neighbour arrays are the
// same diagonal-axis layout in a compact form (at most 68 samples per side).  No lane of a row waits for another row: LDS traffic of a wave is in order.
// ---------------------------------------------------------------------------------------------------------
#define SB_C0 71             // odd, like NB_C0: pairs of up[] are 4-byte aligned
#define SB_UR 144
#define SB_LEN (SB_UR + 72)
#define SMALL_GROUPS (4 * INTRA_WAVES)      // CUs per workgroup
__device__ __forceinline__ void intra_small_body(const IntraArgs &a, uint32_t block, int16_t *s_nb_)
{
    int16_t (*s_nb)[3][SB_LEN] = (int16_t (*)[3][SB_LEN])s_nb_;
    const int g = threadIdx.x >> 4, t = threadIdx.x & 15;
    const uint32_t k = block * SMALL_GROUPS + (uint32_t)g;
    if (k >= (uint32_t)a.n_small) return;
    const uint32_t item = (uint32_t)(a.first + a.count - a.n_small) + k;
    int16_t (*nb)[SB_LEN] = s_nb[g];
    const int mid = 1 << (a.bd_l - 1), maxv = (1 << a.bd_l) - 1;
    const uint4 *rec = (const uint4 *)&a.list[item];
    const uint4 q0 = rec[0], q1 = rec[1], q2 = rec[2];
    const uint32_t nflags = q0.y, avail_ul = nflags & 1;
    const int lrf = (int)((nflags >> 23) & 3);
    const uint64_t avail_up = (uint64_t)q0.z | ((lrf & 2) ? 0ull : (uint64_t)q0.w << 32);      // (the Baseline predictors have no right-hand form: a right mask in the word is not the up mask's upper half)
    const uint64_t avail_le = (uint64_t)q1.x | ((uint64_t)q1.y << 32);
    const uint32_t gg = q2.x, m = q2.y, ipm = q2.z, coef_off = q2.w;
    const int cu_x = gg & 0xFFFF, cu_y = gg >> 16;
    const int lw = m & 0xFF, lh = (m >> 8) & 0xFF, cbf = (m >> 16) & 0xFF;
    const int mode_l = ipm & 0xFF, mode_c = (ipm >> 8) & 0xFF;
    const int cw = 1 << lw, chh = 1 << lh, scuw = cw >> 2, nscu = scuw * (chh >> 2), cwc = cw >> 1;
    const uint32_t off_u = coef_off + ((cbf & 1) ? (uint32_t)(cw * chh) : 0u);
    const uint32_t off_v = off_u + ((cbf & 2) ? (uint32_t)(cwc * (chh >> 1)) : 0u);
    const bool has = t < nscu;
    const int lx = (t & (scuw - 1)) << 2, ly = (t >> (lw - 2)) << 2;

    // the residual of the lane's SCU depends on no neighbour: requested first
    uint2 rl[4] = { { 0, 0 }, { 0, 0 }, { 0, 0 }, { 0, 0 } };
    uint32_t rc[2][2] = { { 0, 0 }, { 0, 0 } };
    if (has) {
        if (cbf & 1)
#pragma unroll
            for (int r = 0; r < 4; r++) rl[r] = *(const uint2 *)(a.resid + coef_off + (ly + r) * cw + lx);
#pragma unroll
        for (int c = 1; c < 3; c++)
            if ((cbf >> c) & 1)
#pragma unroll
                for (int r = 0; r < 2; r++) rc[c - 1][r] = *(const uint32_t *)(a.resid + (c == 1 ? off_u : off_v) + ((ly >> 1) + r) * cwc + (lx >> 1));
    }

    // ---- neighbour staging (xevd_get_nbr_b): every position of the arrays lies inside the padded picture, so the loads are unconditional (an element beyond the
    //      side's length repeats its last one: same address, same request) and availability is a select afterwards - no branch around a load, all of them in flight
    //      together.  One round covers a side of 32 samples (up: a dword per lane; left: two samples per lane); the wide shapes (32x8, 64x4) take a second pass. ----
#pragma unroll
    for (int c = 0; c < 3; c++) {
        const int16_t *plane = c == 0 ? a.cur_y : (c == 1 ? a.cur_u : a.cur_v);
        const int s = c ? a.s_c : a.s_l, sh = c ? 1 : 0, ush = c ? 1 : 2;
        const int16_t *org = plane + (cu_y >> sh) * s + (cu_x >> sh);
        const int n = (cw + chh) >> sh;
        const int eu = min(2 * t, n - 2), e0 = min(t, n - 1), e1 = min(t + 16, n - 1);
        const uint32_t vu = *(const uint32_t *)(org - s + eu);
        const uint32_t v0 = (uint16_t)org[e0 * s - 1], v1 = (uint16_t)org[e1 * s - 1], vc = (uint16_t)org[-s - 1];
        const uint32_t midp = (uint32_t)mid * 0x10001u;
        if (2 * t < n) *(uint32_t *)&nb[c][SB_C0 + 1 + 2 * t] = ((avail_up >> (eu >> ush)) & 1) ? vu : midp;
        if (t < n) nb[c][SB_C0 - 1 - t] = (int16_t)(((avail_le >> (e0 >> ush)) & 1) ? v0 : (uint32_t)mid);
        if (t + 16 < n) nb[c][SB_C0 - 1 - (t + 16)] = (int16_t)(((avail_le >> (e1 >> ush)) & 1) ? v1 : (uint32_t)mid);
        if (t == 0) nb[c][SB_C0] = (int16_t)(avail_ul ? vc : (uint32_t)mid);
        if (n > 32) {
            for (int e = 2 * t + 32; e < n; e += 32)
                *(uint32_t *)&nb[c][SB_C0 + 1 + e] = ((avail_up >> (e >> ush)) & 1) ? *(const uint32_t *)(org - s + e) : midp;
            for (int e = t + 32; e < n; e += 16)
                nb[c][SB_C0 - 1 - e] = ((avail_le >> (e >> ush)) & 1) ? org[e * s - 1] : (int16_t)mid;
        }
    }
    wave_lds_sync();
    // ---- DC values (ipred_dc_b): (sum of h left + w up samples + w) >> (log2 w + 1); IPD_UR_B: the averaged diagonal ----
#pragma unroll
    for (int c = 0; c < 3; c++) {
        const int mode = c ? mode_c : mode_l;
        const int w = c ? cw >> 1 : cw, h = c ? chh >> 1 : chh;
        if (mode == 0) {
            int acc = 0;
            for (int e = t; e < w + h; e += 16) acc += e < h ? nb[c][SB_C0 - 1 - e] : nb[c][SB_C0 + 1 + e - h];
            acc = row_sum16(acc);                                // (rows whose CU has another mode run the instruction with a zero: DPP reads active lanes only)
            if (t == 0) nb[c][NB_DC] = (int16_t)((acc + w) >> ((c ? lw - 1 : lw) + 1));
        } else if (mode == 4) {
            for (int e = t; e < w + h; e += 16) nb[c][SB_UR + e] = (int16_t)((nb[c][SB_C0 + 1 + e] + nb[c][SB_C0 - 1 - e]) >> 1);
        }
    }
    wave_lds_sync();
    if (!has) return;
    // ---- prediction + reconstruction of the lane's SCU (as intra_body's Baseline pass) ----
    const int x = cu_x + lx, y = cu_y + ly;
    int pl[4][4], pc[2][2][2];
    int vl[7], vc[2][3];
    nb_fetch<7, SB_C0, SB_UR>(nb[0], mode_l, lx, ly, vl);
    nb_fetch<3, SB_C0, SB_UR>(nb[1], mode_c, lx >> 1, ly >> 1, vc[0]);
    nb_fetch<3, SB_C0, SB_UR>(nb[2], mode_c, lx >> 1, ly >> 1, vc[1]);
#pragma unroll
    for (int md = 0; md < 5; md++) {
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
    int16_t *dy = a.cur_y + y * a.s_l + x;
    const int coff = (y >> 1) * a.s_c + (x >> 1);
    if (!(nflags & 32u))                                         // (local dual tree: a chroma-only CU leaves luma alone, a luma-only one chroma)
#pragma unroll
        for (int r = 0; r < 4; r++) {
            // also without coefficients: the reference clips the prediction (xevd_recon.c:44-51)
            const uint32_t o0 = recon2i(pack2i(pl[r][0], pl[r][1]), (cbf & 1) ? rl[r].x : 0u, maxv), o1 = recon2i(pack2i(pl[r][2], pl[r][3]), (cbf & 1) ? rl[r].y : 0u, maxv);
            *(uint2 *)(dy + r * a.s_l) = make_uint2(o0, o1);
        }
    if (!(nflags & 64u))
#pragma unroll
        for (int c = 1; c < 3; c++) {
            int16_t *d = (c == 1 ? a.cur_u : a.cur_v) + coff;
#pragma unroll
            for (int r = 0; r < 2; r++)
                *(uint32_t *)(d + r * a.s_c) = recon2i(pack2i(pc[c - 1][r][0], pc[c - 1][r][1]), ((cbf >> c) & 1) ? rc[c - 1][r] : 0u, maxv);      // the luma depth clips chroma too (xevd_recon.c:75-90)
        }
}

// the level-1 launch of pictures with the Baseline predictors and none of the other node kinds: workgroups [0, big) take a large CU per wave, the rest 32 small CUs each
__global__ __launch_bounds__(64 * INTRA_WAVES) void k_intra_l1(const IntraArgs a, uint32_t big)
{
    constexpr int BIG_LDS = INTRA_WAVES * IntraLds<false>::WAVE, SMALL_LDS = SMALL_GROUPS * 3 * SB_LEN;
    __shared__ __attribute__((aligned(16))) int16_t s_nb[BIG_LDS > SMALL_LDS ? BIG_LDS : SMALL_LDS];
    if (blockIdx.x < big) {
        IntraArgs b = a;
        b.count = a.count - a.n_small;
        intra_body<false, 0, false, false, INTRA_WAVES>(b, blockIdx.x, s_nb, nullptr);
    } else intra_small_body(a, blockIdx.x - big, s_nb);
}

template <bool DEP, int EIPD, bool IBC, bool HTDF>
__global__ __launch_bounds__(64 * INTRA_WAVES) void k_intra(const IntraArgs a)
{
    __shared__ __attribute__((aligned(16))) int16_t s_nb[INTRA_WAVES * IntraLds<HTDF>::WAVE];
    __shared__ uint32_t s_chunk;
    intra_body<DEP, EIPD, IBC, HTDF, INTRA_WAVES>(a, blockIdx.x, s_nb, &s_chunk);
}

// k_intra_itdq - the data-flow launch of this picture and the residual pass of the NEXT picture in one grid.  The data-flow kernel is a chain of
// dependent memory round trips (4 K waves at 8K, the SIMDs idle most of its 45 us) and the residual pass depends on nothing but its batch, so its
// work items fill the machine under the chain: workgroups [0, n_intra_wg) run intra_body (tickets order them, whatever the dispatcher does), the rest
// one residual work item each.  256 threads: the residual pass's workgroup shape; four CUs in flight per intra workgroup (18.6 KB of LDS either way).
// Measured at 8K (profiles/round3_*): 45 us + 39 us as two launches (43 + 55 when the residual pass ran beside the level-1 launch on a second stream,
// with two cross-stream event waits of 6 us each), 64 us as one.  (Letting the level-1 launch carry a share of the work items too - it is 23 us of memory latency
// as well - measured nothing at 25 % and 1 - 3 % slower at 35 - 70 %: that launch is short and dense enough to be slowed down by the company.)
// (Round 5, measured and dropped: the level-1 launch held to 64 VGPRs - eight waves per SIMD instead of six, 92 bytes of scratch per lane: 41 us instead of 24 at cfg4, 31.6 instead of
//  28.8 at 1080p.)
// (Round 3, measured and dropped: ONE launch for all levels - the level-1 CUs at the head of this launch's list, publishing flags like everybody else, strand
// members waiting for their level-1 CUs - instead of the plain level-1 launch in front: bit-exact, 8K 2764 -> 2587 frames/s, 4K 8172 -> 7860, 1080p 10996 -> 11163.)
#define FUSED_WAVES 4
// (Round 5: the residual pass alone needs 44 VGPRs, this kernel 121 - 145 because of the chain's side, so the pass rides at half its own occupancy.  Holding the kernel to
//  96 / 80 / 64 VGPRs (amdgpu_waves_per_eu 5 / 6 / 8: 72 / 172 / 236 bytes of scratch, all in the chain's side) measured 3035 / 2740 / 2602 frames/s against 3052 at 8K, 8050 /
//  6820 / 6300 against 8600 at 4K: the launch is as long as its chain, and spills lengthen every link of it.  tools/archive/r5_w.sh)
template <int EIPD, bool IBC, bool IQT>
__global__ __launch_bounds__(64 * FUSED_WAVES) void k_intra_itdq(const IntraArgs a, const ItdqArgs r, uint32_t n_intra_wg, uint64_t rate)
{
    constexpr int ITDQ_DW = (IQT ? ITDQ_LDS_DWORDS - ITDQ_PLANES_DWORDS / 2 : ITDQ_LDS_DWORDS) + 2 * ITDQ_MAX_G, INTRA_DW = (FUSED_WAVES * IntraLds<false>::WAVE + 1) / 2 + 4;
    __shared__ __attribute__((aligned(16))) uint32_t raw[ITDQ_DW > INTRA_DW ? ITDQ_DW : INTRA_DW];
    // The chain's workgroups are spread evenly over the first `span` blocks of the grid (the host passes half of it) instead of all in front: the list is sorted by
    // level and tickets are drawn in the order the workgroups start, so a level's CUs arrive about when the level before them is done - in front, 4 400 waves that
    // mostly wait took half the machine's wave slots from the residual pass for the whole length of the chain, and the pass was the long pole of the launch.
    // One box, share of the grid the chain is spread over 0 (all in front) / 25 / 50 / 75 / 100 %: 8K 2697 / 2772 / 2790 / 2792 / 2735 frames/s, 4K 8183 / 8066 / 8194 /
    // 8068 / 7795.
    // (`rate` = ceil(2^32 n_intra_wg / span), from the host: the chain's share of the blocks before this one is a multiplication - the two 64-bit divisions that stood here
    //  were in front of every work item of the residual pass too)
    const uint32_t before = min(n_intra_wg, (uint32_t)(((uint64_t)blockIdx.x * rate) >> 32)), after = min(n_intra_wg, (uint32_t)(((uint64_t)(blockIdx.x + 1) * rate) >> 32));
    if (after > before) {
        // (s_setprio 3 for the chain's waves, so that a wave whose flags have arrived is not kept waiting by the residual pass's: measured, nothing - 8K 3053 / 3100 / 3054
        //  frames/s without, 3040 / 3059 / 3117 with, tools/archive/r5_w.sh)
        intra_body<true, EIPD, IBC, false, FUSED_WAVES>(a, blockIdx.x, (int16_t *)raw, raw + INTRA_DW - 1);
    } else {
        const int wi = (int)(blockIdx.x - before);
        if (wi >= r.n_waves) return;
        uint32_t *s_rm = raw + ITDQ_DW - 2 * ITDQ_MAX_G;
        itdq_dispatch<IQT>(r, wi, raw, s_rm, s_rm + ITDQ_MAX_G);
    }
}

void upload_transform_tables_intra(const int *tm, const int16_t *ats, hipStream_t s) { upload_transform_tables_tu(tm, ats, s); }

int intra_chunk(bool with_itdq) { return with_itdq ? FUSED_WAVES : INTRA_WAVES; }

// dep launch with `next` != NULL: k_intra_itdq (callers check intra_itdq_fusable first)
void launch_intra(xgpu_ctx *c, const IntraArgs &a, bool dep, bool ibc, bool htdf, const ItdqArgs *next, bool right)
{
    right = right && c->sp.tool_eipd;                   // (the Baseline predictors have no right-hand form)
    if (next) {
        const uint32_t n_wg = (uint32_t)((a.count + FUSED_WAVES - 1) / FUSED_WAVES);
        const dim3 g(n_wg + (uint32_t)next->n_waves), b(64 * FUSED_WAVES);
        // the chain's workgroups spread evenly over the first half of the grid (k_intra_itdq)
        const uint32_t span = std::max(n_wg, g.x / 2);
        const uint64_t rate = (((uint64_t)n_wg << 32) + span - 1) / span;      // <= 2^32: every chain workgroup 0 .. n_wg - 1 is some block's, in order, inside the first `span` blocks
        // (fewer workgroups per CU by dynamic LDS, so that the pass leaves the chain's round trips more room: 4 / 3 / 2 per CU = 3166 / 3024 / 2800 frames/s at 8K, 8718 / 8607 / 8500
        //  at 4K - the launch needs the pass's occupancy as much as the chain's latency)
#define LAUNCHF(E, I) do { if (next->iqt) hipLaunchKernelGGL((k_intra_itdq<E, I, true>), g, b, 0, c->stream, a, *next, n_wg, rate); else hipLaunchKernelGGL((k_intra_itdq<E, I, false>), g, b, 0, c->stream, a, *next, n_wg, rate); } while (0)
        if (right) LAUNCHF(2, true);
        else if (c->sp.tool_eipd) { if (ibc) LAUNCHF(1, true); else LAUNCHF(1, false); }
        else                 { if (ibc) LAUNCHF(0, true); else LAUNCHF(0, false); }
#undef LAUNCHF
        return;
    }
    const int per = INTRA_WAVES;
    // (pictures whose level 1 fits into the machine's wave slots a few times over keep the wave per CU: 1080p, 1059 CUs: 28.9 us against 29.6; 4K, 4129: 35.5 -> 34.8;
    //  8K, 15727: 58.1 -> 51.9 - both intra launches, tools/archive/r5_v.sh; XEVD_HIP_INTRA_SMALL_MIN moves the limit - the GPU tests run the small pictures with 1)
    if (!dep && !right && !htdf && !ibc && !c->sp.tool_eipd && a.n_small >= c->intra_small_min) {
        const uint32_t big = (uint32_t)((a.count - a.n_small + per - 1) / per), small_blocks = (uint32_t)((a.n_small + SMALL_GROUPS - 1) / SMALL_GROUPS);
        hipLaunchKernelGGL(k_intra_l1, dim3(big + small_blocks), dim3(64 * INTRA_WAVES), 0, c->stream, a, big);
        return;
    }
    const int blocks = (a.count + per - 1) / per;
    const dim3 g(blocks), b(64 * INTRA_WAVES);
#define LAUNCH(D, E, I, H) hipLaunchKernelGGL((k_intra<D, E, I, H>), g, b, 0, c->stream, a)
    if (right) { if (dep) LAUNCH(true, 2, true, true); else LAUNCH(false, 2, true, true); }      // SUCO pictures: one instantiation that knows everything
    else if (htdf) {     // pictures with HTDF nodes (they use the IBC-capable instantiation)
        if (c->sp.tool_eipd) { if (dep) LAUNCH(true, 1, true, true); else LAUNCH(false, 1, true, true); }
        else                 { if (dep) LAUNCH(true, 0, true, true); else LAUNCH(false, 0, true, true); }
    } else if (ibc) {      // pictures with intra-block-copy CUs: the instantiation that knows the copy path
        if (c->sp.tool_eipd) { if (dep) LAUNCH(true, 1, true, false); else LAUNCH(false, 1, true, false); }
        else                 { if (dep) LAUNCH(true, 0, true, false); else LAUNCH(false, 0, true, false); }
    } else {
        if (c->sp.tool_eipd) { if (dep) LAUNCH(true, 1, false, false); else LAUNCH(false, 1, false, false); }
        else                 { if (dep) LAUNCH(true, 0, false, false); else LAUNCH(false, 0, false, false); }
    }
#undef LAUNCH
#ifdef INTRA_PROFILE
    static int shots = 0;
    if (dep && ++shots == 5) {
        hipStreamSynchronize(c->stream);
        uint32_t h[8 * 64];
        hipMemcpyFromSymbol(h, HIP_SYMBOL(g_intra_prof), sizeof(h));
        for (int i = 0; i < 64; i++)
            fprintf(stderr, "  pos %d start %10u  wait %6u stage %5u (EIPD luma: issue %5u, return %5u | HTDF: to the block loads %5u from stamp 3, block in LDS %5u) plan %5u predict %5u ack %5u\n", INTRA_PROFILE + i, h[8 * i], h[8 * i + 1] - h[8 * i],
                    h[8 * i + 2] - h[8 * i + 1], h[8 * i + 6] - h[8 * i + 1], h[8 * i + 7] - h[8 * i + 6], h[8 * i + 6] - h[8 * i + 3], h[8 * i + 7] - h[8 * i + 6], h[8 * i + 3] - h[8 * i + 2], h[8 * i + 4] - h[8 * i + 3], h[8 * i + 5] - h[8 * i + 4]);
    }
#endif
}
