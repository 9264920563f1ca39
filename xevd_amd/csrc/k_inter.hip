// k_inter.hip - motion compensation + residual add + clip + SCU-map update for every inter CU of a picture.
//
// Replaces, per picture, the reference's per-CU sequence  xevd_mc -> xevd_recon_yuv -> xevd_set_dec_info
// (src_base/xevd.c:725-754, xevd_mc.c:469-557, xevd_recon.c:35-92, xevd_util.c:1574-1660).
//
// MI355X mapping (not a translation of the per-CU C/AVX loops):
//   * one 256-thread workgroup per 64x64 luma region, one LANE per 4x4 SCU (+ its two 2x2 chroma blocks).  MC is
//     a per-sample function of the covering CU's motion, so an SCU can be predicted independently of the CU it
//     belongs to: every lane runs the same straight-line code whatever the CU sizes are (no size classes, no
//     divergence on block shape), and the 16 lanes of an SCU row store 128 contiguous bytes per picture row.
//   * the covering CU comes from a per-picture SCU -> CU owner map that the host paints in xgpu_batch_create; the reference table and the tap
//     tables are staged once per workgroup in LDS, so a lane's chain is owner -> CU record -> reference samples;
//   * a wave whose 32x32 tile lies inside ONE CU takes the tile path instead: window fetched once into the wave's own LDS, shared
//     horizontal pass, vertical pass from LDS (mc_luma_tile / mc_chroma_tile below);
//   * the 11x11 (luma) / 5x5 (chroma) reference windows are read straight from HBM/L2 at the 2-byte-aligned sample
//     address (16 + 8 bytes per luma row, 12 per chroma row; gfx950 runs in unaligned-access mode) - by every lane only
//     the rows and samples its taps do not multiply by zero (mc_scu_list); neighbouring lanes share the halo
//     through the vector L1, workgroups are mapped to XCDs in contiguous bands so vertical halos share an L2.
//   * FIRs run on packed s16 pairs with v_dot2c_i32_i16 (two taps per instruction); the four rounding regimes of
//     the reference (copy / H-only / V-only / 2-D) are one code path with per-lane tap vectors, shifts and
//     offsets, so lanes with different sub-pel classes do not diverge.
//   * no MFMA: these are 4/8-tap integer FIRs, bounded by load/issue rate and HBM, not by dense contraction.
#include "xgpu_internal.h"

#include <hip/hip_ext.h>
#include <type_traits>
#include "mc_filters.h"

#define LDS_AS __attribute__((address_space(3)))

// ---------------------------------------------------------------------------------------------------------
// Wave-uniform motion (every SCU of a wave's 32x32 tile belongs to one CU - three quarters of the samples of a typical
// picture): the separable filter runs as a tile through the wave's own LDS instead of per lane.  The 39x39 reference window
// is fetched ONCE with 16-byte loads (the per-lane path fetches 11 rows x 24 bytes per SCU: 5x the vector-memory
// instructions, which bound that path), the horizontal pass produces each of the 39x32 intermediate values once
// (per lane: 11 rows for 4 output rows), and the vertical pass reads 11 rows x 8 bytes per lane from LDS.  Same integer
// arithmetic and rounding regimes as mc_luma_4x4 / mc_chroma_2x2, only the work is shared between the lanes.
// LDS accesses of one wave execute in order, so no workgroup barrier is involved - only compiler ordering.
// ---------------------------------------------------------------------------------------------------------
#define UW_STRIDE 48                 // luma window row stride in samples: 96 B, 16-byte aligned chunk stores
#define UI_STRIDE 40                 // intermediate rows: 80 B = 20 banks, so the 8 x 4 SCUs of a half wave hit 64 different banks
#define UC_STRIDE 24                 // chroma rows (window and intermediate): 12 banks, conflict-free for the 64 lanes' dword reads
#define UCW_STRIDE 32                // chroma window rows (the intermediate rows keep UC_STRIDE)
// k_inter_tile's block per wave: luma window (4 requests x 60 chunks = 40 rows), the two chroma windows (152 chunks; 3 requests x 64 slots); the intermediate rows
// of the passes are written over the window rows they come from (inter_tile)
static_assert(UI_STRIDE <= UW_STRIDE && UC_STRIDE <= UCW_STRIDE, "an intermediate row is not longer than the window row it replaces");
#define UT_L_SAMPLES (40 * UW_STRIDE)
#define UT_C_SAMPLES (3 * 64 * 8)
#define UNI_SAMPLES   (UT_L_SAMPLES + UT_C_SAMPLES)
static_assert(2 * 19 * UCW_STRIDE <= UT_C_SAMPLES, "both chroma windows fit");

__device__ __forceinline__ void wave_lds_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// a lane's chunk of the tile's luma window: offsets into the reference plane (samples) and into the wave's LDS window - 10 rows x 6 chunks of 8 samples per request
// (lanes 60..63 idle), four requests cover the 39 rows.  The offsets depend on the lane alone: computed once per kernel, every request adds a constant.
// The chunks are 16-byte ALIGNED pieces of the reference rows (rows start 256-byte aligned): an unaligned 16-byte load costs the vector L1 2.4x the accesses (round 4).
// Luma: up to 7 + 39 samples = 6 chunks per row; chroma: up to 7 + 19 samples = 4 chunks, 2 planes x 19 rows x 4 chunks = 152 chunks in three requests of 64 lanes.
struct LaneMap { int gy, ly; };

// the two passes over a 39x39 window in LDS: Wn = the window's first sample, WS = its row stride in samples (UW_STRIDE: the wave's own window; REG_W_STRIDE: the
// wave's part of the 71x71 window its workgroup shares, see inter_tile<2>)
// SH: the window was fetched in 16-byte ALIGNED chunks, so its first sample sits at an arbitrary sample of the LDS row: Wn = that sample rounded down to a dword,
// odd = whether it is the dword's high half (wave-uniform: the rows are read dword-wise and shifted by 16 bits then)
template <bool H, bool V, int WS, bool SH = false>
__device__ __forceinline__ void luma_tile_filter(const int16_t *Wn, const uint32_t ch[4], const uint32_t cv[4], Regime rg, int maxv, int16_t *I, int lane, uint32_t o[8], int odd = 0)
{
#pragma unroll
    for (int it = 0; it < 5; it++) {                                // horizontal pass: 39 rows x 8 groups of 4 columns
        const int idx = lane + 64 * it, row = idx >> 3, g = idx & 7;
        if (idx < 312) {
            // E[k] = dword k of the row = samples (2k, 2k + 1) from the dword that holds the window's first sample, O[k] = samples (2k + 1, 2k + 2).  A window that starts
            // on the dword's low half takes its even taps' pairs from E and its odd ones from O; one that starts on the high half (`odd`, wave-uniform) the other
            // way round - the same five funnel shifts either way, no extra ones to re-align the row first
            uint32_t E[6];
            if (SH) {
                const uint32_t *w = (const uint32_t *)(Wn + row * WS + 4 * g);
#pragma unroll
                for (int q = 0; q < 6; q++) E[q] = w[q];
            } else {
                const uint2 *w = (const uint2 *)(Wn + row * WS + 4 * g);
                const uint2 a = w[0], b = w[1], c = w[2];
                E[0] = a.x; E[1] = a.y; E[2] = b.x; E[3] = b.y; E[4] = c.x; E[5] = c.y;
            }
            uint2 r;
            if (H) {
                const uint32_t O[5] = { hi_lo(E[1], E[0]), hi_lo(E[2], E[1]), hi_lo(E[3], E[2]), hi_lo(E[4], E[3]), hi_lo(E[5], E[4]) };
                int t[4];
                if (SH && odd) {
                    t[0] = dot2(ch[3], O[3], dot2(ch[2], O[2], dot2(ch[1], O[1], dot2z(ch[0], O[0]))));
                    t[2] = dot2(ch[3], O[4], dot2(ch[2], O[3], dot2(ch[1], O[2], dot2z(ch[0], O[1]))));
                    t[1] = dot2(ch[3], E[4], dot2(ch[2], E[3], dot2(ch[1], E[2], dot2z(ch[0], E[1]))));
                    t[3] = dot2(ch[3], E[5], dot2(ch[2], E[4], dot2(ch[1], E[3], dot2z(ch[0], E[2]))));
                } else {
                    t[0] = dot2(ch[3], E[3], dot2(ch[2], E[2], dot2(ch[1], E[1], dot2z(ch[0], E[0]))));
                    t[2] = dot2(ch[3], E[4], dot2(ch[2], E[3], dot2(ch[1], E[2], dot2z(ch[0], E[1]))));
                    t[1] = dot2(ch[3], O[3], dot2(ch[2], O[2], dot2(ch[1], O[1], dot2z(ch[0], O[0]))));
                    t[3] = dot2(ch[3], O[4], dot2(ch[2], O[3], dot2(ch[1], O[2], dot2z(ch[0], O[1]))));
                }
#pragma unroll
                for (int q = 0; q < 4; q++) { t[q] >>= rg.sh1; if (!V) t[q] = clip3(0, maxv, t[q]); }
                r = make_uint2(pack2(t[0], t[1]), pack2(t[2], t[3]));
            } else {
                if (SH && odd) r = make_uint2(E[2], E[3]);                                         // samples 3..6 of the window
                else r = make_uint2(hi_lo(E[2], E[1]), hi_lo(E[3], E[2]));
            }
            *(uint2 *)(I + row * UI_STRIDE + 4 * g) = r;
        }
    }
    wave_lds_sync();
    const int16_t *base = I + ((lane >> 3) << 2) * UI_STRIDE + ((lane & 7) << 2);
    if (!V) {
#pragma unroll
        for (int r = 0; r < 4; r++) { const uint2 q = *(const uint2 *)(base + (r + 3) * UI_STRIDE); o[r * 2] = q.x; o[r * 2 + 1] = q.y; }
    } else {
        uint2 q[11];
#pragma unroll
        for (int j = 0; j < 11; j++) q[j] = *(const uint2 *)(base + j * UI_STRIDE);
        int acc[4][4];
#pragma unroll
        for (int j = 1; j < 11; j++) {
            const uint32_t pr[4] = { __builtin_amdgcn_perm(q[j].x, q[j - 1].x, 0x05040100u), __builtin_amdgcn_perm(q[j].x, q[j - 1].x, 0x07060302u),
                                     __builtin_amdgcn_perm(q[j].y, q[j - 1].y, 0x05040100u), __builtin_amdgcn_perm(q[j].y, q[j - 1].y, 0x07060302u) };
#pragma unroll
            for (int c = 0; c < 4; c++)
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    const int d = j - 1 - r;
                    if (d == 0) acc[r][c] = dot2a(cv[0], pr[c], rg.off2);
                    else if (d > 0 && d <= 6 && (d & 1) == 0) acc[r][c] = dot2(cv[d >> 1], pr[c], acc[r][c]);
                }
        }
#pragma unroll
        for (int r = 0; r < 4; r++) {
            int v[4];
#pragma unroll
            for (int c = 0; c < 4; c++) v[c] = clip3(0, maxv, acc[r][c] >> rg.sh2);
            o[r * 2 + 0] = pack2(v[0], v[1]);
            o[r * 2 + 1] = pack2(v[2], v[3]);
        }
    }
    wave_lds_sync();
}
// Both chroma planes of the tile (16x16 each): the passes over the two 19x19 windows Wu / Wv (row stride WS) in LDS
template <bool H, bool V, int WS, bool SH = false>
__device__ __forceinline__ void chroma_tile_filter(const int16_t *Wu, const int16_t *Wv, const uint32_t ch[2], const uint32_t cv[2],
                                                   Regime rg, int maxv, int16_t *I, int lane, uint32_t ou[2], uint32_t ov[2], int odd = 0)
{
#pragma unroll
    for (int it = 0; it < 3; it++) {                                // horizontal pass: 2 x 19 rows x 4 groups of 4 columns
        const int idx = lane + 64 * it, prow = idx >> 2, g = idx & 3;
        if (idx < 152) {
            uint32_t E[4];                                               // as in luma_tile_filter
            if (SH) {
                const uint32_t *w = (const uint32_t *)((prow >= 19 ? Wv + (prow - 19) * WS : Wu + prow * WS) + 4 * g);
#pragma unroll
                for (int q = 0; q < 4; q++) E[q] = w[q];
            } else {
                const uint2 *w = (const uint2 *)((prow >= 19 ? Wv + (prow - 19) * WS : Wu + prow * WS) + 4 * g);
                const uint2 a = w[0], b = w[1];
                E[0] = a.x; E[1] = a.y; E[2] = b.x; E[3] = b.y;
            }
            uint2 r;
            if (H) {
                const uint32_t O[3] = { hi_lo(E[1], E[0]), hi_lo(E[2], E[1]), hi_lo(E[3], E[2]) };
                int t[4];
                if (SH && odd) {
                    t[0] = dot2(ch[1], O[1], dot2z(ch[0], O[0]));
                    t[1] = dot2(ch[1], E[2], dot2z(ch[0], E[1]));
                    t[2] = dot2(ch[1], O[2], dot2z(ch[0], O[1]));
                    t[3] = dot2(ch[1], E[3], dot2z(ch[0], E[2]));
                } else {
                    t[0] = dot2(ch[1], E[1], dot2z(ch[0], E[0]));
                    t[1] = dot2(ch[1], O[1], dot2z(ch[0], O[0]));
                    t[2] = dot2(ch[1], E[2], dot2z(ch[0], E[1]));
                    t[3] = dot2(ch[1], O[2], dot2z(ch[0], O[1]));
                }
#pragma unroll
                for (int q = 0; q < 4; q++) { t[q] >>= rg.sh1; if (!V) t[q] = clip3(0, maxv, t[q]); }
                r = make_uint2(pack2(t[0], t[1]), pack2(t[2], t[3]));
            } else {
                if (SH && odd) r = make_uint2(E[1], E[2]);                                         // samples 1..4
                else r = make_uint2(hi_lo(E[1], E[0]), hi_lo(E[2], E[1]));
            }
            *(uint2 *)(I + prow * UC_STRIDE + 4 * g) = r;
        }
    }
    wave_lds_sync();
#pragma unroll
    for (int pl = 0; pl < 2; pl++) {
        const int16_t *base = I + (pl * 19 + ((lane >> 3) << 1)) * UC_STRIDE + ((lane & 7) << 1);
        uint32_t *o = pl ? ov : ou;
        if (!V) {
            o[0] = *(const uint32_t *)(base + UC_STRIDE); o[1] = *(const uint32_t *)(base + 2 * UC_STRIDE);
        } else {
            uint32_t q[5];
#pragma unroll
            for (int j = 0; j < 5; j++) q[j] = *(const uint32_t *)(base + j * UC_STRIDE);
            int acc[2][2];
#pragma unroll
            for (int j = 1; j < 5; j++) {
                const uint32_t pr[2] = { __builtin_amdgcn_perm(q[j], q[j - 1], 0x05040100u), __builtin_amdgcn_perm(q[j], q[j - 1], 0x07060302u) };
#pragma unroll
                for (int c = 0; c < 2; c++)
#pragma unroll
                    for (int r = 0; r < 2; r++) {
                        const int d = j - 1 - r;
                        if (d == 0) acc[r][c] = dot2a(cv[0], pr[c], rg.off2);
                        else if (d == 2) acc[r][c] = dot2(cv[1], pr[c], acc[r][c]);
                    }
            }
#pragma unroll
            for (int r = 0; r < 2; r++)
                o[r] = pack2(clip3(0, maxv, acc[r][0] >> rg.sh2), clip3(0, maxv, acc[r][1] >> rg.sh2));
        }
    }
    wave_lds_sync();
}
// ---------------------------------------------------------------------------------------------------------
// A 64x64 region inside ONE CU (half of the samples of a typical picture lie in CUs of 64x64 and above): its four waves would each fetch their own 39x39 window -
// 78-byte rows that touch 1.44 cache lines, and the 7-sample halos between the four tiles twice.  The workgroup fetches the region's 71x71 window (+ 2 x 35x35 chroma)
// ONCE into LDS it shares - rows of 142 bytes: 2.1 lines - and every wave filters its own 39x39 part of it with the tile passes above.  Per region and list 149 + 104
// line requests instead of 225 + 198 (round 4: the kernel is bound by the rate of L1 -> L2 line requests, ~70 G/s of the ~85 G/s this access pattern reaches with no
// arithmetic at all - tools/ubench/win_bw.hip).  Two workgroup barriers per list.
// ---------------------------------------------------------------------------------------------------------
#define REG_W_STRIDE 80              // up to 7 + 71 columns in 10 chunks of 8 samples, every chunk 16-byte aligned in the reference picture
#define REG_C_STRIDE 48              // up to 7 + 35 columns in 6 chunks
#define REG_W_SAMPLES (71 * REG_W_STRIDE)
#define REG_C_SAMPLES (35 * REG_C_STRIDE)
#define REG_I_SAMPLES (39 * UI_STRIDE)
#define REG_SAMPLES   (REG_W_SAMPLES + 2 * REG_C_SAMPLES + 4 * REG_I_SAMPLES)
// a thread's chunks of the region windows: luma 71 rows x 10 chunks in three rounds of 256 threads, chroma 2 planes x 35 rows x 6 chunks in two; offsets < 0: none
struct RegionMap { int y[3], c[2]; };      // row << 8 | chunk (chroma: | plane << 7); < 0: none.  The offsets are formed where they are used: five registers, not ten
__device__ __forceinline__ RegionMap region_map(int t)
{
    RegionMap m;
#pragma unroll
    for (int it = 0; it < 3; it++) {
        const int idx = t + 256 * it, row = idx / 10, k = idx - row * 10;
        m.y[it] = idx < 71 * 10 ? (row << 8) | k : -1;
    }
#pragma unroll
    for (int it = 0; it < 2; it++) {
        const int idx = t + 256 * it, pl = idx >= 210, j = idx - 210 * pl, row = j / 6, k = j - row * 6;
        m.c[it] = idx < 420 ? (row << 8) | (pl << 7) | k : -1;
    }
    return m;
}

// ---------------------------------------------------------------------------------------------------------
// k_inter_split: one SCU per lane, one list.  The arithmetic of mc_luma_4x4 + 2 x mc_chroma_2x2 (mc_filters.h) with the list's requests in two instalments in front
// of it: a memory round trip for the luma window, one for both chroma windows.  (Left to itself the compiler fetched the luma window in four instalments and each
// chroma plane on its own - six dependent round trips per list.)
// Round 5, second half: the pass runs at what the memory path delivers for its requests, so the requests were cut - lanes sit out of the rows and samples their
// identity taps multiply by zero, a chroma row is one 12-byte request instead of 8 + 4 bytes: 42 -> 25.5 requests per lane and list, k_inter 143.7 -> 135.6 us on one
// box (tools/archive/r5_u.sh, profiles/round5_exp_split_requests.txt; the same structure with every lane requesting everything: 141.0).  The number of round trips does not
// show: luma in instalments of 6 + 5 rows with the chroma planes one after the other (five round trips per list), 6 + 5 rows and both planes (three), 9 + 2 rows,
// all eleven (two) measured 138.2 / 135.6 / 135.3 / 136.0 us - what shows is the fourth wave per SIMD (133 VGPRs: 144.9 us).  Also measured and dropped: a lane whose
// SCU has one of the same CU under it (lane + 8) taking its rows 4..10 from that lane's registers (ds_bpermute) instead of requesting them - 23 rows instead of 44
// per column of a 16x16 CU, bit-exact, and no faster (133.1 against 132.3 us): the requests that cost are those of the 4x4 CUs.
// H / V: does ANY lane of the wave filter luma in that direction (wave-uniform, as mc_luma_4x4); chroma is fetched as for the full filter and its variant chosen after.
// ---------------------------------------------------------------------------------------------------------
template <bool H, bool V>
__device__ __forceinline__ void mc_scu_list(gs16 pl_, int s_l, gs16 pu_, gs16 pv_, int s_c, const uint32_t ch[4], const uint32_t cv[4], Regime rgl, int maxl,
                                            const uint32_t c2h[2], const uint32_t c2v[2], Regime rgc, int maxc, bool cwh, bool cwv, uint32_t o[8], uint32_t ou[2], uint32_t ov[2],
                                            bool fx, bool fy, bool cfx, bool cfy)
{
    constexpr int J0 = V ? 0 : 3, J1 = V ? 11 : 7;
#ifdef XGPU_NO_LANE_PRED
    fx = fy = cfx = cfy = true;                                  // (measurement build: every lane requests the whole window)
#endif
    // fx / fy (cfx / cfy): does THIS lane filter luma (chroma) horizontally / vertically.  A lane whose vector has a whole-sample component carries the identity
    // taps in that direction (mc_filters.h: one tap of 1, zeros around it): the rows above and below its block and the samples right of sample 7 are multiplied by
    // zero, so the lane does not request them - it sits out of those load instructions and the registers hold whatever they held.  With quarter-sample vectors a
    // quarter of the lanes has a whole-sample component in each direction: 16.2 requests per lane and luma window instead of 22.  A chroma row is ONE 12-byte request
    // (six samples, of which the filters use five; the round-4 form asked for 8 + 4 bytes): 10 per list instead of 20, 9.3 with the rows a lane sits out of.
    // A conditional load is a branch of its own, and what consumes the loaded value next is moved INTO that branch by the compiler - a wait for the round trip per
    // row.  `fence` (an empty asm that takes the rows' registers in and out) sits behind all requests of an instalment and in front of their consumers, and the
    // loaded value stays one register tuple until then (mc_filters.h: any_value, gload*_if).
    // Two instalments per list: all luma rows (66 registers), then - the luma block finished - both chroma windows (30); the kernel stays at four waves per SIMD.
    v4u32 A[11]; v2u32 B[11];
    auto request = [&](int ja, int jb) {
#pragma unroll
        for (int j = ja; j < jb; j++) {
            const bool row_on = !V || fy || (j >= 3 && j < 7);
            if (H) {
                A[j] = any_value<v4u32>(); B[j] = any_value<v2u32>();
                gload16_if(A[j], pl_ + j * s_l, row_on);
                gload8_if(B[j], pl_ + j * s_l + 8, row_on && fx);
            } else {
                B[j] = any_value<v2u32>();
                gload8_if(B[j], pl_ + j * s_l + 3, row_on);     // no lane filters horizontally: samples 3..6 of the window
            }
        }
    };
    auto fence = [&](int ja, int jb) {
#pragma unroll
        for (int j = ja; j < jb; j++) {
            if (H) asm volatile("" : "+v"(A[j]), "+v"(B[j]));
            else   asm volatile("" : "+v"(B[j]));
        }
    };
    int acc[4][4];
    int tp[4] = {0, 0, 0, 0};
    auto row = [&](int j) {
        int t[4];
        if (H) {
            const uint32_t D0 = A[j].x, D1 = A[j].y, D2 = A[j].z, D3 = A[j].w, D4 = B[j].x, D5 = B[j].y;
            const uint32_t Q0 = hi_lo(D1, D0), Q1 = hi_lo(D2, D1), Q2 = hi_lo(D3, D2), Q3 = hi_lo(D4, D3), Q4 = hi_lo(D5, D4);
            t[0] = dot2(ch[3], D3, dot2(ch[2], D2, dot2(ch[1], D1, dot2z(ch[0], D0))));
            t[2] = dot2(ch[3], D4, dot2(ch[2], D3, dot2(ch[1], D2, dot2z(ch[0], D1))));
            t[1] = dot2(ch[3], Q3, dot2(ch[2], Q2, dot2(ch[1], Q1, dot2z(ch[0], Q0))));
            t[3] = dot2(ch[3], Q4, dot2(ch[2], Q3, dot2(ch[1], Q2, dot2z(ch[0], Q1))));
#pragma unroll
            for (int c = 0; c < 4; c++) t[c] = clip3(rgl.lo1, rgl.hi1, t[c] >> rgl.sh1);
        } else {
            t[0] = (int)(int16_t)(B[j].x & 0xFFFF); t[1] = (int)(int16_t)(B[j].x >> 16);
            t[2] = (int)(int16_t)(B[j].y & 0xFFFF); t[3] = (int)(int16_t)(B[j].y >> 16);
        }
        if (!V) {
#pragma unroll
            for (int c = 0; c < 4; c++) acc[j - 3][c] = t[c];
            return;
        }
        if (j > 0) {
            // row pair (j-1, j) feeds output row r with tap pair (j-1-r)/2 when j-1-r is even and in 0..6
#pragma unroll
            for (int c = 0; c < 4; c++) {
                const uint32_t pr = pack2(tp[c], t[c]);
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    const int d = j - 1 - r;
                    if (d == 0) acc[r][c] = dot2a(cv[0], pr, rgl.off2);          // first tap pair carries the rounding offset
                    else if (d > 0 && d <= 6 && (d & 1) == 0) acc[r][c] = dot2(cv[d >> 1], pr, acc[r][c]);
                }
            }
        }
#pragma unroll
        for (int c = 0; c < 4; c++) tp[c] = t[c];
    };
    auto rows = [&](int ja, int jb) {
#pragma unroll
        for (int j = ja; j < jb; j++) row(j);
    };
    auto luma_out = [&]() {
#pragma unroll
        for (int r = 0; r < 4; r++) {
            int v[4];
#pragma unroll
            for (int c = 0; c < 4; c++) v[c] = clip3(0, maxl, V ? acc[r][c] >> rgl.sh2 : acc[r][c]);
            o[r * 2 + 0] = pack2(v[0], v[1]);
            o[r * 2 + 1] = pack2(v[2], v[3]);
        }
    };
    v3u32 CW[2][5];
    auto request_c = [&](int pl) {
#pragma unroll
        for (int j = 0; j < 5; j++) {
            const bool row_on = cfy || j == 1 || j == 2;      // (as for luma: identity taps select row 1 + r and sample 1 + c)
            CW[pl][j] = any_value<v3u32>();
            gload12_if(CW[pl][j], (pl ? pv_ : pu_) + j * s_c, row_on);
        }
    };
    auto fence_c = [&](int pl) {
#pragma unroll
        for (int j = 0; j < 5; j++) asm volatile("" : "+v"(CW[pl][j]));
    };
    // chroma, from the registers: mc_chroma_2x2's arithmetic on window row j = the dwords of CW[j] = samples 0..5
    auto chroma = [&](auto hc, auto vc, int pl, uint32_t oo[2]) {
        constexpr bool CH = decltype(hc)::value, CV = decltype(vc)::value;
        int a2[2][2];
        int tq[2] = {0, 0};
#pragma unroll
        for (int j = CV ? 0 : 1; j < (CV ? 5 : 3); j++) {
            int t[2];
            const uint32_t D0 = CW[pl][j].x, D1 = CW[pl][j].y, D2 = CW[pl][j].z;
            if (CH) {
                const uint32_t Q0 = hi_lo(D1, D0), Q1 = hi_lo(D2, D1);
                t[0] = dot2(c2h[1], D1, dot2z(c2h[0], D0));
                t[1] = dot2(c2h[1], Q1, dot2z(c2h[0], Q0));
#pragma unroll
                for (int c = 0; c < 2; c++) t[c] = clip3(rgc.lo1, rgc.hi1, t[c] >> rgc.sh1);
            } else {
                const uint32_t q = hi_lo(D1, D0);                                 // samples 1..2
                t[0] = (int)(int16_t)(q & 0xFFFF); t[1] = (int)(int16_t)(q >> 16);
            }
            if (!CV) { a2[j - 1][0] = t[0]; a2[j - 1][1] = t[1]; continue; }
            if (j > 0) {
#pragma unroll
                for (int c = 0; c < 2; c++) {
                    const uint32_t pr = pack2(tq[c], t[c]);
#pragma unroll
                    for (int r = 0; r < 2; r++) {
                        const int d = j - 1 - r;
                        if (d == 0) a2[r][c] = dot2a(c2v[0], pr, rgc.off2);
                        else if (d == 2) a2[r][c] = dot2(c2v[1], pr, a2[r][c]);
                    }
                }
            }
            tq[0] = t[0]; tq[1] = t[1];
        }
#pragma unroll
        for (int r = 0; r < 2; r++)
            oo[r] = pack2(clip3(0, maxc, CV ? a2[r][0] >> rgc.sh2 : a2[r][0]), clip3(0, maxc, CV ? a2[r][1] >> rgc.sh2 : a2[r][1]));
    };
    using T = std::true_type; using F = std::false_type;
    auto chroma_pl = [&](int pl, uint32_t oo[2]) {
        fence_c(pl);
        if (cwh) { if (cwv) chroma(T{}, T{}, pl, oo); else chroma(T{}, F{}, pl, oo); }
        else     { if (cwv) chroma(F{}, T{}, pl, oo); else chroma(F{}, F{}, pl, oo); }
    };
    request(J0, J1); fence(J0, J1);
    rows(J0, J1);
    luma_out();
    __builtin_amdgcn_sched_barrier(0);
    request_c(0); request_c(1);
    chroma_pl(0, ou); chroma_pl(1, ov);
}

#define OWNER_NONE 0xFFFFFFFFu

// One 32x32 tile (one wave): the SCU map records, and for plain inter CUs prediction + residual + store.  UNI: the whole tile lies in one CU -
// the CU record is made wave-uniform (scalar registers: its decoding, the vector clipping and every branch on it run on the scalar unit) and
// the filters run as a tile through the wave's LDS; otherwise every lane works on the CU that covers its SCU.
// Returns whether the lane has samples to store: pl / pu / pv = its 4x4 luma and 2x2 + 2x2 chroma samples (the caller stores them AFTER it has taken
// the prefetched records of the next tile out of their registers: stores and loads share one counter, and a wait behind the stores would be a
// wait for their acknowledgement - a memory round trip per tile).
// MODE 0: per lane; 1: the wave's tile inside one CU (UNI above); 2: the workgroup's whole 64x64 region inside one CU - the reference windows are fetched once per
// workgroup into LDS the four waves share (W = that block, rm = the thread's chunks of it, wave = the tile's place in the region), everything else as in mode 1
// MODE 0 filters through mc_scu_list: the list's luma window requested at once, then both chroma windows, every lane only what its taps do not multiply by zero
#ifdef XGPU_INTER_TRACE
// measurement build (make EXTRA=-DXGPU_INTER_TRACE): where the life of a wave goes, per role - shader cycles between marks, summed over the waves' lane 0
__device__ unsigned long long g_inter_trace[3][16];
struct InterTrace { unsigned long long prev; int role; bool on; };
#define TRACE_MARK(tr, p) do { if ((tr).on) { const unsigned long long now_ = __builtin_amdgcn_s_memtime(); atomicAdd(&g_inter_trace[(tr).role][p], now_ - (tr).prev); (tr).prev = __builtin_amdgcn_s_memtime(); } } while (0)
#define TRACE_ARG , InterTrace &tr
#define TRACE_PASS , tr
#define TRACE_OFF InterTrace tr = { 0, 0, false };
extern "C" int xgpu_test_inter_trace(unsigned long long out[48], int reset)
{
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_inter_trace), sizeof(g_inter_trace)) != hipSuccess) return -1;
    if (reset) { static unsigned long long z[48]; if (hipMemcpyToSymbol(HIP_SYMBOL(g_inter_trace), z, sizeof(z)) != hipSuccess) return -1; }
    return 0;
}
#else
#define TRACE_MARK(tr, p) do { } while (0)
#define TRACE_ARG
#define TRACE_PASS
#define TRACE_OFF
#endif
template <int MODE>
__device__ __forceinline__ bool inter_tile(const InterArgs &a, uint4 r0, uint4 r1, bool lane_ok, int sx, int sy, int lane, int16_t *W, const LaneMap fm,
                                           const uint4 (*s_ref)[2], const uint4 *s_ltap, const uint2 *s_ctap, uint32_t pl[8], uint32_t pu[2], uint32_t pv[2],
                                           const RegionMap *rm, int wave, uint32_t own TRACE_ARG)
{
    constexpr bool UNI = MODE != 0;
    if (UNI) {
        own = (uint32_t)__builtin_amdgcn_readfirstlane((int)own);
        r0.x = __builtin_amdgcn_readfirstlane(r0.x); r0.y = __builtin_amdgcn_readfirstlane(r0.y); r0.z = __builtin_amdgcn_readfirstlane(r0.z); r0.w = __builtin_amdgcn_readfirstlane(r0.w);
        r1.x = __builtin_amdgcn_readfirstlane(r1.x); r1.y = __builtin_amdgcn_readfirstlane(r1.y); r1.z = __builtin_amdgcn_readfirstlane(r1.z); r1.w = __builtin_amdgcn_readfirstlane(r1.w);
    } else if (!lane_ok) return false;
    const int cu_x = r0.x & 0xFFFF, cu_y = r0.x >> 16;
    const int lw = r0.y & 0xFF, lh = (r0.y >> 8) & 0xFF, pred_mode = (r0.y >> 16) & 0xF, cbf = r0.y >> 24;
    const int refi0 = (int)(int8_t)(r0.z & 0xFF), refi1 = (int)(int8_t)((r0.z >> 8) & 0xFF), qp_map = (r0.z >> 16) & 0xFF;
    const uint32_t coef_off = r0.w;
    const int cw = 1 << lw, chh = 1 << lh;
    const int x = sx << 2, y = sy << 2;
    const bool intra = pred_mode == XGPU_MODE_INTRA;
    // ATS-inter: the coded TU is one half/quarter of the CU at its start or end (xevdm_get_tu_size / get_tu_pos_offset,
    // src_main/xevdm_util.c:3585-3634); residual and luma cbf exist only there (xevdm_recon.c:62-112, xevdm_util.c:3670-3712)
    const int ai = (intra || pred_mode == XGPU_MODE_IBC) ? 0 : (int)((r1.w >> 8) & 0xFF);
    int tu_x = 0, tu_y = 0, tu_w = cw, tu_h = chh;
    if (ai) {
        const int idx = ai & 15, pos = ai >> 4;
        if (idx == 2 || idx == 4) { tu_h = chh >> (idx == 4 ? 2 : 1); tu_y = pos ? chh - tu_h : 0; }
        else                      { tu_w = cw >> (idx == 3 ? 2 : 1);  tu_x = pos ? cw - tu_w : 0; }
    }
    const int lx = x - cu_x - tu_x, ly = y - cu_y - tu_y;                 // position inside the TU
    const bool in_tu = (uint32_t)lx < (uint32_t)tu_w && (uint32_t)ly < (uint32_t)tu_h;

    // ---- SCU map update (xevd_set_dec_info): intra flag, QP, skip flag, luma cbf, COD + CU-edge flags ----
    {
        uint32_t m = ((uint32_t)qp_map << 16) | ((uint32_t)intra << 15) | (1u << 31);
        m |= SCU_RANK(own);                                      // the CU's place in decoding order: which of two neighbouring CUs the baseline deblocking filter reaches later (k_deblock.hip)
        if (pred_mode == XGPU_MODE_SKIP) m |= 1u << 23;
        const bool ibc = pred_mode == XGPU_MODE_IBC;
        if (ibc) m |= 1u << 26;                                  // MCU_SET_IBC (xevdm_def.h:325)
        if (((r0.z >> 24) & 1) && in_tu) m |= 1u << 24;          // CuRec.map_cbf
        // CU boundary, or the 64-sample transform boundary inside a wider CU (deblock_tree splits those, xevdm.c:1989-2037)
        if (((x - cu_x) & 63) == 0) m |= SCU_EDGE_L;
        if (((y - cu_y) & 63) == 0) m |= SCU_EDGE_T;
        if (x == cu_x && ((r0.y >> 16) & CU_NOCH_L)) m |= SCU_NOCH_L;          // luma-only CU of a local dual tree: no chroma edge inside the chroma block
        if (y == cu_y && ((r0.y >> 16) & CU_NOCH_T)) m |= SCU_NOCH_T;
        uint4 rec;
        rec.x = m;
        rec.y = (intra || ibc) ? 0x0000FFFFu : ((r0.z & 0xFFFFu) | ((uint32_t)ai << 16));
        rec.z = intra ? 0u : r1.x;                               // IBC keeps its block vector in list 0 (xevdm.c:1098-1110)
        rec.w = (intra || ibc) ? 0u : r1.y;
        // k_affine and k_dmvr run BESIDE this kernel on another stream (xgpu_batch_recon): the vector words they write - the sub-block vectors of an affine CU's used
        // lists, the refined vectors of a DMVR CU when the baseline filter is to see them - are theirs alone, this kernel leaves them out (the other bytes of the
        // record are written here only)
        bool other0 = false, other1 = false;
        if (!intra && !ibc) {
            if ((r1.w >> 16) & 0xFF) { other0 = refi0 >= 0; other1 = refi1 >= 0; }
            else if (a.dmvr_to_map && (r1.w >> 24)) {
                const int q0 = refi0 >= 0 ? (int)s_ref[refi0 * 2][1].z : 0, q1 = refi1 >= 0 ? (int)s_ref[refi1 * 2 + 1][1].z : 0;
                other0 = other1 = dmvr_applies(a.cur_poc, q0, q1);
            }
        }
        ScuRec *const mp = &a.maps[sy * a.w_scu + sx];
        if (!other0 && !other1) *(uint4 *)mp = rec;
        else {
            *(uint2 *)mp = make_uint2(rec.x, rec.y);
            if (!other0) ((uint32_t *)mp)[2] = rec.z;
            if (!other1) ((uint32_t *)mp)[3] = rec.w;
        }
    }
    if (intra || pred_mode == XGPU_MODE_IBC || ((r1.w >> 16) & 0xFF)) return false;   // IBC CUs are reconstructed with the intra CUs (k_intra); affine CUs: samples and sub-block vectors come from k_affine

    // ---- motion: clip like xevd_mv_clip (xevd_mc.c:435-467), variant from the UNCLIPPED vector ----
    const int maxl = (1 << a.bd_l) - 1, maxc = (1 << a.bd_c) - 1;
    int nl = 0;
    int mvt[2][2];
    const int mvs[2][2] = { { (int)(int16_t)(r1.x & 0xFFFF), (int)(int16_t)(r1.x >> 16) },
                            { (int)(int16_t)(r1.y & 0xFFFF), (int)(int16_t)(r1.y >> 16) } };
    const int refis[2] = { refi0, refi1 };
#pragma unroll
    for (int l = 0; l < 2; l++) {
        int mx = mvs[l][0], my = mvs[l][1];
        const int qx = cu_x << 2, qy = cu_y << 2, qw = cw << 2, qh = chh << 2;
        const int min_c = -(128 << 2), max_x = (a.pic_w - 1 + 128) << 2, max_y = (a.pic_h - 1 + 128) << 2;
        if (qx + mvs[l][0] < min_c) mx = min_c - qx;
        if (qy + mvs[l][1] < min_c) my = min_c - qy;
        if (qx + mvs[l][0] + qw - 4 > max_x) mx = max_x - qx - qw + 4;
        if (qy + mvs[l][1] + qh - 4 > max_y) my = max_y - qy - qh + 4;
        mvt[l][0] = (int)(int16_t)mx; mvt[l][1] = (int)(int16_t)my;
    }
    bool use[2] = { refi0 >= 0, refi1 >= 0 };
    // POC of the two references (0 when the list is unused: the tests below look at them only with both lists in use)
    const int poc0 = use[0] ? (int)s_ref[refi0 * 2][1].z : 0, poc1 = use[1] ? (int)s_ref[refi1 * 2 + 1][1].z : 0;
    if (use[0] && use[1] && poc0 == poc1 && mvt[0][0] == mvt[1][0] && mvt[0][1] == mvt[1][1])
        use[1] = false;                                               // identical motion, xevd_mc.c:512-519
    // a DMVR candidate whose references are POC-symmetric is refined and predicted by k_dmvr (its map record, with the unrefined vectors, is written)
    if ((r1.w >> 24) && dmvr_applies(a.cur_poc, poc0, poc1)) return false;

    // residual of this SCU (zero where nothing is coded); the loads are issued before the filtering
    uint32_t rl[8] = {0, 0, 0, 0, 0, 0, 0, 0}, ru[2] = {0, 0}, rv[2] = {0, 0};
    auto load_resid = [&]() {
        if (!in_tu) return;
        uint32_t off = coef_off;
        const int cwc = tu_w >> 1;
        if (cbf & 1) {
            const int16_t *r = a.resid + off + ly * tu_w + lx;
#pragma unroll
            for (int k = 0; k < 4; k++) { const uint2 v = *(const uint2 *)(r + k * tu_w); rl[k * 2] = v.x; rl[k * 2 + 1] = v.y; }
            off += tu_w * tu_h;
        }
        if (cbf & 2) {
            const int16_t *r = a.resid + off + (ly >> 1) * cwc + (lx >> 1);
            ru[0] = *(const uint32_t *)r; ru[1] = *(const uint32_t *)(r + cwc);
            off += cwc * (tu_h >> 1);
        }
        if (cbf & 4) {
            const int16_t *r = a.resid + off + (ly >> 1) * cwc + (lx >> 1);
            rv[0] = *(const uint32_t *)r; rv[1] = *(const uint32_t *)(r + cwc);
        }
    };
    if (MODE != 0) {
        // ---- wave-uniform CU (k_inter_region / k_inter_tile): the reference windows travel straight into LDS (global_load_lds: no staging registers, no LDS
        //      store pass), list by list through ONE window block, the second list's luma window requested as soon as the first list's luma passes are done with
        //      the block (it lands under the chroma passes), its chroma windows behind the chroma passes (they land under the second list's luma passes).
        //      Loads return in order, so `vmcnt(n)` = everything but the last n requests of this wave has landed.
        //      MODE 2: the block is the workgroup's (71x71 + 2 x 35x35, every thread requests its chunks of it - rm), a workgroup barrier stands between landing and
        //      reading and between reading and the next request into the same place.  MODE 1: the wave's own block (39x39 + 2 x 19x19), wave-local ordering only.
        constexpr bool REGION = MODE == 2;
        constexpr int WS_L = REGION ? REG_W_STRIDE : UW_STRIDE, WS_C = REGION ? REG_C_STRIDE : UCW_STRIDE;
        constexpr int L_SAMPLES = REGION ? REG_W_SAMPLES : UT_L_SAMPLES, C_SAMPLES = REGION ? REG_C_SAMPLES : 19 * UCW_STRIDE;
        // the intermediate rows of the two passes.  MODE 2: a block per wave behind the windows.  MODE 1: IN the window they come from - intermediate row r (80 bytes) lies in
        // window rows <= r (96 bytes each) and a pass of the wave reads all its window rows before it writes (LDS executes a wave's instructions in order), so a row
        // is overwritten only after it has been read: 6.9 KB of LDS per wave instead of 10
        int16_t *const I = REGION ? W + L_SAMPLES + 2 * C_SAMPLES + wave * REG_I_SAMPLES : W;
        int16_t *const IC = REGION ? I : W + L_SAMPLES;
        // the wave's 39x39 (19x19) part of the window block
        const int16_t *const Wy0 = W + (REGION ? ((wave >> 1) << 5) * REG_W_STRIDE + ((wave & 1) << 5) : 0);
        const int16_t *const Wu0 = W + L_SAMPLES + (REGION ? ((wave >> 1) << 4) * REG_C_STRIDE + ((wave & 1) << 4) : 0);
        const int wx = __builtin_amdgcn_readfirstlane(x) & (REGION ? ~63 : ~31), wy = __builtin_amdgcn_readfirstlane(y) & (REGION ? ~63 : ~31);      // the block's first sample
        const int tid = (wave << 6) + lane;
        auto list_pos = [&](int l, int &px, int &py) { px = (wx << 2) + (l ? mvt[1][0] : mvt[0][0]); py = (wy << 2) + (l ? mvt[1][1] : mvt[0][1]); };
        auto list_refs = [&](int l, gs16 &ry_, gs16 &ru_, gs16 &rv_) {
            const int ri = l ? refi1 : refi0;
            const uint4 e0 = s_ref[ri * 2 + l][0], e1 = s_ref[ri * 2 + l][1];
            ry_ = (gs16)(((uint64_t)e0.y << 32) | e0.x); ru_ = (gs16)(((uint64_t)e0.w << 32) | e0.z); rv_ = (gs16)(((uint64_t)e1.y << 32) | e1.x);
        };
        // every chunk at a 16-byte aligned address (rows start 256-byte aligned): the window's first sample is then `& 7` samples into the first chunk - an
        // unaligned 16-byte load costs the vector L1 2.4x the accesses of an aligned one (tools/ubench/win_bw.hip)
        auto request_luma = [&](int l) {
            gs16 ry_, ru_, rv_; int px, py;
            list_refs(l, ry_, ru_, rv_); list_pos(l, px, py);
            const gs16 p = ry_ + ((py >> 2) - 3) * a.s_l + (((px >> 2) - 3) & ~7);
            char LDS_AS *const d = (char LDS_AS *)W;
            // A vector with a whole-sample component needs neither the three rows above and four below the block (the vertical pass only copies rows 3 .. 3 + size) nor
            // the samples left of sample 3 and right of sample 3 + size of a row (the horizontal pass only picks them): those chunks are not requested (wave- /
            // workgroup-uniform here, the CU is one; the slots keep what they held and nothing reads it).  Same idea as the split role's lanes (mc_scu_list).
#ifdef XGPU_NO_LANE_PRED
            const bool fy = true, fx = true;
#else
            const bool fy = ((l ? mvs[1][1] : mvs[0][1]) & 3) != 0, fx = ((l ? mvs[1][0] : mvs[0][0]) & 3) != 0;
#endif
            constexpr int SIZE = REGION ? 64 : 32;
            // the samples of a row, counted from its first chunk, that the list needs: the window's SIZE + 7 from where it starts inside that chunk - the last chunk of
            // a row is only there for windows that start at its sample 2 and beyond -, or the SIZE of a copy
            const int mis = ((px >> 2) - 3) & 7, s0 = fx ? mis : mis + 3, s1 = fx ? mis + SIZE + 7 : mis + 3 + SIZE;
            auto wanted = [&](int row, int k) { return (fy || (row >= 3 && row < 3 + SIZE)) && 8 * k + 8 > s0 && 8 * k < s1; };
            if (REGION) {
#pragma unroll
                for (int it = 0; it < 3; it++)       // 71 rows x 10 chunks, thread t takes chunks t, t + 256, t + 512: LDS offset = chunk * 16
                    if (rm->y[it] >= 0 && wanted(rm->y[it] >> 8, rm->y[it] & 127)) __builtin_amdgcn_global_load_lds((const GAS void *)(p + (rm->y[it] >> 8) * a.s_l + 8 * (rm->y[it] & 127)), (LDS_AS void *)(d + (256 * it + 64 * wave) * 16), 16, 0, 0);
            } else {
                const int row0 = (lane * 171) >> 10, k = lane - row0 * 6;
#pragma unroll
                for (int it = 0; it < 4; it++)       // 39 rows x 6 chunks, 10 rows per request (lanes 60..63 idle)
                    if (lane < (it < 3 ? 60 : 54) && wanted(row0 + 10 * it, k)) __builtin_amdgcn_global_load_lds((const GAS void *)(p + fm.gy + 10 * it * a.s_l), (LDS_AS void *)(d + it * 960), 16, 0, 0);
            }
        };
        auto request_chroma = [&](int l) {
            gs16 ry_, ru_, rv_; int px, py;
            list_refs(l, ry_, ru_, rv_); list_pos(l, px, py);
            const int off = ((py >> 3) - 1) * a.s_c + (((px >> 3) - 1) & ~7);
            char LDS_AS *const d = (char LDS_AS *)(W + L_SAMPLES);
            // (as request_luma: the chroma windows are SIZE / 2 + 3 samples from sample `cmis` of the first chunk - three times out of four they end before the row's
            //  last chunk -, a whole-sample component needs SIZE / 2 rows / samples from 1)
#ifdef XGPU_NO_LANE_PRED
            const bool fy = true, fx = true;
#else
            const bool fy = ((l ? mvs[1][1] : mvs[0][1]) & 7) != 0, fx = ((l ? mvs[1][0] : mvs[0][0]) & 7) != 0;
#endif
            constexpr int CSIZE = REGION ? 32 : 16;
            const int cmis = ((px >> 3) - 1) & 7, s0 = fx ? cmis : cmis + 1, s1 = fx ? cmis + CSIZE + 3 : cmis + 1 + CSIZE;
            auto wanted = [&](int row, int k) { return (fy || (row >= 1 && row < 1 + CSIZE)) && 8 * k + 8 > s0 && 8 * k < s1; };
            if (REGION) {
#pragma unroll
                for (int it = 0; it < 2; it++)       // 2 planes x 35 rows x 6 chunks
                    if (rm->c[it] >= 0 && wanted(rm->c[it] >> 8, rm->c[it] & 127)) __builtin_amdgcn_global_load_lds((const GAS void *)(((rm->c[it] & 128) ? rv_ : ru_) + off + (rm->c[it] >> 8) * a.s_c + 8 * (rm->c[it] & 127)), (LDS_AS void *)(d + (256 * it + 64 * wave) * 16), 16, 0, 0);
            } else {
#pragma unroll
                for (int it = 0; it < 3; it++) {     // 2 planes x 19 rows x 4 chunks
                    const int idx = lane + 64 * it, plane = idx >= 76, j = idx - 76 * plane;
                    if (idx < 152 && wanted(j >> 2, j & 3)) __builtin_amdgcn_global_load_lds((const GAS void *)((plane ? rv_ : ru_) + off + (j >> 2) * a.s_c + 8 * (j & 3)), (LDS_AS void *)(d + it * 1024), 16, 0, 0);
                }
            }
        };
        constexpr int N_LUMA_REQ = REGION ? 3 : 4, N_CHROMA_REQ = REGION ? 2 : 3;
        const int l_first = use[0] ? 0 : 1, n_lists = (int)use[0] + (int)use[1];
        TRACE_MARK(tr, 2);                                     // records decoded, map written
        load_resid();
        if (n_lists) { request_luma(l_first); __builtin_amdgcn_sched_barrier(0); request_chroma(l_first); }
        TRACE_MARK(tr, 3);                                     // first requests issued
#pragma unroll 1
        for (int i = 0; i < n_lists; i++) {
            const int l = i ? 1 : l_first;
            const bool more = i + 1 < n_lists;
            int px, py;
            list_pos(l, px, py);
            const int mvx = l ? mvs[1][0] : mvs[0][0], mvy = l ? mvs[1][1] : mvs[0][1];
            const int ldx = (mvx & 3) != 0, ldy = (mvy & 3) != 0, cdx = (mvx & 7) != 0, cdy = (mvy & 7) != 0;
            uint32_t o[8], ou[2], ov[2];
            if (REGION) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); __syncthreads(); }      // this list's windows have landed, all four waves' requests
            else { asm volatile("s_waitcnt vmcnt(%0)" :: "n"(N_CHROMA_REQ) : "memory"); wave_lds_sync(); }      // the luma window has (the chroma requests may still be out)
            TRACE_MARK(tr, i ? 8 : 4);                         // waited for the list's luma window
            {
                const uint4 th = s_ltap[ldx ? ((px & 3) << 2) : 16], tv = s_ltap[ldy ? ((py & 3) << 2) : 16];
                const uint32_t ch[4] = { th.x, th.y, th.z, th.w }, cv[4] = { tv.x, tv.y, tv.z, tv.w };
                const Regime rg = regime(ldx, ldy, a.bd_l);
                const int mis = ((px >> 2) - 3) & 7, odd = mis & 1;
                const int16_t *const Wy = Wy0 + (mis & ~1);
                if (ldx) { if (ldy) luma_tile_filter<true, true, WS_L, true>(Wy, ch, cv, rg, maxl, I, lane, o, odd); else luma_tile_filter<true, false, WS_L, true>(Wy, ch, cv, rg, maxl, I, lane, o, odd); }
                else     { if (ldy) luma_tile_filter<false, true, WS_L, true>(Wy, ch, cv, rg, maxl, I, lane, o, odd); else luma_tile_filter<false, false, WS_L, true>(Wy, ch, cv, rg, maxl, I, lane, o, odd); }
            }
            TRACE_MARK(tr, 5);                                 // luma passes
            if (more) {                                             // the luma window block is free: the second list's goes there while the chroma passes run
                if (REGION) __syncthreads(); else wave_lds_sync();
                request_luma(1);
            }
            if (!REGION) {
                if (more) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(N_LUMA_REQ) : "memory"); else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                wave_lds_sync();
            }
            TRACE_MARK(tr, 6);                                 // barrier / chroma window wait
            {
                const uint2 th = s_ctap[cdx ? ((px & 7) << 2) : 32], tv = s_ctap[cdy ? ((py & 7) << 2) : 32];
                const uint32_t c2h[2] = { th.x, th.y }, c2v[2] = { tv.x, tv.y };
                const Regime rg = regime(cdx, cdy, a.bd_c);
                const int mis = ((px >> 3) - 1) & 7, odd = mis & 1;
                const int16_t *const Wu = Wu0 + (mis & ~1), *const Wv = Wu + C_SAMPLES;
                if (cdx) { if (cdy) chroma_tile_filter<true, true, WS_C, true>(Wu, Wv, c2h, c2v, rg, maxc, IC, lane, ou, ov, odd); else chroma_tile_filter<true, false, WS_C, true>(Wu, Wv, c2h, c2v, rg, maxc, IC, lane, ou, ov, odd); }
                else     { if (cdy) chroma_tile_filter<false, true, WS_C, true>(Wu, Wv, c2h, c2v, rg, maxc, IC, lane, ou, ov, odd); else chroma_tile_filter<false, false, WS_C, true>(Wu, Wv, c2h, c2v, rg, maxc, IC, lane, ou, ov, odd); }
            }
            TRACE_MARK(tr, 7);                                 // chroma passes
            if (more) {
                if (REGION) __syncthreads(); else wave_lds_sync();
                request_chroma(1);
            }
            if (nl == 0) {
#pragma unroll
                for (int k = 0; k < 8; k++) pl[k] = o[k];
                pu[0] = ou[0]; pu[1] = ou[1]; pv[0] = ov[0]; pv[1] = ov[1];
            } else {
#pragma unroll
                for (int k = 0; k < 8; k++) pl[k] = avg2(pl[k], o[k]);
                pu[0] = avg2(pu[0], ou[0]); pu[1] = avg2(pu[1], ou[1]);
                pv[0] = avg2(pv[0], ov[0]); pv[1] = avg2(pv[1], ov[1]);
            }
            nl++;
        }
        (void)tid;
    } else {
        TRACE_MARK(tr, 2);
#pragma unroll
        for (int l = 0; l < 2; l++) {
            if (!use[l]) continue;
            const uint4 e0 = s_ref[refis[l] * 2 + l][0], e1 = s_ref[refis[l] * 2 + l][1];
            const gs16 ry_ = (gs16)(((uint64_t)e0.y << 32) | e0.x), ru_ = (gs16)(((uint64_t)e0.w << 32) | e0.z), rv_ = (gs16)(((uint64_t)e1.y << 32) | e1.x);
            const int mvx = mvs[l][0], mvy = mvs[l][1];
            // luma: quarter-pel position of this SCU = (x<<2) + clipped mv; phase in 1/16 = (pos&3)<<2
            const int px = (x << 2) + mvt[l][0], py = (y << 2) + mvt[l][1];
            const int ldx = (mvx & 3) != 0, ldy = (mvy & 3) != 0;
            const int cdx = (mvx & 7) != 0, cdy = (mvy & 7) != 0;
            uint32_t ch[4], cv[4], o[8], ou[2], ov[2];
            {
                const uint4 th = s_ltap[ldx ? ((px & 3) << 2) : 16], tv = s_ltap[ldy ? ((py & 3) << 2) : 16];
                ch[0] = th.x; ch[1] = th.y; ch[2] = th.z; ch[3] = th.w; cv[0] = tv.x; cv[1] = tv.y; cv[2] = tv.z; cv[3] = tv.w;
                // chroma: 1/8-pel position (x<<2)+mv in luma quarter-pel == chroma eighth-pel; phase in 1/32 = (pos&7)<<2
                const uint2 cth = s_ctap[cdx ? ((px & 7) << 2) : 32], ctv = s_ctap[cdy ? ((py & 7) << 2) : 32];
                const uint32_t c2h[2] = { cth.x, cth.y }, c2v[2] = { ctv.x, ctv.y };
                const gs16 p = ry_ + ((py >> 2) - 3) * a.s_l + (px >> 2) - 3;
                const int off = ((py >> 3) - 1) * a.s_c + (px >> 3) - 1;
                const Regime rgl = regime(ldx, ldy, a.bd_l), rgc = regime(cdx, cdy, a.bd_c);
                const bool wh = __ballot(ldx) != 0, wvv = __ballot(ldy) != 0;      // over the lanes that run this list
                const bool cwh = __ballot(cdx) != 0, cwv = __ballot(cdy) != 0;
#define MC_S(H, V) mc_scu_list<H, V>(p, a.s_l, ru_ + off, rv_ + off, a.s_c, ch, cv, rgl, maxl, c2h, c2v, rgc, maxc, cwh, cwv, o, ou, ov, ldx != 0, ldy != 0, cdx != 0, cdy != 0)
                if (wh) { if (wvv) MC_S(true, true); else MC_S(true, false); }
                else    { if (wvv) MC_S(false, true); else MC_S(false, false); }
#undef MC_S
            }
            if (nl == 0) {
#pragma unroll
                for (int k = 0; k < 8; k++) pl[k] = o[k];
                pu[0] = ou[0]; pu[1] = ou[1]; pv[0] = ov[0]; pv[1] = ov[1];
            } else {
#pragma unroll
                for (int k = 0; k < 8; k++) pl[k] = avg2(pl[k], o[k]);
                pu[0] = avg2(pu[0], ou[0]); pu[1] = avg2(pu[1], ou[1]);
                pv[0] = avg2(pv[0], ov[0]); pv[1] = avg2(pv[1], ov[1]);
            }
            nl++;
        }
    }
    TRACE_MARK(tr, 9);             // (split role: both lists' requests and arithmetic)
    if (MODE == 0) load_resid();      // (split role: requested behind the lists - its twelve registers in front of them were the kernel's fourth wave per SIMD)
    if (nl == 0) return false;     // inter CU without a valid reference: nothing predicted (does not occur in valid streams)

    // ---- residual add + clip (xevd_recon.c:35-71; the LUMA bit depth clips all three components, :75-90) ----
    if (cbf & 1) {
#pragma unroll
        for (int k = 0; k < 8; k++) pl[k] = recon2(pl[k], rl[k], maxl);
    }
    if (cbf & 2) { pu[0] = recon2(pu[0], ru[0], maxl); pu[1] = recon2(pu[1], ru[1], maxl); }
    if (cbf & 4) { pv[0] = recon2(pv[0], rv[0], maxl); pv[1] = recon2(pv[1], rv[1], maxl); }

    return true;
}

// ---------------------------------------------------------------------------------------------------------
// k_inter - ONE launch per picture: a workgroup per 64x64 region of the picture, in vertical strips XGPU_INTER_STRIP regions wide, row by row inside a strip; its four
// waves take one of three ROLES, which xgpu_batch_create has written into the region's entry of InterArgs.work:
//   region role  the region lies inside ONE CU: the four waves together, the reference windows requested once per workgroup into LDS they share (inter_tile<2>);
//   tile role    the wave's 32x32 tile lies inside one CU (whose region does not): the window in the wave's own LDS block (inter_tile<1>);
//   split role   every other tile: one lane per SCU, every lane on the CU that covers its SCU (owner map), with windows of its own (inter_tile<0>).
// Neighbouring tiles of different roles run at the same time on the same XCD and share the reference lines of their halos in its L2: as three launches, one per role
// (round 5's first form, each with its own register budget: 110 / 101 / 123 VGPRs), every class swept the whole picture on its own and the pass read 505 MB from HBM
// instead of 337 (profiles/round5_a_pmc.json), 156 us instead of 145.
// What a wave must know before it can ask for reference samples is ONE round trip away: the region's position follows from the workgroup's number, and the entry of
// `work`, the wave's item (region and tile role: position-indexed, with the CU's record in it) and the owner-map entries of its SCUs (split role) are all requested
// at once, whatever the role turns out to be.  (With the role-specific lists of the first form - work entry, then list item or list entry, owner entry, CU record -
// the chain in front of the first request was 2 - 4 dependent loads at ~2000 cycles each under load: 40 - 50 % of the life of a region- or tile-role wave, tools/archive/r5_n.sh,
// profiles/round5_exp_inter_wave_life.txt.)  The split role's tables (reference entries, filter taps: looked up per lane) are staged per WAVE: no workgroup barrier
// outside the region role, a tile-role wave does not wait for its neighbours' table loads.
// XCD-aware mapping: workgroup b runs on XCD b % 8 and every XCD has its own L2; XCD k takes the k-th contiguous eighth of the strip order - a compact patch of
// the picture whose vertical halos are still in its L2 when the row below is processed.
// Round 5 also built, measured bit-exact and dropped finer classes for the split tiles (16x16 blocks inside one CU sharing a 23x23 window by global_load_lds, as a path
// of this kernel, as tasks sorted inside the workgroup, and as a fourth and fifth launch), and several entries per workgroup with the next entry's chain loaded
// during the current one (tools/patches/k_inter_items_loop_r5.diff: the loop keeps the scalar state of all three roles alive side by side - 165 - 170 VGPRs for
// constants the scalar registers no longer hold, three waves per SIMD): DESIGN.md 3 has the numbers and what they say about the bound.
// History of the single-kernel form (rounds 1-4: persistent waves, class-sorted pieces, occupancy sweeps, non-temporal hints, region order) is in DESIGN.md 3 too.
// ---------------------------------------------------------------------------------------------------------
__device__ __forceinline__ int xcd_slice(int block, int grid) { return (block & 7) * (grid >> 3) + (block >> 3); }

__device__ __forceinline__ void store_scu(const InterArgs &a, int sx, int sy, const uint32_t pl[8], const uint32_t pu[2], const uint32_t pv[2])
{
    const int x = sx << 2, y = sy << 2;
    int16_t *dy = a.cur_y + y * a.s_l + x;
#pragma unroll
    for (int k = 0; k < 4; k++) *(uint2 *)(dy + k * a.s_l) = make_uint2(pl[k * 2], pl[k * 2 + 1]);
    const int coff = (y >> 1) * a.s_c + (x >> 1);
    *(uint32_t *)(a.cur_u + coff) = pu[0];
    *(uint32_t *)(a.cur_u + coff + a.s_c) = pu[1];
    *(uint32_t *)(a.cur_v + coff) = pv[0];
    *(uint32_t *)(a.cur_v + coff + a.s_c) = pv[1];
}
// the reference table as inter_tile reads it ([index * 2 + list][2 halves]) and the tap tables, where they lie for wave-uniform indices
#define ARG_REFS(a)  ((const uint4 (*)[2])&(a).refp[0][0])
#define ARG_LTAPS(a) ((const uint4 *)&k_luma_taps[(a).admvp][0][0])
#define ARG_CTAPS(a) ((const uint2 *)&k_chroma_taps[(a).admvp][0][0])
static_assert(sizeof(RefEntry) == 32, "a reference entry is two 16-byte halves");
static_assert(XGPU_INTER_STRIP == 16, "the strip arithmetic below shifts by four");

struct SplitTables { uint4 ref[XGPU_MAX_REFS * 2][2]; uint4 ltap[17]; uint2 ctap[33]; uint2 pad; };      // RefEntry [idx][list]; luma taps of this sequence's table, [16] = identity; chroma taps

__device__ __forceinline__ void inter_fused_body(const InterArgs &a)
{
    __shared__ __attribute__((aligned(16))) int16_t s_win[REG_SAMPLES];      // region role: the shared windows + intermediates; tile role: four wave blocks
    static_assert(4 * UNI_SAMPLES <= REG_SAMPLES, "the four waves' tile blocks fit into the region block");
    __shared__ SplitTables s_tab[4];                                         // split role: per wave
    const int idx = xcd_slice(blockIdx.x, gridDim.x);
    if (idx >= a.n_work) return;
    const int t = threadIdx.x, lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6);
#ifdef XGPU_INTER_TRACE
    InterTrace tr = { __builtin_amdgcn_s_memtime(), 0, lane == 0 && (idx % 61) == 7 };      // a sample of the workgroups: the atomics of all of them on 33 addresses stretched the kernel sixfold
#endif
    // the region's place in the strip order (xgpu_batch_create's): full strips of 16 x regions_y entries, then the narrower last strip
    int rx, ry;
    if (idx < a.full_entries) {
        const int s = (int)__umulhi((uint32_t)idx, a.magic_strip), r = idx - s * a.strip_entries;
        ry = r >> 4; rx = (s << 4) + (r & 15);
    } else {
        const int r = idx - a.full_entries;
        ry = a.magic_last ? (int)__umulhi((uint32_t)r, a.magic_last) : r; rx = (a.regions_x & ~15) + r - ry * (a.regions_x & 15);
    }
    const int sx = (rx << 4) + ((wave & 1) << 3) + (lane & 7), sy = (ry << 4) + ((wave >> 1) << 3) + (lane >> 3);
    // everything a role needs first, requested at once
    const uint32_t kinds = a.work[idx];
    const uint4 *const item = (const uint4 *)&a.items[idx * 4 + wave];
    const uint4 e = item[0], i0 = item[1], i1 = item[2];
    const uint32_t own = (sx < (a.pic_w >> 2) && sy < (a.pic_h >> 2)) ? a.owner[sy * a.w_scu + sx] : OWNER_NONE;
    const LaneMap fm0 = { 0, 0 };
    uint32_t pl[8], pu[2], pv[2];
    if (kinds == XGPU_WORK_REGION) {
#ifdef XGPU_INTER_TRACE
        tr.role = 0;
        if (e.y == 0xFFFFFFFFu) tr.on = false;                 // (uses the item: the mark stands behind its arrival)
        if (tr.on) atomicAdd(&g_inter_trace[0][15], 1ull);
        TRACE_MARK(tr, 0);
#endif
        const RegionMap rmap = region_map(t);
        if (inter_tile<2>(a, i0, i1, true, sx, sy, lane, s_win, fm0, ARG_REFS(a), ARG_LTAPS(a), ARG_CTAPS(a), pl, pu, pv, &rmap, wave, e.y TRACE_PASS)) store_scu(a, sx, sy, pl, pu, pv);
        TRACE_MARK(tr, 10);
        return;
    }
    const uint32_t kind = (kinds >> (2 * wave)) & 3u;
    if (kind == 1) {
#ifdef XGPU_INTER_TRACE
        tr.role = 1;
        if (e.y == 0xFFFFFFFFu) tr.on = false;
        if (tr.on) atomicAdd(&g_inter_trace[1][15], 1ull);
        TRACE_MARK(tr, 0);
#endif
        LaneMap fm;
        {
            const int row0 = (lane * 171) >> 10, k = lane - row0 * 6;            // luma window: 10 rows x 6 aligned chunks of 8 samples per request (lanes 60..63 idle)
            fm.gy = row0 * a.s_l + 8 * k; fm.ly = row0 * UW_STRIDE + 8 * k;
        }
        if (inter_tile<1>(a, i0, i1, true, sx, sy, lane, s_win + wave * UNI_SAMPLES, fm, ARG_REFS(a), ARG_LTAPS(a), ARG_CTAPS(a), pl, pu, pv, nullptr, 0, e.y TRACE_PASS)) store_scu(a, sx, sy, pl, pu, pv);
        TRACE_MARK(tr, 10);
    } else if (kind == 2) {
#ifdef XGPU_INTER_TRACE
        tr.role = 2;
        if (tr.on) atomicAdd(&g_inter_trace[2][15], 1ull);
        TRACE_MARK(tr, 0);
#endif
        // second (and last) link of the chain: the CU records of the lanes' owners, and beside them the wave's copy of the tables: 68 + 17 + 33 entries, two per lane
        SplitTables &tb = s_tab[wave];
        static_assert(XGPU_MAX_REFS * 4 == 68, "lanes 0..63 and 0..3 of the second round take the reference entries' halves");
        const uint4 tab0 = ((const uint4 *)&a.refp[0][0])[lane];
        uint4 tab1 = make_uint4(0, 0, 0, 0);
        if (lane < 4) tab1 = ((const uint4 *)&a.refp[0][0])[64 + lane];
        else if (lane < 4 + 17) tab1 = *(const uint4 *)k_luma_taps[a.admvp][lane - 4];
        else if (lane < 4 + 17 + 33) { const uint2 v = *(const uint2 *)k_chroma_taps[a.admvp][lane - 21]; tab1.x = v.x; tab1.y = v.y; }
        const bool ok = own < (uint32_t)a.n_cu;                    // unowned (another batch's SCU) or not an index of this batch
        uint4 c0 = make_uint4(0, 0, 0, 0), c1 = c0;
        if (ok) { c0 = ((const uint4 *)&a.cus[own])[0]; c1 = ((const uint4 *)&a.cus[own])[1]; }
        tb.ref[lane >> 1][lane & 1] = tab0;
        if (lane < 4) tb.ref[32 + (lane >> 1)][lane & 1] = tab1;
        else if (lane < 4 + 17) tb.ltap[lane - 4] = tab1;
        else if (lane < 4 + 17 + 33) tb.ctap[lane - 21] = make_uint2(tab1.x, tab1.y);
        wave_lds_sync();
        TRACE_MARK(tr, 1);
        if (inter_tile<0>(a, c0, c1, ok, sx, sy, lane, nullptr, fm0, tb.ref, tb.ltap, tb.ctap, pl, pu, pv, nullptr, 0, own TRACE_PASS)) store_scu(a, sx, sy, pl, pu, pv);
        TRACE_MARK(tr, 10);
    }
}
__global__ __launch_bounds__(256) void k_inter(const InterArgs a) { inter_fused_body(a); }

// (the measurement knob XEVD_HIP_INTER_ALL_FIRST and its second kernel are gone: mc_scu_list's one form requests a list's luma window at once at four waves per SIMD)
void launch_inter(xgpu_ctx *c, const InterArgs &a)
{
    if (!a.n_work) return;
    const dim3 grid((unsigned)(((a.n_work + 7) >> 3) << 3));
    hipLaunchKernelGGL(k_inter, grid, dim3(256), 0, c->stream, a);
}

// ---------------------------------------------------------------------------------------------------------
// fine-grained shim: one block through the same device functions with the reference's XEVD_MC_L / XEVD_MC_C
// call shape (xevd_mc.h:47-49).  One lane per 4x4 (luma) / 2x2 (chroma) sub-block.
// ---------------------------------------------------------------------------------------------------------
__global__ void k_test_mc(const int16_t *plane, int stride, int ref_x, int ref_y, int has_dx, int has_dy, int gmv_x, int gmv_y,
                          int16_t *pred, int w, int h, int bd, int luma, int admvp)
{
    const int bs = luma ? 4 : 2;
    const int nbx = w / bs, nby = h / bs;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nbx * nby) return;
    const int bx = (i % nbx) * bs, by = (i / nbx) * bs;
    const int maxv = (1 << bd) - 1;
    const int16_t *ref = plane + ref_y * stride + ref_x;
    if (luma) {
        const uint32_t *th = k_luma_taps[admvp][has_dx ? (gmv_x & 15) : 16], *tv = k_luma_taps[admvp][has_dy ? (gmv_y & 15) : 16];
        uint32_t ch[4] = { th[0], th[1], th[2], th[3] }, cv[4] = { tv[0], tv[1], tv[2], tv[3] }, o[8];
        const gs16 p = (gs16)ref + ((gmv_y >> 4) - 3 + by) * stride + (gmv_x >> 4) - 3 + bx;
        if (has_dx) { if (has_dy) mc_luma_4x4<true, true>(p, stride, ch, cv, regime(has_dx, has_dy, bd), maxv, o); else mc_luma_4x4<true, false>(p, stride, ch, cv, regime(has_dx, has_dy, bd), maxv, o); }
        else        { if (has_dy) mc_luma_4x4<false, true>(p, stride, ch, cv, regime(has_dx, has_dy, bd), maxv, o); else mc_luma_4x4<false, false>(p, stride, ch, cv, regime(has_dx, has_dy, bd), maxv, o); }
        for (int r = 0; r < 4; r++) {
            *(uint32_t *)(pred + (by + r) * w + bx) = o[r * 2];
            *(uint32_t *)(pred + (by + r) * w + bx + 2) = o[r * 2 + 1];
        }
    } else {
        const uint32_t *th = k_chroma_taps[admvp][has_dx ? (gmv_x & 31) : 32], *tv = k_chroma_taps[admvp][has_dy ? (gmv_y & 31) : 32];
        uint32_t ch[2] = { th[0], th[1] }, cv[2] = { tv[0], tv[1] }, o[2];
        const gs16 p = (gs16)ref + ((gmv_y >> 5) - 1 + by) * stride + (gmv_x >> 5) - 1 + bx;
        if (has_dx) { if (has_dy) mc_chroma_2x2<true, true>(p, stride, ch, cv, regime(has_dx, has_dy, bd), maxv, o); else mc_chroma_2x2<true, false>(p, stride, ch, cv, regime(has_dx, has_dy, bd), maxv, o); }
        else        { if (has_dy) mc_chroma_2x2<false, true>(p, stride, ch, cv, regime(has_dx, has_dy, bd), maxv, o); else mc_chroma_2x2<false, false>(p, stride, ch, cv, regime(has_dx, has_dy, bd), maxv, o); }
        *(uint32_t *)(pred + by * w + bx) = o[0];
        *(uint32_t *)(pred + (by + 1) * w + bx) = o[1];
    }
}

void launch_test_mc(xgpu_ctx *c, const int16_t *plane, int stride, int ref_x, int ref_y, int has_dx, int has_dy,
                    int gmv_x, int gmv_y, int16_t *pred, int w, int h, int bd, int luma)
{
    const int bs = luma ? 4 : 2;
    const int n = (w / bs) * (h / bs);
    hipLaunchKernelGGL(k_test_mc, dim3((n + 63) / 64), dim3(64), 0, c->stream, plane, stride, ref_x, ref_y, has_dx, has_dy,
                       gmv_x, gmv_y, pred, w, h, bd, luma, c->sp.tool_admvp);
}

// fn_recon's call shape (src_base/xevd_def.h:1466, xevd_recon, xevd_recon.c:35-71) on the kernels' packed residual add: rec = clip(pred + coef)
// with the 16-bit wrap of the reference's s16 sum, or the clipped prediction when the block has no coefficients.  One lane per sample pair.
__global__ void k_test_recon(const int16_t *coef, const int16_t *pred, int is_coef, int cuw, int cuh, int16_t *rec, int s_rec, int bd)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x, pw = cuw >> 1;
    if (i >= pw * cuh) return;
    const int y = i / pw, x = (i - y * pw) * 2;
    const uint32_t p = *(const uint32_t *)(pred + y * cuw + x);
    // without coefficients the reference still clips the prediction (xevd_recon.c:41-48)
    const uint32_t r = recon2(p, is_coef ? *(const uint32_t *)(coef + y * cuw + x) : 0u, (1 << bd) - 1);
    rec[y * s_rec + x] = (int16_t)(r & 0xFFFF); rec[y * s_rec + x + 1] = (int16_t)(r >> 16);
}
void launch_test_recon(xgpu_ctx *c, const int16_t *coef, const int16_t *pred, int is_coef, int cuw, int cuh, int16_t *rec, int s_rec, int bd)
{
    const int n = (cuw >> 1) * cuh;
    hipLaunchKernelGGL(k_test_recon, dim3((n + 63) / 64), dim3(64), 0, c->stream, coef, pred, is_coef, cuw, cuh, rec, s_rec, bd);
}

