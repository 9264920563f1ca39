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
//   * the 11x11 (luma) / 5x5 (chroma) reference windows are read straight from HBM/L2 with 16-byte loads at the
//     2-byte-aligned sample address (gfx950 runs in unaligned-access mode); neighbouring lanes share the halo
//     through the vector L1, workgroups are mapped to XCDs in contiguous bands so vertical halos share an L2.
//   * FIRs run on packed s16 pairs with v_dot2c_i32_i16 (two taps per instruction); the four rounding regimes of
//     the reference (copy / H-only / V-only / 2-D) are one code path with per-lane tap vectors, shifts and
//     offsets, so lanes with different sub-pel classes do not diverge.
//   * no MFMA: these are 4/8-tap integer FIRs, bounded by load/issue rate and HBM, not by dense contraction.
#include "xgpu_internal.h"

#include <hip/hip_ext.h>
#include "mc_filters.h"

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
// k_inter_tile's block per wave: luma window (4 requests x 60 chunks = 40 rows), the two chroma windows (152 chunks; 3 requests x 64 slots), intermediate rows
#define UT_L_SAMPLES (40 * UW_STRIDE)
#define UT_C_SAMPLES (3 * 64 * 8)
#define UNI_SAMPLES   (UT_L_SAMPLES + UT_C_SAMPLES + 39 * UI_STRIDE)
static_assert(2 * 19 * UCW_STRIDE <= UT_C_SAMPLES, "both chroma windows fit");

__device__ __forceinline__ void wave_lds_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// p = reference sample at (tile x - 3, tile y - 3); lane l owns the SCU (l & 7, l >> 3) of the tile.  o[] as mc_luma_4x4.
// The fetch is split from the filtering so that a wave has the windows of both lists (and its residual) in flight at once.
struct TileFetch { uint4 y[4]; uint4 c[3]; };
// a lane's chunk of the tile windows: offsets into the reference plane (samples) and into the wave's LDS window.  Luma: 12 rows x 5 chunks of
// 8 samples per pass (lanes 60..63 idle), four passes cover the 39 rows; chroma: 19 rows x 3 chunks per plane (lanes 57..63 idle).  The offsets
// depend on the lane alone: computed once per kernel, every pass adds a constant.
struct LaneMap { int gy, ly; };
// The chunks are 16-byte ALIGNED pieces of the reference rows (p / pu / pv = the chunk that holds the window's first sample; rows start 256-byte aligned): an unaligned
// 16-byte load costs the vector L1 2.4x the accesses (round 4).  Luma: up to 7 + 39 samples = 6 chunks per row, 10 rows x 6 chunks per pass (lanes 60..63 idle), four
// passes; chroma: up to 7 + 19 samples = 4 chunks, 2 planes x 19 rows x 4 chunks = 152 loads in three passes of 64 lanes.
__device__ __forceinline__ void tile_fetch(gs16 p, int s, gs16 pu, gs16 pv, int sc, int lane, const LaneMap fm, TileFetch &f)
{
#pragma unroll
    for (int it = 0; it < 4; it++)
        if (lane < (it < 3 ? 60 : 54)) f.y[it] = gload16(p + fm.gy + 10 * it * s);
#pragma unroll
    for (int it = 0; it < 3; it++) {
        const int idx = lane + 64 * it, plane = idx >= 76, j = idx - 76 * plane;
        if (idx < 152) f.c[it] = gload16((plane ? pv : pu) + (j >> 2) * sc + 8 * (j & 3));
    }
}

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
template <bool H, bool V>
__device__ __forceinline__ void mc_luma_tile(const uint4 v[4], const uint32_t ch[4], const uint32_t cv[4], Regime rg, int maxv,
                                             int16_t *W, int16_t *I, int lane, const LaneMap fm, uint32_t o[8], int mis)      // mis: the window's first sample inside its first chunk
{
#pragma unroll
    for (int it = 0; it < 4; it++)
        if (lane < (it < 3 ? 60 : 54)) *(uint4 *)(W + fm.ly + 10 * it * UW_STRIDE) = v[it];
    wave_lds_sync();
    luma_tile_filter<H, V, UW_STRIDE, true>(W + (mis & ~1), ch, cv, rg, maxv, I, lane, o, mis & 1);
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
// pu / pv = reference sample at (tile x - 1, tile y - 1) of the plane.
template <bool H, bool V>
__device__ __forceinline__ void mc_chroma_tile(const uint4 v[3], const uint32_t ch[2], const uint32_t cv[2],
                                               Regime rg, int maxv, int16_t *W, int16_t *I, int lane, uint32_t ou[2], uint32_t ov[2], int mis)
{
#pragma unroll
    for (int it = 0; it < 3; it++) {
        const int idx = lane + 64 * it, plane = idx >= 76, j = idx - 76 * plane;
        if (idx < 152) *(uint4 *)(W + (plane ? 19 * UCW_STRIDE : 0) + (j >> 2) * UCW_STRIDE + 8 * (j & 3)) = v[it];
    }
    wave_lds_sync();
    chroma_tile_filter<H, V, UCW_STRIDE, true>(W + (mis & ~1), W + 19 * UCW_STRIDE + (mis & ~1), ch, cv, rg, maxv, I, lane, ou, ov, mis & 1);
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
struct RegionFetch { uint4 y[3]; uint4 c[2]; };
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
__device__ __forceinline__ void region_fetch(gs16 p, int s_l, gs16 pu, gs16 pv, int s_c, const RegionMap m, RegionFetch &f)
{
#pragma unroll
    for (int it = 0; it < 3; it++) if (m.y[it] >= 0) f.y[it] = gload16(p + (m.y[it] >> 8) * s_l + 8 * (m.y[it] & 127));
#pragma unroll
    for (int it = 0; it < 2; it++) if (m.c[it] >= 0) f.c[it] = gload16(((m.c[it] & 128) ? pv : pu) + (m.c[it] >> 8) * s_c + 8 * (m.c[it] & 127));
}
__device__ __forceinline__ void region_store(const RegionFetch &f, const RegionMap m, int16_t *SH)
{
#pragma unroll
    for (int it = 0; it < 3; it++) if (m.y[it] >= 0) *(uint4 *)(SH + (m.y[it] >> 8) * REG_W_STRIDE + 8 * (m.y[it] & 127)) = f.y[it];
#pragma unroll
    for (int it = 0; it < 2; it++) if (m.c[it] >= 0) *(uint4 *)(SH + REG_W_SAMPLES + ((m.c[it] & 128) ? REG_C_SAMPLES : 0) + (m.c[it] >> 8) * REG_C_STRIDE + 8 * (m.c[it] & 127)) = f.c[it];
}

// ---------------------------------------------------------------------------------------------------------
// k_inter_split's shared quadrant windows.  A request (global_load_lds, 16 bytes per lane) writes 64 x 16 = 1024 consecutive bytes: lane (q, rs, col) - quadrant, row
// slot, column chunk - lands at q * 256 + rs * 64 + col * 16, i.e. the block of request k holds rows 4k .. 4k + 3 of the four quadrants' windows as 64-byte rows
// (luma: 32 samples from the 16-byte aligned chunk that holds the window's first sample; chroma: 16 samples of U from the even sample at or below the window's
// first, then 16 of V).  Blocks start SQ_BLK_B = 1024 + 16 bytes apart, so that rows 4 apart - the lanes jy, jy + 1 of a quadrant read them at the same moment -
// start 4 banks apart.  Row r of quadrant q: q * 256 + sq_row(r).  (tools/ubench/glds_probe.hip: a 12-byte request also strides 16 bytes per lane - its rows would
// have holes; inactive lanes write nothing.)
// ---------------------------------------------------------------------------------------------------------
#define LDS_AS __attribute__((address_space(3)))
#define SQ_BLK_B   1040
#define SQ_L_BYTES (6 * SQ_BLK_B)         // 23 luma rows (24 slots)
#define SQ_C_BYTES (3 * SQ_BLK_B)         // 11 chroma rows (12 slots)
#define SQ_BYTES   (SQ_L_BYTES + SQ_C_BYTES)
__device__ __forceinline__ int sq_row(int r) { return (r >> 2) * SQ_BLK_B + (r & 3) * 64; }

// mc_luma_4x4 / mc_chroma_2x2 (mc_filters.h) with the window read from the shared block instead of the reference picture: w = the dword of row 0 that holds the
// lane's first window sample (the sample is the dword's high half when `odd`), r0 = the lane's first row in the quadrant's window.
template <bool H, bool V>
__device__ __forceinline__ void mc_luma_4x4_w(const char LDS_AS *w, int r0, int odd, const uint32_t ch[4], const uint32_t cv[4], Regime rg, int maxv, uint32_t o[8])
{
    int acc[4][4];
    int tp[4] = {0, 0, 0, 0};
#pragma unroll
    for (int j = V ? 0 : 3; j < (V ? 11 : 7); j++) {
        int t[4];
        const uint32_t LDS_AS *r = (const uint32_t LDS_AS *)(w + sq_row(r0 + j));
        if (H) {
            const uint32_t R[6] = { r[0], r[1], r[2], r[3], r[4], r[5] };
            uint32_t D[6];
#pragma unroll
            for (int k = 0; k < 5; k++) D[k] = odd ? hi_lo(R[k + 1], R[k]) : R[k];
            D[5] = odd ? R[5] >> 16 : R[5];                          // (only its low half is a tap's sample)
            const uint32_t Q0 = hi_lo(D[1], D[0]), Q1 = hi_lo(D[2], D[1]), Q2 = hi_lo(D[3], D[2]), Q3 = hi_lo(D[4], D[3]), Q4 = hi_lo(D[5], D[4]);
            t[0] = dot2(ch[3], D[3], dot2(ch[2], D[2], dot2(ch[1], D[1], dot2z(ch[0], D[0]))));
            t[2] = dot2(ch[3], D[4], dot2(ch[2], D[3], dot2(ch[1], D[2], dot2z(ch[0], D[1]))));
            t[1] = dot2(ch[3], Q3, dot2(ch[2], Q2, dot2(ch[1], Q1, dot2z(ch[0], Q0))));
            t[3] = dot2(ch[3], Q4, dot2(ch[2], Q3, dot2(ch[1], Q2, dot2z(ch[0], Q1))));
#pragma unroll
            for (int c_ = 0; c_ < 4; c_++) t[c_] = clip3(rg.lo1, rg.hi1, t[c_] >> rg.sh1);
        } else {
            // samples 3..6 of the window: dwords 1..3 (shifted) when the window starts on an even sample, dwords 2..3 when it starts on an odd one
            const uint32_t d1 = r[1], d2 = r[2], d3 = r[3];
            const uint32_t a0 = odd ? d2 : hi_lo(d2, d1), a1 = odd ? d3 : hi_lo(d3, d2);
            t[0] = (int)(int16_t)(a0 & 0xFFFF); t[1] = (int)(int16_t)(a0 >> 16);
            t[2] = (int)(int16_t)(a1 & 0xFFFF); t[3] = (int)(int16_t)(a1 >> 16);
        }
        if (!V) {
#pragma unroll
            for (int c_ = 0; c_ < 4; c_++) acc[j - 3][c_] = t[c_];
            continue;
        }
        if (j > 0) {
#pragma unroll
            for (int c_ = 0; c_ < 4; c_++) {
                const uint32_t pr = pack2(tp[c_], t[c_]);
#pragma unroll
                for (int r_ = 0; r_ < 4; r_++) {
                    const int d = j - 1 - r_;
                    if (d == 0) acc[r_][c_] = dot2a(cv[0], pr, rg.off2);
                    else if (d > 0 && d <= 6 && (d & 1) == 0) acc[r_][c_] = dot2(cv[d >> 1], pr, acc[r_][c_]);
                }
            }
        }
#pragma unroll
        for (int c_ = 0; c_ < 4; c_++) tp[c_] = t[c_];
        if (j & 1) __builtin_amdgcn_sched_barrier(0);      // two rows at a time: the scheduler otherwise fetches all eleven rows first (66 registers: 160 VGPRs for the kernel)
    }
#pragma unroll
    for (int r_ = 0; r_ < 4; r_++) {
        int v[4];
#pragma unroll
        for (int c_ = 0; c_ < 4; c_++) v[c_] = clip3(0, maxv, V ? acc[r_][c_] >> rg.sh2 : acc[r_][c_]);
        o[r_ * 2 + 0] = pack2(v[0], v[1]);
        o[r_ * 2 + 1] = pack2(v[2], v[3]);
    }
}
template <bool H, bool V>
__device__ __forceinline__ void mc_chroma_2x2_w(const char LDS_AS *w, int r0, int odd, const uint32_t ch[2], const uint32_t cv[2], Regime rg, int maxv, uint32_t o[2])
{
    int acc[2][2];
    int tp[2] = {0, 0};
#pragma unroll
    for (int j = V ? 0 : 1; j < (V ? 5 : 3); j++) {
        int t[2];
        const uint32_t LDS_AS *r = (const uint32_t LDS_AS *)(w + sq_row(r0 + j));
        if (H) {
            const uint32_t R0 = r[0], R1 = r[1], R2 = r[2];
            const uint32_t D0 = odd ? hi_lo(R1, R0) : R0, D1 = odd ? hi_lo(R2, R1) : R1, D2 = odd ? R2 >> 16 : R2;      // (D2: only its low half is a tap's sample)
            const uint32_t Q0 = hi_lo(D1, D0), Q1 = hi_lo(D2, D1);
            t[0] = dot2(ch[1], D1, dot2z(ch[0], D0));
            t[1] = dot2(ch[1], Q1, dot2z(ch[0], Q0));
#pragma unroll
            for (int c_ = 0; c_ < 2; c_++) t[c_] = clip3(rg.lo1, rg.hi1, t[c_] >> rg.sh1);
        } else {
            const uint32_t R0 = r[0], R1 = r[1];
            const uint32_t a0 = odd ? R1 : hi_lo(R1, R0);          // samples 1..2 of the window
            t[0] = (int)(int16_t)(a0 & 0xFFFF); t[1] = (int)(int16_t)(a0 >> 16);
        }
        if (!V) { acc[j - 1][0] = t[0]; acc[j - 1][1] = t[1]; continue; }
        if (j > 0) {
#pragma unroll
            for (int c_ = 0; c_ < 2; c_++) {
                const uint32_t pr = pack2(tp[c_], t[c_]);
#pragma unroll
                for (int r_ = 0; r_ < 2; r_++) {
                    const int d = j - 1 - r_;
                    if (d == 0) acc[r_][c_] = dot2a(cv[0], pr, rg.off2);
                    else if (d == 2) acc[r_][c_] = dot2(cv[1], pr, acc[r_][c_]);
                }
            }
        }
        tp[0] = t[0]; tp[1] = t[1];
    }
#pragma unroll
    for (int r_ = 0; r_ < 2; r_++)
        o[r_] = pack2(clip3(0, maxv, V ? acc[r_][0] >> rg.sh2 : acc[r_][0]), clip3(0, maxv, V ? acc[r_][1] >> rg.sh2 : acc[r_][1]));
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
// Q16 (MODE 0): the wave holds four 16x16 blocks inside one CU each - 16 consecutive lanes share a window (k_inter_quad) - else every lane fetches its own (k_inter_small)
template <int MODE, bool Q16 = false>
__device__ __forceinline__ bool inter_tile(const InterArgs &a, uint4 r0, uint4 r1, bool lane_ok, int sx, int sy, int lane, int16_t *W, const LaneMap fm,
                                           const uint4 (*s_ref)[2], const uint4 *s_ltap, const uint2 *s_ctap, uint32_t pl[8], uint32_t pu[2], uint32_t pv[2],
                                           const RegionMap *rm = nullptr, int wave = 0, uint32_t own = 0)
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
        int16_t *const I = W + L_SAMPLES + 2 * C_SAMPLES + (REGION ? wave * REG_I_SAMPLES : 0);
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
            if (REGION) {
#pragma unroll
                for (int it = 0; it < 3; it++)       // 71 rows x 10 chunks, thread t takes chunks t, t + 256, t + 512: LDS offset = chunk * 16
                    if (rm->y[it] >= 0) __builtin_amdgcn_global_load_lds((const GAS void *)(p + (rm->y[it] >> 8) * a.s_l + 8 * (rm->y[it] & 127)), (LDS_AS void *)(d + (256 * it + 64 * wave) * 16), 16, 0, 0);
            } else {
#pragma unroll
                for (int it = 0; it < 4; it++)       // 39 rows x 6 chunks, 10 rows per request (lanes 60..63 idle)
                    if (lane < (it < 3 ? 60 : 54)) __builtin_amdgcn_global_load_lds((const GAS void *)(p + fm.gy + 10 * it * a.s_l), (LDS_AS void *)(d + it * 960), 16, 0, 0);
            }
        };
        auto request_chroma = [&](int l) {
            gs16 ry_, ru_, rv_; int px, py;
            list_refs(l, ry_, ru_, rv_); list_pos(l, px, py);
            const int off = ((py >> 3) - 1) * a.s_c + (((px >> 3) - 1) & ~7);
            char LDS_AS *const d = (char LDS_AS *)(W + L_SAMPLES);
            if (REGION) {
#pragma unroll
                for (int it = 0; it < 2; it++)       // 2 planes x 35 rows x 6 chunks
                    if (rm->c[it] >= 0) __builtin_amdgcn_global_load_lds((const GAS void *)(((rm->c[it] & 128) ? rv_ : ru_) + off + (rm->c[it] >> 8) * a.s_c + 8 * (rm->c[it] & 127)), (LDS_AS void *)(d + (256 * it + 64 * wave) * 16), 16, 0, 0);
            } else {
#pragma unroll
                for (int it = 0; it < 3; it++) {     // 2 planes x 19 rows x 4 chunks
                    const int idx = lane + 64 * it, plane = idx >= 76, j = idx - 76 * plane;
                    if (idx < 152) __builtin_amdgcn_global_load_lds((const GAS void *)((plane ? rv_ : ru_) + off + (j >> 2) * a.s_c + 8 * (j & 3)), (LDS_AS void *)(d + it * 1024), 16, 0, 0);
                }
            }
        };
        constexpr int N_LUMA_REQ = REGION ? 3 : 4, N_CHROMA_REQ = REGION ? 2 : 3;
        const int l_first = use[0] ? 0 : 1, n_lists = (int)use[0] + (int)use[1];
        load_resid();
        if (n_lists) { request_luma(l_first); __builtin_amdgcn_sched_barrier(0); request_chroma(l_first); }
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
            {
                const uint4 th = s_ltap[ldx ? ((px & 3) << 2) : 16], tv = s_ltap[ldy ? ((py & 3) << 2) : 16];
                const uint32_t ch[4] = { th.x, th.y, th.z, th.w }, cv[4] = { tv.x, tv.y, tv.z, tv.w };
                const Regime rg = regime(ldx, ldy, a.bd_l);
                const int mis = ((px >> 2) - 3) & 7, odd = mis & 1;
                const int16_t *const Wy = Wy0 + (mis & ~1);
                if (ldx) { if (ldy) luma_tile_filter<true, true, WS_L, true>(Wy, ch, cv, rg, maxl, I, lane, o, odd); else luma_tile_filter<true, false, WS_L, true>(Wy, ch, cv, rg, maxl, I, lane, o, odd); }
                else     { if (ldy) luma_tile_filter<false, true, WS_L, true>(Wy, ch, cv, rg, maxl, I, lane, o, odd); else luma_tile_filter<false, false, WS_L, true>(Wy, ch, cv, rg, maxl, I, lane, o, odd); }
            }
            if (more) {                                             // the luma window block is free: the second list's goes there while the chroma passes run
                if (REGION) __syncthreads(); else wave_lds_sync();
                request_luma(1);
            }
            if (!REGION) {
                if (more) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(N_LUMA_REQ) : "memory"); else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                wave_lds_sync();
            }
            {
                const uint2 th = s_ctap[cdx ? ((px & 7) << 2) : 32], tv = s_ctap[cdy ? ((py & 7) << 2) : 32];
                const uint32_t c2h[2] = { th.x, th.y }, c2v[2] = { tv.x, tv.y };
                const Regime rg = regime(cdx, cdy, a.bd_c);
                const int mis = ((px >> 3) - 1) & 7, odd = mis & 1;
                const int16_t *const Wu = Wu0 + (mis & ~1), *const Wv = Wu + C_SAMPLES;
                if (cdx) { if (cdy) chroma_tile_filter<true, true, WS_C, true>(Wu, Wv, c2h, c2v, rg, maxc, I, lane, ou, ov, odd); else chroma_tile_filter<true, false, WS_C, true>(Wu, Wv, c2h, c2v, rg, maxc, I, lane, ou, ov, odd); }
                else     { if (cdy) chroma_tile_filter<false, true, WS_C, true>(Wu, Wv, c2h, c2v, rg, maxc, I, lane, ou, ov, odd); else chroma_tile_filter<false, false, WS_C, true>(Wu, Wv, c2h, c2v, rg, maxc, I, lane, ou, ov, odd); }
            }
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
        load_resid();
        // quadrant-major lanes (k_inter_split): q = lane >> 4 is the SCU's 16x16 quadrant of the tile, (jx, jy) its place in the quadrant
        const int jx = lane & 3, jy = (lane >> 2) & 3;
        char LDS_AS *const Wb = (char LDS_AS *)W;                          // the wave's window block (SQ_BYTES)
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
            const uint4 lth = s_ltap[ldx ? ((px & 3) << 2) : 16], ltv = s_ltap[ldy ? ((py & 3) << 2) : 16];
            ch[0] = lth.x; ch[1] = lth.y; ch[2] = lth.z; ch[3] = lth.w; cv[0] = ltv.x; cv[1] = ltv.y; cv[2] = ltv.z; cv[3] = ltv.w;
            // chroma: 1/8-pel position (x<<2)+mv in luma quarter-pel == chroma eighth-pel; phase in 1/32 = (pos&7)<<2
            const uint2 cth = s_ctap[cdx ? ((px & 7) << 2) : 32], ctv = s_ctap[cdy ? ((py & 7) << 2) : 32];
            uint32_t c2h[2] = { cth.x, cth.y }, c2v[2] = { ctv.x, ctv.y };
            const Regime rgl = regime(ldx, ldy, a.bd_l), rgc = regime(cdx, cdy, a.bd_c);
            // ---- quadrants inside ONE CU (a 16x16 CU, or a 16x16 part of a 32x16 / 16x64 ... one): the 16 lanes want overlapping 11x11 windows of one 23x23 window
            //      (+ two 11x11 chroma windows).  They fetch it ONCE, straight into the wave's LDS block (global_load_lds, 16 bytes per lane and instruction: no staging
            //      registers, all nine requests of the list in flight together), 6 + 3 requests per lane instead of 22 + 20; the other lanes' per-lane loads and
            //      filtering run while those are in flight, then every quadrant lane filters its part of the shared window - the per-lane predictors' arithmetic,
            //      sample for sample (mc_luma_4x4_w / mc_chroma_2x2_w).  All 16 lanes of such a quadrant are here together: they share the CU, hence every test above.
            constexpr bool any_q = Q16;
            int misl = 0, oddc = 0;
            if (any_q) {
                if (Q16) {
                    const int pxq = px - (jx << 4), pyq = py - (jy << 4);          // the quadrant's first sample, quarter samples
                    const int rs = lane >> 2 & 3;                                  // this lane's row slot / column chunk inside a request
                    {
                        const int xi = (pxq >> 2) - 3, col = lane & 3;
                        misl = xi & 7;
                        const gs16 src = ry_ + ((pyq >> 2) - 3 + rs) * a.s_l + (xi & ~7) + col * 8;
#pragma unroll
                        for (int k = 0; k < 6; k++)
                            if (k < 5 || rs < 3) __builtin_amdgcn_global_load_lds((const GAS void *)(src + 4 * k * a.s_l), (LDS_AS void *)(Wb + k * SQ_BLK_B), 16, 0, 0);
                    }
                    {
                        const int xc = (pxq >> 3) - 1, plane = lane >> 1 & 1, col = lane & 1;
                        oddc = xc & 1;
                        const gs16 src = (plane ? rv_ : ru_) + ((pyq >> 3) - 1 + rs) * a.s_c + (xc & ~1) + col * 8;
#pragma unroll
                        for (int k = 0; k < 3; k++)
                            if (k < 2 || rs < 3) __builtin_amdgcn_global_load_lds((const GAS void *)(src + 4 * k * a.s_c), (LDS_AS void *)(Wb + SQ_L_BYTES + k * SQ_BLK_B), 16, 0, 0);
                    }
                }
            }
            if (!Q16) {
                {
                    const gs16 p = ry_ + ((py >> 2) - 3) * a.s_l + (px >> 2) - 3;
                    const bool wh = __ballot(ldx) != 0, wvv = __ballot(ldy) != 0;      // over the lanes that run this list per lane
                    if (wh) { if (wvv) mc_luma_4x4<true, true>(p, a.s_l, ch, cv, rgl, maxl, o); else mc_luma_4x4<true, false>(p, a.s_l, ch, cv, rgl, maxl, o); }
                    else    { if (wvv) mc_luma_4x4<false, true>(p, a.s_l, ch, cv, rgl, maxl, o); else mc_luma_4x4<false, false>(p, a.s_l, ch, cv, rgl, maxl, o); }
                }
                {
                    const int off = ((py >> 3) - 1) * a.s_c + (px >> 3) - 1;
                    const bool wh = __ballot(cdx) != 0, wvv = __ballot(cdy) != 0;
#define MC_C(H, V) do { mc_chroma_2x2<H, V>(ru_ + off, a.s_c, c2h, c2v, rgc, maxc, ou); mc_chroma_2x2<H, V>(rv_ + off, a.s_c, c2h, c2v, rgc, maxc, ov); } while (0)
                    if (wh) { if (wvv) MC_C(true, true); else MC_C(true, false); }
                    else    { if (wvv) MC_C(false, true); else MC_C(false, false); }
#undef MC_C
                }
            }
            if (any_q) {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");               // the window requests have landed (nothing else orders an LDS read behind them)
                wave_lds_sync();
                if (Q16) {
                    const char LDS_AS *const wq = Wb + (lane >> 4) * 256;
                    {
                        const char LDS_AS *const w = wq + (((misl >> 1) + (jx << 1)) << 2);      // the dword of the row that holds the lane's first window sample
                        const int oddl = misl & 1;
                        const bool wh = __ballot(ldx) != 0, wvv = __ballot(ldy) != 0;
                        if (wh) { if (wvv) mc_luma_4x4_w<true, true>(w, jy << 2, oddl, ch, cv, rgl, maxl, o); else mc_luma_4x4_w<true, false>(w, jy << 2, oddl, ch, cv, rgl, maxl, o); }
                        else    { if (wvv) mc_luma_4x4_w<false, true>(w, jy << 2, oddl, ch, cv, rgl, maxl, o); else mc_luma_4x4_w<false, false>(w, jy << 2, oddl, ch, cv, rgl, maxl, o); }
                    }
                    {
                        const char LDS_AS *const w = wq + SQ_L_BYTES + (jx << 2);
                        const bool wh = __ballot(cdx) != 0, wvv = __ballot(cdy) != 0;
#define MC_CW(H, V) do { mc_chroma_2x2_w<H, V>(w, jy << 1, oddc, c2h, c2v, rgc, maxc, ou); mc_chroma_2x2_w<H, V>(w + 32, jy << 1, oddc, c2h, c2v, rgc, maxc, ov); } while (0)
                        if (wh) { if (wvv) MC_CW(true, true); else MC_CW(true, false); }
                        else    { if (wvv) MC_CW(false, true); else MC_CW(false, false); }
#undef MC_CW
                    }
                }
                wave_lds_sync();                                               // the block is written again by the second list's requests
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
// Four launches per picture, one per CLASS of block, each with its own register budget (rounds 2 - 4: as paths of one kernel every path paid the registers of the
// others - 128 -> 143 -> 156 VGPRs, three waves per SIMD - and a wave that held two kinds of block ran both instruction streams).  xgpu_batch_create sorts the picture
// into four work lists in ONE spatial order (vertical strips XGPU_INTER_STRIP regions wide, row by row inside a strip):
//   k_inter_region  64x64 regions inside one CU: one workgroup per region, the reference windows requested once per workgroup into LDS the four waves share;
//   k_inter_tile    32x32 tiles inside one CU (whose region is not): one wave per tile, the window in the wave's own LDS;
//   k_inter_quad    16x16 blocks inside one CU (whose tile is not): four per wave, 16 lanes share a block's window;
//   k_inter_small   every other 16x16 block: four per wave, one lane per SCU, every lane on the CU that covers its SCU (owner map) with windows of its own.
// The classes are disjoint sets of whole 16x16 blocks, so the launches write disjoint 32-byte row segments and disjoint SCU-map records: the second to fourth are
// launched without the barrier bit and overlap the tail of the one before (launch_inter).  The first three know their CU from the list item, which carries the CU's
// record: the chain of a wave is list item -> reference windows; the reference table (kernel arguments) and the tap tables (constant memory) are read with wave-uniform
// addresses in the first two - no owner-map link, no LDS tables, no barrier in front of the work.
// XCD-aware mapping (all four): workgroup b runs on XCD b % 8 and every XCD has its own L2; XCD k takes the k-th contiguous eighth of the list - a compact patch of
// the picture whose vertical halos are still in its L2 when the row below is processed.
// History of the single-kernel form (rounds 1-4: persistent waves, class-sorted pieces, occupancy sweeps, non-temporal hints, region order) is in DESIGN.md 3.
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

__global__ __launch_bounds__(256) void k_inter_region(const InterArgs a)
{
    __shared__ __attribute__((aligned(16))) int16_t s_win[REG_SAMPLES];      // the region's shared windows + the four waves' intermediates
    const int idx = xcd_slice(blockIdx.x, gridDim.x);
    if (idx >= a.n_regions) return;
    const uint4 *const item = (const uint4 *)&a.regions[idx];
    const uint4 e = item[0], c0 = item[1], c1 = item[2];
    const int rx = e.x & 0xFFFF, ry = e.x >> 16;
    const int t = threadIdx.x, lane = t & 63;
    const int sx = (rx << 4) + ((t >> 6 & 1) << 3) + (lane & 7), sy = (ry << 4) + ((t >> 7) << 3) + (lane >> 3);
    const RegionMap rmap = region_map(t);
    const LaneMap fm = { 0, 0 };
    uint32_t pl[8], pu[2], pv[2];
    if (inter_tile<2>(a, c0, c1, true, sx, sy, lane, s_win, fm, ARG_REFS(a), ARG_LTAPS(a), ARG_CTAPS(a), pl, pu, pv, &rmap, t >> 6, e.y)) store_scu(a, sx, sy, pl, pu, pv);
}

__global__ __launch_bounds__(256) void k_inter_tile(const InterArgs a)
{
    __shared__ __attribute__((aligned(16))) int16_t s_win[4 * UNI_SAMPLES];  // per wave: window + intermediate
    const int t = threadIdx.x, lane = t & 63;
    const int idx = xcd_slice(blockIdx.x, gridDim.x) * 4 + (t >> 6);
    if (idx >= a.n_tiles) return;                                            // (no barrier in this kernel)
    const uint4 *const item = (const uint4 *)&a.tiles[idx];
    const uint4 e = item[0], c0 = item[1], c1 = item[2];
    const int sx = ((e.x & 0xFFFF) << 3) + (lane & 7), sy = ((e.x >> 16) << 3) + (lane >> 3);
    LaneMap fm;
    {
        const int row0 = (lane * 171) >> 10, k = lane - row0 * 6;            // luma window: 10 rows x 6 aligned chunks of 8 samples per request (lanes 60..63 idle)
        fm.gy = row0 * a.s_l + 8 * k; fm.ly = row0 * UW_STRIDE + 8 * k;
    }
    uint32_t pl[8], pu[2], pv[2];
    if (inter_tile<1>(a, c0, c1, true, sx, sy, lane, s_win + (t >> 6) * UNI_SAMPLES, fm, ARG_REFS(a), ARG_LTAPS(a), ARG_CTAPS(a), pl, pu, pv, nullptr, 0, e.y)) store_scu(a, sx, sy, pl, pu, pv);
}

// the reference table and the tap tables in LDS, for lanes that look them up with indices of their own; one load per lane, all before the first wait
struct SplitTables { uint4 ref[XGPU_MAX_REFS * 2][2]; uint4 ltap[17]; uint2 ctap[33]; };
__device__ __forceinline__ uint4 tables_load(const InterArgs &a, int t)
{
    static_assert(XGPU_MAX_REFS * 4 <= 96, "one 16-byte half of a reference entry per lane, in front of the tap tables' lanes");
    uint4 tab = make_uint4(0, 0, 0, 0);
    if (t < XGPU_MAX_REFS * 4) tab = ((const uint4 *)&a.refp[0][0])[t];
    else if (t >= 96 && t < 96 + 17) tab = *(const uint4 *)k_luma_taps[a.admvp][t - 96];
    else if (t >= 128 && t < 128 + 33) { const uint2 v = *(const uint2 *)k_chroma_taps[a.admvp][t - 128]; tab.x = v.x; tab.y = v.y; }
    return tab;
}
__device__ __forceinline__ void tables_store(SplitTables &s, int t, uint4 tab)
{
    if (t < XGPU_MAX_REFS * 4) s.ref[t >> 1][t & 1] = tab;
    else if (t >= 96 && t < 96 + 17) s.ltap[t - 96] = tab;
    else if (t >= 128 && t < 128 + 33) s.ctap[t - 128] = make_uint2(tab.x, tab.y);
}

// four 16x16 blocks per wave, each inside one CU: lanes 16 q .. 16 q + 15 are block q's 4 x 4 SCUs and share its windows (inter_tile<0, true>)
__global__ __launch_bounds__(256) void k_inter_quad(const InterArgs a)
{
    __shared__ SplitTables s_tab;
    __shared__ __attribute__((aligned(16))) char s_win[4 * SQ_BYTES];      // per wave: the shared windows of its four blocks
    const int t = threadIdx.x, lane = t & 63;
    const int idx = (xcd_slice(blockIdx.x, gridDim.x) * 4 + (t >> 6)) * 4 + (lane >> 4);
    const bool have = idx < a.n_quads;
    uint4 e = make_uint4(0, 0, 0, 0), c0 = e, c1 = e;
    if (have) { const uint4 *const item = (const uint4 *)&a.quads[idx]; e = item[0]; c0 = item[1]; c1 = item[2]; }
    const uint4 tab = tables_load(a, t);
    const int sx = ((e.x & 0xFFFF) << 2) + (lane & 3), sy = ((e.x >> 16) << 2) + (lane >> 2 & 3);
    tables_store(s_tab, t, tab);
    __syncthreads();
    const LaneMap fm = { 0, 0 };
    uint32_t pl[8], pu[2], pv[2];
    if (inter_tile<0, true>(a, c0, c1, have, sx, sy, lane, (int16_t *)(s_win + (t >> 6) * SQ_BYTES), fm, s_tab.ref, s_tab.ltap, s_tab.ctap, pl, pu, pv, nullptr, 0, e.y)) store_scu(a, sx, sy, pl, pu, pv);
}

// four 16x16 blocks per wave, one lane per SCU, every lane on the CU that covers its SCU: owner entry -> CU record -> windows of its own (inter_tile<0, false>)
__global__ __launch_bounds__(256) void k_inter_small(const InterArgs a)
{
    __shared__ SplitTables s_tab;
    const int t = threadIdx.x, lane = t & 63;
    const int idx = (xcd_slice(blockIdx.x, gridDim.x) * 4 + (t >> 6)) * 4 + (lane >> 4);
    const bool have = idx < a.n_smalls;
    const uint32_t e = have ? a.smalls[idx] : 0u;
    const int sx = ((e & 0xFFFF) << 2) + (lane & 3), sy = ((e >> 16) << 2) + (lane >> 2 & 3);
    const bool active = have && sx < (a.pic_w >> 2) && sy < (a.pic_h >> 2);
    // the first link of the chain goes out before the tables are staged
    const uint32_t own = active ? a.owner[sy * a.w_scu + sx] : OWNER_NONE;
    const uint4 tab = tables_load(a, t);
    uint4 c0 = make_uint4(0, 0, 0, 0), c1 = c0;
    const bool ok = own < (uint32_t)a.n_cu;                    // unowned (another batch's SCU) or not an index of this batch
    if (ok) { c0 = ((const uint4 *)&a.cus[own])[0]; c1 = ((const uint4 *)&a.cus[own])[1]; }
    tables_store(s_tab, t, tab);
    __syncthreads();
    const LaneMap fm = { 0, 0 };
    uint32_t pl[8], pu[2], pv[2];
    if (inter_tile<0, false>(a, c0, c1, ok, sx, sy, lane, nullptr, fm, s_tab.ref, s_tab.ltap, s_tab.ctap, pl, pu, pv, nullptr, 0, own)) store_scu(a, sx, sy, pl, pu, pv);
}

void launch_inter(xgpu_ctx *c, const InterArgs &a, bool any_order)
{
    auto grid = [](int n, int per) { return dim3((unsigned)((((n + per - 1) / per + 7) >> 3) << 3)); };
    // the single-SCU blocks first: their per-lane chains run longest.  any_order: the later launches without the barrier bit (hipExtAnyOrderLaunch)
    bool first = true;
    auto go = [&](auto kernel, dim3 g) {
        if (any_order && !first) hipExtLaunchKernelGGL(kernel, g, dim3(256), 0, c->stream, nullptr, nullptr, hipExtAnyOrderLaunch, a);
        else hipLaunchKernelGGL(kernel, g, dim3(256), 0, c->stream, a);
        first = false;
    };
    if (a.n_smalls) go(k_inter_small, grid(a.n_smalls, 16));
    if (a.n_regions) go(k_inter_region, grid(a.n_regions, 1));
    if (a.n_tiles) go(k_inter_tile, grid(a.n_tiles, 4));
    if (a.n_quads) go(k_inter_quad, grid(a.n_quads, 16));
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

