// mc_filters.h - the interpolation primitives shared by k_inter.hip and k_affine.hip: tap tables, rounding regimes, the per-lane 4x4 luma /
// 2x2 chroma separable filters, bi-prediction average and residual add on packed s16 pairs.
#pragma once
#include "xgpu_internal.h"

typedef short v2s __attribute__((ext_vector_type(2)));
struct __attribute__((packed, aligned(2))) U32x4u { uint32_t a, b, c, d; };
struct __attribute__((packed, aligned(2))) U32x2u { uint32_t a, b; };
struct __attribute__((packed, aligned(2))) U32x1u { uint32_t a; };
// Reference samples are read through GLOBAL pointers (address space 1): a pointer rebuilt from a 64-bit value (the reference table in LDS) is
// otherwise a generic one and every load a flat_load - which also counts on the LDS counter, so each LDS wait of the filter passes would wait
// for all the window and residual loads still in flight.
#define GAS __attribute__((address_space(1)))
typedef const GAS int16_t *gs16;
typedef uint32_t v4u32 __attribute__((ext_vector_type(4)));
typedef uint32_t v2u32 __attribute__((ext_vector_type(2)));
typedef uint32_t v3u32 __attribute__((ext_vector_type(3)));
typedef v4u32 __attribute__((aligned(2))) v4u32_u;          // 16 / 8 / 4 bytes at a 2-byte aligned sample address (gfx950 unaligned-access mode)
typedef v2u32 __attribute__((aligned(2))) v2u32_u;
typedef v3u32 __attribute__((aligned(2))) v3u32_u;
typedef uint32_t __attribute__((aligned(2))) u32_u;
__device__ __forceinline__ uint4 gload16(gs16 p) { const v4u32 v = *(const GAS v4u32_u *)p; return make_uint4(v.x, v.y, v.z, v.w); }
__device__ __forceinline__ uint2 gload8(gs16 p) { const v2u32 v = *(const GAS v2u32_u *)p; return make_uint2(v.x, v.y); }
__device__ __forceinline__ uint32_t gload4(gs16 p) { return *(const GAS u32_u *)p; }
// Loads a lane may sit out of (its result is multiplied by zero then): the value stays ONE vector register tuple from the load to the fence behind the requests,
// so that the place where the lane's branch joins the wave again needs no copy (a copy there waits for the load: one memory round trip per window row); a lane that
// sat out keeps whatever the registers held (`any`: a definition without an instruction).
template <typename T> __device__ __forceinline__ T any_value() { T v; asm volatile("" : "=v"(v)); return v; }
__device__ __forceinline__ void gload16_if(v4u32 &v, gs16 p, bool on) { if (on) v = *(const GAS v4u32_u *)p; }
__device__ __forceinline__ void gload8_if(v2u32 &v, gs16 p, bool on) { if (on) v = *(const GAS v2u32_u *)p; }
__device__ __forceinline__ void gload12_if(v3u32 &v, gs16 p, bool on) { if (on) v = *(const GAS v3u32_u *)p; }
__device__ __forceinline__ void gload4_if(uint32_t &v, gs16 p, bool on) { if (on) v = *(const GAS u32_u *)p; }

__device__ __forceinline__ int dot2(uint32_t a, uint32_t b, int c)
{
    return __builtin_amdgcn_sdot2(__builtin_bit_cast(v2s, a), __builtin_bit_cast(v2s, b), c, false);
}
__device__ __forceinline__ int dot2z(uint32_t a, uint32_t b)       // first tap pair of a chain: VOP3P form with a literal 0 addend
{                                                                  // (the builtin picks v_dot2c, which needs a v_mov 0 first)
    int r;
    asm("v_dot2_i32_i16 %0, %1, %2, 0" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ int dot2a(uint32_t a, uint32_t b, int c)   // VOP3P form, addend not tied to the destination
{
    int r;
    asm("v_dot2_i32_i16 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}
__device__ __forceinline__ uint32_t hi_lo(uint32_t hi, uint32_t lo)   // (lo.hi16, hi.lo16): samples 2m+1, 2m+2
{
    return __builtin_amdgcn_alignbit(hi, lo, 16);
}
__device__ __forceinline__ uint32_t pack2(int lo, int hi)               // two s32 -> packed (lo16, hi16), wraps
{
    return __builtin_amdgcn_perm((uint32_t)hi, (uint32_t)lo, 0x05040100u);
}
__device__ __forceinline__ int clip3(int lo, int hi, int v)
{
    int r;                                   // one v_med3_i32 instead of v_max + v_min (the kernel is VALU-issue bound)
    asm("v_med3_i32 %0, %1, %2, %3" : "=v"(r) : "v"(v), "v"(lo), "v"(hi));
    return r;
}

// Interpolation taps as packed s16 pairs.  Luma rows: phase (1/16 pel) -> 4 dwords; chroma: phase (1/32) -> 2 dwords.
// Baseline tables: src_base/xevd_mc.c:80-134 (phases 0,4,8,12 / 0,4,..,28); Main (sps_admvp_flag):
// src_main/xevdm_mc.c:121-175.  Entry [.][16] / [.][32] is the identity (tap 3 / tap 1 = 1) used when a
// direction is not filtered.
#define PK(a, b) ((uint32_t)(uint16_t)(int16_t)(a) | ((uint32_t)(uint16_t)(int16_t)(b) << 16))
#define L8(a, b, c, d, e, f, g, h) { PK(a, b), PK(c, d), PK(e, f), PK(g, h) }
#define C4(a, b, c, d) { PK(a, b), PK(c, d) }
static __constant__ uint32_t k_luma_taps[2][17][4] = {
    { L8(0,0,0,64,0,0,0,0), L8(0,0,0,0,0,0,0,0), L8(0,0,0,0,0,0,0,0), L8(0,0,0,0,0,0,0,0),
      L8(0,1,-5,52,20,-5,1,0), L8(0,0,0,0,0,0,0,0), L8(0,0,0,0,0,0,0,0), L8(0,0,0,0,0,0,0,0),
      L8(0,2,-10,40,40,-10,2,0), L8(0,0,0,0,0,0,0,0), L8(0,0,0,0,0,0,0,0), L8(0,0,0,0,0,0,0,0),
      L8(0,1,-5,20,52,-5,1,0), L8(0,0,0,0,0,0,0,0), L8(0,0,0,0,0,0,0,0), L8(0,0,0,0,0,0,0,0),
      L8(0,0,0,1,0,0,0,0) },
    { L8(0,0,0,64,0,0,0,0), L8(0,1,-3,63,4,-2,1,0), L8(-1,2,-5,62,8,-3,1,0), L8(-1,3,-8,60,13,-4,1,0),
      L8(-1,4,-10,58,17,-5,1,0), L8(-1,4,-11,52,26,-8,3,-1), L8(-1,3,-9,47,31,-10,4,-1), L8(-1,4,-11,45,34,-10,4,-1),
      L8(-1,4,-11,40,40,-11,4,-1), L8(-1,4,-10,34,45,-11,4,-1), L8(-1,4,-10,31,47,-9,3,-1), L8(-1,3,-8,26,52,-11,4,-1),
      L8(0,1,-5,17,58,-10,4,-1), L8(0,1,-4,13,60,-8,3,-1), L8(0,1,-3,8,62,-5,2,-1), L8(0,1,-2,4,63,-3,1,0),
      L8(0,0,0,1,0,0,0,0) },
};
static __constant__ uint32_t k_chroma_taps[2][33][2] = {
    { C4(0,64,0,0), C4(0,0,0,0), C4(0,0,0,0), C4(0,0,0,0), C4(-2,58,10,-2), C4(0,0,0,0), C4(0,0,0,0), C4(0,0,0,0),
      C4(-4,52,20,-4), C4(0,0,0,0), C4(0,0,0,0), C4(0,0,0,0), C4(-6,46,30,-6), C4(0,0,0,0), C4(0,0,0,0), C4(0,0,0,0),
      C4(-8,40,40,-8), C4(0,0,0,0), C4(0,0,0,0), C4(0,0,0,0), C4(-6,30,46,-6), C4(0,0,0,0), C4(0,0,0,0), C4(0,0,0,0),
      C4(-4,20,52,-4), C4(0,0,0,0), C4(0,0,0,0), C4(0,0,0,0), C4(-2,10,58,-2), C4(0,0,0,0), C4(0,0,0,0), C4(0,0,0,0),
      C4(0,1,0,0) },
    { C4(0,64,0,0), C4(-1,63,2,0), C4(-2,62,4,0), C4(-2,60,7,-1), C4(-2,58,10,-2), C4(-3,57,12,-2), C4(-4,56,14,-2), C4(-4,55,15,-2),
      C4(-4,54,16,-2), C4(-5,53,18,-2), C4(-6,52,20,-2), C4(-6,49,24,-3), C4(-6,46,28,-4), C4(-5,44,29,-4), C4(-4,42,30,-4), C4(-4,39,33,-4),
      C4(-4,36,36,-4), C4(-4,33,39,-4), C4(-4,30,42,-4), C4(-4,29,44,-5), C4(-4,28,46,-6), C4(-3,24,49,-6), C4(-2,20,52,-6), C4(-2,18,53,-5),
      C4(-2,16,54,-4), C4(-2,15,55,-4), C4(-2,14,56,-4), C4(-2,12,57,-3), C4(-2,10,58,-2), C4(-1,7,60,-2), C4(0,4,62,-2), C4(0,2,63,-1),
      C4(0,1,0,0) },
};

// The conditions of xevdm_mc's apply_DMVR that depend on the picture (src_main/xevdm_mc.c:1895-1911): the two references at equal POC distances
// on either side of the current picture (which also rules out the identical-motion case).  The static ones - merge mode, two references, at
// least 8x8 - are CuRec.dmvr.  k_inter (skips the CU's samples) and k_dmvr (predicts them) both call this.
__device__ __forceinline__ bool dmvr_applies(int poc_c, int poc0, int poc1)
{
    return (poc_c - poc0) * (poc_c - poc1) < 0 && abs(poc_c - poc0) == abs(poc_c - poc1);
}

// Per-lane description of one separable interpolation in the reference's four rounding regimes
// (xevd_mc.c:169-288 / :290-408, shifts xevd_mc.h:34-38):
//   stage 1: t = (sum_h) >> sh1, clipped to [0,max] only in the H-only regime, then truncated to s16
//   stage 2: out = clip((sum_v + off2) >> sh2)
struct Regime { int sh1, lo1, hi1, sh2, off2; };     // stage-1 clamp bounds: [0,max] in the H-only regime, else the whole s32 range
__device__ __forceinline__ Regime regime(int has_dx, int has_dy, int bd)
{
    Regime r;
    const int shift1 = min(4, bd - 8), shift2 = max(8, 20 - bd);
    r.sh1   = has_dx ? (has_dy ? shift1 : 6) : 0;
    r.lo1   = (has_dx && !has_dy) ? 0 : (int)0x80000000;
    r.hi1   = (has_dx && !has_dy) ? (1 << bd) - 1 : 0x7FFFFFFF;
    r.sh2   = has_dy ? (has_dx ? shift2 : 6) : 0;
    r.off2  = (has_dy && has_dx) ? (1 << (shift2 - 1)) : 0;
    return r;
}

// 4x4 luma prediction of one SCU.  `p` = reference sample at (block x - 3, block y - 3).
// out[r] = packed (c0,c1),(c2,c3) as two dwords per row -> o[r*2+0], o[r*2+1]; values are clipped s16.
// H / V say whether ANY lane of the wave filters in that direction (wave-uniform): without V only the block's own four rows are
// touched (a lane's identity vertical tap would select exactly row r+3 and add nothing), without H the row is the samples
// themselves.  The results are those of the full path with identity taps - the variants only skip work that cancels.
template <bool H, bool V>
__device__ __forceinline__ void mc_luma_4x4(gs16 p, int s, const uint32_t ch[4], const uint32_t cv[4],
                                            Regime rg, int maxv, uint32_t o[8])
{
    int acc[4][4];
    int tp[4] = {0, 0, 0, 0};
#pragma unroll
    for (int j = V ? 0 : 3; j < (V ? 11 : 7); j++) {
        int t[4];
        if (H) {
            const uint4 a = gload16(p + j * s);
            const uint2 b = gload8(p + j * s + 8);
            const uint32_t D0 = a.x, D1 = a.y, D2 = a.z, D3 = a.w, D4 = b.x, D5 = b.y;
            const uint32_t Q0 = hi_lo(D1, D0), Q1 = hi_lo(D2, D1), Q2 = hi_lo(D3, D2), Q3 = hi_lo(D4, D3), Q4 = hi_lo(D5, D4);
            t[0] = dot2(ch[3], D3, dot2(ch[2], D2, dot2(ch[1], D1, dot2z(ch[0], D0))));
            t[2] = dot2(ch[3], D4, dot2(ch[2], D3, dot2(ch[1], D2, dot2z(ch[0], D1))));
            t[1] = dot2(ch[3], Q3, dot2(ch[2], Q2, dot2(ch[1], Q1, dot2z(ch[0], Q0))));
            t[3] = dot2(ch[3], Q4, dot2(ch[2], Q3, dot2(ch[1], Q2, dot2z(ch[0], Q1))));
#pragma unroll
            for (int c = 0; c < 4; c++) t[c] = clip3(rg.lo1, rg.hi1, t[c] >> rg.sh1);
        } else {
            const uint2 a = gload8(p + j * s + 3);      // no lane filters horizontally: samples 3..6 of the window
            t[0] = (int)(int16_t)(a.x & 0xFFFF); t[1] = (int)(int16_t)(a.x >> 16);
            t[2] = (int)(int16_t)(a.y & 0xFFFF); t[3] = (int)(int16_t)(a.y >> 16);
        }
        if (!V) {
#pragma unroll
            for (int c = 0; c < 4; c++) acc[j - 3][c] = t[c];
            continue;
        }
        if (j > 0) {
            // row pair (j-1, j) feeds output row r with tap pair (j-1-r)/2 when j-1-r is even and in 0..6
#pragma unroll
            for (int c = 0; c < 4; c++) {
                const uint32_t pr = pack2(tp[c], t[c]);
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    const int d = j - 1 - r;
                    if (d == 0) acc[r][c] = dot2a(cv[0], pr, rg.off2);          // first tap pair carries the rounding offset
                    else if (d > 0 && d <= 6 && (d & 1) == 0) acc[r][c] = dot2(cv[d >> 1], pr, acc[r][c]);
                }
            }
        }
#pragma unroll
        for (int c = 0; c < 4; c++) tp[c] = t[c];
    }
#pragma unroll
    for (int r = 0; r < 4; r++) {
        int v[4];
#pragma unroll
        for (int c = 0; c < 4; c++) v[c] = clip3(0, maxv, V ? acc[r][c] >> rg.sh2 : acc[r][c]);
        o[r * 2 + 0] = pack2(v[0], v[1]);
        o[r * 2 + 1] = pack2(v[2], v[3]);
    }
}

// 2x2 chroma prediction of one SCU.  `p` = reference sample at (block x - 1, block y - 1).  o[r] = packed row r.
template <bool H, bool V>
__device__ __forceinline__ void mc_chroma_2x2(gs16 p, int s, const uint32_t ch[2], const uint32_t cv[2],
                                              Regime rg, int maxv, uint32_t o[2])
{
    int acc[2][2];
    int tp[2] = {0, 0};
#pragma unroll
    for (int j = V ? 0 : 1; j < (V ? 5 : 3); j++) {
        int t[2];
        if (H) {
            const v3u32 a = *(const GAS v3u32_u *)(p + j * s);      // six samples in ONE 12-byte request (rounds 1 - 4: 8 + 4 bytes)
            const uint32_t D0 = a.x, D1 = a.y, D2 = a.z;
            const uint32_t Q0 = hi_lo(D1, D0), Q1 = hi_lo(D2, D1);
            t[0] = dot2(ch[1], D1, dot2z(ch[0], D0));
            t[1] = dot2(ch[1], Q1, dot2z(ch[0], Q0));
#pragma unroll
            for (int c = 0; c < 2; c++) t[c] = clip3(rg.lo1, rg.hi1, t[c] >> rg.sh1);
        } else {
            const uint32_t a = gload4(p + j * s + 1);
            t[0] = (int)(int16_t)(a & 0xFFFF); t[1] = (int)(int16_t)(a >> 16);
        }
        if (!V) { acc[j - 1][0] = t[0]; acc[j - 1][1] = t[1]; continue; }
        if (j > 0) {
#pragma unroll
            for (int c = 0; c < 2; c++) {
                const uint32_t pr = pack2(tp[c], t[c]);
#pragma unroll
                for (int r = 0; r < 2; r++) {
                    const int d = j - 1 - r;
                    if (d == 0) acc[r][c] = dot2a(cv[0], pr, rg.off2);
                    else if (d == 2) acc[r][c] = dot2(cv[1], pr, acc[r][c]);
                }
            }
        }
        tp[0] = t[0]; tp[1] = t[1];
    }
#pragma unroll
    for (int r = 0; r < 2; r++)
        o[r] = pack2(clip3(0, maxv, V ? acc[r][0] >> rg.sh2 : acc[r][0]), clip3(0, maxv, V ? acc[r][1] >> rg.sh2 : acc[r][1]));
}

// (p0 + p1 + 1) >> 1 on packed non-negative s16 pairs (xevd_average_16b_no_clip, xevd_mc.c:145-167): packed 16-bit adds, the sums stay below 2^16
typedef unsigned short v2us __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t avg2(uint32_t a, uint32_t b)
{
    const v2us r = (__builtin_bit_cast(v2us, a) + __builtin_bit_cast(v2us, b) + (v2us)(1)) >> (v2us)(1);
    return __builtin_bit_cast(uint32_t, r);
}
// rec = clip(0, max, (s16)(res + pred)) on packed pairs: the 16-bit sum wraps (xevd_recon.c:39,60)
__device__ __forceinline__ uint32_t recon2(uint32_t pred, uint32_t res, int maxv)
{
    // (the sum as UNSIGNED halves: it has to wrap like the reference's (s16) cast, and signed vector overflow would be undefined - as recon2i, intra_pred.h)
    const v2s sum = __builtin_bit_cast(v2s, __builtin_bit_cast(v2us, pred) + __builtin_bit_cast(v2us, res));
    const v2s r = __builtin_elementwise_min(__builtin_elementwise_max(sum, (v2s)(0)), (v2s)((short)maxv));
    return __builtin_bit_cast(uint32_t, r);
}
