// affine_model.h - the affine motion model's scalar arithmetic, shared by the host batch builder (which sorts the tiles of affine CUs into the
// EIF and the sub-block-translation work lists) and k_affine.hip (which re-derives it per tile): deltas, sub-block size, EIF decision.
#pragma once
#ifdef __HIPCC__
#include <hip/hip_runtime.h>
#else                    // the host front end (plain C++) derives the sub-block vectors of affine CUs for its motion maps with the same arithmetic
#define __host__
#define __device__
#endif
#include <stdint.h>

static __host__ __device__ inline int aff_imax(int a, int b) { return a > b ? a : b; }
static __host__ __device__ inline int aff_imin(int a, int b) { return a < b ? a : b; }
static __host__ __device__ inline int aff_iabs(int a) { return a < 0 ? -a : a; }

#define AFF_BIT 7                    // MAX_CU_LOG2: the model keeps 2 + 7 fractional bits

struct AffModel { int dh[2], dv[2]; };

static __host__ __device__ inline int aff_round(int v, int shift) { return (v + (1 << (shift - 1)) - (v >= 0)) >> shift; }     // xevdm_mv_rounding_s32
static __host__ __device__ inline int aff_clip18(int v) { return v < -(1 << 17) ? -(1 << 17) : (v > (1 << 17) - 1 ? (1 << 17) - 1 : v); }

static __host__ __device__ inline AffModel aff_model(const int16_t *mv, int lw, int lh, int vn)          // mv[vertex][x/y]
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
static __host__ __device__ inline bool aff_eif_applicable(const AffModel &m, bool &mem_band)
{
    const int P = 2 + AFF_BIT, one = 1 << P;
    const int x1 = 5 * (m.dh[0] + one), x2 = 5 * m.dv[0], x3 = x1 + x2;
    const int y1 = 5 * m.dh[1], y2 = 5 * (m.dv[1] + one), y3 = y1 + y2;
    const int mx = aff_imax(aff_imax(0, x1), aff_imax(x2, x3)), nx = aff_imin(aff_imin(0, x1), aff_imin(x2, x3));
    const int my = aff_imax(aff_imax(0, y1), aff_imax(y2, y3)), ny = aff_imin(aff_imin(0, y1), aff_imin(y2, y3));
    mem_band = (((mx - nx + one - 1) >> P) + 2) * (((my - ny + one - 1) >> P) + 2) <= 72;
    if (m.dv[1] < -one) return false;
    return (aff_imax(0, m.dv[1]) + aff_iabs(m.dh[1])) * 5 <= one;
}

// xevdm_derive_affine_subblock_size_bi (xevdm_util.c:1870-1945)
static __host__ __device__ inline void aff_subblock(const AffModel m[2], const bool use[2], int lw, int lh, int &sub_w, int &sub_h, bool &mem_band)
{
    sub_w = 1 << lw; sub_h = 1 << lh;
    bool apply = true;
    mem_band = true;
    for (int l = 0; l < 2; l++) {
        if (!use[l]) continue;
        const int wx = aff_imax(aff_iabs(m[l].dh[0]), aff_iabs(m[l].dh[1])), wy = aff_imax(aff_iabs(m[l].dv[0]), aff_iabs(m[l].dv[1]));
        const int w = wx > 4 ? 4 : (wx == 0 ? 1 << lw : (wx == 1 ? 32 : (wx == 2 ? 16 : 8)));
        const int h = wy > 4 ? 4 : (wy == 0 ? 1 << lh : (wy == 1 ? 32 : (wy == 2 ? 16 : 8)));
        sub_w = aff_imin(sub_w, w); sub_h = aff_imin(sub_h, h);
    }
    for (int l = 0; l < 2; l++) {
        if (!use[l] || !apply) continue;             // the reference stops at the first list that fails
        bool mb;
        if (!aff_eif_applicable(m[l], mb)) apply = false;
        mem_band = mem_band && mb;
    }
    if (!apply) { sub_w = aff_imax(sub_w, 8); sub_h = aff_imax(sub_h, 8); }
}

