// k_mc.hip - motion compensation + residual add + clip + SCU-map update for every CU of a picture, from class-sorted work lists.
//
// Replaces, per picture, the reference's per-CU sequence  xevd_mc -> xevd_recon_yuv -> xevd_set_dec_info
// (src_base/xevd.c:725-754, xevd_mc.c:469-557, xevd_recon.c:35-92, xevd_util.c:1574-1660; Main tables xevdm_mc.c:121-175).
//
// MI355X mapping (not a translation of the per-CU C/AVX loops):
//   * the host batch builder cuts every CU into PIECES of at most 32x32 luma samples and sorts them by (band of CTU rows, reference lists used,
//     piece shape).  One wave = 64 lanes = 64 SCUs (4x4) = 64 / ((w/4)(h/4)) pieces of ONE shape and list set: identical control flow in every
//     lane whatever the partitioning of the picture is, no owner map, no per-region scan;
//   * the lanes of a piece fetch its (w+7)x(h+7) reference window ONCE, with 16-byte loads at the 2-byte-aligned sample address, into the
//     wave's own LDS; the horizontal pass produces every intermediate value of the window once (shared by the lanes of the piece), the
//     vertical pass reads 11 rows x 8 bytes per lane.  A 16x16 CU costs 69 window loads for its 16 lanes (the per-SCU formulation: 352),
//     an 8x8 CU 30 for 4 lanes.  LDS operations of one wave execute in order: no workgroup barrier anywhere, the four waves of a block are
//     independent work items;
//   * pieces 4 samples wide or high (two SCUs sharing nothing worth staging) filter per lane straight from L1/L2 (mc_luma_4x4);
//   * FIRs run on packed s16 pairs with v_dot2_i32_i16; the reference's four rounding regimes (copy / H / V / 2-D: variant from the UNCLIPPED
//     vector, phase from the clipped one) are per-lane tap vectors, shifts and clamps, and a per-wave ballot picks one of four code variants
//     that skip the passes no lane needs;
//   * blocks map to XCDs in contiguous runs of the (band-major) work list, so the pieces an XCD's L2 sees are spatial neighbours;
//   * no MFMA: 4/8-tap integer FIRs, bounded by instruction issue and HBM, not by dense contraction.
#include "xgpu_internal.h"

#include "mc_filters.h"

#define MC_LDS_SAMPLES 3840           // per wave: the largest window set - 16 pieces of 8x8: 16 x 15 rows x 16 samples

__device__ __forceinline__ void mc_wave_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// Geometry of the square piece classes (S = 8, 16, 32), all compile-time: L lanes per piece, 64 / L pieces per wave.  A luma window row holds
// CHUNKS 16-byte chunks (S + 7 samples rounded up to 8), a chroma window row the same number of 8-byte chunks (S/2 + 3 samples rounded up to 4).
template <int S> struct Geo {
    static constexpr int LW2 = S == 8 ? 1 : (S == 16 ? 2 : 3), LOGL = 2 * LW2, L = 1 << LOGL;
    static constexpr int ROWS = S + 7, CHUNKS = S == 8 ? 2 : (S == 16 ? 3 : 5), STRIDE = CHUNKS * 8;
    static constexpr int NT = ROWS * CHUNKS, NIT = (NT + L - 1) / L;                 // window fetch tasks per piece, iterations of its lanes
    static constexpr int NH = ROWS * (S / 4), NHIT = (NH + L - 1) / L;                // horizontal-pass tasks: rows x groups of 4 columns
    static constexpr int ROWS_C = S / 2 + 3, STRIDE_C = CHUNKS * 4;
    static constexpr int NTP = ROWS_C * CHUNKS, NTC = 2 * NTP, NITC = (NTC + L - 1) / L;      // both chroma planes
    static constexpr int GC = S / 8, NHC = 2 * ROWS_C * GC, NHCIT = (NHC + L - 1) / L;
    static constexpr int SLOT = NT * 8, SLOT_C = NTC * 4;                            // LDS samples per piece
};

// ---------------------------------------------------------------------------------------------------------
// Cooperative prediction of the pieces of one wave from one reference list.  q = lane index inside its piece, (qx, qy) its SCU there.
// base* = reference sample at (piece x - 3, piece y - 3) / (piece x/2 - 1, piece y/2 - 1) of the lane's piece.  Lanes past the last task of a
// pass repeat it (same bytes to the same place): straight-line code.  The intermediate rows are written over the window rows - every lane
// holds its results until all reads are done.
// ---------------------------------------------------------------------------------------------------------
template <int S> struct LumaFetch { uint4 v[Geo<S>::NIT]; };
template <int S> struct ChromaFetch { uint2 v[Geo<S>::NITC]; };

template <int S>
__device__ __forceinline__ void luma_fetch(gs16 base, int s, int q, LumaFetch<S> &f)
{
    typedef Geo<S> G;
#pragma unroll
    for (int k = 0; k < G::NIT; k++) {
        const int t = min(q + k * G::L, G::NT - 1);
        const int row = t / G::CHUNKS, c = t - row * G::CHUNKS;
        f.v[k] = gload16(base + row * s + 8 * c);
    }
}
template <int S>
__device__ __forceinline__ void chroma_fetch(gs16 bu, gs16 bv, int s, int q, ChromaFetch<S> &f)
{
    typedef Geo<S> G;
#pragma unroll
    for (int k = 0; k < G::NITC; k++) {
        const int t = min(q + k * G::L, G::NTC - 1);
        const int pl = t >= G::NTP, tt = t - pl * G::NTP, row = tt / G::CHUNKS, c = tt - row * G::CHUNKS;
        f.v[k] = gload8((pl ? bv : bu) + row * s + 4 * c);
    }
}

template <int S, bool H, bool V>
__device__ __forceinline__ void coop_luma(const LumaFetch<S> &f, int q, int qx, int qy, int16_t *Wp,
                                          const uint32_t ch[4], const uint32_t cv[4], Regime rg, int maxv, uint32_t o[8])
{
    typedef Geo<S> G;
#pragma unroll
    for (int k = 0; k < G::NIT; k++) {
        const int t = min(q + k * G::L, G::NT - 1);
        *(uint4 *)(Wp + t * 8) = f.v[k];                             // row-major with CHUNKS chunks per row: offset = task index
    }
    mc_wave_sync();
    uint2 r[G::NHIT];
#pragma unroll
    for (int k = 0; k < G::NHIT; k++) {
        const int t = min(q + k * G::L, G::NH - 1);
        const int row = t >> G::LW2, gq = t & (S / 4 - 1);
        const uint2 *w = (const uint2 *)(Wp + row * G::STRIDE + 4 * gq);
        const uint2 a = w[0], b = w[1], c = w[2];
        const uint32_t D0 = a.x, D1 = a.y, D2 = b.x, D3 = b.y, D4 = c.x, D5 = c.y;
        if (H) {
            const uint32_t Q0 = hi_lo(D1, D0), Q1 = hi_lo(D2, D1), Q2 = hi_lo(D3, D2), Q3 = hi_lo(D4, D3), Q4 = hi_lo(D5, D4);
            int tt[4];
            tt[0] = dot2(ch[3], D3, dot2(ch[2], D2, dot2(ch[1], D1, dot2z(ch[0], D0))));
            tt[2] = dot2(ch[3], D4, dot2(ch[2], D3, dot2(ch[1], D2, dot2z(ch[0], D1))));
            tt[1] = dot2(ch[3], Q3, dot2(ch[2], Q2, dot2(ch[1], Q1, dot2z(ch[0], Q0))));
            tt[3] = dot2(ch[3], Q4, dot2(ch[2], Q3, dot2(ch[1], Q2, dot2z(ch[0], Q1))));
#pragma unroll
            for (int m = 0; m < 4; m++) tt[m] = clip3(rg.lo1, rg.hi1, tt[m] >> rg.sh1);
            r[k] = make_uint2(pack2(tt[0], tt[1]), pack2(tt[2], tt[3]));
        } else {
            r[k] = make_uint2(hi_lo(D2, D1), hi_lo(D3, D2));         // samples 3..6 of the window
        }
    }
    mc_wave_sync();
#pragma unroll
    for (int k = 0; k < G::NHIT; k++) {
        const int t = min(q + k * G::L, G::NH - 1);
        const int row = t >> G::LW2, gq = t & (S / 4 - 1);
        *(uint2 *)(Wp + row * G::STRIDE + 4 * gq) = r[k];
    }
    mc_wave_sync();
    const int16_t *base = Wp + (qy << 2) * G::STRIDE + (qx << 2);
    if (!V) {
#pragma unroll
        for (int rr = 0; rr < 4; rr++) { const uint2 w = *(const uint2 *)(base + (rr + 3) * G::STRIDE); o[rr * 2] = w.x; o[rr * 2 + 1] = w.y; }
    } else {
        uint2 w[11];
#pragma unroll
        for (int j = 0; j < 11; j++) w[j] = *(const uint2 *)(base + j * G::STRIDE);
        int acc[4][4];
#pragma unroll
        for (int j = 1; j < 11; j++) {
            const uint32_t pr[4] = { __builtin_amdgcn_perm(w[j].x, w[j - 1].x, 0x05040100u), __builtin_amdgcn_perm(w[j].x, w[j - 1].x, 0x07060302u),
                                     __builtin_amdgcn_perm(w[j].y, w[j - 1].y, 0x05040100u), __builtin_amdgcn_perm(w[j].y, w[j - 1].y, 0x07060302u) };
#pragma unroll
            for (int c = 0; c < 4; c++)
#pragma unroll
                for (int rr = 0; rr < 4; rr++) {
                    const int d = j - 1 - rr;
                    if (d == 0) acc[rr][c] = dot2a(cv[0], pr[c], rg.off2);
                    else if (d > 0 && d <= 6 && (d & 1) == 0) acc[rr][c] = dot2(cv[d >> 1], pr[c], acc[rr][c]);
                }
        }
#pragma unroll
        for (int rr = 0; rr < 4; rr++) {
            int v[4];
#pragma unroll
            for (int c = 0; c < 4; c++) v[c] = clip3(0, maxv, acc[rr][c] >> rg.sh2);
            o[rr * 2 + 0] = pack2(v[0], v[1]);
            o[rr * 2 + 1] = pack2(v[2], v[3]);
        }
    }
    mc_wave_sync();
}

// Both chroma planes of the pieces.  Wp = the piece's slot: plane U rows, then plane V rows, CHUNKS 8-byte chunks per row.
template <int S, bool H, bool V>
__device__ __forceinline__ void coop_chroma(const ChromaFetch<S> &f, int q, int qx, int qy, int16_t *Wp,
                                            const uint32_t ch[2], const uint32_t cv[2], Regime rg, int maxv, uint32_t ou[2], uint32_t ov[2])
{
    typedef Geo<S> G;
#pragma unroll
    for (int k = 0; k < G::NITC; k++) {
        const int t = min(q + k * G::L, G::NTC - 1);
        *(uint2 *)(Wp + t * 4) = f.v[k];
    }
    mc_wave_sync();
    uint2 r[G::NHCIT];
#pragma unroll
    for (int k = 0; k < G::NHCIT; k++) {
        const int t = min(q + k * G::L, G::NHC - 1);
        const int prow = t / G::GC, gq = t - prow * G::GC;
        const uint2 *w = (const uint2 *)(Wp + prow * G::STRIDE_C + 4 * gq);
        const uint2 a = w[0], b = w[1];
        const uint32_t D0 = a.x, D1 = a.y, D2 = b.x, D3 = b.y;
        const uint32_t Q0 = hi_lo(D1, D0), Q1 = hi_lo(D2, D1);
        if (H) {
            const uint32_t Q2 = hi_lo(D3, D2);
            int tt[4];
            tt[0] = dot2(ch[1], D1, dot2z(ch[0], D0));
            tt[1] = dot2(ch[1], Q1, dot2z(ch[0], Q0));
            tt[2] = dot2(ch[1], D2, dot2z(ch[0], D1));
            tt[3] = dot2(ch[1], Q2, dot2z(ch[0], Q1));
#pragma unroll
            for (int m = 0; m < 4; m++) tt[m] = clip3(rg.lo1, rg.hi1, tt[m] >> rg.sh1);
            r[k] = make_uint2(pack2(tt[0], tt[1]), pack2(tt[2], tt[3]));
        } else {
            r[k] = make_uint2(Q0, Q1);                                // samples 1..4
        }
    }
    mc_wave_sync();
#pragma unroll
    for (int k = 0; k < G::NHCIT; k++) {
        const int t = min(q + k * G::L, G::NHC - 1);
        const int prow = t / G::GC, gq = t - prow * G::GC;
        *(uint2 *)(Wp + prow * G::STRIDE_C + 4 * gq) = r[k];
    }
    mc_wave_sync();
#pragma unroll
    for (int pl = 0; pl < 2; pl++) {
        const int16_t *base = Wp + (pl * G::ROWS_C + (qy << 1)) * G::STRIDE_C + (qx << 1);
        uint32_t *o = pl ? ov : ou;
        if (!V) {
            o[0] = *(const uint32_t *)(base + G::STRIDE_C); o[1] = *(const uint32_t *)(base + 2 * G::STRIDE_C);
        } else {
            uint32_t w[5];
#pragma unroll
            for (int j = 0; j < 5; j++) w[j] = *(const uint32_t *)(base + j * G::STRIDE_C);
            int acc[2][2];
#pragma unroll
            for (int j = 1; j < 5; j++) {
                const uint32_t pr[2] = { __builtin_amdgcn_perm(w[j], w[j - 1], 0x05040100u), __builtin_amdgcn_perm(w[j], w[j - 1], 0x07060302u) };
#pragma unroll
                for (int c = 0; c < 2; c++)
#pragma unroll
                    for (int rr = 0; rr < 2; rr++) {
                        const int d = j - 1 - rr;
                        if (d == 0) acc[rr][c] = dot2a(cv[0], pr[c], rg.off2);
                        else if (d == 2) acc[rr][c] = dot2(cv[1], pr[c], acc[rr][c]);
                    }
            }
#pragma unroll
            for (int rr = 0; rr < 2; rr++)
                o[rr] = pack2(clip3(0, maxv, acc[rr][0] >> rg.sh2), clip3(0, maxv, acc[rr][1] >> rg.sh2));
        }
    }
    mc_wave_sync();
}

// one list of a wave of S x S pieces, luma: window fetch, then the passes through the wave's LDS.  (gx, gy) = quarter-pel position of the piece origin
template <int S>
__device__ __forceinline__ void coop_list_luma(gs16 ry, int s_l, int gx, int gy, int mvx, int mvy, int bd, int q, int qx, int qy, int16_t *buf, int p,
                                               const uint4 *s_ltap, uint32_t o[8])
{
    typedef Geo<S> G;
    const int ldx = (mvx & 3) != 0, ldy = (mvy & 3) != 0;
    LumaFetch<S> fl;
    luma_fetch<S>(ry + ((gy >> 2) - 3) * s_l + (gx >> 2) - 3, s_l, q, fl);
    const uint4 th = s_ltap[ldx ? ((gx & 3) << 2) : 16], tv = s_ltap[ldy ? ((gy & 3) << 2) : 16];
    const uint32_t ch[4] = { th.x, th.y, th.z, th.w }, cv[4] = { tv.x, tv.y, tv.z, tv.w };
    const Regime rg = regime(ldx, ldy, bd);
    const bool wh = __ballot(ldx) != 0, wvv = __ballot(ldy) != 0;
    int16_t *Wp = buf + p * G::SLOT;
    const int maxv = (1 << bd) - 1;
    if (wh) { if (wvv) coop_luma<S, true, true>(fl, q, qx, qy, Wp, ch, cv, rg, maxv, o); else coop_luma<S, true, false>(fl, q, qx, qy, Wp, ch, cv, rg, maxv, o); }
    else    { if (wvv) coop_luma<S, false, true>(fl, q, qx, qy, Wp, ch, cv, rg, maxv, o); else coop_luma<S, false, false>(fl, q, qx, qy, Wp, ch, cv, rg, maxv, o); }
}
template <int S>
__device__ __forceinline__ void coop_list_chroma(gs16 ru, gs16 rv, int s_c, int gx, int gy, int mvx, int mvy, int bd, int q, int qx, int qy, int16_t *buf, int p,
                                                 const uint2 *s_ctap, uint32_t ou[2], uint32_t ov[2])
{
    typedef Geo<S> G;
    const int cdx = (mvx & 7) != 0, cdy = (mvy & 7) != 0;
    ChromaFetch<S> fc;
    const int coff = ((gy >> 3) - 1) * s_c + (gx >> 3) - 1;
    chroma_fetch<S>(ru + coff, rv + coff, s_c, q, fc);
    const uint2 th = s_ctap[cdx ? ((gx & 7) << 2) : 32], tv = s_ctap[cdy ? ((gy & 7) << 2) : 32];
    const uint32_t c2h[2] = { th.x, th.y }, c2v[2] = { tv.x, tv.y };
    const Regime rg = regime(cdx, cdy, bd);
    const bool wh = __ballot(cdx) != 0, wvv = __ballot(cdy) != 0;
    int16_t *Wp = buf + p * G::SLOT_C;
    const int maxv = (1 << bd) - 1;
    if (wh) { if (wvv) coop_chroma<S, true, true>(fc, q, qx, qy, Wp, c2h, c2v, rg, maxv, ou, ov); else coop_chroma<S, true, false>(fc, q, qx, qy, Wp, c2h, c2v, rg, maxv, ou, ov); }
    else    { if (wvv) coop_chroma<S, false, true>(fc, q, qx, qy, Wp, c2h, c2v, rg, maxv, ou, ov); else coop_chroma<S, false, false>(fc, q, qx, qy, Wp, c2h, c2v, rg, maxv, ou, ov); }
}

__global__ __launch_bounds__(256, 4) void k_mc(const InterArgs a)
{
    __shared__ uint4    s_ref[XGPU_MAX_REFS * 2][2];        // RefEntry [idx][list]
    __shared__ uint4    s_ltap[17];                         // luma taps of this sequence's table, [16] = identity
    __shared__ uint2    s_ctap[33];
    __shared__ __attribute__((aligned(16))) int16_t s_buf[4][MC_LDS_SAMPLES];      // per wave: the windows of its pieces

    const int t = threadIdx.x;
    if (t < XGPU_MAX_REFS * 2) {
        const uint4 *re = (const uint4 *)&a.refp[t >> 1][t & 1];
        s_ref[t][0] = re[0]; s_ref[t][1] = re[1];
    } else if (t >= 64 && t < 64 + 17) {
        s_ltap[t - 64] = *(const uint4 *)k_luma_taps[a.admvp][t - 64];
    } else if (t >= 128 && t < 128 + 33) {
        s_ctap[t - 128] = *(const uint2 *)k_chroma_taps[a.admvp][t - 128];
    }
    __syncthreads();

    // XCD-aware mapping: workgroup b runs on XCD b % 8; every XCD takes a contiguous run of the work list (band-major: spatial neighbours)
    const int per = gridDim.x >> 3;
    const int wi = __builtin_amdgcn_readfirstlane(((blockIdx.x & 7) * per + (blockIdx.x >> 3)) * 4 + (t >> 6));
    if (wi >= a.n_waves) return;
    const uint2 wd = *(const uint2 *)&a.waves[wi];
    const int first = __builtin_amdgcn_readfirstlane((int)wd.x), wpk = __builtin_amdgcn_readfirstlane((int)wd.y);
    const int n = wpk & 0xFF, lists = wpk >> 24;
    const int lw2 = ((wpk >> 8) & 0xFF) - 2, lh2 = ((wpk >> 16) & 0xFF) - 2, logL = lw2 + lh2;      // piece size in SCUs (log2), lanes per piece (log2)

    const int lane = t & 63;
    const int p = lane >> logL, q = lane & ((1 << logL) - 1);
    const bool valid = p < n;                                        // lanes past the last piece of a class redo that piece without storing
    const uint2 it = *(const uint2 *)&a.items[first + (valid ? p : n - 1)];
    const uint4 r0 = ((const uint4 *)&a.cus[it.x])[0], r1 = ((const uint4 *)&a.cus[it.x])[1];
    const int cu_x = r0.x & 0xFFFF, cu_y = r0.x >> 16;
    const int lw = r0.y & 0xFF, lh = (r0.y >> 8) & 0xFF, pred_mode = (r0.y >> 16) & 0xFF, cbf = r0.y >> 24;
    const int refi0 = (int)(int8_t)(r0.z & 0xFF), refi1 = (int)(int8_t)((r0.z >> 8) & 0xFF), qp_map = (r0.z >> 16) & 0xFF;
    const uint32_t coef_off = r0.w;
    const int cw = 1 << lw, chh = 1 << lh;
    const int qx = q & ((1 << lw2) - 1), qy = q >> lw2;
    const int px0 = cu_x + ((int)(it.y & 0xFF) << 2), py0 = cu_y + ((int)((it.y >> 8) & 0xFF) << 2);      // piece origin
    const int x = px0 + (qx << 2), y = py0 + (qy << 2);
    const int sx = x >> 2, sy = y >> 2;
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
    if (valid) {
        uint32_t m = ((uint32_t)qp_map << 16) | ((uint32_t)intra << 15) | (1u << 31);
        if (pred_mode == XGPU_MODE_SKIP) m |= 1u << 23;
        const bool ibc = pred_mode == XGPU_MODE_IBC;
        if (ibc) m |= 1u << 26;                                  // MCU_SET_IBC (xevdm_def.h:325)
        if (((r0.z >> 24) & 1) && in_tu) m |= 1u << 24;          // CuRec.map_cbf
        // CU boundary, or the 64-sample transform boundary inside a wider CU (deblock_tree splits those, xevdm.c:1989-2037)
        if (((x - cu_x) & 63) == 0) m |= SCU_EDGE_L;
        if (((y - cu_y) & 63) == 0) m |= SCU_EDGE_T;
        uint4 rec;
        rec.x = m;
        rec.y = (intra || ibc) ? 0x0000FFFFu : ((r0.z & 0xFFFFu) | ((uint32_t)ai << 16));
        rec.z = intra ? 0u : r1.x;                               // IBC keeps its block vector in list 0 (xevdm.c:1098-1110)
        rec.w = (intra || ibc) ? 0u : r1.y;
        *(uint4 *)&a.maps[sy * a.w_scu + sx] = rec;
    }
    if (lists == 0) return;      // intra / IBC CUs are reconstructed by k_intra, affine CUs by k_affine; an inter CU without a reference predicts nothing

    // ---- motion: clip like xevd_mv_clip (xevd_mc.c:435-467), variant from the UNCLIPPED vector ----
    // both lists' vectors: unclipped (filter variant) and clipped (position, phase); the loop over the lists below is a real loop with a
    // wave-uniform counter, so everything it indexes by list is selected, not subscripted
    const int mv0x = (int)(int16_t)(r1.x & 0xFFFF), mv0y = (int)(int16_t)(r1.x >> 16), mv1x = (int)(int16_t)(r1.y & 0xFFFF), mv1y = (int)(int16_t)(r1.y >> 16);
    int ct0x, ct0y, ct1x, ct1y;
    {
        const int qxx = cu_x << 2, qyy = cu_y << 2, qw = cw << 2, qh = chh << 2;
        const int min_c = -(128 << 2), max_x = (a.pic_w - 1 + 128) << 2, max_y = (a.pic_h - 1 + 128) << 2;
        auto clipx = [&](int v) { int m = v; if (qxx + v < min_c) m = min_c - qxx; if (qxx + v + qw - 4 > max_x) m = max_x - qxx - qw + 4; return m; };
        auto clipy = [&](int v) { int m = v; if (qyy + v < min_c) m = min_c - qyy; if (qyy + v + qh - 4 > max_y) m = max_y - qyy - qh + 4; return m; };
        ct0x = clipx(mv0x); ct0y = clipy(mv0y); ct1x = clipx(mv1x); ct1y = clipy(mv1y);
    }
    // identical motion in both lists: list 1 is skipped (xevd_mc.c:512-519)
    const bool same = lists == 3 && s_ref[refi0 * 2][1].z == s_ref[refi1 * 2 + 1][1].z && (int16_t)ct0x == (int16_t)ct1x && (int16_t)ct0y == (int16_t)ct1y;

    const bool coop = lw2 > 0 && lh2 > 0;                                        // wave-uniform: square pieces of 8x8, 16x16, 32x32 stage their windows in LDS
    int16_t *buf = s_buf[t >> 6];
    const int l_first = (lists & 1) ? 0 : 1, l_end = (lists & 2) ? 2 : 1;      // wave-uniform
    const int maxl = (1 << a.bd_l) - 1;
    // residual position of this SCU (nothing where nothing is coded)
    const int cwc = tu_w >> 1;
    const uint32_t off_u = coef_off + ((cbf & 1) ? tu_w * tu_h : 0), off_v = off_u + ((cbf & 2) ? cwc * (tu_h >> 1) : 0);

    // ================= luma: both lists, average, residual add + clip (xevd_recon.c:35-71), store =================
    {
        uint32_t rl[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        if (in_tu && (cbf & 1)) {                                     // requested before the reference windows
            const int16_t *r = a.resid + coef_off + ly * tu_w + lx;
#pragma unroll
            for (int k = 0; k < 4; k++) { const uint2 v = *(const uint2 *)(r + k * tu_w); rl[k * 2] = v.x; rl[k * 2 + 1] = v.y; }
        }
        uint32_t pl[8];
        int nl = 0;
#pragma nounroll
        for (int l = l_first; l < l_end; l++) {
            const bool on = !(l && same);
            int qq = q, pp = p;                                       // opaque per iteration: keeps the per-task offsets of the passes below from being
            asm volatile("" : "+v"(qq), "+v"(pp));                    // hoisted out of this loop into a hundred live registers
            const int refi = l ? refi1 : refi0;
            const uint4 e0 = s_ref[(refi < 0 ? 0 : refi) * 2 + l][0];
            const gs16 ry_ = (gs16)(((uint64_t)e0.y << 32) | e0.x);
            const int mvx = l ? mv1x : mv0x, mvy = l ? mv1y : mv0y, ctx = l ? ct1x : ct0x, cty = l ? ct1y : ct0y;
            uint32_t o[8];
            if (coop) {
                // quarter-pel position of the PIECE origin; the phase is the same for every sample of the CU.  A piece with identical motion in
                // both lists runs list 1 like the others of its wave (its lanes share the passes) and drops the result
                const int gx = (px0 << 2) + ctx, gy = (py0 << 2) + cty;
                if (lw2 == 3)      coop_list_luma<32>(ry_, a.s_l, gx, gy, mvx, mvy, a.bd_l, qq, qx, qy, buf, pp, s_ltap, o);
                else if (lw2 == 2) coop_list_luma<16>(ry_, a.s_l, gx, gy, mvx, mvy, a.bd_l, qq, qx, qy, buf, pp, s_ltap, o);
                else               coop_list_luma<8>(ry_, a.s_l, gx, gy, mvx, mvy, a.bd_l, qq, qx, qy, buf, pp, s_ltap, o);
            } else if (on) {
                // pieces 4 samples wide or high: per lane, straight from L1/L2.  Quarter-pel position of this SCU = (x<<2) + clipped mv
                const int px = (x << 2) + ctx, py = (y << 2) + cty;
                const int ldx = (mvx & 3) != 0, ldy = (mvy & 3) != 0;
                const uint4 th = s_ltap[ldx ? ((px & 3) << 2) : 16], tv = s_ltap[ldy ? ((py & 3) << 2) : 16];
                const uint32_t ch[4] = { th.x, th.y, th.z, th.w }, cv[4] = { tv.x, tv.y, tv.z, tv.w };
                const gs16 pp_ = ry_ + ((py >> 2) - 3) * a.s_l + (px >> 2) - 3;
                const Regime rg = regime(ldx, ldy, a.bd_l);
                const bool wh = __ballot(ldx) != 0, wvv = __ballot(ldy) != 0;      // over the lanes that run this list
                if (wh) { if (wvv) mc_luma_4x4<true, true>(pp_, a.s_l, ch, cv, rg, maxl, o); else mc_luma_4x4<true, false>(pp_, a.s_l, ch, cv, rg, maxl, o); }
                else    { if (wvv) mc_luma_4x4<false, true>(pp_, a.s_l, ch, cv, rg, maxl, o); else mc_luma_4x4<false, false>(pp_, a.s_l, ch, cv, rg, maxl, o); }
            }
            if (on) {
#pragma unroll
                for (int k = 0; k < 8; k++) pl[k] = nl ? avg2(pl[k], o[k]) : o[k];
                nl++;
            }
        }
        if (valid && nl) {
            if (cbf & 1) {
#pragma unroll
                for (int k = 0; k < 8; k++) pl[k] = recon2(pl[k], rl[k], maxl);
            }
            int16_t *dy = a.cur_y + y * a.s_l + x;
#pragma unroll
            for (int k = 0; k < 4; k++) *(uint2 *)(dy + k * a.s_l) = make_uint2(pl[k * 2], pl[k * 2 + 1]);
        }
    }
    // ================= chroma: the same for both planes (the LUMA bit depth clips the reconstruction, xevd_recon.c:75-90) =================
    {
        uint32_t ru[2] = {0, 0}, rv[2] = {0, 0};
        if (in_tu && (cbf & 2)) { const int16_t *r = a.resid + off_u + (ly >> 1) * cwc + (lx >> 1); ru[0] = *(const uint32_t *)r; ru[1] = *(const uint32_t *)(r + cwc); }
        if (in_tu && (cbf & 4)) { const int16_t *r = a.resid + off_v + (ly >> 1) * cwc + (lx >> 1); rv[0] = *(const uint32_t *)r; rv[1] = *(const uint32_t *)(r + cwc); }
        uint32_t pu[2], pv[2];
        int nl = 0;
        const int maxc = (1 << a.bd_c) - 1;
#pragma nounroll
        for (int l = l_first; l < l_end; l++) {
            const bool on = !(l && same);
            int qq = q, pp = p;
            asm volatile("" : "+v"(qq), "+v"(pp));
            const int refi = l ? refi1 : refi0;
            const uint4 e0 = s_ref[(refi < 0 ? 0 : refi) * 2 + l][0], e1 = s_ref[(refi < 0 ? 0 : refi) * 2 + l][1];
            const gs16 ru_ = (gs16)(((uint64_t)e0.w << 32) | e0.z), rv_ = (gs16)(((uint64_t)e1.y << 32) | e1.x);
            const int mvx = l ? mv1x : mv0x, mvy = l ? mv1y : mv0y, ctx = l ? ct1x : ct0x, cty = l ? ct1y : ct0y;
            uint32_t ou[2], ov[2];
            if (coop) {
                const int gx = (px0 << 2) + ctx, gy = (py0 << 2) + cty;
                if (lw2 == 3)      coop_list_chroma<32>(ru_, rv_, a.s_c, gx, gy, mvx, mvy, a.bd_c, qq, qx, qy, buf, pp, s_ctap, ou, ov);
                else if (lw2 == 2) coop_list_chroma<16>(ru_, rv_, a.s_c, gx, gy, mvx, mvy, a.bd_c, qq, qx, qy, buf, pp, s_ctap, ou, ov);
                else               coop_list_chroma<8>(ru_, rv_, a.s_c, gx, gy, mvx, mvy, a.bd_c, qq, qx, qy, buf, pp, s_ctap, ou, ov);
            } else if (on) {
                // chroma: 1/8-pel position (x<<2)+mv in luma quarter-pel == chroma eighth-pel; phase in 1/32 = (pos&7)<<2
                const int px = (x << 2) + ctx, py = (y << 2) + cty;
                const int cdx = (mvx & 7) != 0, cdy = (mvy & 7) != 0;
                const uint2 th = s_ctap[cdx ? ((px & 7) << 2) : 32], tv = s_ctap[cdy ? ((py & 7) << 2) : 32];
                uint32_t c2h[2] = { th.x, th.y }, c2v[2] = { tv.x, tv.y };
                const int off = ((py >> 3) - 1) * a.s_c + (px >> 3) - 1;
                const Regime rg = regime(cdx, cdy, a.bd_c);
                const bool wh = __ballot(cdx) != 0, wvv = __ballot(cdy) != 0;
#define MC_C(H, V) do { mc_chroma_2x2<H, V>(ru_ + off, a.s_c, c2h, c2v, rg, maxc, ou); mc_chroma_2x2<H, V>(rv_ + off, a.s_c, c2h, c2v, rg, maxc, ov); } while (0)
                if (wh) { if (wvv) MC_C(true, true); else MC_C(true, false); }
                else    { if (wvv) MC_C(false, true); else MC_C(false, false); }
#undef MC_C
            }
            if (on) {
                pu[0] = nl ? avg2(pu[0], ou[0]) : ou[0]; pu[1] = nl ? avg2(pu[1], ou[1]) : ou[1];
                pv[0] = nl ? avg2(pv[0], ov[0]) : ov[0]; pv[1] = nl ? avg2(pv[1], ov[1]) : ov[1];
                nl++;
            }
        }
        if (valid && nl) {
            if (cbf & 2) { pu[0] = recon2(pu[0], ru[0], maxl); pu[1] = recon2(pu[1], ru[1], maxl); }
            if (cbf & 4) { pv[0] = recon2(pv[0], rv[0], maxl); pv[1] = recon2(pv[1], rv[1], maxl); }
            const int coff = (y >> 1) * a.s_c + (x >> 1);
            *(uint32_t *)(a.cur_u + coff) = pu[0];
            *(uint32_t *)(a.cur_u + coff + a.s_c) = pu[1];
            *(uint32_t *)(a.cur_v + coff) = pv[0];
            *(uint32_t *)(a.cur_v + coff + a.s_c) = pv[1];
        }
    }
}

void launch_mc(xgpu_ctx *c, const InterArgs &a)
{
    if (a.n_waves <= 0) return;
    const int blocks = (a.n_waves + 3) >> 2;
    hipLaunchKernelGGL(k_mc, dim3(((blocks + 7) >> 3) << 3), dim3(256), 0, c->stream, a);
}
