// k_alf.hip - adaptive loop filter (Main profile, sps->tool_alf): 4x4 block classification, 7x7-diamond luma filter,
// 5x5-diamond chroma filter.
//
// Replaces xevd_alf -> call_dec_alf_process_aps -> alf_process -> alf_process_tile -> alf_derive_classification_blk /
// alf_filter_blk_7 / alf_filter_blk_5 (src_main/xevdm.c:2105, src_main/xevdm_alf.c:38-429, 901-1249).  Coefficient
// reconstruction from the APS (alf_recon_coef :700-794) stays on the host; the kernel receives coef_final.
//
// Semantics reproduced: every sample is filtered from the PRE-filter picture; a CTU's 3-sample halo follows the
// reference's per-CTU window rules (own rows mirror at unavailable left/right borders; the rows above/below are
// taken whole from the replicate-extended copy when available - so corner samples next to a picture side border
// are replicated, not mirrored - else mirror the window's own rows; with loop_filter_across_tiles the right and
// bottom picture borders count as available).  With several tiles every tile is filtered from its OWN replicate-extended copy
// (alf_process_tile :934-967): the windows of a CTU end at its tile - "available" across a tile border means the tile's
// replicated edge, "not available" (loop_filter_across_tiles = 0) the mirrored window, never the neighbouring tile's samples.  Class = f(sum of |Laplacians| V,H,D0,D1 over the 8x8 window around
// the 4x4 block, activity), transpose index from the dominant directions, 25 classes x 13 coefficients.
//
// MI355X mapping: out of place (SRC -> DST) like the deblocking passes; one 256-thread workgroup per 64x64 luma
// tile stages the tile + halo of all three planes in LDS (interior by 8-byte coalesced loads, the halo ring through
// the border rule), then one LANE per 4x4 SCU classifies its block with packed-s16 arithmetic (two samples per
// v_pk_* op, row sums via v_dot2 with a ones vector), fetches its 13 coefficients from an LDS copy of the filter
// set and filters its 16 luma + 2x4 chroma samples.  LDS rows are read as aligned 8-byte pieces.
#include "xgpu_internal.h"
#include "addb_filter.h"

typedef short v2s __attribute__((ext_vector_type(2)));

#define LROWS 72          // 70 window rows (-3..66); the fused form keeps the deblocking filter's 72 (-4..67)
#define LSTR  72          // luma LDS row stride in s16: window col c (-3..66) at index c+4
#define CROWS 36
#define CSTR  40          // chroma: window col c (-2..33) at index c+4 (keeps the 8-byte interior pieces aligned)

struct CtuRect { int x0, y0, cw, ch, aL, aR, aT, aB; int tx0, tx1, ty0, ty1; };      // the CTU, its border availability, its tile [tx0,tx1) x [ty0,ty1)

// sample of the reference's per-CTU window at absolute plane position (y,x)  (alf_process_tile :1000-1052)
__device__ __forceinline__ void alf_pos(const CtuRect k, int y, int x, int &yy, int &xx)
{
    yy = y; xx = x;
    if (y < k.y0 && !k.aT) yy = 2 * k.y0 - y;
    else if (y >= k.y0 + k.ch && !k.aB) yy = 2 * (k.y0 + k.ch - 1) - y;
    if (yy >= k.y0 && yy < k.y0 + k.ch) {
        if (x < k.x0 && !k.aL) xx = 2 * k.x0 - x;
        else if (x >= k.x0 + k.cw && !k.aR) xx = 2 * (k.x0 + k.cw - 1) - x;
    }
    yy = min(max(yy, k.ty0), k.ty1 - 1); xx = min(max(xx, k.tx0), k.tx1 - 1);           // the replicate extension of the tile's copy
}
__device__ __forceinline__ int alf_fetch(const int16_t *__restrict__ p, int s, const CtuRect k, int y, int x)
{
    int yy, xx;
    alf_pos(k, y, x, yy, xx);
    return p[yy * s + xx];
}

// stage a T x T tile (+H halo) of one plane into LDS.  lds index of window (r,c) = (r+H)*STR + c + OFF
template <int T, int H, int STR, int OFF>
__device__ __forceinline__ void alf_stage(int16_t *lds, const int16_t *__restrict__ p, int s, const CtuRect k,
                                          int tx0, int ty0, int t0, int nthr)
{
    // fast path (every tile that touches neither the picture border nor an unavailable CTU side - almost all of them): the window rule
    // does nothing, so the tile AND its halo are 8-byte copies of (T + 2H) rows x (T + 8) / 4 pieces (cols -4 .. T+3, window cols -H .. T+H-1 inside)
    static_assert(OFF == 4 && STR == T + 8, "the fast path copies whole LDS rows");
    const bool plain = (k.aL || tx0 > k.x0) && (k.aR || tx0 + T < k.x0 + k.cw) && (k.aT || ty0 > k.y0) && (k.aB || ty0 + T < k.y0 + k.ch) &&
                       tx0 - H >= k.tx0 && tx0 + T + H <= k.tx1 && ty0 - H >= k.ty0 && ty0 + T + H <= k.ty1;
    if (plain) {
        constexpr int PCS = (T + 8) / 4;
        for (int i = t0; i < (T + 2 * H) * PCS; i += nthr) {
            const int r = i / PCS, c4 = (i % PCS) * 4;
            *(uint2 *)(lds + r * STR + c4) = *(const uint2 *)(p + (ty0 - H + r) * s + tx0 - 4 + c4);
        }
        return;
    }
    // interior: T rows x T/4 pieces of 4 samples.  Pieces inside the CTU are rule-free 8-byte copies; when the CTU is
    // cut by the picture border, the part of the tile beyond it belongs to the window's halo and goes through the rule
    for (int i = t0; i < T * (T / 4); i += nthr) {
        const int r = i / (T / 4), c4 = (i % (T / 4)) * 4;
        if (ty0 + r < k.y0 + k.ch && tx0 + c4 + 3 < k.x0 + k.cw) {
            const uint2 v = *(const uint2 *)(p + (ty0 + r) * s + tx0 + c4);
            *(uint2 *)(lds + (r + H) * STR + c4 + OFF) = v;
        } else if (ty0 + r < k.y0 + k.ch + H && tx0 + c4 < k.x0 + k.cw + H) {
#pragma unroll
            for (int e = 0; e < 4; e++) lds[(r + H) * STR + c4 + e + OFF] = (int16_t)alf_fetch(p, s, k, ty0 + r, tx0 + c4 + e);
        }
    }
    // halo ring through the window rule
    constexpr int RING = (T + 2 * H) * (T + 2 * H) - T * T;
    for (int i = t0; i < RING; i += nthr) {
        int r, c;
        if (i < H * (T + 2 * H)) { r = i / (T + 2 * H) - H; c = i % (T + 2 * H) - H; }
        else if (i < 2 * H * (T + 2 * H)) { const int j = i - H * (T + 2 * H); r = T + j / (T + 2 * H); c = j % (T + 2 * H) - H; }
        else { const int j = i - 2 * H * (T + 2 * H); r = j / (2 * H); const int q = j % (2 * H); c = q < H ? q - H : T + q - H; }
        lds[(r + H) * STR + c + OFF] = (int16_t)alf_fetch(p, s, k, ty0 + r, tx0 + c);
    }
}

__device__ __forceinline__ v2s asv(uint32_t x) { return __builtin_bit_cast(v2s, x); }
__device__ __forceinline__ uint32_t asu(v2s x) { return __builtin_bit_cast(uint32_t, x); }
__device__ __forceinline__ v2s vabs(v2s x) { const v2s z = {0, 0}; const v2s n = z - x; return __builtin_elementwise_max(x, n); }
// D = S0.i16 * S1.i16 + S2.i32 on the low / high half of a packed pair: two neighbouring output samples share every packed pair sum
__device__ __forceinline__ int mad_lo(uint32_t ps, int f, int acc) { int r; asm("v_mad_i32_i16 %0, %1, %2, %3" : "=v"(r) : "v"(ps), "v"(f), "v"(acc)); return r; }
__device__ __forceinline__ int mad_hi(uint32_t ps, int f, int acc) { int r; asm("v_mad_i32_i16 %0, %1, %2, %3 op_sel:[1,0,0,0]" : "=v"(r) : "v"(ps), "v"(f), "v"(acc)); return r; }
// ... with the coefficient in the HIGH half of its register (the packed table's pairs)
__device__ __forceinline__ int mad_lo_fh(uint32_t ps, uint32_t f, int acc) { int r; asm("v_mad_i32_i16 %0, %1, %2, %3 op_sel:[0,1,0,0]" : "=v"(r) : "v"(ps), "v"(f), "v"(acc)); return r; }
__device__ __forceinline__ int mad_hi_fh(uint32_t ps, uint32_t f, int acc) { int r; asm("v_mad_i32_i16 %0, %1, %2, %3 op_sel:[1,1,0,0]" : "=v"(r) : "v"(ps), "v"(f), "v"(acc)); return r; }
// ... with a UNIFORM coefficient register (the chroma filter: kernel arguments) as the instruction's scalar operand - an "v" constraint made a v_mov per use
#define SC_ "s"
__device__ __forceinline__ int smad_lo(uint32_t ps, uint32_t f, int acc) { int r; asm("v_mad_i32_i16 %0, %1, %2, %3" : "=v"(r) : "v"(ps), SC_(f), "v"(acc)); return r; }
__device__ __forceinline__ int smad_hi(uint32_t ps, uint32_t f, int acc) { int r; asm("v_mad_i32_i16 %0, %1, %2, %3 op_sel:[1,0,0,0]" : "=v"(r) : "v"(ps), SC_(f), "v"(acc)); return r; }
__device__ __forceinline__ int smad_lo_fh(uint32_t ps, uint32_t f, int acc) { int r; asm("v_mad_i32_i16 %0, %1, %2, %3 op_sel:[0,1,0,0]" : "=v"(r) : "v"(ps), SC_(f), "v"(acc)); return r; }
__device__ __forceinline__ int smad_hi_fh(uint32_t ps, uint32_t f, int acc) { int r; asm("v_mad_i32_i16 %0, %1, %2, %3 op_sel:[1,1,0,0]" : "=v"(r) : "v"(ps), SC_(f), "v"(acc)); return r; }
// the first term of a filter sum: f . x + 256 with the rounding constant as the scalar operand of the three-source form (the accumulating v_dot2c needed a v_mov of 256 per sum)
__device__ __forceinline__ int dot2s_first(uint32_t f, uint32_t x) { int r; asm("v_dot2_i32_i16 %0, %1, %2, %3" : "=v"(r) : "v"(f), "v"(x), SC_(256)); return r; }
__device__ __forceinline__ int sdot2s(uint32_t f, uint32_t x, int acc) { int r; asm("v_dot2_i32_i16 %0, %1, %2, %3" : "=v"(r) : SC_(f), "v"(x), "v"(acc)); return r; }
// two filter sums -> two samples: >> 9, clip to [0, maxv] (one v_med3_i32 each: the compiler cannot know 0 <= maxv and made a max and a min), packed by one byte permute
__device__ __forceinline__ uint32_t clip_pack(int o0, int o1, int maxv)
{
    int a, b;
    asm("v_med3_i32 %0, %1, 0, %2" : "=v"(a) : "v"(o0 >> 9), SC_(maxv));
    asm("v_med3_i32 %0, %1, 0, %2" : "=v"(b) : "v"(o1 >> 9), SC_(maxv));
    return __builtin_amdgcn_perm((uint32_t)b, (uint32_t)a, 0x05040100u);
}
typedef unsigned short v2us __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t padd(uint32_t x, uint32_t y) { return __builtin_bit_cast(uint32_t, __builtin_bit_cast(v2us, x) + __builtin_bit_cast(v2us, y)); }
// (x.lo + y.hi, x.hi + y.lo): the pair sums of two horizontally adjacent taps of ONE output - their mirrored samples sit in a packed pair in reverse order
__device__ __forceinline__ uint32_t padd_x(uint32_t x, uint32_t y) { uint32_t r; asm("v_pk_add_u16 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0]" : "=v"(r) : "v"(x), "v"(y)); return r; }
__device__ __forceinline__ uint32_t pack_f(int lo, int hi) { return ((uint32_t)lo & 0xFFFFu) | ((uint32_t)hi << 16); }
__device__ __forceinline__ int dot2s(uint32_t f, uint32_t x, int acc) { return __builtin_amdgcn_sdot2(__builtin_bit_cast(v2s, f), __builtin_bit_cast(v2s, x), acc, false); }
// |Laplacian| of two neighbouring positions in four directions, accumulated: |2c - a - b| = |2c - (a + b)| on packed u16 pairs is one v_pk_add_u16 and one
// v_sad_u16 (sum of the two absolute differences + accumulator) per direction.  up / mid / dn: three dwords of the rows above, at and below the pair
// (the pair itself = dword 1; dwords 0 / 2 supply its left / right neighbours).  acc = V, H, D0, D1 (alf_derive_classification_blk :83-86)
__device__ __forceinline__ void lap_pair(const uint32_t up[3], const uint32_t mid[3], const uint32_t dn[3], uint32_t acc[4])
{
    const uint32_t c2 = padd(mid[1], mid[1]);
    const uint32_t l0 = __builtin_amdgcn_alignbit(mid[1], mid[0], 16), r0 = __builtin_amdgcn_alignbit(mid[2], mid[1], 16);
    const uint32_t lu = __builtin_amdgcn_alignbit(up[1], up[0], 16), ru = __builtin_amdgcn_alignbit(up[2], up[1], 16);
    const uint32_t ld = __builtin_amdgcn_alignbit(dn[1], dn[0], 16), rd = __builtin_amdgcn_alignbit(dn[2], dn[1], 16);
    acc[0] = __builtin_amdgcn_sad_u16(c2, padd(up[1], dn[1]), acc[0]);
    acc[1] = __builtin_amdgcn_sad_u16(c2, padd(l0, r0), acc[1]);
    acc[2] = __builtin_amdgcn_sad_u16(c2, padd(lu, rd), acc[2]);
    acc[3] = __builtin_amdgcn_sad_u16(c2, padd(ld, ru), acc[3]);
}

// FUSED: the ADDB deblocking filter runs in front of ALF inside the same kernel.  ALF filters a 64x64 tile from the DEBLOCKED samples of the 70x70 window around
// it; the deblocking filter's 8-sample windows (centred on the grid lines, independent of each other, see k_addb.hip) that cover this window are the 72x72 region
// [x0 - 4, x0 + 68) x [y0 - 4, y0 + 68): 9 x 18 vertical-edge segments and 18 x 9 horizontal-edge segments - 162 lanes each - filtered in place in the LDS tile that
// ALF then reads.  The deblocked picture never goes to memory (k_addb_fused writes 100 MB that k_alf reads back: 433 MB of traffic for the two become 270 MB); the
// price is 27 % more deblocking arithmetic (72^2 / 64^2: the halo is filtered by both neighbours).  LDS layout: window row r at row r + RO, luma column c at c + 4,
// chroma column c at c + CO.
// PK (fused form): the deblocking line filters on packed pairs of lines (addb_filter.h; bit depths up to 10)
#ifdef XGPU_ALF_TRACE
// measurement build (make EXTRA=-DXGPU_ALF_TRACE): shader cycles between the marks of alf_kernel, summed over lane 0 of the waves of every 61st workgroup
__device__ unsigned long long g_alf_trace[16];
#define ATR(p) do { if (atr_on) { const unsigned long long now_ = __builtin_amdgcn_s_memtime(); atomicAdd(&g_alf_trace[p], now_ - atr_prev); atr_prev = __builtin_amdgcn_s_memtime(); } } while (0)
extern "C" int xgpu_test_alf_trace(unsigned long long out[16], int reset)
{
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_alf_trace), sizeof(g_alf_trace)) != hipSuccess) return -1;
    if (reset) { static unsigned long long z[16]; if (hipMemcpyToSymbol(HIP_SYMBOL(g_alf_trace), z, sizeof(z)) != hipSuccess) return -1; }
    return 0;
}
#else
#define ATR(p) do { } while (0)
#endif
#ifndef XGPU_ALF_WAVES
#define XGPU_ALF_WAVES 6          // waves per SIMD the fused kernel is compiled for (its registers); the workgroup's LDS allows seven workgroups per CU
#endif
template <bool FUSED, bool PK = false>
__device__ __forceinline__ void alf_kernel(const AlfArgs &a, const AddbArgs *d, const int16_t *__restrict__ sy_, const int16_t *__restrict__ su_,
                                           const int16_t *__restrict__ sv_, int16_t *__restrict__ dy_, int16_t *__restrict__ du_,
                                           int16_t *__restrict__ dv_)
{
    constexpr int RO = FUSED ? 4 : 3, CO = FUSED ? 2 : 4;
    __shared__ __attribute__((aligned(16))) int16_t l_y[LROWS * LSTR];
    __shared__ __attribute__((aligned(16))) int16_t l_c[2][CROWS * CSTR];
    // sums of |Laplacian| (V, H, D0, D1 as one uint4) over the 4x4 blocks of the lattice OFFSET by two samples from the tile's 4x4 blocks - block (m, n) = samples
    // [4m - 2, 4m + 2) x [4n - 2, 4n + 2), m, n = 0..16 - see phase 1 below; the fused form's deblocking state (records, tables, lists: 7.0 KB) lies in the same bytes
    constexpr int LAP_BYTES = 7168;
    static_assert(17 * 17 * 16 <= LAP_BYTES, "the offset-lattice sums fit");
    __shared__ __attribute__((aligned(16))) uint8_t l_lap[LAP_BYTES];
    uint4 (*const l_o)[17] = (uint4 (*)[17])l_lap;
    // the filtered samples on the picture border that this tile holds: [plane][first / last row of the picture][column of the tile] and [first / last column][row],
    // for the border replication below (what k_pad did in a launch of its own)
    // They live in l_y: every lane has its luma window in registers before the barrier behind phase 1, so the staged tile is dead by the time the first filtered
    // sample exists.  (Round 6, measured and not kept: the window in three instalments - rows -3..2 for the Laplacians, 3..4 behind them, 5..6 behind output row 1 -
    // to fit 72 registers = seven waves per SIMD, which the 22.6 KB of LDS would allow: 108.5 us against 104.0 with six; the instalments alone at six waves 105.3.
    // The kernel is bound by instruction issue, a seventh wave only adds contention - tools/exp_variants.sh "lattice new w7".)
    int16_t (*b_row)[2][64] = (int16_t (*)[2][64])l_y, (*b_col)[2][64] = (int16_t (*)[2][64])(l_y + 3 * 2 * 64);
    static_assert(2 * 3 * 2 * 64 <= LROWS * LSTR, "the border arrays fit into the staged luma tile");

    // XCD-aware mapping: workgroup b runs on XCD b % 8 and every XCD has its own L2.  A tile reads its neighbours' edge samples as halo (the 8-byte
    // pieces at cols -4 and 64 sit in the cache lines of the tiles to the left and right, 3 rows above / below in those of the tiles there), so
    // with tiles handed out round-robin every line was fetched from HBM by up to three XCDs: 331 MB read per 8K picture for 126 MB of window
    // (TCC_EA0_RDREQ_128B, profiles/round2_*).  Each XCD now takes a contiguous eighth of the raster tile order: neighbours share an L2.
    const int tiles_x = (a.pic_w + 63) >> 6, n_tiles = tiles_x * ((a.pic_h + 63) >> 6);
    // The XCDs that hold the lower half of the picture walk their eighth backwards: the tiles on the top and bottom picture border also write the padding (below:
    // up to ten times the stores of an inner tile) and should be the first of their XCD, not its tail.
    const int per_xcd = gridDim.x >> 3, xcd = blockIdx.x & 7, seq = blockIdx.x >> 3;
    const int tile = xcd * per_xcd + (xcd >= 4 ? per_xcd - 1 - seq : seq);
    if (tile >= n_tiles) return;
    // (tile / tiles_x by a multiplication: tiles_x is only known at run time, and the division was a reciprocal sequence of ~30 instructions in front of every tile of a
    //  kernel that is bound by instruction issue; floor(2^32 / d) + 1 is exact for n x d < 2^32)
    const int ty = a.magic_tiles_x ? (int)__umulhi((uint32_t)tile, a.magic_tiles_x) : tile, tx = tile - ty * tiles_x;      // (magic 0: one tile per row)
    const int tx0 = tx << 6, ty0 = ty << 6;
    const int t = threadIdx.x;
#ifdef XGPU_ALF_TRACE
    unsigned long long atr_prev = __builtin_amdgcn_s_memtime();
    const bool atr_on = (t & 63) == 0 && tile % 61 == 7;
    if (atr_on) atomicAdd(&g_alf_trace[15], 1ull);
#endif
    // the CTU this tile belongs to and its border availability (alf_process_tile :984-999).  A chain of dependent scalar loads (kernel arguments, the tile masks, the
    // CTU's luma flag): the fused form runs it BEHIND the requests for the tile's region (round 5 ran it in front: a dozen scalar round trips before the first sample
    // was asked for - the "setup" fifth of a wave's life in profiles/round5_exp_addb_alf_wave_life.txt)
    CtuRect k, kc;
    bool luma_on;
    auto ctu_setup = [&]() {
    const int ctu = 1 << a.log2_ctu;
    k.x0 = tx0 & ~(ctu - 1); k.y0 = ty0 & ~(ctu - 1);
    k.cw = min(ctu, a.pic_w - k.x0); k.ch = min(ctu, a.pic_h - k.y0);
    k.tx0 = 0; k.tx1 = a.pic_w; k.ty0 = 0; k.ty1 = a.pic_h;
    if (a.multi_tile) {
        const int cx = k.x0 >> a.log2_ctu, cy = k.y0 >> a.log2_ctu, h_ctu = (a.pic_h + ctu - 1) >> a.log2_ctu;
        k.tx0 = TileMask::tile_first(a.tiles.vb, cx) << a.log2_ctu; k.tx1 = min(TileMask::tile_end(a.tiles.vb, cx, a.w_ctu) << a.log2_ctu, a.pic_w);
        k.ty0 = TileMask::tile_first(a.tiles.hb, cy) << a.log2_ctu; k.ty1 = min(TileMask::tile_end(a.tiles.hb, cy, h_ctu) << a.log2_ctu, a.pic_h);
    }
    // tile_boundary_check against the tile, or - across tiles - against (0, pic_w - 1, 0, pic_h - 1) (:990-999)
    k.aL = a.across_tiles ? k.x0 != 0 : k.x0 != k.tx0; k.aT = a.across_tiles ? k.y0 != 0 : k.y0 != k.ty0;
    k.aR = a.across_tiles ? 1 : (k.x0 + k.cw != k.tx1);
    k.aB = a.across_tiles ? 1 : (k.y0 + k.ch != k.ty1);
    const int ctu_idx = (k.y0 >> a.log2_ctu) * a.w_ctu + (k.x0 >> a.log2_ctu);
    luma_on = a.enable[0] && (a.ctb_in_args ? ((a.ctb_bits[ctu_idx >> 5] >> (ctu_idx & 31)) & 1) != 0 : (a.ctb_flag == nullptr || a.ctb_flag[ctu_idx] != 0));
    kc = k;
    kc.x0 >>= 1; kc.y0 >>= 1; kc.cw >>= 1; kc.ch >>= 1; kc.tx0 >>= 1; kc.tx1 >>= 1; kc.ty0 >>= 1; kc.ty1 >>= 1;
    };
    if (!FUSED) ctu_setup();
    if (!FUSED) {
        if (luma_on) alf_stage<64, 3, LSTR, 4>(l_y, sy_, a.s_l, k, tx0, ty0, t, 256);
        if (a.enable[1]) alf_stage<32, 2, CSTR, 4>(l_c[0], su_, a.s_c, kc, tx0 >> 1, ty0 >> 1, t, 256);
        if (a.enable[2]) alf_stage<32, 2, CSTR, 4>(l_c[1], sv_, a.s_c, kc, tx0 >> 1, ty0 >> 1, t, 256);
        __syncthreads();
    } else {
        // ---- the deblocking filter on the 72 x 72 region around the tile; its SCU records, tables and edge lists borrow l_lap (written by ALF's classification afterwards) ----
        // Edge segments with something to filter are a minority (one grid line in eight inside a 64x64 CU; strength 0 between CUs that move alike) and a wave pays for
        // the whole filter code whenever ONE of its lanes has such an edge.  So the region is moved into LDS unfiltered, every segment decides its strength, the ones
        // to filter are compacted into a list (LDS atomic counter) and as many lanes as the list is long filter them in place: one wave pass instead of three per direction
        // where CUs are large.  This kernel is bound by VALU issue, unlike k_addb_fused, where the same compaction changed nothing.
        const AddbArgs &da = *d;
        uint4 (*s_map)[18] = (uint4 (*)[18])l_lap;
        uint8_t *s_alpha = (uint8_t *)l_lap + 18 * 18 * 16, *s_beta = s_alpha + 52, *s_clip = s_beta + 52, *s_pic = s_clip + 260;
        int8_t *s_cqp = (int8_t *)(s_pic + XGPU_MAX_REFS * 2);
        static_assert(52 + 52 + 260 + XGPU_MAX_REFS * 2 + 192 == ADDB_LDS_TABLE_BYTES, "the LDS tables are AddbArgs.lds_tables");
        constexpr int LIST_OFF = (18 * 18 * 16 + ADDB_LDS_TABLE_BYTES + 3) & ~3;
        uint32_t *s_cnt = (uint32_t *)((uint8_t *)l_lap + LIST_OFF);
        uint32_t (*s_list)[164] = (uint32_t (*)[164])(s_cnt + 2);            // entry = segment | strength << 8 (dwords, like everything else the strength pass stores)
        TileMask *s_tm = (TileMask *)((uint8_t *)l_lap + LIST_OFF + 8 + 2 * 164 * 4);      // the tile-border masks: a lane-indexed read of the kernel arguments is a global load
        static_assert(LIST_OFF + 8 + 2 * 164 * 4 + (int)sizeof(TileMask) <= LAP_BYTES, "the deblocking state fits into l_lap");
#define PK2(lo, hi) ((uint32_t)(uint16_t)(lo) | ((uint32_t)(uint16_t)(hi) << 16))
        // An INTERIOR tile (its 72 x 72 region and the 18 x 18 records around it lie inside the picture: all but the tiles on the picture border) brings region and records
        // straight into LDS (global_load_lds, 16 bytes per lane and request: no staging registers, no LDS store pass - k_inter's region role does the same): l_y is 72 rows of
        // nine chunks, the chroma tiles 2 x 36 rows of five, the records 324 chunks, each array contiguous, chunk c of an array requested by thread c mod 256.  Both
        // directions' strengths are then decided in ONE pass of all lanes (324 segments) instead of the vertical ones behind the store pass and the horizontal ones behind
        // the vertical filters.
        const bool interior = tx0 > 0 && ty0 > 0 && tx0 + 68 <= a.pic_w && ty0 + 68 <= a.pic_h;
        if (interior) {
            const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
            typedef __attribute__((address_space(3))) char lds_char;
            typedef const __attribute__((address_space(1))) char glb_char;
            {
                const int r0 = (t * 57) >> 9, c0 = t - r0 * 9;                        // t / 9, t % 9 (t < 256)
                glb_char *const g0 = (glb_char *)(sy_ + (ty0 - 4) * a.s_l + tx0 - 4);
                lds_char *const d0 = (lds_char *)l_y + wave * 1024;
                const uint32_t rs = (uint32_t)a.s_l << 1;
                // chunk t + 256 it: 256 = 28 rows + 4 chunks, 512 = 56 rows + 8 chunks
                const int c1 = c0 + 4 >= 9 ? c0 - 5 : c0 + 4, r1 = r0 + 28 + (c0 + 4 >= 9), c2 = c0 + 8 >= 9 ? c0 - 1 : c0 + 8, r2 = r0 + 56 + (c0 + 8 >= 9);
                __builtin_amdgcn_global_load_lds((glb_char *)(g0 + r0 * rs + c0 * 16), (lds_char *)d0, 16, 0, 0);
                __builtin_amdgcn_global_load_lds((glb_char *)(g0 + r1 * rs + c1 * 16), (lds_char *)(d0 + 4096), 16, 0, 0);
                if (t < 648 - 512) __builtin_amdgcn_global_load_lds((glb_char *)(g0 + r2 * rs + c2 * 16), (lds_char *)(d0 + 8192), 16, 0, 0);
            }
            {
                const uint32_t rs = (uint32_t)a.s_c << 1;
                const uint32_t cbase = (uint32_t)(((ty0 >> 1) - 2) * a.s_c + (tx0 >> 1) - 2) << 1;
                lds_char *const d0 = (lds_char *)&l_c[0][0] + wave * 1024;
#pragma unroll
                for (int it = 0; it < 2; it++) {
                    const int idx = t + 256 * it, pl = idx >= 180, j = idx - 180 * pl, r = (j * 205) >> 10, c = j - r * 5;      // j / 5 (j < 180)
                    if (it == 0 || t < 360 - 256)
                        __builtin_amdgcn_global_load_lds((glb_char *)((glb_char *)(pl ? sv_ : su_) + cbase + r * rs + c * 16), (lds_char *)(d0 + 4096 * it), 16, 0, 0);
                }
            }
            {
                glb_char *const g0 = (glb_char *)(da.maps + ((ty0 >> 2) - 1) * da.w_scu + (tx0 >> 2) - 1);
                lds_char *const d0 = (lds_char *)l_lap + wave * 1024;
                const uint32_t rs = (uint32_t)da.w_scu << 4;
#pragma unroll
                for (int it = 0; it < 2; it++) {
                    const int idx = t + 256 * it, r = (idx * 57) >> 10, c = idx - r * 18;      // idx / 18 (idx < 324)
                    if (it == 0 || t < 324 - 256) __builtin_amdgcn_global_load_lds((glb_char *)(g0 + r * rs + c * 16), (lds_char *)(d0 + 4096 * it), 16, 0, 0);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
            {
                // the deblocking tables (alpha, beta, clip, reference identities, chroma QP mapping: one block the host lays out, AddbArgs.lds_tables) and the tile masks, a
                // dword per thread (round 5 copied them byte by byte from five places: six loads and six byte stores per wave)
                static_assert(ADDB_LDS_TABLE_DWORDS + 16 <= 256, "one table dword per thread");
                if (t < ADDB_LDS_TABLE_DWORDS) ((uint32_t *)s_alpha)[t] = da.lds_tables[t];
                else if (t < ADDB_LDS_TABLE_DWORDS + 16) { const int k_ = t - ADDB_LDS_TABLE_DWORDS; ((uint32_t *)s_tm)[k_] = k_ < 8 ? da.no_filter.vb[k_] : da.no_filter.hb[k_ - 8]; }
                if (t < 2) s_cnt[t] = 0;
            }
            ctu_setup();
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // this wave's chunks have landed (the compiler does not know that the requests write LDS)
            ATR(1);
            __syncthreads();
            ATR(2);
            for (int it = t; it < 324; it += 256) {
                const bool hor = it >= 162;
                const int j = hor ? it - 162 : it;
                int bs;
                if (!hor) { const int wx = j % 9, sr = j / 9; bs = addb_edge_strength_rt(da, *s_tm, s_map[sr][2 * wx + 1], s_map[sr][2 * wx], (tx0 >> 2) + 2 * wx, false, s_pic); }
                else      { const int sx = j % 18, g = j / 18; bs = addb_edge_strength_rt(da, *s_tm, s_map[2 * g + 1][sx], s_map[2 * g][sx], (ty0 >> 2) + 2 * g, true, s_pic); }
                if (bs) s_list[hor][atomicAdd(&s_cnt[hor], 1u)] = (uint32_t)(j | (bs << 8));
            }
        } else
        {   // phase A: lane = vertical-edge window wx (grid line x0 + 8 wx) x SCU row sr of the region: window and SCU records from memory to LDS
            const int wx = t % 9, sr = t / 9;
            const int gx = tx0 + 8 * wx, sxq = gx >> 2, srow = (ty0 >> 2) - 1 + sr;
            const bool ok = t < 162 && gx <= a.pic_w && srow >= 0 && srow < da.h_scu;
            const bool has_p = ok && gx > 0, has_q = ok && gx < a.pic_w;
            const uint4 *maps = (const uint4 *)da.maps;
            uint4 rq = make_uint4(0, 0, 0, 0), rp = rq;
            uint4 L[4] = { rq, rq, rq, rq };
            uint2 C[2][2] = { { make_uint2(0, 0), make_uint2(0, 0) }, { make_uint2(0, 0), make_uint2(0, 0) } };
            if (ok) {
                const int kq = srow * da.w_scu + sxq;
                if (has_q) rq = maps[kq];
                if (has_p) rp = maps[kq - 1];
                const int y = srow << 2, cy = srow << 1;
#pragma unroll
                for (int r = 0; r < 4; r++) { const U32x4a8 v = *(const U32x4a8 *)(sy_ + (y + r) * a.s_l + gx - 4); L[r] = make_uint4(v.a, v.b, v.c, v.d); }
#pragma unroll
                for (int pl = 0; pl < 2; pl++)
#pragma unroll
                    for (int r = 0; r < 2; r++) { const U32x2a4 v = *(const U32x2a4 *)((pl ? sv_ : su_) + (cy + r) * a.s_c + (gx >> 1) - 2); C[pl][r] = make_uint2(v.a, v.b); }
            }
            // the small tables into LDS only now: each of these copies waits for its own loads, and in front of the window loads they were four memory round trips
            // in a row before the first window load of the tile had even been issued
            // (and all of their loads before the first of their stores: one more round trip, not six)
            {
                // the deblocking tables (alpha, beta, clip, reference identities, chroma QP mapping: one block the host lays out, AddbArgs.lds_tables) and the tile masks, a
                // dword per thread (round 5 copied them byte by byte from five places: six loads and six byte stores per wave)
                static_assert(ADDB_LDS_TABLE_DWORDS + 16 <= 256, "one table dword per thread");
                if (t < ADDB_LDS_TABLE_DWORDS) ((uint32_t *)s_alpha)[t] = da.lds_tables[t];
                else if (t < ADDB_LDS_TABLE_DWORDS + 16) { const int k_ = t - ADDB_LDS_TABLE_DWORDS; ((uint32_t *)s_tm)[k_] = k_ < 8 ? da.no_filter.vb[k_] : da.no_filter.hb[k_ - 8]; }
                if (t < 2) s_cnt[t] = 0;
            }
            ctu_setup();
            ATR(1);
            __syncthreads();                                 // the tables and the counters (the loads above are in flight across it)
            ATR(2);
            if (t < 162) {
                s_map[sr][2 * wx] = rp; s_map[sr][2 * wx + 1] = rq;
#pragma unroll
                for (int r = 0; r < 4; r++) *(uint4 *)(l_y + (4 * sr + r) * LSTR + 8 * wx) = L[r];
#pragma unroll
                for (int pl = 0; pl < 2; pl++)
#pragma unroll
                    for (int r = 0; r < 2; r++) *(uint2 *)(l_c[pl] + (2 * sr + r) * CSTR + 4 * wx) = C[pl][r];
                const int bs = has_p && has_q ? addb_edge_strength<0>(da, *s_tm, rq, rp, sxq, s_pic) : 0;
                if (bs) s_list[0][atomicAdd(&s_cnt[0], 1u)] = (uint32_t)(t | (bs << 8));
            }
        }
        ATR(3);
        __syncthreads();
        ATR(4);
        // phase B: the vertical edges to filter, in place.
        // PK form (round 6): the four WAVES of the workgroup take the four parts of a listed segment - luma lines 0 / 1, luma lines 2 / 3, the U lines, the V lines - and lane
        // l of each wave takes list entry l.  Round 5 gave a whole segment to one lane: with the usual 20 .. 60 entries one wave ran all ~580 instructions of the filter code
        // while the other three waited at the barrier - two such phases were 40 % of a workgroup's life (profiles/round5_exp_addb_alf_wave_life.txt), during which it held its
        // sixth of the CU's LDS with a quarter of its lanes.  The parts touch disjoint samples; the instructions issued are about the same, the phase is a quarter as long.
        if (PK) {
            const int role = __builtin_amdgcn_readfirstlane(t >> 6);
            const int maxl = (1 << da.bd_l) - 1, maxc = (1 << da.bd_c) - 1;
            for (int i = t & 63; i < (int)s_cnt[0]; i += 64) {
                const int e = s_list[0][i], seg = e & 255, bs = e >> 8, wx = seg % 9, sr = seg / 9;
                const uint32_t rpx = s_map[sr][2 * wx].x, rqx = s_map[sr][2 * wx + 1].x;
                if (role < 2) {
                    // lines = rows here: the pair (row 2 role, row 2 role + 1) is formed with byte permutes, sample by sample across the edge
                    int alpha, beta, c1;
                    addb_luma_params(da, rqx, rpx, bs, s_alpha, s_beta, s_clip, alpha, beta, c1);
                    int16_t *const w0 = l_y + (4 * sr + 2 * role) * LSTR + 8 * wx;
                    const uint4 R0 = *(const uint4 *)w0, R1 = *(const uint4 *)(w0 + LSTR);
                    const uint32_t a0[4] = { R0.x, R0.y, R0.z, R0.w }, a1[4] = { R1.x, R1.y, R1.z, R1.w };
                    uint32_t LP[8], b0[4], b1[4];
#pragma unroll
                    for (int j = 0; j < 4; j++) { LP[2 * j] = __builtin_amdgcn_perm(a1[j], a0[j], 0x05040100u); LP[2 * j + 1] = __builtin_amdgcn_perm(a1[j], a0[j], 0x07060302u); }
                    addb_line_luma_pk(LP, bs, alpha, beta, c1, da.bd_l, maxl);
#pragma unroll
                    for (int j = 0; j < 4; j++) { b0[j] = __builtin_amdgcn_perm(LP[2 * j + 1], LP[2 * j], 0x05040100u); b1[j] = __builtin_amdgcn_perm(LP[2 * j + 1], LP[2 * j], 0x07060302u); }
                    *(uint4 *)w0 = make_uint4(b0[0], b0[1], b0[2], b0[3]);
                    *(uint4 *)(w0 + LSTR) = make_uint4(b1[0], b1[1], b1[2], b1[3]);
                } else {
                    const int pl = role - 2;
                    int alpha, beta, c0v;
                    if (addb_chroma_params<0>(da, rqx, rpx, bs, pl, s_alpha, s_beta, s_clip, s_cqp, alpha, beta, c0v)) {
                        int16_t *const w0 = l_c[pl] + (2 * sr) * CSTR + 4 * wx;
                        const uint2 c0 = *(const uint2 *)w0, c1 = *(const uint2 *)(w0 + CSTR);
                        uint32_t CP[4] = { __builtin_amdgcn_perm(c1.x, c0.x, 0x05040100u), __builtin_amdgcn_perm(c1.x, c0.x, 0x07060302u),
                                           __builtin_amdgcn_perm(c1.y, c0.y, 0x05040100u), __builtin_amdgcn_perm(c1.y, c0.y, 0x07060302u) };
                        addb_line_chroma_pk(CP, bs, alpha, beta, c0v, maxc);
                        *(uint2 *)w0 = make_uint2(__builtin_amdgcn_perm(CP[1], CP[0], 0x05040100u), __builtin_amdgcn_perm(CP[3], CP[2], 0x05040100u));
                        *(uint2 *)(w0 + CSTR) = make_uint2(__builtin_amdgcn_perm(CP[1], CP[0], 0x07060302u), __builtin_amdgcn_perm(CP[3], CP[2], 0x07060302u));
                    }
                }
            }
        }
        for (int i = t; !PK && i < (int)s_cnt[0]; i += 256) {
            const int e = s_list[0][i], seg = e & 255, bs = e >> 8, wx = seg % 9, sr = seg / 9;
            const uint4 rp = s_map[sr][2 * wx], rq = s_map[sr][2 * wx + 1];
            int L[4][8], Cc[2][2][4];
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const uint4 v = *(const uint4 *)(l_y + (4 * sr + r) * LSTR + 8 * wx);
                L[r][0] = (int16_t)(v.x & 0xFFFF); L[r][1] = (int16_t)(v.x >> 16); L[r][2] = (int16_t)(v.y & 0xFFFF); L[r][3] = (int16_t)(v.y >> 16);
                L[r][4] = (int16_t)(v.z & 0xFFFF); L[r][5] = (int16_t)(v.z >> 16); L[r][6] = (int16_t)(v.w & 0xFFFF); L[r][7] = (int16_t)(v.w >> 16);
            }
#pragma unroll
            for (int pl = 0; pl < 2; pl++)
#pragma unroll
                for (int r = 0; r < 2; r++) {
                    const uint2 v = *(const uint2 *)(l_c[pl] + (2 * sr + r) * CSTR + 4 * wx);
                    Cc[pl][r][0] = (int16_t)(v.x & 0xFFFF); Cc[pl][r][1] = (int16_t)(v.x >> 16); Cc[pl][r][2] = (int16_t)(v.y & 0xFFFF); Cc[pl][r][3] = (int16_t)(v.y >> 16);
                }
            addb_edge_filter<0>(da, rq, rp, bs, L, Cc, s_alpha, s_beta, s_clip, s_cqp);
#pragma unroll
            for (int r = 0; r < 4; r++)
                *(uint4 *)(l_y + (4 * sr + r) * LSTR + 8 * wx) = make_uint4(PK2(L[r][0], L[r][1]), PK2(L[r][2], L[r][3]), PK2(L[r][4], L[r][5]), PK2(L[r][6], L[r][7]));
#pragma unroll
            for (int pl = 0; pl < 2; pl++)
#pragma unroll
                for (int r = 0; r < 2; r++)
                    *(uint2 *)(l_c[pl] + (2 * sr + r) * CSTR + 4 * wx) = make_uint2(PK2(Cc[pl][r][0], Cc[pl][r][1]), PK2(Cc[pl][r][2], Cc[pl][r][3]));
        }
        if (!interior && t < 162) {   // (border tiles) the horizontal edges to filter (the records are complete since the barrier): lane = SCU column sx x grid line y0 + 8 g of the region
            const int sx = t % 18, g = t / 18;
            const int scol = (tx0 >> 2) - 1 + sx, gy = ty0 + 8 * g;
            if (scol >= 0 && scol < da.w_scu && gy > 0 && gy < a.pic_h) {
                const int bs = addb_edge_strength<1>(da, *s_tm, s_map[2 * g + 1][sx], s_map[2 * g][sx], gy >> 2, s_pic);
                if (bs) s_list[1][atomicAdd(&s_cnt[1], 1u)] = (uint32_t)(t | (bs << 8));
            }
        }
        ATR(5);
        __syncthreads();
        ATR(6);
        // phase C: the horizontal edges, in place (PK: the four parts of a segment on the four waves, as in phase B)
        if (PK) {
            const int role = __builtin_amdgcn_readfirstlane(t >> 6);
            const int maxl = (1 << da.bd_l) - 1, maxc = (1 << da.bd_c) - 1;
            for (int i = t & 63; i < (int)s_cnt[1]; i += 64) {
                const int e = s_list[1][i], seg = e & 255, bs = e >> 8, sx = seg % 18, g = seg / 18;
                const uint32_t rqx = s_map[2 * g + 1][sx].x, rpx = s_map[2 * g][sx].x;
                if (role < 2) {
                    // lines = columns here: a row's dwords ARE the pairs (column 0, column 1) and (column 2, column 3) of the segment - no unpacking at all
                    int alpha, beta, c1;
                    addb_luma_params(da, rqx, rpx, bs, s_alpha, s_beta, s_clip, alpha, beta, c1);
                    int16_t *const w0 = l_y + (8 * g) * LSTR + 4 * sx + 2 * role;
                    uint32_t LP[8];
#pragma unroll
                    for (int r = 0; r < 8; r++) LP[r] = *(const uint32_t *)(w0 + r * LSTR);
                    addb_line_luma_pk(LP, bs, alpha, beta, c1, da.bd_l, maxl);
#pragma unroll
                    for (int r = 0; r < 8; r++) *(uint32_t *)(w0 + r * LSTR) = LP[r];
                } else {
                    const int pl = role - 2;
                    int alpha, beta, c0v;
                    if (addb_chroma_params<1>(da, rqx, rpx, bs, pl, s_alpha, s_beta, s_clip, s_cqp, alpha, beta, c0v)) {
                        int16_t *const w0 = l_c[pl] + (4 * g) * CSTR + 2 * sx;
                        uint32_t CP[4];
#pragma unroll
                        for (int r = 0; r < 4; r++) CP[r] = *(const uint32_t *)(w0 + r * CSTR);
                        addb_line_chroma_pk(CP, bs, alpha, beta, c0v, maxc);
#pragma unroll
                        for (int r = 0; r < 4; r++) *(uint32_t *)(w0 + r * CSTR) = CP[r];
                    }
                }
            }
        }
        for (int i = t; !PK && i < (int)s_cnt[1]; i += 256) {
            const int e = s_list[1][i], seg = e & 255, bs = e >> 8, sx = seg % 18, g = seg / 18;
            const uint4 rq = s_map[2 * g + 1][sx], rp = s_map[2 * g][sx];
            int L[4][8], Cc[2][2][4];
#pragma unroll
            for (int r = 0; r < 8; r++) {
                const uint2 v = *(const uint2 *)(l_y + (8 * g + r) * LSTR + 4 * sx);
                L[0][r] = (int16_t)(v.x & 0xFFFF); L[1][r] = (int16_t)(v.x >> 16); L[2][r] = (int16_t)(v.y & 0xFFFF); L[3][r] = (int16_t)(v.y >> 16);
            }
#pragma unroll
            for (int pl = 0; pl < 2; pl++)
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    const uint32_t v = *(const uint32_t *)(l_c[pl] + (4 * g + r) * CSTR + 2 * sx);
                    Cc[pl][0][r] = (int16_t)(v & 0xFFFF); Cc[pl][1][r] = (int16_t)(v >> 16);
                }
            addb_edge_filter<1>(da, rq, rp, bs, L, Cc, s_alpha, s_beta, s_clip, s_cqp);
#pragma unroll
            for (int r = 0; r < 8; r++) *(uint2 *)(l_y + (8 * g + r) * LSTR + 4 * sx) = make_uint2(PK2(L[0][r], L[1][r]), PK2(L[2][r], L[3][r]));
#pragma unroll
            for (int pl = 0; pl < 2; pl++)
#pragma unroll
                for (int r = 0; r < 4; r++) *(uint32_t *)(l_c[pl] + (4 * g + r) * CSTR + 2 * sx) = PK2(Cc[pl][0][r], Cc[pl][1][r]);
        }
#undef PK2
        ATR(7);
        __syncthreads();
        ATR(8);
        // ---- ALF's window rule on the tiles that touch a picture / tile border or an unavailable CTU side: every window position takes the sample the rule names
        //      (alf_fetch's position mapping is idempotent - a position it names maps to itself - so the pass runs in place) ----
        const bool plain = (k.aL || tx0 > k.x0) && (k.aR || tx0 + 64 < k.x0 + k.cw) && (k.aT || ty0 > k.y0) && (k.aB || ty0 + 64 < k.y0 + k.ch) &&
                           tx0 - 3 >= k.tx0 && tx0 + 67 <= k.tx1 && ty0 - 3 >= k.ty0 && ty0 + 67 <= k.ty1;
        if (!plain) {
            for (int i = t; i < 70 * 70; i += 256) {
                const int r = i / 70 - 3, cc = i % 70 - 3;
                int yy, xx;
                alf_pos(k, ty0 + r, tx0 + cc, yy, xx);
                l_y[(r + RO) * LSTR + cc + 4] = l_y[(yy - ty0 + RO) * LSTR + xx - tx0 + 4];
            }
            for (int i = t; i < 2 * 36 * 36; i += 256) {
                const int pl = i / (36 * 36), j = i - pl * 36 * 36, r = j / 36 - 2, cc = j % 36 - 2;
                int yy, xx;
                alf_pos(kc, (ty0 >> 1) + r, (tx0 >> 1) + cc, yy, xx);
                l_c[pl][(r + 2) * CSTR + cc + CO] = l_c[pl][(yy - (ty0 >> 1) + 2) * CSTR + xx - (tx0 >> 1) + CO];
            }
            __syncthreads();
        }
    }

    // lane -> 4x4 block: a wave takes an 8 x 8 quarter of the tile's 16 x 16 blocks (round 5: 16 x 4).  A lane reads window rows as 8-byte pieces at
    // (4 ly + i) * 144 + 8 lx bytes: with eight blocks per row a half-wave covers banks 16 ly + 2 lx + {0, 1} exactly once - sixteen blocks per row put two lanes
    // on every bank (SQ_LDS_BANK_CONFLICT 41 % of the LDS cycles, profiles/round5_exp_inter_counters.txt); the sub-block sums (72-byte rows) and the chroma tile (80-byte
    // rows) spread the same way
    const int lx = (t & 7) | ((t >> 3) & 8), ly = ((t >> 3) & 7) | ((t >> 4) & 8);
    const int x = tx0 + (lx << 2), y = ty0 + (ly << 2);
    const bool inside = x < a.pic_w && y < a.pic_h;
    const int maxv = (1 << a.bd) - 1;

    // ------------------------------------------------ luma -----------------------------------------------
    // window rows -3..6, each 12 samples (cols -4..7) as 6 dwords; col j sits at sample j+4
    uint32_t W[10][6];
    uint2 keep_w[4];                  // fused, luma filter off in this CTU: the lane's deblocked block, read before b_row / b_col reuse the staged tile
    if (FUSED && !luma_on) {
#pragma unroll
        for (int ii = 0; ii < 4; ii++) keep_w[ii] = *(const uint2 *)(l_y + ((ly << 2) + ii + RO) * LSTR + (lx << 2) + 4);
    }
    if (luma_on) {
#pragma unroll
        for (int i = 0; i < 10; i++) {
            const uint2 *row = (const uint2 *)(l_y + ((ly << 2) + i + RO - 3) * LSTR + (lx << 2));
            const uint2 v0 = row[0], v1 = row[1], v2 = row[2];
            W[i][0] = v0.x; W[i][1] = v0.y; W[i][2] = v1.x; W[i][3] = v1.y; W[i][4] = v2.x; W[i][5] = v2.y;
        }
        // Classification, phase 1 (alf_derive_classification_blk :38-208): the reference sums |Laplacian| (vertical, horizontal, two diagonals) over the 8x8 window around a
        // 4x4 block - the block and two samples on every side.  That window is exactly four blocks of the lattice shifted by (-2, -2): with O(m, n) = the sums over samples
        // [4m - 2, 4m + 2) x [4n - 2, 4n + 2), window(lx, ly) = O(lx, ly) + O(lx + 1, ly) + O(lx, ly + 1) + O(lx + 1, ly + 1).  So a lane computes the sixteen Laplacians
        // of ITS offset block (rows / columns -2 .. 1 of its window: the same work as its own sixteen positions), keeps the four sums and shares them as one 16-byte store;
        // phase 2 is three 16-byte loads and eight additions.  (Rounds 1-5 shared 2x2 sub-block sums: eight stores, 24 two-dword loads and 48 dot products per lane, and
        // the array was 9.8 KB.)  The 33 blocks with m = 16 or n = 16 belong to no lane: the first 33 threads take them from the staged tile.  Sums as u32 (a block's sum
        // reaches 16 x 2 x 4095 at 12 bits).
        {
            uint32_t acc[4] = { 0, 0, 0, 0 };
#pragma unroll
            for (int i = 1; i < 5; i++)
#pragma unroll
                for (int sc = 0; sc < 2; sc++) lap_pair(&W[i - 1][sc], &W[i][sc], &W[i + 1][sc], acc);
            l_o[ly][lx] = make_uint4(acc[0], acc[1], acc[2], acc[3]);
        }
        if (t < 33) {
            const int m = t < 17 ? 16 : t - 17, n = t < 17 ? t : 16;
            // sample rows 4n - 3 .. 4n + 2, samples 4m - 4 .. 4m + 3 = four aligned dwords from LDS column index 4m
            uint32_t R[6][4], acc[4] = { 0, 0, 0, 0 };
#pragma unroll
            for (int r = 0; r < 6; r++) {
                const uint2 *row = (const uint2 *)(l_y + (4 * n - 3 + r + RO) * LSTR + 4 * m);
                const uint2 v0 = row[0], v1 = row[1];
                R[r][0] = v0.x; R[r][1] = v0.y; R[r][2] = v1.x; R[r][3] = v1.y;
            }
#pragma unroll
            for (int i = 1; i < 5; i++)
#pragma unroll
                for (int sc = 0; sc < 2; sc++) lap_pair(&R[i - 1][sc], &R[i][sc], &R[i + 1][sc], acc);
            l_o[n][m] = make_uint4(acc[0], acc[1], acc[2], acc[3]);
        }
    }
    ATR(9);
    __syncthreads();
    ATR(10);
    // output addresses as unsigned byte offsets from the (uniform) plane pointers: one scalar base + a 32-bit lane offset per store, one add per row (the 64-bit
    // form was a multiply, a sign extension and a 64-bit shift-add per row)
    const uint32_t ol0 = (uint32_t)(y * a.s_l + x) << 1, oc0 = (uint32_t)((y >> 1) * a.s_c + (x >> 1)) << 1, sl2 = (uint32_t)a.s_l << 1, sc2 = (uint32_t)a.s_c << 1;
#ifdef XGPU_EXP_PAD_VALU
    {   // measurement: what does one more VALU instruction per wave cost this kernel?  N dependent byte permutes on a value that ends up in an (always false) store condition
        uint32_t pv = (uint32_t)t;
#pragma unroll
        for (int k_ = 0; k_ < XGPU_EXP_PAD_VALU; k_++) asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(pv) : "v"(x), "v"(y));
        if (pv == 0x12345678u && x < 0) l_lap[0] = 1;
    }
#endif
    const bool edge_l = tx0 == 0, edge_r = tx0 + 64 >= a.pic_w, edge_t = ty0 == 0, edge_b = ty0 + 64 >= a.pic_h;
    const bool keep_border = a.pad && (edge_l || edge_r || edge_t || edge_b);      // workgroup-uniform: an inner tile's lanes do not test their rows against the picture borders
    do {
    if (!inside) break;
    // a lane's block on the picture border leaves its outermost samples for the replication
    auto keep_luma = [&](int ii, uint2 w) {
        if (keep_border) {
            if (y + ii == 0) *(uint2 *)&b_row[0][0][lx << 2] = w;
            if (y + ii == a.pic_h - 1) *(uint2 *)&b_row[0][1][lx << 2] = w;
            if (x == 0) b_col[0][0][(ly << 2) + ii] = (int16_t)(w.x & 0xFFFF);
            if (x + 4 == a.pic_w) b_col[0][1][(ly << 2) + ii] = (int16_t)(w.y >> 16);
        }
    };

    if (luma_on) {
        // phase 2: the 8x8 window = the lane's offset block and its right, lower and lower-right neighbours
        const uint4 own_lap = l_o[ly][lx], o10 = l_o[ly][lx + 1], o01 = l_o[ly + 1][lx], o11 = l_o[ly + 1][lx + 1];      // (the lane's own sums too: four registers less across the barrier)
        const int sum[4] = { (int)(own_lap.x + o10.x + o01.x + o11.x), (int)(own_lap.y + o10.y + o01.y + o11.y), (int)(own_lap.z + o10.z + o01.z + o11.z),
                             (int)(own_lap.w + o10.w + o01.w + o11.w) };
        const int sv = sum[0], sh = sum[1], sd0 = sum[2], sd1 = sum[3];
        int cls, tr;
        {
            const int act = min(max((sv + sh) >> (a.bd - 2), 0), 15);
            cls = (0x4333333332222210ull >> (act * 4)) & 0xF;                 // th[16] = {0,1,2,2,2,2,2,3,3,3,3,3,3,3,3,4}
            int hv1, hv0, d1, d0, dir_hv, dir_d, hvd1, hvd0, main_dir, sec_dir;
            if (sv > sh) { hv1 = sv; hv0 = sh; dir_hv = 1; } else { hv1 = sh; hv0 = sv; dir_hv = 3; }
            if (sd0 > sd1) { d1 = sd0; d0 = sd1; dir_d = 0; } else { d1 = sd1; d0 = sd0; dir_d = 2; }
            if (d1 * hv0 > hv1 * d0) { hvd1 = d1; hvd0 = d0; main_dir = dir_d; sec_dir = dir_hv; }
            else { hvd1 = hv1; hvd0 = hv0; main_dir = dir_hv; sec_dir = dir_d; }
            int strength = 0;
            if (hvd1 > 2 * hvd0) strength = 1;
            if (hvd1 * 2 > 9 * hvd0) strength = 2;
            if (strength) cls += (((main_dir & 1) << 1) + strength) * 5;
            tr = (0x31322010u >> ((main_dir * 2 + (sec_dir >> 1)) * 4)) & 0xF;  // trans_tbl = {0,1,0,2,2,3,1,3}
        }
        // the block's filter: two 16-byte loads from the packed class x transpose table in the kernel arguments (L1 / L2 resident: 3.2 KB for the whole picture)
        const uint4 *const fe = (const uint4 *)a.ctab[(cls << 2) + tr];
        const uint4 fe0 = fe[0], fe1 = fe[1];
        // The filter of alf_filter_blk_7 (xevdm_alf.c:210-337): sum_k f[k] * (S(i+dy_k, j+dx_k) + S(i-dy_k, j-dx_k)) + f[12] * S(i, j), + 256 >> 9.
        // Two neighbouring outputs (j, j+1) at a time: their symmetric sample pairs are PACKED pairs of the window - P(i, c) = (S(i, c),
        // S(i, c+1)), a window dword or one v_alignbit - so one v_pk_add_i16 makes both pair sums (<= 2 * 4095, exact in s16) and two
        // v_mad_i32_i16 (low / high half) accumulate them: 38 VALU per output pair instead of ~100 with scalar extracts.
#define P(i, c) ((((c) + 4) & 1) ? __builtin_amdgcn_alignbit(W[(i) + 3][(((c) + 4) >> 1) + 1], W[(i) + 3][((c) + 4) >> 1], 16) : W[(i) + 3][((c) + 4) >> 1])
        // Round 3: ten of the twelve symmetric taps come in horizontally adjacent couples (3,2) (8,7) (6,5) (10,9) + (centre,11).  For ONE output the two
        // first samples of a couple are a packed pair of the window and the two mirrored samples are a packed pair in reverse order, so one
        // v_pk_add_u16 with swapped halves forms both pair sums and one v_dot2_i32_i16 against the packed coefficient couple accumulates them: 2
        // instructions per output and couple (was 3 per output PAIR and tap = 3 per output and couple).  The taps without a neighbour (0, 1, 4) stay
        // on the two-outputs-per-register form: one v_pk_add_u16 + two v_mad_i32_i16 per output pair.
        const uint32_t F32 = fe0.x, F87 = fe0.y, F65 = fe0.z, FA9 = fe0.w, FCB = fe1.x, F10 = fe1.y, F4 = fe1.z;
#pragma unroll
        for (int ii = 0; ii < 4; ii++) {
            int o[4];
#pragma unroll
            for (int jj = 0; jj < 4; jj++) {
                int acc = dot2s_first(F32, padd_x(P(ii + 2, jj - 1), P(ii - 2, jj)));
                acc = dot2s(F87, padd_x(P(ii + 1, jj - 2), P(ii - 1, jj + 1)), acc);
                acc = dot2s(F65, padd_x(P(ii + 1, jj), P(ii - 1, jj - 1)), acc);
                acc = dot2s(FA9, padd_x(P(ii, jj + 2), P(ii, jj - 3)), acc);
                acc = dot2s(FCB, P(ii, jj), acc);
                o[jj] = mad_lo_fh(P(ii, jj - 1), FCB, acc);
            }
#pragma unroll
            for (int jj = 0; jj < 4; jj += 2) {
                const uint32_t p0 = padd(P(ii + 3, jj), P(ii - 3, jj)), p1 = padd(P(ii + 2, jj + 1), P(ii - 2, jj - 1)), p4 = padd(P(ii + 1, jj + 2), P(ii - 1, jj - 2));
                o[jj] = mad_lo(p4, F4, mad_lo_fh(p1, F10, mad_lo(p0, F10, o[jj])));
                o[jj + 1] = mad_hi(p4, F4, mad_hi_fh(p1, F10, mad_hi(p0, F10, o[jj + 1])));
            }
            uint2 w;
            w.x = clip_pack(o[0], o[1], maxv); w.y = clip_pack(o[2], o[3], maxv);
            *(uint2 *)((char *)dy_ + (ol0 + ii * sl2)) = w;
            keep_luma(ii, w);
        }
#undef P
    } else {
#pragma unroll
        for (int ii = 0; ii < 4; ii++) {
            const uint2 w = FUSED ? keep_w[ii] : *(const uint2 *)(sy_ + (y + ii) * a.s_l + x);      // (fused: the deblocked samples only exist in LDS)
            *(uint2 *)((char *)dy_ + (ol0 + ii * sl2)) = w; keep_luma(ii, w);
        }
    }

    // ------------------------------------------------ chroma ---------------------------------------------
    const int cx = x >> 1, cy = y >> 1;
#pragma unroll
    for (int pl = 0; pl < 2; pl++) {
        const int16_t *src = pl ? sv_ : su_;
        int16_t *dst = pl ? dv_ : du_;
        auto keep_chroma = [&](int ii, uint32_t w) {
            if (keep_border) {
                if (cy + ii == 0) *(uint32_t *)&b_row[1 + pl][0][lx << 1] = w;
                if (cy + ii == (a.pic_h >> 1) - 1) *(uint32_t *)&b_row[1 + pl][1][lx << 1] = w;
                if (cx == 0) b_col[1 + pl][0][(ly << 1) + ii] = (int16_t)(w & 0xFFFF);
                if (cx + 2 == (a.pic_w >> 1)) b_col[1 + pl][1][(ly << 1) + ii] = (int16_t)(w >> 16);
            }
        };
        if (!a.enable[1 + pl]) {
#pragma unroll
            for (int ii = 0; ii < 2; ii++) {
                const uint32_t w = FUSED ? *(const uint32_t *)(l_c[pl] + ((ly << 1) + ii + 2) * CSTR + (lx << 1) + CO) : *(const uint32_t *)(src + (cy + ii) * a.s_c + cx);
                *(uint32_t *)((char *)dst + (oc0 + ii * sc2)) = w; keep_chroma(ii, w);
            }
            continue;
        }
        // window rows -2..3, cols -2..3 -> 3 dwords per row; col j at sample j+2
        uint32_t C[6][3];
#pragma unroll
        for (int i = 0; i < 6; i++) {
            const uint32_t *row = (const uint32_t *)(l_c[pl] + ((ly << 1) + i) * CSTR + (lx << 1) + CO - 2);
            C[i][0] = row[0]; C[i][1] = row[1]; C[i][2] = row[2];
        }
        // alf_filter_blk_5 (xevdm_alf.c:339-429), the same way: the SCU's two outputs of a row are one packed pair
#define PC(i, c) ((((c) + 2) & 1) ? __builtin_amdgcn_alignbit(C[(i) + 2][(((c) + 2) >> 1) + 1], C[(i) + 2][((c) + 2) >> 1], 16) : C[(i) + 2][((c) + 2) >> 1])
        const uint32_t G32 = a.cchroma[0], G54 = a.cchroma[1], G10 = a.cchroma[2], G6 = a.cchroma[3];      // the couples of the 5x5 diamond: taps (3,2) and (5,4); scalars of the kernel arguments
        const uint32_t G32v = G32;      // (one vector copy: the first sum's scalar operand is the rounding constant - an instruction takes one scalar register)
#pragma unroll
        for (int ii = 0; ii < 2; ii++) {
            int o[2];
#pragma unroll
            for (int jj = 0; jj < 2; jj++) {
                int acc = sdot2s(G54, padd_x(PC(ii, jj + 1), PC(ii, jj - 2)), dot2s_first(G32v, padd_x(PC(ii + 1, jj - 1), PC(ii - 1, jj))));
                o[jj] = smad_lo(PC(ii, jj), G6, acc);
            }
            const uint32_t p0 = padd(PC(ii + 2, 0), PC(ii - 2, 0)), p1 = padd(PC(ii + 1, 1), PC(ii - 1, -1));
            o[0] = smad_lo_fh(p1, G10, smad_lo(p0, G10, o[0]));
            o[1] = smad_hi_fh(p1, G10, smad_hi(p0, G10, o[1]));
            const uint32_t w = clip_pack(o[0], o[1], maxv);
            *(uint32_t *)((char *)dst + (oc0 + ii * sc2)) = w;
            keep_chroma(ii, w);
        }
#undef PC
    }
    } while (0);

    // ------------------------------------------------ border replication (xevd_picbuf_expand, src_base/xevd_util.c:365-427) ---------------
    // A tile on the picture border writes its share of the padding from the samples it has just produced: the margin left / right of its rows, the rows above /
    // below its columns, and - a corner tile - the corner block.  All 256 threads of the workgroup take part (the tile's own lanes outside the picture too).
    ATR(11);
    if (!a.pad || !(edge_l || edge_r || edge_t || edge_b)) return;
    __syncthreads();
#pragma unroll
    for (int c = 0; c < 3; c++) {
        const int sh = c ? 1 : 0, pad = c ? XGPU_PAD_C : XGPU_PAD_L, s = c ? a.s_c : a.s_l;
        const int pw = a.pic_w >> sh, ph = a.pic_h >> sh, x0 = tx0 >> sh, y0 = ty0 >> sh, T = 64 >> sh;
        const int cols = min(T, pw - x0), rows = min(T, ph - y0);        // the tile's part of the picture
        int16_t *pl = c == 0 ? dy_ : (c == 1 ? du_ : dv_);
        const int q = pad >> 2;                                          // margin width in 8-byte pieces (144 / 72 samples)
        // left / right of the tile's rows
#pragma unroll
        for (int side = 0; side < 2; side++) {
            if (side ? !edge_r : !edge_l) continue;
            for (int i = t; i < rows * q; i += 256) {
                const int r = i / q, k = i - r * q;
                const uint32_t v = (uint16_t)b_col[c][side][r] * 0x10001u;
                *(uint2 *)(pl + (y0 + r) * s + (side ? pw : -pad) + 4 * k) = make_uint2(v, v);
            }
        }
        // above / below the tile's columns, and the corner blocks next to them
#pragma unroll
        for (int side = 0; side < 2; side++) {
            if (side ? !edge_b : !edge_t) continue;
            int16_t *base = pl + (side ? ph : -pad) * s;
            const int cq = cols >> 2;
            for (int i = t; i < pad * cq; i += 256) {
                const int r = i / cq, k = i - r * cq;
                *(uint2 *)(base + r * s + x0 + 4 * k) = *(const uint2 *)&b_row[c][side][4 * k];
            }
#pragma unroll
            for (int cs = 0; cs < 2; cs++) {
                if (cs ? !edge_r : !edge_l) continue;
                const uint32_t v = (uint16_t)b_row[c][side][cs ? cols - 1 : 0] * 0x10001u;
                for (int i = t; i < pad * q; i += 256) {
                    const int r = i / q, k = i - r * q;
                    *(uint2 *)(base + r * s + (cs ? pw : -pad) + 4 * k) = make_uint2(v, v);
                }
            }
        }
    }
}

__global__ __launch_bounds__(256) void k_alf(const AlfArgs a, const int16_t *__restrict__ sy_, const int16_t *__restrict__ su_, const int16_t *__restrict__ sv_,
                                             int16_t *__restrict__ dy_, int16_t *__restrict__ du_, int16_t *__restrict__ dv_)
{
    alf_kernel<false>(a, nullptr, sy_, su_, sv_, dy_, du_, dv_);
}

// ADDB deblocking + ALF of a picture in one pass: SRC = the reconstruction, DST = the output picture
// (Round 3, measured and not kept: the NEXT picture's residual pass in this kernel's grid instead of the data-flow intra launch's - the pass's work items in groups of
//  eight spread evenly between the groups of tiles, its LDS laid over the tile's, 78 VGPRs and six workgroups per CU as before.  The idea: this kernel is bound by VALU issue
//  with 2 TB/s of traffic, the pass moves 107 MB with 19 us of VALU work.  Same box, ride in the intra launch / here: 8K 2679, 2718 / 2708, 2748 frames/s (+1 %),
//  4K 8155, 8165 / 7725, 7849 (-4.5 %).)
// PK: the deblocking line filters on packed pairs of lines (bit depths up to 10; XEVD_HIP_ADDB_SCALAR=1 keeps the scalar form for A/B runs)
template <bool PK>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(XGPU_ALF_WAVES, XGPU_ALF_WAVES))) void k_addb_alf(const AlfArgs a, const AddbArgs d, const int16_t *__restrict__ sy_, const int16_t *__restrict__ su_,
                                                  const int16_t *__restrict__ sv_, int16_t *__restrict__ dy_, int16_t *__restrict__ du_, int16_t *__restrict__ dv_)
{
    alf_kernel<true, PK>(a, &d, sy_, su_, sv_, dy_, du_, dv_);
}

void launch_alf(xgpu_ctx *c, const AlfArgs &a_, const AddbArgs *deblock, const DevPic &src, const DevPic &dst)
{
    AlfArgs a = a_;
    const int tiles = ((a.pic_w + 63) >> 6) * ((a.pic_h + 63) >> 6);
    a.magic_tiles_x = a.pic_w > 64 ? (uint32_t)((1ull << 32) / (uint64_t)((a.pic_w + 63) >> 6)) + 1u : 0u;      // floor(2^32 / 1) + 1 does not fit: 0 = one tile per row
    const dim3 grid(((tiles + 7) >> 3) << 3);
    const bool scalar_knob = c->addb_scalar != 0;
    static const int lds_pad = getenv("XEVD_HIP_ALF_LDSPAD") ? atoi(getenv("XEVD_HIP_ALF_LDSPAD")) : 0;      // measurement knob: dynamic LDS that only lowers the occupancy
    if (deblock && !scalar_knob && deblock->bd_l <= 10 && deblock->bd_c <= 10) hipLaunchKernelGGL(k_addb_alf<true>, grid, dim3(256), lds_pad, c->stream, a, *deblock, src.y, src.u, src.v, dst.y, dst.u, dst.v);
    else if (deblock) hipLaunchKernelGGL(k_addb_alf<false>, grid, dim3(256), 0, c->stream, a, *deblock, src.y, src.u, src.v, dst.y, dst.u, dst.v);
    else hipLaunchKernelGGL(k_alf, grid, dim3(256), 0, c->stream, a, src.y, src.u, src.v, dst.y, dst.u, dst.v);
}
