// evc_host.cc - MPEG-5 EVC Baseline-profile bitstream parser (-> CU batches for the GPU path) and writer (<- synthetic CU
// batches), see include/xevd_host.h.  Plain C++ (no HIP): it is the serial, bit-level host half of the decoder.
//
// Design (not the reference's): the CU-level syntax is written ONCE as a template over a "coder" that is either the
// arithmetic decoder or the arithmetic encoder - every syntax element is `v = c.bin(v, model)` - so the writer and the parser
// cannot drift apart; the reference decoder itself (oracle/_ref) is what pins them to the standard in tests.  A picture is
// parsed CU by CU in ONE pass (entropy decoding + motion/QP derivation + map update), straight into the structure-of-arrays
// batch that xgpu_batch_create takes; the reference's two passes over XEVD_CU_DATA (xevd_tile_eco, then xevd_ctu_row_rec_mt)
// see the same neighbour state because both walk the CUs in the same order and "reconstructed" (COD) == "already parsed".
#include "../../include/xevd_host.h"
#include "alf_fixed_tables.h"
#include "dmvr_search.h"
#include "cm_init_tables.h"
#include "../csrc/affine_model.h"

#include <algorithm>
#include <atomic>
#include <map>
#include <memory>
#include <thread>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

namespace {

// ------------------------------------------------------------------------------------------------ bit I/O
struct BitWriter {
    std::vector<uint8_t> buf;
    uint32_t acc = 0;
    int n = 0;
    void put1(int b) { acc = (acc << 1) | (uint32_t)(b & 1); if (++n == 8) { buf.push_back((uint8_t)acc); acc = 0; n = 0; } }
    void put(uint32_t v, int len) { for (int i = len - 1; i >= 0; i--) put1((int)((v >> i) & 1)); }
    void ue(uint32_t v) { const uint64_t x = (uint64_t)v + 1; int len = 0; while ((x >> len) > 1) len++; for (int i = 0; i < len; i++) put1(0); for (int i = len; i >= 0; i--) put1((int)((x >> i) & 1)); }
    void se(int v) { ue(v > 0 ? (uint32_t)(2 * v - 1) : (uint32_t)(-2 * v)); }      // xevd_bsr_read_se, xevd_bsr.c:322-328
    void align_zero() { while (n) put1(0); }
};
struct BitReader {
    const uint8_t *p = nullptr;
    size_t size = 0, pos = 0;       // pos in bits
    bool overrun = false;
    int get1() { if (pos >= size * 8) { overrun = true; return 0; } const int b = (p[pos >> 3] >> (7 - (pos & 7))) & 1; pos++; return b; }
    uint32_t get(int len) { uint32_t v = 0; for (int i = 0; i < len; i++) v = (v << 1) | (uint32_t)get1(); return v; }
    uint32_t ue() { int z = 0; while (!get1()) { if (++z > 32 || overrun) { overrun = true; return 0; } } uint64_t v = 1; for (int i = 0; i < z; i++) v = (v << 1) | (uint64_t)get1(); return (uint32_t)(v - 1); }
    int se() { const uint32_t k = ue(); return (k & 1) ? (int)((k + 1) >> 1) : -(int)(k >> 1); }
    bool aligned() const { return (pos & 7) == 0; }
};

// ------------------------------------------------------------------------------------------------ arithmetic coder
// Context model = (state << 1) | mps, state 9 bits, initial 512 = state 256 (p = 1/2) - xevd_eco.c:35-87, xevd_def.h:76
typedef uint16_t Model;
static inline void model_update(Model &m, bool lps)
{
    int state = m >> 1, mps = m & 1;
    if (lps) { state = state + ((512 - state + 16) >> 5); if (state > 256) { mps = 1 - mps; state = 512 - state; } }
    else state = state - ((state + 16) >> 5);
    m = (Model)((state << 1) | mps);
}
static inline uint32_t lps_range(Model m, uint32_t range) { const uint32_t l = ((uint32_t)(m >> 1) * range) >> 9; return l < 437 ? 437 : l; }

struct Dec {      // xevd_sbac_decode_bin / sbac_decode_bin_ep / xevd_sbac_decode_bin_trm, xevd_eco.c:35-165
    BitReader *br;
    uint32_t range, value;
    void start() { range = 16384; value = 0; for (int i = 0; i < 14; i++) value = ((value << 1) | (uint32_t)br->get1()) & 0xFFFF; }
    int bin(int, Model &m)
    {
        const int mps = m & 1;
        const uint32_t lps = lps_range(m, range);
        int b = mps;
        range -= lps;
        if (value >= range) { b = 1 - mps; value -= range; range = lps; model_update(m, true); }
        else model_update(m, false);
        while (range < 8192) { range <<= 1; value = ((value << 1) | (uint32_t)br->get1()) & 0xFFFF; }
        return b;
    }
    int ep(int)
    {
        int b = 0;
        range >>= 1;
        if (value >= range) { b = 1; value -= range; }
        range <<= 1;
        value = ((value << 1) | (uint32_t)br->get1()) & 0xFFFF;
        return b;
    }
    int tile_end()      // terminating bin, then zero bits up to the byte boundary and zero words up to the end (xevd_eco.c:100-140,1683-1695)
    {
        range--;
        if (value < range) return 0;
        while (!br->aligned()) if (br->get1()) return -1;
        return 1;
    }
};
struct Enc {      // the mirror image: MPS takes the lower part of the interval, carries resolved with outstanding bits
    BitWriter *bw;
    uint32_t low = 0, range = 16384;
    int outstanding = 0;
    bool first = true;
    void start() { low = 0; range = 16384; outstanding = 0; first = true; }
    void emit(int b) { if (first) first = false; else bw->put1(b); while (outstanding) { bw->put1(!b); outstanding--; } }
    void shift_out()
    {
        if (low < 8192) emit(0);
        else if (low >= 16384) { low -= 16384; emit(1); }
        else { low -= 8192; outstanding++; }
        low <<= 1;
    }
    int bin(int v, Model &m)
    {
        const int mps = m & 1;
        const uint32_t lps = lps_range(m, range);
        range -= lps;
        if ((v & 1) != mps) { low += range; range = lps; model_update(m, true); }
        else model_update(m, false);
        while (range < 8192) { shift_out(); range <<= 1; }
        return v & 1;
    }
    int ep(int v)
    {
        const uint32_t half = range >> 1;
        if (v & 1) low += half;
        shift_out();
        range = half << 1;
        return v & 1;
    }
    int tile_end()
    {
        range--;
        low += range;                               // the top unit of the interval: the decoder sees value >= range
        emit((int)((low >> 14) & 1));
        for (int i = 13; i >= 0; i--) bw->put1((int)((low >> i) & 1));
        bw->align_zero();
        return 1;
    }
};

// Symbol binarisations shared by both coders (xevd_eco.c:167-258, 452-489)
template <class C> static int sym_unary(C &c, int v, Model *m, int num_ctx)       // sbac_read_unary_sym
{
    int sym = 0, ctx = 0;
    if (!c.bin(v > 0, m[0])) return 0;
    for (;;) {
        if (ctx < num_ctx - 1) ctx++;
        sym++;
        if (!c.bin(v > sym, m[ctx])) break;
        if (sym >= 0x7FFF) break;                   // no valid symbol is longer (levels are s16, runs below 4096): malformed input ends here
    }
    return sym;
}
template <class C> static int sym_trunc_unary(C &c, int v, Model *m, int num_ctx, int max_num)      // sbac_read_truncate_unary_sym
{
    int i = 0;
    if (max_num > 1)
        for (; i < max_num - 1; ++i)
            if (!c.bin(v > i, m[i > num_ctx - 1 ? num_ctx - 1 : i])) break;
    return i;
}
template <class C> static int sym_abs_mvd(C &c, int v, Model &m)      // xevd_eco_abs_mvd: 1 = zero; else (len-1) zeros + 1, then len suffix bits
{
    if (c.bin(v == 0, m)) return 0;
    int len_v = 0;
    while (((v + 1) >> (len_v + 1)) > 0) len_v++;            // floor(log2(v + 1)) on the encoder side
    int len = 0, code;
    do { code = len == 0 ? c.bin(len + 1 == len_v, m) : c.ep(len + 1 == len_v); len++; } while (!code && len < 24);      // a valid |mvd| has at most 16 prefix bins
    int val = (1 << len) - 1;
    const int suffix = v + 1 - (1 << len_v);
    while (len != 0) { len--; val += c.ep((suffix >> len) & 1) << len; }
    return val;
}

template <class C> static int sym_unary_ep(C &c, int v, int max_val)                 // sbac_read_unary_sym_ep, xevd_eco.c:166-189
{
    if (!c.ep(v > 0)) return 0;
    int sym = 0, counter = 1, t;
    do { t = counter == max_val ? 0 : c.ep(v > sym + 1); counter++; sym++; } while (t);
    return sym;
}
template <class C> static int sym_bits_ep(C &c, int v, int n)                         // sbac_decode_bins_ep: most significant bin first
{
    int r = 0;
    for (int i = n - 1; i >= 0; i--) r = (r << 1) | c.ep((v >> i) & 1);
    return r;
}

struct Rpl { int n = 0, active = 0; int ref[XGPU_MAX_REFS + 4] = { 0 }; };      // XEVD_RPL: POC differences cur - ref of the list's pictures (the first `active` are indexed by refi)
// ref_pic_list_struct (xevdm_eco_rlp, src_main/xevdm_eco.c:1820-1844): entry count, then per entry the POC difference to the previous entry with a sign
static bool read_rpl(BitReader &br, Rpl &r)
{
    r = Rpl();
    r.n = (int)br.ue();
    if (br.overrun || r.n > XGPU_MAX_REFS) return false;
    int sign = 0;                                             // the reference keeps the last sign flag across entries whose difference is zero
    for (int i = 0; i < r.n; i++) {
        const int d = (int)br.ue();
        if (d != 0) sign = br.get1();
        r.ref[i] = (i ? r.ref[i - 1] : 0) + d * (1 - 2 * sign);
    }
    return !br.overrun;
}
static void write_rpl(BitWriter &bw, const Rpl &r)
{
    bw.ue((uint32_t)r.n);
    for (int i = 0; i < r.n; i++) {
        const int d = r.ref[i] - (i ? r.ref[i - 1] : 0);
        bw.ue((uint32_t)(d < 0 ? -d : d));
        if (d != 0) bw.put1(d < 0);
    }
}

struct Models {
    Model split[1], run[24], last[2], level[24], cbf_luma[1], cbf_cb[1], cbf_cr[1], cbf_all[1], pred_mode[3], direct[1], inter_dir[2],
          intra_dir[2], mvp_idx[3], mvd[1], refi[2], dqp[1], skip[2],
          ats_mode[1], ats_inter_flag[2], ats_inter_quad[1], ats_inter_hor[3], ats_inter_pos[1],      // Main: xevd_def.h:559-563
          alf_ctb[1],
          mmvd_flag[1], mmvd_merge_idx[3], mmvd_dist_idx[7], mmvd_dir_idx[2], mmvd_group_idx[2],           // tool_mmvd: xevd_def.h:478-482
          mvr_idx[4],                                                                                // tool_amvr: xevd_def.h:493
          merge_mode[1], merge_idx[5], bi_idx[2],                                                      // tool_admvp: xevd_def.h:461-465
          ibc_flag[2],                                                                               // sps->ibc_flag: xevd_def.h:485
          affine_flag[2], affine_mode[1], affine_mrg[5], affine_mvp_idx[1], affine_mvd_flag[2],       // tool_affine: xevd_def.h:483-498
          ipm_mpm_flag[1], ipm_mpm_idx[1], ipm_chroma[1],                                            // tool_eipd: xevd_def.h intra_luma_pred_mpm_flag / _idx, intra_chroma_pred_mode
          btt_split_flag[15], btt_split_dir[5], btt_split_type[1], mode_cons[3],                        // sps_btt_flag: xevd_def.h:486-491
          sig_coeff[47], gt_ab[18], last_x[21], last_y[21];                                          // tool_adcc: xevd_def.h sig_coeff_flag, coeff_abs_level_greaterAB_flag, last_sig_coeff_{x,y}_prefix
    void reset() { Model *p = (Model *)this; for (size_t i = 0; i < sizeof(Models) / sizeof(Model); i++) p[i] = 512; }     // PROB_INIT, xevd_eco.c:769-803
    // sps->tool_cm_init: every context starts from its initValue, the slice kind and the slice QP (xevd_eco_sbac_ctx_initialize, src_base/xevd_util.c:1243-1274;
    // the list of xevdm_eco_sbac_reset, src_main/xevdm_eco.c:1012-1065)
    template <int N> static void init(Model (&m)[N], const int16_t (&tbl)[2][N], int b_slice, int qp)
    {
        for (int i = 0; i < N; i++) {
            const int v = tbl[b_slice][i];
            int slope = (v & 14) << 4, offset = ((v >> 4) & 62) << 7;
            if (v & 1) slope = -slope;
            if ((v >> 4) & 1) offset = -offset;
            int state = std::min(std::max((slope * qp + offset + 4096) >> 4, 1), 511), mps = 1;
            if (state > 256) { state = 512 - state; mps = 0; }
            m[i] = (Model)((state << 1) + mps);
        }
    }
    void reset_cm(int b_slice, int qp)
    {
        qp = std::min(std::max(qp, 0), 51);
#define CM(name) init(name, k_cm_##name, b_slice, qp)
        CM(split); CM(run); CM(last); CM(level); CM(cbf_luma); CM(cbf_cb); CM(cbf_cr); CM(cbf_all); CM(pred_mode); CM(direct); CM(inter_dir); CM(intra_dir); CM(mvp_idx);
        CM(mvd); CM(refi); CM(dqp); CM(skip); CM(ats_mode); CM(ats_inter_flag); CM(ats_inter_quad); CM(ats_inter_hor); CM(ats_inter_pos); CM(alf_ctb); CM(mmvd_flag);
        CM(mmvd_merge_idx); CM(mmvd_dist_idx); CM(mmvd_dir_idx); CM(mmvd_group_idx); CM(mvr_idx); CM(merge_mode); CM(merge_idx); CM(bi_idx); CM(ibc_flag); CM(affine_flag);
        CM(affine_mode); CM(affine_mrg); CM(affine_mvp_idx); CM(affine_mvd_flag); CM(ipm_mpm_flag); CM(ipm_mpm_idx); CM(ipm_chroma); CM(btt_split_flag); CM(btt_split_dir); CM(btt_split_type); CM(mode_cons); CM(sig_coeff); CM(gt_ab); CM(last_x); CM(last_y);
#undef CM
    }
};

// ------------------------------------------------------------------------------------------------ constants of the standard
// most-probable-mode code numbers by (left mode + 1, upper mode + 1), 0 = not intra/available: xevd_tbl_mpm, xevd_tbl.c:46-54
static const uint8_t k_mpm[6][6][5] = {
    { { 0, 2, 3, 1, 4 }, { 0, 2, 1, 3, 4 }, { 0, 2, 1, 3, 4 }, { 1, 2, 0, 3, 4 }, { 0, 2, 1, 3, 4 }, { 0, 1, 2, 3, 4 } },
    { { 1, 0, 2, 3, 4 }, { 0, 1, 2, 3, 4 }, { 0, 1, 2, 3, 4 }, { 1, 2, 0, 3, 4 }, { 0, 1, 3, 2, 4 }, { 0, 2, 1, 4, 3 } },
    { { 1, 0, 2, 3, 4 }, { 1, 0, 2, 3, 4 }, { 1, 0, 2, 3, 4 }, { 2, 0, 1, 3, 4 }, { 1, 0, 3, 2, 4 }, { 0, 1, 2, 4, 3 } },
    { { 1, 0, 2, 3, 4 }, { 0, 2, 1, 3, 4 }, { 1, 0, 2, 3, 4 }, { 1, 2, 0, 3, 4 }, { 0, 1, 2, 3, 4 }, { 0, 2, 1, 4, 3 } },
    { { 0, 1, 2, 3, 4 }, { 0, 3, 2, 1, 4 }, { 1, 0, 2, 3, 4 }, { 1, 2, 0, 3, 4 }, { 1, 2, 3, 0, 4 }, { 0, 2, 1, 4, 3 } },
    { { 0, 1, 2, 3, 4 }, { 0, 1, 2, 4, 3 }, { 0, 1, 2, 4, 3 }, { 0, 2, 1, 4, 3 }, { 0, 1, 2, 3, 4 }, { 0, 1, 2, 4, 3 } } };
// default chroma QP mapping: xevd_tbl_qp_chroma_adjust_base, xevd_tbl.c:345-354
static const int8_t k_chroma_qp[58] = { 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16, 17, 18, 19, 20, 21, 22, 23, 24, 25, 26, 27, 28, 29,
    29, 29, 30, 31, 32, 32, 33, 33, 34, 34, 35, 35, 36, 36, 36, 37, 37, 37, 38, 38, 39, 39, 40, 40, 40, 41, 41, 41 };

// ... and with sps->tool_iqt: xevd_tbl_qp_chroma_adjust_main, xevd_tbl.c:334-342 (src_main/xevdm.c:471-479)
static const int8_t k_chroma_qp_main[58] = { 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16, 17, 18, 19, 20, 21, 22, 23, 24, 25, 26, 27, 28, 29,
    29, 30, 31, 32, 33, 34, 35, 36, 37, 37, 38, 39, 40, 40, 41, 42, 43, 44, 45, 46, 47, 48, 49, 50, 51, 52, 53, 54 };

// zig-zag scan of a w x h block (init_scan, xevd_util.c:1004-1047): anti-diagonals, odd ones top-right -> bottom-left
static void make_zigzag(std::vector<uint16_t> &scan, int w, int h)
{
    scan.resize((size_t)w * h);
    int pos = 0;
    scan[pos++] = 0;
    for (int l = 1; l < w + h - 1; l++) {
        if (l & 1) { int x = std::min(l, w - 1), y = std::max(0, l - (w - 1)); while (x >= 0 && y < h) { scan[pos++] = (uint16_t)(y * w + x); x--; y++; } }
        else       { int y = std::min(l, h - 1), x = std::max(0, l - (h - 1)); while (y >= 0 && x < w) { scan[pos++] = (uint16_t)(y * w + x); x++; y--; } }
    }
}

enum { MODE_INTRA = XGPU_MODE_INTRA, MODE_INTER = XGPU_MODE_INTER, MODE_SKIP = XGPU_MODE_SKIP, MODE_IBC = XGPU_MODE_IBC };

// ------------------------------------------------------------------------------------------------ stream / picture state
struct Sps { int width = 0, height = 0, bd_l = 8, bd_c = 8, log2_sub_gop = 0, log2_ref_gap = 0, max_num_ref_pics = 1;
             int profile_main = 0, tool_iqt = 0, tool_ats = 0, tool_addb = 0, tool_alf = 0, tool_eipd = 0, tool_dra = 0, tool_htdf = 0;
             int tool_mmvd = 0;                      // sps->tool_mmvd: merge with vector difference (a base candidate plus one of 32 offsets)
             int tool_dmvr = 0;                      // sps->tool_dmvr: merge-mode motion is refined by the backend (no syntax of its own)
             // ... and by the front end itself when refined vectors are decoder state inside the picture: the history buffer (tool_hmvp) and the merge
             // list of MMVD CUs (tool_mmvd) read them (dmvr_search.h)
             bool host_dmvr() const { return tool_dmvr && (tool_hmvp || tool_mmvd); }
             int tool_amvr = 0, tool_hmvp = 0;       // sub-tools of tool_admvp: adaptive vector resolution (mvr_idx), history-based candidates
             int tool_rpl = 0, tool_pocs = 0, poc_lsb_bits = 4;      // sps->tool_rpl: reference lists and marking from signalled RPLs; tool_pocs: POC from poc_lsb in the slice header
             int n_rpl[2] = { 0, 0 }; Rpl rpls[2][32];               // RPL candidates of the SPS (sps->rpls_l0 / rpls_l1)
             int btt = 0, log2_min_cb = 2, split_tbl[4][2] = { { 0, 0 }, { 0, 0 }, { 0, 0 }, { 0, 0 } };      // sps_btt_flag: binary / ternary splits; allowed long sides (min, max) per shape 1:1, 1:2, 1:4, TT
             int btt_raw[4] = { 0, 0, 0, 0 };         // the SPS fields behind split_tbl (min cb - 2, diff max 1:4, diff max TT, diff min TT - 2)
             int tool_cm_init = 0, tool_adcc = 0;     // sps->tool_cm_init: contexts start from tables (slice kind, QP) and several flags pick theirs from the neighbours; tool_adcc
             int dquant = 0;                         // sps->dquant_flag (Main): QP deltas per quantisation group of pps.cu_qp_delta_area instead of per coded CU
             int tool_affine = 0;                    // sps->tool_affine: affine merge / affine inter CUs (4- or 6-parameter models from 2 / 3 control points)
             int tool_admvp = 0;                     // sps->tool_admvp: merge / resolution-indexed predictors instead of the Baseline candidate lists, 8-tap MC tables
             int ibc = 0, ibc_log_max = 0;            // sps->ibc_flag, sps->ibc_log_max_size (log2 of the largest IBC CU; xevdm_eco.c:1890-1898)
             int crop[4] = { 0, 0, 0, 0 };            // picture_crop_left / right / top / bottom_offset (xevd_eco.c:1349-1357), as xevd_pull reports them
             bool cqt = false; int8_t cq[2][96] = { { 0 } }; };      // chroma QP mapping tables signalled in the SPS: [c][qp + 6*(bd_c-8)], qp = -6*(bd_c-8) .. 57
struct Pps { int rpl1_idx_present = 0, default_active[2] = { 1, 1 }; int constrained_intra = 0, cu_qp_delta = 0, qp_delta_area = 6, dra_on = 0, dra_aps_id = 0;      // qp_delta_area: log2 of the group's sample count (6 = 8x8)
             // tiles (xevdm_eco_pps, xevdm_eco.c:2019-2052): a grid of CTU columns x rows, uniform or with explicit sizes
             int tile_cols = 1, tile_rows = 1, tile_uniform = 1, across_tiles = 0, offset_bits = 1, id_bits = 1, arbitrary_slices = 0;
             int tile_col_w[XGPU_MAX_TILE_COLS] = { 0 }, tile_row_h[XGPU_MAX_TILE_ROWS] = { 0 }; };
struct Slice { int type = XHOST_SLICE_I, qp = 32, qp_u_offset = 0, qp_v_offset = 0, deblock = 1, alpha_off = 0, beta_off = 0;
               int alf_on = 0, aps_id_y = 0, aps_id_ch = 0, alf_chroma_idc = 0, alf_ctb_map = 0;
               int poc_lsb = 0; Rpl rpl[2];                                                // tool_pocs / tool_rpl (xevdm_eco.c:2658-2733)
               int mmvd_group = 0;                                                         // mmvd_group_enable_flag (tool_mmvd, xevdm_eco.c:2592-2599)
               int tmvp_assigned = 0, col_list = 0, col_src_list = 0, col_ref = 0; };      // temporal_mvp_asigned_flag + collocated_* (tool_admvp, xevdm_eco.c:2748-2760)

// ---- ALF parameter sets (XEVD_ALF_SLICE_PARAM / ac_alf_line_buf[32], src_main/xevdm_alf.c:587-698) ----
// zig-zag position of the coded coefficients inside the 13-tap (7x7 diamond) layout and Exp-Golomb order class of each coefficient
// (pattern_to_large_filter5/7, golombIdx5/7: constants of the EVC specification, src_main/xevdm_alf.h:165-194)
static const int k_alf_to_large[2][13] = { { 0, 0, 1, 0, 0, 2, 3, 4, 0, 0, 5, 6, 7 }, { 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13 } };
static const int k_alf_golomb_idx[2][13] = { { 0, 0, 1, 0, 0, 1, 0, 0, 0, 0, 0, 0, 0 }, { 0, 0, 1, 0, 0, 1, 2, 1, 0, 0, 1, 2, 0 } };
struct AlfAps {
    bool valid = false;
    int luma_present = 0, chroma_present = 0, type7 = 0, num_filters = 1, coef_delta_flag = 0, pred_mode_flag = 0;
    uint8_t delta_idx[25] = { 0 }, filter_coef_flag[25] = { 0 };
    int fixed_pattern = 0;                                        // alf_luma_fixed_filter_usage_pattern: 0 none, 1 every class, 2 per-class flags
    uint8_t fixed_usage[25] = { 0 }, fixed_idx[25] = { 0 };      // which classes start from a fixed filter, and which of the class's 16
    int16_t luma[25][13] = { { 0 } }, chroma[7] = { 0 };
};
// alfGolombDecode / its inverse (xevdm_eco.c:2154-2187): q zeros, a one, q + k suffix bits, sign bit (1 = positive) for non-zero values
static int alf_golomb_read(BitReader &br, int k, bool is_signed)
{
    int q = 0;
    while (!br.get1()) { if (++q > 24 || br.overrun) { br.overrun = true; return 0; } }
    int v = ((1 << q) - 1) << k;
    if (q + k > 0) v += (int)br.get(q + k);
    if (is_signed && v != 0) v = br.get1() ? v : -v;
    return v;
}
static void alf_golomb_write(BitWriter &bw, int v, int k, bool is_signed)
{
    const int a = v < 0 ? -v : v;
    int q = 0;
    while (a >= (((1 << (q + 1)) - 1) << k)) q++;
    for (int i = 0; i < q; i++) bw.put1(0);
    bw.put1(1);
    if (q + k > 0) bw.put((uint32_t)(a - (((1 << q) - 1) << k)), q + k);
    if (is_signed && a != 0) bw.put1(v > 0);
}
static int ilog2i(int v) { int l = 0; while ((v >> (l + 1)) > 0) l++; return l; }

struct RefPic {          // what a decoded picture leaves behind for later pictures (XEVD_PIC map_mv / list_poc, xevd_picman.c:213-221)
    int poc = 0, tid = 0;
    int list0_poc = 0;               // POC of reference 0 of ITS list 0 (pic->list_poc[0]); temporal direct mode scales by it
    std::vector<int16_t> mv0;        // [f_scu][2]: list-0 motion of every SCU (refp.map_mv[scup][REFP_0])
    std::vector<int16_t> mv;         // [f_scu][2][2] and
    std::vector<int8_t> refi;        // [f_scu][2]: both lists (tool_admvp's temporal candidates read them)
    int list_poc[16] = { 0 };        // pic->list_poc[]: POCs of ITS list-0 references (indexed by reference indices of EITHER list, xevdm_util.c:3760-3761)
    const int16_t *luma = nullptr;   // the decoded picture's luma samples on the host (sample (0, 0), >= 144 samples of replicated border), registered by the caller
    int luma_stride = 0;             //   when the front end refines vectors itself (xhost_parser_set_ref_luma; dmvr_search.h)
};

struct Cu {
    int x, y, log2w, log2h;
    int mode;                        // MODE_INTRA / MODE_INTER / MODE_SKIP / MODE_IBC (mv[0] = the block vector, whole samples)
    int direct;                      // B slices: temporal direct mode (inter_dir = PRED_DIR), no motion syntax
    int refi[2], mvp_idx[2];
    int16_t mvd[2][2], mv[2][2];
    int ipm, ipm_c, cbf[3], qp;       // ipm_c: chroma mode with tool_eipd (DM 0, BI 1, DC 2, HOR 3, VER 4)
    int ats;                         // bit 0 ats_intra_cu, bit 1 ats_intra_mode_v, bit 2 ats_intra_mode_h (layout of xgpu_cu_batch.ats)
    int ats_inter;                   // ats_inter_info: idx | pos << 4
    int mmvd, mmvd_idx;              // mmvd_flag; group << 7 | base candidate << 5 | distance << 2 | direction
    int dmvr;                        // tool_dmvr and a skip / merge-mode CU: mcore->dmvr_enable (xevdm.c:1272-1288)
    int only_inter;                  // mode constraint eOnlyInter of a local tree (sps_btt_flag with tool_admvp): no pred_mode_flag, no IBC
    int tree;                        // local dual tree (mode constraint eOnlyIntra below a split whose chroma blocks would get too small): 0 = luma + chroma,
                                     // 1 = luma only (TREE_L; intra or IBC), 2 = chroma only (TREE_C: the split node's chroma block, intra, after its luma CUs)
    int qp_code;                     // core->cu_qp_delta_code (sps->dquant_flag): 0 - , 1 a CU of at least a quantisation group, 2 a CU inside a group
    int affine;                      // mcore->affine_flag: 0 translational, 1 / 2 = 2 / 3 control points (4- / 6-parameter model)
    int16_t aff_mv[2][3][2];         // mcore->affine_mv[list][vertex][x/y]: top-left, top-right, bottom-left control-point vectors
    int aff_idx[2];                  // affine merge index ([0]) / affine predictor index per list
    int16_t aff_mvd[2][3][2];        // coded control-point differences of an affine inter CU
};

struct Picture {         // SCU maps of the picture being parsed / written (ctx->map_scu, map_ipm, map_mv, map_refi; cod_eco)
    int w_scu = 0, h_scu = 0;
    std::vector<uint8_t> cod, intra, ibc;      // ibc: MCU_GET_IBC
    std::vector<uint8_t> cu_size;    // log2w | log2h << 4 of the CU over the SCU (map_cu_mode; only kept for the split-flag contexts: sps_btt_flag with tool_cm_init)
    std::vector<uint8_t> skip;       // MCU_GET_SF (only kept with sps->tool_cm_init: the skip flag's context counts skipped neighbours)
    std::vector<uint8_t> tidx;       // ctx->map_tidx: the tile of every SCU (empty: one tile) - neighbours in another tile are not available
    bool same_tile(int a, int b) const { return tidx.empty() || tidx[(size_t)a] == tidx[(size_t)b]; }
    std::vector<uint8_t> aff;        // sps->tool_affine: 0, or affine_flag | log2w << 2 | log2h << 5 of the affine CU the SCU belongs to (MCU_GET_AFF + map_affine)
    std::vector<uint32_t> aff_tl;    //   ... and the SCU address of that CU's top-left corner (MCU_GET_AFF_XOFF / _YOFF)
    std::vector<int8_t> ipm;
    std::vector<int16_t> mv;         // [f_scu][2][2]: the CUs' own vectors (mctx->map_unrefined_mv)
    std::vector<int16_t> mv_ref;     // host-side DMVR (Sps::host_dmvr): ctx->map_mv - the refined vectors of refined sub-blocks, the CU's own elsewhere; else empty
    std::vector<int8_t> refi;        // [f_scu][2]
    void reset(int w, int h, bool refined_map = false)
    {
        w_scu = w >> 2; h_scu = h >> 2;
        const size_t f = (size_t)w_scu * h_scu;
        if (refined_map) mv_ref.assign(f * 4, 0); else mv_ref.clear();
        cod.assign(f, 0); intra.assign(f, 0); ibc.assign(f, 0); ipm.assign(f, 0); mv.assign(f * 4, 0); refi.assign(f * 2, -1); tidx.clear(); aff.clear(); skip.clear(); cu_size.clear();
    }
};

struct Batch {           // the xgpu_cu_batch under construction
    std::vector<uint16_t> x, y;
    std::vector<uint8_t> log2w, log2h, pred_mode, qp, cbf, ipm, ats, ats_inter, dmvr, affine, tree;
    bool has_tree = false;
    std::vector<int8_t> refi;
    std::vector<int16_t> mv, coef, affine_mv;
    std::vector<uint32_t> coef_off, ctu_start;
    void clear() { x.clear(); y.clear(); log2w.clear(); log2h.clear(); pred_mode.clear(); qp.clear(); cbf.clear(); ipm.clear(); ats.clear(); ats_inter.clear(); dmvr.clear(); affine.clear(); tree.clear(); has_tree = false; affine_mv.clear(); refi.clear(); mv.clear(); coef.clear(); coef_off.clear(); ctu_start.clear(); }
};

// ---- DRA parameter sets (APS type 1, SIG_PARAM_DRA) and the inverse-mapping tables the output stage applies (src_main/xevdm_dra.c) ----
struct DraAps { bool valid = false; int num_ranges = 0, in_ranges[33] = { 0 }, scale[32] = { 0 }, cb_scale = 0, cr_scale = 0, table_idx = 0; };
// approximations of log / exp at 9 fractional bits used by the chroma scale correction (constants of the specification, src_main/xevdm_tbl.c:410-421)
static const int k_dra_log_tbl[55] = { 0, 1, 1, 1, 1, 1, 2, 2, 3, 4, 4, 6, 7, 9, 11, 14, 18, 23, 29, 36, 45, 57, 72, 91, 114, 144, 181, 228, 287, 362, 456, 575, 724, 912, 1149,
                                       1448, 1825, 2299, 2896, 3649, 4598, 5793, 7298, 9195, 11585, 14596, 18390, 23170, 29193, 36781, 46341, 58386, 73562, 92682, 116772 };
static const int k_dra_exp_tbl[25] = { 128, 144, 161, 181, 203, 228, 256, 287, 322, 362, 406, 456, 512, 574, 645, 724, 812, 912, 1024, 1149, 1290, 1448, 1625, 1825, 2048 };
static int dra_range_idx(int sample, const int *ranges, int n)           // xevd_get_dra_range_idx_gen: first i with sample < ranges[i + 1], else n - 1
{
    for (int i = 0; i < n - 1; i++) if (sample < ranges[i + 1]) return i;
    return n - 1;
}
// luts: [3][1024] = luma_inv_scale_lut, int_chroma_inv_scale_lut[Cb], [Cr] of DRA_CONTROL after xevd_init_dra (xevdm_dra.c:39-270).
// cq[c] + off = the sequence's chroma QP mapping (xevd_qp_chroma_dynamic[c]), indexable from -off.
static void dra_build_luts(const DraAps &a, int bd, const int8_t *cq_u, const int8_t *cq_v, int off, int32_t *luts)
{
    const int n = a.num_ranges;
    int out[34] = { 0 }, inv_scale[32], inv_off[32], cinv[2][32];
    for (int i = 1; i <= n; i++) out[i] = out[i - 1] + (a.in_ranges[i] - a.in_ranges[i - 1]) * a.scale[i - 1];       // xevd_construct_dra
    for (int i = 0; i < n; i++) {
        const int sc = a.scale[i] ? a.scale[i] : 1;
        inv_scale[i] = ((1 << 18) + (sc >> 1)) / sc;
        inv_off[i] = (int)((((int64_t)a.in_ranges[i + 1] << 18) - (int64_t)out[i + 1] * inv_scale[i] + (1 << 8)) >> 9);
    }
    for (int i = 0; i <= n; i++) out[i] = (out[i] + (1 << 8)) >> 9;
    auto scaled_qp = [&](int ch, int qp) { qp = std::min(std::max(qp, -off), 57); return (int)(ch == 1 ? cq_u : cq_v)[qp]; };   // xevd_get_scaled_chroma_qp2
    for (int i = 0; i < n; i++)
        for (int ch = 1; ch <= 2; ch++) {                            // xevd_correct_local_chroma_scale (:83-163)
            const int base = ch == 1 ? a.cb_scale : a.cr_scale;
            int cs;
            if (a.table_idx == 58) cs = base;
            else {
                const int scale_dra = base * a.scale[i];
                const int shift1 = a.table_idx - scaled_qp(ch, a.table_idx);
                const int s9 = (scale_dra + (1 << 8)) >> 9;
                const int idx = dra_range_idx(s9, k_dra_log_tbl, 54);
                const int num = s9 - k_dra_log_tbl[idx], den = k_dra_log_tbl[idx + 1] - k_dra_log_tbl[idx];
                int qp_int = 2 * idx - 60, qp_frac = 0;
                if (num == 0) qp_int -= 1;
                else { qp_frac = 512 * (num << 1) / den; qp_int += qp_frac / 512; qp_frac = 512 - (qp_frac % 512); }
                const int local_qp = a.table_idx - qp_int;
                const int q0 = scaled_qp(ch, std::min(std::max(local_qp, -off), 57)), q1 = scaled_qp(ch, std::min(std::max(local_qp + 1, -off), 57));
                const int dec = (q1 - q0) * qp_frac;
                int frac_adj = qp_frac - dec % 512;
                int shift = (local_qp - q0 - (dec >> 9)) - shift1;
                if (frac_adj < 0) { shift -= 1; frac_adj += 512; }
                const int sc = std::min(std::max(shift, -12), 12);
                const int e0 = k_dra_exp_tbl[sc + 12];
                const int de = shift >= 0 ? k_dra_exp_tbl[std::min(std::max(shift + 1, -12), 12) + 12] - e0 : e0 - k_dra_exp_tbl[std::min(std::max(shift - 1, -12), 12) + 12];
                const int out_scale = e0 + ((de * frac_adj + (1 << 8)) >> 9);
                cs = (int)(((int64_t)scale_dra * out_scale + (1 << 17)) >> 18);
            }
            if (cs == 0) cs = 1;
            cinv[ch - 1][i] = ((1 << 18) + (cs >> 1)) / cs;          // xevd_compensate_chroma_shift_table
        }
    for (int v = 0; v < 1024; v++) {                                 // xevd_build_dra_luma_lut
        const int r = dra_range_idx(v, out, n);
        luts[v] = std::min(std::max((inv_off[r] + v * inv_scale[r] + (1 << 8)) >> 9, 0), 1023);
    }
    for (int ch = 0; ch < 2; ch++) {                                 // xevd_build_dra_chroma_lut
        int r2[35] = { 0 }, msc[34], mof[34];
        r2[0] = out[0];
        for (int i = 1; i <= n; i++) r2[i] = (out[i - 1] + out[i]) / 2;
        msc[0] = 0; mof[0] = cinv[ch][0];
        for (int i = 1; i < n; i++) {
            const int delta = r2[i + 1] - r2[i];
            mof[i] = cinv[ch][i - 1];
            msc[i] = delta ? (((cinv[ch][i] - mof[i]) << bd) + (delta >> 1)) / delta : 0;
        }
        msc[n] = 0; mof[n] = cinv[ch][n - 1];
        for (int v = 0; v < 1024; v++) {
            int r = n;                                               // (the reference scans n + 1 ranges; past the last pivot the index is n)
            for (int i = 0; i < n; i++) if (v < r2[i + 1]) { r = i; break; }
            luts[(1 + ch) * 1024 + v] = mof[r] + ((msc[r] * (v - r2[r]) + (1 << (bd - 1))) >> bd);
        }
    }
}

struct Stream {          // everything both directions share
    Sps sps;
    Pps pps;
    Slice sh;
    Picture pic;
    std::vector<RefPic> dpb;         // reference pictures in coding order (pm->pic[] restricted to IS_REF)
    std::vector<const RefPic *> refp[2];
    int poc = 0, prev_poc = 0, prev_doc_offset = -1, tid = 0, last_intra_poc = 0, stale_list0_poc = 0;
    int stale_list_poc[16] = { 0 };      // list_poc[] entries past a picture's own list keep what earlier pictures wrote
    bool have_sps = false, have_pps = false, need_idr = false;
    std::vector<uint16_t> scan[6][6];      // zig-zag tables by log2 size - 1
    std::vector<uint16_t> scan_inv[6][6];  // raster position -> scan position (tool_adcc)
    AlfAps alf_aps[32];
    DraAps dra_aps[32];
    int32_t dra_luts[3 * 1024];            // of the current picture (when the PPS switches DRA on)
    std::vector<uint8_t> alf_ctb_flag;     // luma CTB flags: all on at the start of a picture (xevdm.c:3001-3005), coded ones overwrite (:2411-2418)
    int16_t alf_luma_final[25][13], alf_chroma_final[7];
    xgpu_tile_grid grid;                   // tiles of the current picture (set_tile_info, src_main/xevdm.c:2162-2330)

    // tile grid of the current picture from the PPS, and the SCU -> tile map (after pic.reset); false: the PPS does not fit the picture
    bool setup_tiles()
    {
        const int w_ctu = (sps.width + 63) >> 6, h_ctu = (sps.height + 63) >> 6;
        memset(&grid, 0, sizeof(grid));
        grid.n_cols = pps.tile_cols; grid.n_rows = pps.tile_rows; grid.loop_filter_across_tiles = pps.across_tiles;
        if (grid.n_cols > w_ctu || grid.n_rows > h_ctu) return false;
        for (int i = 0; i < grid.n_cols; i++) {
            const int wd = pps.tile_uniform ? ((i + 1) * w_ctu) / grid.n_cols - (i * w_ctu) / grid.n_cols : i + 1 < grid.n_cols ? pps.tile_col_w[i] : w_ctu - grid.col_bd[i];
            if (wd < 1) return false;
            grid.col_bd[i + 1] = grid.col_bd[i] + wd;
        }
        for (int j = 0; j < grid.n_rows; j++) {
            const int ht = pps.tile_uniform ? ((j + 1) * h_ctu) / grid.n_rows - (j * h_ctu) / grid.n_rows : j + 1 < grid.n_rows ? pps.tile_row_h[j] : h_ctu - grid.row_bd[j];
            if (ht < 1) return false;
            grid.row_bd[j + 1] = grid.row_bd[j] + ht;
        }
        if (grid.col_bd[grid.n_cols] != w_ctu || grid.row_bd[grid.n_rows] != h_ctu) return false;
        if (sps.tool_affine) { pic.aff.assign((size_t)pic.w_scu * pic.h_scu, 0); pic.aff_tl.resize((size_t)pic.w_scu * pic.h_scu); }
        if (sps.tool_cm_init) pic.skip.assign((size_t)pic.w_scu * pic.h_scu, 0);
        if (sps.tool_cm_init && sps.btt) pic.cu_size.assign((size_t)pic.w_scu * pic.h_scu, 0);
        pic.tidx.clear();
        if (grid.n_cols * grid.n_rows > 1) {
            pic.tidx.assign((size_t)pic.w_scu * pic.h_scu, 0);
            for (int j = 0; j < grid.n_rows; j++) for (int i = 0; i < grid.n_cols; i++)
                for (int y = grid.row_bd[j] * 16; y < std::min(grid.row_bd[j + 1] * 16, pic.h_scu); y++)
                    memset(&pic.tidx[(size_t)y * pic.w_scu + grid.col_bd[i] * 16], j * grid.n_cols + i, (size_t)(std::min(grid.col_bd[i + 1] * 16, pic.w_scu) - grid.col_bd[i] * 16));
        }
        return true;
    }

    // what alf_process hands to the filter (alf_load_paramline_from_aps_buffer2 + alf_recon_coef, xevdm_alf.c:682-794)
    bool alf_finalise()
    {
        const AlfAps &y = alf_aps[sh.aps_id_y & 31];
        if (!y.valid || !y.luma_present) return false;
        int16_t coef[25][13];
        memcpy(coef, y.luma, sizeof(coef));
        const int ncm1 = y.type7 ? 12 : 6;
        if (y.pred_mode_flag)
            for (int i = 1; i < y.num_filters; i++) for (int j = 0; j < ncm1; j++) coef[i][j] = (int16_t)(coef[i][j] + coef[i - 1][j]);
        for (int c = 0; c < 25; c++) {
            int sum = 0;
            for (int i = 0; i < 12; i++) {
                const int pos = k_alf_to_large[y.type7][i];
                // the class's fixed filter (one of the 16 its class may use), plus the coded coefficient (alf_recon_coef, xevdm_alf.c:724-752)
                const int fixed = y.fixed_usage[c] ? k_alf_fixed_coef[k_alf_class_to_fixed[c][y.fixed_idx[c] & 15]][i] : 0;
                alf_luma_final[c][i] = (int16_t)(fixed + (pos > 0 ? coef[y.delta_idx[c]][pos - 1] : 0));
                sum += alf_luma_final[c][i] * 2;
            }
            alf_luma_final[c][12] = (int16_t)(512 - sum);
        }
        memset(alf_chroma_final, 0, sizeof(alf_chroma_final));
        if (sh.alf_chroma_idc) {
            const AlfAps &ch = alf_aps[sh.aps_id_ch & 31];
            if (!ch.valid || !ch.chroma_present) return false;
            int sum = 0;
            for (int i = 0; i < 6; i++) { alf_chroma_final[i] = ch.chroma[i]; sum += ch.chroma[i] * 2; }
            alf_chroma_final[6] = (int16_t)(512 - sum);
        }
        return true;
    }
    // APS payload after aps_id / aps_type (xevdm_eco_alf_aps_param + xevdm_eco_alf_filter), reading or writing
    template <bool WR> bool alf_aps_syntax(BitReader *br, BitWriter *bw, AlfAps &a)
    {
        auto bit = [&](int v) -> int { if (WR) { bw->put1(v); return v & 1; } return br->get1(); };
        auto ue = [&](int v) -> int { if (WR) { bw->ue((uint32_t)v); return v; } return (int)br->ue(); };
        auto gol = [&](int v, int k, bool sg) -> int { if (WR) { alf_golomb_write(*bw, v, k, sg); return v; } return alf_golomb_read(*br, k, sg); };
        a.luma_present = bit(a.luma_present);
        a.chroma_present = bit(a.chroma_present);
        for (int pass = 0; pass < 2; pass++) {
            const bool chroma = pass == 1;
            if (chroma ? !a.chroma_present : !a.luma_present) continue;
            int type7 = 0;
            if (!chroma) {
                a.num_filters = ue(a.num_filters - 1) + 1;
                if (a.num_filters < 1 || a.num_filters > 25) return false;
                a.type7 = bit(a.type7);
                if (a.num_filters > 1) {
                    const int nb = ilog2i(a.num_filters - 1) + 1;
                    for (int c = 0; c < 25; c++) {
                        if (WR) bw->put(a.delta_idx[c], nb); else a.delta_idx[c] = (uint8_t)br->get(nb);
                        if (a.delta_idx[c] >= a.num_filters) return false;
                    }
                } else memset(a.delta_idx, 0, sizeof(a.delta_idx));
                // fixed filter sets (xevdm_eco.c:2436-2466): pattern, per-class usage flags with pattern 2, a 4-bit set index per using class
                a.fixed_pattern = gol(a.fixed_pattern, 0, false);
                if (a.fixed_pattern < 0 || a.fixed_pattern > 2) return false;
                for (int c = 0; c < 25; c++) a.fixed_usage[c] = (uint8_t)(a.fixed_pattern == 2 ? bit(a.fixed_usage[c]) : a.fixed_pattern == 1);
                for (int c = 0; c < 25; c++) {
                    if (!a.fixed_usage[c]) { a.fixed_idx[c] = 0; continue; }
                    if (WR) bw->put(a.fixed_idx[c] & 15, 4); else a.fixed_idx[c] = (uint8_t)br->get(4);
                }
                a.coef_delta_flag = bit(a.coef_delta_flag);
                a.pred_mode_flag = (!a.coef_delta_flag && a.num_filters > 1) ? bit(a.pred_mode_flag) : 0;
                type7 = a.type7;
            }
            int kmin = ue(0 + (WR ? alf_kmin_minus1 : 0)) + 1, ktab[3];
            if (kmin > 7) return false;
            for (int i = 0; i < (type7 ? 3 : 2); i++) { ktab[i] = kmin + bit(0); kmin = ktab[i]; }
            const int nf = chroma ? 1 : a.num_filters, nc = type7 ? 12 : 6;
            if (!chroma) {
                if (a.coef_delta_flag) for (int f = 0; f < nf; f++) a.filter_coef_flag[f] = (uint8_t)bit(a.filter_coef_flag[f]);
                else memset(a.filter_coef_flag, 1, sizeof(a.filter_coef_flag));
            }
            for (int f = 0; f < nf; f++) {
                int16_t *dst = chroma ? a.chroma : a.luma[f];
                if (!chroma && !a.filter_coef_flag[f]) { memset(dst, 0, sizeof(int16_t) * 13); continue; }
                for (int i = 0; i < nc; i++) dst[i] = (int16_t)gol(dst[i], ktab[k_alf_golomb_idx[type7][i]], true);
            }
        }
        return WR || !br->overrun;
    }
    int alf_kmin_minus1 = 0;

    Stream()
    {
        for (int a = 0; a < 6; a++) for (int b = 0; b < 6; b++) {
            make_zigzag(scan[a][b], 2 << a, 2 << b);
            scan_inv[a][b].resize(scan[a][b].size());
            for (size_t i = 0; i < scan[a][b].size(); i++) scan_inv[a][b][scan[a][b][i]] = (uint16_t)i;
        }
    }

    // POC of the next picture (xevd.c:1846-1861, xevd_poc_derivation xevd_util.c:429-467)
    void derive_poc(bool idr, int t)
    {
        tid = t;
        if (sps.tool_pocs && !enc_side) {
            // POC from the slice header's poc_lsb (xevdm.c:3044-3074): msb from the previous temporal-layer-0 picture (an IDR picture does not reset that one)
            if (idr) { poc = 0; return; }
            const int max_lsb = 1 << sps.poc_lsb_bits, prev_lsb = prev_poc & (max_lsb - 1), prev_msb = prev_poc - prev_lsb;
            int msb = prev_msb;
            if (sh.poc_lsb < prev_lsb && prev_lsb - sh.poc_lsb >= max_lsb / 2) msb = prev_msb + max_lsb;
            else if (sh.poc_lsb > prev_lsb && sh.poc_lsb - prev_lsb > max_lsb / 2) msb = prev_msb - max_lsb;
            poc = msb + sh.poc_lsb;
            if (t == 0) prev_poc = poc;
            return;
        }
        if (idr) { poc = 0; prev_doc_offset = -1; prev_poc = 0; return; }
        const int sub = 1 << sps.log2_sub_gop;
        if (t == 0) { poc = prev_poc + sub; prev_doc_offset = 0; prev_poc = poc; return; }
        auto ilog2 = [](int v) { int l = 0; while ((v >> (l + 1)) > 0) l++; return l; };
        int doc = (prev_doc_offset + 1) % sub, expected = 0;
        if (doc == 0) prev_poc += sub; else expected = 1 + ilog2(doc);
        for (int guard = 0; t != expected && guard < 4 * sub; guard++) { doc = (doc + 1) % sub; expected = doc == 0 ? 0 : 1 + ilog2(doc); }
        poc = prev_poc + (int)(sub * ((2.0 * doc + 1) / (double)(1 << t) - 2));
        prev_doc_offset = doc;
    }
    bool enc_side = false;               // the writer: POCs, marking and lists by the sub-GOP scheme (it then DESCRIBES them with poc_lsb / RPLs when those tools are on)
    bool is_ref_picture() const { return (sps.tool_pocs && !enc_side) || tid == 0 || tid < sps.log2_sub_gop; }      // ctx->slice_ref_flag, xevd.c:1853 (tool_pocs: every picture, xevdm.c:3076)
    std::vector<int> rpl_released;       // POCs the current slice's RPLs dropped from the DPB (reported with the picture)

    // reference lists without RPL (xevd_picman_refp_init, xevd_picman.c:291-437) over the reference pictures by descending POC
    // later_slice: a further slice of the picture being parsed - what the earlier slices' marking released is kept, and an I slice leaves the lists alone
    // (xevdm_picman_refp_rpl_based_init / xevdm_picman_refp_init return before touching refp for SLICE_I: the lists of an earlier P / B slice stay)
    bool build_ref_lists(bool idr, bool later_slice = false)
    {
        const bool keep_lists = later_slice && sh.type == XHOST_SLICE_I;
        if (!keep_lists) { refp[0].clear(); refp[1].clear(); }
        if (!later_slice) rpl_released.clear();
        if (sps.tool_rpl && !enc_side) {
            // marking (xevdm_picman_refpic_marking, xevdm_picman.c:542-588): a reference picture that neither list of THIS slice names (active or not) is
            // dropped; lists (xevdm_picman_refp_rpl_based_init :315-368): entry i = the picture with POC cur - ref[i], which must be there
            if (!idr)
                for (size_t i = 0; i < dpb.size();) {
                    bool named = false;
                    for (int l = 0; l < 2 && !named; l++) for (int j = 0; j < sh.rpl[l].n && !named; j++) named = dpb[i].poc == poc - sh.rpl[l].ref[j];
                    if (named) i++; else { rpl_released.push_back(dpb[i].poc); dpb.erase(dpb.begin() + (long)i); }
                }
            if (sh.type == XHOST_SLICE_I) return true;
            for (int l = 0; l < (sh.type == XHOST_SLICE_B ? 2 : 1); l++)
                for (int i = 0; i < sh.rpl[l].active; i++) {
                    if (i >= sh.rpl[l].n || i >= XGPU_MAX_REFS) return false;
                    const RefPic *hit = nullptr;
                    for (const RefPic &r : dpb) if (r.poc == poc - sh.rpl[l].ref[i]) { hit = &r; break; }
                    // a reference with the current picture's own POC (an RPL delta of 0, or tool_pocs repeating a poc_lsb) would put a zero POC distance
                    // into every scaling of mmvd_motion / the temporal candidates: not a stream a conformant encoder writes - refused, not divided by
                    if (!hit || hit->poc == poc) return false;
                    refp[l].push_back(hit);
                }
            return true;
        }
        if (sh.type == XHOST_SLICE_I) return true;
        std::vector<const RefPic *> ref;
        for (const RefPic &r : dpb) ref.push_back(&r);
        std::stable_sort(ref.begin(), ref.end(), [](const RefPic *a, const RefPic *b) { return a->poc > b->poc; });
        const int maxn = sps.max_num_ref_pics, n = (int)ref.size();
        auto old = [&](const RefPic *r) { return poc >= last_intra_poc && r->poc < last_intra_poc; };
        if (sh.type == XHOST_SLICE_P) {
            for (int i = 0; i < n && (int)refp[0].size() < maxn; i++) {
                const RefPic *r = ref[i];
                if (tid > 0) {
                    if (tid == 1) { if (r->poc < poc && r->tid <= tid) refp[0].push_back(r); }
                    else if (r->poc < poc && refp[0].empty()) refp[0].push_back(r);
                    else if (!refp[0].empty() && r->poc < poc && r->tid <= 1) refp[0].push_back(r);
                } else {
                    if (old(r)) continue;
                    if (r->poc < poc) refp[0].push_back(r);
                }
            }
            return true;
        }
        // B: nearest pictures first, each step allowed one temporal layer further down than the picture just taken
        for (int l = 0; l < 2; l++) {
            for (int pass = 0; pass < 2; pass++) {
                int next_layer = std::max(tid - 1, 0);
                const bool backward = (l == 0) == (pass == 0);             // list 0: earlier pictures first; list 1: later pictures first
                for (int k = 0; k < n && (int)refp[l].size() < maxn; k++) {
                    const RefPic *r = backward ? ref[k] : ref[n - 1 - k];
                    if (old(r)) continue;
                    if ((backward ? r->poc < poc : r->poc > poc) && r->tid <= next_layer) { refp[l].push_back(r); next_layer = std::max(r->tid - 1, 0); }
                }
            }
        }
        return true;
    }
    // picture marking + insertion (xevd_picman_put_pic / pic_marking_no_rpl, xevd_picman.c:68-110,462-509); released POCs reported
    void store_picture(bool idr, std::vector<int> &released, int any_inter_slice = -1)
    {
        // pic->list_poc[i] = POC of refp[i][REFP_0] (xevd_picman.c:213-221); an I slice leaves num_refp untouched, so the previous picture's values stay.
        // Taken BEFORE the DPB below is edited: refp[] points into it.  (Several slices: the lists of the picture's P / B slices, whichever slice came last.)
        if (any_inter_slice < 0 ? sh.type != XHOST_SLICE_I : any_inter_slice != 0) {
            stale_list0_poc = refp[0].empty() ? 0 : refp[0][0]->poc;
            for (size_t i = 0; i < refp[0].size() && i < 16; i++) stale_list_poc[i] = refp[0][i]->poc;
        }
        released.insert(released.end(), rpl_released.begin(), rpl_released.end());
        rpl_released.clear();
        if (idr) { for (const RefPic &r : dpb) released.push_back(r.poc); dpb.clear(); }
        else if (tid == 0 && (!sps.tool_rpl || enc_side)) {            // sliding-window marking only without RPLs (xevdm_picman_put_pic, xevdm_picman.c:595-606)
            const int gap = 1 << sps.log2_ref_gap;
            for (size_t i = 0; i < dpb.size();) {
                if (dpb[i].tid > 0 || (i > 0 && gap > 0 && dpb[i].poc % gap != 0)) { released.push_back(dpb[i].poc); dpb.erase(dpb.begin() + (long)i); }
                else i++;
            }
            while (dpb.size() >= 5) { released.push_back(dpb[0].poc); dpb.erase(dpb.begin()); }      // XEVD_MAX_NUM_ACTIVE_REF_FRAME
        }
        if (!is_ref_picture()) return;
        // bound on a damaged stream that keeps sending tid > 0 reference pictures without a tid-0 picture between them
        while (dpb.size() >= 32) { released.push_back(dpb[0].poc); dpb.erase(dpb.begin()); }
        RefPic r;
        r.poc = poc; r.tid = tid; r.list0_poc = stale_list0_poc;
        const size_t f = (size_t)pic.w_scu * pic.h_scu;
        const std::vector<int16_t> &kept = pic.mv_ref.empty() ? pic.mv : pic.mv_ref;      // ctx->map_mv: with host-side DMVR the refined vectors
        r.mv0.resize(f * 2);
        for (size_t k = 0; k < f; k++) { r.mv0[k * 2] = kept[k * 4]; r.mv0[k * 2 + 1] = kept[k * 4 + 1]; }
        if (sps.tool_admvp) { r.mv = kept; r.refi = pic.refi; memcpy(r.list_poc, stale_list_poc, sizeof(r.list_poc)); }
        dpb.push_back(std::move(r));
    }

};

// The part of the front end that works inside ONE tile: CU syntax in both directions, motion derivation, the SCU maps of the tile's CUs.
// Everything a tile changes while it is coded lives here (context models, QP predictor, motion history) or in the tile's own region of the
// picture maps, so the tiles of a picture can be parsed by different threads (xevdm_dec_slice hands tiles to its thread pool the same way,
// src_main/xevdm.c:2640-2690).  The references name the state of the Stream the coder belongs to.
struct TileCoder {
    const Sps &sps;
    const Pps &pps;
    const Slice &sh;
    Picture &pic;
    const std::vector<const RefPic *> (&refp)[2];
    const int &poc;
    const std::vector<uint16_t> (&scan)[6][6], (&scan_inv)[6][6];
    Models models;
    int qp_prev = 0;
    int qp_coded = 0;                // core->cu_qp_delta_is_coded: the current quantisation group has sent its delta
    explicit TileCoder(Stream &s) : sps(s.sps), pps(s.pps), sh(s.sh), pic(s.pic), refp(s.refp), poc(s.poc), scan(s.scan), scan_inv(s.scan_inv) { history_reset(); }

    // motion vector predictor candidates of one list (xevd_get_motion, xevd_util.c:469-515; availability xevd_get_avail_inter :632-687):
    // left, up, up-right neighbour SCU (1,1 when not available) and the co-located list-0 motion of reference 0 of that list
    void mvp_candidates(const Cu &cu, int lidx, int16_t cand[4][2]) const
    {
        const int xs = cu.x >> 2, ys = cu.y >> 2, scuw = (1 << cu.log2w) >> 2, ws = pic.w_scu;
        const int scup = ys * ws + xs;
        auto take = [&](int k, bool ok, int s) {
            cand[k][0] = ok ? pic.mv[(size_t)s * 4 + lidx * 2] : (int16_t)1;
            cand[k][1] = ok ? pic.mv[(size_t)s * 4 + lidx * 2 + 1] : (int16_t)1;
        };
        // an intra-block-copy neighbour does not count on the left and above - but does above-right, where the Main library's availability test
        // only asks "coded and not intra" (xevdm_get_avail_inter, xevdm_util.c:1468-1503): its stored vector is the block vector (list 1: zero)
        take(0, xs > 0 && pic.same_tile(scup, scup - 1) && !pic.intra[scup - 1] && pic.cod[scup - 1] && !pic.ibc[scup - 1], scup - 1);
        take(1, ys > 0 && pic.same_tile(scup, scup - ws) && !pic.intra[scup - ws] && !pic.ibc[scup - ws], scup - ws);
        take(2, ys > 0 && xs + scuw < ws && pic.same_tile(scup, scup - ws + scuw) && pic.cod[scup - ws + scuw] && !pic.intra[scup - ws + scuw], scup - ws + scuw);
        const RefPic *col = refp[lidx].empty() ? nullptr : refp[lidx][0];
        cand[3][0] = col ? col->mv0[(size_t)scup * 2] : (int16_t)0;
        cand[3][1] = col ? col->mv0[(size_t)scup * 2 + 1] : (int16_t)0;
    }
    // ------------------------------------------------------------------------------------------------------------------------------
    // Main profile, sps->tool_admvp: merge candidates (skip and merge-mode CUs) and the predictor of explicitly coded motion.
    // No SUCO here, so the right-hand neighbours are never decoded before the CU: the avail_lr LR_10 / LR_00 branch of the reference.
    // ------------------------------------------------------------------------------------------------------------------------------
    struct Motion { int8_t refi[2]; int16_t mv[2][2]; };
    // sps->tool_hmvp: the motion of the last 23 inter CUs of the CTU row (XEVD_HISTORY_BUFFER; xevdm_hmvp_init at the start of every CTU row,
    // xevdm.c:553-566, 2499-2503; update_history_buffer_parse_affine after every inter CU, :657-778)
    Motion hist[23];
    int hist_cnt = 0;
    void history_reset() { hist_cnt = 0; for (Motion &m : hist) { m.refi[0] = m.refi[1] = -1; memset(m.mv, 0, sizeof(m.mv)); } }
    void history_push(const Cu &cu)
    {
        if (hist_cnt == 23) { for (int i = 1; i < 23; i++) hist[i - 1] = hist[i]; hist_cnt = 22; }
        Motion &m = hist[hist_cnt++];
        for (int l = 0; l < 2; l++) { m.refi[l] = (int8_t)cu.refi[l]; m.mv[l][0] = cu.mv[l][0]; m.mv[l][1] = cu.mv[l][1]; }
        if (cu.affine) for (int l = 0; l < 2; l++) { m.mv[l][0] = m.mv[l][1] = 0; if (cu.refi[l] >= 0) aff_centre(cu, l, m.mv[l]); }      // an affine CU leaves the vector at its centre
    }
    bool bi_applicable(const Cu &cu) const { return sh.type == XHOST_SLICE_B && (1 << cu.log2w) + (1 << cu.log2h) > 12; }      // xevdm_check_bi_applicability, xevdm_util.c:1083-1096
    // the five spatial neighbours H, D, E, I, A (xevdm_check_motion_availability, xevdm_util.c:594-748, last branch): decoded, inter, not IBC
    void adm_neighbours(const Cu &cu, int neb[5], bool valid[5]) const
    {
        const int ws = pic.w_scu, hs = pic.h_scu, xs = cu.x >> 2, ys = cu.y >> 2, scuw = (1 << cu.log2w) >> 2, scuh = (1 << cu.log2h) >> 2, scup = ys * ws + xs;
        neb[0] = scup + (scuh - 1) * ws - 1; neb[1] = scup - ws + scuw - 1; neb[2] = scup - ws + scuw; neb[3] = scup + scuh * ws - 1; neb[4] = scup - ws - 1;
        const bool in[5] = { xs > 0, ys > 0, ys > 0 && xs + scuw < ws, ys + scuh < hs && xs > 0, ys > 0 && xs > 0 };
        for (int k = 0; k < 5; k++) valid[k] = in[k] && pic.same_tile(scup, neb[k]) && pic.cod[neb[k]] && !pic.intra[neb[k]] && !pic.ibc[neb[k]];      // the tile test first: another tile's maps may be written right now
    }
    static void scale_mv(int ratio, const int16_t in[2], int16_t out[2])      // scaling_mv, xevdm_util.c:180-190 (MVP_SCALING_PRECISION 5)
    {
        for (int d = 0; d < 2; d++) {
            int t = in[d] * ratio;
            t = t == 0 ? 0 : t > 0 ? (t + 16) >> 5 : -((-t + 16) >> 5);
            out[d] = (int16_t)std::min(std::max(t, -32768), 32767);
        }
    }
    // temporal candidate from the collocated picture at SCU `scu_col` (xevdm_get_mv_collocated, xevdm_util.c:3729-3818): 0 none, bit 0 list 0, bit 1 list 1
    int collocated(const Cu &cu, int scu_col, int16_t mvp[2][2]) const
    {
        int list = sh.type == XHOST_SLICE_P ? 0 : 1, ref = 0, src = 0;
        if (sh.tmvp_assigned) { list = sh.col_list; ref = sh.col_ref; src = sh.col_src_list; }
        memset(mvp, 0, sizeof(int16_t) * 4);
        if (ref >= (int)refp[list].size()) return 0;
        const RefPic *col = refp[list][ref];
        if (col->refi.empty()) return 0;
        const int dpoc[2] = { refp[0].empty() ? 0 : poc - refp[0][0]->poc, refp[1].empty() ? 0 : poc - refp[1][0]->poc };
        int have[2] = { 0, 0 };
        if (!sh.tmvp_assigned) {
            for (int l = 0; l < 2; l++) {
                const int r = col->refi[(size_t)scu_col * 2 + l];
                if (r < 0 || r >= 16) continue;
                const int dco = col->poc - col->list_poc[r];
                if (dco == 0) continue;
                have[l] = 1;
                scale_mv((dpoc[l] << 5) / dco, &col->mv[(size_t)scu_col * 4 + l * 2], mvp[l]);
            }
        } else {
            const int r = col->refi[(size_t)scu_col * 2 + src];
            const int dco = (r >= 0 && r < 16) ? col->poc - col->list_poc[r] : 0;
            if (dco != 0) {
                have[0] = have[1] = 1;
                for (int l = 0; l < 2; l++) scale_mv((dpoc[l] << 5) / dco, &col->mv[(size_t)scu_col * 4 + src * 2], mvp[l]);
            }
        }
        // xevdm_clip_mv_pic (xevdm_util.c:1409-1421) at the CU's position - samples and quarter samples mixed exactly like the reference does
        const int x = cu.x, y = cu.y, max_x = 144 + (pic.w_scu << 2) - 1, max_y = 144 + (pic.h_scu << 2) - 1, mn = -144;      // PIC_PAD_SIZE_L = MAX_CU_SIZE + 16
        for (int l = 0; l < 2; l++) { if (x + mvp[l][0] < mn) mvp[l][0] = (int16_t)-(x + mn); }
        for (int l = 0; l < 2; l++) { if (y + mvp[l][1] < mn) mvp[l][1] = (int16_t)-(y + mn); }
        for (int l = 0; l < 2; l++) { if (x + mvp[l][0] > max_x) mvp[l][0] = (int16_t)(max_x - x); }
        for (int l = 0; l < 2; l++) { if (y + mvp[l][1] > max_y) mvp[l][1] = (int16_t)(max_y - y); }
        return have[0] | (have[1] << 1);
    }
    // xevdm_get_motion_merge_main (xevdm_util.c:1169-1391) without the history candidates (sps->tool_hmvp off): up to 6 candidates (4 for CUs of 32 samples)
    // refined: the spatial candidates come from the refined map (ctx->map_mv) - the list xevdm_get_mmvd_mvp_list builds for an MMVD CU (xevdm_util.c:246-247);
    // the ordinary merge list reads the CUs' own vectors (map_unrefined_mv for refined neighbours, :1212-1216)
    void merge_candidates(const Cu &cu, Motion cand[6], bool refined = false) const
    {
        const std::vector<int16_t> &map_mv = (refined && !pic.mv_ref.empty()) ? pic.mv_ref : pic.mv;
        const int ws = pic.w_scu, hs = pic.h_scu, xs = cu.x >> 2, ys = cu.y >> 2, cuw = 1 << cu.log2w, cuh = 1 << cu.log2h, scup = ys * ws + xs;
        const int max_n = cuw * cuh <= 32 ? 4 : 6;
        const bool is_b = sh.type == XHOST_SLICE_B, bi = bi_applicable(cu);
        for (int k = 0; k < 6; k++) { cand[k].refi[0] = cand[k].refi[1] = -1; memset(cand[k].mv, 0, sizeof(cand[k].mv)); }
        int cnt = 0;
        auto insert = [&](const int8_t r[2], const int16_t *mv) {       // xevdm_get_merge_insert_mv + check_redundancy
            Motion &d = cand[cnt];
            d.refi[0] = r[0] >= 0 ? r[0] : (int8_t)-1; d.mv[0][0] = mv[0]; d.mv[0][1] = mv[1];
            if (is_b) {
                if (r[0] >= 0 && !bi) { d.refi[1] = -1; d.mv[1][0] = d.mv[1][1] = 0; }
                else { d.refi[1] = r[1] >= 0 ? r[1] : (int8_t)-1; d.mv[1][0] = mv[2]; d.mv[1][1] = mv[3]; }
            }
            bool dup = false;
            for (int i = cnt - 1; i >= 0 && !dup; i--)
                dup = d.refi[0] == cand[i].refi[0] && d.mv[0][0] == cand[i].mv[0][0] && d.mv[0][1] == cand[i].mv[0][1] &&
                      (!is_b || (d.refi[1] == cand[i].refi[1] && d.mv[1][0] == cand[i].mv[1][0] && d.mv[1][1] == cand[i].mv[1][1]));
            if (!dup) cnt++;
            return !dup;
        };
        int neb[5]; bool valid[5];
        adm_neighbours(cu, neb, valid);
        for (int k = 0; k < 5; k++) {
            if (valid[k]) insert(&pic.refi[(size_t)neb[k] * 2], &map_mv[(size_t)neb[k] * 4]);
            if (cnt == max_n - 1) break;
        }
        // temporal: the centre of the CU on the 8x8 grid, else below, else to the right (inside the CTU row / column)
        bool tmvp_added = false;
        auto temporal = [&](int scu_col) -> bool {      // true: candidate list complete
            int16_t t[2][2];
            const int av = collocated(cu, scu_col, t);
            if (!av) return false;
            const int8_t r[2] = { (int8_t)((av & 1) ? 0 : -1), (int8_t)((av & 2) ? 0 : -1) };
            tmvp_added = insert(r, &t[0][0]);
            return cnt >= max_n;
        };
        if (temporal(((xs + (cuw >> 3)) >> 1 << 1) + ((ys + (cuh >> 3)) >> 1 << 1) * ws)) return;
        const int xe = xs + (cuw >> 2) - 1, ye = ys + (cuh >> 2) - 1;
        if (!tmvp_added && ye + 1 < hs && ((ye + 1) << 2 >> 6) == (ye << 2 >> 6))
            if (temporal(((ye + 1) >> 1 << 1) * ws + (xe >> 1 << 1))) return;
        if (!tmvp_added && xe + 1 < ws && ((xe + 1) << 2 >> 6) == (xe << 2 >> 6))
            if (temporal((ye >> 1 << 1) * ws + ((xe + 1) >> 1 << 1))) return;
        // every fourth entry of the history, newest first (with tool_hmvp off the buffer is empty)
        for (int k = 3; k <= std::min(hist_cnt, max_n == 4 ? 15 : 23); k += 4) {
            insert(hist[hist_cnt - k].refi, &hist[hist_cnt - k].mv[0][0]);
            if (cnt >= max_n) return;
        }
        if (bi) {       // combinations of the list-0 part of one candidate with the list-1 part of another
            static const int p0[20] = { 0, 1, 0, 2, 1, 2, 0, 3, 1, 3, 2, 3, 0, 4, 1, 4, 2, 4, 3, 4 }, p1[20] = { 1, 0, 2, 0, 2, 1, 3, 0, 3, 1, 3, 2, 4, 0, 4, 1, 4, 2, 4, 3 };
            const int cur = cnt;
            for (int i = 0; i < cur * (cur - 1) && cnt != max_n; i++) {
                const Motion a = cand[p0[i]], b = cand[p1[i]];
                if (a.refi[0] >= 0 && b.refi[1] >= 0) {
                    cand[cnt].refi[0] = a.refi[0]; cand[cnt].mv[0][0] = a.mv[0][0]; cand[cnt].mv[0][1] = a.mv[0][1];
                    cand[cnt].refi[1] = b.refi[1]; cand[cnt].mv[1][0] = b.mv[1][0]; cand[cnt].mv[1][1] = b.mv[1][1];
                    cnt++;
                }
            }
            if (cnt == max_n) return;
        }
        for (int k = cnt; k < max_n; k++) { cand[k].refi[0] = 0; cand[k].refi[1] = bi ? 0 : -1; memset(cand[k].mv, 0, sizeof(cand[k].mv)); }
        (void)scup;
    }
    // mmvd_group_idx (only with the slice's group flag and above 32 samples), mmvd_merge_idx, mmvd_distance_idx, mmvd_direction_idx (xevdm_eco_mmvd_data, xevdm_eco.c:767-812)
    template <class C> void code_mmvd_idx(C &c, Cu &cu)
    {
        int grp = cu.mmvd_idx >> 7, base = (cu.mmvd_idx >> 5) & 3, dist = (cu.mmvd_idx >> 2) & 7, dir = cu.mmvd_idx & 3;
        if (sh.mmvd_group && (1 << (cu.log2w + cu.log2h)) > 32) {
            if (c.bin(grp > 0, models.mmvd_group_idx[0])) grp = 1 + c.bin(grp > 1, models.mmvd_group_idx[1]); else grp = 0;
        } else grp = 0;
        base = sym_trunc_unary(c, base, models.mmvd_merge_idx, 3, 4);
        dist = sym_trunc_unary(c, dist, models.mmvd_dist_idx, 7, 8);
        dir = (c.bin(dir >> 1, models.mmvd_dir_idx[0]) << 1) | c.bin(dir & 1, models.mmvd_dir_idx[1]);
        cu.mmvd_idx = (grp << 7) | (base << 5) | (dist << 2) | dir;
    }
    // merge with vector difference (xevdm_get_mmvd_motion, xevdm_util.c:4682-4716; xevdm_get_mmvd_mvp_list :191-592, for ONE index): base candidate
    // `base` of the merge list, turned into the prediction type of its group (bi-predictive / list 0 / list 1, the missing list mirrored and scaled by
    // POC distance; P slices: the same, another or a third reference), plus an offset of 1 .. 128 quarter samples in one of four directions, scaled
    // between the lists by their POC distances and mirrored when the references lie on either side
    void mmvd_motion(Cu &cu) const
    {
        Motion cand[6];
        merge_candidates(cu, cand, true);
        const int grp = cu.mmvd_idx >> 7, base = (cu.mmvd_idx >> 5) & 3, kk = cu.mmvd_idx & 31;
        const bool is_b = sh.type == XHOST_SLICE_B, small = (1 << (cu.log2w + cu.log2h)) <= 32;
        auto rpoc = [&](int l, int r) -> int { return (r >= 0 && r < (int)refp[l].size()) ? refp[l][(size_t)r]->poc : 0; };      // REF_SET
        // POC distances are non-zero for the lists build_ref_lists accepts; a REF_SET miss (rpoc = 0 at POC 0) must still not trap in the parser
        auto sdiv = [](int a, int b) -> int { return b ? a / b : 0; };
        auto scaled = [&](int w, int v, int sg) -> int { return std::min(std::max(sg * ((abs(w * v) + 16) >> 5), -32768), 32767); };
        // base_mv_t: the candidate (P slices take list 1 of candidate 0: unused), types per group
        int t[2][3] = { { cand[base].mv[0][0], cand[base].mv[0][1], cand[base].refi[0] },
                        { is_b ? cand[base].mv[1][0] : cand[0].mv[1][0], is_b ? cand[base].mv[1][1] : cand[0].mv[1][1], is_b ? cand[base].refi[1] : cand[0].refi[1] } };
        int b[2][3] = { { t[0][0], t[0][1], t[0][2] }, { t[1][0], t[1][1], t[1][2] } };      // base_mv: starts as the candidate
        int pm[3][3] = { { 0 } }, type[3];
        const int n0 = (int)refp[0].size(), n1 = (int)refp[1].size();
        if (t[0][2] >= 0 && t[1][2] >= 0) { type[0] = 0; type[1] = 1; type[2] = 2; }
        else if (t[0][2] >= 0) {
            if (!is_b) {
                type[0] = type[1] = type[2] = 1;
                pm[0][2] = t[0][2];
                pm[1][2] = n0 == 1 ? t[0][2] : !t[0][2];
                pm[2][2] = n0 < 3 ? t[0][2] : (t[0][2] < 2 ? 2 : 1);
                pm[0][0] = t[0][0]; pm[0][1] = t[0][1];
                if (n0 == 1) { pm[1][0] = t[0][0] + 3; pm[1][1] = t[0][1]; pm[2][0] = t[0][0] - 3; pm[2][1] = t[0][1]; }
                else {
                    for (int g = 1; g <= (n0 == 2 ? 1 : 2); g++) {
                        const int w = sdiv((poc - rpoc(0, pm[0][2])) << 5, poc - rpoc(0, pm[g][2]));
                        pm[g][0] = scaled(w, t[0][0], 1); pm[g][1] = scaled(w, t[0][1], 1);
                    }
                    if (n0 == 2) { pm[2][0] = t[0][0] - 3; pm[2][1] = t[0][1]; }
                }
            } else {
                type[0] = 1; type[1] = 0; type[2] = 2;
                const int p0 = rpoc(0, t[0][2]);
                t[1][2] = (n1 > 1 && rpoc(1, 1) - poc == poc - p0) ? 1 : 0;
                const int w = sdiv((poc - rpoc(1, t[1][2])) << 5, poc - p0);
                t[1][0] = scaled(w, t[0][0], w * t[0][0] < 0 ? -1 : 1); t[1][1] = scaled(w, t[0][1], w * t[0][1] < 0 ? -1 : 1);
            }
        } else if (t[1][2] >= 0) {
            type[0] = 2; type[1] = 0; type[2] = 1;
            const int p1 = rpoc(1, t[1][2]);
            t[0][2] = (n0 > 1 && rpoc(0, 1) - poc == poc - p1) ? 1 : 0;
            const int w = sdiv((poc - rpoc(0, t[0][2])) << 5, poc - p1);
            t[0][0] = scaled(w, t[1][0], w * t[1][0] < 0 ? -1 : 1); t[0][1] = scaled(w, t[1][1], w * t[1][1] < 0 ? -1 : 1);
        } else type[0] = type[1] = type[2] = 3;
        if (small) type[0] = 1;
        switch (type[grp]) {
        case 0: for (int l = 0; l < 2; l++) for (int d = 0; d < 3; d++) b[l][d] = t[l][d]; break;
        case 1: if (!is_b) { b[0][0] = pm[grp][0]; b[0][1] = pm[grp][1]; b[0][2] = pm[grp][2]; } else { b[0][0] = t[0][0]; b[0][1] = t[0][1]; b[0][2] = t[0][2]; } b[1][2] = -1; break;
        case 2: b[0][2] = -1; b[1][0] = t[1][0]; b[1][1] = t[1][1]; b[1][2] = t[1][2]; break;
        default: b[0][2] = b[1][2] = -1; break;
        }
        const int r0 = b[0][2], r1 = b[1][2], step = 1 << (kk >> 2);
        int sign = 1, d0 = step, d1 = step;
        if (r0 != -1 && r1 != -1) {
            const int p0 = rpoc(0, r0), p1 = rpoc(1, r1);
            if (is_b && (p0 - poc) * (poc - p1) > 0) sign = -1;
            if (abs(p1 - poc) >= abs(p0 - poc)) d0 = std::min(std::max((sdiv(abs(p0 - poc) << 5, abs(p1 - poc)) * step + 16) >> 5, -32768), 32767);
            else d1 = std::min(std::max((sdiv(abs(p1 - poc) << 5, abs(p0 - poc)) * step + 16) >> 5, -32768), 32767);
        }
        const int dir = kk & 3, s0 = (dir & 1) ? -d0 : d0, s1 = ((dir & 1) ? -d1 : d1) * sign;
        const int real[2][2] = { { b[0][0] + (dir < 2 ? s0 : 0), b[0][1] + (dir < 2 ? 0 : s0) }, { b[1][0] + (dir < 2 ? s1 : 0), b[1][1] + (dir < 2 ? 0 : s1) } };
        cu.refi[0] = r0; cu.mv[0][0] = (int16_t)real[0][0]; cu.mv[0][1] = (int16_t)real[0][1];
        if (is_b) { cu.refi[1] = r1; cu.mv[1][0] = (int16_t)real[1][0]; cu.mv[1][1] = (int16_t)real[1][1]; }
        else { cu.refi[1] = -1; cu.mv[1][0] = cu.mv[1][1] = 0; }
    }
    // skip / merge-mode motion = candidate `idx` (xevd_get_skip_motion / xevd_get_direct_motion, xevdm.c:800-883); entries past the list stay "no reference, zero"
    void merge_motion(Cu &cu, int idx) const
    {
        Motion cand[6];
        merge_candidates(cu, cand);
        const Motion &m = cand[std::min(std::max(idx, 0), 5)];
        cu.refi[0] = m.refi[0]; cu.mv[0][0] = m.mv[0][0]; cu.mv[0][1] = m.mv[0][1];
        if (sh.type == XHOST_SLICE_P) { cu.refi[1] = -1; cu.mv[1][0] = cu.mv[1][1] = 0; }
        else { cu.refi[1] = m.refi[1]; cu.mv[1][0] = m.mv[1][0]; cu.mv[1][1] = m.mv[1][1]; }
    }
    // the fallback motion of a list among the first two neighbours (xevdm_get_default_motion, xevdm_util.c:783-867, no history): the one with `cur_refi`, else any
    void default_motion(const int neb[5], const bool valid[5], int cur_refi, int l, int &refi, int16_t mv[2]) const
    {
        refi = 0; mv[0] = mv[1] = 0;
        for (int pass = 0; pass < 2; pass++)
            for (int k = 0; k < 2; k++) {
                if (!valid[k]) continue;
                const int r = pic.refi[(size_t)neb[k] * 2 + l];
                if (r >= 0 && (pass == 1 || r == cur_refi)) { refi = r; mv[0] = pic.mv[(size_t)neb[k] * 4 + l * 2]; mv[1] = pic.mv[(size_t)neb[k] * 4 + l * 2 + 1]; return; }
            }
        if (!sps.tool_hmvp) return;
        for (int pass = 0; pass < 2; pass++)                       // ... then the four newest history entries the same way
            for (int k = 1; k <= std::min(hist_cnt, 4); k++) {
                const int r = hist[hist_cnt - k].refi[l];
                if (r >= 0 && (pass == 1 || r == cur_refi)) { refi = r; mv[0] = hist[hist_cnt - k].mv[l][0]; mv[1] = hist[hist_cnt - k].mv[l][1]; return; }
            }
    }
    // reference index of a list of a bi-predicted CU that does not code it (bi_idx FL0 / FL1: xevdm_get_first_refi, xevdm_util.c:750-781), resolution index 0
    int first_refi(const Cu &cu, int l, int mvr) const
    {
        int neb[5], dref; bool valid[5]; int16_t dmv[2];
        adm_neighbours(cu, neb, valid);
        default_motion(neb, valid, 0, l, dref, dmv);
        if (valid[mvr] && pic.refi[(size_t)neb[mvr] * 2 + l] >= 0) return pic.refi[(size_t)neb[mvr] * 2 + l];      // the neighbour position is coupled with the resolution index
        return dref;
    }
    // the predictor of explicitly coded motion at resolution index 0 (xevdm_get_motion_from_mvr, xevdm_util.c:869-951): neighbour H, scaled to the
    // CU's reference when it points elsewhere, else the fallback motion
    void mvr_predictor(const Cu &cu, int l, int cur_refi, int mvr, int16_t mvp[2]) const
    {
        int neb[5], dref; bool valid[5]; int16_t dmv[2];
        adm_neighbours(cu, neb, valid);
        default_motion(neb, valid, cur_refi, l, dref, dmv);
        const int n = (int)refp[l].size(), pc = refp[l][std::min(std::max(cur_refi, 0), n - 1)]->poc;
        auto ratio = [&](int r) -> int { const int t0 = poc - refp[l][std::min(std::max(r, 0), n - 1)]->poc; return t0 ? ((poc - pc) * 32) / t0 : 0; };
        int r = valid[mvr] ? (int)pic.refi[(size_t)neb[mvr] * 2 + l] : -1;
        if (r >= 0) {
            const int16_t *m = &pic.mv[(size_t)neb[mvr] * 4 + l * 2];
            if (r == cur_refi) { mvp[0] = m[0]; mvp[1] = m[1]; } else scale_mv(ratio(r), m, mvp);
        } else {
            if (dref == cur_refi) { mvp[0] = dmv[0]; mvp[1] = dmv[1]; } else scale_mv(ratio(dref), dmv, mvp);
        }
        const int rnd = mvr > 0 ? 1 << (mvr - 1) : 0;             // the predictor on the grid of the resolution, rounded away from zero at the half
        for (int d = 0; d < 2; d++) mvp[d] = (int16_t)(mvp[d] >= 0 ? ((mvp[d] + rnd) >> mvr) << mvr : -(((-mvp[d] + rnd) >> mvr) << mvr));
    }

    // temporal direct motion of a B CU (xevd_get_mv_dir, xevd_util.c:540-566; call site xevd.c:713-717): the list-0 motion the
    // co-located picture (reference 0 of list 1) stored at the CU's bottom-right SCU, scaled by POC distances (C division)
    void direct_motion(Cu &cu) const
    {
        const int ws = pic.w_scu, scup = (cu.y >> 2) * ws + (cu.x >> 2);
        const int c_scu = scup + (((1 << cu.log2w) >> 2) - 1) + (((1 << cu.log2h) >> 2) - 1) * ws;
        const RefPic *r0 = refp[0][0], *col = refp[1][0];
        const int mvx = col->mv0[(size_t)c_scu * 2], mvy = col->mv0[(size_t)c_scu * 2 + 1];
        const int dco = col->poc - col->list0_poc, d0 = poc - r0->poc, d1 = col->poc - poc;
        cu.refi[0] = cu.refi[1] = 0;
        if (dco == 0) { memset(cu.mv, 0, sizeof(cu.mv)); return; }
        cu.mv[0][0] = (int16_t)(d0 * mvx / dco); cu.mv[0][1] = (int16_t)(d0 * mvy / dco);
        cu.mv[1][0] = (int16_t)(-d1 * mvx / dco); cu.mv[1][1] = (int16_t)(-d1 * mvy / dco);
    }
    // code-number table of the luma intra mode (xevd_get_mpm_b, xevd_ipred.c:678-692): neighbours count when intra and already parsed
    const uint8_t *mpm_list(const Cu &cu) const
    {
        const int xs = cu.x >> 2, ys = cu.y >> 2, ws = pic.w_scu, scup = ys * ws + xs;
        int l = 0, u = 0;
        if (xs > 0 && pic.same_tile(scup, scup - 1) && pic.intra[scup - 1] && pic.cod[scup - 1]) l = pic.ipm[scup - 1] + 1;
        if (ys > 0 && pic.same_tile(scup, scup - ws) && pic.intra[scup - ws] && pic.cod[scup - ws]) u = pic.ipm[scup - ws] + 1;
        return k_mpm[l][u];
    }
    // tool_eipd: the two most probable modes, eight "extended" ones and the ordering of all 33 (xevdm_get_mpm, src_main/xevdm_ipred.c:
    // 320-767) for a CU whose right-hand neighbour is not coded yet - always the case without SUCO, which this front end rejects.
    void eipd_mpm(const Cu &cu, int mpm[2], int ext[8], int pims[33]) const
    {
        enum { DC = 0, PLN = 1, BI = 2, VER = 12, HOR = 24, DIA_R = 18, DIA_L = 6, DIA_U = 30, CNT = 33 };
        const int xs = cu.x >> 2, ys = cu.y >> 2, ws = pic.w_scu, scup = ys * ws + xs;
        int l = DC, u = DC;
        if (xs > 0 && pic.same_tile(scup, scup - 1) && pic.intra[scup - 1] && pic.cod[scup - 1]) l = pic.ipm[scup - 1];
        if (ys > 0 && pic.same_tile(scup, scup - ws) && pic.intra[scup - ws] && pic.cod[scup - ws]) u = pic.ipm[scup - ws];
        mpm[0] = std::min(l, u); mpm[1] = std::max(l, u);
        if (mpm[0] == mpm[1]) { mpm[0] = DC; mpm[1] = mpm[1] == DC ? BI : mpm[1]; }
        const int m0 = mpm[0], m1 = mpm[1];
        if (m1 < 3) {                                         // two non-angular modes: the third one, then the main directions
            const int e[8] = { m0 == DC ? (m1 == BI ? PLN : BI) : DC, VER, HOR, DIA_R, DIA_L, DIA_U, VER + 4, HOR - 4 };
            memcpy(ext, e, sizeof(e));
        } else if (m0 < 3) {                                  // one angular mode: the other non-angular ones, then its neighbourhood
            ext[0] = m0 == PLN ? BI : (m0 == BI ? DC : BI);
            ext[1] = m0 == PLN ? DC : PLN;
            if (m1 > CNT - 3)  { const int e[6] = { m1 == CNT - 1 ? CNT - 2 : CNT - 1, CNT - 3, CNT - 4, CNT - 5, HOR, DIA_R }; memcpy(ext + 2, e, sizeof(e)); }
            else if (m1 < 5)   { const int e[6] = { m1 == 3 ? 4 : 3, 5, 6, 7, VER, DIA_R }; memcpy(ext + 2, e, sizeof(e)); }
            else {
                ext[2] = m1 + 2; ext[3] = m1 - 2; ext[4] = m1 + 1; ext[5] = m1 - 1;
                if (m1 <= 23 && m1 >= 13) { ext[6] = m1 - 5; ext[7] = m1 + 5; }
                else { ext[6] = m1 > 23 ? m1 - 5 : m1 + 5; ext[7] = m1 > 23 ? m1 - 10 : m1 + 10; }
            }
        } else {                                              // two angular modes: their neighbours and means, then the main directions
            int list[15] = { (m0 == 3 || m0 == 4) ? m0 + 1 : m0 - 2, m0 == CNT - 2 ? m0 - 1 : m0 + 2, m1 == 4 ? m1 + 1 : m1 - 2,
                             (m1 == CNT - 1 || m1 == CNT - 2) ? m1 - 1 : m1 + 2, (m0 + m1 + 1) >> 1, 0, 0,
                             VER, HOR, DIA_R, PLN, DIA_L, DIA_U, VER + 4, HOR - 4 };
            list[5] = (list[4] + m0 + 1) >> 1; list[6] = (list[4] + m1 + 1) >> 1;
            ext[0] = BI; ext[1] = DC;
            int n = 2;
            for (int i = 0; i < 15 && n < 8; i++) {
                bool dup = list[i] == m0 || list[i] == m1;
                for (int j = 0; j < n && !dup; j++) dup = list[i] == ext[j];
                if (!dup) ext[n++] = list[i];
            }
        }
        // the order of all modes: the two, the eight, then a fixed list (intra_mode_list, xevdm_ipred.c:307-318), duplicates dropped
        static const int k_default[33] = { DC, BI, VER, PLN, HOR, VER - 1, VER + 1, VER - 2, VER + 2, VER - 3, VER + 3, HOR - 1, HOR + 1, HOR - 2, HOR + 2,
                                           HOR - 3, HOR + 3, DIA_R, DIA_L, DIA_L - 3, DIA_L - 2, DIA_L - 1, DIA_U, DIA_U + 1, DIA_U + 2, VER + 5, VER + 4,
                                           HOR - 4, HOR - 5, VER - 5, VER - 4, HOR + 5, HOR + 4 };
        bool in[33] = { false };
        int n = 0;
        auto add = [&](int m) { if (m >= 0 && m < 33 && !in[m]) { in[m] = true; pims[n++] = m; } };
        add(m0); add(m1);
        for (int i = 0; i < 8; i++) add(ext[i]);
        for (int i = 0; i < 33; i++) add(k_default[i]);
    }
    void chroma_qps(int qp, int &qp_u, int &qp_v) const      // xevd_eco.c:663-666
    {
        const int off = 6 * (sps.bd_c - 8);
        const int iu = std::min(std::max(qp + sh.qp_u_offset, -off), 57), iv = std::min(std::max(qp + sh.qp_v_offset, -off), 57);
        if (sps.cqt) { qp_u = sps.cq[0][iu + off] + off; qp_v = sps.cq[1][iv + off] + off; return; }
        const int8_t *tbl = sps.tool_iqt ? k_chroma_qp_main : k_chroma_qp;
        qp_u = (iu >= 0 ? tbl[iu] : 0) + off;                 // entries below 0 of the default table are zero-initialised storage
        qp_v = (iv >= 0 ? tbl[iv] : 0) + off;
    }

    // ------------------------------------------------------------------------------------------------------------------------------
    // sps->tool_affine: control-point vectors of affine merge / affine inter CUs (xevd_get_affine_motion, src_main/xevdm.c:937-1013;
    // candidates src_main/xevdm_util.c:2145-3187).  Without SUCO the right-hand neighbours are never decoded before the CU: avail_lr is
    // LR_10 or LR_00, and the branches the reference keeps for LR_01 / LR_11 are not restated.
    // ------------------------------------------------------------------------------------------------------------------------------
    static int aff_rnd(int v, int shift) { return (v + (1 << (shift - 1)) - (v >= 0)) >> shift; }      // xevdm_mv_rounding_s32, xevdm_util.c:1856-1861
    static int16_t clip16(int v) { return (int16_t)std::min(std::max(v, -32768), 32767); }
    // neighbour SCU n of the CU at scup: decoded, inter, same tile - and affine (model-based candidates) or not IBC (corner vectors)
    bool aff_nb(int scup, int n, bool inside, bool need_affine) const
    {
        if (!inside || !pic.same_tile(scup, n) || !pic.cod[n] || pic.intra[n]) return false;
        return need_affine ? pic.aff[n] != 0 : !pic.ibc[n];
    }
    // the model of the affine CU that covers SCU scun, evaluated at this CU's corners (xevdm_derive_affine_model_mv, xevdm_util.c:2270-2363)
    void aff_inherit(const Cu &cu, int scun, int l, int cp_num, int16_t mvp[3][2]) const
    {
        const int ws = pic.w_scu, a = pic.aff[(size_t)scun], nlw = (a >> 2) & 7, nlh = (a >> 5) & 7, nw = 1 << nlw, nh = 1 << nlh, tl = (int)pic.aff_tl[(size_t)scun];
        const int addr[4] = { tl, tl + (nw >> 2) - 1, tl + ((nh >> 2) - 1) * ws, tl + ((nh >> 2) - 1) * ws + (nw >> 2) - 1 };
        int nmv[4][2];
        for (int i = 0; i < 4; i++) { nmv[i][0] = pic.mv[(size_t)addr[i] * 4 + l * 2]; nmv[i][1] = pic.mv[(size_t)addr[i] * 4 + l * 2 + 1]; }
        const int nx = (tl % ws) << 2;
        int ny = (tl / ws) << 2;
        bool top = false;
        if ((ny + nh) % 64 == 0 && ny + nh == cu.y) {          // the neighbour sits in the CTU row above: only its bottom row of vectors is used (line buffer)
            top = true; ny += nh;
            nmv[0][0] = nmv[2][0]; nmv[0][1] = nmv[2][1]; nmv[1][0] = nmv[3][0]; nmv[1][1] = nmv[3][1];
        }
        const int dhx = (nmv[1][0] - nmv[0][0]) * (1 << (7 - nlw)), dhy = (nmv[1][1] - nmv[0][1]) * (1 << (7 - nlw));
        int dvx = -dhy, dvy = dhx;
        if (cp_num == 3 && !top) { dvx = (nmv[2][0] - nmv[0][0]) * (1 << (7 - nlh)); dvy = (nmv[2][1] - nmv[0][1]) * (1 << (7 - nlh)); }
        const int hb = nmv[0][0] * 128, vb = nmv[0][1] * 128;
        auto at = [&](int px, int py, int16_t out[2]) {
            out[0] = clip16(aff_rnd(dhx * px + dvx * py + hb, 7)); out[1] = clip16(aff_rnd(dhy * px + dvy * py + vb, 7));
        };
        at(cu.x - nx, cu.y - ny, mvp[0]);
        at(cu.x - nx + (1 << cu.log2w), cu.y - ny, mvp[1]);
        if (cp_num == 3) at(cu.x - nx, cu.y - ny + (1 << cu.log2h), mvp[2]);
    }
    // the two predictor candidates of an affine inter CU for list l / reference cur_refi (xevdm_get_affine_motion_scaling, xevdm_util.c:2367-2761)
    void aff_amvp(const Cu &cu, int l, int cur_refi, int vn, int16_t mvp[2][3][2]) const
    {
        const int ws = pic.w_scu, hs = pic.h_scu, xs = cu.x >> 2, ys = cu.y >> 2, scuw = (1 << cu.log2w) >> 2, scuh = (1 << cu.log2h) >> 2, scup = ys * ws + xs;
        memset(mvp, 0, sizeof(int16_t) * 2 * 3 * 2);
        int cnt = 0;
        auto model = [&](const int *nb, const bool *in, int n) {          // the first neighbour of a group that is affine and uses this reference
            for (int k = 0; k < n; k++)
                if (aff_nb(scup, nb[k], in[k], true) && pic.refi[(size_t)nb[k] * 2 + l] >= 0 && pic.refi[(size_t)nb[k] * 2 + l] == cur_refi) {
                    int16_t t[3][2] = { { 0, 0 }, { 0, 0 }, { 0, 0 } };
                    aff_inherit(cu, nb[k], l, vn, t);
                    memcpy(mvp[cnt++], t, sizeof(t));
                    return;
                }
        };
        { const int nb[2] = { scup + ws * scuh - 1, scup + ws * (scuh - 1) - 1 }; const bool in[2] = { xs > 0 && ys + scuh < hs, xs > 0 }; model(nb, in, 2); }      // A0, A1
        if (cnt >= 2) return;
        { const int nb[3] = { scup - ws + scuw, scup - ws + scuw - 1, scup - ws - 1 }; const bool in[3] = { ys > 0 && xs + scuw < ws, ys > 0, xs > 0 && ys > 0 }; model(nb, in, 3); }      // B0, B1, B2
        if (cnt >= 2) return;
        { const int nb[2] = { scup + ws * scuh + scuw, scup + ws * (scuh - 1) + scuw }; const bool in[2] = { xs + scuw < ws && ys + scuh < hs, xs + scuw < ws }; model(nb, in, 2); }      // C0, C1
        if (cnt >= 2) return;
        // corner vectors: the first neighbour of each corner that uses this reference
        auto corner = [&](const int *nb, const bool *in, int n, int16_t out[2]) -> bool {
            out[0] = out[1] = 0;
            for (int k = 0; k < n; k++)
                if (aff_nb(scup, nb[k], in[k], false) && pic.refi[(size_t)nb[k] * 2 + l] == cur_refi && cur_refi >= 0) {
                    out[0] = pic.mv[(size_t)nb[k] * 4 + l * 2]; out[1] = pic.mv[(size_t)nb[k] * 4 + l * 2 + 1];
                    return true;
                }
            return false;
        };
        int16_t lt[2], rt[2], lb[2], rb[2];
        const int nlt[3] = { scup - ws - 1, scup - ws, scup - 1 }; const bool ilt[3] = { xs > 0 && ys > 0, ys > 0, xs > 0 };
        const int nrt[3] = { scup - ws + scuw, scup - ws + scuw - 1, scup + scuw }; const bool irt[3] = { ys > 0 && xs + scuw < ws, ys > 0, xs + scuw < ws };
        const int nlb[2] = { scup + ws * scuh - 1, scup + ws * (scuh - 1) - 1 }; const bool ilb[2] = { xs > 0 && ys + scuh < hs, xs > 0 };
        const int nrb[2] = { scup + ws * scuh + scuw, scup + ws * (scuh - 1) + scuw }; const bool irb[2] = { xs + scuw < ws && ys + scuh < hs, xs + scuw < ws };
        const bool c_lt = corner(nlt, ilt, 3, lt), c_rt = corner(nrt, irt, 3, rt), c_lb = corner(nlb, ilb, 2, lb), c_rb = corner(nrb, irb, 2, rb);
        auto put = [&](const int16_t a[2], const int16_t b[2], const int16_t c2[2]) { memcpy(mvp[cnt][0], a, 4); memcpy(mvp[cnt][1], b, 4); memcpy(mvp[cnt][2], c2, 4); cnt++; };
        if (c_lt && c_rt && (vn == 2 || c_lb || c_rb)) {
            int16_t third[2] = { lb[0], lb[1] };
            if (!c_lb && c_rb) { third[0] = clip16(rb[0] + lt[0] - rt[0]); third[1] = clip16(rb[1] + lt[1] - rt[1]); }
            put(lt, rt, third);
        }
        if (cnt == 2) return;
        if (c_lb) put(lb, lb, lb); else if (c_rb) put(rb, rb, rb);        // translational candidates: left, (right,) above, above-left
        if (cnt == 2) return;
        if (c_rt) put(rt, rt, rt);
        if (cnt == 2) return;
        if (c_lt) put(lt, lt, lt);
    }
    // one constructed candidate from corner vectors (xevdm_derive_affine_constructed_candidate, xevdm_util.c:2145-2268)
    void aff_constructed(const Cu &cu, const int cp_valid[4], const int16_t cp_mv[2][4][2], const int cp_refi[2][4], const int *idx, int model, int vn,
                         int8_t refi[5][2], int16_t cpmv[5][2][3][2], int cpn[5], int &cnt) const
    {
        if (cnt >= 5) return;
        for (int i = 0; i < vn; i++) if (!cp_valid[idx[i]]) return;
        bool ok[2];
        for (int l = 0; l < 2; l++) {
            ok[l] = true;
            for (int i = 0; i < vn; i++) ok[l] = ok[l] && cp_refi[l][idx[i]] >= 0 && cp_refi[l][idx[i]] == cp_refi[l][idx[0]];
        }
        if (!ok[0] && !ok[1]) return;
        cpn[cnt] = vn;
        const int sh_hw = 7 + cu.log2w - cu.log2h;
        for (int l = 0; l < 2; l++) {
            memset(cpmv[cnt][l], 0, sizeof(cpmv[cnt][l]));
            refi[cnt][l] = -1;
            if (!ok[l]) continue;
            refi[cnt][l] = (int8_t)cp_refi[l][idx[0]];
            int t[4][2] = { { 0, 0 }, { 0, 0 }, { 0, 0 }, { 0, 0 } };
            for (int i = 0; i < vn; i++) { t[idx[i]][0] = cp_mv[l][idx[i]][0]; t[idx[i]][1] = cp_mv[l][idx[i]][1]; }
            switch (model) {      // to top-left, top-right (, bottom-left)
            case 1: t[2][0] = t[3][0] + t[0][0] - t[1][0]; t[2][1] = t[3][1] + t[0][1] - t[1][1]; break;
            case 2: t[1][0] = t[3][0] + t[0][0] - t[2][0]; t[1][1] = t[3][1] + t[0][1] - t[2][1]; break;
            case 3: t[0][0] = t[1][0] + t[2][0] - t[3][0]; t[0][1] = t[1][1] + t[2][1] - t[3][1]; break;
            case 5: {
                const int h = (t[2][1] - t[0][1]) * (1 << sh_hw) + t[0][0] * 128, v = -((t[2][0] - t[0][0]) * (1 << sh_hw)) + t[0][1] * 128;
                t[1][0] = aff_rnd(h, 7); t[1][1] = aff_rnd(v, 7);
                break; }
            default: break;
            }
            for (int i = 0; i < vn; i++) { cpmv[cnt][l][i][0] = clip16(t[i][0]); cpmv[cnt][l][i][1] = clip16(t[i][1]); }
        }
        cnt++;
    }
    // the five affine merge candidates (xevdm_get_affine_merge_candidate, xevdm_util.c:2763-3187)
    void aff_merge(const Cu &cu, int8_t refi[5][2], int16_t cpmv[5][2][3][2], int cpn[5]) const
    {
        const int ws = pic.w_scu, hs = pic.h_scu, xs = cu.x >> 2, ys = cu.y >> 2, scuw = (1 << cu.log2w) >> 2, scuh = (1 << cu.log2h) >> 2, scup = ys * ws + xs;
        const bool left_avail = xs > 0 && pic.same_tile(scup, scup - 1) && pic.cod[(size_t)scup - 1];      // avail_lr LR_10 (xevd_check_nev_avail, xevd_util.c:1156-1174)
        int cnt = 0;
        memset(cpmv, 0, sizeof(int16_t) * 5 * 2 * 3 * 2);
        {   // model based: A1, B1, B0, A0, B2 - one candidate per distinct affine neighbour CU
            const int nb[5] = { scup + ws * (scuh - 1) - 1, scup - ws + scuw - 1, scup - ws + scuw, scup + ws * scuh - 1, scup - ws - 1 };
            const bool in[5] = { xs > 0, ys > 0, ys > 0 && xs + scuw < ws, xs > 0 && ys + scuh < hs, xs > 0 && ys > 0 };
            bool valid[5];
            uint32_t tl[5] = { 0, 0, 0, 0, 0 };
            for (int k = 0; k < 5; k++) { valid[k] = aff_nb(scup, nb[k], in[k], true); if (valid[k]) tl[k] = pic.aff_tl[(size_t)nb[k]]; }
            if (valid[2] && valid[1] && tl[1] == tl[2]) valid[2] = false;
            if (valid[3] && valid[0] && tl[0] == tl[3]) valid[3] = false;
            if ((valid[4] && valid[0] && tl[4] == tl[0]) || (valid[4] && valid[1] && tl[4] == tl[1])) valid[4] = false;
            for (int k = 0; k < 5 && cnt < 5; k++) {
                if (!valid[k]) continue;
                cpn[cnt] = (pic.aff[(size_t)nb[k]] & 3) == 1 ? 2 : 3;
                for (int l = 0; l < 2; l++) {
                    refi[cnt][l] = pic.refi[(size_t)nb[k] * 2 + l];
                    if (refi[cnt][l] >= 0) aff_inherit(cu, nb[k], l, cpn[cnt], cpmv[cnt][l]); else refi[cnt][l] = -1;
                }
                cnt++;
            }
        }
        {   // constructed from the vectors at the four corners
            int16_t cp_mv[2][4][2];
            int cp_refi[2][4], cp_valid[4] = { 0, 0, 0, 0 };
            memset(cp_mv, 0, sizeof(cp_mv));
            for (int l = 0; l < 2; l++) for (int i = 0; i < 4; i++) cp_refi[l][i] = -1;
            auto spatial = [&](const int *nb, const bool *in, int n, int v) {
                for (int k = 0; k < n; k++)
                    if (aff_nb(scup, nb[k], in[k], false)) {
                        for (int l = 0; l < 2; l++) {
                            cp_refi[l][v] = pic.refi[(size_t)nb[k] * 2 + l];
                            cp_mv[l][v][0] = pic.mv[(size_t)nb[k] * 4 + l * 2]; cp_mv[l][v][1] = pic.mv[(size_t)nb[k] * 4 + l * 2 + 1];
                        }
                        cp_valid[v] = 1;
                        return;
                    }
            };
            auto temporal = [&](int scu_col, int v) {          // the co-located vectors at an 8x8-aligned position, reference 0 of each list
                int16_t t[2][2];
                const int have = collocated(cu, scu_col, t);
                for (int l = 0; l < 2; l++) {
                    const bool on = (have >> l) & 1 && (l == 0 || sh.type == XHOST_SLICE_B);
                    cp_refi[l][v] = on ? 0 : -1;
                    cp_mv[l][v][0] = on ? t[l][0] : (int16_t)0; cp_mv[l][v][1] = on ? t[l][1] : (int16_t)0;
                }
            };
            const bool same_ctu_row = (((ys + scuh) << 2) >> 6) == ((ys << 2) >> 6);
            { const int nb[3] = { scup - ws - 1, scup - ws, scup - 1 }; const bool in[3] = { xs > 0 && ys > 0, ys > 0, xs > 0 }; spatial(nb, in, 3, 0); }
            { const int nb[3] = { scup - ws + scuw, scup - ws + scuw - 1, scup + scuw }; const bool in[3] = { ys > 0 && xs + scuw < ws, ys > 0, xs + scuw < ws }; spatial(nb, in, 3, 1); }
            if (left_avail) { const int nb[2] = { scup + ws * scuh - 1, scup + ws * (scuh - 1) - 1 }; const bool in[2] = { xs > 0 && ys + scuh < hs, xs > 0 }; spatial(nb, in, 2, 2); }
            else {
                if (xs > 0 && ys + scuh < hs && same_ctu_row && pic.same_tile(scup, scup + ws * scuh - 1) && pic.same_tile(scup, scup - 1))
                    temporal(((xs - 1) >> 1 << 1) + ((ys + scuh) >> 1 << 1) * ws, 2);
                if (cp_refi[0][2] >= 0 || cp_refi[1][2] >= 0) cp_valid[2] = 1;
            }
            {
                const int col = ((xs + scuw) >> 1 << 1) + ((ys + scuh) >> 1 << 1) * ws;
                if (xs + scuw < ws && ys + scuh < hs && same_ctu_row && pic.same_tile(scup, col)) temporal(col, 3);
                if (cp_refi[0][3] >= 0 || cp_refi[1][3] >= 0) cp_valid[3] = 1;
            }
            static const int models[6][3] = { { 0, 1, 2 }, { 0, 1, 3 }, { 0, 2, 3 }, { 1, 2, 3 }, { 0, 1, 0 }, { 0, 2, 0 } };
            static const int vns[6] = { 3, 3, 3, 3, 2, 2 };
            for (int m = 0; m < 6; m++) aff_constructed(cu, cp_valid, cp_mv, cp_refi, models[m], m, vns[m], refi, cpmv, cpn, cnt);
        }
        for (; cnt < 5; cnt++) { cpn[cnt] = 2; refi[cnt][0] = 0; refi[cnt][1] = sh.type == XHOST_SLICE_B ? 0 : -1; }      // zero candidates
    }
    // affine merge: candidate idx decides the model and the references
    void aff_merge_motion(Cu &cu, int idx) const
    {
        int8_t refi[5][2]; int16_t cpmv[5][2][3][2]; int cpn[5];
        aff_merge(cu, refi, cpmv, cpn);
        cu.affine = cpn[idx] - 1;
        memset(cu.aff_mv, 0, sizeof(cu.aff_mv));
        for (int l = 0; l < 2; l++) {
            cu.refi[l] = refi[idx][l] >= 0 ? refi[idx][l] : -1;
            cu.mv[l][0] = cu.mv[l][1] = 0;
            if (cu.refi[l] >= 0) for (int v = 0; v < cpn[idx]; v++) { cu.aff_mv[l][v][0] = cpmv[idx][l][v][0]; cu.aff_mv[l][v][1] = cpmv[idx][l][v][1]; }
        }
    }
    // what the history keeps of an affine CU: the model's vector at the CU centre (update_history_buffer_parse_affine, xevdm.c:657-796)
    void aff_centre(const Cu &cu, int l, int16_t out[2]) const
    {
        const int16_t (*m)[2] = cu.aff_mv[l];
        const int dhx = (m[1][0] - m[0][0]) * (1 << (7 - cu.log2w)), dhy = (m[1][1] - m[0][1]) * (1 << (7 - cu.log2w));
        int dvx = -dhy, dvy = dhx;
        if (cu.affine == 2) { dvx = (m[2][0] - m[0][0]) * (1 << (7 - cu.log2h)); dvy = (m[2][1] - m[0][1]) * (1 << (7 - cu.log2h)); }
        const int px = 1 << (cu.log2w - 1), py = 1 << (cu.log2h - 1);
        out[0] = clip16(aff_rnd(m[0][0] * 128 + dhx * px + dvx * py, 7)); out[1] = clip16(aff_rnd(m[0][1] * 128 + dhy * px + dvy * py, 7));
    }
    // the vectors later CUs and pictures see of an affine CU: one per sub-block, the control-point vectors themselves at the corners
    // (xevdm_set_affine_mvf, xevdm_util.c:4095-4195; sub-block size xevdm_derive_affine_subblock_size_bi :1870-1945)
    void aff_store(const Cu &cu)
    {
        const int ws = pic.w_scu, w_cu = (1 << cu.log2w) >> 2, h_cu = (1 << cu.log2h) >> 2, scup = (cu.y >> 2) * ws + (cu.x >> 2), vn = cu.affine + 1;
        AffModel m[2];
        const bool use[2] = { cu.refi[0] >= 0, cu.refi[1] >= 0 };
        for (int l = 0; l < 2; l++) if (use[l]) m[l] = aff_model(&cu.aff_mv[l][0][0], cu.log2w, cu.log2h, vn);
        int sub_w, sub_h; bool mem_band;
        aff_subblock(m, use, cu.log2w, cu.log2h, sub_w, sub_h, mem_band);
        const int sw = sub_w >> 2, shh = sub_h >> 2;
        for (int l = 0; l < 2; l++) {
            if (!use[l]) continue;
            const int16_t (*mv)[2] = cu.aff_mv[l];
            for (int h = 0; h < h_cu; h += shh) for (int w = 0; w < w_cu; w += sw) {
                int vx, vy;
                if (w == 0 && h == 0) { vx = mv[0][0]; vy = mv[0][1]; }
                else if (w + sw == w_cu && h == 0) { vx = mv[1][0]; vy = mv[1][1]; }
                else if (w == 0 && h + shh == h_cu && vn == 3) { vx = mv[2][0]; vy = mv[2][1]; }
                else {
                    const int px = (w << 2) + (sub_w >> 1), py = (h << 2) + (sub_h >> 1);
                    vx = aff_clip18(aff_round(mv[0][0] * 128 + m[l].dh[0] * px + m[l].dv[0] * py, 5)) >> 2;
                    vy = aff_clip18(aff_round(mv[0][1] * 128 + m[l].dh[1] * px + m[l].dv[1] * py, 5)) >> 2;
                }
                for (int y = h; y < h + shh; y++) for (int x = w; x < w + sw; x++) {
                    const size_t k = (size_t)scup + (size_t)y * ws + x;
                    pic.mv[k * 4 + l * 2] = (int16_t)vx; pic.mv[k * 4 + l * 2 + 1] = (int16_t)vy;
                    if (!pic.mv_ref.empty()) { pic.mv_ref[k * 4 + l * 2] = (int16_t)vx; pic.mv_ref[k * 4 + l * 2 + 1] = (int16_t)vy; }      // map_mv and map_unrefined_mv agree outside refined CUs
                }
            }
        }
        const uint8_t tag = (uint8_t)(cu.affine | (cu.log2w << 2) | (cu.log2h << 5));
        for (int y = 0; y < h_cu; y++) for (int x = 0; x < w_cu; x++) { pic.aff[(size_t)scup + (size_t)y * ws + x] = tag; pic.aff_tl[(size_t)scup + (size_t)y * ws + x] = (uint32_t)scup; }
    }
    // SCU maps after a CU (xevd_set_dec_info, xevd_util.c:1574-1660; cod_eco xevd.c:797-803)
    std::vector<int16_t> dmvr_scratch;
    bool missing_ref = false;                            // host-side DMVR needed the samples of a reference picture the caller has not registered
    void commit(const Cu &cu)
    {
        if (cu.tree == 2) return;                        // a chroma-only CU leaves every map as its luma CUs wrote it (xevdm_set_dec_info, xevdm_util.c:4241)
        const int xs = cu.x >> 2, ys = cu.y >> 2, w = (1 << cu.log2w) >> 2, h = (1 << cu.log2h) >> 2;
        // Host-side DMVR (Sps::host_dmvr): the refinement search of a merge-mode bi-predicted CU runs HERE, because the refined vectors are state of this
        // picture's parse - the refined map below (MMVD merge lists, temporal candidates of later pictures) and the history buffer
        int16_t refined[64][2][2];
        bool is_refined = false;
        if (sps.host_dmvr() && cu.dmvr && (cu.mode == MODE_INTER || cu.mode == MODE_SKIP) && !cu.affine && cu.refi[0] >= 0 && cu.refi[1] >= 0 &&
            cu.refi[0] < (int)refp[0].size() && cu.refi[1] < (int)refp[1].size()) {
            const RefPic *r0 = refp[0][(size_t)cu.refi[0]], *r1 = refp[1][(size_t)cu.refi[1]];
            if (dmvr_search_applies(poc, r0->poc, r1->poc, 1 << cu.log2w, 1 << cu.log2h)) {
                if (!r0->luma || !r1->luma) missing_ref = true;
                else {
                    const DmvrRefPlane rp[2] = { { r0->luma, r0->luma_stride, r0->poc }, { r1->luma, r1->luma_stride, r1->poc } };
                    dmvr_search_cu(pic.w_scu << 2, pic.h_scu << 2, sps.bd_l, cu.x, cu.y, 1 << cu.log2w, 1 << cu.log2h, cu.mv, rp, refined, dmvr_scratch);
                    is_refined = true;
                }
            }
        }
        if (sps.tool_hmvp && (cu.mode == MODE_INTER || cu.mode == MODE_SKIP)) {
            if (is_refined) {                            // core->mv = map_mv[first SCU] before the history update (xevdm_util.c:4384-4387, xevdm.c:1335-1342)
                Cu first = cu;
                memcpy(first.mv, refined[0], sizeof(first.mv));
                history_push(first);
            } else history_push(cu);
        }
        if (!pic.mv_ref.empty()) {
            const int sbw = std::min(w, 4), sbh = std::min(h, 4), per_row = w / sbw;      // 16x16 sub-blocks in SCUs
            for (int r = 0; r < h; r++) for (int c = 0; c < w; c++) {
                const size_t k = (size_t)(ys + r) * pic.w_scu + xs + c;
                const int16_t (*v)[2] = is_refined ? refined[(r / sbh) * per_row + c / sbw] : cu.mv;
                for (int l = 0; l < 2; l++) { pic.mv_ref[k * 4 + l * 2] = v[l][0]; pic.mv_ref[k * 4 + l * 2 + 1] = v[l][1]; }
            }
        }
        for (int r = 0; r < h; r++) for (int c = 0; c < w; c++) {
            const size_t k = (size_t)(ys + r) * pic.w_scu + xs + c;
            pic.cod[k] = 1; pic.intra[k] = cu.mode == MODE_INTRA; pic.ibc[k] = cu.mode == MODE_IBC; pic.ipm[k] = (int8_t)cu.ipm;
            if (!pic.skip.empty()) pic.skip[k] = cu.mode == MODE_SKIP;
            if (!pic.cu_size.empty()) pic.cu_size[k] = (uint8_t)(cu.log2w | (cu.log2h << 4));
            for (int l = 0; l < 2; l++) { pic.refi[k * 2 + l] = (int8_t)cu.refi[l]; pic.mv[k * 4 + l * 2] = cu.mv[l][0]; pic.mv[k * 4 + l * 2 + 1] = cu.mv[l][1]; }
        }
        if (cu.affine && cu.mode != MODE_INTRA && cu.mode != MODE_IBC) aff_store(cu);
    }

    // ---- coefficient block, run-length coding in zig-zag order (xevd_eco_run_length_cc, xevd_eco.c:343-395) ----
    template <class C> void code_coefs(C &c, int16_t *coef, int log2w, int log2h, int chroma, bool enc)
    {
        const std::vector<uint16_t> &sc = scan[log2w - 1][log2h - 1];
        const int n = 1 << (log2w + log2h);
        int pos = 0, prev_level = 6;
        for (;;) {
            int run = 0, level = 1, sign = 0, last = 1;
            // context pair of run and level: with sps->tool_cm_init by the level before (xevdm_eco.c:319)
            const int t0 = sps.tool_cm_init ? (std::min(prev_level - 1, 5) << 1) + (chroma ? 12 : 0) : (chroma ? 2 : 0);
            if (enc) {
                while (pos + run < n && coef[sc[pos + run]] == 0) run++;
                const int v = coef[sc[pos + run]];
                level = v < 0 ? -v : v; sign = v < 0;
                for (int q = pos + run + 1; q < n; q++) if (coef[sc[q]]) { last = 0; break; }
            }
            run = sym_unary(c, run, models.run + t0, 2);
            if (!enc) for (int i = pos; i < pos + run && i < n; i++) coef[sc[i]] = 0;
            pos += run;
            if (pos >= n) return;                                   // malformed input; the caller checks the reader's overrun flag
            level = sym_unary(c, level - 1, models.level + t0, 2) + 1;
            prev_level = level;
            sign = c.ep(sign);
            if (!enc) coef[sc[pos]] = (int16_t)(sign ? -level : level);
            if (pos >= n - 1) break;
            pos++;
            last = c.bin(last, models.last[chroma ? 1 : 0]);
            if (last) break;
        }
    }


    // ---- coefficient block with sps->tool_adcc (xevdm_eco_adcc, src_main/xevdm_eco.c:482-689): position of the last coefficient in scan order, then per
    //      group of 16 scan positions (last group first): significance flags, greater-than-1 flags of the first 8 coefficients, one greater-than-2 flag,
    //      Golomb-Rice remainders, signs.  Contexts from the five already-coded neighbours to the right and below; the block itself is the working state ----
    static int adcc_nb(const int16_t *coef, int blkpos, int width, int height, int log2w, int what, int base = 0)
    {
        const int16_t *p = coef + blkpos;
        const int py = blkpos >> log2w, px = blkpos - (py << log2w);
        int n = 0;
        auto take = [&](int v) { const int a = v < 0 ? -v : v; n += what == 0 ? v != 0 : what == 1 ? a > 1 : what == 2 ? a > 2 : a; };
        if (px < width - 1) { take(p[1]); if (px < width - 2) take(p[2]); if (py < height - 1) take(p[width + 1]); }
        if (py < height - 1) { take(p[width]); if (py < height - 2) take(p[2 * width]); }
        (void)base;
        return n;
    }
    template <class C> void code_adcc(C &c, int16_t *coef, int log2w, int log2h, int chroma, bool enc)
    {
        static const int group_idx[64] = { 0, 1, 2, 3, 4, 4, 5, 5, 6, 6, 6, 6, 7, 7, 7, 7, 8, 8, 8, 8, 8, 8, 8, 8, 9, 9, 9, 9, 9, 9, 9, 9, 10, 10, 10, 10, 10, 10, 10, 10, 10, 10, 10, 10, 10,
                                           10, 10, 10, 11, 11, 11, 11, 11, 11, 11, 11, 11, 11, 11, 11, 11, 11, 11, 11 };      // g_group_idx, xevdm_tbl.c:390
        static const int min_in_group[14] = { 0, 1, 2, 3, 4, 6, 8, 12, 16, 24, 32, 48, 64, 96 };
        static const int rice_range[10] = { 6, 5, 6, 3, 3, 3, 3, 3, 3, 3 };
        static const int rice_para[32] = { 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 1, 1, 1, 2, 2, 2, 2, 2, 2, 2, 2, 2, 2, 2, 2, 2, 2, 3, 3, 3, 3 };
        const int width = 1 << log2w, height = 1 << log2h, n = width * height;
        const std::vector<uint16_t> &sc = scan[log2w - 1][log2h - 1], &inv = scan_inv[log2w - 1][log2h - 1];
        std::vector<int16_t> want;
        int last_x = 0, last_y = 0;
        if (enc) {
            want.assign(coef, coef + n);
            int last = 0;
            for (int i = 0; i < n; i++) if (want[sc[(size_t)i]]) last = i;
            last_x = sc[(size_t)last] & (width - 1); last_y = sc[(size_t)last] >> log2w;
            memset(coef, 0, sizeof(int16_t) * (size_t)n);
        }
        {   // last_sig_coeff_{x,y}_prefix / _suffix (xevdm_parse_positionLastXY :395-457; context sets by block size, xevd_get_ctx_last_pos_xy_para)
            auto para = [&](int size, int log2s, int &off, int &shift) {
                const int cv = std::max(log2s - 2, 0);
                if (chroma) { off = 0; shift = cv - (size >= 16 ? log2s - 4 : 0); }      // cv - log2(size >> 4), the reference's log2 table maps 0 to 0
                else { off = cv * 3 + ((cv + 1) >> 2); shift = (cv + 3) >> 2; if (cv >= 4) { off += ((size >> 6) << 1) + (size >> 7); shift = 2; } }
            };
            int off[2], shift[2], pos[2] = { last_x, last_y };
            para(width, log2w, off[0], shift[0]); para(height, log2h, off[1], shift[1]);
            Model *cm[2] = { models.last_x + (chroma ? 18 : 0), models.last_y + (chroma ? 18 : 0) };
            int grp[2];
            for (int d = 0; d < 2; d++) {
                const int gmax = group_idx[(d ? height : width) - 1], g = enc ? group_idx[pos[d]] : 0;
                int k = 0;
                for (; k < gmax; k++) if (!c.bin(enc ? k < g : 0, cm[d][off[d] + (k >> shift[d])])) break;
                grp[d] = k;
            }
            for (int d = 0; d < 2; d++) {
                if (grp[d] > 3) { const int cnt = (grp[d] - 2) >> 1; pos[d] = min_in_group[grp[d]] + sym_bits_ep(c, enc ? pos[d] - min_in_group[grp[d]] : 0, cnt); }
                else pos[d] = grp[d];
            }
            last_x = std::min(pos[0], width - 1); last_y = std::min(pos[1], height - 1);
        }
        const int num_coeff = inv[(size_t)(last_x + last_y * width)] + 1, scan_pos_last = num_coeff - 1;
        const int log2_min = std::min(log2w, log2h);
        Model *cm_sig = chroma ? models.sig_coeff + 39 : models.sig_coeff + (log2_min <= 2 ? 0 : 13 << std::min(1, log2_min - 3));
        Model *cm_gt = chroma ? models.gt_ab + 13 : models.gt_ab;
        int ipos = scan_pos_last, pos_last = -1, ctx_gta = 0, ctx_gtb = 0;
        for (int sub = scan_pos_last >> 4; sub >= 0; sub--) {
            int num_nz = 0, pos[16], abs_coef[16];
            for (; ipos >= (sub << 4); ipos--) {
                const int blk = sc[(size_t)ipos];
                int sig = 1;
                if (ipos != scan_pos_last) {
                    const int py = blk >> log2w, px = blk - (py << log2w), diag = px + py;
                    int idx = std::min(adcc_nb(coef, blk, width, height, log2w, 0), 4) + 1;
                    if (diag < 2) idx = std::min(idx, 2);
                    const int ofs = chroma ? (diag < 2 ? 0 : 2) : (diag < 2 ? 0 : diag < 5 ? 2 : 7);
                    sig = c.bin(enc ? want[(size_t)blk] != 0 : 0, cm_sig[ofs + idx]);
                }
                coef[blk] = (int16_t)sig;
                if (sig) { pos[num_nz++] = blk; if (pos_last < 0) pos_last = blk; }
            }
            if (!num_nz) continue;
            auto gt_ctx = [&](int blk, int what) {
                const int py = blk >> log2w, px = blk - (py << log2w), diag = px + py;
                int v = std::min(adcc_nb(coef, blk, width, height, log2w, what), 3) + 1;
                if (!chroma) v += diag < 3 ? 0 : diag < 10 ? 4 : 8;
                return v;
            };
            auto true_abs = [&](int blk) { const int v = want[(size_t)blk]; return v < 0 ? -v : v; };
            bool escape = false;
            int first_c2 = -1;
            for (int i = 0; i < num_nz; i++) abs_coef[i] = 1;
            for (int i = 0; i < std::min(num_nz, 8); i++) {
                if (pos[i] != pos_last) ctx_gta = gt_ctx(pos[i], 1);
                const int f = c.bin(enc ? true_abs(pos[i]) > 1 : 0, cm_gt[ctx_gta]);
                coef[pos[i]] = (int16_t)(coef[pos[i]] + f);
                abs_coef[i] = f + 1;
                if (f) { if (first_c2 < 0) first_c2 = i; else escape = true; }
            }
            if (first_c2 >= 0) {
                if (pos[first_c2] != pos_last) ctx_gtb = gt_ctx(pos[first_c2], 2);
                const int f = c.bin(enc ? true_abs(pos[first_c2]) > 2 : 0, cm_gt[ctx_gtb]);
                coef[pos[first_c2]] = (int16_t)(coef[pos[first_c2]] + f);
                abs_coef[first_c2] = f + 2;
                if (f) escape = true;
            }
            escape = escape || num_nz > 8;
            int first_coeff2 = 1;
            if (escape)
                for (int i = 0; i < num_nz; i++) {
                    const int base = i < 8 ? 2 + first_coeff2 : 1;
                    if (abs_coef[i] >= base) {
                        const int r = rice_para[std::max(std::min(adcc_nb(coef, pos[i], width, height, log2w, 3) - 5 * base, 31), 0)];
                        // xevdm_parse_coef_remain_exgolomb (:458-481): unary prefix, then r bins - or, past the prefix limit of r, an escape with growing suffix
                        int sym = enc ? true_abs(pos[i]) - base : 0, prefix = 0;
                        if (enc) {
                            if (sym < (rice_range[r] << r)) prefix = sym >> r;
                            else { int k = 0; while ((((1 << (k + 1)) + rice_range[r] - 1) << r) <= sym) k++; prefix = rice_range[r] + k; }
                        }
                        int k = 0;
                        while (c.ep(enc ? k < prefix : 0)) { if (++k > 32) break; }
                        prefix = k;
                        if (prefix < rice_range[r]) sym = (prefix << r) + sym_bits_ep(c, sym & ((1 << r) - 1), r);
                        else {
                            const int e = prefix - rice_range[r], b0 = ((1 << e) + rice_range[r] - 1) << r;
                            sym = b0 + sym_bits_ep(c, sym - b0, std::min(e + r, 30));
                        }
                        abs_coef[i] = sym + base;
                        coef[pos[i]] = (int16_t)std::min(abs_coef[i], 32767);
                    }
                    if (abs_coef[i] >= 2) first_coeff2 = 0;
                }
            uint32_t signs = 0;
            if (enc) for (int i = 0; i < num_nz; i++) signs = (signs << 1) | (want[(size_t)pos[i]] < 0);
            signs = (uint32_t)sym_bits_ep(c, (int)signs, num_nz);
            for (int i = 0; i < num_nz; i++) {
                const int neg = (signs >> (num_nz - 1 - i)) & 1, a = std::min(abs_coef[i], 32767);
                coef[pos[i]] = (int16_t)(neg ? -a : a);
            }
        }
    }

    // ---- one CU: syntax (xevd_eco_cu, xevd_eco.c:1048-1176; cbf :260-341; coefficients/QP :593-767) + derivations ----
    // enc: `cu` and `coef` carry the wanted values (mv of an INTER CU is met through mvd, a SKIP CU takes its predictor's motion);
    // dec: they are filled.  coef[c]: w*h (w/2*h/2) values of component c, zero-initialised by the caller when decoding.
    // ---- sps_btt_flag: which splits a node may take (xevdm_check_split_mode, src_main/xevdm_util.c:1575-1680).  Index = split mode: 0 none, 1 binary with a
    //      vertical cut, 2 binary horizontal, 3 ternary vertical (1/4, 1/2, 1/4), 4 ternary horizontal; there is no quad split with BTT.  Shapes are limited
    //      through the SPS table of long sides per aspect ratio; a node crossing the picture border takes a binary split towards it ----
    enum { NO_SPLIT = 0, BI_VER = 1, BI_HOR = 2, TRI_VER = 3, TRI_HOR = 4, QUAD = 5 };
    bool ratio_ok(int long_side, int ratio) const { return ratio <= 2 && long_side <= sps.split_tbl[ratio][1] && long_side >= sps.split_tbl[ratio][0]; }
    bool tri_ok(int long_side) const { return long_side <= sps.split_tbl[3][1] && long_side >= sps.split_tbl[3][0]; }
    static bool small_child_is_4x4(int split, int w, int h) { if (split == BI_HOR) h >>= 1; else if (split == BI_VER) w >>= 1; else if (split == TRI_HOR) h >>= 2; else w >>= 2; return w == 4 && h == 4; }
    static bool chroma_split_ok(int split, int w, int h) { if (split == BI_HOR) h >>= 1; else if (split == BI_VER) w >>= 1; else if (split == TRI_HOR) h >>= 2; else w >>= 2; return w * h >= 64; }
    void split_allowed(int allow[6], int lw, int lh, int x, int y, bool only_inter) const
    {
        const int W = sps.width, H = sps.height, w = 1 << lw, h = 1 << lh;
        const bool boundary = !(x + w <= W && y + h <= H), boundary_r = boundary && x + w > W && !(y + h > H);
        const bool from_boundary_b = y >= H - H % 32 && !(x >= W - W % 32);      // cu_max = half a CTU
        for (int i = 0; i < 6; i++) allow[i] = 0;
        allow[NO_SPLIT] = 1;                     // (the reference leaves this entry unset and never reads it for a node inside the picture)
        const bool tv = tri_ok(lw) && (lw > lh || (lw == lh && ratio_ok(lw, 2))), th = tri_ok(lh) && (lh > lw || (lw == lh && ratio_ok(lh, 2)));
        if (lw == lh) { allow[BI_HOR] = allow[BI_VER] = ratio_ok(lw, 1); }
        else if (lw > lh) {
            allow[BI_HOR] = ratio_ok(lw, lw - lh + 1);
            const int sw = lw - 1, ratio = sw > lh ? sw - lh : lh - sw;
            allow[BI_VER] = ratio_ok(std::max(sw, lh), ratio) || (from_boundary_b && (ratio == 3 || ratio == 4));
        } else {
            const int sh_ = lh - 1, ratio = lw > sh_ ? lw - sh_ : sh_ - lw;
            allow[BI_HOR] = ratio_ok(std::max(lw, sh_), ratio);
            allow[BI_VER] = ratio_ok(lh, lh - lw + 1);
        }
        allow[TRI_VER] = tv; allow[TRI_HOR] = th;
        if (boundary) {
            allow[NO_SPLIT] = allow[TRI_VER] = allow[TRI_HOR] = 0;
            if (boundary_r) allow[BI_HOR] = !allow[BI_VER]; else allow[BI_VER] = !allow[BI_HOR];
        }
        if (only_inter) for (int m = BI_VER; m <= TRI_HOR; m++) allow[m] = allow[m] && !small_child_is_4x4(m, w, h);
    }
    // split syntax of a node inside the picture (xevdm_eco_split_mode, src_main/xevdm_eco.c:1173-1296): btt_split_flag, direction, type - each only when
    // both alternatives are allowed; with tool_cm_init the flag's context counts the neighbours (above / left) that are narrower / lower than the node
    template <class C> int code_split(C &c, int want, int x, int y, int lw, int lh, bool only_inter)
    {
        if (lw < 3 && lh < 3) return NO_SPLIT;
        int allow[6];
        split_allowed(allow, lw, lh, x, y, only_inter);
        if (!(allow[BI_VER] || allow[BI_HOR] || allow[TRI_VER] || allow[TRI_HOR])) return NO_SPLIT;
        int ctx = 0;
        if (sps.tool_cm_init) {
            static const uint8_t shape_ctx[6][6] = { { 255, 4, 4, 14, 15, 15 }, { 4, 4, 3, 3, 2, 2 }, { 4, 3, 3, 2, 2, 1 }, { 14, 3, 2, 2, 1, 1 }, { 15, 2, 2, 1, 1, 0 }, { 15, 2, 1, 1, 0, 0 } };      // xevd_tbl_split_flag_ctx
            const int ws = pic.w_scu, xs = x >> 2, ys = y >> 2, scup = ys * ws + xs;
            int smaller = 0;
            if (ys > 0 && pic.same_tile(scup, scup - ws)) smaller += (1 << (pic.cu_size[(size_t)scup - ws] & 15)) < (1 << lw);                                  // above: parsed whenever it is in the tile
            if (xs > 0 && pic.same_tile(scup, scup - 1) && pic.cod[(size_t)scup - 1]) smaller += (1 << (pic.cu_size[(size_t)scup - 1] >> 4)) < (1 << lh);
            ctx = std::min(smaller, 2) + 3 * shape_ctx[lw - 2][lh - 2];
            if (ctx > 14) ctx = 14;
        }
        if (!c.bin(want != NO_SPLIT, models.btt_split_flag[ctx])) return NO_SPLIT;
        const bool ver_ok = allow[BI_VER] || allow[TRI_VER], hor_ok = allow[BI_HOR] || allow[TRI_HOR];
        int dir = want == BI_VER || want == TRI_VER;
        if (ver_ok && hor_ok) dir = c.bin(dir, models.btt_split_dir[sps.tool_cm_init ? lw - lh + 2 : 0]); else dir = ver_ok;
        int tri = want == TRI_VER || want == TRI_HOR;
        if ((dir && allow[BI_VER] && allow[TRI_VER]) || (!dir && allow[BI_HOR] && allow[TRI_HOR])) tri = c.bin(tri, models.btt_split_type[0]);
        else tri = (dir && allow[TRI_VER]) || (!dir && allow[TRI_HOR]);
        return tri ? (dir ? TRI_VER : TRI_HOR) : (dir ? BI_VER : BI_HOR);
    }
    // children of a split node: position and size
    static int split_parts(int split, int x, int y, int lw, int lh, int px[3], int py[3], int plw[3], int plh[3])
    {
        const int n = split == BI_VER || split == BI_HOR ? 2 : 3, ver = split == BI_VER || split == TRI_VER;
        int off = 0;
        for (int i = 0; i < n; i++) {
            const int shrink = n == 2 ? 1 : (i == 1 ? 1 : 2);
            plw[i] = ver ? lw - shrink : lw; plh[i] = ver ? lh : lh - shrink;
            px[i] = ver ? x + off : x; py[i] = ver ? y : y + off;
            off += 1 << (ver ? plw[i] : plh[i]);
        }
        return n;
    }
    // the mode constraint a split node hands to its children with sps_btt_flag and tool_admvp (xevd_entropy_decode_tree, src_main/xevdm.c:1775-1802):
    // 0 none, 1 inter only (signalled), -1 intra only = a local dual tree (luma CUs, then one chroma CU), which the batch format cannot express
    template <class C> int code_mode_cons(C &c, int split, int lw, int lh, bool only_inter, int want_only_inter)
    {
        if (!(sps.btt && sps.tool_admvp) || only_inter) return only_inter ? 1 : 0;
        const int w = 1 << lw, h = 1 << lh;
        if (chroma_split_ok(split, w, h)) return 0;
        if (sh.type == XHOST_SLICE_I || small_child_is_4x4(split, w, h)) return -1;
        return c.bin(!want_only_inter, models.mode_cons[0]) ? -1 : 1;      // the flag says "intra only"; its context counts nothing (always 0)
    }

    // sps->tool_cm_init: the contexts of skip_flag / pred_mode_flag / ibc_flag / affine_flag count the neighbours that have the property - above the top-left
    // SCU, left of the bottom-left one (and right of the bottom-right one, never parsed before the CU without SUCO) - in the same tile and already parsed
    // (xevdm_get_ctx_some_flags, src_main/xevdm_util.c:1729-1853)
    enum { CTX_SKIP, CTX_PRED, CTX_IBC, CTX_AFF };
    int nb_ctx(const Cu &cu, int what) const
    {
        if (!sps.tool_cm_init) return 0;
        const int ws = pic.w_scu, xs = cu.x >> 2, ys = cu.y >> 2, scuh = (1 << cu.log2h) >> 2, scup = ys * ws + xs;
        const int nb[2] = { scup - ws, scup - 1 + (scuh - 1) * ws };
        const bool in[2] = { ys > 0, xs > 0 };
        int n = 0;
        for (int k = 0; k < 2; k++) {
            if (!in[k] || !pic.same_tile(scup, nb[k]) || !pic.cod[(size_t)nb[k]]) continue;
            n += what == CTX_SKIP ? pic.skip[(size_t)nb[k]] : what == CTX_PRED ? pic.intra[(size_t)nb[k]] : what == CTX_IBC ? pic.ibc[(size_t)nb[k]] : (!pic.aff.empty() && pic.aff[(size_t)nb[k]] != 0);
        }
        return std::min(n, what == CTX_PRED ? 2 : 1);
    }
    template <class C> int code_refi(C &c, int want, int nref)      // xevd_eco_refi, xevd_eco.c:409-436
    {
        if (nref <= 1) return 0;
        const int r = std::min(std::max(want, 0), nref - 1);
        int v = 0;
        if (c.bin(r > 0, models.refi[0])) {
            v = 1;
            if (nref > 2 && c.bin(r > 1, models.refi[1])) {
                v = 2;
                for (; v < nref - 1; v++) if (!c.ep(r > v)) break;
            }
        }
        return v;
    }
    template <class C> void code_mvd(C &c, int16_t mvd[2])          // xevd_eco_get_mvd, xevd_eco.c:491-536
    {
        for (int d = 0; d < 2; d++) {
            const int v = mvd[d], a = sym_abs_mvd(c, v < 0 ? -v : v, models.mvd[0]);
            int sg = v < 0;
            if (a) sg = c.ep(sg);
            mvd[d] = (int16_t)(sg ? -a : a);
        }
    }
    // affine_flag + affine merge index of a skip / merge-mode CU of at least 8x8 (xevdm_eco.c:1528-1537, 1622-1632); true: the CU is affine
    template <class C> bool code_affine_merge(C &c, Cu &cu)
    {
        int aff = 0;
        if (sps.tool_affine && cu.log2w >= 3 && cu.log2h >= 3) aff = c.bin(cu.affine != 0, models.affine_flag[nb_ctx(cu, CTX_AFF)]);
        if (!aff) { cu.affine = 0; return false; }
        cu.aff_idx[0] = sym_trunc_unary(c, cu.aff_idx[0], models.affine_mrg, 5, 5);
        aff_merge_motion(cu, cu.aff_idx[0]);
        cu.dmvr = 0;
        return true;
    }
    // the luma intra mode a chroma-only CU refers to: the one stored at the centre of its block (xevd_get_luma_cup, xevd_util.c:1481); Baseline modes: that CU is intra
    // by constraint; EIPD: DC when it is not (an IBC CU)
    int luma_mode_of(const Cu &cu) const
    {
        const size_t k = (size_t)((cu.y >> 2) + ((1 << cu.log2h) >> 3)) * pic.w_scu + (cu.x >> 2) + ((1 << cu.log2w) >> 3);
        return pic.intra[k] ? pic.ipm[k] : 0;
    }
    template <class C> void code_cu(C &c, Cu &cu, int16_t *coef[3], bool enc)
    {
        // mode constraint eOnlyIntra: I slices, and with tool_admvp every 4x4 CU (xevdm.c:1838-1843) - no skip flag, no pred_mode_flag
        // ... and the CUs of a local dual tree (TREE_L / TREE_C)
        const bool inter_slice = sh.type != XHOST_SLICE_I && !(sps.tool_admvp && cu.log2w == 2 && cu.log2h == 2) && cu.tree == 0;
        const int keep_only_inter = cu.only_inter;
        int skip = 0;
        if (inter_slice) skip = c.bin(cu.mode == MODE_SKIP, models.skip[nb_ctx(cu, CTX_SKIP)]);
        if (!enc) { cu.mode = skip ? MODE_SKIP : MODE_INTRA; cu.refi[0] = cu.refi[1] = -1; memset(cu.mv, 0, sizeof(cu.mv)); memset(cu.mvd, 0, sizeof(cu.mvd));
                    cu.mvp_idx[0] = cu.mvp_idx[1] = 0; cu.cbf[0] = cu.cbf[1] = cu.cbf[2] = 0; cu.ipm = cu.ipm_c = 0; cu.ats = cu.ats_inter = 0; cu.dmvr = 0; cu.mmvd = cu.mmvd_idx = 0; cu.affine = 0; memset(cu.aff_mv, 0, sizeof(cu.aff_mv)); memset(cu.aff_mvd, 0, sizeof(cu.aff_mvd)); cu.aff_idx[0] = cu.aff_idx[1] = 0; }
        int16_t cand[4][2];
        const int n_lists = sh.type == XHOST_SLICE_B ? 2 : 1;
        if (skip && sps.tool_admvp) {
            // Main: one merge index, truncated unary over five contexts (xevdm_eco_merge_idx, xevdm_eco.c:731-744; call site :1550-1551)
            if (sps.tool_mmvd) cu.mmvd = c.bin(cu.mmvd, models.mmvd_flag[0]);
            if (cu.mmvd) { cu.affine = 0; code_mmvd_idx(c, cu); mmvd_motion(cu); }
            else if (code_affine_merge(c, cu)) { }
            else {
                cu.mvp_idx[0] = cu.mvp_idx[1] = sym_trunc_unary(c, cu.mvp_idx[0], models.merge_idx, 5, 6);
                merge_motion(cu, cu.mvp_idx[0]);
                cu.dmvr = sps.tool_dmvr;
            }
            cu.cbf[0] = cu.cbf[1] = cu.cbf[2] = 0;
            cu.qp = qp_prev;
            return;
        }
        if (skip) {
            // the motion of candidate mvp_idx of every list, reference 0 (xevd_get_skip_motion, xevd.c:502-531; syntax xevd_eco.c:1079-1085)
            for (int l = 0; l < n_lists; l++) cu.mvp_idx[l] = sym_trunc_unary(c, cu.mvp_idx[l], models.mvp_idx, 3, 4);
            cu.refi[1] = -1; cu.mv[1][0] = cu.mv[1][1] = 0;
            for (int l = 0; l < n_lists; l++) {
                mvp_candidates(cu, l, cand);
                cu.refi[l] = 0; cu.mv[l][0] = cand[cu.mvp_idx[l]][0]; cu.mv[l][1] = cand[cu.mvp_idx[l]][1];
            }
            cu.cbf[0] = cu.cbf[1] = cu.cbf[2] = 0;
            cu.qp = qp_prev;                                         // xevd_eco.c:1091-1115 (cu_qp_delta on: previous QP; off: slice QP = the same)
            return;
        }
        int intra = 1;
        if (inter_slice && keep_only_inter) intra = 0;                 // eOnlyInter: no pred_mode_flag (xevdm_eco_pred_mode, xevdm_eco.c:1401-1438)
        else if (inter_slice) intra = c.bin(cu.mode == MODE_INTRA, models.pred_mode[nb_ctx(cu, CTX_PRED)]);
        // xevdm_eco_pred_mode (xevdm_eco.c:1401-1438): with sps->ibc_flag every CU up to the IBC size limit that is not already known to be
        // intra-predicted carries ibc_flag - in I slices all of them (mode constraint eOnlyIntra: no pred_mode_flag); context 0 without cm_init
        int ibc = 0;
        if (sps.ibc && cu.log2w <= sps.ibc_log_max && cu.log2h <= sps.ibc_log_max && !(inter_slice && intra) && !keep_only_inter && cu.tree != 2)
            ibc = c.bin(cu.mode == MODE_IBC, models.ibc_flag[nb_ctx(cu, CTX_IBC)]);
        if (!enc) { cu.mode = ibc ? MODE_IBC : intra ? MODE_INTRA : MODE_INTER; cu.direct = 0; }
        if (ibc) {
            // the block vector itself is sent as a motion vector difference (xevdm_eco.c:1789-1800); no references, no predictor
            intra = 0;
            cu.refi[0] = cu.refi[1] = -1; cu.mv[1][0] = cu.mv[1][1] = 0; cu.direct = 0;
            for (int d = 0; d < 2; d++) {
                const int v = cu.mv[0][d], a = sym_abs_mvd(c, v < 0 ? -v : v, models.mvd[0]);
                int sg = v < 0;
                if (a) sg = c.ep(sg);
                cu.mv[0][d] = (int16_t)(sg ? -a : a);
            }
        } else if (!intra && sps.tool_admvp) {
            // xevdm_eco.c:1595-1726 with the sub-tools off: merge_mode_flag (the CU takes a merge candidate: pred_mode MODE_DIR), else
            // inter_pred_idc (no bi-prediction for CUs of 4x4 / 4x8 / 8x4), bi_idx of a bi-predicted CU (normal / list 0 / list 1 without a
            // coded difference; the latter two also derive their reference indices), per list reference index and vector difference.
            // The predictor is the resolution-indexed one (index 0 without AMVR): mv = predictor + mvd (xevd_get_inter_motion, xevdm.c:885-932)
            int mvr = 0;                                             // xevdm_eco_mvr_idx (xevdm_eco.c:814-817): quarter, half, 1, 2, 4 samples
            if (sps.tool_amvr) {
                if (enc) mvr = (cu.direct || cu.affine) ? 0 : ((cu.x >> 3) * 5 + (cu.y >> 3) * 3) % 11 % 5 * (((cu.x ^ cu.y) >> 2) & 1);      // about every second coded vector on a coarser grid
                mvr = sym_trunc_unary(c, mvr, models.mvr_idx, 4, 5);
            }
            cu.direct = mvr == 0 ? c.bin(cu.direct, models.merge_mode[0]) : 0;
            if (cu.direct) {
                if (sps.tool_mmvd) cu.mmvd = c.bin(cu.mmvd, models.mmvd_flag[0]);
                if (cu.mmvd) { cu.affine = 0; code_mmvd_idx(c, cu); mmvd_motion(cu); }
                else if (code_affine_merge(c, cu)) { }
                else {
                    cu.mvp_idx[0] = cu.mvp_idx[1] = sym_trunc_unary(c, cu.mvp_idx[0], models.merge_idx, 5, 6);
                    merge_motion(cu, cu.mvp_idx[0]);
                    cu.dmvr = sps.tool_dmvr;
                }
            } else {
                int dir = 0;
                if (n_lists == 2) {
                    if (enc) dir = (cu.refi[0] >= 0 && cu.refi[1] >= 0 && bi_applicable(cu)) ? 2 : ((cu.refi[1] >= 0 && cu.refi[0] < 0) ? 1 : 0);
                    int not_bi = 1;
                    if (bi_applicable(cu)) not_bi = c.bin(dir != 2, models.inter_dir[0]);
                    if (!not_bi) dir = 2;
                    else dir = c.bin(dir == 1, models.inter_dir[1]) ? 1 : 0;
                }
                // affine inter CU (xevdm_eco.c:1649-1682): 16x16 and larger, quarter-sample vectors only; affine_mode picks 2 or 3 control points, then
                // per list the reference, one of two predictors, and the control-point differences (all zero with affine_mvd_flag)
                int aff = 0;
                if (sps.tool_affine && cu.log2w >= 4 && cu.log2h >= 4 && mvr == 0) aff = c.bin(cu.affine != 0, models.affine_flag[nb_ctx(cu, CTX_AFF)]);
                if (!aff) cu.affine = 0;
                else {
                    cu.affine = 1 + c.bin(cu.affine == 2, models.affine_mode[0]);
                    const int vn = cu.affine + 1;
                    for (int l = 0; l < 2; l++) {
                        cu.mv[l][0] = cu.mv[l][1] = 0;
                        if (!(((dir + 1) >> l) & 1)) { cu.refi[l] = -1; memset(cu.aff_mv[l], 0, sizeof(cu.aff_mv[l])); continue; }
                        cu.refi[l] = code_refi(c, cu.refi[l], (int)refp[l].size());
                        cu.aff_idx[l] = c.bin(cu.aff_idx[l] != 0, models.affine_mvp_idx[0]);
                        int16_t mvp[2][3][2];
                        aff_amvp(cu, l, cu.refi[l], vn, mvp);
                        int16_t (*pp)[2] = mvp[cu.aff_idx[l]];
                        if (enc) {      // differences that reproduce the requested control points: the first one also moves the other predictors
                            for (int d = 0; d < 2; d++) {
                                cu.aff_mvd[l][0][d] = (int16_t)(cu.aff_mv[l][0][d] - pp[0][d]);
                                for (int v = 1; v < vn; v++) cu.aff_mvd[l][v][d] = (int16_t)(cu.aff_mv[l][v][d] - (int16_t)(pp[v][d] + cu.aff_mvd[l][0][d]));
                            }
                        }
                        int zero = 1;
                        for (int v = 0; v < vn; v++) zero &= cu.aff_mvd[l][v][0] == 0 && cu.aff_mvd[l][v][1] == 0;
                        zero = c.bin(zero, models.affine_mvd_flag[l]);
                        for (int v = 0; v < vn; v++) {
                            if (zero) cu.aff_mvd[l][v][0] = cu.aff_mvd[l][v][1] = 0; else code_mvd(c, cu.aff_mvd[l][v]);
                            for (int d = 0; d < 2; d++) cu.aff_mv[l][v][d] = (int16_t)(pp[v][d] + cu.aff_mvd[l][v][d]);
                            if (v == 0) for (int d = 0; d < 2; d++) { pp[1][d] = (int16_t)(pp[1][d] + cu.aff_mvd[l][0][d]); pp[2][d] = (int16_t)(pp[2][d] + cu.aff_mvd[l][0][d]); }
                        }
                    }
                }
                int bi_idx = 0;                                      // BI_NON 0, BI_NORMAL 1, BI_FL0 2, BI_FL1 3
                if (dir == 2 && !aff) {
                    if (enc) bi_idx = 1 + ((cu.x >> 2) + (cu.y >> 2) * 3) % 7 % 3;      // a spread of the three kinds over the picture
                    const int v = bi_idx - 1;
                    if (c.bin(v == 0, models.bi_idx[0])) bi_idx = 1;
                    else bi_idx = c.bin(v == 1, models.bi_idx[1]) ? 2 : 3;
                }
                for (int l = 0; l < 2 && !aff; l++) {
                    if (!(((dir + 1) >> l) & 1)) { cu.refi[l] = -1; cu.mv[l][0] = cu.mv[l][1] = 0; continue; }
                    const int nref = (int)refp[l].size();
                    if (bi_idx != 2 && bi_idx != 3) {
                        if (nref > 1) {                              // xevd_eco_refi, xevd_eco.c:409-436
                            const int r = std::min(std::max(cu.refi[l], 0), nref - 1);
                            int v = 0;
                            if (c.bin(r > 0, models.refi[0])) {
                                v = 1;
                                if (nref > 2 && c.bin(r > 1, models.refi[1])) {
                                    v = 2;
                                    for (; v < nref - 1; v++) if (!c.ep(r > v)) break;
                                }
                            }
                            cu.refi[l] = v;
                        } else cu.refi[l] = 0;
                    } else cu.refi[l] = first_refi(cu, l, mvr);
                    int16_t mvp[2];
                    mvr_predictor(cu, l, cu.refi[l], mvr, mvp);
                    if (bi_idx == 2 + l) cu.mvd[l][0] = cu.mvd[l][1] = 0;
                    else {
                        if (enc) { cu.mvd[l][0] = (int16_t)((cu.mv[l][0] - mvp[0]) >> mvr); cu.mvd[l][1] = (int16_t)((cu.mv[l][1] - mvp[1]) >> mvr); }
                        for (int d = 0; d < 2; d++) {                // xevd_eco_get_mvd, xevd_eco.c:491-536
                            const int v = cu.mvd[l][d], a = sym_abs_mvd(c, v < 0 ? -v : v, models.mvd[0]);
                            int sg = v < 0;
                            if (a) sg = c.ep(sg);
                            cu.mvd[l][d] = (int16_t)(sg ? -a : a);
                        }
                    }
                    cu.mv[l][0] = (int16_t)(mvp[0] + cu.mvd[l][0] * (1 << mvr)); cu.mv[l][1] = (int16_t)(mvp[1] + cu.mvd[l][1] * (1 << mvr));
                }
            }
        } else if (!intra) {
            // xevd_eco.c:1120-1148: B: direct_mode_flag, else inter_pred_idc; per list in use: ref index, predictor index, mvd; mv = mvp + mvd (xevd.c:533-556)
            int dir = 0;                                             // PRED_L0 0, PRED_L1 1, PRED_BI 2
            if (n_lists == 2) cu.direct = c.bin(cu.direct, models.direct[0]);
            if (cu.direct) direct_motion(cu);
            else {
                if (n_lists == 2) {
                    if (enc) dir = (cu.refi[0] >= 0 && cu.refi[1] >= 0) ? 2 : (cu.refi[1] >= 0 ? 1 : 0);
                    if (!c.bin(dir != 2, models.inter_dir[0])) dir = 2;
                    else dir = c.bin(dir == 1, models.inter_dir[1]) ? 1 : 0;
                }
                for (int l = 0; l < 2; l++) {
                    if (!(((dir + 1) >> l) & 1)) { cu.refi[l] = -1; cu.mv[l][0] = cu.mv[l][1] = 0; continue; }
                    const int nref = (int)refp[l].size();
                    if (nref > 1) {                                  // xevd_eco_refi, xevd_eco.c:409-436
                        const int r = cu.refi[l];
                        int v = 0;
                        if (c.bin(r > 0, models.refi[0])) {
                            v = 1;
                            if (nref > 2 && c.bin(r > 1, models.refi[1])) {
                                v = 2;
                                for (; v < nref - 1; v++) if (!c.ep(r > v)) break;
                            }
                        }
                        cu.refi[l] = v;
                    } else cu.refi[l] = 0;
                    mvp_candidates(cu, l, cand);
                    if (enc) {                                       // cheapest predictor
                        int best = 0, cost = 1 << 30;
                        for (int k = 0; k < 4; k++) { const int d = abs(cu.mv[l][0] - cand[k][0]) + abs(cu.mv[l][1] - cand[k][1]); if (d < cost) { cost = d; best = k; } }
                        cu.mvp_idx[l] = best;
                        cu.mvd[l][0] = (int16_t)(cu.mv[l][0] - cand[best][0]); cu.mvd[l][1] = (int16_t)(cu.mv[l][1] - cand[best][1]);
                    }
                    cu.mvp_idx[l] = sym_trunc_unary(c, cu.mvp_idx[l], models.mvp_idx, 3, 4);
                    for (int d = 0; d < 2; d++) {                    // xevd_eco_get_mvd, xevd_eco.c:491-536
                        const int v = cu.mvd[l][d], a = sym_abs_mvd(c, v < 0 ? -v : v, models.mvd[0]);
                        int sg = v < 0;
                        if (a) sg = c.ep(sg);
                        cu.mvd[l][d] = (int16_t)(sg ? -a : a);
                    }
                    cu.mv[l][0] = (int16_t)(cand[cu.mvp_idx[l]][0] + cu.mvd[l][0]); cu.mv[l][1] = (int16_t)(cand[cu.mvp_idx[l]][1] + cu.mvd[l][1]);
                }
            }
        } else if (cu.tree == 2 && !sps.tool_eipd) {
            cu.ipm_c = cu.ipm = luma_mode_of(cu);                  // no syntax: the chroma block takes the luma mode at its centre (xevdm_eco.c:1763-1781)
        } else if (sps.tool_eipd) {
            // xevd_eco_intra_dir (xevd_eco.c:842-879): one of the 2 most probable modes, one of the 8 extended ones (bypass), or the index
            // among the remaining 23 in truncated binary (4 or 5 bypass bins)
            int mpm[2], ext[8], pims[33];
            if (cu.tree == 2) cu.ipm = luma_mode_of(cu);           // chroma-only CU: DM refers to the luma mode at the block's centre, DC when that CU is not intra (xevdm_eco.c:1738-1752)
            else {
            eipd_mpm(cu, mpm, ext, pims);
            int pos = 0;
            while (enc && pos < 33 && pims[pos] != cu.ipm) pos++;
            const int in_mpm = enc ? (cu.ipm == mpm[0] || cu.ipm == mpm[1]) : 0;
            if (c.bin(in_mpm, models.ipm_mpm_flag[0])) {
                cu.ipm = mpm[c.bin(cu.ipm == mpm[1], models.ipm_mpm_idx[0])];
            } else {
                int ei = 0;
                while (enc && ei < 8 && ext[ei] != cu.ipm) ei++;
                if (c.ep(enc ? ei < 8 : 0)) {
                    cu.ipm = ext[sym_bits_ep(c, ei & 7, 3)];
                } else {
                    const int rem = pos - 10;                        // 23 symbols: 9 with 4 bins, 14 with 5
                    int v = sym_bits_ep(c, rem < 9 ? rem : (rem + 9) >> 1, 4);
                    if (v >= 9) v = ((v << 1) | c.ep((rem + 9) & 1)) - 9;
                    cu.ipm = pims[10 + std::min(v, 22)];
                }
            }
            }
            if (cu.tree == 1) cu.ipm_c = 0;
            else {
            // xevd_eco_intra_dir_c (:881-910): DM, or one of the other modes - the one DM stands for (a luma DC / BI / HOR / VER) is skipped
            const int lc = cu.ipm == 12 ? 4 : cu.ipm == 24 ? 3 : cu.ipm == 0 ? 2 : cu.ipm == 2 ? 1 : 0;
            if (enc && lc && cu.ipm_c == lc) cu.ipm_c = 0;
            if (c.bin(cu.ipm_c == 0, models.ipm_chroma[0])) cu.ipm_c = 0;
            else {
                int v = sym_unary_ep(c, (lc && cu.ipm_c > lc ? cu.ipm_c - 1 : cu.ipm_c) - 1, 4) + 1;
                if (lc && v >= lc) v++;
                cu.ipm_c = std::min(v, 4);
            }
            }
        } else {
            const uint8_t *mpm = mpm_list(cu);                       // xevd_eco_intra_dir_b, xevd_eco.c:826-846: the code number is sent
            const int code = sym_unary(c, mpm[cu.ipm], models.intra_dir, 2);
            if (!enc) for (int i = 0; i < 5; i++) if (mpm[i] == code) cu.ipm = i;
        }
        // coded block flags (eco_cbf, xevd_eco.c:260-341), CU <= 64: no sub-blocks
        bool all_zero = false;
        if (!intra && cu.tree == 0) {
            // a merge-mode CU (MODE_DIR with tool_admvp) has coefficients by definition - without them it would be a skip CU: no cbf_all (xevdm_eco.c:831)
            const bool merge_cu = sps.tool_admvp && cu.direct && cu.mode == MODE_INTER;
            const int any = merge_cu ? 1 : c.bin((cu.cbf[0] | cu.cbf[1] | cu.cbf[2]) != 0, models.cbf_all[0]);
            if (!any) { cu.cbf[0] = cu.cbf[1] = cu.cbf[2] = 0; all_zero = true; }
            else {
                cu.cbf[1] = c.bin(cu.cbf[1], models.cbf_cb[0]);
                cu.cbf[2] = c.bin(cu.cbf[2], models.cbf_cr[0]);
                if (cu.cbf[1] + cu.cbf[2] == 0) cu.cbf[0] = 1;
                else cu.cbf[0] = c.bin(cu.cbf[0], models.cbf_luma[0]);
            }
        } else {                                                     // intra CUs, and every CU of a dual tree (an IBC luma CU too): one flag per component it has (xevdm_eco_cbf, xevdm_eco.c:266-296)
            cu.cbf[1] = cu.tree == 1 ? 0 : c.bin(cu.cbf[1], models.cbf_cb[0]);
            cu.cbf[2] = cu.tree == 1 ? 0 : c.bin(cu.cbf[2], models.cbf_cr[0]);
            cu.cbf[0] = cu.tree == 2 ? 0 : c.bin(cu.cbf[0], models.cbf_luma[0]);
        }
        // QP (xevd_eco.c:640-668, xevd_eco_dqp :460-479): a delta only when the CU has coefficients
        // sps->dquant_flag (Main; xevdm_eco.c:882-897): one delta per quantisation group - a CU of at least the group size sends it when it has
        // coefficients (code 1), the first CU that gets this far inside a group of smaller CUs sends it in any case (code 2), the others take the predictor
        const bool any_cbf = cu.cbf[0] || cu.cbf[1] || cu.cbf[2];
        const bool qp_here = sps.profile_main && sps.dquant ? ((cu.qp_code == 1 && !qp_coded && any_cbf) || (cu.qp_code == 2 && !qp_coded)) : any_cbf;
        if (!all_zero && pps.cu_qp_delta && qp_here) {
            int dqp = 0;
            if (enc) { dqp = cu.qp - qp_prev; while (dqp > 25) dqp -= 52; while (dqp < -26) dqp += 52; }
            const int a = sym_unary(c, dqp < 0 ? -dqp : dqp, models.dqp, 1);
            int s = dqp < 0;
            if (a) s = c.ep(s);
            dqp = s ? -a : a;
            cu.qp = (qp_prev + dqp + 52) % 52;
            qp_prev = cu.qp;
            qp_coded = 1;
        } else cu.qp = qp_prev;
        if (all_zero) return;
        // Main, tool_ats (xevdm_eco_coef, xevdm_eco.c:902-934): transform selection of intra luma blocks up to 32x32, sub-block
        // transform of inter CUs (the coefficient blocks of all components then have the TU size, xevdm_eco_xcoef :697-703)
        int tlw = cu.log2w, tlh = cu.log2h;
        if (sps.tool_ats) {
            if (intra && cu.cbf[0] && cu.log2w <= 5 && cu.log2h <= 5) {
                int on = c.ep(cu.ats & 1), mh = 0, mv = 0;
                if (on) { mh = c.bin((cu.ats >> 2) & 1, models.ats_mode[0]); mv = c.bin((cu.ats >> 1) & 1, models.ats_mode[0]); }
                cu.ats = on | (mv << 1) | (mh << 2);
            } else cu.ats = 0;
            const int w = 1 << cu.log2w, h = 1 << cu.log2h;
            const int avail = (intra || ibc || w > 64 || h > 64) ? 0 : ((w >= 8) | ((h >= 8) << 1) | ((w >= 16) << 2) | ((h >= 16) << 3));      // xevdm_util.c:3565-3583
            cu.ats_inter = avail ? code_ats_inter(c, cu.ats_inter, avail, cu.log2w, cu.log2h) : 0;
            const int idx = cu.ats_inter & 15;
            if (idx == 1 || idx == 3) tlw -= idx == 3 ? 2 : 1;
            if (idx == 2 || idx == 4) tlh -= idx == 4 ? 2 : 1;
        }
        for (int k = 0; k < 3; k++)
            if (cu.cbf[k]) { if (sps.tool_adcc) code_adcc(c, coef[k], tlw - (k ? 1 : 0), tlh - (k ? 1 : 0), k != 0, enc); else code_coefs(c, coef[k], tlw - (k ? 1 : 0), tlh - (k ? 1 : 0), k != 0, enc); }
    }

    // ats_inter_info syntax (xevdm_eco_ats_inter_info, xevdm_eco.c:128-190; with cm_init the flag's context goes by the CU's area, the direction's by its shape)
    template <class C> int code_ats_inter(C &c, int info, int avail, int log2w, int log2h)
    {
        const int mv_ = avail & 1, mh_ = (avail >> 1) & 1, vq = (avail >> 2) & 1, hq = (avail >> 3) & 1;
        const int ctx_flag = sps.tool_cm_init ? (log2w + log2h >= 8 ? 0 : 1) : 0, ctx_hor = sps.tool_cm_init ? (log2w == log2h ? 0 : log2w < log2h ? 1 : 2) : 0;
        if (!c.bin(info != 0, models.ats_inter_flag[ctx_flag])) return 0;
        const int idx = info & 15;
        int quad = idx >= 3, hor = idx == 2 || idx == 4, pos = (info >> 4) & 1;
        if ((vq || hq) && (mv_ || mh_)) quad = c.bin(quad, models.ats_inter_quad[0]); else quad = 0;
        if ((quad && vq && hq) || (!quad && mv_ && mh_)) hor = c.bin(hor, models.ats_inter_hor[ctx_hor]);
        else hor = (quad && hq) || (!quad && mh_);
        pos = c.bin(pos, models.ats_inter_pos[0]);
        return ((quad ? 2 : 0) + (hor ? 1 : 0) + 1) | (pos << 4);
    }
};

// ------------------------------------------------------------------------------------------------ NAL plumbing
enum { NUT_NONIDR = 0, NUT_IDR = 1, NUT_SPS = 24, NUT_PPS = 25, NUT_SEI = 28 };

static void write_nal(std::vector<uint8_t> &out, int nut, int tid, const BitWriter &payload)
{
    const uint32_t len = (uint32_t)payload.buf.size() + 2;
    for (int i = 3; i >= 0; i--) out.push_back((uint8_t)(len >> (8 * i)));
    const uint32_t hdr = ((uint32_t)(nut + 1) << 9) | ((uint32_t)tid << 6);      // 1 zero bit, type + 1 (6), tid (3), 5 reserved zero bits, 1 extension bit
    out.push_back((uint8_t)(hdr >> 8)); out.push_back((uint8_t)hdr);
    out.insert(out.end(), payload.buf.begin(), payload.buf.end());
}

}   // namespace

// =============================================================================================================== parser
// sps->dquant_flag: where a quantisation group starts in the split tree (xevd_entropy_decode_tree, src_main/xevdm.c:1739-1759) - at a leaf of at least
// pps.cu_qp_delta_area samples (code 1: the delta goes with the CU's coefficients), or at the split node of exactly that size (code 2: the first CU below it
// that reaches its QP syntax sends the delta).  -> the code for the node's children / the leaf
static int qp_group(const Stream &st, TileCoder &tc, int split, int lw, int lh, int qp_code)
{
    if (!(st.pps.cu_qp_delta && st.sps.dquant && st.sps.profile_main)) return qp_code;
    if (!split && lw + lh >= st.pps.qp_delta_area && qp_code != 2) { tc.qp_coded = 0; return (lw == 7 || lh == 7) ? 2 : 1; }
    const bool tri = split == TileCoder::TRI_VER || split == TileCoder::TRI_HOR;
    if ((tri && lw + lh == st.pps.qp_delta_area + 1) || (lw + lh == st.pps.qp_delta_area && qp_code != 2)) { tc.qp_coded = 0; return 2; }
    return qp_code;
}

// One tile of a picture being parsed: its coder state, its share of the batch, its scratch blocks.  Objects are kept between pictures (the vectors keep
// their capacity); with several tiles and xhost_parser_set_threads() > 1 they run on different threads.
struct TileParser {
    Stream &st;
    TileCoder tc;
    Batch batch;
    std::vector<int16_t> blk[3];
    std::string err;
    size_t n_coef = 0;
    explicit TileParser(Stream &s) : st(s), tc(s) { for (int k = 0; k < 3; k++) blk[k].assign(64 * 64, 0); }
    int fail(const char *m) { err = m; return XHOST_ERR_MALFORMED; }
    // xevd_tile_eco (src_main/xevdm.c:2363-2461) for tile (tc_, tr) of the grid, from bit position `pos` of the slice NAL
    int parse_tile(const BitReader &br0, size_t pos, int tcol, int trow)
    {
        BitReader br = br0;
        br.pos = pos;
        if (br.pos > br.size * 8) return fail("tile entry point past the end of the slice");
        const Slice &sh = st.sh;
        const int w_ctu = (st.sps.width + 63) >> 6;
        batch.clear();
        n_coef = 0;
        tc.missing_ref = false;
        if (st.sps.tool_cm_init) tc.models.reset_cm(sh.type == XHOST_SLICE_B, sh.qp); else tc.models.reset();
        tc.qp_prev = sh.qp;
        Dec dec;
        dec.br = &br;
        dec.start();
        for (int cy = st.grid.row_bd[trow]; cy < st.grid.row_bd[trow + 1]; cy++) for (int cx = st.grid.col_bd[tcol]; cx < st.grid.col_bd[tcol + 1]; cx++) {
            if (cx == st.grid.col_bd[tcol]) tc.history_reset();
            batch.ctu_start.push_back((uint32_t)batch.x.size());
            if (sh.alf_on && sh.alf_ctb_map) st.alf_ctb_flag[(size_t)cy * w_ctu + cx] = (uint8_t)dec.bin(0, tc.models.alf_ctb[0]);      // xevdm.c:2411-2418
            const int rc = st.sps.btt ? parse_node(dec, cx << 6, cy << 6, 6, 6, 0, false) : parse_tree(dec, cx << 6, cy << 6, 6);
            if (rc != XGPU_OK) return rc;
            if (br.overrun) return fail("slice data ends early");
            if (tc.missing_ref) return fail("tool_dmvr with tool_hmvp / tool_mmvd: the samples of a reference picture are needed (xhost_parser_set_ref_luma)");
        }
        if (dec.tile_end() != 1) return fail("missing end-of-tile flag");
        return XGPU_OK;
    }
    int parse_tree(Dec &dec, int x, int y, int log2s, int qp_code = 0)
    {
        const int s = 1 << log2s;
        int split = 0;
        if (s > 4 && !(s < 8)) split = dec.bin(0, tc.models.split[0]);
        qp_code = qp_group(st, tc, split ? TileCoder::QUAD : 0, log2s, log2s, qp_code);
        if (split) {
            const int h = s >> 1;
            for (int i = 0; i < 4; i++) {
                const int nx = x + (i & 1) * h, ny = y + (i >> 1) * h;
                if (nx < st.sps.width && ny < st.sps.height) { const int rc = parse_tree(dec, nx, ny, log2s - 1, qp_code); if (rc != XGPU_OK) return rc; }
            }
            return XGPU_OK;
        }
        return leaf(dec, x, y, log2s, log2s, qp_code, 0);
    }
    // sps_btt_flag: a node of the binary / ternary split tree (xevd_entropy_decode_tree, src_main/xevdm.c:1644-1850, without SUCO)
    int parse_node(Dec &dec, int x, int y, int lw, int lh, int qp_code, bool only_inter, bool only_intra = false)
    {
        const int W = st.sps.width, H = st.sps.height, w = 1 << lw, h = 1 << lh, mn = 1 << st.sps.log2_min_cb;
        int split = TileCoder::NO_SPLIT;
        if (w > mn || h > mn) {
            if (x + w <= W && y + h <= H) split = tc.code_split(dec, 0, x, y, lw, lh, only_inter);
            else {      // across the picture border: the binary split towards it, no syntax (:1687-1713)
                int allow[6];
                tc.split_allowed(allow, lw, lh, x, y, only_inter);
                split = allow[TileCoder::BI_VER] ? TileCoder::BI_VER : allow[TileCoder::BI_HOR] ? TileCoder::BI_HOR : -1;
                if (split < 0) return fail("a node across the picture border cannot be split");
            }
        }
        qp_code = qp_group(st, tc, split, lw, lh, qp_code);
        if (split == TileCoder::NO_SPLIT) { last_qp_code = qp_code; return leaf(dec, x, y, lw, lh, qp_code, only_inter, only_intra ? 1 : 0); }
        const int mc = only_intra ? 0 : tc.code_mode_cons(dec, split, lw, lh, only_inter, 0);
        int px[3], py[3], plw[3], plh[3];
        const int n = TileCoder::split_parts(split, x, y, lw, lh, px, py, plw, plh);
        for (int i = 0; i < n; i++)
            if (px[i] < W && py[i] < H) { const int rc = parse_node(dec, px[i], py[i], plw[i], plh[i], qp_code, mc == 1, only_intra || mc < 0); if (rc != XGPU_OK) return rc; }
        // local dual tree: the luma CUs above, now the node's chroma block as one CU (xevdm.c:1828-1833; core->cu_qp_delta_code keeps the last leaf's value)
        if (mc < 0) return leaf(dec, x, y, lw, lh, last_qp_code, 0, 2);
        return XGPU_OK;
    }
    int last_qp_code = 0;
    int leaf(Dec &dec, int x, int y, int lw, int lh, int qp_code, int only_inter, int tree = 0)
    {
        if (x + (1 << lw) > st.sps.width || y + (1 << lh) > st.sps.height) return fail("a CU crosses the picture border");
        Cu cu;
        memset(&cu, 0, sizeof(cu));
        cu.x = x; cu.y = y; cu.log2w = lw; cu.log2h = lh; cu.qp_code = qp_code; cu.only_inter = only_inter; cu.tree = tree;
        int16_t *coef[3] = { blk[0].data(), blk[1].data(), blk[2].data() };
        memset(coef[0], 0, sizeof(int16_t) << (lw + lh));
        memset(coef[1], 0, sizeof(int16_t) << (lw + lh - 2));
        memset(coef[2], 0, sizeof(int16_t) << (lw + lh - 2));
        tc.code_cu(dec, cu, coef, false);
        tc.commit(cu);
        // append to the batch
        if (batch.x.empty()) n_coef = 0;
        batch.x.push_back((uint16_t)x); batch.y.push_back((uint16_t)y); batch.log2w.push_back((uint8_t)lw); batch.log2h.push_back((uint8_t)lh);
        batch.pred_mode.push_back((uint8_t)cu.mode);
        batch.refi.push_back((int8_t)cu.refi[0]); batch.refi.push_back((int8_t)cu.refi[1]);
        for (int l = 0; l < 2; l++) { batch.mv.push_back(cu.mv[l][0]); batch.mv.push_back(cu.mv[l][1]); }
        int qp_u, qp_v;
        tc.chroma_qps(cu.qp, qp_u, qp_v);
        batch.qp.push_back((uint8_t)(cu.qp + 6 * (st.sps.bd_l - 8))); batch.qp.push_back((uint8_t)qp_u); batch.qp.push_back((uint8_t)qp_v);
        batch.cbf.push_back((uint8_t)(cu.cbf[0] | (cu.cbf[1] << 1) | (cu.cbf[2] << 2)));
        batch.ipm.push_back((uint8_t)cu.ipm); batch.ipm.push_back((uint8_t)(st.sps.tool_eipd ? cu.ipm_c : cu.ipm));      // Baseline: chroma mode = luma mode, xevd_eco.c:1154
        batch.ats.push_back((uint8_t)cu.ats); batch.ats_inter.push_back((uint8_t)cu.ats_inter); batch.dmvr.push_back((uint8_t)cu.dmvr);
        batch.tree.push_back((uint8_t)tree); batch.has_tree |= tree != 0;
        if (st.sps.tool_affine) {      // xgpu_cu_batch.affine: 0 / 2 / 3 control points, .affine_mv[list][vertex][x/y]
            batch.affine.push_back((uint8_t)(cu.affine ? cu.affine + 1 : 0));
            batch.affine_mv.insert(batch.affine_mv.end(), &cu.aff_mv[0][0][0], &cu.aff_mv[0][0][0] + 12);
        }
        batch.coef_off.push_back((uint32_t)n_coef);
        const int tu_shift = (cu.ats_inter & 15) == 0 ? 0 : (((cu.ats_inter & 15) >= 3) ? 2 : 1);      // the TU is 1/2 or 1/4 of the CU
        for (int k = 0; k < 3; k++)
            if (cu.cbf[k]) {
                const size_t n = (size_t)1 << (lw + lh - (k ? 2 : 0) - tu_shift);
                batch.coef.insert(batch.coef.end(), coef[k], coef[k] + n);
                n_coef += n;
            }
        return XGPU_OK;
    }
};

struct xhost_parser {
    std::vector<uint8_t> data;
    size_t pos = 0;
    Stream st;
    std::vector<std::unique_ptr<TileParser>> tiles;      // one per tile of the current picture (kept between pictures)
    Batch merged;                                        // several tiles: their batches, tile by tile
    // What a handed-out xhost_picture points at lives in a ring of `held.size()` slots (xhost_parser_set_depth, default 1): the arrays of picture k
    // stay untouched until the call that hands out picture k + depth - so a caller can build and launch picture k on one thread while another
    // one is inside xhost_parser_next for picture k + 1 (depth 2).  The batch arrays are swapped into the slot (no copy), the small tables copied.
    struct Held {
        Batch batch;
        int16_t alf_luma[25][13], alf_chroma[7];
        std::vector<uint8_t> ctb;
        std::vector<int32_t> dra;
        xgpu_tile_grid grid;
        int16_t *arena = nullptr;                        // xhost_parser_set_arena: the coefficients of a picture with several tiles are gathered here
        size_t arena_cap = 0;                            //   (in samples) instead of in batch.coef
    };
    void *(*arena_alloc)(void *, size_t) = nullptr;
    void (*arena_release)(void *, void *) = nullptr;
    void *arena_user = nullptr;
    ~xhost_parser() { for (Held &h : held) if (h.arena && arena_release) arena_release(arena_user, h.arena); }
    std::vector<Held> held = std::vector<Held>(1);
    size_t n_handed = 0;
    int16_t *merged_arena = nullptr;                     // the arena the coefficients of the picture being handed out were gathered in (else batch.coef)
    Batch *cur = &merged;                                // the batch of the picture handed out last (in its ring slot)
    size_t n_coef = 0;
    int n_threads = 1;                                   // xhost_parser_set_threads
    std::string err;
    int fail(const char *m) { err = m; pic_tiles_left = 0; return XHOST_ERR_MALFORMED; }      // (a picture half assembled from slices is dropped)
    // runs fn(0 .. n-1) on up to n_threads threads (the calling one included)
    template <class F> void parallel_for(int n, F fn)
    {
        const int nt = std::min(n_threads, n);
        if (nt <= 1) { for (int i = 0; i < n; i++) fn(i); return; }
        std::atomic<int> next(0);
        auto work = [&]() { for (int i = next.fetch_add(1); i < n; i = next.fetch_add(1)) fn(i); };
        std::vector<std::thread> th;
        for (int t = 1; t < nt; t++) th.emplace_back(work);
        work();
        for (std::thread &t : th) t.join();
    }

    // A picture-signature SEI directly after the slice NAL belongs to that picture (xevd_dec_nalu checks it against ctx->pic,
    // src_base/xevd.c:2010-2026).  Payload (xevd_eco_sei, xevd_eco.c:1617-1678): type 0x10, size 16, then 16 bytes PER PLANE.
    void attach_signature(xhost_picture *out)
    {
        out->has_md5 = 0;
        if (pos + 6 > data.size()) return;
        const uint8_t *d = data.data() + pos;
        const size_t len = ((size_t)d[0] << 24) | ((size_t)d[1] << 16) | ((size_t)d[2] << 8) | d[3];
        if (len < 2 + 2 + 48 || pos + 4 + len > data.size()) return;
        if ((((d[4] << 8 | d[5]) >> 9) & 63) - 1 != NUT_SEI) return;
        if (d[6] != 0x10 || d[7] != 16) return;
        memcpy(out->md5, d + 8, 48);
        out->has_md5 = 1;
        pos += 4 + len;
    }

    int parse_sps(BitReader &br)
    {
        Sps tmp = st.sps;                                // parsed into a copy and committed on success only: a damaged SPS leaves the active one intact
        Sps &s = tmp;
        br.ue();                                         // sps_seq_parameter_set_id
        const int profile = (int)br.get(8);              // 0 Baseline, 1 Main, 2/3 still picture
        if (profile < 0 || profile > 3) return fail("unknown profile");
        s.profile_main = profile == 1 || profile == 3;
        br.get(8); br.get(32); br.get(32);               // level, toolset_idc_h/l
        if (br.ue() != 1) return fail("only 4:2:0 is supported");
        s.width = (int)br.ue(); s.height = (int)br.ue();
        s.bd_l = (int)br.ue() + 8; s.bd_c = (int)br.ue() + 8;
        // range checks before anything is sized or indexed by these fields (the chroma QP table below is)
        if (br.overrun || (s.width & 7) || (s.height & 7) || s.width <= 0 || s.height <= 0 || s.width > 16384 || s.height > 16384 ||
            s.bd_l < 8 || s.bd_l > 12 || s.bd_c < 8 || s.bd_c > 12) return fail("bad SPS");
        int unsupported = 0, rpl = 0, pocs = 0;
        s.tool_iqt = s.tool_ats = s.tool_addb = s.tool_alf = s.tool_eipd = s.tool_dra = s.tool_htdf = 0;
        if (!s.profile_main) {
            for (int i = 0; i < 13; i++) { const int f = br.get1(); if (i != 11) unsupported |= f; }      // btt suco admvp eipd cm_init iqt addb alf htdf rpl pocs dquant dra
        } else {                                          // xevdm_eco_sps, xevdm_eco.c:1863-1937: sub-flags follow their tool flag
            s.btt = br.get1();                           // sps_btt_flag + the limits of the split tree (xevdm_eco.c:1863-1871; table xevdm_util.c:4393-4400)
            if (s.btt) {
                if (br.ue() != 1) return fail("sps_btt_flag with a CTU size other than 64 is not supported");      // log2_ctu_size_minus5
                for (int i = 0; i < 4; i++) s.btt_raw[i] = (int)br.ue();
                if (br.overrun || s.btt_raw[0] > 4 || s.btt_raw[1] > 6 || s.btt_raw[2] > 6 || s.btt_raw[3] > 6) return fail("bad SPS: split limits");
                s.log2_min_cb = s.btt_raw[0] + 2;
                s.split_tbl[0][1] = 6; s.split_tbl[0][0] = s.log2_min_cb;
                s.split_tbl[1][1] = 6; s.split_tbl[1][0] = s.log2_min_cb + 1;
                s.split_tbl[2][1] = std::min(6 - s.btt_raw[1], 6); s.split_tbl[2][0] = s.log2_min_cb + 2;
                s.split_tbl[3][1] = std::min(6 - s.btt_raw[2], 6); s.split_tbl[3][0] = s.log2_min_cb + s.btt_raw[3] + 2;
            }
            unsupported |= br.get1();                    // sps_suco_flag
            s.tool_admvp = br.get1();
            s.tool_amvr = s.tool_hmvp = s.tool_dmvr = s.tool_mmvd = s.tool_affine = 0;
            if (s.tool_admvp) { s.tool_affine = br.get1(); s.tool_amvr = br.get1(); s.tool_dmvr = br.get1(); s.tool_mmvd = br.get1(); s.tool_hmvp = br.get1(); }      // tool_affine, tool_amvr, tool_dmvr, tool_mmvd, tool_hmvp
            s.tool_eipd = br.get1();
            s.ibc = s.ibc_log_max = 0;
            if (s.tool_eipd && (s.ibc = br.get1())) { s.ibc_log_max = (int)br.ue() + 2; if (s.ibc_log_max > 7) return fail("bad SPS"); }
            s.tool_cm_init = br.get1(); s.tool_adcc = s.tool_cm_init ? br.get1() : 0;      // tool_cm_init, tool_adcc (xevdm_eco.c:1900-1904)
            s.tool_iqt = br.get1();
            if (s.tool_iqt) s.tool_ats = br.get1();
            s.tool_addb = br.get1();
            s.tool_alf = br.get1();
            s.tool_htdf = br.get1();                     // no syntax of its own: the backend filters with the slice QP (xevdm.c:1381-1392)
            rpl = br.get1(); pocs = br.get1();
            s.dquant = br.get1();                        // dquant_flag: QP deltas per quantisation group of pps.cu_qp_delta_area (xevdm.c:1739-1759, xevdm_eco.c:882-897)
            s.tool_dra = br.get1();
        }
        // tool_dmvr with tool_hmvp / tool_mmvd: xevdm_set_dec_info ends by copying map_mv[first SCU] - the REFINED vector of the CU's first sub-block - back into
        // core->mv (xevdm_util.c:4384-4387), which is what the history buffer then receives (xevdm.c:1335-1342), and an MMVD CU builds its merge list from
        // ctx->map_mv, the refined vectors (xevdm_util.c:246-247): the syntax of later CUs of the SAME picture depends on the refinement search.  The
        // front end then searches itself (TileCoder::commit, dmvr_search.h) on the reference samples the caller registers (xhost_parser_set_ref_luma)
        if (unsupported) return fail("the stream uses tools this front end does not parse (sps_suco_flag)");
        // xevdm_eco.c:1920-1961: POC lsb width (tool_pocs), the sub-GOP description unless both tools are on, and either the sliding-window size or the RPL candidates
        s.tool_rpl = rpl; s.tool_pocs = pocs;
        s.log2_sub_gop = s.log2_ref_gap = 0;
        if (pocs) { s.poc_lsb_bits = (int)br.ue() + 4; if (s.poc_lsb_bits > 16) return fail("bad SPS: log2_max_pic_order_cnt_lsb"); }
        if (!rpl || !pocs) {
            s.log2_sub_gop = (int)br.ue();
            if (s.log2_sub_gop == 0) s.log2_ref_gap = (int)br.ue();
            if (s.log2_sub_gop > 5) return fail("bad SPS");
        }
        if (!rpl) s.max_num_ref_pics = (int)br.ue();
        else {
            s.max_num_ref_pics = std::min((int)br.ue() + 1, XGPU_MAX_REFS);      // sps_max_dec_pic_buffering_minus1
            br.get1();                                                          // long_term_ref_pics_flag (no syntax of its own in this decoder)
            if (br.get1()) return fail("rpl1_same_as_rpl0_flag is not supported (nor by the reference)");
            for (int l = 0; l < 2; l++) {
                s.n_rpl[l] = (int)br.ue();
                if (br.overrun || s.n_rpl[l] > 32) return fail("bad SPS: num_ref_pic_lists_in_sps");
                for (int i = 0; i < s.n_rpl[l]; i++) if (!read_rpl(br, s.rpls[l][i])) return fail("bad SPS: reference picture list");
            }
        }
        s.crop[0] = s.crop[1] = s.crop[2] = s.crop[3] = 0;
        if (br.get1()) for (int i = 0; i < 4; i++) s.crop[i] = (int)br.ue();      // left, right, top, bottom (handed to the caller's output stage)
        s.cqt = br.get1() != 0;
        if (s.cqt) {
            // chroma_qp_table_struct (xevd_eco.c:1361-1376) -> mapping tables (xevd_derived_chroma_qp_mapping_tables, xevd_tbl.c:375-425): pivot
            // points, linear interpolation with rounding between them, slope 1 outside, clipped to the QP range
            const int same = br.get1(), global_offset = br.get1(), off = 6 * (s.bd_c - 8), start = global_offset ? 16 : -off;
            for (int c = 0; c < (same ? 1 : 2); c++) {
                const int np = (int)br.ue() + 1;
                if (np < 1 || np > 58 + off) return fail("bad chroma QP table");
                int din[96], qin[96], qout[96];                      // np <= 58 + 24
                for (int j = 0; j < np; j++) {
                    din[j] = (int)br.get(6);
                    const int dout = br.se();
                    qin[j] = j ? qin[j - 1] + din[j] + 1 : start + din[0];
                    qout[j] = j ? qout[j - 1] + din[j] + 1 + dout : start + din[0] + dout;
                    if (qin[j] < -off || qin[j] > 57 || qout[j] < -off || qout[j] > 57) return fail("bad chroma QP table");
                }
                int8_t *t = s.cq[c] + off;                          // t[qp], qp = -off .. 57
                t[qin[0]] = (int8_t)qout[0];
                for (int k = qin[0] - 1; k >= -off; k--) t[k] = (int8_t)std::min(std::max(t[k + 1] - 1, -off), 57);
                for (int j = 0; j + 1 < np; j++) {
                    const int den = din[j + 1] + 1, rnd = den >> 1;
                    for (int k = qin[j] + 1, m = 1; k <= qin[j + 1]; k++, m++) t[k] = (int8_t)(t[qin[j]] + ((qout[j + 1] - qout[j]) * m + rnd) / den);
                }
                for (int k = qin[np - 1] + 1; k <= 57; k++) t[k] = (int8_t)std::min(std::max(t[k - 1] + 1, -off), 57);
            }
            if (same) memcpy(s.cq[1], s.cq[0], sizeof(s.cq[0]));
        }
        br.get1();      // vui_parameters_present_flag: the VUI (display metadata, xevd_eco.c:1226-1304) is the last SPS element and is not needed here
        if (br.overrun || (s.width & 7) || (s.height & 7) || s.width <= 0 || s.height <= 0 || s.width > 16384 || s.height > 16384 ||
            s.bd_l < 8 || s.bd_l > 12 || s.bd_c < 8 || s.bd_c > 12 || s.max_num_ref_pics < 0 || s.max_num_ref_pics > XGPU_MAX_REFS) return fail("bad SPS");
        // a new geometry / bit depth invalidates every stored picture (their motion fields have the old SCU grid): drop the DPB and take
        // nothing but an IDR picture until one arrives
        if (st.have_sps && (s.width != st.sps.width || s.height != st.sps.height || s.bd_l != st.sps.bd_l || s.bd_c != st.sps.bd_c)) {
            st.dpb.clear(); st.refp[0].clear(); st.refp[1].clear();
            st.need_idr = true;
        }
        st.sps = s;
        st.have_sps = true;
        return XGPU_OK;
    }
    int parse_pps(BitReader &br)
    {
        Pps q = st.pps;                                  // parsed into a copy: a damaged PPS leaves the active one untouched (as parse_sps does)
        br.ue(); br.ue();                                // pps id, sps id
        q.default_active[0] = (int)br.ue() + 1; q.default_active[1] = (int)br.ue() + 1;      // num_ref_idx_default_active_minus1
        br.ue();                                         // additional_lt_poc_lsb_len
        q.rpl1_idx_present = br.get1();
        q.tile_cols = q.tile_rows = q.tile_uniform = 1; q.across_tiles = 0;      // one tile: the flag is not sent and reads as 0
        if (!br.get1()) {                                // single_tile_in_pic_flag == 0 (xevdm_eco.c:2021-2039)
            q.tile_cols = (int)br.ue() + 1; q.tile_rows = (int)br.ue() + 1;
            if (br.overrun || q.tile_cols > XGPU_MAX_TILE_COLS || q.tile_rows > XGPU_MAX_TILE_ROWS) return fail("bad PPS: tile grid");
            q.tile_uniform = br.get1();
            if (!q.tile_uniform) {
                for (int i = 0; i + 1 < q.tile_cols; i++) q.tile_col_w[i] = (int)br.ue() + 1;
                for (int i = 0; i + 1 < q.tile_rows; i++) q.tile_row_h[i] = (int)br.ue() + 1;
            }
            q.across_tiles = br.get1();
            q.offset_bits = (int)br.ue() + 1;
            if (q.offset_bits > 32) return fail("bad PPS: tile_offset_lens_minus1");
        }
        q.id_bits = (int)br.ue() + 1;                    // tile_id_len_minus1
        if (q.id_bits > 15) return fail("bad PPS: tile_id_len_minus1");
        if (br.get1())                                   // explicit_tile_id_flag: tile_id_val[row][col] (xevdm_eco.c:2042-2052).  The reference decoder reads the values and never
            for (int i = 0; i < q.tile_cols * q.tile_rows; i++) br.get(q.id_bits);      // looks at them again - first / last_tile_id stay raster indices (set_tile_info) - so neither do we
        q.dra_on = br.get1();                       // pic_dra_enabled_flag, pic_dra_aps_id (xevdm_eco.c:2054-2060)
        if (q.dra_on) q.dra_aps_id = (int)br.get(5);
        q.arbitrary_slices = br.get1();                  // arbitrary_slice_present_flag
        q.constrained_intra = br.get1();
        q.cu_qp_delta = br.get1();
        q.qp_delta_area = 6;
        if (q.cu_qp_delta) { q.qp_delta_area = (int)br.ue() + 6; if (q.qp_delta_area > 14) return fail("bad PPS: cu_qp_delta_area"); }
        if (br.overrun) return fail("bad PPS");
        st.pps = q;
        st.have_pps = true;
        return XGPU_OK;
    }
    // the tiles of a slice, in the order of their entry points (xevdm_eco.c:2520-2550, set_tile_info src_main/xevdm.c:2185-2236): one tile, the rectangle of
    // tiles between first_tile_id and last_tile_id (it may wrap around the picture's right / bottom border), or an arbitrary ascending list.  br stands behind
    // slice_pic_parameter_set_id
    int slice_tile_list(BitReader &br, std::vector<int> &tl)
    {
        const int n_tiles = st.pps.tile_cols * st.pps.tile_rows;
        tl.assign(1, 0);
        if (n_tiles > 1) {
            const int single = br.get1(), first = (int)br.get(st.pps.id_bits);
            if (first >= n_tiles) return fail("bad slice header: first_tile_id");
            tl[0] = first;
            if (!single) {
                const int arbitrary = st.pps.arbitrary_slices ? br.get1() : 0;
                if (!arbitrary) {
                    const int last = (int)br.get(st.pps.id_bits), wt = st.pps.tile_cols, ht = st.pps.tile_rows;
                    if (last >= n_tiles) return fail("bad slice header: last_tile_id");
                    int delta = last - first;
                    if (last < first) delta += first % wt > last % wt ? n_tiles + wt : n_tiles;
                    else if (first % wt > last % wt) delta += wt;
                    const int ws = delta % wt + 1, hs = delta / wt + 1;
                    if (ws > wt || hs > ht) return fail("bad slice header: tile rectangle");
                    tl.clear();
                    for (int r = 0; r < hs; r++) for (int c2 = 0; c2 < ws; c2++) tl.push_back(((first / wt + r) % ht) * wt + (first % wt + c2) % wt);
                } else {
                    const uint32_t more = br.ue() + 1;             // num_remaining_tiles_in_slice_minus1 + 1
                    if (br.overrun || more >= (uint32_t)n_tiles) return fail("bad slice header: arbitrary slice");
                    for (uint32_t i = 0; i < more; i++) {
                        const uint32_t d = br.ue();
                        if (br.overrun || d >= (uint32_t)n_tiles || tl.back() + (int)d + 1 >= n_tiles) return fail("bad slice header: delta_tile_id");
                        tl.push_back(tl.back() + (int)d + 1);
                    }
                }
            }
        }
        return XGPU_OK;
    }
    int parse_slice(BitReader &br, int nut, int tid, xhost_picture *out)
    {
        if (!st.have_sps || !st.have_pps) return fail("slice before SPS/PPS");
        if (st.need_idr && nut != NUT_IDR) return fail("the sequence parameters changed: waiting for an IDR picture");
        st.need_idr = false;
        Slice &sh = st.sh;
        br.ue();                                         // slice_pic_parameter_set_id
        const int n_tiles = st.pps.tile_cols * st.pps.tile_rows;
        if (n_tiles > 1 && !st.sps.profile_main) return fail("tiles in a Baseline stream are not supported");
        std::vector<int> tl;
        { const int rc = slice_tile_list(br, tl); if (rc != XGPU_OK) return rc; }
        const int n_slice_tiles = (int)tl.size();
        const bool first_slice = pic_tiles_left == 0;
        if (!first_slice) {
            // The reference decoder takes slice NALs as parts of one picture until every CTU is covered (ctx->num_ctb, src_main/xevdm.c:2995-2999, 3106, 3138); without
            // sps_pocs_flag it derives a NEW picture order count for every slice NAL (xevd_poc_derivation is called per NAL, :3030-3040), so such streams need the flag
            if (!st.sps.tool_pocs) return fail("several slices per picture need sps_pocs_flag (the reference decoder counts a picture per slice NAL otherwise)");
            if (nut != pic_nut || tid != pic_tid) return fail("the slices of a picture differ in NAL unit type or temporal id");
        }
        for (int t : tl) if (!first_slice && tile_done[(size_t)t]) return fail("a tile is coded twice in one picture");
        if (!first_slice && n_slice_tiles > pic_tiles_left) return fail("a tile is coded twice in one picture");
        sh.type = (int)br.ue();
        if (sh.type < 0 || sh.type > 2) return fail("bad slice type");
        if (nut == NUT_IDR) br.get1();                   // no_output_of_prior_pics_flag
        sh.mmvd_group = (st.sps.tool_mmvd && sh.type != XHOST_SLICE_I) ? br.get1() : 0;
        sh.alf_on = sh.alf_chroma_idc = sh.alf_ctb_map = 0;
        if (st.sps.tool_alf) {                           // xevdm_eco.c:2608-2657 (4:2:0)
            sh.alf_on = br.get1();
            if (sh.alf_on) {
                sh.aps_id_y = (int)br.get(5);
                sh.alf_ctb_map = br.get1();
                sh.alf_chroma_idc = (int)br.get(2);
                if (sh.alf_chroma_idc) sh.aps_id_ch = (int)br.get(5);
            }
        }
        sh.poc_lsb = 0; sh.rpl[0] = Rpl(); sh.rpl[1] = Rpl();
        if (nut != NUT_IDR) {                            // xevdm_eco.c:2658-2733
            if (st.sps.tool_pocs) sh.poc_lsb = (int)br.get(st.sps.poc_lsb_bits);
            if (st.sps.tool_rpl) {
                int from_sps[2] = { 0, 0 }, idx[2] = { 0, 0 };
                for (int l = 0; l < 2; l++) {
                    if (l == 0 || st.pps.rpl1_idx_present) from_sps[l] = st.sps.n_rpl[l] > 0 ? br.get1() : 0;
                    else from_sps[1] = from_sps[0];
                    if (from_sps[l]) {
                        if (l == 0 || st.pps.rpl1_idx_present) { if (st.sps.n_rpl[l] > 1) idx[l] = (int)br.ue(); }
                        else idx[1] = idx[0];
                        // (the reference copies list 0 only when the SPS holds more than one candidate - with exactly one it keeps the previous slice's list)
                        if (idx[l] >= st.sps.n_rpl[l] || (l == 0 && st.sps.n_rpl[0] == 1)) return fail("reference picture list index");
                        sh.rpl[l] = st.sps.rpls[l][idx[l]];
                    } else if (!read_rpl(br, sh.rpl[l])) return fail("bad slice header: reference picture list");
                }
            }
        }
        sh.rpl[0].active = st.pps.default_active[0]; sh.rpl[1].active = st.pps.default_active[1];
        if (sh.type != XHOST_SLICE_I && br.get1()) { sh.rpl[0].active = (int)br.ue() + 1; if (sh.type == XHOST_SLICE_B) sh.rpl[1].active = (int)br.ue() + 1; }      // num_ref_idx_active override (only used with tool_rpl)
        sh.tmvp_assigned = sh.col_list = sh.col_src_list = sh.col_ref = 0;
        if (sh.type != XHOST_SLICE_I && st.sps.tool_admvp && (sh.tmvp_assigned = br.get1())) {                // xevdm_eco.c:2748-2760
            if (sh.type == XHOST_SLICE_B) { sh.col_list = br.get1(); sh.col_src_list = br.get1(); }
            sh.col_ref = br.get1();
        }
        sh.deblock = br.get1();
        sh.alpha_off = sh.beta_off = 0;
        if (sh.deblock && st.sps.tool_addb) { sh.alpha_off = br.se(); sh.beta_off = br.se(); }      // xevdm_eco.c:2767-2772
        sh.qp = (int)br.get(6);
        sh.qp_u_offset = br.se(); sh.qp_v_offset = br.se();
        std::vector<size_t> tile_size((size_t)n_slice_tiles, 0);      // entry_point_offset_minus1 + 1: bytes of every tile of the slice but the last (xevdm_eco.c:2789-2795)
        for (int i = 0; i + 1 < n_slice_tiles; i++) tile_size[(size_t)i] = (size_t)br.get(st.pps.offset_bits) + 1;
        while (!br.aligned()) if (br.get1()) return fail("slice header alignment");
        if (br.overrun || sh.qp > 51) return fail("bad slice header");
        st.derive_poc(nut == NUT_IDR, tid);
        if (!first_slice && st.poc != pic_poc) return fail("a slice of another picture before every tile of the picture was coded");
        if (sh.type == XHOST_SLICE_I) st.last_intra_poc = st.poc;
        if (!st.build_ref_lists(nut == NUT_IDR, !first_slice)) return fail("a reference picture list names a picture that is not in the DPB");
        if (sh.type != XHOST_SLICE_I && st.refp[0].empty()) return fail("P/B slice without a reference picture");
        if (sh.type == XHOST_SLICE_B && st.refp[1].empty()) return fail("B slice without a list-1 reference picture");
        for (int l = 0; l < 2; l++)
            for (const RefPic *r : st.refp[l])
                if (r->mv0.size() != (size_t)(st.sps.width >> 2) * (st.sps.height >> 2) * 2) return fail("reference picture of another geometry");
        const int W = st.sps.width, H = st.sps.height, w_ctu = (W + 63) >> 6, h_ctu = (H + 63) >> 6;
        if (first_slice) {
            st.pic.reset(st.sps.width, st.sps.height, st.sps.host_dmvr());
            if (!st.setup_tiles()) return fail("the tile grid of the PPS does not fit the picture");
            st.alf_ctb_flag.assign((size_t)w_ctu * h_ctu, 1);
            tile_done.assign((size_t)n_tiles, 0);
            pic_nut = nut; pic_tid = tid; pic_poc = st.poc; pic_inter = 0; pic_qp = sh.qp;
        }
        // The backend takes ONE pair of reference lists per picture (and the reference decoder deblocks the whole picture with the lists of its last P / B slice,
        // ctx->refp at src_main/xevdm.c:3138-3199): the P / B slices of a picture must name the same pictures; I slices may be mixed in
        if (sh.type != XHOST_SLICE_I) {
            std::vector<int> lp;
            for (int l = 0; l < 2; l++) { lp.push_back(-1 - l); for (const RefPic *r : st.refp[l]) lp.push_back(r->poc); }
            if (pic_inter && lp != pic_lists) return fail("slices of one picture with different reference picture lists are not supported");
            pic_lists = lp; pic_inter = 1;
        }
        if (st.sps.tool_htdf && sh.qp != pic_qp) return fail("slices of one picture with different slice QPs together with HTDF are not supported");

        // ---- tile data (xevdm_dec_slice + xevd_tile_eco, src_main/xevdm.c:2363-2461, 2614-2718): every tile is its own arithmetic-coder
        //      run - contexts, QP predictor and motion history start afresh - at the byte offset the slice header gave; tiles share nothing but
        //      the picture maps (disjoint regions), so they are parsed in parallel when the caller allows threads ----
        while ((int)tiles.size() < n_tiles) tiles.emplace_back(new TileParser(st));
        std::vector<size_t> tile_pos((size_t)n_slice_tiles, br.pos);
        for (int t = 1; t < n_slice_tiles; t++) tile_pos[(size_t)t] = tile_pos[(size_t)t - 1] + tile_size[(size_t)t - 1] * 8;
        std::vector<int> tile_rc((size_t)n_slice_tiles, XGPU_OK);
        parallel_for(n_slice_tiles, [&](int i) { const int t = tl[(size_t)i]; tile_rc[(size_t)i] = tiles[(size_t)t]->parse_tile(br, tile_pos[(size_t)i], t % st.grid.n_cols, t / st.grid.n_cols); });
        for (int i = 0; i < n_slice_tiles; i++) if (tile_rc[(size_t)i] != XGPU_OK) { err = tiles[(size_t)tl[(size_t)i]]->err; pic_tiles_left = 0; return tile_rc[(size_t)i]; }
        if (first_slice) pic_tiles_left = n_tiles;
        for (int t : tl) tile_done[(size_t)t] = 1;
        pic_tiles_left -= n_slice_tiles;
        if (pic_tiles_left > 0) return XGPU_OK;                // more slices of this picture follow; the in-loop filters run with the LAST slice's header, as in the
                                                                 // reference decoder (ctx->sh at src_main/xevdm.c:3138-3199: deblocking switch and offsets, chroma QP offsets, ALF)
        merged_arena = nullptr;
        if (n_tiles == 1) { cur = &tiles[0]->batch; n_coef = tiles[0]->n_coef; }
        else {
            // one batch for the backend: the tiles' arrays one after the other, coefficient offsets and CTU starts moved along
            std::vector<size_t> cu0((size_t)n_tiles + 1, 0), cf0((size_t)n_tiles + 1, 0), ct0((size_t)n_tiles + 1, 0);
            for (int t = 0; t < n_tiles; t++) {
                cu0[(size_t)t + 1] = cu0[(size_t)t] + tiles[(size_t)t]->batch.x.size();
                cf0[(size_t)t + 1] = cf0[(size_t)t] + tiles[(size_t)t]->n_coef;
                ct0[(size_t)t + 1] = ct0[(size_t)t] + tiles[(size_t)t]->batch.ctu_start.size();
            }
            if (cf0[(size_t)n_tiles] > 0xFFFFFFFFull) return fail("coefficient arena beyond 32-bit offsets");
            Batch &m = merged;
            const size_t n = cu0[(size_t)n_tiles];
            // the caller's arena (pinned memory of the backend: xgpu_batch_create then sends the coefficients from where they lie) of the slot this picture
            // will be handed out in; without one - or when the allocator has nothing yet - the merged batch's own vector
            Held &slot = held[n_handed % held.size()];
            int16_t *arena = nullptr;
            if (arena_alloc) {
                const size_t need = std::max(cf0[(size_t)n_tiles], (size_t)8);
                if (slot.arena_cap < need) {
                    if (slot.arena && arena_release) arena_release(arena_user, slot.arena);
                    slot.arena_cap = need + need / 4;
                    slot.arena = (int16_t *)arena_alloc(arena_user, slot.arena_cap * sizeof(int16_t));
                    if (!slot.arena) slot.arena_cap = 0;
                }
                arena = slot.arena;
            }
            m.x.resize(n); m.y.resize(n); m.log2w.resize(n); m.log2h.resize(n); m.pred_mode.resize(n); m.qp.resize(n * 3); m.cbf.resize(n); m.ipm.resize(n * 2);
            m.ats.resize(n); m.ats_inter.resize(n); m.dmvr.resize(n); m.tree.resize(n); m.has_tree = false;
            for (int t = 0; t < n_tiles; t++) m.has_tree |= tiles[(size_t)t]->batch.has_tree;
            m.affine.resize(st.sps.tool_affine ? n : 0); m.affine_mv.resize(st.sps.tool_affine ? n * 12 : 0); m.refi.resize(n * 2); m.mv.resize(n * 4); m.coef_off.resize(n);
            if (!arena) m.coef.resize(cf0[(size_t)n_tiles]);
            m.ctu_start.resize(ct0[(size_t)n_tiles]);
            int16_t *coef_dst = arena ? arena : m.coef.data();
            merged_arena = arena;
            parallel_for(n_tiles, [&](int t) {
                const Batch &b = tiles[(size_t)t]->batch;
                const size_t o = cu0[(size_t)t], k = b.x.size();
                auto put = [&](auto &dst, const auto &src, size_t per) { if (k) memcpy(dst.data() + o * per, src.data(), k * per * sizeof(src[0])); };
                put(m.x, b.x, 1); put(m.y, b.y, 1); put(m.log2w, b.log2w, 1); put(m.log2h, b.log2h, 1); put(m.pred_mode, b.pred_mode, 1); put(m.qp, b.qp, 3);
                put(m.cbf, b.cbf, 1); put(m.ipm, b.ipm, 2); put(m.ats, b.ats, 1); put(m.ats_inter, b.ats_inter, 1); put(m.dmvr, b.dmvr, 1); put(m.tree, b.tree, 1); put(m.refi, b.refi, 2); put(m.mv, b.mv, 4);
                if (st.sps.tool_affine) { put(m.affine, b.affine, 1); put(m.affine_mv, b.affine_mv, 12); }
                for (size_t i = 0; i < k; i++) m.coef_off[o + i] = b.coef_off[i] + (uint32_t)cf0[(size_t)t];
                if (tiles[(size_t)t]->n_coef) memcpy(coef_dst + cf0[(size_t)t], b.coef.data(), tiles[(size_t)t]->n_coef * sizeof(int16_t));
                for (size_t i = 0; i < b.ctu_start.size(); i++) m.ctu_start[ct0[(size_t)t] + i] = b.ctu_start[i] + (uint32_t)o;
            });
            cur = &merged; n_coef = cf0[(size_t)n_tiles];
        }
        cur->ctu_start.push_back((uint32_t)cur->x.size());
        Held &hd = held[n_handed++ % held.size()];
        std::swap(hd.batch, *cur);                       // the parser's working vectors take over the slot's old storage (cleared / resized at their next use)
        cur = &hd.batch;
        Batch &batch = *cur;

        // ---- hand-over ----
        memset(out, 0, sizeof(*out));
        out->width = W; out->height = H; out->bit_depth_luma = st.sps.bd_l; out->bit_depth_chroma = st.sps.bd_c;
        out->poc = st.poc; out->temporal_id = tid; out->slice_type = sh.type; out->is_idr = nut == NUT_IDR; out->is_ref = st.is_ref_picture();
        for (int l = 0; l < 2; l++) {
            out->num_refp[l] = (int)st.refp[l].size();
            for (size_t i = 0; i < st.refp[l].size(); i++) out->refp_poc[i][l] = st.refp[l][i]->poc;
        }
        out->slice_qp = sh.qp; out->qp_u_offset = sh.qp_u_offset; out->qp_v_offset = sh.qp_v_offset; out->deblock_on = sh.deblock;
        out->profile_main = st.sps.profile_main; out->tool_iqt = st.sps.tool_iqt; out->tool_ats = st.sps.tool_ats; out->tool_addb = st.sps.tool_addb;
        out->deblock_alpha_offset = sh.alpha_off; out->deblock_beta_offset = sh.beta_off;
        out->tool_alf = st.sps.tool_alf; out->alf_on = sh.alf_on; out->tool_eipd = st.sps.tool_eipd; out->tool_admvp = st.sps.tool_admvp;
        out->dra_lut[0] = out->dra_lut[1] = out->dra_lut[2] = nullptr;
        if (st.sps.tool_dra && st.pps.dra_on) {          // what xevd_pull applies to its copy of this picture (xevd_apply_filter, xevdm.c:3305-3349)
            const DraAps &d = st.dra_aps[st.pps.dra_aps_id & 31];
            if (!d.valid) return fail("the PPS names a DRA parameter set that was not sent");
            if (st.sps.bd_l > 10) return fail("DRA tables cover 10 bits");
            const int off = 6 * (st.sps.bd_c - 8);
            int8_t dflt[2][96];
            const int8_t *cq[2];
            for (int c = 0; c < 2; c++) {
                if (st.sps.cqt) cq[c] = st.sps.cq[c] + off;
                else { for (int q = -off; q <= 57; q++) dflt[c][q + off] = q >= 0 ? (st.sps.tool_iqt ? k_chroma_qp_main : k_chroma_qp)[q] : 0; cq[c] = dflt[c] + off; }
            }
            hd.dra.resize(3 * 1024);
            dra_build_luts(d, st.sps.bd_l, cq[0], cq[1], off, hd.dra.data());
            for (int c = 0; c < 3; c++) out->dra_lut[c] = hd.dra.data() + 1024 * c;
        }
        for (int i = 0; i < 4; i++) out->crop[i] = st.sps.crop[i];
        out->chroma_qp_table[0] = st.sps.cqt ? st.sps.cq[0] : nullptr; out->chroma_qp_table[1] = st.sps.cqt ? st.sps.cq[1] : nullptr;
        if (sh.alf_on) {
            if (!st.alf_finalise()) return fail("slice refers to an ALF parameter set that was not sent");
            out->alf.enable[0] = 1; out->alf.enable[1] = sh.alf_chroma_idc & 1; out->alf.enable[2] = (sh.alf_chroma_idc >> 1) & 1;
            memcpy(hd.alf_luma, st.alf_luma_final, sizeof(hd.alf_luma)); memcpy(hd.alf_chroma, st.alf_chroma_final, sizeof(hd.alf_chroma));
            hd.ctb = st.alf_ctb_flag;
            out->alf.luma_coef = &hd.alf_luma[0][0]; out->alf.chroma_coef = hd.alf_chroma;
            out->alf.ctb_flag = hd.ctb.data(); out->alf.across_tiles = st.pps.across_tiles; out->alf.tiles = n_tiles > 1 ? &hd.grid : nullptr;
        }
        std::vector<int> released;
        st.store_picture(nut == NUT_IDR, released, pic_inter);
        out->n_release = (int)std::min(released.size(), (size_t)32);
        for (int i = 0; i < out->n_release; i++) out->release_poc[i] = released[(size_t)i];
        xgpu_cu_batch &b = out->batch;
        b.n_cu = (int)batch.x.size();
        b.x = batch.x.data(); b.y = batch.y.data(); b.log2w = batch.log2w.data(); b.log2h = batch.log2h.data();
        b.pred_mode = batch.pred_mode.data(); b.refi = batch.refi.data(); b.mv = batch.mv.data(); b.qp = batch.qp.data();
        b.cbf = batch.cbf.data(); b.ipm = batch.ipm.data(); b.coef_off = batch.coef_off.data();
        if (st.sps.tool_ats) { b.ats = batch.ats.data(); b.ats_inter = batch.ats_inter.data(); }
        if (st.sps.tool_affine) { b.affine = batch.affine.data(); b.affine_mv = batch.affine_mv.data(); }
        if (batch.has_tree) b.tree = batch.tree.data();
        if (batch.coef.empty()) batch.coef.push_back(0);
        b.coef = merged_arena ? merged_arena : batch.coef.data(); b.n_coef = batch.x.empty() ? 0 : n_coef;
        b.n_ctu = w_ctu * h_ctu; b.ctu_cu_start = batch.ctu_start.data();
        b.constrained_intra_pred = st.pps.constrained_intra;
        hd.grid = st.grid;
        b.tiles = n_tiles > 1 ? &hd.grid : nullptr;
        b.htdf_slice_qp = st.sps.tool_htdf ? sh.qp : 0;
        // sps->tool_dmvr: the merge-mode flags, and how many 16x16 sub-blocks of candidates (flag, two references, at least 8x8 - the order of
        // xgpu_batch_dmvr_mvs) the backend will report vectors for; they go back in through xhost_parser_set_dmvr_mvs before the next picture
        out->n_dmvr_sub = 0;
        last_poc = st.poc; last_stored = st.is_ref_picture();
        out->needs_ref_luma = st.sps.host_dmvr() && st.is_ref_picture();
        if (st.sps.tool_dmvr) b.dmvr = batch.dmvr.data();
        if (st.sps.tool_dmvr && !st.sps.host_dmvr()) {      // (host-side refinement: nothing comes back from the backend)
            for (int i = 0; i < b.n_cu; i++)
                if (batch.dmvr[(size_t)i] && batch.refi[(size_t)i * 2] >= 0 && batch.refi[(size_t)i * 2 + 1] >= 0 && batch.log2w[(size_t)i] >= 3 && batch.log2h[(size_t)i] >= 3)
                    out->n_dmvr_sub += (batch.log2w[(size_t)i] > 4 ? 1 << (batch.log2w[(size_t)i] - 4) : 1) * (batch.log2h[(size_t)i] > 4 ? 1 << (batch.log2h[(size_t)i] - 4) : 1);
        }
        last_n_dmvr = out->n_dmvr_sub;
        return 1;
    }
    int last_poc = 0, last_n_dmvr = 0;
    bool last_stored = false;
    // the picture being assembled from several slice NALs: tiles still to come, and what its slices must agree on
    int pic_tiles_left = 0, pic_nut = 0, pic_tid = 0, pic_poc = 0, pic_inter = 0, pic_qp = 0;
    std::vector<uint8_t> tile_done;
    std::vector<int> pic_lists;
    // the refined vectors of the picture just handed out: what the reference keeps in map_mv for the temporal candidates of later pictures
    // (dmvr_mv -> map_mv, src_main/xevdm_util.c:4327-4338); the picture's own CUs, the history and the deblocking filter use the unrefined ones
    int set_dmvr_mvs(const int16_t *mv, int n)
    {
        if (n != last_n_dmvr || (n > 0 && !mv)) return fail("xhost_parser_set_dmvr_mvs: not the sub-block count of the last picture");
        if (!last_stored || n == 0 || st.sps.host_dmvr()) return 0;
        RefPic *r = nullptr;
        for (RefPic &q : st.dpb) if (q.poc == last_poc) r = &q;
        if (!r || r->mv.empty()) return 0;
        const int ws = st.sps.width >> 2;
        const Batch &batch = *cur;
        for (size_t i = 0; i < batch.x.size(); i++) {
            if (!(batch.dmvr[i] && batch.refi[i * 2] >= 0 && batch.refi[i * 2 + 1] >= 0 && batch.log2w[i] >= 3 && batch.log2h[i] >= 3)) continue;
            const int w = 1 << batch.log2w[i], h = 1 << batch.log2h[i], dx = std::min(w, 16), dy = std::min(h, 16);
            for (int sy = 0; sy < h; sy += dy) for (int sx = 0; sx < w; sx += dx, mv += 4)
                for (int v = 0; v < dy >> 2; v++) for (int u = 0; u < dx >> 2; u++)
                    memcpy(&r->mv[((size_t)((batch.y[i] + sy) >> 2) + v) * ws * 4 + ((size_t)((batch.x[i] + sx) >> 2) + u) * 4], mv, sizeof(int16_t) * 4);
        }
        return 0;
    }
};

extern "C" xhost_parser *xhost_parser_open(const uint8_t *bytes, size_t size)
{
    if (!bytes) return nullptr;
    xhost_parser *p = new xhost_parser();
    p->data.assign(bytes, bytes + size);
    return p;
}
// the decoded luma samples of the picture with this POC, for the front end's own refinement search (Sps::host_dmvr)
static int set_ref_luma(Stream &st, int poc, const int16_t *plane, int stride)
{
    if (!plane || stride <= 0) return XGPU_ERR_INVALID_ARGUMENT;
    for (RefPic &r : st.dpb) if (r.poc == poc) { r.luma = plane; r.luma_stride = stride; return XGPU_OK; }
    return XGPU_OK;                                      // not kept as a reference: nothing will read it
}
extern "C" int xhost_parser_set_ref_luma(xhost_parser *p, int poc, const int16_t *plane, int stride) { return p ? set_ref_luma(p->st, poc, plane, stride) : XGPU_ERR_INVALID_ARGUMENT; }
extern "C" int xhost_parser_set_depth(xhost_parser *p, int depth)
{
    if (!p || depth < 1 || depth > 8 || p->n_handed) return XGPU_ERR_INVALID_ARGUMENT;      // before the first picture
    p->held = std::vector<xhost_parser::Held>((size_t)depth);
    return XGPU_OK;
}
extern "C" int xhost_parser_set_arena(xhost_parser *p, void *(*alloc)(void *, size_t), void (*release)(void *, void *), void *user)
{
    if (!p || !alloc || !release || p->n_handed) return XGPU_ERR_INVALID_ARGUMENT;
    p->arena_alloc = alloc; p->arena_release = release; p->arena_user = user;
    return XGPU_OK;
}
extern "C" int xhost_parser_set_threads(xhost_parser *p, int n) { if (!p || n < 1) return XGPU_ERR_INVALID_ARGUMENT; p->n_threads = std::min(n, 64); return XGPU_OK; }
extern "C" int xhost_parser_set_dmvr_mvs(xhost_parser *p, const int16_t *mv, int n_sub) { return p ? p->set_dmvr_mvs(mv, n_sub) : XHOST_ERR_MALFORMED; }
extern "C" const char *xhost_parser_error(const xhost_parser *p) { return p ? p->err.c_str() : "null parser"; }
extern "C" void xhost_parser_close(xhost_parser *p) { delete p; }

// one NAL unit (2-byte header + payload, no length prefix): 1 = `out` holds a picture, 0 = consumed without a picture, < 0 error
static int parser_nal(xhost_parser *p, const uint8_t *nal, size_t len, xhost_picture *out, bool peek_signature)
{
    if (len < 2) return p->fail("bad NAL length");
    BitReader br;
    br.p = nal; br.size = len;
    if (br.get1()) return p->fail("forbidden_zero_bit");
    const int nut = (int)br.get(6) - 1, tid = (int)br.get(3);
    if (br.get(5) != 0 || br.get1() != 0) return p->fail("reserved NAL header bits");
    if (nut == NUT_SPS) return p->parse_sps(br);
    if (nut == NUT_PPS) return p->parse_pps(br);
    if (nut == NUT_IDR || nut == NUT_NONIDR) {
        const int rc = p->parse_slice(br, nut, tid, out);
        if (rc == 1) { out->has_md5 = 0; if (peek_signature) p->attach_signature(out); }
        return rc;
    }
    if (nut == 26) {                                     // APS (xevdm_eco_aps_gen, xevdm_eco.c:2082-2135)
        const int id = (int)br.get(5), type = (int)br.get(3);
        if (type == 1) {                                 // DRA parameters (xevdm_eco_dra_aps_param, :2319-2375)
            if (!p->st.have_sps) return p->fail("DRA APS before the SPS");
            DraAps d;
            if (br.get(4) != 4 || br.get(4) != 9) return p->fail("unsupported DRA descriptors");
            d.num_ranges = (int)br.ue() + 1;
            if (d.num_ranges > 32) return p->fail("bad DRA APS");
            const int equal = br.get1(), sh = std::max(0, p->st.sps.bd_l - 10);
            d.in_ranges[0] = (int)br.get(10) << sh;
            int delta[32];
            if (equal) delta[0] = (int)br.get(10);
            else for (int i = 0; i < d.num_ranges; i++) delta[i] = (int)br.get(10);
            for (int i = 0; i < d.num_ranges; i++) d.scale[i] = (int)br.get(13);
            d.cb_scale = (int)br.get(13); d.cr_scale = (int)br.get(13);
            d.table_idx = (int)br.ue();
            for (int i = 1; i <= d.num_ranges; i++) d.in_ranges[i] = d.in_ranges[i - 1] + (delta[equal ? 0 : i - 1] << sh);
            if (br.overrun || d.table_idx > 58) return p->fail("bad DRA APS");
            d.valid = true;
            p->st.dra_aps[id] = d;
            return XGPU_OK;
        }
        if (type != 0) return p->fail("unknown APS type");
        AlfAps a;
        if (!p->st.alf_aps_syntax<false>(&br, nullptr, a)) return p->fail("bad ALF APS");
        a.valid = true;
        p->st.alf_aps[id] = a;
        return XGPU_OK;
    }
    if (nut == NUT_SEI || nut == 27) return XGPU_OK;     // SEI (picture signatures are the caller's to check), filler data
    return p->fail("unsupported NAL unit type");
}

extern "C" int xhost_parser_next(xhost_parser *p, xhost_picture *out)
{
    if (!p || !out) return XGPU_ERR_INVALID_ARGUMENT;
    while (p->pos + 4 <= p->data.size()) {
        const uint8_t *d = p->data.data() + p->pos;
        const size_t len = ((size_t)d[0] << 24) | ((size_t)d[1] << 16) | ((size_t)d[2] << 8) | d[3];
        if (len < 2 || p->pos + 4 + len > p->data.size()) return p->fail("bad NAL length");
        p->pos += 4 + len;
        const int rc = parser_nal(p, d + 4, len, out, true);
        if (rc != XGPU_OK) return rc;
    }
    return 0;
}

// Picture boundaries without decoding anything (the GOP splitter of the work queue, xwq.cc): a slice NAL belongs to the picture of the slice before it until
// the PPS's tiles are all covered (ctx->num_ctb, src_main/xevdm.c:2995-2999).  The scanner reads PPS NALs (tile grid) and the first fields of slice headers.
struct xhost_scan { xhost_parser p; int tiles_left = 0; };
extern "C" xhost_scan *xhost_scan_open(void) { return new xhost_scan(); }
extern "C" void xhost_scan_close(xhost_scan *s) { delete s; }
// -> 0: not a slice NAL, 1: the first (or only) slice of a picture, 2: a further slice of the picture, < 0: malformed
extern "C" int xhost_scan_nal(xhost_scan *s, const uint8_t *nal, size_t len)
{
    if (!s || !nal || len < 2) return XGPU_ERR_INVALID_ARGUMENT;
    BitReader br;
    br.p = nal; br.size = len;
    br.get1();
    const int nut = (int)br.get(6) - 1;
    br.get(3); br.get(5); br.get1();
    if (nut == NUT_PPS) { const int rc = s->p.parse_pps(br); return rc < 0 ? rc : 0; }
    if (nut != NUT_IDR && nut != NUT_NONIDR) return 0;
    br.ue();                                             // slice_pic_parameter_set_id
    std::vector<int> tl;
    const int rc = s->p.slice_tile_list(br, tl);
    if (rc != XGPU_OK || br.overrun) return XHOST_ERR_MALFORMED;
    const bool first = s->tiles_left <= 0;
    if (first) s->tiles_left = s->p.st.pps.tile_cols * s->p.st.pps.tile_rows;
    s->tiles_left -= (int)tl.size();
    return first ? 1 : 2;
}

// NAL-at-a-time interface (what xevd_decode takes: one NAL unit without its length prefix, src_base/xevd.c:1786-2024)
extern "C" xhost_parser *xhost_parser_open_nal(void) { return new xhost_parser(); }
extern "C" int xhost_parser_nal(xhost_parser *p, const uint8_t *nal, size_t size, xhost_picture *out)
{
    if (!p || !nal || !out) return XGPU_ERR_INVALID_ARGUMENT;
    return parser_nal(p, nal, size, out, false);
}

// =============================================================================================================== writer
struct xhost_writer {
    xhost_stream_params sp;
    Stream st;
    TileCoder coder{st};          // the tiles are written one after the other
    bool plain_inter_seen = false; // a translational inter CU has been written (see the affine CUs in TreeWriter::node)
    std::vector<uint8_t> out;
    int n_pics = 0, last_tid = 0;
    bool headers_done = false;
    xhost_slice_alf next_alf = { 0, 0, 0, 0, 0, nullptr };
    std::vector<uint8_t> next_alf_ctb;
    std::vector<xhost_slice_desc> slices;      // xhost_writer_set_slices; empty = one slice with every tile
    bool arbitrary_slices = false;             // xhost_writer_set_arbitrary_slices: slices of several tiles list their tiles (arbitrary_slice_flag) instead of naming a rectangle

    void write_sps()
    {
        BitWriter bw;
        bw.ue(0); bw.put(sp.profile_main ? 1 : 0, 8); bw.put(0, 8); bw.put(0, 32); bw.put(0, 32);       // id, profile, level, toolset
        bw.ue(1); bw.ue((uint32_t)sp.width); bw.ue((uint32_t)sp.height);
        bw.ue((uint32_t)(sp.bit_depth - 8)); bw.ue((uint32_t)(sp.bit_depth - 8));
        if (!sp.profile_main) for (int i = 0; i < 13; i++) bw.put1(i == 11 ? (sp.cu_qp_delta ? 1 : 0) : 0);       // all tools off; dquant_flag with cu_qp_delta
        else {
            bw.put1(st.sps.btt);                         // sps_btt_flag: log2_ctu_size_minus5 (1 = 64), log2_min_cb_size_minus2, the three limits of the split table
            if (st.sps.btt) { bw.ue(1); for (int i = 0; i < 4; i++) bw.ue((uint32_t)st.sps.btt_raw[i]); }
            bw.put1(0);                                  // sps_suco_flag
            bw.put1(sp.tool_admvp ? 1 : 0);
            if (sp.tool_admvp) { bw.put1(sp.tool_affine ? 1 : 0); bw.put1(sp.tool_amvr ? 1 : 0); bw.put1(sp.tool_dmvr ? 1 : 0); bw.put1(sp.tool_mmvd ? 1 : 0); bw.put1(sp.tool_hmvp ? 1 : 0); }      // affine amvr dmvr mmvd hmvp
            bw.put1(sp.tool_eipd ? 1 : 0);
            if (sp.tool_eipd) { bw.put1(sp.ibc_log_max_size ? 1 : 0); if (sp.ibc_log_max_size) bw.ue((uint32_t)(sp.ibc_log_max_size - 2)); }      // ibc_flag, ibc_log_max_size - 2
            bw.put1(st.sps.tool_cm_init); if (st.sps.tool_cm_init) bw.put1(st.sps.tool_adcc);      // cm_init (+ adcc)
            bw.put1(sp.tool_iqt ? 1 : 0);
            if (sp.tool_iqt) bw.put1(sp.tool_ats ? 1 : 0);
            bw.put1(sp.tool_addb ? 1 : 0);
            bw.put1(sp.tool_alf ? 1 : 0);
            bw.put1(sp.tool_htdf ? 1 : 0);
            bw.put1(st.sps.tool_rpl); bw.put1(st.sps.tool_pocs); bw.put1(st.sps.dquant);      // rpl pocs dquant
            bw.put1(sp.tool_dra ? 1 : 0);
        }
        if (st.sps.tool_pocs) bw.ue((uint32_t)st.sps.poc_lsb_bits - 4);      // log2_max_pic_order_cnt_lsb_minus4
        if (!st.sps.tool_rpl || !st.sps.tool_pocs) {
            bw.ue((uint32_t)sp.log2_sub_gop_length);
            if (sp.log2_sub_gop_length == 0) bw.ue(0);      // log2_ref_pic_gap_length
        }
        if (!st.sps.tool_rpl) bw.ue((uint32_t)sp.max_num_ref_pics);
        else {
            bw.ue(15); bw.put1(0); bw.put1(0);           // sps_max_dec_pic_buffering_minus1, long_term_ref_pics_flag, rpl1_same_as_rpl0_flag
            for (int l = 0; l < 2; l++) { bw.ue((uint32_t)st.sps.n_rpl[l]); for (int i = 0; i < st.sps.n_rpl[l]; i++) write_rpl(bw, st.sps.rpls[l][i]); }
        }
        const bool crop = sp.crop[0] | sp.crop[1] | sp.crop[2] | sp.crop[3];
        bw.put1(crop);
        if (crop) for (int i = 0; i < 4; i++) bw.ue((uint32_t)sp.crop[i]);
        bw.put1(sp.cqt_present ? 1 : 0);
        if (sp.cqt_present) {
            bw.put1(sp.cqt_same ? 1 : 0); bw.put1(sp.cqt_global_offset ? 1 : 0);
            for (int c = 0; c < (sp.cqt_same ? 1 : 2); c++) {
                bw.ue((uint32_t)(sp.cqt_num_points[c] - 1));
                for (int j = 0; j < sp.cqt_num_points[c]; j++) { bw.put((uint32_t)sp.cqt_delta_in[c][j], 6); bw.se(sp.cqt_delta_out[c][j]); }
            }
        }
        bw.put1(0);                                      // no VUI
        bw.align_zero();
        write_nal(out, NUT_SPS, 0, bw);
    }
    void write_pps()
    {
        BitWriter bw;
        bw.ue(0); bw.ue(0); bw.ue(0); bw.ue(0); bw.ue(0);
        bw.put1(st.sps.tool_rpl);                        // rpl1_idx_present_flag
        const Pps &q = st.pps;
        bw.put1(q.tile_cols * q.tile_rows == 1);         // single_tile_in_pic_flag
        if (q.tile_cols * q.tile_rows > 1) {
            bw.ue((uint32_t)q.tile_cols - 1); bw.ue((uint32_t)q.tile_rows - 1); bw.put1(q.tile_uniform);
            if (!q.tile_uniform) {
                for (int i = 0; i + 1 < q.tile_cols; i++) bw.ue((uint32_t)q.tile_col_w[i] - 1);
                for (int i = 0; i + 1 < q.tile_rows; i++) bw.ue((uint32_t)q.tile_row_h[i] - 1);
            }
            bw.put1(q.across_tiles); bw.ue((uint32_t)q.offset_bits - 1);
        }
        bw.ue((uint32_t)q.id_bits - 1); bw.put1(0);      // tile_id_len_minus1, explicit_tile_id_flag
        bw.put1(sp.tool_dra ? 1 : 0);                    // pic_dra_enabled_flag
        if (sp.tool_dra) bw.put((uint32_t)sp.dra_aps_id, 5);
        st.pps.arbitrary_slices = arbitrary_slices ? 1 : 0;
        bw.put1(st.pps.arbitrary_slices);                // arbitrary_slice_present_flag
        bw.put1(0);                                      // constrained_intra_pred_flag
        bw.put1(sp.cu_qp_delta ? 1 : 0);
        if (sp.cu_qp_delta) bw.ue((uint32_t)(st.pps.qp_delta_area - 6));      // cu_qp_delta_area - 6
        bw.align_zero();
        write_nal(out, NUT_PPS, 0, bw);
    }
};

extern "C" xhost_writer *xhost_writer_open(const xhost_stream_params *sp)
{
    if (!sp || sp->width <= 0 || sp->height <= 0 || (sp->width & 7) || (sp->height & 7) || sp->bit_depth < 8 || sp->bit_depth > 12 ||
        sp->log2_sub_gop_length < 0 || sp->log2_sub_gop_length > 5 || sp->max_num_ref_pics < 1) return nullptr;
    xhost_writer *w = new xhost_writer();
    w->sp = *sp;
    Sps &s = w->st.sps;
    s.width = sp->width; s.height = sp->height; s.bd_l = s.bd_c = sp->bit_depth; s.max_num_ref_pics = sp->max_num_ref_pics;
    s.log2_sub_gop = sp->log2_sub_gop_length;
    s.profile_main = sp->profile_main ? 1 : 0;
    s.tool_iqt = s.profile_main && sp->tool_iqt; s.tool_ats = s.tool_iqt && sp->tool_ats; s.tool_addb = s.profile_main && sp->tool_addb;
    s.tool_alf = s.profile_main && sp->tool_alf;
    s.tool_eipd = s.profile_main && sp->tool_eipd;
    w->sp.tool_iqt = s.tool_iqt; w->sp.tool_ats = s.tool_ats; w->sp.tool_addb = s.tool_addb; w->sp.tool_alf = s.tool_alf; w->sp.tool_eipd = s.tool_eipd;
    w->sp.tool_dra = s.profile_main && sp->tool_dra; s.tool_dra = w->sp.tool_dra;
    w->sp.tool_htdf = s.profile_main && sp->tool_htdf; s.tool_htdf = w->sp.tool_htdf;
    w->sp.tool_admvp = s.profile_main && sp->tool_admvp; s.tool_admvp = w->sp.tool_admvp;
    w->sp.tool_amvr = s.tool_admvp && sp->tool_amvr; s.tool_amvr = w->sp.tool_amvr;
    w->sp.tool_hmvp = s.tool_admvp && sp->tool_hmvp; s.tool_hmvp = w->sp.tool_hmvp;
    w->sp.tool_affine = s.tool_admvp && sp->tool_affine; s.tool_affine = w->sp.tool_affine;
    w->st.enc_side = true;
    s.btt = s.profile_main && sp->btt;
    if (s.btt) {
        s.btt_raw[0] = std::min(std::max(sp->btt_log2_min_cb - 2, 0), 4); s.btt_raw[1] = std::min(std::max(sp->btt_diff_max_14, 0), 6);
        s.btt_raw[2] = std::min(std::max(sp->btt_diff_max_tt, 0), 6); s.btt_raw[3] = std::min(std::max(sp->btt_diff_min_tt, 0), 6);
        s.log2_min_cb = s.btt_raw[0] + 2;
        s.split_tbl[0][1] = 6; s.split_tbl[0][0] = s.log2_min_cb; s.split_tbl[1][1] = 6; s.split_tbl[1][0] = s.log2_min_cb + 1;
        s.split_tbl[2][1] = std::min(6 - s.btt_raw[1], 6); s.split_tbl[2][0] = s.log2_min_cb + 2;
        s.split_tbl[3][1] = std::min(6 - s.btt_raw[2], 6); s.split_tbl[3][0] = s.log2_min_cb + s.btt_raw[3] + 2;
    }
    s.tool_cm_init = s.profile_main && (sp->tool_cm_init || sp->tool_adcc); s.tool_adcc = s.tool_cm_init && sp->tool_adcc;      // tool_adcc is a sub-flag of tool_cm_init
    s.tool_rpl = s.profile_main && sp->tool_rpl; s.tool_pocs = s.profile_main && sp->tool_pocs; s.poc_lsb_bits = 8;
    if (s.tool_rpl && sp->rpl_in_sps && sp->log2_sub_gop_length == 0 && sp->max_num_ref_pics >= 2) {
        // low delay: list 0 of a picture with k references is { 1 .. k } - as candidates of the SPS (both lists), picked by index in the slice headers
        for (int l = 0; l < 2; l++) {
            s.n_rpl[l] = std::min(sp->max_num_ref_pics, 5);
            for (int k = 0; k < s.n_rpl[l]; k++) { s.rpls[l][k] = Rpl(); s.rpls[l][k].n = k + 1; for (int j = 0; j <= k; j++) s.rpls[l][k].ref[j] = j + 1; }
        }
    }
    w->sp.tool_mmvd = s.tool_admvp && sp->tool_mmvd; s.tool_mmvd = w->sp.tool_mmvd;
    w->sp.tool_dmvr = s.tool_admvp && sp->tool_dmvr; s.tool_dmvr = w->sp.tool_dmvr;      // with tool_hmvp / tool_mmvd the writer needs the reference samples too (xhost_writer_set_ref_luma)
    w->sp.ibc_log_max_size = (s.tool_eipd && sp->ibc_log_max_size >= 2 && sp->ibc_log_max_size <= 7) ? sp->ibc_log_max_size : 0;
    s.ibc = w->sp.ibc_log_max_size != 0; s.ibc_log_max = w->sp.ibc_log_max_size;
    {
        Pps &q = w->st.pps;
        const int w_ctu = (sp->width + 63) >> 6, h_ctu = (sp->height + 63) >> 6;
        q.tile_cols = std::max(sp->tile_cols, 1); q.tile_rows = std::max(sp->tile_rows, 1);
        if (!s.profile_main || q.tile_cols > std::min(w_ctu, XGPU_MAX_TILE_COLS) || q.tile_rows > std::min(h_ctu, XGPU_MAX_TILE_ROWS)) q.tile_cols = q.tile_rows = 1;
        q.tile_uniform = sp->tile_col_w[0] == 0;
        for (int i = 0; i < XGPU_MAX_TILE_COLS; i++) q.tile_col_w[i] = sp->tile_col_w[i];
        for (int i = 0; i < XGPU_MAX_TILE_ROWS; i++) q.tile_row_h[i] = sp->tile_row_h[i];
        const int n = q.tile_cols * q.tile_rows;
        q.across_tiles = n > 1 && sp->loop_filter_across_tiles; q.offset_bits = 24; q.id_bits = 1;
        while ((1 << q.id_bits) < n) q.id_bits++;
        w->sp.tile_cols = q.tile_cols; w->sp.tile_rows = q.tile_rows;
    }
    w->st.pps.cu_qp_delta = sp->cu_qp_delta;
    // Main: sps->dquant_flag with quantisation groups of 2^cu_qp_delta_area samples (6 = 8x8 ... 12 = 64x64); 0 = off (a delta per coded CU)
    s.dquant = (s.profile_main && sp->cu_qp_delta && sp->cu_qp_delta_area >= 6 && sp->cu_qp_delta_area <= 13) ? 1 : 0;
    w->st.pps.qp_delta_area = s.dquant ? sp->cu_qp_delta_area : 6;
    return w;
}
extern "C" void xhost_writer_close(xhost_writer *w) { delete w; }
extern "C" int xhost_writer_set_slice_alf(xhost_writer *w, const xhost_slice_alf *sa)
{
    if (!w || !sa) return XGPU_ERR_INVALID_ARGUMENT;
    w->next_alf = *sa;
    w->next_alf_ctb.clear();
    const size_t n_ctu = (size_t)((w->sp.width + 63) >> 6) * (size_t)((w->sp.height + 63) >> 6);
    if (sa->ctb_flag) w->next_alf_ctb.assign(sa->ctb_flag, sa->ctb_flag + n_ctu);
    w->next_alf.ctb_flag = nullptr;
    return XGPU_OK;
}
extern "C" int xhost_writer_set_slices(xhost_writer *w, int n, const xhost_slice_desc *d)
{
    if (!w || n < 0 || n > XGPU_MAX_TILE_COLS * XGPU_MAX_TILE_ROWS || (n > 0 && !d)) return XGPU_ERR_INVALID_ARGUMENT;
    w->slices.assign(d, d + n);
    return XGPU_OK;
}
extern "C" int xhost_writer_set_arbitrary_slices(xhost_writer *w, int on)
{
    if (!w || w->headers_done) return XGPU_ERR_INVALID_ARGUMENT;      // the PPS carries arbitrary_slice_present_flag
    w->arbitrary_slices = on != 0;
    return XGPU_OK;
}
extern "C" int xhost_writer_add_dra_aps(xhost_writer *w, const xhost_dra_aps *in)
{
    if (!w || !in || !w->sp.tool_dra || in->aps_id < 0 || in->aps_id > 31 || in->num_ranges < 1 || in->num_ranges > 32) return XGPU_ERR_INVALID_ARGUMENT;
    if (w->n_pics == 0 && w->out.empty()) { w->write_sps(); w->write_pps(); w->headers_done = true; }
    BitWriter bw;
    bw.put((uint32_t)in->aps_id, 5); bw.put(1, 3);
    bw.put(4, 4); bw.put(9, 4);
    bw.ue((uint32_t)(in->num_ranges - 1));
    bw.put1(0);                                          // dra_equal_ranges_flag
    bw.put((uint32_t)in->in_ranges[0], 10);
    for (int i = 0; i < in->num_ranges; i++) bw.put((uint32_t)(in->in_ranges[i + 1] - in->in_ranges[i]), 10);
    for (int i = 0; i < in->num_ranges; i++) bw.put((uint32_t)in->scale[i], 13);
    bw.put((uint32_t)in->cb_scale, 13); bw.put((uint32_t)in->cr_scale, 13);
    bw.ue((uint32_t)in->table_idx);
    bw.put1(0);                                          // aps_extension_flag
    bw.align_zero();
    write_nal(w->out, 26, 0, bw);
    return XGPU_OK;
}
extern "C" int xhost_writer_add_alf_aps(xhost_writer *w, const xhost_alf_aps *in)
{
    if (!w || !in || !w->st.sps.tool_alf || in->aps_id < 0 || in->aps_id > 31 || in->num_luma_filters < 1 || in->num_luma_filters > 25) return XGPU_ERR_INVALID_ARGUMENT;
    if (w->n_pics == 0 && w->out.empty()) { w->write_sps(); w->write_pps(); w->headers_done = true; }
    AlfAps a;
    a.valid = true;
    a.luma_present = in->luma_present != 0; a.chroma_present = in->chroma_present != 0; a.type7 = in->luma_type_7x7 != 0;
    a.num_filters = in->num_luma_filters; a.coef_delta_flag = in->coef_delta_flag != 0; a.pred_mode_flag = in->pred_mode_flag != 0;
    for (int c = 0; c < 25; c++) { a.delta_idx[c] = (uint8_t)(in->delta_idx[c] % a.num_filters); a.filter_coef_flag[c] = in->filter_coef_flag[c] != 0; }
    for (int f = 0; f < 25; f++) for (int i = 0; i < 12; i++) a.luma[f][i] = in->luma_coef[f][i];
    for (int i = 0; i < 6; i++) a.chroma[i] = in->chroma_coef[i];
    a.fixed_pattern = in->fixed_filter_pattern < 0 || in->fixed_filter_pattern > 2 ? 0 : in->fixed_filter_pattern;
    for (int c = 0; c < 25; c++) { a.fixed_usage[c] = (uint8_t)(a.fixed_pattern == 1 || (a.fixed_pattern == 2 && in->fixed_filter_usage[c])); a.fixed_idx[c] = (uint8_t)(in->fixed_filter_idx[c] & 15); }
    BitWriter bw;
    bw.put((uint32_t)in->aps_id, 5); bw.put(0, 3);
    AlfAps coded = a;
    if (!w->st.alf_aps_syntax<true>(nullptr, &bw, coded)) return XGPU_ERR_INVALID_ARGUMENT;
    bw.put1(0);                                          // aps_extension_flag
    bw.align_zero();
    write_nal(w->out, 26, 0, bw);
    // keep what a decoder will hold after parsing (filters without coefficients are zero, flags normalised)
    BitReader br;
    br.p = bw.buf.data(); br.size = bw.buf.size();
    br.get(8);
    AlfAps parsed;
    w->st.alf_aps_syntax<false>(&br, nullptr, parsed);
    parsed.valid = true;
    w->st.alf_aps[in->aps_id] = parsed;
    return XGPU_OK;
}
extern "C" int xhost_writer_set_ref_luma(xhost_writer *w, int poc, const int16_t *plane, int stride) { return w ? set_ref_luma(w->st, poc, plane, stride) : XGPU_ERR_INVALID_ARGUMENT; }
extern "C" int xhost_writer_add_md5_sei(xhost_writer *w, const uint8_t md5[3][16])
{
    if (!w || !md5 || w->n_pics == 0) return XGPU_ERR_INVALID_ARGUMENT;
    BitWriter bw;
    bw.put(0x10, 8); bw.put(16, 8);
    for (int c = 0; c < 3; c++) for (int i = 0; i < 16; i++) bw.put(md5[c][i], 8);
    bw.put(0x80, 8);                                      // rbsp trailing bits
    write_nal(w->out, NUT_SEI, w->last_tid, bw);
    return XGPU_OK;
}

extern "C" int xhost_writer_bytes(xhost_writer *w, const uint8_t **bytes, size_t *size)
{
    if (!w || !bytes || !size) return XGPU_ERR_INVALID_ARGUMENT;
    *bytes = w->out.data(); *size = w->out.size();
    return XGPU_OK;
}

namespace {
struct TreeWriter {
    xhost_writer *w;
    const xgpu_cu_batch *b;
    Enc *enc;
    std::vector<int> leaf;        // CU index by SCU position of its top-left corner, -1 elsewhere
    int bd_off;
    int error = 0;
    void node(int x, int y, int log2s, int qp_code = 0)
    {
        Stream &st = w->st;
        TileCoder &tcd = w->coder;
        const int s = 1 << log2s, ws = st.pic.w_scu;
        const int i = (x < st.sps.width && y < st.sps.height) ? leaf[(size_t)(y >> 2) * ws + (x >> 2)] : -1;
        const bool is_leaf = i >= 0 && b->log2w[i] == log2s;
        if (s >= 8) enc->bin(!is_leaf, tcd.models.split[0]);
        else if (!is_leaf) { error = 1; return; }
        qp_code = qp_group(st, tcd, is_leaf ? 0 : TileCoder::QUAD, log2s, log2s, qp_code);
        if (!is_leaf) {
            const int h = s >> 1;
            for (int q = 0; q < 4; q++) {
                const int nx = x + (q & 1) * h, ny = y + (q >> 1) * h;
                if (nx < st.sps.width && ny < st.sps.height) node(nx, ny, log2s - 1, qp_code);
            }
            return;
        }
        write_leaf(i, qp_code, 0);
    }
    // ---- sps_btt_flag: the split tree is found from the leaves - at every node the first allowed split whose cuts no CU crosses and below which the
    //      same search succeeds (plan), then coded (node_btt) ----
    std::vector<int> owner;                              // CU index of every SCU
    std::map<uint64_t, int> chosen;                      // node -> split mode
    static uint64_t node_key(int x, int y, int lw, int lh, int cons) { return ((uint64_t)x << 40) | ((uint64_t)y << 16) | ((uint64_t)lw << 8) | ((uint64_t)lh << 4) | (uint64_t)cons; }      // cons: 0 none, 1 inter only, 2 intra only (dual tree)
    std::map<uint64_t, int> chroma_cu;                   // node -> its chroma-only CU (xgpu_cu_batch.tree == 2): the local dual tree starts at that node
    int last_qp_code = 0;
    bool whole_cus(int x, int y, int wd, int ht, bool &any_non_inter) const      // the rectangle (clipped to the picture) is a union of whole CUs
    {
        const Stream &st = w->st;
        for (int yy = y; yy < std::min(y + ht, st.sps.height); yy += 4) for (int xx = x; xx < std::min(x + wd, st.sps.width); xx += 4) {
            const int i = owner[(size_t)(yy >> 2) * st.pic.w_scu + (xx >> 2)];
            if (i < 0 || b->x[i] < x || b->y[i] < y || b->x[i] + (1 << b->log2w[i]) > x + wd || b->y[i] + (1 << b->log2h[i]) > y + ht) return false;
            if (b->pred_mode[i] == XGPU_MODE_INTRA || b->pred_mode[i] == XGPU_MODE_IBC || st.sh.type == XHOST_SLICE_I) any_non_inter = true;
        }
        return true;
    }
    bool plan(int x, int y, int lw, int lh, bool only_inter, bool only_intra = false)
    {
        Stream &st = w->st;
        TileCoder &tcd = w->coder;
        const uint64_t key = node_key(x, y, lw, lh, only_inter ? 1 : only_intra ? 2 : 0);
        if (chosen.count(key)) return chosen[key] >= 0;
        const int W = st.sps.width, H = st.sps.height, wd = 1 << lw, ht = 1 << lh;
        const bool inside = x + wd <= W && y + ht <= H;
        if (inside) {
            const int i = leaf[(size_t)(y >> 2) * st.pic.w_scu + (x >> 2)];
            if (i >= 0 && b->log2w[i] == lw && b->log2h[i] == lh) {
                // a luma-only CU is a leaf of a dual tree and nothing else
                if (((b->tree ? b->tree[i] : 0) == 1) != only_intra) { chosen[key] = -1; return false; }
                chosen[key] = TileCoder::NO_SPLIT; return true;
            }
        }
        chosen[key] = -1;
        if (!(wd > (1 << st.sps.log2_min_cb) || ht > (1 << st.sps.log2_min_cb)) || (lw < 3 && lh < 3)) return false;
        int allow[6];
        tcd.split_allowed(allow, lw, lh, x, y, only_inter);
        for (int sp = TileCoder::BI_VER; sp <= TileCoder::TRI_HOR; sp++) {
            if (!allow[sp]) continue;
            if (!inside && sp != (allow[TileCoder::BI_VER] ? TileCoder::BI_VER : TileCoder::BI_HOR)) continue;      // the forced split
            int px[3], py[3], plw[3], plh[3];
            const int n = TileCoder::split_parts(sp, x, y, lw, lh, px, py, plw, plh);
            bool ok = true, non_inter = false;
            for (int k = 0; k < n && ok; k++) if (px[k] < W && py[k] < H) ok = whole_cus(px[k], py[k], 1 << plw[k], 1 << plh[k], non_inter);
            if (!ok) continue;
            bool child_oi = only_inter, child_intra = only_intra;
            if (st.sps.btt && st.sps.tool_admvp && !only_inter && !only_intra && !TileCoder::chroma_split_ok(sp, wd, ht)) {
                // the children need a mode constraint: a local dual tree when the batch has the node's chroma-only CU (inferred in I slices and for 4x4 children,
                // else signalled), "inter only" when every CU below is an inter CU
                if (chroma_cu.count(node_key(x, y, lw, lh, 0))) child_intra = true;
                else if (st.sh.type == XHOST_SLICE_I || TileCoder::small_child_is_4x4(sp, wd, ht) || non_inter) continue;
                else child_oi = true;
            } else if (!only_intra && chroma_cu.count(node_key(x, y, lw, lh, 0))) continue;      // a chroma-only CU where no dual tree can start
            for (int k = 0; k < n && ok; k++) if (px[k] < W && py[k] < H) ok = plan(px[k], py[k], plw[k], plh[k], child_oi, child_intra);
            if (ok) { chosen[key] = sp; return true; }
        }
        return false;
    }
    void node_btt(int x, int y, int lw, int lh, int qp_code, bool only_inter, bool only_intra = false)
    {
        Stream &st = w->st;
        TileCoder &tcd = w->coder;
        const int W = st.sps.width, H = st.sps.height, wd = 1 << lw, ht = 1 << lh, mn = 1 << st.sps.log2_min_cb;
        const int split = chosen[node_key(x, y, lw, lh, only_inter ? 1 : only_intra ? 2 : 0)];
        if ((wd > mn || ht > mn) && x + wd <= W && y + ht <= H) tcd.code_split(*enc, split, x, y, lw, lh, only_inter);
        qp_code = qp_group(st, tcd, split, lw, lh, qp_code);
        if (split == TileCoder::NO_SPLIT) { last_qp_code = qp_code; write_leaf(leaf[(size_t)(y >> 2) * st.pic.w_scu + (x >> 2)], qp_code, only_inter, only_intra ? 1 : 0); return; }
        const auto cc = chroma_cu.find(node_key(x, y, lw, lh, 0));
        const bool dual = !only_inter && !only_intra && cc != chroma_cu.end();
        const int mc = only_intra ? 0 : tcd.code_mode_cons(*enc, split, lw, lh, only_inter, dual ? 0 : 1);
        int px[3], py[3], plw[3], plh[3];
        const int n = TileCoder::split_parts(split, x, y, lw, lh, px, py, plw, plh);
        for (int k = 0; k < n; k++) if (px[k] < W && py[k] < H) node_btt(px[k], py[k], plw[k], plh[k], qp_code, mc == 1, only_intra || mc < 0);
        if (mc < 0) { if (cc == chroma_cu.end()) { error = 1; return; } write_leaf(cc->second, last_qp_code, 0, 2); }
    }
    void write_leaf(int i, int qp_code, int only_inter, int tree = 0)
    {
        Stream &st = w->st;
        TileCoder &tcd = w->coder;
        const int x = b->x[i], y = b->y[i], lw = b->log2w[i], lh = b->log2h[i], log2s = std::min(lw, lh);
        Cu cu;
        memset(&cu, 0, sizeof(cu));
        cu.x = x; cu.y = y; cu.log2w = lw; cu.log2h = lh; cu.qp_code = qp_code; cu.only_inter = only_inter; cu.tree = tree;
        cu.mode = b->pred_mode[i] == XGPU_MODE_INTRA ? MODE_INTRA : (b->pred_mode[i] == XGPU_MODE_SKIP ? MODE_SKIP : MODE_INTER);
        // (a luma-only IBC CU only with EIPD: a Baseline-mode chroma CU must find an intra mode at its centre, xevdm_eco.c:1771-1775)
        bool ibc = b->pred_mode[i] == XGPU_MODE_IBC && st.sps.ibc && std::max(lw, lh) <= st.sps.ibc_log_max && !only_inter && tree != 2 && !(tree == 1 && !st.sps.tool_eipd);
        if (ibc) {
            // the source block must lie in what THIS stream has coded before the CU: the writer's split tree may order the leaves differently from the
            // batch (several trees have the same leaves), and a block copied from samples that do not exist yet is whatever the decoder's buffer held
            const int bvx = b->mv[i * 4], bvy = b->mv[i * 4 + 1];
            const int x0 = x + (bvx & ~1), y0 = y + (bvy & ~1), x1 = x + bvx + (1 << lw) - 1, y1 = y + bvy + (1 << lh) - 1;
            if (x0 < 0 || y0 < 0 || x1 >= st.sps.width || y1 >= st.sps.height) ibc = false;
            for (int sy = y0 >> 2; ibc && sy <= y1 >> 2; sy++) for (int sx = x0 >> 2; sx <= x1 >> 2; sx++) if (!st.pic.cod[(size_t)sy * st.pic.w_scu + sx]) { ibc = false; break; }
        }
        if (st.sh.type == XHOST_SLICE_I || (st.sps.tool_admvp && lw == 2 && lh == 2) || tree) cu.mode = MODE_INTRA;
        if (ibc) cu.mode = MODE_IBC;
        cu.direct = (st.sh.type == XHOST_SLICE_B || (st.sps.tool_admvp && st.sh.type == XHOST_SLICE_P)) && b->pred_mode[i] == XGPU_MODE_DIR;
        for (int l = 0; l < 2; l++) {
            const int nref = (int)st.refp[l].size();
            cu.refi[l] = (b->refi[i * 2 + l] < 0 || nref == 0) ? -1 : std::min((int)b->refi[i * 2 + l], nref - 1);
            cu.mv[l][0] = b->mv[i * 4 + l * 2]; cu.mv[l][1] = b->mv[i * 4 + l * 2 + 1];
        }
        if (cu.mode == MODE_INTER && cu.refi[0] < 0 && cu.refi[1] < 0) cu.refi[0] = 0;
        if (st.sh.type == XHOST_SLICE_P) { cu.refi[1] = -1; if (cu.refi[0] < 0) cu.refi[0] = 0; }
        if (ibc) { cu.refi[0] = cu.refi[1] = -1; cu.mv[1][0] = cu.mv[1][1] = 0; }
        cu.mvp_idx[0] = (x >> 2) & 3; cu.mvp_idx[1] = (y >> 2) & 3;   // a SKIP CU: some predictor per list
        if (st.sps.tool_admvp) cu.mvp_idx[0] = cu.mvp_idx[1] = ((x >> 2) + 2 * (y >> 2)) % 6;
        // (not for CUs of up to 32 samples - 4x8 / 8x4 with sps_btt_flag: the reference's candidate list for them comes out with uninitialised entries in P slices,
        //  refi 85 / vector 13235 seen in xevdm_get_mmvd_mvp_list's output; decoder behaviour on them is undefined, so the streams stay away)
        if (st.sps.tool_mmvd && ((x >> 3) + (y >> 2)) % 3 == 0 && (1 << (lw + lh)) > 32) { cu.mmvd = 1; cu.mmvd_idx = ((x >> 2) * 37 + (y >> 2) * 101 + i) % 384; }      // a third of the skip / merge-mode CUs: any group, base, distance, direction      // ... or one of the six merge candidates (also of a merge-mode CU)
        if (st.sps.tool_affine && b->affine && b->affine[i] >= 2 && (cu.mode == MODE_SKIP || cu.mode == MODE_INTER) && !cu.mmvd) {
            // an affine CU: merge candidates of a skip / merge-mode CU (8x8 and larger), or the batch's control points coded against a predictor (16x16 and larger)
            // (not before the stream's first translational inter CU: the reference's sub-block affine prediction reads its interpolation taps through a
            //  process-global pointer that only the translational path switches from the Baseline table - zeros at the sixteenth-sample phases - to the
            //  Main one, src_main/xevdm_mc.c:1914-1924; an affine CU decoded before that predicts zeros.  A decoder state, not a stream property: avoided)
            const bool merge = cu.mode == MODE_SKIP || cu.direct;
            if (log2s >= (merge ? 3 : 4) && w->plain_inter_seen) {
                cu.affine = b->affine[i] - 1;
                cu.aff_idx[0] = merge ? ((x >> 3) + (y >> 3) * 3) % 5 : (x >> 4) & 1; cu.aff_idx[1] = (y >> 4) & 1;
                if (b->affine_mv) memcpy(cu.aff_mv, b->affine_mv + (size_t)i * 12, sizeof(cu.aff_mv));
            }
        }
        if ((cu.mode == MODE_SKIP || cu.mode == MODE_INTER) && !cu.affine) w->plain_inter_seen = true;
        cu.ipm = b->ipm ? b->ipm[i * 2] % (st.sps.tool_eipd ? 33 : 5) : 0;
        cu.ipm_c = (b->ipm && st.sps.tool_eipd) ? b->ipm[i * 2 + 1] % 5 : 0;
        cu.qp = std::min(std::max((int)b->qp[i * 3] - bd_off, 0), 51);
        for (int k = 0; k < 3; k++) cu.cbf[k] = (cu.mode == MODE_SKIP || (tree == 1 && k > 0) || (tree == 2 && k == 0)) ? 0 : (b->cbf[i] >> k) & 1;
        cu.ats = (st.sps.tool_ats && b->ats && cu.mode == MODE_INTRA) ? b->ats[i] & 7 : 0;
        cu.ats_inter = 0;
        if (st.sps.tool_ats && b->ats_inter && cu.mode == MODE_INTER && !cu.direct) {
            const int info = b->ats_inter[i], idx = info & 15, dim = (idx == 1 || idx == 3) ? 1 << lw : 1 << lh;      // the side the TU split cuts
            if (idx >= 1 && idx <= 4 && dim >= (idx >= 3 ? 16 : 8) && std::max(lw, lh) <= 6) cu.ats_inter = info & 0x1F;
        }
        const int tu_shift = (cu.ats_inter & 15) == 0 ? 0 : (((cu.ats_inter & 15) >= 3) ? 2 : 1);
        // coefficient blocks: a coded component needs at least one non-zero value to be representable
        std::vector<int16_t> blk[3];
        int16_t *coef[3];
        size_t off = b->coef_off[i];
        for (int k = 0; k < 3; k++) {
            const size_t n = (size_t)1 << (lw + lh - (k ? 2 : 0) - tu_shift);
            blk[k].assign(n, 0);
            if ((b->cbf[i] >> k) & 1) {
                blk[k].assign(b->coef + off, b->coef + off + n);
                off += n;
                bool nz = false;
                for (int16_t v : blk[k]) nz |= v != 0;
                if (!nz) cu.cbf[k] = 0;
            }
            coef[k] = blk[k].data();
        }
        if (st.sps.tool_admvp && cu.mode == MODE_INTER && cu.direct && !(cu.cbf[0] | cu.cbf[1] | cu.cbf[2])) { cu.mode = MODE_SKIP; cu.direct = 0; }      // merge mode without coefficients IS skip
        const bool cbf_all_path = (cu.mode == MODE_INTER || cu.mode == MODE_IBC) && tree == 0;          // eco_cbf's non-intra branch
        if (cbf_all_path && !(cu.cbf[0] | cu.cbf[1] | cu.cbf[2])) { /* all-zero flag path */ }
        else if (cbf_all_path && cu.cbf[1] + cu.cbf[2] == 0) cu.cbf[0] = 1;      // implied luma cbf needs a luma coefficient
        if (cbf_all_path && cu.cbf[0]) { bool nz = false; for (int16_t v : blk[0]) nz |= v != 0; if (!nz) blk[0][0] = 1; }
        if (cu.mmvd && (cu.mode == MODE_SKIP || cu.direct)) {
            // an index whose candidate has no reference in either list (P slices: prediction type "none" of a group, xevdm_util.c:499-503) would leave the
            // decoder without a prediction - its buffer keeps what the last CU left: not a stream to compare decoders with
            Cu t = cu;
            if (!(st.sh.mmvd_group && (1 << (lw + lh)) > 32)) t.mmvd_idx &= 127;
            tcd.mmvd_motion(t);
            if (t.refi[0] < 0 && t.refi[1] < 0) cu.mmvd = 0;
        }
        if (st.sps.tool_admvp && !cu.mmvd && !cu.affine && (cu.mode == MODE_SKIP || cu.direct)) {
            // ... and the same for a merge candidate (e.g. a temporal one that only has list-1 motion, in a P slice): take the next index that predicts from something
            const int n_cand = (1 << (lw + lh)) <= 32 ? 4 : 6;      // CUs of up to 32 samples have four candidates (MAX_NUM_MVP_SMALL_CU)
            cu.mvp_idx[0] = cu.mvp_idx[1] = cu.mvp_idx[0] % n_cand;
            for (int tries = 0; tries < n_cand; tries++) {
                Cu t = cu;
                tcd.merge_motion(t, cu.mvp_idx[0]);
                if (t.refi[0] >= 0 || t.refi[1] >= 0) break;
                cu.mvp_idx[0] = cu.mvp_idx[1] = (cu.mvp_idx[0] + 1) % n_cand;
            }
        }
        tcd.code_cu(*enc, cu, coef, true);
        tcd.commit(cu);
    }
};
}

// which splits the writer's stream allows at a node (for generators of CU batches): allow[0..4] = none, binary vertical / horizontal cut, ternary vertical / horizontal
extern "C" int xhost_writer_split_allowed(xhost_writer *w, int x, int y, int log2w, int log2h, int allow[5])
{
    if (!w || !allow || log2w < 2 || log2w > 6 || log2h < 2 || log2h > 6) return XGPU_ERR_INVALID_ARGUMENT;
    for (int i = 0; i < 5; i++) allow[i] = i == 0;
    if (!w->st.sps.btt) return XGPU_OK;
    const int wd = 1 << log2w, ht = 1 << log2h, mn = 1 << w->st.sps.log2_min_cb;
    if (!(wd > mn || ht > mn) || (log2w < 3 && log2h < 3)) { allow[0] = x + wd <= w->st.sps.width && y + ht <= w->st.sps.height; return XGPU_OK; }
    int a[6];
    w->coder.split_allowed(a, log2w, log2h, x, y, false);
    for (int i = 0; i < 5; i++) allow[i] = a[i];
    // splits whose children need a mode constraint (sps_btt_flag with tool_admvp, smallest child under 64 luma samples): 2 = either every CU below is an inter CU
    // (P / B pictures) or the node starts a local dual tree - luma-only intra / IBC CUs below, then the node's chroma-only CU; 3 = the dual tree only (4x4 children)
    if (w->st.sps.tool_admvp) for (int i = 1; i < 5; i++) if (allow[i] && !TileCoder::chroma_split_ok(i, wd, ht)) allow[i] = TileCoder::small_child_is_4x4(i, wd, ht) ? 3 : 2;
    return XGPU_OK;
}

extern "C" int xhost_writer_add_picture(xhost_writer *w, int idr, int slice_type, int slice_qp, int temporal_id, const xgpu_cu_batch *b)
{
    if (!w || !b || slice_qp < 0 || slice_qp > 51 || temporal_id < 0 || temporal_id > w->sp.log2_sub_gop_length) return XGPU_ERR_INVALID_ARGUMENT;
    Stream &st = w->st;
    if (w->n_pics == 0) { idr = 1; if (!w->headers_done) { w->write_sps(); w->write_pps(); w->headers_done = true; } }
    if (idr) { slice_type = XHOST_SLICE_I; temporal_id = 0; }
    if (slice_type < 0 || slice_type > 2) return XGPU_ERR_INVALID_ARGUMENT;
    st.sh.type = slice_type; st.sh.qp = slice_qp; st.sh.qp_u_offset = w->sp.qp_u_offset; st.sh.qp_v_offset = w->sp.qp_v_offset;
    st.sh.deblock = w->sp.deblock_on ? 1 : 0;
    st.derive_poc(idr != 0, temporal_id);
    if (slice_type == XHOST_SLICE_I) st.last_intra_poc = st.poc;
    st.build_ref_lists(idr != 0);
    if (slice_type != XHOST_SLICE_I && st.refp[0].empty()) return XGPU_ERR_INVALID_ARGUMENT;
    // tool_pocs / tool_rpl: the same pictures and lists, described to the decoder - poc_lsb, and RPLs whose leading entries are the lists above and whose
    // tail (list 0) names every other picture the sub-GOP scheme still keeps, so that the decoder's marking drops exactly what that scheme drops
    st.sh.poc_lsb = st.poc & ((1 << st.sps.poc_lsb_bits) - 1);
    for (int l = 0; l < 2; l++) {
        Rpl &r = st.sh.rpl[l];
        r = Rpl();
        for (const RefPic *q : st.refp[l]) r.ref[r.n++] = st.poc - q->poc;
        r.active = r.n;
    }
    for (const RefPic &q : st.dpb) {
        bool named = false;
        for (int l = 0; l < 2; l++) for (int j = 0; j < st.sh.rpl[l].n; j++) named |= st.sh.rpl[l].ref[j] == st.poc - q.poc;
        if (!named && st.sh.rpl[0].n < XGPU_MAX_REFS) st.sh.rpl[0].ref[st.sh.rpl[0].n++] = st.poc - q.poc;
    }
    if (slice_type == XHOST_SLICE_B && st.refp[1].empty()) return XGPU_ERR_INVALID_ARGUMENT;

    const int n_tiles = st.pps.tile_cols * st.pps.tile_rows;
    // the slices of this picture (xhost_writer_set_slices; default: one slice with every tile) and, per tile, the slice it belongs to
    std::vector<xhost_slice_desc> slices = w->slices;
    if (slices.empty()) { xhost_slice_desc d; d.first_tile = 0; d.last_tile = n_tiles - 1; d.slice_qp = -1; d.deblock_on = -1; slices.push_back(d); }
    std::vector<std::vector<int>> slice_tiles(slices.size());
    std::vector<int> tile_slice((size_t)n_tiles, -1);
    for (size_t k = 0; k < slices.size(); k++) {
        const int first = slices[k].first_tile, last = slices[k].last_tile, wt = st.pps.tile_cols;
        if (first < 0 || last < first || last >= n_tiles || first % wt > last % wt) return XGPU_ERR_INVALID_ARGUMENT;      // rectangles inside the grid (no wrap-around)
        for (int r = first / wt; r <= last / wt; r++) for (int c2 = first % wt; c2 <= last % wt; c2++) {
            if (tile_slice[(size_t)r * wt + c2] >= 0) return XGPU_ERR_INVALID_ARGUMENT;
            tile_slice[(size_t)r * wt + c2] = (int)k; slice_tiles[k].push_back(r * wt + c2);
        }
        if (slices[k].slice_qp > 51) return XGPU_ERR_INVALID_ARGUMENT;
    }
    for (int t = 0; t < n_tiles; t++) if (tile_slice[(size_t)t] < 0) return XGPU_ERR_INVALID_ARGUMENT;
    if (slices.size() > 1 && !st.sps.tool_pocs) return XGPU_ERR_INVALID_ARGUMENT;      // the reference decoder counts a picture per slice NAL without poc_lsb (see the parser)
    st.sh.mmvd_group = (st.sps.tool_mmvd && slice_type != XHOST_SLICE_I) ? (w->n_pics & 1) : 0;      // every other picture with the candidate groups
    const int w_ctu = (st.sps.width + 63) >> 6, h_ctu = (st.sps.height + 63) >> 6;
    st.alf_ctb_flag.assign((size_t)w_ctu * h_ctu, 1);
    if (st.sps.tool_alf) {
        st.sh.alf_on = w->next_alf.alf_on ? 1 : 0; st.sh.aps_id_y = w->next_alf.aps_id_y & 31; st.sh.aps_id_ch = w->next_alf.aps_id_ch & 31;
        st.sh.alf_chroma_idc = w->next_alf.chroma_idc & 3; st.sh.alf_ctb_map = w->next_alf.ctb_map ? 1 : 0;
        if (st.sh.alf_on && (!st.alf_aps[st.sh.aps_id_y].valid || !st.alf_aps[st.sh.aps_id_y].luma_present ||
                             (st.sh.alf_chroma_idc && (!st.alf_aps[st.sh.aps_id_ch].valid || !st.alf_aps[st.sh.aps_id_ch].chroma_present))))
            return XGPU_ERR_INVALID_ARGUMENT;
    } else st.sh.alf_on = 0;
    st.sh.tmvp_assigned = st.sh.col_list = st.sh.col_src_list = st.sh.col_ref = 0;
    st.sh.alpha_off = st.sps.tool_addb ? w->sp.deblock_alpha_offset : 0; st.sh.beta_off = st.sps.tool_addb ? w->sp.deblock_beta_offset : 0;
    // the slice header of slice k up to the entry points (xevdm_eco_sh, src_main/xevdm_eco.c:2510-2797)
    auto slice_header = [&](BitWriter &bw, size_t k, int qp, int deblock) {
        bw.ue(0);                                        // slice_pic_parameter_set_id
        if (n_tiles > 1) {
            // single_tile_in_slice_flag stays 0 also for a slice of one tile: the reference decoder does not reset last_tile_id for such a header and computes the
            // slice's rectangle from the PREVIOUS slice's value (set_tile_info, src_main/xevdm.c:2185-2210 after xevdm_eco_sh :2519-2538)
            bw.put1(0); bw.put((uint32_t)slices[k].first_tile, st.pps.id_bits);
            if (st.pps.arbitrary_slices && slice_tiles[k].size() > 1) {      // the same tiles as an ascending list (xevdm_eco.c:2540-2548)
                bw.put1(1); bw.ue((uint32_t)slice_tiles[k].size() - 2);
                for (size_t i = 1; i < slice_tiles[k].size(); i++) bw.ue((uint32_t)(slice_tiles[k][i] - slice_tiles[k][i - 1] - 1));
            } else {
                if (st.pps.arbitrary_slices) bw.put1(0);
                bw.put((uint32_t)slices[k].last_tile, st.pps.id_bits);
            }
        }
        bw.ue((uint32_t)slice_type);
        if (idr) bw.put1(0);                             // no_output_of_prior_pics_flag
        if (st.sps.tool_mmvd && slice_type != XHOST_SLICE_I) bw.put1(st.sh.mmvd_group);
        if (st.sps.tool_alf) {
            bw.put1(st.sh.alf_on);
            if (st.sh.alf_on) {
                bw.put((uint32_t)st.sh.aps_id_y, 5); bw.put1(st.sh.alf_ctb_map); bw.put((uint32_t)st.sh.alf_chroma_idc, 2);
                if (st.sh.alf_chroma_idc) bw.put((uint32_t)st.sh.aps_id_ch, 5);
            }
        }
        if (!idr) {                                      // xevdm_eco.c:2658-2733
            if (st.sps.tool_pocs) bw.put((uint32_t)st.sh.poc_lsb, st.sps.poc_lsb_bits);
            if (st.sps.tool_rpl)
                for (int l = 0; l < 2; l++) {
                    int hit = -1;                        // a candidate of the SPS with these entries (the index is only sent when there are at least two)
                    for (int i = 0; i < st.sps.n_rpl[l] && st.sps.n_rpl[l] > 1 && hit < 0; i++)
                        if (st.sps.rpls[l][i].n == st.sh.rpl[l].n && !memcmp(st.sps.rpls[l][i].ref, st.sh.rpl[l].ref, sizeof(int) * (size_t)st.sh.rpl[l].n)) hit = i;
                    if (st.sps.n_rpl[l] > 0) bw.put1(hit >= 0);
                    if (hit >= 0) bw.ue((uint32_t)hit); else write_rpl(bw, st.sh.rpl[l]);
                }
        }
        if (slice_type != XHOST_SLICE_I) {               // num_ref_idx_active_override_flag (+ the list sizes, which the decoder only uses with tool_rpl)
            bw.put1(st.sps.tool_rpl);
            if (st.sps.tool_rpl) { bw.ue((uint32_t)st.sh.rpl[0].active - 1); if (slice_type == XHOST_SLICE_B) bw.ue((uint32_t)st.sh.rpl[1].active - 1); }
        }
        if (slice_type != XHOST_SLICE_I && st.sps.tool_admvp) bw.put1(0);      // temporal_mvp_asigned_flag: the collocated picture is reference 0 of list 1 (P: list 0)
        bw.put1(deblock);
        if (deblock && st.sps.tool_addb) { bw.se(st.sh.alpha_off); bw.se(st.sh.beta_off); }
        bw.put((uint32_t)qp, 6);
        bw.se(st.sh.qp_u_offset); bw.se(st.sh.qp_v_offset);
    };

    st.pic.reset(st.sps.width, st.sps.height, st.sps.host_dmvr());
    if (!st.setup_tiles()) return XGPU_ERR_INVALID_ARGUMENT;
    TreeWriter tw;
    tw.w = w; tw.b = b; tw.bd_off = 6 * (st.sps.bd_l - 8);
    tw.leaf.assign((size_t)st.pic.w_scu * st.pic.h_scu, -1);
    if (st.sps.btt) tw.owner.assign((size_t)st.pic.w_scu * st.pic.h_scu, -1);
    for (int i = 0; i < b->n_cu; i++) {
        if ((!st.sps.btt && b->log2w[i] != b->log2h[i]) || b->log2w[i] < 2 || b->log2w[i] > 6 || b->log2h[i] < 2 || b->log2h[i] > 6 || b->x[i] + (1 << b->log2w[i]) > st.sps.width ||
            b->y[i] + (1 << b->log2h[i]) > st.sps.height || (b->x[i] & 3) || (b->y[i] & 3) || (!st.sps.btt && ((b->x[i] & ((1 << b->log2w[i]) - 1)) || (b->y[i] & ((1 << b->log2h[i]) - 1)))))
            return XGPU_ERR_INVALID_ARGUMENT;
        if (b->tree && b->tree[i] == 2) {      // the chroma block of a local dual tree: found again through its node
            if (!(st.sps.btt && st.sps.tool_admvp) || b->pred_mode[i] != XGPU_MODE_INTRA) return XGPU_ERR_INVALID_ARGUMENT;
            tw.chroma_cu[TreeWriter::node_key(b->x[i], b->y[i], b->log2w[i], b->log2h[i], 0)] = i;
            continue;
        }
        if (b->tree && b->tree[i] > 2) return XGPU_ERR_INVALID_ARGUMENT;
        tw.leaf[(size_t)(b->y[i] >> 2) * st.pic.w_scu + (b->x[i] >> 2)] = i;
        if (st.sps.btt)
            for (int r = 0; r < (1 << b->log2h[i]) >> 2; r++) for (int c2 = 0; c2 < (1 << b->log2w[i]) >> 2; c2++) tw.owner[(size_t)((b->y[i] >> 2) + r) * st.pic.w_scu + (b->x[i] >> 2) + c2] = i;
    }
    if (st.sps.btt)      // the split tree of every CTU must exist before anything is written
        for (int cy = 0; cy < h_ctu; cy++) for (int cx = 0; cx < w_ctu; cx++) if (!tw.plan(cx << 6, cy << 6, 6, 6, false)) return XGPU_ERR_INVALID_ARGUMENT;
    // every tile is its own arithmetic-coder run (contexts, QP predictor, motion history); the header carries the byte sizes of all but the last
    TileCoder &tcd = w->coder;
    std::vector<BitWriter> tile_bits((size_t)n_tiles);
    for (int t = 0; t < n_tiles; t++) {
        const int tc = t % st.grid.n_cols, tr = t / st.grid.n_cols;
        const int tile_qp = slices[(size_t)tile_slice[(size_t)t]].slice_qp >= 0 ? slices[(size_t)tile_slice[(size_t)t]].slice_qp : slice_qp;
        st.sh.qp = tile_qp;                              // (the coder of a CU reads the slice QP of ITS slice)
        if (st.sps.tool_cm_init) tcd.models.reset_cm(slice_type == XHOST_SLICE_B, tile_qp); else tcd.models.reset();
        tcd.qp_prev = tile_qp;
        Enc enc;
        enc.bw = &tile_bits[(size_t)t];
        enc.start();
        tw.enc = &enc;
        for (int cy = st.grid.row_bd[tr]; cy < st.grid.row_bd[tr + 1]; cy++) for (int cx = st.grid.col_bd[tc]; cx < st.grid.col_bd[tc + 1]; cx++) {
            if (cx == st.grid.col_bd[tc]) tcd.history_reset();
            if (st.sh.alf_on && st.sh.alf_ctb_map) {
                const int f = w->next_alf_ctb.empty() ? 1 : (w->next_alf_ctb[(size_t)cy * w_ctu + cx] != 0);
                enc.bin(f, tcd.models.alf_ctb[0]);
                st.alf_ctb_flag[(size_t)cy * w_ctu + cx] = (uint8_t)f;
            }
            if (st.sps.btt) tw.node_btt(cx << 6, cy << 6, 6, 6, 0, false); else tw.node(cx << 6, cy << 6, 6);
        }
        if (tw.error) return XGPU_ERR_INVALID_ARGUMENT;
        enc.tile_end();
        // the reference steps to a tile in 4-byte words from the word its reader stands in (xevdm.c:2665-2678): an entry offset shorter than that breaks it
        while (t != slice_tiles[(size_t)tile_slice[(size_t)t]].back() && tile_bits[(size_t)t].buf.size() < 8) tile_bits[(size_t)t].buf.push_back(0);
    }
    for (size_t k = 0; k < slices.size(); k++) {         // one NAL unit per slice: header, entry points of its tiles, the tiles
        const int qp = slices[k].slice_qp >= 0 ? slices[k].slice_qp : slice_qp, deblock = slices[k].deblock_on >= 0 ? (slices[k].deblock_on != 0) : (w->sp.deblock_on ? 1 : 0);
        st.sh.qp = qp; st.sh.deblock = deblock;          // what stays in st.sh is the LAST slice's header: the one the decoders run the in-loop filters with
        BitWriter bw;
        slice_header(bw, k, qp, deblock);
        const std::vector<int> &tl = slice_tiles[k];
        for (size_t i = 0; i + 1 < tl.size(); i++) {
            const size_t sz = tile_bits[(size_t)tl[i]].buf.size();
            if (st.pps.offset_bits < 32 && (sz - 1) >> st.pps.offset_bits) return XGPU_ERR_UNSUPPORTED;
            bw.put((uint32_t)sz - 1, st.pps.offset_bits);      // entry_point_offset_minus1
        }
        bw.align_zero();
        for (int t : tl) bw.buf.insert(bw.buf.end(), tile_bits[(size_t)t].buf.begin(), tile_bits[(size_t)t].buf.end());
        write_nal(w->out, idr ? NUT_IDR : NUT_NONIDR, temporal_id, bw);
    }
    w->last_tid = temporal_id;
    std::vector<int> released;
    st.store_picture(idr != 0, released);
    w->n_pics++;
    return XGPU_OK;
}
