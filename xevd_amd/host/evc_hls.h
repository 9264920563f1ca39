// evc_hls.h - the state above the CU level that parser and writer share: reference picture lists, context-model sets, SPS / PPS / slice header fields, ALF and
// DRA parameter sets, the SCU maps of the picture in work and of the reference pictures, the CU batch under construction, and `Stream`: POC derivation,
// reference list construction and marking, the tile grid, ALF coefficient reconstruction, the DPB.
#pragma once
#include <condition_variable>
#include <deque>
#include <mutex>
#include "evc_bits.h"
#include "alf_fixed_tables.h"
#include "dmvr_search.h"
#include "cm_init_tables.h"
#include "../csrc/affine_model.h"

namespace {
struct Rpl { int n = 0, active = 0; int ref[XGPU_MAX_REFS + 4] = { 0 }; };      // XEVD_RPL: POC differences cur - ref of the list's pictures (the first `active` are indexed by refi)
// ref_pic_list_struct (xevdm_eco_rlp, src_main/xevdm_eco.c:1820-1844): entry count, then per entry the POC difference to the previous entry with a sign
static inline bool read_rpl(BitReader &br, Rpl &r)
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
static inline void write_rpl(BitWriter &bw, const Rpl &r)
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
          suco_flag[14],                                                                             // sps_suco_flag: xevd_def.h:489
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
        CM(affine_mode); CM(affine_mrg); CM(affine_mvp_idx); CM(affine_mvd_flag); CM(ipm_mpm_flag); CM(ipm_mpm_idx); CM(ipm_chroma); CM(btt_split_flag); CM(btt_split_dir); CM(btt_split_type); CM(mode_cons); CM(suco_flag); CM(sig_coeff); CM(gt_ab); CM(last_x); CM(last_y);
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
static inline void make_zigzag(std::vector<uint16_t> &scan, int w, int h)
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
             int suco = 0, suco_raw[2] = { 0, 0 };     // sps_suco_flag: a split node may code its parts right to left; log2_diff_ctu_size_max_suco_cb_size, log2_diff_max_suco_min_suco_cb_size
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
static inline int alf_golomb_read(BitReader &br, int k, bool is_signed)
{
    int q = 0;
    while (!br.get1()) { if (++q > 24 || br.overrun) { br.overrun = true; return 0; } }
    int v = ((1 << q) - 1) << k;
    if (q + k > 0) v += (int)br.get(q + k);
    if (is_signed && v != 0) v = br.get1() ? v : -v;
    return v;
}
static inline void alf_golomb_write(BitWriter &bw, int v, int k, bool is_signed)
{
    const int a = v < 0 ? -v : v;
    int q = 0;
    while (a >= (((1 << (q + 1)) - 1) << k)) q++;
    for (int i = 0; i < q; i++) bw.put1(0);
    bw.put1(1);
    if (q + k > 0) bw.put((uint32_t)(a - (((1 << q) - 1) << k)), q + k);
    if (is_signed && a != 0) bw.put1(v > 0);
}
static inline int ilog2i(int v) { int l = 0; while ((v >> (l + 1)) > 0) l++; return l; }

struct RefPic {          // what a decoded picture leaves behind for later pictures (XEVD_PIC map_mv / list_poc, xevd_picman.c:213-221)
    int poc = 0, tid = 0;
    int list0_poc = 0;               // POC of reference 0 of ITS list 0 (pic->list_poc[0]); temporal direct mode scales by it
    std::vector<int16_t> mv;         // [f_scu][2][2]: the motion of every SCU, both lists (refp.map_mv; the Baseline temporal predictor reads list 0) and
    std::vector<int8_t> refi;        // [f_scu][2]: the reference indices (tool_admvp's temporal candidates read them; empty otherwise)
    int list_poc[16] = { 0 };        // pic->list_poc[]: POCs of ITS list-0 references (indexed by reference indices of EITHER list, xevdm_util.c:3760-3761)
    const int16_t *luma = nullptr;   // the decoded picture's luma samples on the host (sample (0, 0), >= 144 samples of replicated border), registered by the caller
    int luma_stride = 0;             //   when the front end refines vectors itself (xhost_parser_set_ref_luma; dmvr_search.h)
    long serial = 0;                 // count of reference pictures stored so far in this stream (POCs repeat across IDR periods): names the picture on the LumaBoard
};

// xhost_parser_set_ref_luma_wait: luma planes registered from ANOTHER thread while the parser is already inside a later picture.  The planes of the pictures
// handed out with needs_ref_luma are posted here; the parser thread moves the posted ones into the DPB at the start of every picture, and a CU whose
// refinement search needs a plane that has not been posted yet waits for it (TileCoder::commit) instead of the whole parser parking behind every
// reference picture.  Entries are added and removed on the parser thread only; the registering thread fills in plane / stride.
struct LumaBoard {
    struct Entry { long serial; int poc; const int16_t *plane; int stride; };
    std::mutex mu;
    std::condition_variable cv;
    std::deque<Entry> e;             // oldest first
    bool cancelled = false;
    void expect(long serial, int poc) { std::lock_guard<std::mutex> g(mu); e.push_back({ serial, poc, nullptr, 0 }); }
    int post(int poc, const int16_t *plane, int stride)      // the oldest picture with this POC that still waits for its plane
    {
        std::lock_guard<std::mutex> g(mu);
        for (Entry &x : e) if (x.poc == poc && !x.plane) { x.plane = plane; x.stride = stride; cv.notify_all(); return XGPU_OK; }
        return XGPU_OK;              // not kept as a reference, or already gone from the DPB: nothing will read it
    }
    bool wait(long serial, const int16_t **plane, int *stride)
    {
        std::unique_lock<std::mutex> g(mu);
        for (;;) {
            const Entry *hit = nullptr;
            for (const Entry &x : e) if (x.serial == serial) { hit = &x; break; }
            if (!hit || cancelled) return false;
            if (hit->plane) { *plane = hit->plane; *stride = hit->stride; return true; }
            cv.wait(g);
        }
    }
    void cancel() { std::lock_guard<std::mutex> g(mu); cancelled = true; cv.notify_all(); }
    void reset() { std::lock_guard<std::mutex> g(mu); e.clear(); cancelled = false; }
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
    // size(): the maps get the picture's geometry (contents undefined); clear_rows(): the state every picture starts from, for SCU rows [y0, y1) - the parser
    // clears row bands on its tile threads (at 8K the maps are 29 MB: a serial 10 ms in front of every picture otherwise); reset() = both, serially
    void size(int w, int h, bool refined_map = false)
    {
        w_scu = w >> 2; h_scu = h >> 2;
        const size_t f = (size_t)w_scu * h_scu;
        if (refined_map) mv_ref.resize(f * 4); else mv_ref.clear();
        cod.resize(f); intra.resize(f); ibc.resize(f); ipm.resize(f); mv.resize(f * 4); refi.resize(f * 2); tidx.clear(); aff.clear(); skip.clear(); cu_size.clear();
    }
    void clear_rows(int y0, int y1)
    {
        const size_t a = (size_t)y0 * w_scu, n = (size_t)(y1 - y0) * w_scu;
        if (!mv_ref.empty()) memset(mv_ref.data() + a * 4, 0, n * 4 * sizeof(int16_t));
        memset(cod.data() + a, 0, n); memset(intra.data() + a, 0, n); memset(ibc.data() + a, 0, n); memset(ipm.data() + a, 0, n);
        memset(mv.data() + a * 4, 0, n * 4 * sizeof(int16_t)); memset(refi.data() + a * 2, -1, n * 2);
    }
    void reset(int w, int h, bool refined_map = false) { size(w, h, refined_map); clear_rows(0, h_scu); }
};

// the coefficient arena of a batch grows without being zeroed (a resize of a plain vector value-initialises: 3 bytes per SAMPLE of every CU - 100 MB per 8K picture -
// when the space of a CU's three blocks was taken and given back CU by CU); a coded block is cleared where it is decoded (TileCoder::code_cu)
template <class T> struct NoInitAlloc : std::allocator<T> {
    template <class U> struct rebind { typedef NoInitAlloc<U> other; };
    template <class U> void construct(U *p) noexcept { ::new ((void *)p) U; }
    template <class U, class... A> void construct(U *p, A &&... a) { ::new ((void *)p) U(std::forward<A>(a)...); }
};
typedef std::vector<int16_t, NoInitAlloc<int16_t>> CoefVec;
struct Batch {           // the xgpu_cu_batch under construction
    std::vector<uint16_t> x, y;
    std::vector<uint8_t> log2w, log2h, pred_mode, qp, cbf, ipm, ats, ats_inter, dmvr, affine, tree;
    bool has_tree = false;
    std::vector<int8_t> refi;
    std::vector<int16_t> mv, affine_mv;
    CoefVec coef;
    std::vector<uint32_t> coef_off, ctu_start;
    void clear() { x.clear(); y.clear(); log2w.clear(); log2h.clear(); pred_mode.clear(); qp.clear(); cbf.clear(); ipm.clear(); ats.clear(); ats_inter.clear(); dmvr.clear(); affine.clear(); tree.clear(); has_tree = false; affine_mv.clear(); refi.clear(); mv.clear(); coef.clear(); coef_off.clear(); ctu_start.clear(); }
};

// ---- DRA parameter sets (APS type 1, SIG_PARAM_DRA) and the inverse-mapping tables the output stage applies (src_main/xevdm_dra.c) ----
struct DraAps { bool valid = false; int num_ranges = 0, in_ranges[33] = { 0 }, scale[32] = { 0 }, cb_scale = 0, cr_scale = 0, table_idx = 0; };
// approximations of log / exp at 9 fractional bits used by the chroma scale correction (constants of the specification, src_main/xevdm_tbl.c:410-421)
static const int k_dra_log_tbl[55] = { 0, 1, 1, 1, 1, 1, 2, 2, 3, 4, 4, 6, 7, 9, 11, 14, 18, 23, 29, 36, 45, 57, 72, 91, 114, 144, 181, 228, 287, 362, 456, 575, 724, 912, 1149,
                                       1448, 1825, 2299, 2896, 3649, 4598, 5793, 7298, 9195, 11585, 14596, 18390, 23170, 29193, 36781, 46341, 58386, 73562, 92682, 116772 };
static const int k_dra_exp_tbl[25] = { 128, 144, 161, 181, 203, 228, 256, 287, 322, 362, 406, 456, 512, 574, 645, 724, 812, 912, 1024, 1149, 1290, 1448, 1625, 1825, 2048 };
static inline int dra_range_idx(int sample, const int *ranges, int n)           // xevd_get_dra_range_idx_gen: first i with sample < ranges[i + 1], else n - 1
{
    for (int i = 0; i < n - 1; i++) if (sample < ranges[i + 1]) return i;
    return n - 1;
}
// luts: [3][1024] = luma_inv_scale_lut, int_chroma_inv_scale_lut[Cb], [Cr] of DRA_CONTROL after xevd_init_dra (xevdm_dra.c:39-270).
// cq[c] + off = the sequence's chroma QP mapping (xevd_qp_chroma_dynamic[c]), indexable from -off.
static inline void dra_build_luts(const DraAps &a, int bd, const int8_t *cq_u, const int8_t *cq_v, int off, int32_t *luts)
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
            // refp[] holds pointers into dpb: the lists an I slice keeps from an earlier P / B slice of the picture are re-resolved by POC behind the marking
            // (an erase shifts the elements behind it), and an I slice whose RPLs release one of those pictures is refused
            std::vector<int> kept[2];
            if (keep_lists) for (int l = 0; l < 2; l++) for (const RefPic *r : refp[l]) kept[l].push_back(r->poc);
            if (!idr)
                for (size_t i = 0; i < dpb.size();) {
                    bool named = false;
                    for (int l = 0; l < 2 && !named; l++) for (int j = 0; j < sh.rpl[l].n && !named; j++) named = dpb[i].poc == poc - sh.rpl[l].ref[j];
                    if (named) i++; else { rpl_released.push_back(dpb[i].poc); drop_ref(i); }
                }
            if (keep_lists)
                for (int l = 0; l < 2; l++) {
                    refp[l].clear();
                    for (int q : kept[l]) {
                        const RefPic *hit = nullptr;
                        for (const RefPic &r : dpb) if (r.poc == q) { hit = &r; break; }
                        if (!hit) return false;
                        refp[l].push_back(hit);
                    }
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
        if (idr) { for (const RefPic &r : dpb) released.push_back(r.poc); while (!dpb.empty()) drop_ref(dpb.size() - 1); }
        else if (tid == 0 && (!sps.tool_rpl || enc_side)) {            // sliding-window marking only without RPLs (xevdm_picman_put_pic, xevdm_picman.c:595-606)
            const int gap = 1 << sps.log2_ref_gap;
            for (size_t i = 0; i < dpb.size();) {
                if (dpb[i].tid > 0 || (i > 0 && gap > 0 && dpb[i].poc % gap != 0)) { released.push_back(dpb[i].poc); drop_ref(i); }
                else i++;
            }
            while (dpb.size() >= 5) { released.push_back(dpb[0].poc); drop_ref(0); }      // XEVD_MAX_NUM_ACTIVE_REF_FRAME
        }
        if (!is_ref_picture()) return;
        // bound on a damaged stream that keeps sending tid > 0 reference pictures without a tid-0 picture between them
        while (dpb.size() >= 32) { released.push_back(dpb[0].poc); drop_ref(0); }
        RefPic r;
        r.poc = poc; r.tid = tid; r.list0_poc = stale_list0_poc;
        r.serial = ++pic_serial;
        // The picture's motion field MOVES into the DPB entry (at 8K it is 16.6 MB + 4 MB of reference indices: copying them, and extracting a list-0 copy, was a
        // serial 18 ms behind every reference picture); the maps of the next picture take the vectors of an entry that left the DPB (pool) - no allocation, no
        // page faults in the steady state.  ctx->map_mv: with host-side DMVR the refined vectors
        std::vector<int16_t> &kept = pic.mv_ref.empty() ? pic.mv : pic.mv_ref;
        r.mv = std::move(kept);
        kept.clear();
        if (!mv_pool.empty()) { kept = std::move(mv_pool.back()); mv_pool.pop_back(); }
        if (sps.tool_admvp) {
            r.refi = std::move(pic.refi);
            pic.refi.clear();
            if (!refi_pool.empty()) { pic.refi = std::move(refi_pool.back()); refi_pool.pop_back(); }
            memcpy(r.list_poc, stale_list_poc, sizeof(r.list_poc));
        }
        dpb.push_back(std::move(r));
    }
    long pic_serial = 0;
    // the planes posted on the board so far go to their DPB entries (parser thread, between pictures); posted entries of pictures that left the DPB are dropped
    void take_posted_luma(LumaBoard &b)
    {
        std::lock_guard<std::mutex> g(b.mu);
        for (size_t i = 0; i < b.e.size();) {
            RefPic *r = nullptr;
            for (RefPic &d : dpb) if (d.serial == b.e[i].serial) { r = &d; break; }
            if (!b.e[i].plane) { i++; continue; }        // (also when the picture has left the DPB: its entry waits for its post, so that the post cannot land on a later picture with the same POC)
            if (r) { r->luma = b.e[i].plane; r->luma_stride = b.e[i].stride; }
            b.e.erase(b.e.begin() + (long)i);
        }
    }
    // a picture leaves the DPB: its motion vectors go back to the pool the picture maps draw from
    std::vector<std::vector<int16_t>> mv_pool;
    std::vector<std::vector<int8_t>> refi_pool;
    void drop_ref(size_t i)
    {
        if (!dpb[i].mv.empty() && mv_pool.size() < 4) mv_pool.push_back(std::move(dpb[i].mv));
        if (!dpb[i].refi.empty() && refi_pool.size() < 4) refi_pool.push_back(std::move(dpb[i].refi));
        dpb.erase(dpb.begin() + (long)i);
    }

};


static inline int set_ref_luma(Stream &st, int poc, const int16_t *plane, int stride)
{
    if (!plane || stride <= 0) return XGPU_ERR_INVALID_ARGUMENT;
    for (RefPic &r : st.dpb) if (r.poc == poc) { r.luma = plane; r.luma_stride = stride; return XGPU_OK; }
    return XGPU_OK;                                      // not kept as a reference: nothing will read it
}
}   // namespace
