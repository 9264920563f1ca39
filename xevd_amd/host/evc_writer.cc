// evc_writer.cc - EVC stream writer of the host front end (xhost_writer_*, include/xevd_host.h): parameter sets, APS, slice headers (one or several slices per
// picture) and the CU syntax of caller-supplied CU batches, through the SAME syntax templates the parser reads with (evc_cu.h) - the two cannot drift apart.
#include "evc_cu.h"

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
            bw.put1(st.sps.suco);                        // sps_suco_flag + the two size limits
            if (st.sps.suco) { bw.ue((uint32_t)st.sps.suco_raw[0]); bw.ue((uint32_t)st.sps.suco_raw[1]); }
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
    s.suco = s.profile_main && sp->suco;
    s.suco_raw[0] = s.suco ? std::min(std::max(sp->suco_diff_max, 0), 6) : 0; s.suco_raw[1] = s.suco ? std::min(std::max(sp->suco_diff_min, 0), 6) : 0;
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
    // sps_suco_flag: the writer's choice for a node that may choose - a hash of its position, so about half of them run right to left
    static int suco_want(int x, int y, int lw, int lh) { return (((x >> 3) * 5 + (y >> 3) * 3 + lw + 2 * lh) >> 1) & 1; }
    void node(int x, int y, int log2s, int qp_code = 0, int suco = 0)
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
            suco = tcd.code_suco(*enc, suco_want(x, y, log2s, log2s), TileCoder::QUAD, log2s, log2s, !(x + s <= st.sps.width && y + s <= st.sps.height), suco);
            int order[4];
            TileCoder::part_order(TileCoder::QUAD, suco, 4, order);
            for (int k = 0; k < 4; k++) {
                const int q = order[k], nx = x + (q & 1) * h, ny = y + (q >> 1) * h;
                if (nx < st.sps.width && ny < st.sps.height) node(nx, ny, log2s - 1, qp_code, suco);
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
    void node_btt(int x, int y, int lw, int lh, int qp_code, bool only_inter, bool only_intra = false, int suco = 0)
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
        suco = tcd.code_suco(*enc, suco_want(x, y, lw, lh), split, lw, lh, !(x + wd <= W && y + ht <= H), suco);
        const int mc = only_intra ? 0 : tcd.code_mode_cons(*enc, split, lw, lh, only_inter, dual ? 0 : 1);
        int px[3], py[3], plw[3], plh[3], order[4];
        const int n = TileCoder::split_parts(split, x, y, lw, lh, px, py, plw, plh);
        TileCoder::part_order(split, suco, n, order);
        for (int j = 0; j < n; j++) { const int k = order[j]; if (px[k] < W && py[k] < H) node_btt(px[k], py[k], plw[k], plh[k], qp_code, mc == 1, only_intra || mc < 0, suco); }
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

