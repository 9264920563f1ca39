/*
 * ref_binding.c - INTEGRATION.md section 4 as code that compiles: the reference decoder (mpeg5/xevd v0.7.0, Main-profile library) with the
 * MI355X backend of include/xevd_hip.h installed behind its coarse function-table slots.  TEST INFRASTRUCTURE (oracle/): it shows that the C
 * ABI is the drop-in boundary it claims to be - the reference's OWN front end (NAL / parameter sets / slice header / SBAC / CU syntax of every
 * Main tool / motion derivation / DPB / bumping / xevd_pull) runs unchanged and feeds the HIP backend, which replaces everything
 * xevd_dec_nalu does to a picture after entropy decoding.
 *
 * What is installed (on the XEVD_CTX a plain xevd_create() returned; nothing of the reference is edited):
 *   ctx->fn_dec_slice      <- hip_dec_slice      src_main/xevdm.c:2608-2718 (xevdm_dec_slice): the reference's xevd_tile_eco (:2363-2461) still
 *                                                parses the tile into XEVD_CU_DATA; the reconstruction half (xevd_tile_mt -> xevd_ctu_row_rec_mt ->
 *                                                xevd_recon_tree -> xevd_recon_unit, :2463-2606, :1854-1933, :1230-1405) is replaced by a walk that
 *                                                runs the reference's cu_init + motion derivation + xevdm_set_dec_info per CU (the bookkeeping later
 *                                                CUs and pictures need) and, instead of predicting and reconstructing, appends the CU to an
 *                                                xgpu_cu_batch; then xgpu_frame_begin + xgpu_batch_create + xgpu_batch_recon
 *   ctx->fn_deblock        <- hip_deblock        :2048-2103 (xevdm_deblock): xgpu_deblock, once per picture
 *   mctx->fn_alf           <- hip_alf            :2105-2111 (xevd_alf): the reference's own coefficient reconstruction (alf_load_paramline_from_aps_
 *                                                buffer2 + alf_recon_coef, src_main/xevdm_alf.c:682-794), then xgpu_alf
 *   ctx->fn_picbuf_expand  <- hip_picbuf_expand  src_base/xevd_util.c:365-427: xgpu_pad + xgpu_frame_end, then the picture's active area is copied
 *                                                into the reference's XEVD_PIC (what xevd_pull, the MD5 check and the DRA post-filter read)
 *   ctx->pf                <- the binding's state (the "platform specific data" slot, src_base/xevd_def.h:1452-1470)
 * Device pictures are keyed by the reference's XEVD_PIC pointers (its picture manager recycles a fixed pool).
 *
 * How the statics are reached: this translation unit #includes the reference's xevdm.c and xevdm_alf.c where they lie (it replaces those two
 * objects in the link, oracle/Makefile.ref: libxevd_ref_hip.so) - the way ref_harness.c reaches static helpers; no reference source is copied.
 * The recursion over the split tree below is OUR walk over the reference's exported helpers (xevd_get_split_mode,
 * xevd_split_get_part_structure, xevdm_get_suco_flag, xevdm_split_get_suco_order, xevd_derive_mode_cons).
 * Limits (asserted): 4:2:0.  Several slices per picture are gathered into one batch; with tool_dmvr next to tool_hmvp / tool_mmvd - later CUs of the picture derive
 * their candidates from refined vectors while the picture is still being parsed - the refinement SEARCH runs on the host (xhost_dmvr_search of libxevd_host.so, on luma
 * planes downloaded from the backend's picture slots), the backend repeats it for the prediction.
 */
#include "xevdm.c"
#include "xevdm_alf.c"
#include "../include/xevd_hip.h"
#include "../include/xevd_host.h"

typedef struct { void **p; size_t elem, n, cap; } rb_vec;          /* growable array of a batch field */
static void *rb_push(rb_vec *v, void **base, size_t elem, size_t count)
{
    if (v->n + count > v->cap) {
        v->cap = (v->n + count) * 2 + 1024;
        *base = realloc(*base, v->cap * elem);
    }
    v->n += count;
    return (char *)*base + (v->n - count) * elem;
}

typedef struct {
    xgpu_ctx *g;
    struct { const XEVD_PIC *pic; int slot; } slots[64];
    int n_slots;
    int deblocked;
    /* the batch of the picture being decoded */
    int n_cu, cap_cu;
    uint16_t *x, *y, *cbf_sub;
    uint8_t *log2w, *log2h, *pred_mode, *qp, *cbf, *ipm, *ats, *ats_inter, *affine, *dmvr, *tree;
    int8_t *refi;
    int16_t *mv, *affine_mv;
    uint32_t *coef_off, *ctu_start;
    void *coef; rb_vec coef_v;
    int any_affine, any_dmvr, any_ats, any_ats_inter, any_tree;
    int failed;
    /* sps->tool_dmvr with tool_hmvp / tool_mmvd: the refined vectors steer this parser's own candidate lists CU by CU, so the refinement SEARCH runs here, on host
       copies of the two references' luma planes (xhost_dmvr_search; the backend repeats it for the prediction) - fetched once per picture and reference */
    struct { int slot; int16_t *y; } luma[64];
    int n_luma;
    int n_ctu, pic_inter;         /* CTUs of the picture in the batch so far (several slices: one fn_dec_slice call each); a P / B slice among them */
    xgpu_tile_grid grid;
} rb_state;

static rb_state *rb_of(XEVD_CTX *ctx) { return (rb_state *)ctx->pf; }

static int rb_slot(rb_state *s, const XEVD_PIC *pic)
{
    int i;
    for (i = 0; i < s->n_slots; i++) if (s->slots[i].pic == pic) return s->slots[i].slot;
    if (s->n_slots == 64) return -1;
    s->slots[s->n_slots].pic = pic;
    s->slots[s->n_slots].slot = xgpu_pic_alloc(s->g);
    return s->slots[s->n_slots++].slot;
}

/* the padded luma plane of the picture in `slot` on the host (sample (0, 0); stride = width + 2 * 144) */
static const int16_t *rb_luma(rb_state *s, XEVD_CTX *ctx, int slot, int *stride)
{
    const int pad = 144, w = ctx->w + 2 * pad, h = ctx->h + 2 * pad;
    int i;
    *stride = w;
    for (i = 0; i < s->n_luma; i++) if (s->luma[i].slot == slot) return s->luma[i].y + pad * w + pad;
    if (s->n_luma == 64) return NULL;
    s->luma[s->n_luma].slot = slot;
    s->luma[s->n_luma].y = (int16_t *)malloc(sizeof(int16_t) * (size_t)w * h);
    if (xgpu_pic_download_padded(s->g, slot, s->luma[s->n_luma].y, NULL, NULL) < 0) { free(s->luma[s->n_luma].y); return NULL; }
    return s->luma[s->n_luma++].y + pad * w + pad;
}
static void rb_luma_drop(rb_state *s)
{
    int i;
    for (i = 0; i < s->n_luma; i++) free(s->luma[i].y);
    s->n_luma = 0;
}

static void rb_reserve(rb_state *s)
{
    if (s->n_cu < s->cap_cu) return;
    s->cap_cu = s->cap_cu ? s->cap_cu * 2 : 4096;
#define G(f, k) s->f = realloc(s->f, sizeof(*s->f) * (size_t)s->cap_cu * (k))
    G(x, 1); G(y, 1); G(cbf_sub, 1); G(log2w, 1); G(log2h, 1); G(pred_mode, 1); G(qp, 3); G(cbf, 1); G(ipm, 2); G(ats, 1); G(ats_inter, 1);
    G(affine, 1); G(dmvr, 1); G(tree, 1); G(refi, 2); G(mv, 4); G(affine_mv, 12); G(coef_off, 1);
#undef G
}

/* one leaf CU: the reference's per-CU bookkeeping (what xevd_recon_unit does around the pixel work, xevdm.c:1230-1405), the CU into the batch */
static void hip_recon_unit(XEVD_CTX *ctx, XEVD_CORE *core, int x, int y, int log2_cuw, int log2_cuh, int cup, TREE_CONS_NEW tree_cons)
{
    rb_state *s = rb_of(ctx);
    XEVDM_CORE *mcore = (XEVDM_CORE *)core;
    XEVD_CU_DATA *cu_data = &ctx->map_cu_data[core->lcu_num];
    const int cuw = 1 << log2_cuw, cuh = 1 << log2_cuh;
    int i, c, sb, mode;
    s16 mv_unrefined[REFP_NUM][MV_D];
    int refined_cu = 0;
    mcore->tree_cons = (TREE_CONS) { FALSE, tree_cons.tree_type, tree_cons.mode_cons };
    core->log2_cuw = log2_cuw; core->log2_cuh = log2_cuh;
    core->x_scu = PEL2SCU(x); core->y_scu = PEL2SCU(y);
    core->scup = core->x_scu + core->y_scu * ctx->w_scu;
    cu_init(ctx, core, x, y, cuw, cuh);
    core->avail_lr = xevd_check_nev_avail(core->x_scu, core->y_scu, cuw, cuh, ctx->w_scu, ctx->h_scu, ctx->map_scu, ctx->map_tidx);
    xevdm_get_ctx_some_flags(core->x_scu, core->y_scu, cuw, cuh, ctx->w_scu, ctx->map_scu, ctx->cod_eco, ctx->map_cu_mode, core->ctx_flags, ctx->sh.slice_type,
                             ctx->sps->tool_cm_init, ctx->sps->ibc_flag, ctx->sps->ibc_log_max_size, ctx->map_tidx, 0);
    if (core->pred_mode != MODE_SKIP) coef_rect_to_series(ctx, cu_data->coef, x, y, cuw, cuh, core->coef, core);

    /* motion: xevdm.c:1263-1345 without the prediction calls */
    mcore->dmvr_enable = 0;
    if (core->pred_mode == MODE_IBC) {
        core->avail_cu = xevdm_get_avail_ibc(core->x_scu, core->y_scu, ctx->w_scu, ctx->h_scu, core->scup, cuw, cuh, ctx->map_scu, ctx->map_tidx);
    } else if (core->pred_mode != MODE_INTRA) {
        core->avail_cu = xevdm_get_avail_inter(core->x_scu, core->y_scu, ctx->w_scu, ctx->h_scu, core->scup, cuw, cuh, ctx->map_scu, ctx->map_tidx);
        if (ctx->sps->tool_dmvr) {
            if (core->pred_mode == MODE_SKIP && !mcore->mmvd_flag) mcore->dmvr_enable = 1;
            if (core->inter_dir == PRED_DIR) mcore->dmvr_enable = 1;
            if (mcore->affine_flag) mcore->dmvr_enable = 0;
        }
        if (mcore->affine_flag) xevd_get_affine_motion(ctx, core);
        else if (core->pred_mode == MODE_SKIP) xevd_get_skip_motion(ctx, core);
        else if (core->inter_dir == PRED_DIR) {
            if (ctx->sps->tool_admvp == 0) {
                xevdm_get_mv_dir(ctx->refp[0], ctx->poc.poc_val, core->scup + ((1 << (core->log2_cuw - MIN_CU_LOG2)) - 1) + ((1 << (core->log2_cuh - MIN_CU_LOG2)) - 1) * ctx->w_scu,
                                 core->scup, ctx->w_scu, ctx->h_scu, core->mv, ctx->sps->tool_admvp);
                core->refi[REFP_0] = 0; core->refi[REFP_1] = 0;
            } else if (core->mvr_idx == 0) xevd_get_direct_motion(ctx, core);
        } else if (core->inter_dir == PRED_DIR_MMVD) xevdm_get_mmvd_motion(ctx, core);
        else xevd_get_inter_motion(ctx, core);
        /* what xevdm_mc decides before refining (src_main/xevdm_mc.c:1895-1911): the CU is flagged, its SCUs keep the unrefined vectors for the
           neighbours and the deblocking filter (map_unrefined_mv); map_mv receives the refined ones after the batch has run (hip_dec_slice) */
        if (mcore->dmvr_enable && ctx->sps->tool_dmvr && !mcore->affine_flag && REFI_IS_VALID(core->refi[0]) && REFI_IS_VALID(core->refi[1]) && cuw >= 8 && cuh >= 8) {
            const int d0 = (int)ctx->poc.poc_val - (int)ctx->refp[core->refi[0]][REFP_0].poc, d1 = (int)ctx->poc.poc_val - (int)ctx->refp[core->refi[1]][REFP_1].poc;
            if (d0 * d1 < 0 && abs(d0) == abs(d1)) {
                int k;
                mcore->dmvr_flag = 1;
                for (k = 0; k < (cuw >> MIN_CU_LOG2) * (cuh >> MIN_CU_LOG2); k++) memcpy(mcore->dmvr_mv[k], core->mv, sizeof(s16) * 4);
                if (ctx->sps->tool_hmvp || ctx->sps->tool_mmvd) {
                    /* ... unless later CUs of this picture derive their candidates from the refined vectors (history buffer, the MMVD base list): then the
                       search runs now, and mcore->dmvr_mv is what processDMVR would have left (src_main/xevdm_mc.c:1783-1797) */
                    int16_t sub[64 * 4], in[4];
                    const int dx = cuw < 16 ? cuw : 16, dy = cuh < 16 ? cuh : 16;
                    int st0 = 0, st1 = 0, sx, sy, u, v, n = 0;
                    const int16_t *r0 = rb_luma(s, ctx, rb_slot(s, ctx->refp[core->refi[0]][REFP_0].pic), &st0);
                    const int16_t *r1 = rb_luma(s, ctx, rb_slot(s, ctx->refp[core->refi[1]][REFP_1].pic), &st1);
                    memcpy(in, core->mv, sizeof(in));
                    if (!r0 || !r1 || xhost_dmvr_search(ctx->w, ctx->h, ctx->sps->bit_depth_luma_minus8 + 8, x, y, cuw, cuh, in, r0, st0, r1, st1, sub) < 0) s->failed = 1;
                    else
                        for (sy = 0; sy < cuh; sy += dy) for (sx = 0; sx < cuw; sx += dx, n++)
                            for (v = 0; v < dy >> MIN_CU_LOG2; v++) for (u = 0; u < dx >> MIN_CU_LOG2; u++)
                                memcpy(mcore->dmvr_mv[((sy >> MIN_CU_LOG2) + v) * (cuw >> MIN_CU_LOG2) + (sx >> MIN_CU_LOG2) + u], &sub[n * 4], sizeof(s16) * 4);
                    refined_cu = 1;
                }
            }
        }
        memcpy(mv_unrefined, core->mv, sizeof(mv_unrefined));
        xevdm_set_dec_info(ctx, core);
        mcore->dmvr_flag = 0;
        if (ctx->sps->tool_hmvp) update_history_buffer_parse_affine(core, ctx->sh.slice_type);
    }

    /* ---- the CU record of include/xevd_hip.h ---- */
    rb_reserve(s);
    i = s->n_cu++;
    s->x[i] = (uint16_t)x; s->y[i] = (uint16_t)y; s->log2w[i] = (uint8_t)log2_cuw; s->log2h[i] = (uint8_t)log2_cuh;
    mode = core->pred_mode;
    s->pred_mode[i] = (uint8_t)(mode == MODE_INTRA ? XGPU_MODE_INTRA : mode == MODE_IBC ? XGPU_MODE_IBC :
                                (mode == MODE_SKIP || mode == MODE_SKIP_MMVD) ? XGPU_MODE_SKIP : (mode == MODE_DIR || mode == MODE_DIR_MMVD) ? XGPU_MODE_DIR : XGPU_MODE_INTER);
    s->refi[i * 2] = core->refi[0]; s->refi[i * 2 + 1] = core->refi[1];
    memcpy(&s->mv[i * 4], (refined_cu && mode != MODE_INTRA && mode != MODE_IBC) ? mv_unrefined : core->mv, sizeof(s16) * 4);      /* (the backend refines for itself: xevdm_set_dec_info has put the first sub-block's refined vector into core->mv) */
    s->qp[i * 3] = core->qp_y; s->qp[i * 3 + 1] = core->qp_u; s->qp[i * 3 + 2] = core->qp_v;
    s->ipm[i * 2] = (uint8_t)core->ipm[0]; s->ipm[i * 2 + 1] = (uint8_t)core->ipm[1];
    s->ats[i] = (uint8_t)((mode == MODE_INTRA && mcore->ats_intra_cu) ? (1 | (mcore->ats_intra_mode_v << 1) | (mcore->ats_intra_mode_h << 2)) : 0);
    s->ats_inter[i] = (uint8_t)((mode != MODE_INTRA && mode != MODE_IBC) ? mcore->ats_inter_info : 0);
    s->affine[i] = (uint8_t)((mode != MODE_INTRA && mode != MODE_IBC && mcore->affine_flag) ? mcore->affine_flag + 1 : 0);
    s->dmvr[i] = (uint8_t)(mcore->dmvr_enable && ctx->sps->tool_dmvr);
    s->tree[i] = (uint8_t)(tree_cons.tree_type == TREE_L ? 1 : tree_cons.tree_type == TREE_C ? 2 : 0);      /* local dual tree: xgpu_cu_batch.tree */
    s->any_tree |= s->tree[i] != 0;
    memset(&s->affine_mv[i * 12], 0, sizeof(int16_t) * 12);
    if (s->affine[i]) {
        int l, v;
        for (l = 0; l < 2; l++) for (v = 0; v < 3; v++) { s->affine_mv[i * 12 + l * 6 + v * 2] = mcore->affine_mv[l][v][MV_X]; s->affine_mv[i * 12 + l * 6 + v * 2 + 1] = mcore->affine_mv[l][v][MV_Y]; }
        s->any_affine = 1;
    }
    s->any_dmvr |= s->dmvr[i]; s->any_ats |= s->ats[i] != 0; s->any_ats_inter |= s->ats_inter[i] != 0;
    s->cbf[i] = 0; s->cbf_sub[i] = 0;
    s->coef_off[i] = (uint32_t)s->coef_v.n;
    if (mode != MODE_SKIP) {
        int lt_w = log2_cuw, lt_h = log2_cuh;
        if (s->ats_inter[i]) xevdm_get_tu_size(s->ats_inter[i], log2_cuw, log2_cuh, &lt_w, &lt_h);
        for (c = 0; c < N_C; c++) {
            const size_t n = ((size_t)1 << (lt_w + lt_h)) >> (c ? 2 : 0);
            if (!core->is_coef[c] || (c == 0 ? !xevd_check_luma(ctx, core) : !xevd_check_chroma(ctx, core))) continue;
            s->cbf[i] |= (uint8_t)(1 << c);
            for (sb = 0; sb < MAX_SUB_TB_NUM; sb++) if (core->is_coef_sub[c][sb]) s->cbf_sub[i] |= (uint16_t)(1 << (4 * c + sb));
            memcpy(rb_push(&s->coef_v, &s->coef, sizeof(int16_t), n), core->coef[c], sizeof(int16_t) * n);
        }
    }
    {   /* MCU_SET_COD over the CU, xevdm.c:1394-1402 */
        u32 *map_scu = ctx->map_scu + core->scup;
        int j, k;
        for (j = 0; j < cuh >> MIN_CU_LOG2; j++, map_scu += ctx->w_scu) for (k = 0; k < cuw >> MIN_CU_LOG2; k++) MCU_SET_COD(map_scu[k]);
    }
}

/* our walk over one CTU's split tree, in the reference's decoding order (split modes and SUCO order from its own helpers) */
static void hip_recon_tree(XEVD_CTX *ctx, XEVD_CORE *core, int x, int y, int cuw, int cuh, int cud, int cup, TREE_CONS_NEW tree_cons)
{
    XEVDM_CTX *mctx = (XEVDM_CTX *)ctx;
    s8 split_mode, suco_flag = 0;
    xevd_get_split_mode(&split_mode, cud, cup, cuw, cuh, ctx->max_cuwh, &ctx->map_split[core->lcu_num]);
    xevdm_get_suco_flag(&suco_flag, cud, cup, cuw, cuh, ctx->max_cuwh, &mctx->map_suco[core->lcu_num]);
    if (split_mode != NO_SPLIT) {
        XEVD_SPLIT_STRUCT st;
        int order[SPLIT_MAX_PART_COUNT], k;
        TREE_CONS_NEW child = (TREE_CONS_NEW) { TREE_LC, eAll };
        xevd_split_get_part_structure(split_mode, x, y, cuw, cuh, cup, cud, ctx->log2_max_cuwh - MIN_CU_LOG2, &st);
        if (ctx->sps->tool_admvp && ctx->sps->sps_btt_flag) {
            child = tree_cons;
            if (tree_cons.mode_cons == eAll && !xevd_is_chroma_split_allowed(cuw, cuh, split_mode)) {
                child.mode_cons = xevd_derive_mode_cons(ctx, PEL2SCU(x) + PEL2SCU(y) * ctx->w_scu);
                child.tree_type = child.mode_cons == eOnlyIntra ? TREE_L : TREE_LC;
            }
        }
        xevdm_split_get_suco_order(xevd_split_is_vertical(split_mode) ? suco_flag : 0, split_mode, order);
        for (k = 0; k < st.part_count; k++) {
            const int p = order[k];
            if (st.x_pos[p] < ctx->w && st.y_pos[p] < ctx->h)
                hip_recon_tree(ctx, core, st.x_pos[p], st.y_pos[p], st.width[p], st.height[p], st.cud[p], st.cup[p], child);
        }
        /* a local dual tree started at this node: its chroma block follows the luma CUs as one CU (xevd_recon_tree, src_main/xevdm.c:1910-1917) */
        if (tree_cons.mode_cons == eAll && child.mode_cons == eOnlyIntra && child.tree_type == TREE_L)
            hip_recon_unit(ctx, core, x, y, XEVD_CONV_LOG2(cuw), XEVD_CONV_LOG2(cuh), cup, (TREE_CONS_NEW) { TREE_C, eOnlyIntra });
        return;
    }
    if (ctx->sh.slice_type == SLICE_I || (ctx->sps->tool_admvp && XEVD_CONV_LOG2(cuw) == 2 && XEVD_CONV_LOG2(cuh) == 2)) tree_cons.mode_cons = eOnlyIntra;
    hip_recon_unit(ctx, core, x, y, XEVD_CONV_LOG2(cuw), XEVD_CONV_LOG2(cuh), cup, tree_cons);
}

static int hip_open(XEVD_CTX *ctx)
{
    rb_state *s = rb_of(ctx);
    xgpu_seq_params sp;
    int8_t tbl[2][96];
    const int bdc = ctx->sps->bit_depth_chroma_minus8 + 8, off = 6 * (bdc - 8);
    int c, q;
    memset(&sp, 0, sizeof(sp));
    sp.device = 0; sp.width = ctx->w; sp.height = ctx->h;
    sp.bit_depth_luma = ctx->sps->bit_depth_luma_minus8 + 8; sp.bit_depth_chroma = bdc; sp.chroma_format_idc = ctx->sps->chroma_format_idc;
    sp.log2_ctu = ctx->log2_max_cuwh;
    sp.tool_iqt = ctx->sps->tool_iqt; sp.tool_admvp = ctx->sps->tool_admvp; sp.tool_addb = ctx->sps->tool_addb; sp.tool_alf = ctx->sps->tool_alf;
    sp.tool_eipd = ctx->sps->tool_eipd; sp.max_pics = 34;
    for (c = 0; c < 2; c++) { for (q = -off; q <= 57; q++) tbl[c][q + off] = (int8_t)xevd_qp_chroma_dynamic[c][q]; sp.chroma_qp_table[c] = tbl[c]; }
    return xgpu_open(&sp, &s->g);
}

/* ctx->fn_dec_slice */
/* ctx->tile[] (set_tile_info, xevdm.c:2162-2330) as the backend's tile grid; NULL for one tile */
static const xgpu_tile_grid *hip_tile_grid(XEVD_CTX *ctx, xgpu_tile_grid *g)
{
    int i;
    if (ctx->w_tile * ctx->h_tile <= 1) return NULL;
    memset(g, 0, sizeof(*g));
    g->n_cols = ctx->w_tile; g->n_rows = ctx->h_tile; g->loop_filter_across_tiles = ctx->pps.loop_filter_across_tiles_enabled_flag;
    for (i = 0; i < g->n_cols; i++) g->col_bd[i + 1] = g->col_bd[i] + ctx->tile[i].w_ctb;
    for (i = 0; i < g->n_rows; i++) g->row_bd[i + 1] = g->row_bd[i] + ctx->tile[i * ctx->w_tile].h_ctb;
    return g;
}

static int hip_dec_slice(XEVD_CTX *ctx, XEVD_CORE *core)
{
    rb_state *s = rb_of(ctx);
    XEVDM_CTX *mctx = (XEVDM_CTX *)ctx;
    XEVD_CORE *cm = ctx->core_mt[0];
    XEVD_TILE *tile;
    xgpu_cu_batch b;
    xgpu_frame_params fp;
    xgpu_dbatch *db = NULL;
    int ret, l, i, cx, cy;
    XEVD_BSR bs0;
    XEVD_SBAC sbac0;
    int t;
    /* A picture may come as several slices - one call each, every slice with its own tiles (ctx->tile_in_slice, set_tile_info xevdm.c:2185-2236): the CUs of all
       of them are gathered into ONE batch, which is launched when the last CTU of the picture has been parsed (ctx->num_ctb == 0, the test xevd_dec_nalu itself
       makes before it runs the in-loop filters, :3139) - with that last slice's header for everything picture-level, as the reference decoder filters */
    const int first_slice = ctx->num_ctb == (u32)ctx->f_lcu;
    if (ctx->sps->chroma_format_idc != 1) return XEVD_ERR_UNSUPPORTED;
    if (!s->g && (ret = hip_open(ctx)) < 0) return ret;

    if (first_slice) {
        s->n_cu = 0; s->coef_v.n = 0; s->any_affine = s->any_dmvr = s->any_ats = s->any_ats_inter = s->any_tree = 0; s->failed = 0; s->deblocked = 0;
        s->n_ctu = 0; s->pic_inter = 0;
        rb_luma_drop(s);                                               /* picture slots are reused: the host copies are per picture */
        s->ctu_start = realloc(s->ctu_start, sizeof(uint32_t) * (size_t)(ctx->f_lcu + 1));
    }
    if (ctx->sh.slice_type != SLICE_I) s->pic_inter = 1;
    ctx->sh.qp_prev_eco = ctx->sh.qp;
    xevd_mcpy(&bs0, &ctx->bs, sizeof(XEVD_BSR));                        /* the reader right behind the slice header: where the first tile starts */
    xevd_mcpy(&sbac0, GET_SBAC_DEC(&ctx->bs), sizeof(XEVD_SBAC));
    for (t = 0; t < ctx->num_tiles_in_slice; t++) {
        /* entropy decoding of one tile, as xevdm_dec_slice sets its worker up (xevdm.c:2640-2680): own reader + arithmetic decoder at the tile's
           entry point - the sum of the slice header's entry_point_offset_minus1 + 1 of the tiles before it, in bytes from the first tile */
        int x0, x1, y1;
        const int tile_idx = ctx->tile_in_slice[t];
        xevd_mcpy(cm, core, sizeof(XEVD_CORE));
        cm->ctx = ctx; cm->bs = &ctx->bs_mt[0]; cm->sbac = &ctx->sbac_dec_mt[0]; cm->tile_num = tile_idx; cm->thread_idx = 0;
        tile = &ctx->tile[tile_idx];
        tile->qp_prev_eco = ctx->sh.qp; tile->qp = ctx->sh.qp;
        xevd_mcpy(cm->bs, &bs0, sizeof(XEVD_BSR));
        xevd_mcpy(cm->sbac, &sbac0, sizeof(XEVD_SBAC));
        SET_SBAC_DEC(cm->bs, cm->sbac);
        if (t > 0) {
            int off = 0;
            for (i = 0; i < t; i++) off += ctx->sh.entry_point_offset_minus1[i] + 1;
            cm->bs->cur = bs0.cur - (bs0.leftbits >> 3) + off;          /* bytes the reader has fetched but not consumed lie before bs0.cur */
            cm->bs->leftbits = 0; cm->bs->code = 0;
            if (cm->bs->cur > cm->bs->end) return XEVD_ERR_MALFORMED_BITSTREAM;
        }
        xevd_mset((void *)ctx->sync_row, 0, ctx->h_lcu * sizeof(ctx->sync_row[0]));
        (void)xevd_tile_eco(cm);                                       /* xevd_tile_mt does not look at its status in this configuration either (xevdm.c:2573-2574) */

        /* the CUs of the tile in decoding order, appended to the picture's batch */
        x0 = tile->ctba_rs_first % ctx->w_lcu; cy = tile->ctba_rs_first / ctx->w_lcu;
        x1 = x0 + tile->w_ctb; y1 = cy + tile->h_ctb;
        for (; cy < y1; cy++) for (cx = x0; cx < x1; cx++) {
            cm->x_lcu = cx; cm->y_lcu = cy; cm->lcu_num = cy * ctx->w_lcu + cx;
            cm->x_pel = cx << ctx->log2_max_cuwh; cm->y_pel = cy << ctx->log2_max_cuwh;
            if (ctx->sps->tool_hmvp && cx == x0 && xevdm_hmvp_init(cm) != XEVD_OK) return XEVD_ERR;      /* xevdm.c:2499-2503 */
            if (s->n_ctu >= (int)ctx->f_lcu) return XEVD_ERR_MALFORMED_BITSTREAM;
            s->ctu_start[s->n_ctu++] = (uint32_t)s->n_cu;
            hip_recon_tree(ctx, cm, cm->x_pel, cm->y_pel, ctx->max_cuwh, ctx->max_cuwh, 0, 0, (TREE_CONS_NEW) { TREE_LC, eAll });
            if (s->failed) return XEVD_ERR_UNSUPPORTED;
        }
        ctx->num_ctb -= tile->w_ctb * tile->h_ctb;                    /* xevdm.c:2693-2697 */
    }
    xevd_mcpy(&ctx->bs, cm->bs, sizeof(XEVD_BSR));                      /* :2707-2711 */
    xevd_mcpy(&ctx->sbac_dec, cm->sbac, sizeof(XEVD_SBAC));
    if (ctx->num_ctb > 0) return XEVD_OK;                              /* further slices of this picture follow */
    if (s->n_ctu != (int)ctx->f_lcu) return XEVD_ERR_MALFORMED_BITSTREAM;
    s->ctu_start[ctx->f_lcu] = (uint32_t)s->n_cu;

    memset(&b, 0, sizeof(b));
    b.n_cu = s->n_cu; b.x = s->x; b.y = s->y; b.log2w = s->log2w; b.log2h = s->log2h; b.pred_mode = s->pred_mode; b.refi = s->refi; b.mv = s->mv;
    b.qp = s->qp; b.cbf = s->cbf; b.cbf_sub = s->cbf_sub; b.ipm = s->ipm;
    b.ats = s->any_ats ? s->ats : NULL; b.ats_inter = s->any_ats_inter ? s->ats_inter : NULL;
    b.coef_off = s->coef_off; b.coef = (const int16_t *)s->coef; b.n_coef = s->coef_v.n;
    b.n_ctu = ctx->f_lcu; b.ctu_cu_start = s->ctu_start;
    b.constrained_intra_pred = ctx->pps.constrained_intra_pred_flag;
    if (s->any_affine) { b.affine = s->affine; b.affine_mv = s->affine_mv; }
    if (s->any_dmvr) b.dmvr = s->dmvr;
    if (s->any_tree) b.tree = s->tree;
    b.htdf_slice_qp = ctx->sps->tool_htdf ? ctx->sh.qp : 0;
    b.tiles = hip_tile_grid(ctx, &s->grid);

    memset(&fp, 0, sizeof(fp));
    fp.pic = rb_slot(s, ctx->pic); fp.poc = ctx->poc.poc_val;
    for (l = 0; l < 2; l++) {
        fp.num_refp[l] = s->pic_inter ? mctx->dpm.num_refp[l] : 0;      /* (an I slice leaves the lists of the picture's P / B slices alone) */
        for (i = 0; i < fp.num_refp[l]; i++) { fp.refp_pic[i][l] = rb_slot(s, ctx->refp[i][l].pic); fp.refp_poc[i][l] = ctx->refp[i][l].poc; }
    }
    fp.qp_u_offset = ctx->sh.qp_u_offset; fp.qp_v_offset = ctx->sh.qp_v_offset;
    fp.deblock_alpha_offset = ctx->sh.sh_deblock_alpha_offset; fp.deblock_beta_offset = ctx->sh.sh_deblock_beta_offset;
    fp.deblock_on = ctx->sh.deblocking_filter_on; fp.alf_on = mctx->sh.alf_on;
    if ((ret = xgpu_batch_create(s->g, &b, &db)) < 0) return ret;
    if ((ret = xgpu_frame_begin(s->g, &fp)) >= 0) ret = xgpu_batch_recon(s->g, db);
    if (ret >= 0 && s->any_dmvr && (ret = xgpu_batch_dmvr_mvs(s->g, db, NULL, 0)) > 0) {
        /* the refined vectors of the picture, for the temporal candidates of later pictures: what processDMVR leaves in mcore->dmvr_mv and
           xevdm_set_dec_info copies into map_mv (src_main/xevdm_mc.c:1783-1797, xevdm_util.c:4327-4338) */
        const int n_sub = ret;
        int16_t *sub = (int16_t *)malloc(sizeof(int16_t) * 4 * (size_t)n_sub), *m = sub;
        if ((ret = xgpu_batch_dmvr_mvs(s->g, db, sub, n_sub)) == n_sub) {
            for (i = 0; i < s->n_cu; i++) {
                const int w = 1 << s->log2w[i], h = 1 << s->log2h[i], dx = w < 16 ? w : 16, dy = h < 16 ? h : 16;
                int sx, sy, u, v;
                if (!s->dmvr[i] || s->refi[i * 2] < 0 || s->refi[i * 2 + 1] < 0 || w < 8 || h < 8) continue;
                for (sy = 0; sy < h; sy += dy) for (sx = 0; sx < w; sx += dx, m += 4)
                    for (v = 0; v < dy >> MIN_CU_LOG2; v++) for (u = 0; u < dx >> MIN_CU_LOG2; u++)
                        memcpy(ctx->map_mv[((s->y[i] + sy) >> MIN_CU_LOG2) * ctx->w_scu + ((s->x[i] + sx) >> MIN_CU_LOG2) + v * ctx->w_scu + u], m, sizeof(int16_t) * 4);
            }
            if (m != sub + 4 * (size_t)n_sub) ret = XEVD_ERR;
        }
        free(sub);
    }
    xgpu_batch_destroy(s->g, db);
    return ret < 0 ? ret : XEVD_OK;
}

/* ctx->fn_deblock: xevd_dec_nalu calls it once per edge direction (and tile); both passes run on the device at the first call */
static int hip_deblock(void *arg)
{
    XEVD_CORE *core = (XEVD_CORE *)arg;
    rb_state *s = rb_of(core->ctx);
    if (s->deblocked) return XEVD_OK;
    s->deblocked = 1;
    return xgpu_deblock(s->g) < 0 ? XEVD_ERR : XEVD_OK;
}

/* mctx->fn_alf: call_dec_alf_process_aps + alf_process up to the filtering (xevdm_alf.c:1167-1195, 1251-1275) */
static int hip_alf(XEVD_CTX *ctx, XEVD_PIC *pic)
{
    XEVDM_CTX *mctx = (XEVDM_CTX *)ctx;
    rb_state *s = rb_of(ctx);
    ADAPTIVE_LOOP_FILTER *alf = (ADAPTIVE_LOOP_FILTER *)mctx->alf;
    ALF_SLICE_PARAM ap;
    xgpu_alf_params xp;
    int ret;
    (void)pic;
    ap.alf_ctb_flag = (u8 *)malloc(N_C * ctx->f_lcu * sizeof(u8));
    if (!ap.alf_ctb_flag) return XEVD_ERR;
    alf_load_paramline_from_aps_buffer2(alf, &ap, mctx->sh.aps_id_y, mctx->sh.aps_id_ch, mctx->sh.alf_chroma_idc);
    ap.is_ctb_alf_on = mctx->sh.alf_sh_param.is_ctb_alf_on;
    xevd_mcpy(ap.alf_ctb_flag, mctx->sh.alf_sh_param.alf_ctu_enable_flag, N_C * ctx->f_lcu * sizeof(u8));
    ap.filter_shapes = &alf->filter_shapes[0];
    alf_recon_coef(alf, &ap, CHANNEL_TYPE_LUMA, FALSE, TRUE);
    if (ap.enable_flag[U_C] || ap.enable_flag[V_C]) alf_recon_coef(alf, &ap, CHANNEL_TYPE_CHROMA, FALSE, FALSE);
    memset(&xp, 0, sizeof(xp));
    xp.enable[0] = ap.enable_flag[Y_C]; xp.enable[1] = ap.enable_flag[U_C]; xp.enable[2] = ap.enable_flag[V_C];
    xp.luma_coef = alf->coef_final; xp.chroma_coef = ap.chroma_coef;
    xp.ctb_flag = ap.alf_ctb_flag;                                     /* the luma flags: the first f_lcu entries */
    xp.across_tiles = ctx->pps.loop_filter_across_tiles_enabled_flag;
    xp.tiles = hip_tile_grid(ctx, &s->grid);
    ret = xgpu_alf(s->g, &xp);
    free(ap.alf_ctb_flag);
    return ret < 0 ? XEVD_ERR : XEVD_OK;
}

/* ctx->fn_picbuf_expand: padding on the device, then the active area into the reference's picture (xevd_pull, MD5 SEI, DRA read it there) */
static void hip_picbuf_expand(XEVD_CTX *ctx, XEVD_PIC *pic)
{
    rb_state *s = rb_of(ctx);
    if (xgpu_pad(s->g) < 0 || xgpu_frame_end(s->g) < 0) { s->failed = 1; return; }
    if (xgpu_pic_download(s->g, rb_slot(s, pic), pic->y, pic->s_l, pic->u, pic->v, pic->s_c) < 0) s->failed = 1;
}

/* ---- what the driver calls around xevd_create / xevd_delete ---- */
int refb_install(void *id)
{
    XEVD_CTX *ctx = (XEVD_CTX *)id;
    XEVDM_CTX *mctx = (XEVDM_CTX *)ctx;
    rb_state *s = (rb_state *)calloc(1, sizeof(rb_state));
    if (!s || ctx->pf) return -1;
    ctx->pf = s;
    ctx->fn_dec_slice = hip_dec_slice;
    ctx->fn_deblock = hip_deblock;
    mctx->fn_alf = hip_alf;
    ctx->fn_picbuf_expand = hip_picbuf_expand;
    return 0;
}
int refb_failed(void *id) { XEVD_CTX *ctx = (XEVD_CTX *)id; return ctx->pf ? rb_of(ctx)->failed : 0; }
void refb_uninstall(void *id)
{
    XEVD_CTX *ctx = (XEVD_CTX *)id;
    rb_state *s = rb_of(ctx);
    if (!s) return;
    if (s->g) xgpu_close(s->g);
    free(s->x); free(s->y); free(s->cbf_sub); free(s->log2w); free(s->log2h); free(s->pred_mode); free(s->qp); free(s->cbf); free(s->ipm);
    free(s->ats); free(s->ats_inter); free(s->affine); free(s->dmvr); free(s->refi); free(s->mv); free(s->affine_mv); free(s->coef_off);
    free(s->ctu_start); free(s->coef);
    free(s);
    ctx->pf = NULL;                                                   /* xevd_platform_deinit asserts it (src_base/xevd.c:2153) */
}
