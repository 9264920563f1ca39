/*
 * ref_binding_base.c - ref_binding.c's counterpart for the reference's BASELINE library (libxevdb: src_base/xevd.c + xevd_*.c): the MI355X backend of
 * include/xevd_hip.h behind the coarse function-table slots of the Baseline decoder.  TEST INFRASTRUCTURE (oracle/).  The reference's own Baseline front end
 * (NAL / SPS / PPS / slice header / SBAC / CU syntax / motion derivation / DPB / bumping / xevd_pull) runs unchanged and feeds the HIP backend, which replaces
 * everything xevd_dec_nalu does to a picture after entropy decoding (src_base/xevd.c:1905-1983).
 *
 * What is installed (on the XEVD_CTX a plain xevd_create() returned; nothing of the reference is edited):
 *   ctx->fn_dec_slice      <- hipb_dec_slice      src_base/xevd.c:1608-1657 (xevd_dec_slice): the reference's xevd_tile_eco (:1408-1468) still parses the picture's
 *                                                 one tile into XEVD_CU_DATA; the reconstruction half (xevd_tile_mt -> xevd_ctu_row_rec_mt -> xevd_recon_tree ->
 *                                                 xevd_recon_unit, :1528-1606, :1470-1526, :1019-1055, :678-756) is replaced by a walk that runs the reference's
 *                                                 cu_init + motion derivation + xevd_set_dec_info per CU and appends the CU to an xgpu_cu_batch; then
 *                                                 xgpu_frame_begin + xgpu_batch_create + xgpu_batch_recon
 *   ctx->fn_deblock        <- hipb_deblock        :1116-1243 (xevd_deblock; called per thread and edge direction, :1905-1975): xgpu_deblock, once per picture
 *   ctx->fn_picbuf_expand  <- hipb_picbuf_expand  src_base/xevd_util.c:365-427: xgpu_pad + xgpu_frame_end, then the active area into the reference's XEVD_PIC
 *   ctx->pf                <- the binding's state (src_base/xevd_def.h:1452-1470; xevd_platform_deinit asserts it NULL again, xevd.c:2153)
 * The statics (cu_init, coef_rect_to_series) are reached by #including src_base/xevd.c where it lies (it replaces that object in the link, oracle/Makefile.ref:
 * ref_decode_hip_base); no reference source is copied.  Limits (checked): 4:2:0.
 */
#include "xevd.c"
#include "../include/xevd_hip.h"

typedef struct {
    xgpu_ctx *g;
    struct { const XEVD_PIC *pic; int slot; } slots[64];
    int n_slots;
    int deblocked, failed;
    int n_cu, cap_cu;
    uint16_t *x, *y, *cbf_sub;
    uint8_t *log2w, *log2h, *pred_mode, *qp, *cbf, *ipm;
    int8_t *refi;
    int16_t *mv;
    uint32_t *coef_off, *ctu_start;
    int16_t *coef;
    size_t n_coef, cap_coef;
} rbb_state;

static rbb_state *rbb_of(XEVD_CTX *ctx) { return (rbb_state *)ctx->pf; }

static int rbb_slot(rbb_state *s, const XEVD_PIC *pic)
{
    int i;
    for (i = 0; i < s->n_slots; i++) if (s->slots[i].pic == pic) return s->slots[i].slot;
    if (s->n_slots == 64) return -1;
    s->slots[s->n_slots].pic = pic;
    s->slots[s->n_slots].slot = xgpu_pic_alloc(s->g);
    return s->slots[s->n_slots++].slot;
}

static void rbb_reserve(rbb_state *s)
{
    if (s->n_cu < s->cap_cu) return;
    s->cap_cu = s->cap_cu ? s->cap_cu * 2 : 4096;
#define G(f, k) s->f = realloc(s->f, sizeof(*s->f) * (size_t)s->cap_cu * (k))
    G(x, 1); G(y, 1); G(cbf_sub, 1); G(log2w, 1); G(log2h, 1); G(pred_mode, 1); G(qp, 3); G(cbf, 1); G(ipm, 2); G(refi, 2); G(mv, 4); G(coef_off, 1);
#undef G
}

/* one leaf CU: xevd_recon_unit (xevd.c:678-756) without the pixel work - the bookkeeping later CUs and pictures need, and the CU into the batch */
static void hipb_recon_unit(XEVD_CTX *ctx, XEVD_CORE *core, int x, int y, int log2_cuw, int log2_cuh)
{
    rbb_state *s = rbb_of(ctx);
    XEVD_CU_DATA *cu_data = &ctx->map_cu_data[core->lcu_num];
    const int cuw = 1 << log2_cuw, cuh = 1 << log2_cuh;
    int i, c, sb, mode, j, k;
    u32 *map_scu;
    core->log2_cuw = log2_cuw; core->log2_cuh = log2_cuh;
    core->x_scu = PEL2SCU(x); core->y_scu = PEL2SCU(y);
    core->scup = core->x_scu + core->y_scu * ctx->w_scu;
    cu_init(ctx, core, x, y, cuw, cuh);
    core->avail_lr = xevd_check_nev_avail(core->x_scu, core->y_scu, cuw, cuh, ctx->w_scu, ctx->h_scu, ctx->map_scu, ctx->map_tidx);
    if (core->pred_mode != MODE_SKIP) coef_rect_to_series(ctx, cu_data->coef, x, y, cuw, cuh, core->coef, core);
    if (core->pred_mode != MODE_INTRA) {            /* :702-730 without xevd_mc */
        core->avail_cu = xevd_get_avail_inter(core->x_scu, core->y_scu, ctx->w_scu, ctx->h_scu, core->scup, cuw, cuh, ctx->map_scu, ctx->map_tidx);
        if (core->pred_mode == MODE_SKIP) xevd_get_skip_motion(ctx, core);
        else if (core->inter_dir == PRED_DIR) {
            xevd_get_mv_dir(ctx->refp[0], ctx->poc.poc_val, core->scup + ((1 << (core->log2_cuw - MIN_CU_LOG2)) - 1) + ((1 << (core->log2_cuh - MIN_CU_LOG2)) - 1) * ctx->w_scu,
                            core->scup, ctx->w_scu, ctx->h_scu, core->mv);
            core->refi[REFP_0] = 0; core->refi[REFP_1] = 0;
        } else xevd_get_inter_motion(ctx, core);
        xevd_set_dec_info(ctx, core);
    }

    rbb_reserve(s);
    i = s->n_cu++;
    s->x[i] = (uint16_t)x; s->y[i] = (uint16_t)y; s->log2w[i] = (uint8_t)log2_cuw; s->log2h[i] = (uint8_t)log2_cuh;
    mode = core->pred_mode;
    s->pred_mode[i] = (uint8_t)(mode == MODE_INTRA ? XGPU_MODE_INTRA : mode == MODE_SKIP ? XGPU_MODE_SKIP : mode == MODE_DIR ? XGPU_MODE_DIR : XGPU_MODE_INTER);
    s->refi[i * 2] = core->refi[0]; s->refi[i * 2 + 1] = core->refi[1];
    memcpy(&s->mv[i * 4], core->mv, sizeof(s16) * 4);
    s->qp[i * 3] = core->qp_y; s->qp[i * 3 + 1] = core->qp_u; s->qp[i * 3 + 2] = core->qp_v;
    s->ipm[i * 2] = (uint8_t)core->ipm[0]; s->ipm[i * 2 + 1] = (uint8_t)core->ipm[1];
    s->cbf[i] = 0; s->cbf_sub[i] = 0;
    s->coef_off[i] = (uint32_t)s->n_coef;
    if (mode != MODE_SKIP)
        for (c = 0; c < N_C; c++) {
            const size_t n = ((size_t)1 << (log2_cuw + log2_cuh)) >> (c ? 2 : 0);
            if (!core->is_coef[c]) continue;
            s->cbf[i] |= (uint8_t)(1 << c);
            for (sb = 0; sb < MAX_SUB_TB_NUM; sb++) if (core->is_coef_sub[c][sb]) s->cbf_sub[i] |= (uint16_t)(1 << (4 * c + sb));
            if (s->n_coef + n > s->cap_coef) { s->cap_coef = (s->n_coef + n) * 2 + 4096; s->coef = realloc(s->coef, sizeof(int16_t) * s->cap_coef); }
            memcpy(s->coef + s->n_coef, core->coef[c], sizeof(int16_t) * n);
            s->n_coef += n;
        }
    map_scu = ctx->map_scu + core->scup;        /* MCU_SET_COD over the CU, :745-753 */
    for (j = 0; j < cuh >> MIN_CU_LOG2; j++, map_scu += ctx->w_scu) for (k = 0; k < cuw >> MIN_CU_LOG2; k++) MCU_SET_COD(map_scu[k]);
}

/* our walk over one CTU's split tree in the reference's decoding order (xevd_recon_tree, :1019-1055; split modes from its own helpers) */
static void hipb_recon_tree(XEVD_CTX *ctx, XEVD_CORE *core, int x, int y, int cuw, int cuh, int cud, int cup)
{
    s8 split_mode;
    xevd_get_split_mode(&split_mode, cud, cup, cuw, cuh, ctx->max_cuwh, &ctx->map_split[core->lcu_num]);
    if (split_mode != NO_SPLIT) {
        XEVD_SPLIT_STRUCT st;
        int p;
        xevd_split_get_part_structure(split_mode, x, y, cuw, cuh, cup, cud, ctx->log2_max_cuwh - MIN_CU_LOG2, &st);
        for (p = 0; p < st.part_count; p++)
            if (st.x_pos[p] < ctx->w && st.y_pos[p] < ctx->h) hipb_recon_tree(ctx, core, st.x_pos[p], st.y_pos[p], st.width[p], st.height[p], st.cud[p], st.cup[p]);
        return;
    }
    hipb_recon_unit(ctx, core, x, y, XEVD_CONV_LOG2(cuw), XEVD_CONV_LOG2(cuh));
}

static int hipb_open(XEVD_CTX *ctx)
{
    rbb_state *s = rbb_of(ctx);
    xgpu_seq_params sp;
    int8_t tbl[2][96];
    const int bdc = ctx->sps->bit_depth_chroma_minus8 + 8, off = 6 * (bdc - 8);
    int c, q;
    memset(&sp, 0, sizeof(sp));
    sp.device = 0; sp.width = ctx->w; sp.height = ctx->h;
    sp.bit_depth_luma = ctx->sps->bit_depth_luma_minus8 + 8; sp.bit_depth_chroma = bdc; sp.chroma_format_idc = ctx->sps->chroma_format_idc;
    sp.log2_ctu = ctx->log2_max_cuwh; sp.max_pics = 34;
    for (c = 0; c < 2; c++) { for (q = -off; q <= 57; q++) tbl[c][q + off] = (int8_t)xevd_qp_chroma_dynamic[c][q]; sp.chroma_qp_table[c] = tbl[c]; }
    return xgpu_open(&sp, &s->g);
}

/* ctx->fn_dec_slice: xevd_dec_slice (:1608-1657) with the reconstruction half replaced */
static int hipb_dec_slice(XEVD_CTX *ctx, XEVD_CORE *core)
{
    rbb_state *s = rbb_of(ctx);
    XEVD_CORE *cm = ctx->core_mt[0];
    XEVD_BSR bs0;
    XEVD_SBAC sbac0;
    xgpu_cu_batch b;
    xgpu_frame_params fp;
    xgpu_dbatch *db = NULL;
    int ret, l, i, cx, cy;
    if (ctx->sps->chroma_format_idc != 1) return XEVD_ERR_UNSUPPORTED;
    if (!s->g && (ret = hipb_open(ctx)) < 0) return ret;
    s->n_cu = 0; s->n_coef = 0; s->deblocked = 0;
    s->ctu_start = realloc(s->ctu_start, sizeof(uint32_t) * (size_t)(ctx->f_lcu + 1));

    xevd_mcpy(&bs0, &ctx->bs, sizeof(XEVD_BSR));
    xevd_mcpy(&sbac0, GET_SBAC_DEC(&ctx->bs), sizeof(XEVD_SBAC));
    ctx->sh.qp_prev_eco = ctx->sh.qp;
    xevd_mcpy(cm, core, sizeof(XEVD_CORE));
    cm->ctx = ctx; cm->bs = &ctx->bs_mt[0]; cm->sbac = &ctx->sbac_dec_mt[0]; cm->tile_num = 0; cm->thread_idx = 0;
    ctx->tile[0].qp_prev_eco = ctx->sh.qp; ctx->tile[0].qp = ctx->sh.qp;
    xevd_mcpy(cm->bs, &bs0, sizeof(XEVD_BSR));
    xevd_mcpy(cm->sbac, &sbac0, sizeof(XEVD_SBAC));
    SET_SBAC_DEC(cm->bs, cm->sbac);
    xevd_mset((void *)ctx->sync_row, 0, ctx->tile[0].h_ctb * sizeof(ctx->sync_row[0]));
    ret = xevd_tile_eco(cm);                                           /* entropy decoding of the picture's one tile */
    if (XEVD_FAILED(ret)) return ret;

    for (cy = 0; cy < (int)ctx->h_lcu; cy++) for (cx = 0; cx < (int)ctx->w_lcu; cx++) {
        cm->x_lcu = cx; cm->y_lcu = cy; cm->lcu_num = cy * ctx->w_lcu + cx;
        cm->x_pel = cx << ctx->log2_max_cuwh; cm->y_pel = cy << ctx->log2_max_cuwh;
        s->ctu_start[cm->lcu_num] = (uint32_t)s->n_cu;
        hipb_recon_tree(ctx, cm, cm->x_pel, cm->y_pel, ctx->max_cuwh, ctx->max_cuwh, 0, 0);
    }
    s->ctu_start[ctx->f_lcu] = (uint32_t)s->n_cu;
    ctx->num_ctb -= ctx->tile[0].w_ctb * ctx->tile[0].h_ctb;          /* :1649-1650 */
    xevd_mcpy(&ctx->bs, cm->bs, sizeof(XEVD_BSR));                      /* :1652-1653 */
    xevd_mcpy(&ctx->sbac_dec, cm->sbac, sizeof(XEVD_SBAC));

    memset(&b, 0, sizeof(b));
    b.n_cu = s->n_cu; b.x = s->x; b.y = s->y; b.log2w = s->log2w; b.log2h = s->log2h; b.pred_mode = s->pred_mode; b.refi = s->refi; b.mv = s->mv;
    b.qp = s->qp; b.cbf = s->cbf; b.cbf_sub = s->cbf_sub; b.ipm = s->ipm;
    b.coef_off = s->coef_off; b.coef = s->coef; b.n_coef = s->n_coef;
    b.n_ctu = ctx->f_lcu; b.ctu_cu_start = s->ctu_start;
    b.constrained_intra_pred = ctx->pps.constrained_intra_pred_flag;

    memset(&fp, 0, sizeof(fp));
    fp.pic = rbb_slot(s, ctx->pic); fp.poc = ctx->poc.poc_val;
    for (l = 0; l < 2; l++) {
        fp.num_refp[l] = ctx->sh.slice_type == SLICE_I ? 0 : ctx->dpm.num_refp[l];
        for (i = 0; i < fp.num_refp[l]; i++) { fp.refp_pic[i][l] = rbb_slot(s, ctx->refp[i][l].pic); fp.refp_poc[i][l] = ctx->refp[i][l].poc; }
    }
    fp.qp_u_offset = ctx->sh.qp_u_offset; fp.qp_v_offset = ctx->sh.qp_v_offset;
    fp.deblock_on = ctx->sh.deblocking_filter_on; fp.alf_on = 0;
    if ((ret = xgpu_batch_create(s->g, &b, &db)) < 0) return ret;
    if ((ret = xgpu_frame_begin(s->g, &fp)) >= 0) ret = xgpu_batch_recon(s->g, db);
    xgpu_batch_destroy(s->g, db);
    return ret < 0 ? ret : XEVD_OK;
}

/* ctx->fn_deblock: xevd_dec_nalu calls it per thread and edge direction (:1905-1975); both passes run on the device at the first call */
static int hipb_deblock(void *arg)
{
    XEVD_CORE *core = (XEVD_CORE *)arg;
    rbb_state *s = rbb_of(core->ctx);
    if (s->deblocked) return XEVD_OK;
    s->deblocked = 1;
    return xgpu_deblock(s->g) < 0 ? XEVD_ERR : XEVD_OK;
}

/* ctx->fn_picbuf_expand: padding on the device, then the active area into the reference's picture (xevd_pull and the MD5 check read it there) */
static void hipb_picbuf_expand(XEVD_CTX *ctx, XEVD_PIC *pic)
{
    rbb_state *s = rbb_of(ctx);
    if (xgpu_pad(s->g) < 0 || xgpu_frame_end(s->g) < 0) { s->failed = 1; return; }
    if (xgpu_pic_download(s->g, rbb_slot(s, pic), pic->y, pic->s_l, pic->u, pic->v, pic->s_c) < 0) s->failed = 1;
}

/* ---- what the driver calls around xevd_create / xevd_delete (the same three names as ref_binding.c: ref_decode.c links either) ---- */
int refb_install(void *id)
{
    XEVD_CTX *ctx = (XEVD_CTX *)id;
    rbb_state *s = (rbb_state *)calloc(1, sizeof(rbb_state));
    if (!s || ctx->pf) return -1;
    ctx->pf = s;
    ctx->fn_dec_slice = hipb_dec_slice;
    ctx->fn_deblock = hipb_deblock;
    ctx->fn_picbuf_expand = hipb_picbuf_expand;
    return 0;
}
int refb_failed(void *id) { XEVD_CTX *ctx = (XEVD_CTX *)id; return ctx->pf ? rbb_of(ctx)->failed : 0; }
void refb_uninstall(void *id)
{
    XEVD_CTX *ctx = (XEVD_CTX *)id;
    rbb_state *s = rbb_of(ctx);
    if (!s) return;
    if (s->g) xgpu_close(s->g);
    free(s->x); free(s->y); free(s->cbf_sub); free(s->log2w); free(s->log2h); free(s->pred_mode); free(s->qp); free(s->cbf); free(s->ipm);
    free(s->refi); free(s->mv); free(s->coef_off); free(s->ctu_start); free(s->coef);
    free(s);
    ctx->pf = NULL;
}
