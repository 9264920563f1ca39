/*
 * ref_harness.c - OUR driver around the REAL reference functions (test infrastructure only).
 *
 * Compiled by oracle/Makefile.ref against the reference's headers where they lie (/root/reference) and linked
 * to oracle/_ref/libxevd_ref.so; the output (oracle/_ref/libref_harness.so) exists only in the development
 * container and travels to the GPU box as a prebuilt .so.  No reference source is copied: this file only
 * fills the reference's own structs (XEVD_CTX / XEVD_CORE / XEVD_PIC / XEVD_REFP) from our batch format and
 * calls the reference's exported functions, so that picture-level golden outputs are computed by reference
 * arithmetic:
 *   xevd_sub_block_itdq   src_base/xevd_itdq.c:544      xevdm_sub_block_itdq  src_main/xevdm_itdq.c:790
 *   xevd_mc               src_base/xevd_mc.c:469        xevdm_mc              src_main/xevdm_mc.c:1860
 *   xevd_recon            src_base/xevd_recon.c:35      xevd_set_dec_info     src_base/xevd_util.c:1574
 *   xevd_deblock_cu_ver / _hor  src_base/xevd_df.c:385 / :291
 *   xevd_picbuf_lc_expand src_base/xevd_util.c:420
 * The loop structure mirrors xevd_recon_unit (src_base/xevd.c:678-756) and the two deblocking passes
 * (src_base/xevd.c:1116-1243).
 */
#include <stdlib.h>
#include <string.h>
#include "xevdm_def.h"
#include "xevdm_mc.h"
#include "xevdm_itdq.h"
#include "xevdm_df.h"
#include "xevdm_alf.h"
#include "xevdm_ipred.h"
#include "xevdm_dra.h"
#include "xevd_oracle.h"

/* reference tables selected like xevd_platform_init does (src_base/xevd.c:2074-2149) */
static void select_tables(XEVD_CTX *ctx, int simd)
{
    /* the interpolation-table pointers are process globals (xevd_mc.c:137-138) that only xevdm_mc rewrites
       (xevdm_mc.c:1914-1924); start every picture from their initial Baseline value */
    tbl_mc_l_coeff = xevd_tbl_mc_l_coeff;
    tbl_mc_c_coeff = xevd_tbl_mc_c_coeff;
    if (simd) {
        xevd_func_mc_l = xevd_tbl_mc_l_avx;
        xevd_func_mc_c = xevd_tbl_mc_c_avx;
        xevd_func_average_no_clip = &xevd_average_16b_no_clip_sse;
        ctx->fn_itxb = &xevd_tbl_itxb_avx;
        ctx->fn_recon = &xevd_recon_avx;
        ctx->fn_dbk = &xevd_tbl_dbk_sse;
        ctx->fn_dbk_chroma = &xevd_tbl_dbk_chroma_sse;
        xevdm_fn_itx = &xevdm_tbl_itx_avx;
        xevdm_func_dmvr_mc_l = xevdm_tbl_dmvr_mc_l_sse; xevdm_func_dmvr_mc_c = xevdm_tbl_dmvr_mc_c_sse; xevdm_func_bl_mc_l = xevdm_tbl_bl_mc_l_sse;      /* xevdm.c:3415-3417 */
    } else {
        xevd_func_mc_l = xevd_tbl_mc_l;
        xevd_func_mc_c = xevd_tbl_mc_c;
        xevd_func_average_no_clip = &xevd_average_16b_no_clip;
        ctx->fn_itxb = &xevd_tbl_itxb;
        ctx->fn_recon = &xevd_recon;
        ctx->fn_dbk = &xevd_tbl_dbk;
        ctx->fn_dbk_chroma = &xevd_tbl_dbk_chroma;
        xevdm_fn_itx = &xevdm_tbl_itx;
        xevdm_func_dmvr_mc_l = xevdm_tbl_dmvr_mc_l; xevdm_func_dmvr_mc_c = xevdm_tbl_dmvr_mc_c; xevdm_func_bl_mc_l = xevdm_tbl_bl_mc_l;                  /* xevdm.c:3443-3445 */
    }
    /* ATS: matrices and the function table, as xevd_create does (src_main/xevdm.c:3441, 3581-3582) */
    xevdm_init_multi_tbl();
    xevd_init_multi_inv_tbl();
    xevd_func_itrans = xevdm_itrans_map_tbl;
}

typedef struct {
    XEVD_CTX  *ctx;
    XEVD_CORE *core;
    XEVD_SPS   sps;
    XEVD_PIC   cur;
    XEVD_PIC   rpic[XGPU_MAX_REFS][2];
    u8        *map_tidx;
    s8        *map_ipm;
    u32       *map_cu_mode;
} harness;

static void fill_pic(XEVD_PIC *p, const orc_pic *o, const xgpu_seq_params *sp)
{
    memset(p, 0, sizeof(*p));
    p->y = o->y; p->u = o->u; p->v = o->v;
    p->s_l = o->s_l; p->s_c = o->s_c;
    p->w_l = sp->width; p->h_l = sp->height; p->w_c = sp->width >> 1; p->h_c = sp->height >> 1;
    p->pad_l = XGPU_PAD_L; p->pad_c = XGPU_PAD_C;
    p->poc = o->poc;
}

static harness *harness_new(const xgpu_seq_params *sp, const orc_frame *fr, orc_maps *m, int simd)
{
    harness *h = (harness *)calloc(1, sizeof(harness));
    int i, l, n = m->w_scu * m->h_scu;
    h->ctx = (XEVD_CTX *)calloc(1, sizeof(XEVDM_CTX));     /* XEVDM_CTX begins with XEVD_CTX bctx */
    h->core = (XEVD_CORE *)calloc(1, sizeof(XEVDM_CORE));
    h->map_tidx = (u8 *)calloc(n, 1);
    h->map_ipm = (s8 *)calloc(n, 1);
    h->map_cu_mode = (u32 *)calloc(n, sizeof(u32));
    h->sps.bit_depth_luma_minus8 = sp->bit_depth_luma - 8;
    h->sps.bit_depth_chroma_minus8 = sp->bit_depth_chroma - 8;
    h->sps.chroma_format_idc = sp->chroma_format_idc;
    h->ctx->sps = &h->sps;
    h->ctx->w = sp->width; h->ctx->h = sp->height;
    h->ctx->w_scu = m->w_scu; h->ctx->h_scu = m->h_scu; h->ctx->f_scu = n;
    h->ctx->map_scu = m->map_scu;
    h->ctx->map_refi = (s8 (*)[REFP_NUM])m->map_refi;
    h->ctx->map_mv = (s16 (*)[REFP_NUM][MV_D])m->map_mv;
    h->ctx->map_tidx = h->map_tidx;
    h->ctx->map_ipm = h->map_ipm;
    h->ctx->map_cu_mode = h->map_cu_mode;
    h->ctx->slice_num = 0;
    fill_pic(&h->cur, &fr->cur, sp);
    h->cur.pic_qp_u_offset = fr->qp_u_offset;
    h->cur.pic_qp_v_offset = fr->qp_v_offset;
    h->ctx->pic = &h->cur;
    for (i = 0; i < XGPU_MAX_REFS; i++) for (l = 0; l < 2; l++) {
        int i2, l2, found = 0;
        fill_pic(&h->rpic[i][l], &fr->refp[i][l], sp);
        h->ctx->refp[i][l].pic = &h->rpic[i][l];
        /* one XEVD_PIC per distinct picture: ADDB's get_bs compares reference pictures by pointer (xevdm_df.c:461-464) */
        for (i2 = 0; i2 <= i && !found; i2++) for (l2 = 0; l2 < 2 && !found; l2++) {
            if (i2 == i && l2 >= l) break;
            if (fr->refp[i2][l2].y == fr->refp[i][l].y) { h->ctx->refp[i][l].pic = h->ctx->refp[i2][l2].pic; found = 1; }
        }
        h->ctx->refp[i][l].poc = fr->refp[i][l].poc;
    }
    select_tables(h->ctx, simd);
    /* chroma QP mapping exactly as sequence_init sets it up (src_base/xevd.c:347-358) */
    xevd_set_chroma_qp_tbl_loc(sp->bit_depth_luma);
    if (sp->chroma_qp_table[0] && sp->chroma_qp_table[1]) {
        const int boff = 6 * (sp->bit_depth_chroma - 8);
        for (i = -boff; i <= 57; i++) {
            xevd_qp_chroma_dynamic[0][i] = sp->chroma_qp_table[0][i + boff];
            xevd_qp_chroma_dynamic[1][i] = sp->chroma_qp_table[1][i + boff];
        }
    } else {
        for (i = 0; i < XEVD_MAX_QP_TABLE_SIZE; i++) {
            /* the sequence default: the Main table with tool_iqt, else the Baseline one (xevdm.c:471-479) */
            xevd_qp_chroma_dynamic[0][i] = sp->tool_iqt ? xevd_tbl_qp_chroma_adjust_main[i] : xevd_tbl_qp_chroma_adjust_base[i];
            xevd_qp_chroma_dynamic[1][i] = xevd_qp_chroma_dynamic[0][i];
        }
    }
    return h;
}

static void harness_free(harness *h)
{
    free(h->map_tidx); free(h->map_ipm); free(h->map_cu_mode); free(h->core); free(h->ctx); free(h);
}

int refh_recon_batch_ex(const xgpu_seq_params *sp, const orc_frame *fr, const xgpu_cu_batch *b, orc_maps *m,
                        int16_t *resid_out, int simd, int16_t *dmvr_mv_out);
int refh_recon_batch(const xgpu_seq_params *sp, const orc_frame *fr, const xgpu_cu_batch *b, orc_maps *m,
                     int16_t *resid_out, int simd)
{
    return refh_recon_batch_ex(sp, fr, b, m, resid_out, simd, NULL);
}
/* dmvr_mv_out: as orc_recon_batch_ex - what xevdm_mc leaves in mcore->dmvr_mv (refined) / core->mv (not refined) for the DMVR candidates */
int refh_recon_batch_ex(const xgpu_seq_params *sp, const orc_frame *fr, const xgpu_cu_batch *b, orc_maps *m,
                        int16_t *resid_out, int simd, int16_t *dmvr_mv_out)
{
    size_t dmvr_n = 0;
    int dmvr_applied = 0;
    harness *hn = harness_new(sp, fr, m, simd);
    XEVD_CTX *ctx = hn->ctx; XEVD_CORE *core = hn->core;
    XEVDM_CORE *mcore = (XEVDM_CORE *)core;
    const int main_path = sp->tool_admvp || sp->tool_iqt || b->ats != NULL || b->ats_inter != NULL || b->affine != NULL || b->htdf_slice_qp != 0 || b->dmvr != NULL;
    int i, c;

    if (b->affine) {
        /* xevdm_affine_mc reads the interpolation tables through the process-global pointers that only xevdm_mc sets (xevdm_mc.c:1914-1924): it
           works with whatever the last regular inter CU of the process left there (a stream whose first inter CU is affine gets the Baseline
           table, whose rows for the sixteenth-sample phases are empty).  The batch semantics are those of the steady state: let one regular
           CU go first. */
        s8 refi0[2] = { -1, -1 }; s16 mv0[2][2] = { { 0, 0 }, { 0, 0 } }; u8 dmvr_flag = 0;
        refi0[ctx->refp[0][0].pic ? 0 : 1] = 0;
        xevdm_mc(0, 0, ctx->w, ctx->h, 4, 4, refi0, mv0, ctx->refp, core->pred, fr->cur.poc, mcore->dmvr_template, mcore->dmvr_ref_pred_interpolated,
                 mcore->dmvr_half_pred_interpolated, 0, mcore->dmvr_padding_buf, &dmvr_flag, mcore->dmvr_mv, sp->tool_admvp,
                 sp->bit_depth_luma, sp->bit_depth_chroma, sp->chroma_format_idc);
    }
    for (i = 0; i < b->n_cu; i++) {
        const int x = b->x[i], y = b->y[i], lw = b->log2w[i], lh = b->log2h[i], w = 1 << lw, h = 1 << lh;
        size_t off = b->coef_off[i], o = off;
        dmvr_applied = 0;
        core->log2_cuw = lw; core->log2_cuh = lh; core->cuw = w; core->cuh = h;
        core->x_scu = x >> 2; core->y_scu = y >> 2;
        core->scup = core->x_scu + core->y_scu * ctx->w_scu;
        core->pred_mode = b->pred_mode[i];
        core->qp_y = b->qp[i * 3]; core->qp_u = b->qp[i * 3 + 1]; core->qp_v = b->qp[i * 3 + 2];
        core->qp = core->qp_y - 6 * (sp->bit_depth_luma - 8);
        core->ipm[0] = b->ipm ? b->ipm[i * 2] : 0; core->ipm[1] = b->ipm ? b->ipm[i * 2 + 1] : 0;
        memset(core->is_coef_sub, 0, sizeof(core->is_coef_sub));
        const int ibc = b->pred_mode[i] == XGPU_MODE_IBC;
        const u8 ai = (b->ats_inter && b->pred_mode[i] != XGPU_MODE_INTRA && !ibc) ? b->ats_inter[i] : 0;
        int tuw = w, tuh = h;
        if (ai) { int lt_w, lt_h; xevdm_get_tu_size(ai, lw, lh, &lt_w, &lt_h); tuw = 1 << lt_w; tuh = 1 << lt_h; }
        for (c = 0; c < 3; c++) {
            const int n = c ? (tuw >> 1) * (tuh >> 1) : tuw * tuh;
            int sb;
            core->is_coef[c] = (b->cbf[i] >> c) & 1;
            /* nnz_sub of the 64x64 sub-blocks, (j<<1)|i; a CU <= 64 only has sub-block 0 */
            for (sb = 0; sb < 4; sb++) {
                const int exists = (sb & 1) < (lw > 6 ? 2 : 1) && (sb >> 1) < (lh > 6 ? 2 : 1);
                core->is_coef_sub[c][sb] = core->is_coef[c] && exists &&
                    ((lw <= 6 && lh <= 6) || !b->cbf_sub || ((b->cbf_sub[i] >> (4 * c + sb)) & 1));
            }
            if (core->is_coef[c]) { memcpy(core->coef[c], b->coef + o, sizeof(s16) * n); o += n; }
        }
        if (b->pred_mode[i] == XGPU_MODE_INTRA) {
            core->refi[0] = core->refi[1] = -1;
            memset(core->mv, 0, sizeof(core->mv));
        } else {
            core->refi[0] = b->refi[i * 2]; core->refi[1] = b->refi[i * 2 + 1];
            memcpy(core->mv, &b->mv[i * 4], sizeof(s16) * 4);
        }
        /* inverse quantisation + transform: xevd.c:694-698 (xevd_lc_itdq) */
        if (main_path) {
            const int a = (b->ats && b->pred_mode[i] == XGPU_MODE_INTRA) ? b->ats[i] : 0;      /* xevdm.c:602 */
            xevdm_sub_block_itdq(ctx, core->coef, lw, lh, core->qp_y, core->qp_u, core->qp_v, core->is_coef, core->is_coef_sub,
                                 sp->tool_iqt, a & 1, (u8)((((a >> 2) & 1) << 1) | ((a >> 1) & 1)), ai, sp->bit_depth_luma, sp->chroma_format_idc);
        }
        else
            xevd_sub_block_itdq(ctx, core->coef, lw, lh, core->qp_y, core->qp_u, core->qp_v, core->is_coef, core->is_coef_sub,
                                sp->bit_depth_luma, sp->chroma_format_idc);
        if (resid_out) {
            o = off;
            for (c = 0; c < 3; c++) {
                const int n = c ? (tuw >> 1) * (tuh >> 1) : tuw * tuh;
                if (core->is_coef[c]) { memcpy(resid_out + o, core->coef[c], sizeof(s16) * n); o += n; }
            }
        }
        if (b->pred_mode[i] == XGPU_MODE_INTRA) {
            /* xevd_recon_unit's intra branch (xevd.c:731-741; Main with tool_eipd = 0: xevdm.c:1346-1381): availability from the
               COD flags set so far, neighbour samples, the Baseline predictors, then the same reconstruction */
            int xx = x, yy = y, ww = w, hh = h;
            ctx->pps.constrained_intra_pred_flag = b->constrained_intra_pred;
            core->avail_lr = xevd_check_nev_avail(core->x_scu, core->y_scu, w, h, ctx->w_scu, ctx->h_scu, ctx->map_scu, ctx->map_tidx);
            core->avail_cu = xevd_get_avail_intra(core->x_scu, core->y_scu, ctx->w_scu, ctx->h_scu, core->scup, lw, lh, ctx->map_scu, ctx->map_tidx);
            for (c = 0; c < 3; c++) {
                pel *plane = c == 0 ? ctx->pic->y : (c == 1 ? ctx->pic->u : ctx->pic->v);
                const int s = c ? ctx->pic->s_c : ctx->pic->s_l;
                if (c == 1) { xx >>= 1; yy >>= 1; ww >>= 1; hh >>= 1; }
                if (sp->tool_eipd)      /* get_nbr_yuv, xevdm.c:605-655 */
                    xevdm_get_nbr(xx, yy, ww, hh, plane + yy * s + xx, s, core->avail_cu, core->nb, core->scup, ctx->map_scu, ctx->w_scu, ctx->h_scu,
                                  c, b->constrained_intra_pred, ctx->map_tidx, sp->bit_depth_luma, sp->chroma_format_idc);
                else
                    xevd_get_nbr_b(xx, yy, ww, hh, plane + yy * s + xx, s, core->avail_cu, core->nb, core->scup, ctx->map_scu, ctx->w_scu, ctx->h_scu,
                                   c, b->constrained_intra_pred, ctx->map_tidx, sp->bit_depth_luma, sp->chroma_format_idc);
            }
            if (sp->tool_eipd) {        /* xevdm.c:1352-1360 */
                xevdm_ipred(core->nb[0][0] + 2, core->nb[0][1] + h, core->nb[0][2] + 2, core->avail_lr, core->pred[0][Y_C], core->ipm[0], w, h, sp->bit_depth_luma);
                xevdm_ipred_uv(core->nb[1][0] + 2, core->nb[1][1] + (h >> 1), core->nb[1][2] + 2, core->avail_lr, core->pred[0][U_C], core->ipm[1], core->ipm[0], w >> 1, h >> 1, sp->bit_depth_chroma);
                xevdm_ipred_uv(core->nb[2][0] + 2, core->nb[2][1] + (h >> 1), core->nb[2][2] + 2, core->avail_lr, core->pred[0][V_C], core->ipm[1], core->ipm[0], w >> 1, h >> 1, sp->bit_depth_chroma);
            } else {
                xevd_ipred_b(core->nb[0][0] + 2, core->nb[0][1] + h, core->nb[0][2] + 2, core->avail_lr, core->pred[0][Y_C], core->ipm[0], w, h);
                xevd_ipred_uv_b(core->nb[1][0] + 2, core->nb[1][1] + (h >> 1), core->nb[1][2] + 2, core->avail_lr, core->pred[0][U_C], core->ipm[1], core->ipm[0], w >> 1, h >> 1);
                xevd_ipred_uv_b(core->nb[2][0] + 2, core->nb[2][1] + (h >> 1), core->nb[2][2] + 2, core->avail_lr, core->pred[0][V_C], core->ipm[1], core->ipm[0], w >> 1, h >> 1);
            }
            xevd_recon_yuv(ctx, core, x, y, w, h);
        } else {
            /* prediction: xevd.c:725-726 / xevdm.c:1311-1316 (DMVR off); affine CUs xevdm.c:1290-1295; IBC xevdm.c:1263-1268 */
            const int vn = b->affine ? b->affine[i] : 0;
            if (ibc) {
                TREE_CONS tc = { 0, TREE_LC, eAll };
                core->refi[0] = core->refi[1] = -1;
                core->mv[1][0] = core->mv[1][1] = 0;
                xevdm_IBC_mc(x, y, lw, lh, core->mv[0], ctx->pic, core->pred[0], tc, sp->chroma_format_idc);
            } else if (vn) {
                int l, v;
                mcore->affine_flag = (u8)(vn - 1);
                memset(mcore->affine_mv, 0, sizeof(mcore->affine_mv));
                for (l = 0; l < 2; l++) for (v = 0; v < 3; v++) {
                    mcore->affine_mv[l][v][MV_X] = b->affine_mv[i * 12 + l * 6 + v * 2];
                    mcore->affine_mv[l][v][MV_Y] = b->affine_mv[i * 12 + l * 6 + v * 2 + 1];
                }
                xevdm_affine_mc(x, y, ctx->w, ctx->h, w, h, core->refi, mcore->affine_mv, ctx->refp, core->pred, vn, core->eif_tmp_buffer,
                                sp->bit_depth_luma, sp->bit_depth_chroma, sp->chroma_format_idc);
            } else if (main_path) {
                /* xevdm.c:1326-1331: apply_DMVR = (mcore->dmvr_enable == 1) && sps->tool_dmvr - the batch's flag */
                u8 dmvr_flag = 0;
                const int cand = b->dmvr && b->dmvr[i];
                xevdm_mc(x, y, ctx->w, ctx->h, w, h, core->refi, core->mv, ctx->refp, core->pred, fr->cur.poc,
                         mcore->dmvr_template, mcore->dmvr_ref_pred_interpolated, mcore->dmvr_half_pred_interpolated, cand,
                         mcore->dmvr_padding_buf, &dmvr_flag, mcore->dmvr_mv, sp->tool_admvp,
                         sp->bit_depth_luma, sp->bit_depth_chroma, sp->chroma_format_idc);
                dmvr_applied = dmvr_flag;
                if (cand && core->refi[0] >= 0 && core->refi[1] >= 0 && w >= 8 && h >= 8) {
                    /* dmvr_mv is indexed by SCU inside the CU (:1783-1797): one entry per 16x16 sub-block, raster order */
                    const int dx = w < 16 ? w : 16, dy = h < 16 ? h : 16;
                    int sx, sy;
                    for (sy = 0; sy < h; sy += dy) for (sx = 0; sx < w; sx += dx, dmvr_n++) {
                        const int idx = (sx >> 2) + (sy >> 2) * (w >> 2);
                        if (!dmvr_mv_out) continue;
                        if (dmvr_flag) memcpy(dmvr_mv_out + dmvr_n * 4, mcore->dmvr_mv[idx], 4 * sizeof(s16));
                        else memcpy(dmvr_mv_out + dmvr_n * 4, core->mv, 4 * sizeof(s16));
                    }
                }
            } else {
                xevd_mc(x, y, ctx->w, ctx->h, w, h, core->refi, core->mv, ctx->refp, core->pred, fr->cur.poc,
                        sp->bit_depth_luma, sp->bit_depth_chroma, sp->chroma_format_idc);
            }
            /* reconstruction: xevd_recon_yuv, xevd_recon.c:70-92; with ATS-inter the Main profile's own (xevdm.c:1379) */
            if (ai) {
                TREE_CONS tc = { 0, TREE_LC, eAll };
                xevdm_recon_yuv(x, y, w, h, core->coef, core->pred[0], core->is_coef, ctx->pic, ai, tc, sp->bit_depth_luma, sp->chroma_format_idc);
            } else
                xevd_recon_yuv(ctx, core, x, y, w, h);
        }
        if (b->htdf_slice_qp && !ibc && (core->is_coef[Y_C] || b->pred_mode[i] == XGPU_MODE_INTRA)) {      /* xevdm.c:1381-1392 */
            const u16 avail_cu = xevd_get_avail_intra(core->x_scu, core->y_scu, ctx->w_scu, ctx->h_scu, core->scup, lw, lh, ctx->map_scu, ctx->map_tidx);
            const int cif = b->pred_mode[i] == XGPU_MODE_INTRA && b->constrained_intra_pred;
            pel *rec = ctx->pic->y + y * ctx->pic->s_l + x;
            xevdm_htdf(rec, b->htdf_slice_qp, w, h, ctx->pic->s_l, b->pred_mode[i] == XGPU_MODE_INTRA, rec, ctx->pic->s_l, avail_cu, core->scup,
                       ctx->w_scu, ctx->h_scu, ctx->map_scu, cif, sp->bit_depth_luma);
        }
        xevd_set_dec_info(ctx, core);
        if (dmvr_applied && !sp->tool_addb) {
            /* xevdm_set_dec_info stores dmvr_mv in ctx->map_mv (xevdm_util.c:4327-4332), and that is the array the Main library's baseline deblocking
               filter reads (xevdm_df.c:118,209; the ADDB filter gets map_unrefined_mv): one map here, holding what the sequence's filter reads */
            int r, q;
            for (r = 0; r < h >> 2; r++) for (q = 0; q < w >> 2; q++)
                memcpy(ctx->map_mv[core->scup + r * ctx->w_scu + q], mcore->dmvr_mv[r * (w >> 2) + q], 4 * sizeof(s16));
        }
        {   /* xevdm_set_dec_info's IBC flag (xevdm_util.c:4289-4296) */
            int r, q;
            for (r = 0; r < h >> 2; r++) for (q = 0; q < w >> 2; q++) {
                if (ibc) MCU_SET_IBC(ctx->map_scu[core->scup + r * ctx->w_scu + q]); else MCU_CLR_IBC(ctx->map_scu[core->scup + r * ctx->w_scu + q]);
            }
        }
        if (b->affine && b->affine[i] && b->pred_mode[i] != XGPU_MODE_INTRA)      /* xevdm_set_dec_info's affine tail, xevdm_util.c:4378-4381 */
            xevdm_set_affine_mvf(ctx, core);
        if (ai) {      /* xevdm_set_dec_info's ATS-inter tail (xevdm_util.c:4321, :4375) */
            int r, q;
            xevdm_set_cu_cbf_flags((u8)core->is_coef[Y_C], ai, lw, lh, ctx->map_scu + core->scup, ctx->w_scu);
            if (m->map_ats) for (r = 0; r < h >> 2; r++) for (q = 0; q < w >> 2; q++) m->map_ats[core->scup + r * ctx->w_scu + q] = ai;
        }
        {   /* MCU_SET_COD over the CU, xevd.c:746-754 */
            int r, q;
            for (r = 0; r < h >> 2; r++) for (q = 0; q < w >> 2; q++)
                MCU_SET_COD(ctx->map_scu[core->scup + r * ctx->w_scu + q]);
        }
    }
    harness_free(hn);
    return 0;
}

static int deblock_main(const xgpu_seq_params *sp, const orc_frame *fr, const xgpu_cu_batch *b, orc_maps *m, int addb, int alpha_off, int beta_off);

int refh_deblock_baseline(const xgpu_seq_params *sp, const orc_frame *fr, const xgpu_cu_batch *b, orc_maps *m, int simd)
{
    /* a Main-profile stream without ADDB runs the Main library's own copy of this filter through the same deblock_tree */
    if (sp->tool_admvp || sp->tool_iqt || sp->log2_ctu > 6 || b->ats || b->ats_inter) return deblock_main(sp, fr, b, m, 0, 0, 0);
    harness *hn = harness_new(sp, fr, m, simd);
    XEVD_CTX *ctx = hn->ctx;
    int i, k;
    /* vertical edges, then horizontal edges; COD cleared before each pass (xevd.c:1179-1188, 1221-1227);
       leaf CUs visited in decode order as deblock_tree does (xevd.c:1057-1114) */
    for (k = 0; k < (int)ctx->f_scu; k++) MCU_CLR_COD(ctx->map_scu[k]);
    for (i = 0; i < b->n_cu; i++)
        xevd_deblock_cu_ver(ctx, ctx->pic, b->x[i], b->y[i], 1 << b->log2w[i], 1 << b->log2h[i], 0);
    for (k = 0; k < (int)ctx->f_scu; k++) MCU_CLR_COD(ctx->map_scu[k]);
    for (i = 0; i < b->n_cu; i++)
        xevd_deblock_cu_hor(ctx, ctx->pic, b->x[i], b->y[i], 1 << b->log2w[i], 1 << b->log2h[i], 0);
    harness_free(hn);
    return 0;
}

/* ADDB: xevdm_deblock_cu_ver / _hor (src_main/xevdm_df.c:1137-1168) over the leaf CUs in decode order, vertical edges
   first (is_hor_edge = 0), COD cleared before each pass - the loop of src_main/xevdm.c:3152-3205 / deblock_tree :1935-2040 */
int refh_deblock_addb(const xgpu_seq_params *sp, const orc_frame *fr, const xgpu_cu_batch *b, orc_maps *m, int alpha_off, int beta_off)
{
    return deblock_main(sp, fr, b, m, 1, alpha_off, beta_off);
}

static int deblock_main(const xgpu_seq_params *sp, const orc_frame *fr, const xgpu_cu_batch *b, orc_maps *m, int addb, int alpha_off, int beta_off)
{
    harness *hn = harness_new(sp, fr, m, 0);
    XEVD_CTX *ctx = hn->ctx;
    XEVDM_CTX *mctx = (XEVDM_CTX *)ctx;
    const TREE_CONS tc = { FALSE, TREE_LC, eAll };
    u8 *map_ats = (u8 *)calloc(ctx->f_scu, 1);
    if (m->map_ats) memcpy(map_ats, m->map_ats, ctx->f_scu);
    int i, k;
    hn->sps.tool_addb = addb;
    ctx->pic->pic_deblock_alpha_offset = alpha_off;
    ctx->pic->pic_deblock_beta_offset = beta_off;
    (void)mctx;
    for (k = 0; k < (int)ctx->f_scu; k++) MCU_CLR_COD(ctx->map_scu[k]);
    for (i = 0; i < b->n_cu; i++) {      /* CUs above 64 go in as two halves, deblock_tree xevdm.c:2017-2037 */
        const int cw = 1 << b->log2w[i];
        int hx;
        for (hx = 0; hx < cw; hx += 64)
            xevdm_deblock_cu_ver(ctx, ctx->pic, b->x[i] + hx, b->y[i], cw > 64 ? 64 : cw, 1 << b->log2h[i], ctx->map_scu, ctx->map_refi, ctx->map_mv,
                                 ctx->w_scu, sp->log2_ctu, ctx->map_cu_mode, ctx->refp, 0, tc, ctx->map_tidx, 0, addb, map_ats,
                                 sp->bit_depth_luma, sp->bit_depth_chroma, sp->chroma_format_idc);
    }
    for (k = 0; k < (int)ctx->f_scu; k++) MCU_CLR_COD(ctx->map_scu[k]);
    for (i = 0; i < b->n_cu; i++) {
        const int ch = 1 << b->log2h[i];
        int hy;
        for (hy = 0; hy < ch; hy += 64)
            xevdm_deblock_cu_hor(ctx, ctx->pic, b->x[i], b->y[i] + hy, 1 << b->log2w[i], ch > 64 ? 64 : ch, ctx->map_scu, ctx->map_refi, ctx->map_mv,
                                 ctx->w_scu, sp->log2_ctu, ctx->refp, 0, tc, ctx->map_tidx, 0, addb, map_ats,
                                 sp->bit_depth_luma, sp->bit_depth_chroma, sp->chroma_format_idc);
    }
    free(map_ats);
    harness_free(hn);
    return 0;
}

/* ALF: the reference's alf_process_tile (src_main/xevdm_alf.c:901-1165: copy + extend, per-CTU halo buffer,
   alf_derive_classification, alf_filter_blk_7 / _5) on a single-tile picture, with the final coefficients
   (alf->coef_final, what alf_recon_coef :700-794 produces) supplied by the caller.  The argument record of
   alf_process_tile is private to xevdm_alf.c (:796-803); this mirrors its five fields for the call. */
typedef struct { ADAPTIVE_LOOP_FILTER *alf; CODING_STRUCTURE *cs; ALF_SLICE_PARAM *alf_slice_param; int tile_idx; int tsk_num; } alf_tile_arg;
extern int alf_process_tile(void *arg);

int refh_alf(const xgpu_seq_params *sp, const orc_pic *pic, const xgpu_alf_params *ap)
{
    XEVD_CTX *ctx = (XEVD_CTX *)calloc(1, sizeof(XEVDM_CTX));
    XEVD_SPS sps;
    XEVD_PIC p;
    XEVD_TILE tile;
    CODING_STRUCTURE cs;
    ALF_SLICE_PARAM *asp = (ALF_SLICE_PARAM *)calloc(1, sizeof(ALF_SLICE_PARAM));
    ADAPTIVE_LOOP_FILTER *alf = new_alf(sp->bit_depth_luma);
    const int ctu = 1 << sp->log2_ctu;
    const int w_lcu = (sp->width + ctu - 1) / ctu, h_lcu = (sp->height + ctu - 1) / ctu, f_lcu = w_lcu * h_lcu;
    u8 *flags = (u8 *)calloc(3 * f_lcu, 1);
    alf_tile_arg arg;
    int i, c;

    memset(&sps, 0, sizeof(sps)); memset(&tile, 0, sizeof(tile));
    sps.chroma_format_idc = sp->chroma_format_idc;
    sps.pic_width_in_luma_samples = sp->width; sps.pic_height_in_luma_samples = sp->height;
    ctx->sps = &sps;
    ctx->w = sp->width; ctx->h = sp->height;
    ctx->w_scu = sp->width >> 2; ctx->h_scu = sp->height >> 2;
    ctx->log2_max_cuwh = sp->log2_ctu; ctx->max_cuwh = ctu;
    ctx->w_lcu = w_lcu; ctx->h_lcu = h_lcu; ctx->f_lcu = f_lcu;
    ctx->pps.num_tile_columns_minus1 = 0;
    ctx->pps.loop_filter_across_tiles_enabled_flag = ap->across_tiles;
    tile.ctba_rs_first = 0; tile.w_ctb = w_lcu; tile.h_ctb = h_lcu; tile.f_ctb = f_lcu;
    ctx->tile = &tile;
    fill_pic(&p, pic, sp);
    cs.ctx = ctx; cs.pic = &p; cs.temp_stride = 0; cs.pic_stride = 0;

    xevd_alf_create(alf, sp->width, sp->height, ctu, ctu, 5, sp->chroma_format_idc, sp->bit_depth_luma);
    memcpy(alf->coef_final, ap->luma_coef, sizeof(short) * 25 * 13);
    memcpy(asp->chroma_coef, ap->chroma_coef, sizeof(short) * 7);
    for (c = 0; c < 3; c++) {
        asp->enable_flag[c] = ap->enable[c];
        for (i = 0; i < f_lcu; i++) flags[c * f_lcu + i] = ap->enable[c] ? ((c == 0 && ap->ctb_flag) ? ap->ctb_flag[i] : 1) : 0;
        alf->ctu_enable_flag[c] = flags + c * f_lcu;      /* alf_process :1181-1184 */
    }
    asp->alf_ctb_flag = flags;
    arg.alf = alf; arg.cs = &cs; arg.alf_slice_param = asp; arg.tile_idx = 0; arg.tsk_num = 0;
    alf_process_tile(&arg);

    xevd_alf_destroy(alf); delete_alf(alf);
    free(flags); free(asp); free(ctx);
    return 0;
}

void refh_pad(const xgpu_seq_params *sp, const orc_pic *p)
{
    XEVD_PIC pic;
    fill_pic(&pic, p, sp);
    xevd_picbuf_lc_expand(&pic, XGPU_PAD_L, XGPU_PAD_C);
}

/* DRA (the post-filter xevd_pull applies to a COPY of the picture, src_main/xevdm.c:3305-3385): the real LUT construction
   (xevd_init_dra, xevdm_dra.c:263-270) from signalled parameters, and the real sample processing (chroma planes first - they read
   the unmapped luma - then luma, :272-355) on tight 16-bit planes.  luts: [3][1024] = luma_inv_scale_lut, int_chroma_inv_scale_lut[0..1] */
int refh_dra(int bit_depth, int table_idx, int num_ranges, const int *in_ranges, const int *scale_values, int cb_scale, int cr_scale,
             int16_t *y, int16_t *u, int16_t *v, int w, int h, int32_t *luts)
{
    static DRA_CONTROL dc;
    XEVD_IMGB im;
    int i;
    /* the chroma QP mapping xevd_correct_local_chroma_scale reads: the Main sequence default, as sequence_init leaves it (xevdm.c:471-479) */
    xevd_set_chroma_qp_tbl_loc(bit_depth);
    for (i = 0; i < XEVD_MAX_QP_TABLE_SIZE; i++)
        xevd_qp_chroma_dynamic[0][i] = xevd_qp_chroma_dynamic[1][i] = xevd_tbl_qp_chroma_adjust_main[i];
    memset(&dc, 0, sizeof(dc));
    dc.signalled_dra.signal_dra_flag = 1;
    dc.signalled_dra.dra_table_idx = table_idx;
    dc.signalled_dra.num_ranges = num_ranges;
    dc.signalled_dra.dra_descriptor1 = 4; dc.signalled_dra.dra_descriptor2 = 9;
    dc.signalled_dra.dra_cb_scale_value = cb_scale; dc.signalled_dra.dra_cr_scale_value = cr_scale;
    for (i = 0; i <= num_ranges; i++) dc.signalled_dra.in_ranges[i] = in_ranges[i];
    for (i = 0; i < num_ranges; i++) dc.signalled_dra.dra_scale_value[i] = scale_values[i];
    xevd_init_dra(&dc, bit_depth);
    for (i = 0; i < DRA_LUT_MAXSIZE; i++) {
        luts[i] = dc.luma_inv_scale_lut[i];
        luts[DRA_LUT_MAXSIZE + i] = dc.int_chroma_inv_scale_lut[0][i];
        luts[2 * DRA_LUT_MAXSIZE + i] = dc.int_chroma_inv_scale_lut[1][i];
    }
    if (!y) return 0;
    memset(&im, 0, sizeof(im));
    im.np = 3;
    im.a[0] = y; im.a[1] = u; im.a[2] = v;
    for (i = 0; i < 3; i++) { im.w[i] = i ? w >> 1 : w; im.h[i] = i ? h >> 1 : h; im.s[i] = im.w[i] * 2; }
    xevd_apply_dra_chroma_plane(&im, &im, &dc, 1, TRUE);
    xevd_apply_dra_chroma_plane(&im, &im, &dc, 2, TRUE);
    xevd_apply_dra_luma_plane(&im, &im, &dc, 0, TRUE);
    return 0;
}
