/*
 * xevd_oracle.h - CPU restatement of the reference's per-CU reconstruction path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under xevd_amd/ (the product) includes, links, loads or executes this;
 * only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg do, and only as the checker / the
 * timed CPU baseline.  Parity status: PINNED - every function here is checked against the reference itself
 * (oracle/_ref/libxevd_ref.so, built from /root/reference by oracle/Makefile.ref) in tests/test_oracle_vs_ref.py,
 * and the golden vectors under tests/golden/ were produced by the reference (tests/golden/make_golden.py).
 *
 * Each function cites the reference file:line whose arithmetic it restates (paths relative to the
 * mpeg5/xevd v0.7.0 tree).
 */
#ifndef XEVD_ORACLE_H
#define XEVD_ORACLE_H
#include <stdint.h>
#include "../include/xevd_hip.h"   /* shares the batch / parameter structs with the product ABI */

#ifdef __cplusplus
extern "C" {
#endif

/* a picture: pointers to the first ACTIVE sample of each plane (XEVD_PIC.y/u/v), strides in samples;
   the planes must carry the 144/72-sample padding around them (src_base/xevd_def.h:211-212) */
typedef struct orc_pic {
    int16_t *y, *u, *v;
    int      s_l, s_c;
    int      poc;
} orc_pic;

/* SCU maps written by reconstruction and read by the in-loop filters (ctx->map_scu / map_refi / map_mv,
   src_base/xevd_def.h:372-438, xevd_util.c:1574-1660) */
typedef struct orc_maps {
    uint32_t *map_scu;     /* [w_scu*h_scu] */
    int8_t   *map_refi;    /* [w_scu*h_scu][2] */
    int16_t  *map_mv;      /* [w_scu*h_scu][2][2] */
    uint8_t  *map_ats;     /* [w_scu*h_scu] mctx->map_ats_inter (src_main/xevdm_util.c:4321), may be NULL */
    int       w_scu, h_scu;
    const uint8_t *map_tidx; /* [w_scu*h_scu] ctx->map_tidx or NULL: orc_recon_batch_ex fills it in for its own use from batch->tiles */
} orc_maps;

typedef struct orc_frame {
    orc_pic cur;
    orc_pic refp[XGPU_MAX_REFS][2];     /* ctx->refp[idx][list] */
    int     qp_u_offset, qp_v_offset;
} orc_frame;

/* ---- block level (function-table surface) ---- */
/* xevd_mc_l_{00,n0,0n,nn}: src_base/xevd_mc.c:169-288; tables :80-98 / src_main/xevdm_mc.c:121-139 */
void orc_mc_l(const int16_t *ref, int gmv_x, int gmv_y, int s_ref, int s_pred, int16_t *pred, int w, int h,
              int bit_depth, int has_dx, int has_dy, int admvp);
/* xevd_mc_c_{00,n0,0n,nn}: src_base/xevd_mc.c:290-408; tables :100-134 / src_main/xevdm_mc.c:141-175 */
void orc_mc_c(const int16_t *ref, int gmv_x, int gmv_y, int s_ref, int s_pred, int16_t *pred, int w, int h,
              int bit_depth, int has_dx, int has_dy, int admvp);
/* xevd_mc: src_base/xevd_mc.c:435-557 (MV clip, variant select, identical-motion early-out, bi average).
   pred[list][comp] are contiguous w*h / (w/2)*(h/2) buffers of capacity 128*128. Returns number of lists predicted. */
int  orc_mc_cu(const xgpu_seq_params *sp, const orc_frame *fr, int x, int y, int w, int h,
               const int8_t refi[2], const int16_t mv[2][2], int16_t *pred0[3], int16_t *pred1[3]);
/* xevdm_mc with apply_DMVR: src_main/xevdm_mc.c:1860-2038, processDMVR :1647-1829.  1 = refined and averaged into pred0 (refined[k][list][x/y]:
   quarter-sample vectors of the 16x16 sub-blocks), 0 = the conditions failed, nothing predicted */
int  orc_dmvr_cu(const xgpu_seq_params *sp, const orc_frame *fr, int x, int y, int w, int h, const int8_t refi[2], const int16_t mv[2][2],
                 int16_t *pred0[3], int16_t *pred1[3], int16_t (*refined)[2][2]);
/* xevdm_affine_mc: src_main/xevdm_mc.c:2606-2685 (sub-block size / EIF decision xevdm_util.c:1870-2125; EIF :2108-2150, :2393-2604).
   mv[list][vertex][x/y] quarter-pel control points, vn = 2 or 3.  Same pred layout as orc_mc_cu. */
int  orc_affine_mc_cu(const xgpu_seq_params *sp, const orc_frame *fr, int x, int y, int log2w, int log2h, const int8_t refi[2],
                      const int16_t mv[2][3][2], int vn, int16_t *pred0[3], int16_t *pred1[3]);
/* xevd_dquant + xevd_itrans (xevd_itdq): src_base/xevd_itdq.c:473-542; IQT variant src_main/xevdm_itdq.c:708-788 */
void orc_itdq(int16_t *coef, int log2w, int log2h, int qp, int bit_depth, int iqt);
/* ATS (Main, intra CUs): dequant + DST-VII / DCT-VIII 2-D inverse transform, src_main/xevdm_itdq.c:81-421, 732-788.
   tr_v / tr_h: 0 = DST-VII, 1 = DCT-VIII for the vertical / horizontal stage (xevd_tbl_tr_subset_intra) */
void orc_itdq_ats(int16_t *coef, int log2w, int log2h, int qp, int bit_depth, int iqt, int tr_v, int tr_h);
const int16_t *orc_ats_tm(int type, int log2n);   /* xevd_tbl_tr{4,8,16,32}[DCT8=0|DST7=1] as the init code builds them */
/* 1-D stages exposed for table-level checks: xevd_itx_pb*b (step 0 / step 1), src_base/xevd_itdq.c:48-461 */
void orc_itx_pass0(const int16_t *src, int32_t *dst, int log2n, int line);
void orc_itx_pass1(const int32_t *src, int16_t *dst, int log2n, int line, int shift);
const int8_t *orc_tm(int log2n);  /* xevd_tbl_tm2..64, src_base/xevd_tbl.c:89-243 (generated by formula) */
/* xevd_recon: src_base/xevd_recon.c:35-71 */
void orc_recon(const int16_t *coef, const int16_t *pred, int is_coef, int cuw, int cuh, int s_rec, int16_t *rec, int bit_depth);
/* deblock_scu_{ver,hor}[_chroma]: src_base/xevd_df.c:96-289 */
void orc_dbk_luma(int16_t *buf, int st, int stride, int bit_depth, int is_ver);
void orc_dbk_chroma(int16_t *u, int16_t *v, int st_u, int st_v, int stride, int bit_depth, int is_ver);

/* ---- picture level ---- */
/* every inter CU of the batch: itdq -> mc -> recon -> set_dec_info  (xevd_recon_unit, src_base/xevd.c:678-756);
   intra CUs only update the maps.  `resid_out` (optional, n_coef s16) receives the residual arena. */
int  orc_recon_batch(const xgpu_seq_params *sp, const orc_frame *fr, const xgpu_cu_batch *b, orc_maps *maps, int16_t *resid_out);
int  orc_recon_batch_ex(const xgpu_seq_params *sp, const orc_frame *fr, const xgpu_cu_batch *b, orc_maps *maps, int16_t *resid_out, int16_t *dmvr_mv_out);
/* both baseline deblocking passes in the reference's order (src_base/xevd.c:1116-1243,1909-1976; xevd_df.c:291-546) */
int  orc_deblock_baseline(const xgpu_seq_params *sp, const orc_frame *fr, const xgpu_cu_batch *b, orc_maps *maps);
/* ADDB deblocking, both passes (src_main/xevdm_df.c:361-1135, driver src_main/xevdm.c:1935-2103, 3142-3205) */
int  orc_deblock_addb(const xgpu_seq_params *sp, const orc_frame *fr, const xgpu_cu_batch *b, orc_maps *maps, int alpha_off, int beta_off);
/* adaptive loop filter of one picture, in place (alf_process_tile + classification + filters, src_main/xevdm_alf.c:38-429, 901-1165) */
int  orc_alf(const xgpu_seq_params *sp, const orc_pic *pic, const xgpu_alf_params *ap);
/* picbuf_expand: src_base/xevd_util.c:365-427 */
void orc_pad(const xgpu_seq_params *sp, const orc_pic *p);
/* Output conversion of one plane, the application's imgb_cpy_codec_to_out (app/xevd_app_util.h:665-708 with :464-552):
   to 8 bit: bytes, (v + round) >> shift clipped to [0,255]; to a lower depth: the same clipped to the range, 16 bit;
   to a higher depth: v << shift; equal: copy.  src stride in samples; dst rows are tight. */
/* DRA sample processing in place on tight planes (xevd_apply_dra_chroma_plane / _luma_plane with backward_map,
   src_main/xevdm_dra.c:272-355, in xevd_apply_filter's order xevdm.c:3342-3344: Cb, Cr - reading the UNMAPPED luma at (2j, 2k) -
   then luma).  luts: luma_inv_scale_lut[1024], int_chroma_inv_scale_lut[2][1024] of DRA_CONTROL after xevd_init_dra. */
void orc_dra_apply(int16_t *y, int16_t *u, int16_t *v, int w, int h, const int32_t *luma_inv, const int32_t *cb_inv, const int32_t *cr_inv);
void orc_output_convert(const int16_t *src, int stride, int w, int h, int src_bd, int dst_bd, void *dst);
/* default chroma QP mapping table (static table of xevd_tbl.c:334-357 after xevd_tbl_derived_chroma_qp_mapping_tables, :364-426) */
const int8_t *orc_default_chroma_qp_table(void);   /* 58 entries for qp 0..57 (8-bit) */

#ifdef __cplusplus
}
#endif
#endif
