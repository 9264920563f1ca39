/*
 * xevd_hip.h - C ABI of the MI355X (gfx950) per-CU reconstruction backend for an MPEG-5 EVC decoder.
 *
 * This is the drop-in boundary: plain C, plain pointers and sizes, no C++/torch types.  Every entry point
 * replaces one coarse slot of the reference decoder's function table (all citations are relative to the
 * reference tree mpeg5/xevd v0.7.0):
 *
 *   xgpu_open / xgpu_close        <- xevd_platform_init / xevd_platform_deinit  src_base/xevd.c:2074-2163
 *                                    (the `void *pf` "platform specific data" slot, src_base/xevd_def.h:1452-1470)
 *   xgpu_pic_alloc / xgpu_pic_free<- PICBUF_ALLOCATOR.fn_alloc / fn_free         src_base/xevd_def.h:684-705
 *   xgpu_frame_begin              <- slice_init + refp set-up                    src_base/xevd.c:378-405,1871-1903
 *   xgpu_batch_create/_recon      <- body of xevd_ctu_row_rec_mt -> xevd_recon_unit
 *                                                                                 src_base/xevd.c:1470-1526, 678-756
 *                                    (xevd_sub_block_itdq, xevd_mc, xevd_recon_yuv, xevd_set_dec_info)
 *   xgpu_deblock                  <- ctx->fn_deblock (xevd_deblock)              src_base/xevd.c:1116-1243,1909-1976
 *   xgpu_alf                      <- mctx->fn_alf (xevd_alf -> alf_process)      src_main/xevdm.c:2105, xevdm_alf.c:901-1249
 *   xgpu_pad                      <- ctx->fn_picbuf_expand                       src_base/xevd_util.c:365-427
 *   xgpu_pic_download             <- xevd_pull (picture hand-off)                src_base/xevd.c:2042-2071
 *   xgpu_pic_md5                  <- xevd_picbuf_signature / xevd_md5_imgb (picture signature)       src_base/xevd_util.c:985-1002, 1557-1572
 *   xgpu_pic_output               <- xevd_pull + the application's imgb_cpy_codec_to_out (crop fields xevd.c:2058-2069,
 *                                    bit-depth conversions app/xevd_app_util.h:441-552,656-700)
 *
 * plus fine-grained shims with the reference's per-block function-table signatures
 * (XEVD_MC_L / XEVD_MC_C src_base/xevd_mc.h:47-49, XEVD_ITXB src_base/xevd_def.h:360, fn_recon :1466)
 * so parity tests can drive one block exactly like the reference tables are driven: xgpu_test_*.
 *
 * Conventions (same as the reference's): every function returns XGPU_OK (0) or a negative error code with
 * the numeric values of XEVD_ERR_* (inc/xevd.h:48-77); nothing throws; no global state - one xgpu_ctx per
 * decoder instance, one HIP stream per ctx; a ctx is thread-compatible, not thread-safe.  All sample
 * storage is 16-bit (`pel` = s16, src_base/xevd_port.h:51) also for 8-bit streams.
 */
#ifndef XEVD_HIP_H
#define XEVD_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define XGPU_OK                    0
#define XGPU_ERR                  (-1)    /* XEVD_ERR                      */
#define XGPU_ERR_INVALID_ARGUMENT (-101)  /* XEVD_ERR_INVALID_ARGUMENT     */
#define XGPU_ERR_OUT_OF_MEMORY    (-102)  /* XEVD_ERR_OUT_OF_MEMORY        */
#define XGPU_ERR_UNSUPPORTED      (-104)  /* XEVD_ERR_UNSUPPORTED          */
#define XGPU_ERR_UNEXPECTED       (-105)  /* XEVD_ERR_UNEXPECTED (HIP errors map here) */

#define XGPU_MAX_REFS   17      /* XEVD_MAX_NUM_REF_PICS, src_base/xevd_def.h */
#define XGPU_PAD_L      144     /* PIC_PAD_SIZE_L = MAX_CU_SIZE + 16, src_base/xevd_def.h:211 */
#define XGPU_PAD_C      72      /* PIC_PAD_SIZE_C, :212 */

/* prediction modes, numeric values of MODE_* in src_base/xevd_def.h:284-287 */
#define XGPU_MODE_INTRA 0
#define XGPU_MODE_INTER 1
#define XGPU_MODE_SKIP  2
#define XGPU_MODE_DIR   3
#define XGPU_MODE_IBC   6      /* intra block copy (Main, sps->ibc_flag): mv[0] = whole-sample block vector into the current picture, refi unused */

typedef struct xgpu_ctx    xgpu_ctx;     /* one decoder instance on one GPU          */
typedef struct xgpu_dbatch xgpu_dbatch;  /* a CU batch resident in HBM               */

/* Sequence-level parameters: the SPS fields that change arithmetic on this path. */
typedef struct xgpu_seq_params {
    int device;               /* HIP device ordinal                                              */
    int width, height;        /* luma picture size in samples (ctx->w, ctx->h); multiples of 8  */
    int bit_depth_luma;       /* sps->bit_depth_luma_minus8 + 8   (8..12)                        */
    int bit_depth_chroma;     /* sps->bit_depth_chroma_minus8 + 8                                */
    int chroma_format_idc;    /* 1 (4:2:0) is implemented                                        */
    int log2_ctu;             /* ctx->log2_max_cuwh: 6 Baseline (xevd.c:249-252), 5..7 Main      */
    int tool_iqt;             /* sps->tool_iqt  : 16-bit two-stage transform + scale table ..72  */
    int tool_admvp;           /* sps->tool_admvp: Main 8-tap luma / 4-tap chroma tables          */
    int tool_addb;            /* sps->tool_addb : ADDB deblocking (else baseline filter)         */
    int tool_alf;             /* sps->tool_alf                                                   */
    int max_pics;             /* picture slots to make available (DPB + current), <= 34          */
    /* chroma QP mapping tables (xevd_qp_chroma_dynamic[0..1][-qpBdOffsetC .. 57], xevd_tbl.c:359-426);
       entry [c][qp + 6*(bit_depth_chroma-8)]; NULL = the sequence default: xevd_tbl_qp_chroma_adjust_base, or
       _main when tool_iqt is on (src_main/xevdm.c:471-479)                                            */
    const int8_t *chroma_qp_table[2];
    int tool_eipd;            /* sps->tool_eipd : 33 luma / 5 chroma intra modes, neighbour padding of xevdm_get_nbr; batch.ipm then
                                 holds core->ipm[0] (IPD_DC 0, PLN 1, BI 2, angular 3..32) and core->ipm[1] (DM 0, BI 1, DC 2, HOR 3, VER 4) */
} xgpu_seq_params;

/* Per-picture parameters: what slice_init / the slice header contribute to this path. */
typedef struct xgpu_frame_params {
    int pic;                               /* destination picture slot                                   */
    int poc;                               /* ctx->poc.poc_val                                           */
    int num_refp[2];                       /* ctx->dpm.num_refp[list]                                    */
    int refp_pic[XGPU_MAX_REFS][2];        /* picture slot of ctx->refp[idx][list].pic                   */
    int refp_poc[XGPU_MAX_REFS][2];        /* ctx->refp[idx][list].pic->poc                              */
    int qp_u_offset, qp_v_offset;          /* sh.qp_u_offset / sh.qp_v_offset (pic_qp_*_offset)          */
    int deblock_alpha_offset, deblock_beta_offset;  /* sh_deblock_alpha/beta_offset (ADDB only)          */
    /* which in-loop filters WILL run on this picture (sh.deblocking_filter_on, sh.alf_on).  The filters work
       out of place between the DPB slot and a private scratch picture; knowing the plan up front lets
       reconstruction start in the buffer from which the last filter lands in the DPB slot without a copy.  */
    int deblock_on, alf_on;
} xgpu_frame_params;

/* Tile grid of a picture: the PPS tile syntax after set_tile_info (src_main/xevdm.c:2162-2330).  Tiles are rectangles of CTUs; a CU
   sees no neighbour in another tile (intra prediction, HTDF border, motion candidates), and the in-loop filters treat tile borders
   as loop_filter_across_tiles says: the deblocking filters leave edges on a tile border alone when it is 0 (xevdm_df.c:142, 233,
   274), the ALF windows end at the tile - mirrored when 0, replicated when 1 (alf_process_tile, xevdm_alf.c:901-1160).            */
#define XGPU_MAX_TILE_COLS 20      /* MAX_NUM_TILES_COL / MAX_NUM_TILES_ROW, src_base/xevd_def.h */
#define XGPU_MAX_TILE_ROWS 22
typedef struct xgpu_tile_grid {
    int n_cols, n_rows;
    int col_bd[XGPU_MAX_TILE_COLS + 1];      /* first CTU column of every tile column; col_bd[n_cols] = CTUs per picture row */
    int row_bd[XGPU_MAX_TILE_ROWS + 1];
    int loop_filter_across_tiles;            /* pps.loop_filter_across_tiles_enabled_flag */
} xgpu_tile_grid;

/* Adaptive loop filter parameters of one picture: what alf_process has after alf_recon_coef
   (src_main/xevdm_alf.c:700-794, 1167-1195) - coefficient reconstruction from the APS stays on the host. */
typedef struct xgpu_alf_params {
    int            enable[3];     /* alf_slice_param.enable_flag[Y, U, V]                                    */
    const int16_t *luma_coef;     /* alf->coef_final: [25 classes][13] 7x7-diamond coefficients              */
    const int16_t *chroma_coef;   /* alf_slice_param.chroma_coef: [7] 5x5-diamond coefficients               */
    const uint8_t *ctb_flag;      /* alf_ctb_flag of the luma plane, [n_ctu] raster; NULL = every CTU on     */
    int            across_tiles;  /* pps.loop_filter_across_tiles_enabled_flag (changes right/bottom borders) */
    const xgpu_tile_grid *tiles;  /* NULL = one tile; else its loop_filter_across_tiles must equal across_tiles */
} xgpu_alf_params;

/*
 * One batch of decoded CUs (one tile or one picture), structure-of-arrays, in decode order, grouped by CTU.
 * The ORDER carries meaning: a neighbouring CU counts as reconstructed before CU i when its index is below i (xevd_get_avail_intra / xevdm_get_nbr / xevd_check_nev_avail read the
 * COD flags the reference sets CU by CU) - left, above, and with sps_suco_flag also right-hand neighbours.  Nothing else tells the backend about split order.
 * It must be the order of a walk over the split tree (any node's parts left to right or right to left): all neighbours along one side of a CU are then decoded before it or all
 * after it, which the reference's deblocking walk relies on (xevdm_df.c:237,280) and the backend's filters take for granted.
 * It is the post-entropy-decode record set of XEVD_CU_DATA (src_base/xevd_def.h:1145-1190) after MV
 * derivation (xevd.c:705-728), flattened per leaf CU.  Coefficients are stored CU-contiguous exactly as
 * coef_rect_to_series produces them (xevd.c:640-676): for a CU, the coded components in the order Y, U, V,
 * each `w*h` (luma) / `(w/2)*(h/2)` (chroma) s16 values, row-major; an un-coded component occupies nothing.
 */
typedef struct xgpu_cu_batch {
    int             n_cu;
    const uint16_t *x, *y;        /* [n_cu] top-left luma sample                                          */
    const uint8_t  *log2w, *log2h;/* [n_cu] 2..6 (7 with Main CTU 128)                                    */
    const uint8_t  *pred_mode;    /* [n_cu] XGPU_MODE_*; SKIP/DIR are inter CUs (MVs already derived)     */
    const int8_t   *refi;         /* [n_cu][2]  reference index per list, <0 = unused                     */
    const int16_t  *mv;           /* [n_cu][2][2] quarter-pel (list, x/y), unclipped                      */
    const uint8_t  *qp;           /* [n_cu][3]  core->qp_y/qp_u/qp_v: dequant QPs incl. 6*(bd-8)          */
    const uint8_t  *cbf;          /* [n_cu]  bit c set = component c has coefficients (is_coef[c])        */
    const uint16_t *cbf_sub;      /* [n_cu] or NULL: for CUs wider/taller than 64, bit (4*c + sb) = nnz_sub[c][sb]
                                     of the 64x64 sub-blocks sb = (j<<1)|i (xevd_itdq.c:544-621); NULL = every
                                     sub-block of a coded component is coded                                */
    const uint8_t  *ipm;          /* [n_cu][2] intra luma / chroma mode of intra CUs (core->ipm[0..1]): IPD_DC_B 0, HOR 1, VER 2,
                                     UL 3, UR 4 (src_base/xevd_def.h:332-342); NULL = DC                      */
    const uint8_t  *ats;          /* [n_cu] or NULL: bit 0 = ats_intra_cu, bit 1 = ats_intra_mode_v, bit 2 = ats_intra_mode_h
                                     (0 = DST-VII, 1 = DCT-VIII; luma TB of intra CUs, xevdm.c:602, xevdm_itdq.c:406-421)   */
    const uint8_t  *ats_inter;    /* [n_cu] or NULL: ats_inter_info of inter CUs = idx | pos << 4 (src_main/xevdm_def.h:232-236):
                                     idx 1/3 = left|right half/quarter-width TU, 2/4 = top|bottom half/quarter-height TU, pos 0 =
                                     first part coded, 1 = last part.  The CU's coefficient blocks then have the TU size
                                     (xevdm_get_tu_size, xevdm_util.c:3585-3608) for all three components          */
    const uint32_t *coef_off;     /* [n_cu]  offset (in s16 units) of the CU's first coefficient          */
    const int16_t  *coef;         /* [n_coef] coefficient arena                                           */
    size_t          n_coef;
    int             n_ctu;
    const uint32_t *ctu_cu_start; /* [n_ctu+1] first CU of every CTU in CTU decode order (raster; with tiles: tile by tile) */
    int             constrained_intra_pred;   /* pps.constrained_intra_pred_flag: intra CUs only predict from intra neighbours
                                                 (xevd_get_nbr_b, src_base/xevd_ipred.c:47,61,77)          */
    /* affine motion (Main, sps->tool_affine; xevdm_affine_mc, src_main/xevdm_mc.c:2606): both NULL = no affine CU in the batch */
    const uint8_t  *affine;       /* [n_cu] or NULL: 0 = translational, 2 / 3 = control points of an inter CU (mcore->affine_flag + 1);
                                     CUs of at least 8x8.  `mv` of such a CU is only stored for a list it does not use          */
    const int16_t  *affine_mv;    /* [n_cu][2][3][2] quarter-pel control-point vectors mcore->affine_mv[list][vertex][x/y]
                                     (top-left, top-right, bottom-left; the third ignored with 2 control points)              */
    const uint8_t  *dmvr;         /* [n_cu] or NULL: 1 = the CU's merge mode allows decoder-side motion vector refinement (sps->tool_dmvr and mcore->dmvr_enable:
                                     a skip CU without MMVD or a direct-mode CU, not affine; src_main/xevdm.c:1272-1288).  The backend applies the
                                     remaining conditions of xevdm_mc (two references at equal POC distances on either side of the picture, at least
                                     8x8, src_main/xevdm_mc.c:1895-1911) and refines per 16x16 sub-block (processDMVR :1647-1829).  The SCU map the
                                     deblocking filter reads keeps the UNREFINED vectors (map_unrefined_mv, xevdm.c:2009-2041); the refined ones
                                     come back through xgpu_batch_dmvr_mvs for the host's temporal motion prediction                              */
    int             htdf_slice_qp;/* 0 = no HTDF.  With sps->tool_htdf: ctx->sh.qp of the picture's slice - the Hadamard-domain filter
                                     (xevdm_htdf, src_main/xevdm_recon.c:153-385) then runs on the luma block of every intra CU and every
                                     inter CU with luma coefficients right after its reconstruction, reading one sample of border from
                                     the CUs reconstructed before it (xevdm.c:1381-1392; a slice QP up to 17 switches it off)          */
    const xgpu_tile_grid *tiles;  /* NULL = one tile.  Else the CUs come tile by tile (tiles in raster order, CTUs in raster order inside a
                                     tile - xevdm_dec_slice, src_main/xevdm.c:2614-2718) and neighbours in another tile are unavailable   */
    const uint8_t  *tree;         /* [n_cu] or NULL: local dual tree of Main streams with sps_btt_flag and tool_admvp (mode constraint eOnlyIntra below a split that
                                     would leave chroma blocks under 16 samples, src_main/xevdm.c:1775-1833): 0 = the CU has luma and chroma, 1 = luma only
                                     (TREE_L: intra or IBC; cbf bits 1-2 clear; writes the SCU maps as usual), 2 = chroma only (TREE_C: the split node's
                                     chroma block, intra, ipm[0] = the luma mode its DM refers to, cbf bit 0 clear; follows its luma CUs in decoding order,
                                     leaves the SCU maps alone; chroma edges are deblocked at ITS border, not at those of the luma CUs inside)          */
} xgpu_cu_batch;

/* ------------------------------------------------------------------ lifetime ---------------------- */
int  xgpu_open(const xgpu_seq_params *sp, xgpu_ctx **out);
void xgpu_close(xgpu_ctx *ctx);
int  xgpu_sync(xgpu_ctx *ctx);                         /* wait for everything enqueued on the ctx stream */
const char *xgpu_last_error(const xgpu_ctx *ctx);
const char *xgpu_version(void);

/* Pinned host memory for the arrays a batch points at (north_star: "batches decoded CUs per tile into pinned SoA buffers"): a coefficient
   arena inside such a range is sent to the device straight from the caller's buffer, everything else through the context's staging
   blocks.  The range must stay untouched until xgpu_batch_wait_upload() of the batch that points into it has returned.            */
int  xgpu_host_alloc(xgpu_ctx *ctx, size_t bytes, void **out);
void xgpu_host_free(xgpu_ctx *ctx, void *p);

/* ------------------------------------------------------------------ pictures ---------------------- */
int  xgpu_pic_alloc(xgpu_ctx *ctx);                    /* -> slot >= 0, or error                          */
int  xgpu_pic_free(xgpu_ctx *ctx, int pic);
/* planes point at the first ACTIVE sample (XEVD_PIC.y/u/v); strides in samples.  Upload does not pad.   */
int  xgpu_pic_upload(xgpu_ctx *ctx, int pic, const int16_t *y, int s_y, const int16_t *u, const int16_t *v, int s_c);
int  xgpu_pic_download(xgpu_ctx *ctx, int pic, int16_t *y, int s_y, int16_t *u, int16_t *v, int s_c);
/* The output side in one call: the active area minus a conformance-window crop (luma samples, even: sps picture_crop_*_offset
   as xevd_pull reports them), converted to out_bit_depth - 8: one byte per sample, (v + round) >> shift clipped to 255;
   below the coding depth: the same rounding shift clipped to the range, 16 bit; above: v << shift; equal: copy - and packed
   as Y, U, V planes back to back without row padding (the bytes imgb_write puts in a .yuv file).  Crop, conversion and
   packing run on the device; `dst` (host, >= xgpu_pic_output_size() bytes) receives one contiguous copy.  Blocking.
   `dra` (NULL = none): the DRA post-filter xevd_pull applies to its copy of the picture when sps->tool_dra and the PPS names a
   DRA parameter set (xevd_apply_filter, src_main/xevdm.c:3305-3349): the inverse-mapping tables of DRA_CONTROL after xevd_init_dra
   (the table construction stays host code), applied before the conversion - Cb and Cr scaled around 512 by a factor looked up
   with the UNMAPPED luma sample at (2y, 2x), then luma through its table (xevdm_dra.c:272-355); 4:2:0 at up to 10 bit. */
typedef struct xgpu_dra_luts {
    const int32_t *luma_inv_scale_lut;         /* [1024]  DRA_CONTROL.luma_inv_scale_lut          */
    const int32_t *chroma_inv_scale_lut[2];    /* [1024]  DRA_CONTROL.int_chroma_inv_scale_lut[c] */
} xgpu_dra_luts;
size_t xgpu_pic_output_size(const xgpu_ctx *ctx, int out_bit_depth, int crop_l, int crop_r, int crop_t, int crop_b);   /* 0: invalid */
int  xgpu_pic_output(xgpu_ctx *ctx, int pic, const xgpu_dra_luts *dra, int out_bit_depth, int crop_l, int crop_r, int crop_t, int crop_b,
                     void *dst, size_t dst_size);
/* The same in two halves, for a decode loop that overlaps the output of picture k with the kernels of picture k+1: _async queues the
   conversion behind the picture's kernels and the copy to `dst` (pinned memory for a truly asynchronous copy) on the context's download
   stream and returns a ticket; _wait blocks until `dst` holds the picture.  At most two outputs are in flight (a third call waits for the
   first); the picture slot may be decoded into again as soon as _async has returned.                                                   */
int  xgpu_pic_output_async(xgpu_ctx *ctx, int pic, const xgpu_dra_luts *dra, int out_bit_depth, int crop_l, int crop_r, int crop_t, int crop_b,
                           void *dst, size_t dst_size, int *ticket);
int  xgpu_pic_output_wait(xgpu_ctx *ctx, int ticket);
/* The picture signature on the device: the MD5 of every plane over its rows of width x 2 bytes of 16-bit samples (8-bit pictures too), as xevd_md5_imgb makes it
   (src_base/xevd_util.c:985-1002) and xevd_picbuf_check_signature compares it with the SEI (:1557-1572) - of the DRA-mapped picture when `dra` is given, which is
   what the Main decoder signs when the PPS names a DRA parameter set (src_main/xevdm.c:3256-3287).  digest[plane] = the 16 bytes of the SEI payload.  Blocking; the
   picture's own kernels need not have finished when it is called.  An MD5 is one serial chain per plane: the device walks the three chains in three lanes of one
   wave at ~80 MB/s each (measured: 48 ms for a 1080p picture, 0.22 s at 4K, 0.87 s at 8K; a host core hashes at ~700 MB/s) - it takes the hashing off a CPU-bound
   host, it does not make it faster. */
int  xgpu_pic_md5(xgpu_ctx *ctx, int pic, const xgpu_dra_luts *dra, uint8_t digest[3][16]);
/* whole padded buffers (XEVD_PIC.buf_y/u/v layout: stride = w + 2*pad, rows = h + 2*pad): for tests, and - luma alone, buf_u = buf_v = NULL - for a
   front end that refines merge vectors itself on the reference samples (xhost_parser_set_ref_luma, include/xevd_host.h).  Blocking.                 */
int  xgpu_pic_download_padded(xgpu_ctx *ctx, int pic, int16_t *buf_y, int16_t *buf_u, int16_t *buf_v);
int  xgpu_pic_upload_padded(xgpu_ctx *ctx, int pic, const int16_t *buf_y, const int16_t *buf_u, const int16_t *buf_v);

/* ------------------------------------------------------------------ per picture ------------------- */
int  xgpu_frame_begin(xgpu_ctx *ctx, const xgpu_frame_params *fp);
/* copy a batch into HBM (one pinned staging block + one async H2D copy) and build its device work lists.  The arrays behind `b`
   are read before the call returns.  Device and staging blocks come from a per-context pool: no allocation in steady state. */
int  xgpu_batch_create(xgpu_ctx *ctx, const xgpu_cu_batch *b, xgpu_dbatch **out);
/* xgpu_batch_create / xgpu_batch_destroy are the two entry points that may run on ANOTHER thread than the one driving the context (a builder
   thread preparing picture k+1 while picture k is being launched): the upload goes through the context's own upload stream, and
   xgpu_batch_recon makes the kernels wait for it.  Several threads may be inside xgpu_batch_create of ONE context at a time (pictures built side
   by side: examples/evc_decode --builders, bench.py's end-to-end leg); the staging-block pool is locked, every thread has its own scratch and
   worker pool.  xgpu_batch_wait_upload blocks until the batch's arrays have left host memory.                                             */
int  xgpu_batch_wait_upload(xgpu_ctx *ctx, xgpu_dbatch *db);
/* DMVR: the vectors the decoder stores for temporal prediction (map_mv / dmvr_mv, src_main/xevdm_mc.c:1783-1797, xevdm.c:1553-1563) after
   xgpu_batch_recon: for every CU of the batch with the dmvr flag, two references and at least 8x8 samples - in batch order, its 16x16
   sub-blocks in raster order - mv[list][x/y] in quarter samples: refined where the refinement ran, the CU's own otherwise.  `n` = capacity of
   `mv` in sub-blocks; returns the number of sub-blocks (also with mv = NULL), or a negative error.  Blocking.                                 */
int  xgpu_batch_dmvr_mvs(xgpu_ctx *ctx, xgpu_dbatch *db, int16_t *mv, int n);
/* Host threads xgpu_batch_create may spread its per-CU passes over (validation + counting, record / TB-list construction, the owner map); 1..64, default 1.
   The device arrays it builds do not depend on the count. */
int  xgpu_set_builder_threads(xgpu_ctx *ctx, int n);
/* what the batch builder made of the batch (measurement / diagnostics): info[0] CUs, [1] transform blocks, [2] work items of the transform kernel,
   [3] nodes of the order-dependent kernel (intra / IBC CUs, HTDF nodes), [4] of them without a node among their neighbours, [5] depth of the
   dependency graph (levels), [6] DMVR sub-blocks, [7] affine tiles.                                                                            */
#define XGPU_BATCH_INFO_COUNT 8
int  xgpu_batch_info(xgpu_ctx *ctx, const xgpu_dbatch *db, int info[XGPU_BATCH_INFO_COUNT]);
/* returns the batch's blocks to the pool.  Does not wait for the device: it may follow xgpu_batch_recon immediately (kernels already
   queued keep their data - later batches of this context are written through the same HIP stream, behind them). */
void xgpu_batch_destroy(xgpu_ctx *ctx, xgpu_dbatch *db);
/* Optional, for a caller that has the NEXT picture's batch at hand while the current one is being reconstructed: queues the batch's residual pass (dequant +
   inverse transform - it depends on nothing but the batch) on a side stream behind the k_inter launched last, so that it runs under the current picture's
   dependency kernel and filters; xgpu_batch_recon of that batch then only waits for it.  Call between xgpu_batch_recon of picture k and of picture k + 1.     */
int  xgpu_batch_prepare(xgpu_ctx *ctx, xgpu_dbatch *db);
/* dequant + inverse transform of every coded TB, then MC + residual add + clip of every inter CU, and the
   SCU map update (xevd_set_dec_info) the in-loop filters read.  Asynchronous on the ctx stream.          */
int  xgpu_batch_recon(xgpu_ctx *ctx, xgpu_dbatch *db);
/* xgpu_batch_recon of `db`, with the residual pass of the NEXT picture's batch (`next`, may be NULL; its upload may still be in flight) queued on the
   same stream: inside this picture's data-flow intra launch when it has one - that launch is a chain of memory round trips that leaves most of the GPU
   idle - or behind this picture's last kernel.  xgpu_batch_recon(_ahead) of `next` then starts with its MC kernel.  The preferred form of
   xgpu_batch_prepare: no second stream, no cross-stream event.                                                                                    */
int  xgpu_batch_recon_ahead(xgpu_ctx *ctx, xgpu_dbatch *db, xgpu_dbatch *next);
/* both deblocking passes over the current picture (vertical edges, then horizontal edges)               */
int  xgpu_deblock(xgpu_ctx *ctx);
/* adaptive loop filter: 4x4 block classification + 7x7 luma / 5x5 chroma diamond filters (ctx->fn_alf)    */
int  xgpu_alf(xgpu_ctx *ctx, const xgpu_alf_params *ap);
/* replicate the picture border into the 144/72-sample padding                                            */
int  xgpu_pad(xgpu_ctx *ctx);
int  xgpu_frame_end(xgpu_ctx *ctx);

/* ------------------------------------------------------------------ measurement ------------------- */
/* Kernel families timed with HIP events on the ctx stream (the stream the kernels are launched on).      */
enum { XGPU_K_ITDQ = 0, XGPU_K_INTER = 1, XGPU_K_DBK_V = 2, XGPU_K_DBK_H = 3, XGPU_K_PAD = 4, XGPU_K_INTRA = 5,
       XGPU_K_ALF = 6, XGPU_K_AFFINE = 7, XGPU_K_DMVR = 8, XGPU_K_COUNT = 9 };
int  xgpu_timing_enable(xgpu_ctx *ctx, int on);
int  xgpu_timing_reset(xgpu_ctx *ctx);
/* resolves pending events (synchronises) and returns accumulated milliseconds and launch counts          */
int  xgpu_timing_get(xgpu_ctx *ctx, double ms[XGPU_K_COUNT], long long launches[XGPU_K_COUNT]);
/* plain device-to-device copy bandwidth (bytes read + written per second) - the "measured roofline"      */
int  xgpu_measure_copy_bw(xgpu_ctx *ctx, size_t bytes, int iters, double *gbps);

/* ------------------------------------------------------------------ fine-grained test shims ------- */
/* Host buffers in, host buffers out, one block per call, reference function-table signatures.
   `ref` points INTO a host plane that has at least the filter margin around the addressed window.       */
int xgpu_test_mc_l(xgpu_ctx *ctx, const int16_t *ref_plane, int plane_w, int plane_h, int ref_x, int ref_y,
                   int has_dx, int has_dy, int gmv_x, int gmv_y, int16_t *pred, int w, int h, int bit_depth);
int xgpu_test_mc_c(xgpu_ctx *ctx, const int16_t *ref_plane, int plane_w, int plane_h, int ref_x, int ref_y,
                   int has_dx, int has_dy, int gmv_x, int gmv_y, int16_t *pred, int w, int h, int bit_depth);
/* residual arena of a batch after xgpu_batch_recon (what xevd_sub_block_itdq leaves in core->coef), n_coef s16   */
/* fn_recon (src_base/xevd_def.h:1466; xevd_recon, xevd_recon.c:35-71): rec[cuh][s_rec] = clip(pred + coef) or pred, through the kernels' residual add */
int xgpu_test_recon(xgpu_ctx *ctx, const int16_t *coef, const int16_t *pred, int is_coef, int cuw, int cuh, int s_rec, int16_t *rec, int bit_depth);
/* fn_dbk / fn_dbk_chroma (XEVD_DBK / XEVD_DBK_CH, src_base/xevd_def.h:363-364; deblock_scu_hor / _ver[_chroma], xevd_df.c:96-289): ONE 4-sample (chroma:
   2-sample) edge segment of a host plane of pw x ph samples through the kernels' line filters, in place; (x, y) = first sample on the far side of the
   edge, hor = the edge is horizontal, st = the strength from xevd_tbl_df_st (0 leaves a chroma plane untouched) */
int xgpu_test_dbk(xgpu_ctx *ctx, int16_t *plane, int pw, int ph, int x, int y, int st, int hor, int bit_depth);
int xgpu_test_dbk_chroma(xgpu_ctx *ctx, int16_t *u, int16_t *v, int pw, int ph, int x, int y, int st_u, int st_v, int hor, int bit_depth);
int xgpu_test_batch_resid(xgpu_ctx *ctx, xgpu_dbatch *db, int16_t *resid);
/* The host batch builder alone - no device, no HIP call (runs on a machine without a GPU): builds the staging block of `b` for a sequence `sp` on `threads`
   builder threads and returns an FNV-1a digest per array of the block (CU records, CTU starts, TB records, itdq work items, intra records, dependency lists,
   affine tiles, control points, DMVR sub-blocks, owner map, coefficients, and k_inter's two arrays: the items of the whole tiles and the roles per region), the counts of xgpu_batch_info and the builder's wall time in milliseconds.  The CPU
   suite pins the builder with it: the arrays do not depend on the thread count, and their digests are golden values. */
#define XGPU_TEST_BUILD_DIGESTS 13
int xgpu_test_build_batch(const xgpu_seq_params *sp, const xgpu_cu_batch *b, int threads, uint64_t digest[XGPU_TEST_BUILD_DIGESTS], int info[XGPU_BATCH_INFO_COUNT], double *ms);
/* dequant + 2-D inverse transform of n blocks of one size, in place (xevd_itdq, src_base/xevd_itdq.c:494) */
int xgpu_test_itdq(xgpu_ctx *ctx, int16_t *coef, int n_blocks, int log2w, int log2h, const uint8_t *qp, int bit_depth);

#ifdef __cplusplus
}
#endif
#endif /* XEVD_HIP_H */
