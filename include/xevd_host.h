/*
 * xevd_host.h - C ABI of the host front end that sits before the GPU path: an MPEG-5 EVC (Baseline profile) bitstream
 * parser that turns `.evc` NAL units into the CU batches of include/xevd_hip.h, and a bitstream writer for synthetic
 * CU batches (test/bench streams the reference decoder can decode too).
 *
 * What it mirrors in the reference (mpeg5/xevd v0.7.0), re-designed around the batch hand-over:
 *   parser : xevd_dec_nalu (src_base/xevd.c:1786-2024): xevd_eco_nalu/sps/pps/sh (xevd_eco.c:1178-1581), POC derivation
 *            (xevd_util.c:429-467), reference list set-up without RPL (xevd_picman.c:68-110,291-437), tile entropy decoding
 *            (xevd_tile_eco xevd.c:1408-1468, xevd_eco_cu xevd_eco.c:1048-1176, coefficients :343-395, SBAC :35-165),
 *            motion derivation (xevd.c:476-566, xevd_util.c:469-566,632-687), QP derivation (xevd_eco.c:640-668).
 *   writer : the inverse syntax (this tree has no encoder; the arithmetic encoder is the mirror of xevd_sbac_decode_bin).
 * Wire format: 4-byte big-endian NAL length prefix (XEVD_NAL_UNIT_LENGTH_BYTE, inc/xevd.h:133; app/xevd_app.c:52-107),
 * 2-byte NAL header (xevd_eco.c:1178-1209).
 *
 * Scope: Baseline profile, and Main-profile streams with any of - sps->tool_iqt, tool_ats, tool_addb (syntax of src_main/xevdm_eco.c: SPS :1847-2004,
 * slice header :2510-2800, ATS flags :128-190,354-393,902-934), tool_eipd (intra mode syntax src_base/xevd_eco.c:842-910, most-probable-mode lists
 * src_main/xevdm_ipred.c:320-767), ibc_flag (ibc_flag + block vector per CU, xevdm_eco.c:1401-1438,1789-1800), tool_htdf, tool_dra (DRA APS NAL units
 * :2319-2375, PPS switch :2054-2060, table construction src_main/xevdm_dra.c:39-270), tool_alf (APS NAL units :2082-2135,2376-2477 with the fixed
 * filter sets, coefficient syntax :2154-2318, slice-level parameters :2479-2657, per-CTU flags src_main/xevdm.c:2411-2418, alf_recon_coef
 * src_main/xevdm_alf.c:700-794), and tool_admvp with its sub-tools tool_amvr, tool_hmvp, tool_mmvd and tool_dmvr: merge_idx / merge_mode_flag / mvr_idx /
 * bi_idx / mmvd_flag + mmvd data syntax (xevdm_eco.c:767-812,1519-1726), merge with vector difference (src_main/xevdm_util.c:191-592,4682-4716), merge candidates incl. the temporal and history ones (src_main/xevdm_util.c:594-1391,3729-3818), the resolution-indexed
 * predictor (:750-951), intra-only 4x4 CUs; tool_dmvr needs the backend's refined vectors back per picture (xhost_parser_set_dmvr_mvs) - or, together with
 * tool_hmvp or tool_mmvd, the decoded luma samples of the reference pictures (xhost_parser_set_ref_luma: the front end then refines itself).  tool_affine (affine merge / affine inter CUs, xevdm_util.c:2145-3187) is parsed too.
 * sps_btt_flag (binary / ternary split trees with CTU 64, "inter only" mode constraints, local dual trees: xgpu_cu_batch.tree) and sps_suco_flag (parts of a split coded right to left, right-hand neighbours) are parsed.
 * tool_cm_init (context initialisation tables, neighbour-dependent contexts) and tool_adcc
 * (advanced coefficient coding) are parsed.
 * dquant_flag (QP deltas per quantisation group), tool_rpl (reference picture lists in SPS / slice headers, RPL-based marking) and tool_pocs
 * (POC from poc_lsb) are parsed.  SPS chroma QP mapping tables and cropping offsets are parsed, a VUI is skipped; 4:2:0, one slice per picture (with all of its
 * tiles - uniform or explicit PPS tile grids, entry points in the slice header; explicit tile ids and arbitrary slices are refused), I / P / B slices incl. temporal layers (hierarchical sub-GOPs).
 * Conventions as xevd_hip.h: 0 / negative XEVD_ERR_* codes, nothing throws, one object per stream.
 */
#ifndef XEVD_HOST_H
#define XEVD_HOST_H

#include "xevd_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

#define XHOST_ERR_MALFORMED (-202)      /* XEVD_ERR_MALFORMED_BITSTREAM */

#define XHOST_SLICE_B 0                 /* XEVD_ST_B / _P / _I, inc/xevd.h:181-183 */
#define XHOST_SLICE_P 1
#define XHOST_SLICE_I 2

typedef struct xhost_parser xhost_parser;
typedef struct xhost_writer xhost_writer;

/* One decoded-syntax picture: everything xgpu_frame_begin + xgpu_batch_create need, plus DPB bookkeeping by POC.
   The arrays behind `batch`, `alf` and `dra_lut` belong to the parser and stay valid until the next xhost_parser_next / close - or, after
   xhost_parser_set_depth(p, d), until the call that hands out the d-th picture after this one. */
typedef struct xhost_picture {
    int width, height, bit_depth_luma, bit_depth_chroma;
    int poc, temporal_id, slice_type, is_idr;
    int is_ref;                            /* the picture stays in the DPB as a reference (ctx->slice_ref_flag)         */
    int num_refp[2];
    int refp_poc[XGPU_MAX_REFS][2];        /* POC of ctx->refp[idx][list]: the caller maps POC -> picture slot           */
    int slice_qp, qp_u_offset, qp_v_offset;
    int deblock_on;
    int profile_main;                      /* 1: Main-profile stream (sps->profile_idc)                                   */
    int tool_iqt, tool_ats, tool_addb;     /* sps->tool_* flags that change arithmetic on the GPU path                      */
    int deblock_alpha_offset, deblock_beta_offset;      /* sh.sh_deblock_alpha/beta_offset (ADDB)                          */
    int tool_alf;
    int tool_eipd;                         /* sps->tool_eipd: batch.ipm holds Main mode numbers, xgpu_seq_params.tool_eipd must be set      */
    int tool_admvp;                        /* sps->tool_admvp: xgpu_seq_params.tool_admvp must be set (Main interpolation tables)            */
    int crop[4];                           /* sps picture_crop_left / right / top / bottom_offset, what xevd_pull puts in the XEVD_IMGB      */
    const int8_t *chroma_qp_table[2];      /* SPS chroma QP mapping tables in xgpu_seq_params.chroma_qp_table layout, NULL = sequence default */
    const int32_t *dra_lut[3];             /* sps->tool_dra + pps.pic_dra_enabled_flag: the tables xgpu_pic_output takes (xgpu_dra_luts: luma, Cb, Cr; 1024
                                              entries each), built as xevd_init_dra does from the DRA parameter set; NULL: no DRA for this picture  */
    int alf_on;                            /* sh.alf_on: `alf` below is what xgpu_alf takes (final coefficients, CTB flags)      */
    xgpu_alf_params alf;
    int has_md5;                           /* a picture-signature SEI follows the slice: MD5 of every plane's 16-bit samples    */
    uint8_t md5[3][16];                    /*   (xevd_eco_sei xevd_eco.c:1617-1678, xevd_md5_imgb xevd_util.c:985-1002)          */
    int n_dmvr_sub;                        /* sps->tool_dmvr: the number of sub-blocks xgpu_batch_dmvr_mvs reports for this picture's batch (batch.dmvr flags the
                                              merge-mode CUs); their vectors must come back through xhost_parser_set_dmvr_mvs before the next picture is parsed */
    int needs_ref_luma;                    /* sps->tool_dmvr together with tool_hmvp or tool_mmvd, and this picture is kept as a reference: the front end refines merge-mode
                                              vectors itself while it parses LATER pictures (their syntax depends on the refined vectors) and reads this picture's
                                              decoded luma samples for that - register them with xhost_parser_set_ref_luma before the next xhost_parser_next */
    int n_release;                         /* reference pictures unmarked before this one was stored (pic_marking_no_rpl) */
    int release_poc[32];
    xgpu_cu_batch batch;
} xhost_picture;

xhost_parser *xhost_parser_open(const uint8_t *bytes, size_t size);
/* The same parser object on another independent byte string (parameter sets + closed GOPs, e.g. the next unit of a work queue): every bit of stream state
   starts afresh exactly as in a new parser, but the object keeps its memory (picture maps, motion fields, tile batches, the arenas of xhost_parser_set_arena)
   and its tile threads - a new parser per GOP pays for ~300 MB of fresh pages at 8K.  Pictures handed out before are invalid afterwards.  0 or < 0. */
int  xhost_parser_rebind(xhost_parser *p, const uint8_t *bytes, size_t size);
/* 1: `out` holds the next picture in decoding order; 0: end of stream; < 0: error */
int  xhost_parser_next(xhost_parser *p, xhost_picture *out);
/* sps->tool_dmvr: the vectors xgpu_batch_dmvr_mvs returned for the picture the parser handed out last ([n_sub][list][x/y], n_sub = its n_dmvr_sub): the
   parser stores them with the picture, where the temporal merge candidates of later pictures read them (the reference's map_mv holds the REFINED
   vectors, src_main/xevdm_util.c:4327-4338).  0, or < 0 on a count mismatch.  Not needed for pictures with n_dmvr_sub == 0. */
int  xhost_parser_set_dmvr_mvs(xhost_parser *p, const int16_t *mv, int n_sub);
/* The decoded luma plane of the picture with POC `poc` (handed out with needs_ref_luma): `plane` = sample (0, 0), rows `stride` samples apart, with at least
   144 samples of replicated border on every side (what xgpu_pic_download_padded delivers); it must stay valid and unchanged while the picture is a
   reference.  Streams that need it and do not get it fail with an error that says so.  Why the front end reads samples at all: xevd_amd/host/dmvr_search.h. */
int  xhost_parser_set_ref_luma(xhost_parser *p, int poc, const int16_t *plane, int stride);
/* The same hand-over without parking the parser behind every reference picture: after xhost_parser_set_ref_luma_wait(p, 1) (before the first picture)
   xhost_parser_set_ref_luma may be called from ANOTHER thread while xhost_parser_next runs on later pictures - in the order the pictures were handed out - and
   a CU whose refinement search needs a plane that has not been registered yet waits for it inside xhost_parser_next.  The caller must therefore register the
   plane of every picture handed out with needs_ref_luma, or call xhost_parser_cancel_wait (any thread): a waiting xhost_parser_next then fails with
   XHOST_ERR_MALFORMED, as does every later one until xhost_parser_rebind. */
int  xhost_parser_set_ref_luma_wait(xhost_parser *p, int on);
void xhost_parser_cancel_wait(xhost_parser *p);
/* The refinement search of decoder-side motion vector refinement on its own (xevd_amd/host/dmvr_search.h: what processDMVR decides, src_main/xevdm_mc.c:1647-1829), for a
   binding whose OWN parser derives candidates CU by CU from refined vectors (sps->tool_dmvr with tool_hmvp / tool_mmvd: oracle/ref_binding.c is the worked example).  The CU
   (x, y, w x h) with the unrefined quarter-sample vectors mv = { l0x, l0y, l1x, l1y } between two references at equal POC distances on either side; ref0 / ref1 = luma sample
   (0, 0) of the reference planes, each with at least 144 samples of replicated border.  refined[k * 4 ..] = the vectors of sub-block k (16x16, or the CU when smaller; raster
   order inside the CU) = mcore->dmvr_mv.  Returns the number of sub-blocks. */
int  xhost_dmvr_search(int pic_w, int pic_h, int bit_depth, int x, int y, int w, int h, const int16_t mv[4],
                       const int16_t *ref0, int stride0, const int16_t *ref1, int stride1, int16_t *refined);
/* Picture boundaries of a NAL stream without decoding it (what the GOP splitter xwq_split_gops uses): feed every NAL unit (without its length prefix);
   -> 0 not a slice, 1 the first slice of a picture, 2 a further slice of the same picture (several slices per picture), < 0 malformed.               */
typedef struct xhost_scan xhost_scan;
xhost_scan *xhost_scan_open(void);
int  xhost_scan_nal(xhost_scan *s, const uint8_t *nal, size_t size);
void xhost_scan_close(xhost_scan *s);

/* The same parser fed one NAL unit at a time (2-byte NAL header + payload, no length prefix - what xevd_decode receives):
   1: `out` holds a picture (has_md5 is 0: a signature SEI arrives as its own NAL unit); 0: consumed; < 0: error */
xhost_parser *xhost_parser_open_nal(void);
int  xhost_parser_nal(xhost_parser *p, const uint8_t *nal, size_t size, xhost_picture *out);
const char *xhost_parser_error(const xhost_parser *p);
/* Host threads the parser may use for ONE picture (default 1): the tiles of a picture are independent arithmetic-coder runs and are parsed
   in parallel, tile by tile off a shared counter - what xevdm_dec_slice does with the reference's thread pool (src_main/xevdm.c:2640-2690).
   Pictures with one tile are not affected.  The batch handed out is the same for every thread count. */
int  xhost_parser_set_threads(xhost_parser *p, int n_threads);
/* How many handed-out pictures stay valid at once (1..8, default 1; before the first picture).  With 2, picture k can be turned into a device
   batch and launched on one thread while another thread is inside xhost_parser_next for picture k + 1 (examples/evc_decode.c): the stages
   xevd_dec_nalu runs back to back per picture - entropy decoding, then reconstruction - overlap across pictures.  The parser itself is not
   re-entrant: one thread at a time inside its functions.  (Not for streams with sps->tool_dmvr feedback pending: xhost_parser_set_dmvr_mvs of
   picture k has to come before xhost_parser_next for k + 1 - the caller serialises those.) */
int  xhost_parser_set_depth(xhost_parser *p, int depth);
/* Where the coefficient arena of a picture with several tiles is gathered (before the first picture): memory from `alloc(user, bytes)` - e.g. pinned
   host memory of the backend (xgpu_host_alloc), so that xgpu_batch_create sends the largest array of a batch from where the parser wrote it, without
   a staging copy - one arena per picture slot of xhost_parser_set_depth, grown on demand, given back through `release(user, p)` when it is outgrown or
   the parser is closed.  `alloc` may return NULL (nothing available yet): that picture's coefficients then live in the parser's own memory.  The caller
   keeps an arena untouched until the backend has consumed it (xgpu_batch_wait_upload) before it lets the parser reuse the picture slot. */
int  xhost_parser_set_arena(xhost_parser *p, void *(*alloc)(void *user, size_t bytes), void (*release)(void *user, void *mem), void *user);
void xhost_parser_close(xhost_parser *p);

typedef struct xhost_stream_params {
    int width, height;                     /* multiples of 8                                                        */
    int bit_depth;                         /* luma = chroma                                                         */
    int max_num_ref_pics;                  /* sps->max_num_ref_pics                                                 */
    int log2_sub_gop_length;               /* sps->log2_sub_gop_length: 0 = low delay; n = hierarchical sub-GOPs of 2^n pictures
                                              (decoding order per sub-GOP: temporal ids 0, 1, 2, 2, 3, 3, 3, 3, ...)  */
    int qp_u_offset, qp_v_offset;          /* sh.qp_u_offset / qp_v_offset                                          */
    int deblock_on;                        /* sh.deblocking_filter_on                                               */
    int cu_qp_delta;                       /* pps.cu_qp_delta_enabled_flag: per-CU QPs of coded CUs are transmitted  */
    int profile_main;                      /* 1: Main profile (needed for any of the tools below)                    */
    int tool_iqt, tool_ats, tool_addb;     /* sps->tool_iqt / tool_ats (needs iqt) / tool_addb                       */
    int deblock_alpha_offset, deblock_beta_offset;      /* slice-level ADDB offsets                                   */
    int tool_alf;                          /* sps->tool_alf                                                          */
    int tool_eipd;                         /* sps->tool_eipd: ipm[0] = luma mode 0..32, ipm[1] = chroma mode 0..4 (a chroma mode equal to what DM
                                              stands for is written as DM)                                            */
    int crop[4];                           /* picture cropping offsets left / right / top / bottom (all 0: no cropping)  */
    int tool_dra, dra_aps_id;              /* sps->tool_dra; the PPS then switches DRA on for every picture with this parameter set id */
    /* chroma_qp_table_struct of the SPS (xevd_eco.c:1361-1376): pivot points of the chroma QP mapping */
    int cqt_present, cqt_same, cqt_global_offset;
    int cqt_num_points[2];                 /* 1..16 per table                                                        */
    int cqt_delta_in[2][16];               /* delta_qp_in_val_minus1 (6 bits)                                         */
    int cqt_delta_out[2][16];              /* delta_qp_out_val                                                       */
    int tool_htdf;                         /* sps->tool_htdf: no CU syntax of its own; the parser hands the slice QP to the backend (batch.htdf_slice_qp) */
    int tool_admvp;                        /* sps->tool_admvp: skip and merge-mode CUs take one of up to six merge candidates, explicitly coded motion uses the
                                              resolution-indexed predictor and bi_idx (xevdm_eco.c:1519-1726); the backend then interpolates with the Main 8-tap tables */
    int tool_mmvd;                         /* sub-tool of tool_admvp: sps->tool_mmvd - a share of the skip / merge-mode CUs is written as merge with vector difference */
    int tool_dmvr;                         /* sub-tool of tool_admvp: sps->tool_dmvr - skip and merge-mode CUs are flagged for decoder-side refinement */
    int tool_amvr, tool_hmvp;              /* sub-tools of tool_admvp: sps->tool_amvr (mvr_idx: vector differences on a half / 1 / 2 / 4 sample grid, predictor position
                                              coupled with the index), sps->tool_hmvp (history-based merge candidates and fallback predictors)  */
    int ibc_log_max_size;                  /* 0: sps->ibc_flag off.  2..7 (needs tool_eipd): intra block copy for CUs up to 2^n samples - a CU of the batch with
                                              pred_mode XGPU_MODE_IBC is written with ibc_flag and its block vector mv[0] (xevdm_eco.c:1401-1438, 1789-1800)  */
    /* tiles (Main profile; PPS syntax xevdm_eco.c:2019-2052): tile_cols x tile_rows tiles per picture (0 / 1: one tile), one slice with all of them.
       tile_col_w[0] == 0: uniform spacing; else the widths / heights in CTUs of all but the last column / row.  The batch handed to
       xhost_writer_add_picture stays in raster CTU order; the writer codes the CTUs tile by tile.  No IBC CUs with tiles (their block
       vectors would have to respect the tile decode order). */
    int tile_cols, tile_rows;
    int tile_col_w[XGPU_MAX_TILE_COLS], tile_row_h[XGPU_MAX_TILE_ROWS];
    int loop_filter_across_tiles;          /* pps.loop_filter_across_tiles_enabled_flag */
    int tool_affine;                       /* sub-tool of tool_admvp: sps->tool_affine - a CU of the batch with `affine` 2 / 3 is written as an affine CU: skip / merge-mode
                                              CUs of 8x8 and larger take one of the five affine merge candidates (the batch's control points are not used), inter CUs
                                              of 16x16 and larger code their control-point vectors `affine_mv` against one of two predictors (xevdm_eco.c:1528-1537, 1649-1682) */
    int cu_qp_delta_area;                  /* Main with cu_qp_delta: 0 = a QP delta per coded CU; 6..13 = sps->dquant_flag with quantisation groups of 2^n samples
                                              (pps.cu_qp_delta_area; 6 = 8x8): one delta per group (xevdm.c:1739-1759, xevdm_eco.c:882-897)                    */
    int tool_rpl, tool_pocs;               /* Main: sps->tool_rpl - every slice header carries reference picture lists (leading entries = the lists the sub-GOP scheme
                                              builds, tail = the other pictures that scheme still keeps) and the list sizes; sps->tool_pocs - poc_lsb (8 bits) per slice */
    int tool_cm_init;                      /* Main: sps->tool_cm_init - context variables start from the standard's tables (slice kind, QP); skip / pred_mode / ibc /
                                              affine flags, the run / level pair and the ATS-inter flags pick their contexts from neighbours, levels and CU shapes */
    int tool_adcc;                         /* Main, switches tool_cm_init on: sps->tool_adcc - coefficient blocks as last position + significance / greater-than flags /
                                              Golomb-Rice remainders per group of 16 (xevdm_eco_adcc) instead of run-level pairs */
    int btt;                               /* Main: sps_btt_flag (CTU 64) - the batch's CUs are the leaves of a binary / ternary split tree (non-square CUs 4x8 .. 64x16);
                                              the writer finds the tree from the leaves.  Limits of the tree: smallest CU side 2^btt_log2_min_cb (2..6), and the SPS fields
                                              log2_diff_ctu_max_14_cb_size / log2_diff_ctu_max_tt_cb_size / log2_diff_min_cb_min_tt_cb_size_minus2 (xevdm_util.c:4393-4400) */
    int btt_log2_min_cb, btt_diff_max_14, btt_diff_max_tt, btt_diff_min_tt;
    int rpl_in_sps;                        /* with tool_rpl, low delay and at least 2 references: RPL candidates in the SPS, picked by index where they match */
    int suco;                              /* Main: sps_suco_flag - split nodes with a vertical cut may code their parts right to left (the writer lets about half of the
                                              nodes that may choose do so); the CUs of such a part see their RIGHT neighbours as decoded.  suco_diff_max / suco_diff_min:
                                              log2_diff_ctu_size_max_suco_cb_size / log2_diff_max_suco_min_suco_cb_size (xevdm_util.c:1702-1727: the flag is sent for
                                              nodes with sides between 2^max(6 - max - min, 4) and 2^(6 - max))                                                       */
    int suco_diff_max, suco_diff_min;
} xhost_stream_params;

/* ALF parameter set as it is coded in an APS NAL unit (XEVD_ALF_SLICE_PARAM after xevdm_eco_alf_aps_param) */
typedef struct xhost_alf_aps {
    int aps_id;                            /* 0..31                                                                 */
    int luma_present, chroma_present;      /* alf_luma_filter_signal_flag / alf_chroma_filter_signal_flag            */
    int luma_type_7x7;                     /* alf_luma_type_flag: 0 = 5x5 diamond (6 coded coefficients), 1 = 7x7 (12) */
    int num_luma_filters;                  /* 1..25                                                                 */
    uint8_t delta_idx[25];                 /* filter of every class                                                 */
    int coef_delta_flag, pred_mode_flag;
    uint8_t filter_coef_flag[25];          /* with coef_delta_flag: which filters carry coefficients                */
    int16_t luma_coef[25][12];             /* coded values (differences with pred_mode_flag)                        */
    int16_t chroma_coef[6];
    /* fixed filter sets (alf_luma_fixed_filter_usage_pattern, xevdm_eco.c:2436-2466): 0 = none, 1 = every class starts from one of its 16 fixed
       filters, 2 = the classes flagged in fixed_filter_usage; fixed_filter_idx 0..15 selects the filter (alf_class_to_filter_mapping) */
    int fixed_filter_pattern;
    uint8_t fixed_filter_usage[25], fixed_filter_idx[25];
} xhost_alf_aps;
/* DRA parameter set as an APS NAL unit of type 1 carries it (SIG_PARAM_DRA, src_main/xevdm_dra.h:76-89; descriptors 4.9 fixed point) */
typedef struct xhost_dra_aps {
    int aps_id;                            /* 0..31                                                                  */
    int num_ranges;                        /* 1..32                                                                  */
    int in_ranges[33];                     /* luma range borders, increasing, in_ranges[0] >= 1, steps <= 1023          */
    int scale[32];                         /* dra_scale_value per range, 4.9 fixed point (512 = 1.0)                  */
    int cb_scale, cr_scale;                /* dra_cb_scale_value / dra_cr_scale_value                                 */
    int table_idx;                         /* dra_table_idx 0..58 (58: chroma scales taken as they are)               */
} xhost_dra_aps;
/* slice-level ALF parameters of the NEXT picture (sh.alf_on, aps ids, CTB map) */
typedef struct xhost_slice_alf {
    int alf_on, aps_id_y, aps_id_ch, chroma_idc;      /* chroma_idc: bit 0 Cb, bit 1 Cr                              */
    int ctb_map;                           /* alf_sh_param.is_ctb_alf_on: per-CTU luma flags are coded              */
    const uint8_t *ctb_flag;               /* [n_ctu], used with ctb_map                                            */
} xhost_slice_alf;

xhost_writer *xhost_writer_open(const xhost_stream_params *sp);
int  xhost_writer_add_dra_aps(xhost_writer *w, const xhost_dra_aps *aps);       /* appends a DRA APS NAL unit (needs tool_dra), before the pictures */
int  xhost_writer_add_alf_aps(xhost_writer *w, const xhost_alf_aps *aps);       /* appends an APS NAL unit (needs tool_alf)    */
int  xhost_writer_set_slice_alf(xhost_writer *w, const xhost_slice_alf *sa);    /* for the next xhost_writer_add_picture       */
/* Several slices per picture (Main profile with tiles and sps_pocs_flag): every following picture is written as n slice NAL units, slice k holding the
   rectangle of tiles first_tile .. last_tile (tile ids in raster order of the PPS's grid; the rectangles must partition the grid).  slice_qp / deblock_on:
   -1 = what xhost_writer_add_picture / the stream parameters say.  n = 0: back to one slice with every tile.  The decoders run the in-loop filters of
   the whole picture with the LAST slice's header (src_main/xevdm.c:3138-3199).                                                                       */
typedef struct xhost_slice_desc { int first_tile, last_tile, slice_qp, deblock_on; } xhost_slice_desc;
int  xhost_writer_set_slices(xhost_writer *w, int n, const xhost_slice_desc *slices);
/* before the first picture: the PPS announces arbitrary slices and every slice of several tiles lists its tiles (arbitrary_slice_flag, ascending tile ids)
   instead of naming the rectangle's corners - the same slices in the other syntax                                                                  */
int  xhost_writer_set_arbitrary_slices(xhost_writer *w, int on);
/* Appends one picture (SPS + PPS first when it is the first).  `b`: leaf CUs of a quad tree (64..4) in decode order with the
   fields of xgpu_cu_batch; per CU the writer keeps pred_mode (INTRA / INTER / SKIP / DIR), refi and mv of the lists in use
   (INTER: refi[l] >= 0 selects the lists, indices are clamped to the actual list sizes; SKIP / DIR CUs take derived motion),
   qp[0] (luma dequant QP incl. 6*(bd-8), used when the CU has coefficients and cu_qp_delta is on), cbf, ipm[0], coefficients,
   and with tool_ats: ats (intra CUs up to 32x32 with luma coefficients) and ats_inter (inter CUs; dropped when the shape does
   not allow the split; the coefficient blocks then have the TU size).
   idr != 0 forces an IDR picture with an I slice.  temporal_id: nuh_temporal_id (0 for low-delay streams). */
/* sps_btt_flag: which splits the stream allows for the node (x, y, 2^log2w x 2^log2h) - allow[0] none, [1] / [2] binary with a vertical / horizontal cut, [3] / [4]
   ternary; 2 instead of 1: the children get a mode constraint - either every CU below is an inter CU of a P / B picture ("inter only"), or the node starts a local dual
   tree: luma-only intra / IBC CUs below (xgpu_cu_batch.tree 1), then the node's chroma-only CU (tree 2) in the batch; 3: the dual tree only (a child would be 4x4) */
int  xhost_writer_split_allowed(xhost_writer *w, int x, int y, int log2w, int log2h, int allow[5]);
int  xhost_writer_add_picture(xhost_writer *w, int idr, int slice_type, int slice_qp, int temporal_id, const xgpu_cu_batch *b);
/* Appends a picture-signature SEI NAL unit (payload type 0x10) for the picture added last: the MD5 digests of its decoded Y, U, V
   planes (16-bit little-endian samples, rows without padding).  The reference decoder verifies them when
   XEVD_CFG_SET_USE_PIC_SIGNATURE is set (src_base/xevd.c:2010-2026) - the MD5 round trip of SURVEY 8c. */
int  xhost_writer_add_md5_sei(xhost_writer *w, const uint8_t md5[3][16]);
/* tool_dmvr together with tool_hmvp / tool_mmvd: the writer derives the same motion as a decoder, refined vectors included, so it needs the decoded luma
   samples of every reference picture it has written (a test harness decodes the stream so far); same contract as xhost_parser_set_ref_luma */
int  xhost_writer_set_ref_luma(xhost_writer *w, int poc, const int16_t *plane, int stride);
int  xhost_writer_bytes(xhost_writer *w, const uint8_t **bytes, size_t *size);
void xhost_writer_close(xhost_writer *w);

#ifdef __cplusplus
}
#endif
#endif
