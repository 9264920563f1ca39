// xgpu_internal.h - device-side record formats and the context of the MI355X reconstruction backend.
// Everything here is private to xevd_amd/csrc; the public boundary is include/xevd_hip.h.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <condition_variable>
#include <functional>
#include <mutex>
#include <thread>
#include <vector>
#include "../../include/xevd_hip.h"

// ---------------------------------------------------------------------------------------------------------
// Device picture.  16-bit planar 4:2:0 like the reference's XEVD_PIC (src_base/xevd_def.h:630-679), but with a
// geometry chosen for HBM: the first active sample of every row is 256-byte aligned (left margin 192 luma /
// 128 chroma samples >= the reference's 144 / 72 padding) and the row stride is a multiple of 64 samples, so
// that a wave's 128-byte row segments never straddle more cache lines than necessary.
// ---------------------------------------------------------------------------------------------------------
#define XGPU_MARGIN_L 192
#define XGPU_MARGIN_C 128

struct DevPic {
    int16_t *base;        // one allocation: Y, U, V padded planes
    int16_t *y, *u, *v;   // first active sample of each plane
    int      s_l, s_c;    // strides in samples
    int      used;
};

// Per-CU record on the device: 32 bytes, read as two 16-byte loads.  Built on the host from the SoA batch of the
// ABI because every consumer (inter kernel, map update, TB list) needs all fields of a CU together.
struct __attribute__((aligned(16))) CuRec {
    uint16_t x, y;            // luma sample position
    uint8_t  log2w, log2h;
    uint8_t  pred_mode;       // XGPU_MODE_*
    uint8_t  cbf;             // bit c: component c coded
    int8_t   refi[2];
    uint8_t  qp_map;          // core->qp = qp_y - 6*(bd-8): the QP stored in map_scu (xevd_util.c:1626)
    uint8_t  map_cbf;         // luma cbf bit of the SCU map = is_coef_sub[Y_C][0] (xevd_util.c:1615)
    uint32_t coef_off;        // offset of the CU's residual block in the arena (s16 units)
    int16_t  mv[2][2];        // unclipped quarter-pel
    uint8_t  qp[3];           // dequant QPs
    uint8_t  ipm[2];
    uint8_t  ats_inter;       // ats_inter_info of an inter CU (idx | pos << 4), 0 = whole-CU transform
    uint8_t  affine;          // 0, or the number of control points (2 / 3) of an affine CU: predicted by k_affine, not k_inter
    uint8_t  dmvr;            // 1: a DMVR candidate with two references and at least 8x8 samples - predicted by k_dmvr when the picture's POCs allow (else k_inter)
};
static_assert(sizeof(CuRec) == 32, "CuRec must be 32 bytes");

// SCU map record: the reference's map_scu / map_refi / map_mv (xevd_def.h:372-438) fused into one 16-byte
// element so that a deblocking lane fetches a neighbour with one load.
struct __attribute__((aligned(16))) ScuRec {
    uint32_t scu;             // bit 15 intra, 16-22 QP, 23 skip, 24 luma cbf, 31 COD; bits 8/9: left/top CU edge; bits 0-7, 12-14: SCU_RANK
    int8_t   refi[2];
    uint16_t ats_inter;       // mctx->map_ats_inter of the SCU (ADDB: non-zero on either side of an edge -> bS 2)
    int16_t  mv[2][2];
};
static_assert(sizeof(ScuRec) == 16, "ScuRec must be 16 bytes");
#define SCU_EDGE_L (1u << 8)      // the SCU's left edge is a CU boundary  (free bits 7:14 of map_scu)
#define SCU_EDGE_T (1u << 9)      // the SCU's top edge is a CU boundary
#define SCU_NOCH_L (1u << 10)     // ... but not an edge of the chroma block (a luma-only CU inside a local dual tree): the filters leave chroma alone there
#define SCU_NOCH_T (1u << 11)
// bits 0-7 and 12-14: the low 11 bits of the CU's index in the batch = its place in decoding order.  The baseline deblocking filter of the Main library reaches a
// vertical CU edge with the LATER of the two CUs (xevdm_df.c:186-330: left edge when the left neighbour is done, right edge when the right one is), so with
// sps_suco_flag chroma edges 2 samples apart are not always applied left to right; two CUs of one CTU are at most 320 places apart, CTUs follow each other
#define SCU_RANK(idx)  (((uint32_t)(idx) & 0xFFu) | ((((uint32_t)(idx) >> 8) & 7u) << 12))
#define SCU_RANK_OF(m) ((int)(((m) & 0xFFu) | (((m) >> 4) & 0x700u)))
#define CU_NOCH_L 0x40            // CuRec.pred_mode bits 6 / 7 carry the two flags to k_inter's map pass (bits 0-3: XGPU_MODE_*)
#define CU_NOCH_T 0x80

// One coded transform block for the dequant + inverse-transform kernel.
struct TbRec {
    uint32_t off;             // arena offset (s16 units) of the TB's first coefficient
    uint8_t  log2w, log2h, qp;
    uint8_t  log2s;           // log2 of the row stride: = log2w, or log2 of the CU width for a 64x64 sub-block of a larger CU
};
// transform kinds of a work item (wave-uniform): 0 = DCT-II, 1 = DST-VII, 2 = DCT-VIII (ATS, src_main/xevdm_itdq.c:163-421)
#define TR_DCT2 0
#define TR_DST7 1
#define TR_DCT8 2
// One 256-thread work item of the itdq kernel: `count` consecutive TbRecs of one size class.
struct TbWave {
    uint32_t first;
    uint16_t count;
    uint8_t  log2w, log2h;
    uint8_t  tr_v, tr_h;      // TR_* of the vertical (first) and horizontal (second) stage
    uint8_t  pad[2];
};

#define XGPU_WORK_REGION 0x100u
#define XGPU_INTER_STRIP 16       // width, in 64x64 regions, of the vertical strips the inter work lists are ordered in (xgpu_batch_create, k_inter.hip)
// One item of the two uniform work lists: where, which CU, and the CU's record (the kernels need no second fetch to start their window requests)
struct __attribute__((aligned(16))) InterItem { uint32_t pos, cu, pad[2]; CuRec rec; };
static_assert(sizeof(InterItem) == 48, "InterItem must be 48 bytes");
struct RefEntry { const int16_t *y, *u, *v; int poc; int pad; };

// Kernel arguments of the inter reconstruction kernel (passed by value).
struct InterArgs {
    int16_t *cur_y, *cur_u, *cur_v;
    int      s_l, s_c;                 // all pictures of a ctx share the geometry
    int      pic_w, pic_h;
    int      bd_l, bd_c;
    int      admvp;
    // One entry of `work` per 64x64 region of the PICTURE, in vertical strips XGPU_INTER_STRIP regions wide, row by row inside a strip (every XCD takes a contiguous
    // eighth of that order: a compact patch of the picture): XGPU_WORK_REGION = the region lies inside one CU of the batch; else two bits per 32x32 tile of the
    // region (bits 2q.. for tile q = column | row << 1): 0 no SCU of the batch, 1 the tile lies inside one CU, 2 every other tile with SCUs of the batch.
    // items[entry * 4 + q]: the CU (index and record) of tile q where it lies inside one CU (all four the same for a region entry), zero elsewhere.
    const uint32_t  *work;
    const InterItem *items;
    int      n_work;
    int      regions_x, strip_entries, full_entries;      // 16 x regions_y; entries in front of the last, narrower strip
    uint32_t magic_strip, magic_last;                     // floor(2^32 / d) + 1 for d = strip_entries and the last strip's width: n / d = mulhi(n, magic) for n x d < 2^32; magic_last 0 = width 1 (or no last strip)
    const CuRec    *cus;
    const int16_t  *resid;
    ScuRec  *maps;
    int      w_scu;
    const uint32_t *owner;             // [w_scu * h_scu] batch index of the CU covering each SCU (0xFFFFFFFF: none), painted by xgpu_batch_create
    int      n_cu;
    int      cur_poc;                  // POC of the picture being decoded (DMVR's distance test)
    int      dmvr_to_map;              // k_dmvr writes its refined vectors into the map records (DmvrArgs.refined_to_map): k_inter leaves those words alone
    RefEntry refp[XGPU_MAX_REFS][2];
};

// DMVR (k_dmvr.hip): one work item per 16x16 (or smaller) sub-block of a candidate CU
struct DmvrItem { uint32_t cu; uint8_t sx, sy; uint16_t pad; };          // CU record index, sub-block origin inside the CU in 4-sample units
struct DmvrArgs {
    int16_t *cur_y, *cur_u, *cur_v;
    int      s_l, s_c;
    int      pic_w, pic_h;
    int      bd_l, bd_c;
    int      admvp, cur_poc;
    const CuRec    *cus;
    const DmvrItem *items;
    int      n_items;
    const int16_t  *resid;
    int16_t *out_mv;                   // [n_items][2][2] quarter-sample vectors kept for temporal prediction
    ScuRec  *maps;                     // the SCU map the deblocking filter reads, and
    int      w_scu, refined_to_map;    // whether it gets the refined vectors (baseline filter of the Main library) or keeps the unrefined ones (ADDB)
    RefEntry refp[XGPU_MAX_REFS][2];
};

// Affine CUs (k_affine.hip): one work item per tile of an affine CU - 16x16 luma samples in the EIF branch, 32x32 in the translation branch.
struct AffItem { uint32_t cu, aff; uint16_t tx, ty; uint32_t pad; };      // CU record index, index into the control-point array, tile origin in the CU
static_assert(sizeof(AffItem) == 16, "AffItem must be 16 bytes");
struct AffineArgs {
    int16_t *cur_y, *cur_u, *cur_v;
    int      s_l, s_c;
    int      pic_w, pic_h;
    int      bd_l, bd_c;
    int      admvp;
    const CuRec   *cus;
    const int16_t *cpmv;               // [n_affine][2][3][2] quarter-pel control points
    const AffItem *items;              // the EIF tiles (16x16), then the sub-block-translation tiles (32x32)
    int      n_eif, n_sub;
    const int16_t *resid;
    ScuRec  *maps;
    int      w_scu;
    RefEntry refp[XGPU_MAX_REFS][2];
};

// One intra CU of a picture, in dependency-level order.  Availability = the reference's COD flags at the CU's turn
// (xevd_get_nbr_b, src_base/xevd_ipred.c:47-92): bit k of `up` / `le` = the k-th 4-luma-sample unit of the row above /
// the column to the left (cw/4 + ch/4 units each) comes from the picture, else it is mid grey.
struct __attribute__((aligned(16))) IntraRec {
    uint32_t cu;              // index into the CU records
    uint32_t flags;           // bit 0: the up-left sample is available; bit 1: an intra-block-copy CU (then `le` = block vector x | y << 16, `up` = 0);
                              // bit 2: HTDF runs on the CU, bit 3: and nothing else (an inter CU), bit 4: under constrained intra prediction (the up / le masks
                              // pick the border units), bits 8-16: xevd_get_avail_intra's bits 0-8, bits 20-22: table
                              // bits 23 / 24: avail_lr - the SCU left / right of the CU's first row is reconstructed before it (the right one only where
                              // sps_suco_flag reversed a split); with bit 24 the upper half of `up` is the mask of the RIGHT column's units
    uint64_t up, le;
    uint32_t dep_first, dep_count;   // range of the dependency list: positions (in this list) of the intra CUs it reads from
    uint16_t x, y;            // the CU fields the kernel needs, copied here so that one record fetch starts the work
    uint8_t  log2w, log2h, cbf, pad0;
    uint8_t  ipm[2], pad1[2];
    uint32_t coef_off;
};
static_assert(sizeof(IntraRec) == 48, "IntraRec must be 48 bytes");
struct IntraArgs {
    int16_t *cur_y, *cur_u, *cur_v;
    int      s_l, s_c;
    int      bd_l, bd_c;
    const CuRec    *cus;
    const IntraRec *list;
    const uint32_t *deps;
    const int16_t  *resid;
    uint32_t *done;           // [n_intra] = epoch once the CU's samples are published; [n_intra] (one past) = the ticket counter
    uint32_t  epoch, ticket_base;
    int       n_intra;        // list length (the ticket counter sits at done[n_intra])
    int       first, count;   // the list range this launch covers
    int       n_small;        // level-1 launch: the LAST n_small entries of the range are CUs of at most 16 SCUs (the list is sorted by size inside a level): 16 lanes per CU (k_intra.hip)
};

struct ItdqArgs {
    const int16_t *coef;
    int16_t       *resid;
    const TbRec   *tbs;
    const TbWave  *waves;
    int            n_waves;
    int            bd;
    int            iqt;
};

// Tile borders of a picture as bit masks over CTU columns / rows: bit i set = a tile starts at CTU column (row) i > 0.  All zero for one tile.
struct TileMask {
    uint32_t vb[8], hb[8];             // pictures up to 16384 samples = 256 CTUs of 64
    __host__ __device__ bool col_start(int ctu_x) const { return (vb[ctu_x >> 5] >> (ctu_x & 31)) & 1; }
    __host__ __device__ bool row_start(int ctu_y) const { return (hb[ctu_y >> 5] >> (ctu_y & 31)) & 1; }
    // first CTU column / row of the tile that holds CTU i, and one past its last (n = CTUs per picture row / column)
    __host__ __device__ static int tile_first(const uint32_t *m, int i)
    {
        for (int w = i >> 5; w >= 0; w--) {
            uint32_t v = m[w];
            if (w == (i >> 5)) v &= 0xFFFFFFFFu >> (31 - (i & 31));
            if (v) return (w << 5) + 31 - __builtin_clz(v);
        }
        return 0;
    }
    __host__ __device__ static int tile_end(const uint32_t *m, int i, int n)
    {
        for (int w = (i + 1) >> 5; w < 8; w++) {
            uint32_t v = m[w];
            if (w == ((i + 1) >> 5)) v &= 0xFFFFFFFFu << ((i + 1) & 31);
            if (v) { const int e = (w << 5) + __builtin_ctz(v); return e < n ? e : n; }
        }
        return n;
    }
};

struct DbkArgs {
    int      s_l, s_c;
    int      pic_w, pic_h, w_scu, h_scu;
    int      bd_l, bd_c;
    int      ctu_sh;                   // SCUs per CTU = 1 << ctu_sh
    TileMask no_filter;                // tile borders the filter must leave alone (loop_filter_across_tiles = 0; else all zero)
    const ScuRec *maps;
    uint8_t  st[3][4][64];             // strength by component, edge class, map QP (host-built from xevd_tbl_df_st)
};

#define ADDB_LDS_TABLE_BYTES (52 + 52 + 260 + XGPU_MAX_REFS * 2 + 192)
#define ADDB_LDS_TABLE_DWORDS ((ADDB_LDS_TABLE_BYTES + 3) / 4)
struct AddbArgs {
    int      s_l, s_c;
    int      w_scu, h_scu;
    int      bd_l, bd_c, log2_ctu;
    int      alpha_off, beta_off, qp_u_off, qp_v_off;
    TileMask no_filter;                // as in DbkArgs
    const ScuRec *maps;
    int8_t   chroma_qp[2 * 96];        // [c][qp + 6*(bdc-8)]
    uint8_t  pic_id[XGPU_MAX_REFS * 2];// picture slot of refp[idx][list], 255 = none
    // alpha[52] beta[52] clip[260] pic_id[2 * XGPU_MAX_REFS] chroma_qp[192] in the order and at the offsets of k_addb_alf's LDS copy, as dwords
    uint32_t lds_tables[ADDB_LDS_TABLE_DWORDS];
};

#define ALF_CTB_BITS 8704           // 8192 x 4320 in 64 x 64 CTUs
struct AlfArgs {
    int      s_l, s_c, pic_w, pic_h, bd, log2_ctu, w_ctu, across_tiles;
    TileMask tiles;                    // tile starts: a CTU's windows end at its tile (alf_process_tile)
    uint32_t magic_tiles_x;            // floor(2^32 / tiles per row) + 1, set by launch_alf: tile / tiles_x as a multiplication; 0 = one tile per row
    int      multi_tile;               // 0: one tile - the masks are not looked at
    int      pad;                      // 1: the tiles on the picture border also write the 144 / 72-sample padding of the output picture (no k_pad launch)
    int      enable[3];
    const uint8_t *ctb_flag;           // device, [n_ctu] or null
    int      ctb_in_args;              // 1: the per-CTU luma flags travel in ctb_bits (pictures up to ALF_CTB_BITS CTUs): no H2D copy between the kernels
    uint32_t ctb_bits[ALF_CTB_BITS / 32];
    // The luma filter set as the filter loop wants it: for each of the 25 classes and the 4 transpose indices (xevdm_alf.c:268-273) the thirteen coefficients in the
    // block's orientation, packed in the pairs the v_dot2 / v_mad_i32_i16 forms take - (f3,f2) (f8,f7) (f6,f5) (f10,f9) (f12,f11) (f1,f0) (-,f4), high half first - so a
    // lane fetches its filter with two 16-byte loads at (class * 4 + transpose) (xgpu_alf builds it; round 5 read thirteen s16 from an LDS copy of coef_final through a
    // nibble permutation and packed them with run-time shifts, ~70 instructions per lane of a kernel bound by instruction issue).  Argument blocks above 4 KB launch on
    // this runtime (tools/ubench/kernarg_probe.hip: 8 KB tested)
    uint32_t ctab[25 * 4][8];
    uint32_t cchroma[4];               // the chroma filter: (f3,f2) (f5,f4) (f1,f0) (-,f6)
};

// One device block + one pinned staging block of a batch.  xgpu_batch_destroy returns them to the context's pool instead of freeing:
// hipMalloc / hipFree / hipHostMalloc synchronise the device and serialise across the threads of a process, which capped many-stream
// decoding on one GPU (tools/bench_multistream.py) and cost a stream synchronisation per picture.
struct BatchBlock { uint8_t *d_base; size_t d_cap; void *h_stage; size_t h_cap; hipEvent_t uploaded, done, itdq_done; };      // uploaded: the H2D copies have left the host memory; done: the reconstruction kernels have read the block

struct xgpu_dbatch {
    BatchBlock blk;
    int        n_cu, n_ctu, n_tb, n_waves;
    size_t     n_coef;
    CuRec     *d_cus;
    uint32_t  *d_ctu_start;
    uint32_t  *d_owner;               // SCU -> CU index of the batch, over the whole picture
    InterItem *d_inter_items;                        // k_inter's roles per region of the picture and the CUs of its whole tiles (InterArgs)
    uint32_t  *d_inter_work;
    int        n_inter_work;
    int16_t   *d_coef, *d_resid;
    TbRec     *d_tbs;
    TbWave    *d_waves;
    DmvrItem  *d_dmvr_items;          // sub-blocks of the DMVR candidates
    int16_t   *d_dmvr_mv;             // their vectors after xgpu_batch_recon
    int        n_dmvr;
    AffItem   *d_aff_items;           // tiles of the affine CUs
    int16_t   *d_cpmv;
    int        n_aff_eif, n_aff_sub;
    IntraRec  *d_intra;               // intra CUs sorted by dependency level
    uint32_t  *d_intra_deps;          // dependency lists (positions in d_intra)
    uint32_t  *d_intra_done;          // [n_intra] done epochs + [1] ticket counter
    int        has_ibc, has_htdf;     // the intra list holds intra-block-copy CUs / HTDF nodes
    int        order_rl;              // some CU is decoded after its right-hand neighbour (sps_suco_flag; only looked for when the baseline deblocking filter is the one that runs)
    int        has_right;             // ... CUs whose right-hand neighbours are reconstructed first (sps_suco_flag): the instantiations that know the right reference column
    TileMask   tile_starts;           // of the batch's tile grid (zero: one tile)
    int        tiles_across;          // its loop_filter_across_tiles
    int        n_intra_l1_small;      // of the level-1 entries, the ones with at most 16 SCUs (log2 w + log2 h <= 8): they end the level's part of the list
    int        n_intra, n_levels, n_intra_deps, n_intra_l1, n_intra_heads;      // n_intra_l1: CUs of level 1 (head of the list); n_intra_heads: + the strand heads of the data-flow launch (the strand members follow)
    uint32_t   intra_epoch, intra_tickets;
    void      *h_stage;               // pinned staging block
    size_t     stage_bytes;
    int        prepared;              // 1: xgpu_batch_prepare has queued the residual pass (k_itdq) on the side stream: `blk.itdq_done` says when it has finished;
                                      // 2: it ran on the main stream with the previous picture (xgpu_batch_recon_ahead)
    int        upload_waited;         // the main stream already waits for blk.uploaded
    int        used;                  // kernels that read the block have been queued (xgpu_batch_recon, or a residual pass ahead)
};

struct xgpu_ctx {
    xgpu_seq_params sp;
    int8_t          chroma_qp[2][96];  // [c][qp + 6*(bdc-8)]
    hipStream_t     stream;            // kernels
    hipStream_t     up_stream, down_stream;      // batch uploads / output downloads: their own DMA queues, ordered against the kernels by events
    hipStream_t     side_stream;       // xgpu_batch_prepare: the residual pass of the NEXT picture, behind the current picture's k_inter (fills the GPU under the
    hipEvent_t      after_inter;       //   latency-bound dependency kernel); after_inter = the point of the main stream it may start behind
    int             have_after_inter;
    std::mutex      pool_mu;           // xgpu_batch_create / _destroy may run on a builder thread next to the thread that drives the context
    struct HostRange { uint8_t *p; size_t n; };
    std::vector<HostRange> pinned;     // xgpu_host_alloc'ed ranges: batch arrays inside them are sent without a staging copy
    int             w_scu, h_scu, w_ctu, h_ctu;
    int             s_l, s_c, rows_l, rows_c;
    size_t          pic_elems, off_u, off_v;
    std::vector<DevPic> pics;
    ScuRec         *d_maps;
    uint8_t        *d_ctb_flag;       // ALF luma CTB flags of the current picture
    std::vector<BatchBlock> pool;     // blocks of destroyed batches, reused by later ones of the same stream (same HIP stream -> ordered)
    uint8_t        *d_out[2];         // packed output pictures (xgpu_pic_output): two in flight, grown on demand
    uint8_t        *d_md5;            // xgpu_pic_md5: the picture's planes as the signature's message (+ 48 bytes of digest behind it), allocated at the first call
    hipEvent_t      md5_ready;
    size_t          out_caps[2];
    hipEvent_t      out_ready[2], out_done[2];      // conversion kernel finished (kernel stream) / copy to the host finished (download stream)
    int             out_busy[2], out_next;
    int32_t        *d_dra;            // [3][1024] DRA inverse tables of the current output call
    xgpu_frame_params fp;
    int             have_frame;
    TileMask        no_dbk;            // tile borders the deblocking of the current picture leaves alone (set by xgpu_batch_recon)
    hipEvent_t      fork_ev, join_ev;  // k_dmvr / k_affine on the side stream beside k_inter: where they may start, where the kernel stream takes them back
    int             builder_threads;   // xgpu_set_builder_threads: host threads xgpu_batch_create spreads its per-CU passes over (default 1)
    int             pad_done;          // the padding of the current picture has been written (by k_alf's border tiles): xgpu_pad launches nothing
    int             where;             // 0: the picture being built lives in its DPB slot, 1: in the scratch picture
    int             order_rl;          // a batch of the picture has CUs decoded after their right-hand neighbours (xgpu_dbatch.order_rl): k_dbk's order-aware instantiation
    int             intra_small_min;   // level-1 launches with at least this many CUs of at most 16 SCUs give those 16 lanes each (k_intra_l1; XEVD_HIP_INTRA_SMALL_MIN, default 2048)
    int             addb_scalar;                       // XEVD_HIP_ADDB_SCALAR: the scalar line filters (the > 10-bit instantiation) at every bit depth - read per context, tests set it
    int             addb_pending, split_addb_alf;      // ADDB + ALF in one kernel: xgpu_deblock left its arguments in addb_args for xgpu_alf
    AddbArgs        addb_args;
    // timing
    int             timing;
    struct Ev { hipEvent_t a, b; int k; };
    std::vector<Ev> ev_pending;
    std::vector<hipEvent_t> ev_pool;
    double          t_ms[XGPU_K_COUNT];
    long long       t_n[XGPU_K_COUNT];
    char            err[256];
};

// The host threads of a batch builder: workers that stay alive between pictures (a std::thread per phase and picture cost ~0.1 ms each - more than a phase of
// the builder takes).  One pool per CALLING thread (xgpu_batch_create may run on several builder threads of one context at once): static thread_local in
// xgpu_builder.hip.  run(n, f): f(0) on the caller, f(1) .. f(n - 1) on workers, returns when all are done.
class WorkPool {
public:
    ~WorkPool() { { std::lock_guard<std::mutex> g(mu); stop = true; } cv.notify_all(); for (std::thread &t : th) t.join(); }
    template <class F> void run(int n, F &&f)
    {
        if (n <= 1) { f(0); return; }
        while ((int)th.size() < n - 1) { const int idx = (int)th.size(); th.emplace_back([this, idx]() { worker(idx); }); }
        const std::function<void(int)> fn = [&f](int k) { f(k); };
        { std::lock_guard<std::mutex> g(mu); job = &fn; want = n - 1; left = n - 1; gen++; }
        cv.notify_all();
        f(0);
        std::unique_lock<std::mutex> g(mu);
        done_cv.wait(g, [this]() { return left == 0; });
        job = nullptr;
    }
private:
    void worker(int idx)
    {
        uint64_t seen = 0;
        for (;;) {
            const std::function<void(int)> *fn = nullptr;
            {
                std::unique_lock<std::mutex> g(mu);
                cv.wait(g, [&]() { return stop || gen != seen; });
                if (stop) return;
                seen = gen;
                if (idx < want) fn = job;
            }
            if (!fn) continue;
            (*fn)(idx + 1);
            std::lock_guard<std::mutex> g(mu);
            if (--left == 0) done_cv.notify_one();
        }
    }
    std::vector<std::thread> th;
    std::mutex mu;
    std::condition_variable cv, done_cv;
    const std::function<void(int)> *job = nullptr;
    int want = 0, left = 0;
    uint64_t gen = 0;
    bool stop = false;
};

// kernel launchers (one per .hip file)
void launch_itdq(xgpu_ctx *c, const ItdqArgs &a, hipStream_t s);
void launch_inter(xgpu_ctx *c, const InterArgs &a);      // k_inter on the ctx stream
void launch_test_recon(xgpu_ctx *c, const int16_t *coef, const int16_t *pred, int is_coef, int cuw, int cuh, int16_t *rec, int s_rec, int bd);
void launch_test_dbk(xgpu_ctx *c, int16_t *buf, int16_t *buf_v, int st, int st_v, int stride, int bd, int hor, int chroma);
void launch_intra(xgpu_ctx *c, const IntraArgs &a, bool dep, bool ibc, bool htdf, const ItdqArgs *next, bool right = false);      // next != NULL (dep launches): k_intra_itdq
int  intra_chunk(bool with_itdq);                // list positions per ticket of the data-flow launch (= waves per workgroup: 8, with the residual pass riding 4)
void launch_affine(xgpu_ctx *c, const AffineArgs &a, hipStream_t s);
void launch_dmvr(xgpu_ctx *c, const DmvrArgs &a, hipStream_t s);
void launch_dbk(xgpu_ctx *c, const DbkArgs &a, int dir, const DevPic &src, const DevPic &dst, bool order_rl = false);
void upload_transform_tables(const int *tm, const int16_t *ats, hipStream_t s);
int  itdq_group_size(int log2w, int log2h);     // TBs of one size class per 256-thread work item
void launch_addb_fused(xgpu_ctx *c, const AddbArgs &a, const DevPic &src, const DevPic &dst);      // both passes, one read + one write of the picture
void launch_alf(xgpu_ctx *c, const AlfArgs &a, const AddbArgs *deblock, const DevPic &src, const DevPic &dst);      // deblock != NULL: ADDB on SRC first, inside the same kernel
void launch_pad(xgpu_ctx *c, const DevPic &p);
void launch_copy_bw(xgpu_ctx *c, const void *src, void *dst, size_t bytes);
void launch_output(xgpu_ctx *c, const DevPic &pic, const int32_t *d_dra, int out_bd, int crop_l, int crop_r, int crop_t, int crop_b, uint8_t *d_dst, bool raw16 = false);
void launch_md5(xgpu_ctx *c, hipStream_t s, const uint8_t *d_msg, int w, int h, uint32_t *d_digest);      // k_md5.hip: the three planes packed back to back at d_msg -> d_digest[3][4]
void launch_test_mc(xgpu_ctx *c, const int16_t *plane, int stride, int ref_x, int ref_y, int has_dx, int has_dy,
                    int gmv_x, int gmv_y, int16_t *pred, int w, int h, int bd, int luma);
