// xgpu_launch.hip - the per-picture launch sequencing behind xgpu_batch_prepare / _recon / _recon_ahead, xgpu_deblock, xgpu_alf, xgpu_pad: which kernels, on which
// stream, in which order, with which arguments.  The kernels live in k_*.hip.
#include "xgpu_host.h"
#include "addb_filter.h"

// xevd_tbl_df_st (src_base/xevd_tbl.c:306-324): deblocking strength by edge class and QP - a table of the
// MPEG-5 EVC specification.
static const uint8_t k_df_st[4][52] = {
    { 0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,1,1,1,1,1,1,1,1,1,2,2,2,2,2,3,3,3,4,4,4,5,5,6,6,7,8,9,10,11,12,12,12,12,12 },
    { 0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,1,1,1,1,1,2,2,2,3,3,3,4,4,5,5,6,7,8, 9,10,11,11,11,11,11 },
    { 0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,1,1,1,2,2,2,3,3,4,4,5,6,7, 8, 9,10,10,10,10,10 },
    { 0 },
};

static ItdqArgs itdq_args(const xgpu_ctx *c, const xgpu_dbatch *db)
{
    ItdqArgs ia;
    ia.coef = db->d_coef; ia.resid = db->d_resid; ia.tbs = db->d_tbs; ia.waves = db->d_waves; ia.n_waves = db->n_waves;
    ia.bd = c->sp.bit_depth_luma;      // the LUMA depth drives dequant/transform shifts of all components (xevd.c:441-442)
    ia.iqt = c->sp.tool_iqt;
    return ia;
}

// The residual pass of a batch depends on nothing but the batch: queued AHEAD, on the side stream, behind the k_inter of the picture being reconstructed now, it
// runs under that picture's dependency kernel (k_intra's data-flow launch keeps a few thousand waves busy for tens of microseconds - most of the GPU idles)
// and its filters instead of in front of its own k_inter.  Optional: xgpu_batch_recon launches the pass itself for a batch that was not prepared.
int xgpu_batch_prepare(xgpu_ctx *c, xgpu_dbatch *db)
{
    ARGCHK(c, c != NULL); ARGCHK(c, db != NULL);
    if (db->prepared) return XGPU_OK;
    HIPCHK(c, hipStreamWaitEvent(c->side_stream, db->blk.uploaded, 0));
    if (c->have_after_inter) HIPCHK(c, hipStreamWaitEvent(c->side_stream, c->after_inter, 0));
    const ItdqArgs ia = itdq_args(c, db);
    launch_itdq(c, ia, c->side_stream);
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, hipEventRecord(db->blk.itdq_done, c->side_stream));
    db->prepared = 1;
    return XGPU_OK;
}

int xgpu_batch_recon(xgpu_ctx *c, xgpu_dbatch *db) { return xgpu_batch_recon_ahead(c, db, NULL); }

// next != NULL: the residual pass of the NEXT picture's batch is queued with this picture's kernels, on the same stream - inside the data-flow intra launch
// when the picture has one (k_intra_itdq), behind the last kernel otherwise.  No second stream and no event between streams is involved (the side-stream form,
// xgpu_batch_prepare, pays two cross-stream waits of ~6 us per picture and overlaps with the wrong kernel; profiles/round3_trace_window.txt).
int xgpu_batch_recon_ahead(xgpu_ctx *c, xgpu_dbatch *db, xgpu_dbatch *next)
{
    ARGCHK(c, c != NULL); ARGCHK(c, db != NULL); ARGCHK(c, c->have_frame); ARGCHK(c, next != db);
    if (!db->upload_waited) HIPCHK(c, hipStreamWaitEvent(c->stream, db->blk.uploaded, 0));       // the batch's arrays come through the upload stream
    db->upload_waited = 0;
    db->used = 1;                                                        // xgpu_batch_destroy records blk.done: the block may be overwritten behind the kernels queued until then
    if (next && next->prepared) next = NULL;
    const bool ahead = next != NULL;
    // (the stream waits for the NEXT batch's upload only in front of the launch that carries its residual pass - behind this picture's k_inter and tool kernels: a slow
    //  upload of picture k + 1, 45 MB of coefficients at 8K, must not hold up picture k)
    bool next_waited = false;
    auto wait_next = [&]() -> int {
        if (next && !next_waited) { HIPCHK(c, hipStreamWaitEvent(c->stream, next->blk.uploaded, 0)); next_waited = true; }
        return XGPU_OK;
    };
    if (db->tiles_across) memset(&c->no_dbk, 0, sizeof(c->no_dbk)); else c->no_dbk = db->tile_starts;
    if (db->prepared == 2) {                                             // the residual pass ran on this stream with the previous picture
        db->prepared = 0;
    } else if (db->prepared) {                                           // xgpu_batch_prepare ran the residual pass on the side stream
        HIPCHK(c, hipStreamWaitEvent(c->stream, db->blk.itdq_done, 0));
        db->prepared = 0;
    } else {
        const ItdqArgs ia = itdq_args(c, db);
        TIMED(c, XGPU_K_ITDQ, launch_itdq(c, ia, c->stream));
    }

    InterArgs a;
    memset(&a, 0, sizeof(a));
    // out-of-place filter chain: the deblocking passes (ADDB: one fused kernel; baseline filter: two) + one of ALF must end in the DPB slot ->
    // start in the scratch picture when the number of passes is odd.  ADDB followed by ALF is ONE pass (k_addb_alf).
    c->where = (addb_alf_fused(c) ? 1 : (c->fp.deblock_on ? (c->sp.tool_addb ? 1 : 2) : 0) + (c->fp.alf_on ? 1 : 0)) & 1;
    DevPic &cur = c->where ? c->pics[0] : dpic(c, c->fp.pic);
    a.cur_y = cur.y; a.cur_u = cur.u; a.cur_v = cur.v;
    a.s_l = c->s_l; a.s_c = c->s_c; a.pic_w = c->sp.width; a.pic_h = c->sp.height;
    a.bd_l = c->sp.bit_depth_luma; a.bd_c = c->sp.bit_depth_chroma;
    a.admvp = c->sp.tool_admvp ? 1 : 0;
    a.work = db->d_inter_work; a.items = db->d_inter_items; a.n_work = db->n_inter_work;
    {
        const int regions_x = (c->sp.width + 63) >> 6, regions_y = (c->sp.height + 63) >> 6;
        a.regions_x = regions_x; a.strip_entries = XGPU_INTER_STRIP * regions_y; a.full_entries = (regions_x / XGPU_INTER_STRIP) * a.strip_entries;
        a.magic_strip = (uint32_t)((1ull << 32) / (uint64_t)a.strip_entries) + 1u;
        // (a last strip ONE region wide: floor(2^32 / 1) + 1 wraps to 1 - magic 0 tells the kernel that the quotient is the index itself)
        a.magic_last = (regions_x % XGPU_INTER_STRIP) > 1 ? (uint32_t)((1ull << 32) / (uint64_t)(regions_x % XGPU_INTER_STRIP)) + 1u : 0u;
    }
    a.cus = db->d_cus; a.resid = db->d_resid;
    a.maps = c->d_maps; a.w_scu = c->w_scu; a.owner = db->d_owner; a.n_cu = db->n_cu; a.cur_poc = c->fp.poc;
    c->order_rl |= db->order_rl;                            // (the pictures' batches - one per slice - say it for the deblocking pass behind them)
    for (int l = 0; l < 2; l++)
        for (int i = 0; i < XGPU_MAX_REFS; i++) {
            const DevPic &rp = i < c->fp.num_refp[l] ? dpic(c, c->fp.refp_pic[i][l]) : dpic(c, c->fp.pic);
            a.refp[i][l].y = rp.y; a.refp[i][l].u = rp.u; a.refp[i][l].v = rp.v;
            a.refp[i][l].poc = i < c->fp.num_refp[l] ? c->fp.refp_poc[i][l] : 0;
        }
    // k_dmvr and k_affine reconstruct CUs that k_inter skips, from the same references: on the side stream beside it (not while single kernels are timed).  The map records
    // are shared: k_inter leaves the vector words those kernels write alone (k_inter.hip)
    static const bool serial_knob = getenv("XEVD_HIP_TOOLS_SERIAL") != NULL;      // A/B measurements
    const bool beside = !c->timing && !serial_knob && (db->n_dmvr || db->n_aff_eif + db->n_aff_sub);
    hipStream_t tool_stream = c->stream;
    a.dmvr_to_map = c->sp.tool_addb ? 0 : 1;
    if (beside) {
        HIPCHK(c, hipEventRecord(c->fork_ev, c->stream));
        HIPCHK(c, hipStreamWaitEvent(c->side_stream, c->fork_ev, 0));
        tool_stream = c->side_stream;
    }
    // the inter pass (k_inter.hip): one launch, a workgroup per 64x64 region of the picture
    TIMED(c, XGPU_K_INTER, launch_inter(c, a));
    if (!ahead) HIPCHK(c, hipEventRecord(c->after_inter, c->stream));   // where a residual pass prepared on the side stream (xgpu_batch_prepare) may start
    c->have_after_inter = 1;
    if (db->n_dmvr) {
        DmvrArgs d;
        memset(&d, 0, sizeof(d));
        d.cur_y = cur.y; d.cur_u = cur.u; d.cur_v = cur.v; d.s_l = c->s_l; d.s_c = c->s_c; d.pic_w = c->sp.width; d.pic_h = c->sp.height;
        d.bd_l = c->sp.bit_depth_luma; d.bd_c = c->sp.bit_depth_chroma; d.admvp = a.admvp; d.cur_poc = c->fp.poc;
        d.cus = db->d_cus; d.items = db->d_dmvr_items; d.n_items = db->n_dmvr; d.resid = db->d_resid; d.out_mv = db->d_dmvr_mv; d.maps = c->d_maps; d.w_scu = c->w_scu; d.refined_to_map = c->sp.tool_addb ? 0 : 1;
        memcpy(d.refp, a.refp, sizeof(d.refp));
        TIMED(c, XGPU_K_DMVR, launch_dmvr(c, d, tool_stream));
    }
    if (db->n_aff_eif + db->n_aff_sub) {
        AffineArgs f;
        memset(&f, 0, sizeof(f));
        f.cur_y = cur.y; f.cur_u = cur.u; f.cur_v = cur.v; f.s_l = c->s_l; f.s_c = c->s_c; f.pic_w = c->sp.width; f.pic_h = c->sp.height;
        f.bd_l = c->sp.bit_depth_luma; f.bd_c = c->sp.bit_depth_chroma; f.admvp = a.admvp;
        f.cus = db->d_cus; f.cpmv = db->d_cpmv; f.items = db->d_aff_items; f.n_eif = db->n_aff_eif; f.n_sub = db->n_aff_sub;
        f.resid = db->d_resid; f.maps = c->d_maps; f.w_scu = c->w_scu;
        memcpy(f.refp, a.refp, sizeof(f.refp));
        TIMED(c, XGPU_K_AFFINE, launch_affine(c, f, tool_stream));
    }
    if (beside) {
        HIPCHK(c, hipEventRecord(c->join_ev, c->side_stream));
        HIPCHK(c, hipStreamWaitEvent(c->stream, c->join_ev, 0));
    }
    if (db->n_intra) {
        // intra CUs: level 1 as a plain launch, all deeper levels as one data-flow launch (k_intra.hip)
        IntraArgs ta;
        ta.cur_y = cur.y; ta.cur_u = cur.u; ta.cur_v = cur.v; ta.s_l = c->s_l; ta.s_c = c->s_c; ta.bd_l = c->sp.bit_depth_luma; ta.bd_c = c->sp.bit_depth_chroma;
        ta.cus = db->d_cus; ta.list = db->d_intra; ta.deps = db->d_intra_deps; ta.resid = db->d_resid;
        ta.done = db->d_intra_done; ta.n_intra = db->n_intra;
        ta.epoch = ++db->intra_epoch;                      // flags are compared against the epoch: no reset between pictures
        ta.ticket_base = db->intra_tickets;                // the counter keeps running: a launch draws one ticket per workgroup
        const int n_dep = db->n_intra_heads - db->n_intra_l1;      // strand heads: the members behind them in the list are reached through their parents
        {
            // the next picture's residual pass rides in the data-flow launch (not while single kernels are being timed; HTDF's workgroups are a different shape)
            ItdqArgs na;
            const bool ride = next && n_dep > 0 && !c->timing && !db->has_htdf && (na = itdq_args(c, next), na.n_waves > 0);
            if (ride) { const int rc = wait_next(); if (rc != XGPU_OK) return rc; }
            TIMED(c, XGPU_K_INTRA, {
                ta.first = 0; ta.count = db->n_intra_l1; ta.n_small = db->n_intra_l1_small;
                if (ta.count) launch_intra(c, ta, false, db->has_ibc != 0, db->has_htdf != 0, NULL, db->has_right != 0);
                ta.first = db->n_intra_l1; ta.count = n_dep; ta.n_small = 0;
                if (n_dep) {
                    launch_intra(c, ta, true, db->has_ibc != 0, db->has_htdf != 0, ride ? &na : NULL, db->has_right != 0);
                    const int chunk = intra_chunk(ride);
                    db->intra_tickets += (uint32_t)((ta.count + chunk - 1) / chunk);
                    if (ride) { next->upload_waited = 1; next->prepared = 2; next->used = 1; next = NULL; }
                }
            });
        }
    }
    if (next) {
        const int rc = wait_next();
        if (rc != XGPU_OK) return rc;
        const ItdqArgs na = itdq_args(c, next);
        TIMED(c, XGPU_K_ITDQ, launch_itdq(c, na, c->stream));
        next->upload_waited = 1; next->prepared = 2; next->used = 1;
    }
    HIPCHK(c, hipGetLastError());
    return XGPU_OK;
}

int xgpu_batch_dmvr_mvs(xgpu_ctx *c, xgpu_dbatch *db, int16_t *mv, int n)
{
    ARGCHK(c, c != NULL); ARGCHK(c, db != NULL && n >= 0);
    if (mv && db->n_dmvr) {
        ARGCHK(c, n >= db->n_dmvr);
        HIPCHK(c, hipMemcpyAsync(mv, db->d_dmvr_mv, sizeof(int16_t) * 4 * (size_t)db->n_dmvr, hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));
    }
    return db->n_dmvr;
}


int xgpu_deblock(xgpu_ctx *c)
{
    ARGCHK(c, c != NULL); ARGCHK(c, c->have_frame); ARGCHK(c, c->fp.deblock_on);
    DevPic &slot = dpic(c, c->fp.pic), &scratch = c->pics[0];
    DevPic &first = c->where ? scratch : slot, &second = c->where ? slot : scratch;      // first -> second -> first
    const int boff = 6 * (c->sp.bit_depth_chroma - 8);
    if (c->sp.tool_addb) {
        AddbArgs a;
        memset(&a, 0, sizeof(a));
        a.s_l = c->s_l; a.s_c = c->s_c; a.w_scu = c->w_scu; a.h_scu = c->h_scu;
        a.bd_l = c->sp.bit_depth_luma; a.bd_c = c->sp.bit_depth_chroma; a.log2_ctu = c->sp.log2_ctu;
        a.alpha_off = c->fp.deblock_alpha_offset; a.beta_off = c->fp.deblock_beta_offset;
        a.qp_u_off = c->fp.qp_u_offset; a.qp_v_off = c->fp.qp_v_offset; a.maps = c->d_maps; a.no_filter = c->no_dbk;
        memcpy(a.chroma_qp, c->chroma_qp, sizeof(a.chroma_qp));
        for (int l = 0; l < 2; l++)
            for (int i = 0; i < XGPU_MAX_REFS; i++) a.pic_id[i * 2 + l] = i < c->fp.num_refp[l] ? (uint8_t)c->fp.refp_pic[i][l] : 255;
        {
            uint8_t *tb = (uint8_t *)a.lds_tables;
            memcpy(tb, h_addb_alpha, 52); memcpy(tb + 52, h_addb_beta, 52); memcpy(tb + 104, h_addb_clip, 260);
            memcpy(tb + 364, a.pic_id, XGPU_MAX_REFS * 2); memcpy(tb + 364 + XGPU_MAX_REFS * 2, a.chroma_qp, 192);
        }
        if (addb_alf_fused(c)) {
            ARGCHK(c, c->where == 1 && !c->addb_pending);
            c->addb_args = a; c->addb_pending = 1;      // runs inside xgpu_alf's kernel
            return XGPU_OK;
        }
        TIMED(c, XGPU_K_DBK_V, launch_addb_fused(c, a, first, second));      // both edge directions: one read + one write of the picture (timed as "dbk_v")
        c->where ^= 1;
    } else {
        DbkArgs a;
        memset(&a, 0, sizeof(a));
        a.s_l = c->s_l; a.s_c = c->s_c; a.pic_w = c->sp.width; a.pic_h = c->sp.height; a.w_scu = c->w_scu; a.h_scu = c->h_scu;
        a.bd_l = c->sp.bit_depth_luma; a.bd_c = c->sp.bit_depth_chroma; a.maps = c->d_maps; a.ctu_sh = c->sp.log2_ctu - 2; a.no_filter = c->no_dbk;
        // strength LUT: xevd_df.c:347-365 with the table index clamped to 0..51 (see oracle chroma_qp())
        for (int cls = 0; cls < 4; cls++)
            for (int qp = 0; qp < 64; qp++) {
                a.st[0][cls][qp] = (uint8_t)(k_df_st[cls][std::min(qp, 51)] << (c->sp.bit_depth_luma - 8));
                for (int t = 0; t < 2; t++) {
                    const int q = std::min(std::max(qp + (t ? c->fp.qp_v_offset : c->fp.qp_u_offset), -boff), 57);
                    const int v = std::min(std::max((int)c->chroma_qp[t][q + boff], 0), 51);
                    a.st[1 + t][cls][qp] = (uint8_t)(k_df_st[cls][v] << (c->sp.bit_depth_chroma - 8));
                }
            }
        TIMED(c, XGPU_K_DBK_V, launch_dbk(c, a, 0, first, second, c->order_rl != 0));
        TIMED(c, XGPU_K_DBK_H, launch_dbk(c, a, 1, second, first));
    }
    HIPCHK(c, hipGetLastError());
    return XGPU_OK;
}

int xgpu_alf(xgpu_ctx *c, const xgpu_alf_params *ap)
{
    ARGCHK(c, c != NULL); ARGCHK(c, c->have_frame); ARGCHK(c, c->fp.alf_on); ARGCHK(c, ap != NULL && c->where == 1);
    ARGCHK(c, !addb_alf_fused(c) || c->addb_pending);      // deblock_on was announced: xgpu_deblock comes first
    ARGCHK(c, (!ap->enable[0] || ap->luma_coef) && ((!ap->enable[1] && !ap->enable[2]) || ap->chroma_coef));
    AlfArgs a;
    memset(&a, 0, sizeof(a));
    a.s_l = c->s_l; a.s_c = c->s_c; a.pic_w = c->sp.width; a.pic_h = c->sp.height;
    a.bd = c->sp.bit_depth_luma;           // one bit depth for classification and all clip ranges (xevd_alf_init, xevdm_alf.c:431-437)
    a.log2_ctu = c->sp.log2_ctu; a.w_ctu = c->w_ctu; a.across_tiles = ap->across_tiles ? 1 : 0;
    ARGCHK(c, tile_mask(c, ap->tiles, a.tiles));
    a.multi_tile = ap->tiles && ap->tiles->n_cols * ap->tiles->n_rows > 1;
    ARGCHK(c, !ap->tiles || (ap->tiles->loop_filter_across_tiles != 0) == (ap->across_tiles != 0));
    for (int i = 0; i < 3; i++) a.enable[i] = ap->enable[i] ? 1 : 0;
    if (ap->luma_coef) {
        static const uint8_t perm[4][13] = {      // coefficient order per transpose index, xevdm_alf.c:268-273
            { 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12 }, { 9, 4, 10, 8, 1, 5, 11, 7, 3, 0, 2, 6, 12 },
            { 0, 3, 2, 1, 8, 7, 6, 5, 4, 9, 10, 11, 12 }, { 9, 8, 10, 4, 3, 7, 11, 5, 1, 0, 2, 6, 12 } };
        auto pk = [](int hi, int lo) { return ((uint32_t)(uint16_t)(int16_t)hi << 16) | (uint16_t)(int16_t)lo; };
        for (int cls = 0; cls < 25; cls++)
            for (int tr = 0; tr < 4; tr++) {
                int f[13];
                for (int i = 0; i < 13; i++) f[i] = ap->luma_coef[cls * 13 + perm[tr][i]];
                uint32_t *e = a.ctab[cls * 4 + tr];
                e[0] = pk(f[2], f[3]); e[1] = pk(f[7], f[8]); e[2] = pk(f[5], f[6]); e[3] = pk(f[9], f[10]); e[4] = pk(f[11], f[12]); e[5] = pk(f[1], f[0]); e[6] = pk(0, f[4]); e[7] = 0;
            }
    }
    if (ap->chroma_coef) {
        const int16_t *g = ap->chroma_coef;
        a.cchroma[0] = ((uint32_t)(uint16_t)g[2] << 16) | (uint16_t)g[3]; a.cchroma[1] = ((uint32_t)(uint16_t)g[4] << 16) | (uint16_t)g[5];
        a.cchroma[2] = ((uint32_t)(uint16_t)g[1] << 16) | (uint16_t)g[0]; a.cchroma[3] = (uint16_t)g[6];
    }
    if (ap->ctb_flag && ap->enable[0]) {
        const int n_ctu = c->w_ctu * c->h_ctu;
        if (n_ctu <= ALF_CTB_BITS) {        // as kernel arguments: a copy engine transfer between two kernels costs ~12 us of idle device (profiles/round3_trace_window.txt)
            a.ctb_in_args = 1;
            for (int i = 0; i < n_ctu; i++) if (ap->ctb_flag[i]) a.ctb_bits[i >> 5] |= 1u << (i & 31);
        } else {
            HIPCHK(c, hipMemcpyAsync(c->d_ctb_flag, ap->ctb_flag, (size_t)n_ctu, hipMemcpyHostToDevice, c->stream));
            a.ctb_flag = c->d_ctb_flag;
        }
    }
    // the filter chain is planned so that ALF reads the scratch picture and lands in the DPB slot
    a.pad = 1;                              // the border tiles replicate their samples into the padding: xgpu_pad has nothing left to do for this picture
    TIMED(c, XGPU_K_ALF, launch_alf(c, a, c->addb_pending ? &c->addb_args : NULL, c->pics[0], dpic(c, c->fp.pic)));
    c->where = 0;
    c->addb_pending = 0;
    c->pad_done = 1;
    HIPCHK(c, hipGetLastError());
    return XGPU_OK;
}

int xgpu_pad(xgpu_ctx *c)
{
    ARGCHK(c, c != NULL); ARGCHK(c, c->have_frame); ARGCHK(c, c->where == 0);
    if (c->pad_done) return XGPU_OK;        // k_alf wrote the padding with its border tiles
    TIMED(c, XGPU_K_PAD, launch_pad(c, dpic(c, c->fp.pic)));
    HIPCHK(c, hipGetLastError());
    return XGPU_OK;
}

int xgpu_measure_copy_bw(xgpu_ctx *c, size_t bytes, int iters, double *gbps)
{
    ARGCHK(c, c != NULL); ARGCHK(c, gbps != NULL && iters > 0 && bytes >= (1u << 20));
    void *a = NULL, *b = NULL;
    bytes &= ~(size_t)15;
    if (hipMalloc(&a, bytes) != hipSuccess || hipMalloc(&b, bytes) != hipSuccess) { if (a) (void)hipFree(a); return XGPU_ERR_OUT_OF_MEMORY; }
    (void)hipMemsetAsync(a, 1, bytes, c->stream);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    launch_copy_bw(c, a, b, bytes);                       // warm-up
    (void)hipEventRecord(e0, c->stream);
    for (int i = 0; i < iters; i++) launch_copy_bw(c, (i & 1) ? b : a, (i & 1) ? a : b, bytes);
    (void)hipEventRecord(e1, c->stream);
    hipError_t e = hipStreamSynchronize(c->stream);
    float ms = 0;
    (void)hipEventElapsedTime(&ms, e0, e1);
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    (void)hipFree(a); (void)hipFree(b);
    if (e != hipSuccess || ms <= 0) return XGPU_ERR_UNEXPECTED;
    *gbps = 2.0 * (double)bytes * iters / (ms * 1e-3) / 1e9;
    return XGPU_OK;
}

