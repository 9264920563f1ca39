// evc_parser.cc - MPEG-5 EVC bitstream parser (-> CU batches for the GPU path), see include/xevd_host.h; the writer (<- synthetic CU batches) is evc_writer.cc.
// Shared layers: evc_bits.h (bits, CABAC, binarisations), evc_hls.h (parameter sets, slice state, DPB, picture maps), evc_cu.h (CU syntax + motion derivation).  Plain C++ (no HIP): it is the serial, bit-level host half of the decoder.
//
// Design (not the reference's): the CU-level syntax is written ONCE as a template over a "coder" that is either the
// arithmetic decoder or the arithmetic encoder - every syntax element is `v = c.bin(v, model)` - so the writer and the parser
// cannot drift apart; the reference decoder itself (oracle/_ref) is what pins them to the standard in tests.  A picture is
// parsed CU by CU in ONE pass (entropy decoding + motion/QP derivation + map update), straight into the structure-of-arrays
// batch that xgpu_batch_create takes; the reference's two passes over XEVD_CU_DATA (xevd_tile_eco, then xevd_ctu_row_rec_mt)
// see the same neighbour state because both walk the CUs in the same order and "reconstructed" (COD) == "already parsed".
#include "evc_cu.h"
#include <chrono>
#include <condition_variable>
#include <functional>
#include <mutex>

// =============================================================================================================== parser
// One tile of a picture being parsed: its coder state, its share of the batch, its scratch blocks.  Objects are kept between pictures (the vectors keep
// their capacity); with several tiles and xhost_parser_set_threads() > 1 they run on different threads.
struct TileParser {
    Stream &st;
    TileCoder tc;
    Batch batch;
    std::string err;
    size_t n_coef = 0;
    explicit TileParser(Stream &s) : st(s), tc(s) {}
    int fail(const char *m) { err = m; return XHOST_ERR_MALFORMED; }
    // xevd_tile_eco (src_main/xevdm.c:2363-2461) for tile (tc_, tr) of the grid, from bit position `pos` of the slice NAL
    int parse_tile(const BitReader &br0, size_t pos, int tcol, int trow)
    {
        BitReader br = br0;
        br.pos = pos;
        if (br.pos > br.size * 8) return fail("tile entry point past the end of the slice");
        const Slice &sh = st.sh;
        const int w_ctu = (st.sps.width + 63) >> 6;
        batch.clear();
        n_coef = 0;
        tc.missing_ref = false;
        if (st.sps.tool_cm_init) tc.models.reset_cm(sh.type == XHOST_SLICE_B, sh.qp); else tc.models.reset();
        tc.qp_prev = sh.qp;
        Dec dec;
        dec.br = &br;
        dec.start();
        for (int cy = st.grid.row_bd[trow]; cy < st.grid.row_bd[trow + 1]; cy++) for (int cx = st.grid.col_bd[tcol]; cx < st.grid.col_bd[tcol + 1]; cx++) {
            if (cx == st.grid.col_bd[tcol]) tc.history_reset();
            batch.ctu_start.push_back((uint32_t)batch.x.size());
            if (sh.alf_on && sh.alf_ctb_map) st.alf_ctb_flag[(size_t)cy * w_ctu + cx] = (uint8_t)dec.bin(0, tc.models.alf_ctb[0]);      // xevdm.c:2411-2418
            const int rc = st.sps.btt ? parse_node(dec, cx << 6, cy << 6, 6, 6, 0, false) : parse_tree(dec, cx << 6, cy << 6, 6);
            if (rc != XGPU_OK) return rc;
            if (br.overrun) return fail("slice data ends early");
            if (tc.missing_ref) return fail("tool_dmvr with tool_hmvp / tool_mmvd: the samples of a reference picture are needed (xhost_parser_set_ref_luma)");
        }
        if (dec.tile_end() != 1) return fail("missing end-of-tile flag");
        return XGPU_OK;
    }
    int parse_tree(Dec &dec, int x, int y, int log2s, int qp_code = 0, int suco = 0)
    {
        const int s = 1 << log2s;
        int split = 0;
        if (s > 4 && !(s < 8)) split = dec.bin(0, tc.models.split[0]);
        qp_code = qp_group(st, tc, split ? TileCoder::QUAD : 0, log2s, log2s, qp_code);
        if (split) {
            const int h = s >> 1;
            suco = tc.code_suco(dec, 0, TileCoder::QUAD, log2s, log2s, !(x + s <= st.sps.width && y + s <= st.sps.height), suco);
            int order[4];
            TileCoder::part_order(TileCoder::QUAD, suco, 4, order);
            for (int k = 0; k < 4; k++) {
                const int i = order[k], nx = x + (i & 1) * h, ny = y + (i >> 1) * h;
                if (nx < st.sps.width && ny < st.sps.height) { const int rc = parse_tree(dec, nx, ny, log2s - 1, qp_code, suco); if (rc != XGPU_OK) return rc; }
            }
            return XGPU_OK;
        }
        return leaf(dec, x, y, log2s, log2s, qp_code, 0);
    }
    // sps_btt_flag: a node of the binary / ternary split tree (xevd_entropy_decode_tree, src_main/xevdm.c:1644-1850)
    int parse_node(Dec &dec, int x, int y, int lw, int lh, int qp_code, bool only_inter, bool only_intra = false, int suco = 0)
    {
        const int W = st.sps.width, H = st.sps.height, w = 1 << lw, h = 1 << lh, mn = 1 << st.sps.log2_min_cb;
        int split = TileCoder::NO_SPLIT;
        if (w > mn || h > mn) {
            if (x + w <= W && y + h <= H) split = tc.code_split(dec, 0, x, y, lw, lh, only_inter);
            else {      // across the picture border: the binary split towards it, no syntax (:1687-1713)
                int allow[6];
                tc.split_allowed(allow, lw, lh, x, y, only_inter);
                split = allow[TileCoder::BI_VER] ? TileCoder::BI_VER : allow[TileCoder::BI_HOR] ? TileCoder::BI_HOR : -1;
                if (split < 0) return fail("a node across the picture border cannot be split");
            }
        }
        qp_code = qp_group(st, tc, split, lw, lh, qp_code);
        if (split == TileCoder::NO_SPLIT) { last_qp_code = qp_code; return leaf(dec, x, y, lw, lh, qp_code, only_inter, only_intra ? 1 : 0); }
        suco = tc.code_suco(dec, 0, split, lw, lh, !(x + w <= W && y + h <= H), suco);
        const int mc = only_intra ? 0 : tc.code_mode_cons(dec, split, lw, lh, only_inter, 0);
        int px[3], py[3], plw[3], plh[3], order[4];
        const int n = TileCoder::split_parts(split, x, y, lw, lh, px, py, plw, plh);
        TileCoder::part_order(split, suco, n, order);
        for (int k = 0; k < n; k++) {
            const int i = order[k];
            if (px[i] < W && py[i] < H) { const int rc = parse_node(dec, px[i], py[i], plw[i], plh[i], qp_code, mc == 1, only_intra || mc < 0, suco); if (rc != XGPU_OK) return rc; }
        }
        // local dual tree: the luma CUs above, now the node's chroma block as one CU (xevdm.c:1828-1833; core->cu_qp_delta_code keeps the last leaf's value)
        if (mc < 0) return leaf(dec, x, y, lw, lh, last_qp_code, 0, 2);
        return XGPU_OK;
    }
    int last_qp_code = 0;
    int leaf(Dec &dec, int x, int y, int lw, int lh, int qp_code, int only_inter, int tree = 0)
    {
        if (x + (1 << lw) > st.sps.width || y + (1 << lh) > st.sps.height) return fail("a CU crosses the picture border");
        Cu cu;
        memset(&cu, 0, sizeof(cu));
        cu.x = x; cu.y = y; cu.log2w = lw; cu.log2h = lh; cu.qp_code = qp_code; cu.only_inter = only_inter; cu.tree = tree;
        // the coefficient blocks are decoded where they stay: space for all three components at the end of the tile's arena (cleared block by block, where one is coded), given back behind the
        // blocks that turn out to be coded (they are packed together when one in front of them is not)
        if (batch.x.empty()) n_coef = 0;
        const size_t nl = (size_t)1 << (lw + lh), nc = nl >> 2;
        if (batch.coef.size() < n_coef + nl + 2 * nc) batch.coef.resize(std::max(n_coef + nl + 2 * nc, batch.coef.size() * 2));      // (not zeroed: NoInitAlloc)
        int16_t *coef[3] = { batch.coef.data() + n_coef, batch.coef.data() + n_coef + nl, batch.coef.data() + n_coef + nl + nc };
        tc.code_cu(dec, cu, coef, false);
        tc.commit(cu);
        // append to the batch
        batch.x.push_back((uint16_t)x); batch.y.push_back((uint16_t)y); batch.log2w.push_back((uint8_t)lw); batch.log2h.push_back((uint8_t)lh);
        batch.pred_mode.push_back((uint8_t)cu.mode);
        batch.refi.push_back((int8_t)cu.refi[0]); batch.refi.push_back((int8_t)cu.refi[1]);
        for (int l = 0; l < 2; l++) { batch.mv.push_back(cu.mv[l][0]); batch.mv.push_back(cu.mv[l][1]); }
        int qp_u, qp_v;
        tc.chroma_qps(cu.qp, qp_u, qp_v);
        batch.qp.push_back((uint8_t)(cu.qp + 6 * (st.sps.bd_l - 8))); batch.qp.push_back((uint8_t)qp_u); batch.qp.push_back((uint8_t)qp_v);
        batch.cbf.push_back((uint8_t)(cu.cbf[0] | (cu.cbf[1] << 1) | (cu.cbf[2] << 2)));
        batch.ipm.push_back((uint8_t)cu.ipm); batch.ipm.push_back((uint8_t)(st.sps.tool_eipd ? cu.ipm_c : cu.ipm));      // Baseline: chroma mode = luma mode, xevd_eco.c:1154
        batch.ats.push_back((uint8_t)cu.ats); batch.ats_inter.push_back((uint8_t)cu.ats_inter); batch.dmvr.push_back((uint8_t)cu.dmvr);
        batch.tree.push_back((uint8_t)tree); batch.has_tree |= tree != 0;
        if (st.sps.tool_affine) {      // xgpu_cu_batch.affine: 0 / 2 / 3 control points, .affine_mv[list][vertex][x/y]
            batch.affine.push_back((uint8_t)(cu.affine ? cu.affine + 1 : 0));
            batch.affine_mv.insert(batch.affine_mv.end(), &cu.aff_mv[0][0][0], &cu.aff_mv[0][0][0] + 12);
        }
        batch.coef_off.push_back((uint32_t)n_coef);
        const int tu_shift = (cu.ats_inter & 15) == 0 ? 0 : (((cu.ats_inter & 15) >= 3) ? 2 : 1);      // the TU is 1/2 or 1/4 of the CU
        for (int k = 0; k < 3; k++)
            if (cu.cbf[k]) {
                const size_t n = (size_t)1 << (lw + lh - (k ? 2 : 0) - tu_shift);
                int16_t *dst = batch.coef.data() + n_coef;
                if (dst != coef[k]) memmove(dst, coef[k], n * sizeof(int16_t));
                n_coef += n;
            }
        return XGPU_OK;
    }
};

struct xhost_parser {
    std::vector<uint8_t> data;
    size_t pos = 0;
    Stream st;
    std::vector<std::unique_ptr<TileParser>> tiles;      // one per tile of the current picture (kept between pictures)
    Batch merged;                                        // several tiles: their batches, tile by tile
    // What a handed-out xhost_picture points at lives in a ring of `held.size()` slots (xhost_parser_set_depth, default 1): the arrays of picture k
    // stay untouched until the call that hands out picture k + depth - so a caller can build and launch picture k on one thread while another
    // one is inside xhost_parser_next for picture k + 1 (depth 2).  The batch arrays are swapped into the slot (no copy), the small tables copied.
    struct Held {
        Batch batch;
        int16_t alf_luma[25][13], alf_chroma[7];
        std::vector<uint8_t> ctb;
        std::vector<int32_t> dra;
        xgpu_tile_grid grid;
        int16_t *arena = nullptr;                        // xhost_parser_set_arena: the coefficients of a picture with several tiles are gathered here
        size_t arena_cap = 0;                            //   (in samples) instead of in batch.coef
    };
    void *(*arena_alloc)(void *, size_t) = nullptr;
    void (*arena_release)(void *, void *) = nullptr;
    void *arena_user = nullptr;
    ~xhost_parser() { for (Held &h : held) if (h.arena && arena_release) arena_release(arena_user, h.arena); }
    std::vector<Held> held = std::vector<Held>(1);
    size_t n_handed = 0;
    LumaBoard board;                                     // xhost_parser_set_ref_luma_wait
    bool luma_wait = false;
    int16_t *merged_arena = nullptr;                     // the arena the coefficients of the picture being handed out were gathered in (else batch.coef)
    Batch *cur = &merged;                                // the batch of the picture handed out last (in its ring slot)
    size_t n_coef = 0;
    int n_threads = 1;                                   // xhost_parser_set_threads
    std::string err;
    int fail(const char *m) { err = m; pic_tiles_left = 0; return XHOST_ERR_MALFORMED; }      // (a picture half assembled from slices is dropped)
    // runs fn(0 .. n-1) on up to n_threads threads (the calling one included).  The helper threads live as long as the parser: creating and joining 15 threads
    // three times per picture cost milliseconds of a 9 ms picture - more when another thread of the process (a decoder's device thread inside the GPU
    // driver) holds the address-space lock that every thread creation needs.
    struct Pool {
        std::vector<std::thread> th;
        std::mutex mu;
        std::condition_variable cv, done_cv;
        const std::function<void()> *job = nullptr;
        int want = 0, left = 0;
        uint64_t gen = 0;
        bool stop = false;
        ~Pool() { { std::lock_guard<std::mutex> g(mu); stop = true; } cv.notify_all(); for (std::thread &t : th) t.join(); }
        void worker(int idx)
        {
            uint64_t seen = 0;
            for (;;) {
                const std::function<void()> *fn = nullptr;
                {
                    std::unique_lock<std::mutex> g(mu);
                    cv.wait(g, [&]() { return stop || gen != seen; });
                    if (stop) return;
                    seen = gen;
                    if (idx < want) fn = job;
                }
                if (!fn) continue;
                (*fn)();
                std::lock_guard<std::mutex> g(mu);
                if (--left == 0) done_cv.notify_one();
            }
        }
        void run(int helpers, const std::function<void()> &fn)      // fn on `helpers` pool threads and on the caller
        {
            while ((int)th.size() < helpers) { const int idx = (int)th.size(); th.emplace_back([this, idx]() { worker(idx); }); }
            { std::lock_guard<std::mutex> g(mu); job = &fn; want = helpers; left = helpers; gen++; }
            cv.notify_all();
            fn();
            std::unique_lock<std::mutex> g(mu);
            done_cv.wait(g, [this]() { return left == 0; });
            job = nullptr;
        }
    };
    std::unique_ptr<Pool> pool;
    template <class F> void parallel_for(int n, F fn)
    {
        const int nt = std::min(n_threads, n);
        if (nt <= 1) { for (int i = 0; i < n; i++) fn(i); return; }
        std::atomic<int> next(0);
        const std::function<void()> work = [&]() { for (int i = next.fetch_add(1); i < n; i = next.fetch_add(1)) fn(i); };
        if (!pool) pool.reset(new Pool());
        pool->run(nt - 1, work);
    }

    // A picture-signature SEI directly after the slice NAL belongs to that picture (xevd_dec_nalu checks it against ctx->pic,
    // src_base/xevd.c:2010-2026).  Payload (xevd_eco_sei, xevd_eco.c:1617-1678): type 0x10, size 16, then 16 bytes PER PLANE.
    void attach_signature(xhost_picture *out)
    {
        out->has_md5 = 0;
        if (pos + 6 > data.size()) return;
        const uint8_t *d = data.data() + pos;
        const size_t len = ((size_t)d[0] << 24) | ((size_t)d[1] << 16) | ((size_t)d[2] << 8) | d[3];
        if (len < 2 + 2 + 48 || pos + 4 + len > data.size()) return;
        if ((((d[4] << 8 | d[5]) >> 9) & 63) - 1 != NUT_SEI) return;
        if (d[6] != 0x10 || d[7] != 16) return;
        memcpy(out->md5, d + 8, 48);
        out->has_md5 = 1;
        pos += 4 + len;
    }

    int parse_sps(BitReader &br)
    {
        Sps tmp = st.sps;                                // parsed into a copy and committed on success only: a damaged SPS leaves the active one intact
        Sps &s = tmp;
        br.ue();                                         // sps_seq_parameter_set_id
        const int profile = (int)br.get(8);              // 0 Baseline, 1 Main, 2/3 still picture
        if (profile < 0 || profile > 3) return fail("unknown profile");
        s.profile_main = profile == 1 || profile == 3;
        br.get(8); br.get(32); br.get(32);               // level, toolset_idc_h/l
        if (br.ue() != 1) return fail("only 4:2:0 is supported");
        s.width = (int)br.ue(); s.height = (int)br.ue();
        s.bd_l = (int)br.ue() + 8; s.bd_c = (int)br.ue() + 8;
        // range checks before anything is sized or indexed by these fields (the chroma QP table below is)
        if (br.overrun || (s.width & 7) || (s.height & 7) || s.width <= 0 || s.height <= 0 || s.width > 16384 || s.height > 16384 ||
            s.bd_l < 8 || s.bd_l > 12 || s.bd_c < 8 || s.bd_c > 12) return fail("bad SPS");
        int unsupported = 0, rpl = 0, pocs = 0;
        s.tool_iqt = s.tool_ats = s.tool_addb = s.tool_alf = s.tool_eipd = s.tool_dra = s.tool_htdf = 0;
        if (!s.profile_main) {
            for (int i = 0; i < 13; i++) { const int f = br.get1(); if (i != 11) unsupported |= f; }      // btt suco admvp eipd cm_init iqt addb alf htdf rpl pocs dquant dra
        } else {                                          // xevdm_eco_sps, xevdm_eco.c:1863-1937: sub-flags follow their tool flag
            s.btt = br.get1();                           // sps_btt_flag + the limits of the split tree (xevdm_eco.c:1863-1871; table xevdm_util.c:4393-4400)
            if (s.btt) {
                if (br.ue() != 1) return fail("sps_btt_flag with a CTU size other than 64 is not supported");      // log2_ctu_size_minus5
                for (int i = 0; i < 4; i++) s.btt_raw[i] = (int)br.ue();
                if (br.overrun || s.btt_raw[0] > 4 || s.btt_raw[1] > 6 || s.btt_raw[2] > 6 || s.btt_raw[3] > 6) return fail("bad SPS: split limits");
                s.log2_min_cb = s.btt_raw[0] + 2;
                s.split_tbl[0][1] = 6; s.split_tbl[0][0] = s.log2_min_cb;
                s.split_tbl[1][1] = 6; s.split_tbl[1][0] = s.log2_min_cb + 1;
                s.split_tbl[2][1] = std::min(6 - s.btt_raw[1], 6); s.split_tbl[2][0] = s.log2_min_cb + 2;
                s.split_tbl[3][1] = std::min(6 - s.btt_raw[2], 6); s.split_tbl[3][0] = s.log2_min_cb + s.btt_raw[3] + 2;
            } else s.log2_min_cb = 2;
            s.suco = br.get1();                          // sps_suco_flag + the CU sizes that may choose (xevdm_eco.c:1872-1877)
            s.suco_raw[0] = s.suco_raw[1] = 0;
            if (s.suco) { s.suco_raw[0] = (int)br.ue(); s.suco_raw[1] = (int)br.ue(); if (br.overrun || s.suco_raw[0] > 6 || s.suco_raw[1] > 6) return fail("bad SPS: SUCO sizes"); }
            s.tool_admvp = br.get1();
            s.tool_amvr = s.tool_hmvp = s.tool_dmvr = s.tool_mmvd = s.tool_affine = 0;
            if (s.tool_admvp) { s.tool_affine = br.get1(); s.tool_amvr = br.get1(); s.tool_dmvr = br.get1(); s.tool_mmvd = br.get1(); s.tool_hmvp = br.get1(); }      // tool_affine, tool_amvr, tool_dmvr, tool_mmvd, tool_hmvp
            s.tool_eipd = br.get1();
            s.ibc = s.ibc_log_max = 0;
            if (s.tool_eipd && (s.ibc = br.get1())) { s.ibc_log_max = (int)br.ue() + 2; if (s.ibc_log_max > 7) return fail("bad SPS"); }
            s.tool_cm_init = br.get1(); s.tool_adcc = s.tool_cm_init ? br.get1() : 0;      // tool_cm_init, tool_adcc (xevdm_eco.c:1900-1904)
            s.tool_iqt = br.get1();
            if (s.tool_iqt) s.tool_ats = br.get1();
            s.tool_addb = br.get1();
            s.tool_alf = br.get1();
            s.tool_htdf = br.get1();                     // no syntax of its own: the backend filters with the slice QP (xevdm.c:1381-1392)
            rpl = br.get1(); pocs = br.get1();
            s.dquant = br.get1();                        // dquant_flag: QP deltas per quantisation group of pps.cu_qp_delta_area (xevdm.c:1739-1759, xevdm_eco.c:882-897)
            s.tool_dra = br.get1();
        }
        // tool_dmvr with tool_hmvp / tool_mmvd: xevdm_set_dec_info ends by copying map_mv[first SCU] - the REFINED vector of the CU's first sub-block - back into
        // core->mv (xevdm_util.c:4384-4387), which is what the history buffer then receives (xevdm.c:1335-1342), and an MMVD CU builds its merge list from
        // ctx->map_mv, the refined vectors (xevdm_util.c:246-247): the syntax of later CUs of the SAME picture depends on the refinement search.  The
        // front end then searches itself (TileCoder::commit, dmvr_search.h) on the reference samples the caller registers (xhost_parser_set_ref_luma)
        if (unsupported) return fail("a Baseline SPS with a Main profile tool flag");
        // xevdm_eco.c:1920-1961: POC lsb width (tool_pocs), the sub-GOP description unless both tools are on, and either the sliding-window size or the RPL candidates
        s.tool_rpl = rpl; s.tool_pocs = pocs;
        s.log2_sub_gop = s.log2_ref_gap = 0;
        if (pocs) { s.poc_lsb_bits = (int)br.ue() + 4; if (s.poc_lsb_bits > 16) return fail("bad SPS: log2_max_pic_order_cnt_lsb"); }
        if (!rpl || !pocs) {
            s.log2_sub_gop = (int)br.ue();
            if (s.log2_sub_gop == 0) s.log2_ref_gap = (int)br.ue();
            if (s.log2_sub_gop > 5) return fail("bad SPS");
        }
        if (!rpl) s.max_num_ref_pics = (int)br.ue();
        else {
            s.max_num_ref_pics = std::min((int)br.ue() + 1, XGPU_MAX_REFS);      // sps_max_dec_pic_buffering_minus1
            br.get1();                                                          // long_term_ref_pics_flag (no syntax of its own in this decoder)
            if (br.get1()) return fail("rpl1_same_as_rpl0_flag is not supported (nor by the reference)");
            for (int l = 0; l < 2; l++) {
                s.n_rpl[l] = (int)br.ue();
                if (br.overrun || s.n_rpl[l] > 32) return fail("bad SPS: num_ref_pic_lists_in_sps");
                for (int i = 0; i < s.n_rpl[l]; i++) if (!read_rpl(br, s.rpls[l][i])) return fail("bad SPS: reference picture list");
            }
        }
        s.crop[0] = s.crop[1] = s.crop[2] = s.crop[3] = 0;
        if (br.get1()) for (int i = 0; i < 4; i++) s.crop[i] = (int)br.ue();      // left, right, top, bottom (handed to the caller's output stage)
        s.cqt = br.get1() != 0;
        if (s.cqt) {
            // chroma_qp_table_struct (xevd_eco.c:1361-1376) -> mapping tables (xevd_derived_chroma_qp_mapping_tables, xevd_tbl.c:375-425): pivot
            // points, linear interpolation with rounding between them, slope 1 outside, clipped to the QP range
            const int same = br.get1(), global_offset = br.get1(), off = 6 * (s.bd_c - 8), start = global_offset ? 16 : -off;
            for (int c = 0; c < (same ? 1 : 2); c++) {
                const int np = (int)br.ue() + 1;
                if (np < 1 || np > 58 + off) return fail("bad chroma QP table");
                int din[96], qin[96], qout[96];                      // np <= 58 + 24
                for (int j = 0; j < np; j++) {
                    din[j] = (int)br.get(6);
                    const int dout = br.se();
                    qin[j] = j ? qin[j - 1] + din[j] + 1 : start + din[0];
                    qout[j] = j ? qout[j - 1] + din[j] + 1 + dout : start + din[0] + dout;
                    if (qin[j] < -off || qin[j] > 57 || qout[j] < -off || qout[j] > 57) return fail("bad chroma QP table");
                }
                int8_t *t = s.cq[c] + off;                          // t[qp], qp = -off .. 57
                t[qin[0]] = (int8_t)qout[0];
                for (int k = qin[0] - 1; k >= -off; k--) t[k] = (int8_t)std::min(std::max(t[k + 1] - 1, -off), 57);
                for (int j = 0; j + 1 < np; j++) {
                    const int den = din[j + 1] + 1, rnd = den >> 1;
                    for (int k = qin[j] + 1, m = 1; k <= qin[j + 1]; k++, m++) t[k] = (int8_t)(t[qin[j]] + ((qout[j + 1] - qout[j]) * m + rnd) / den);
                }
                for (int k = qin[np - 1] + 1; k <= 57; k++) t[k] = (int8_t)std::min(std::max(t[k - 1] + 1, -off), 57);
            }
            if (same) memcpy(s.cq[1], s.cq[0], sizeof(s.cq[0]));
        }
        br.get1();      // vui_parameters_present_flag: the VUI (display metadata, xevd_eco.c:1226-1304) is the last SPS element and is not needed here
        if (br.overrun || (s.width & 7) || (s.height & 7) || s.width <= 0 || s.height <= 0 || s.width > 16384 || s.height > 16384 ||
            s.bd_l < 8 || s.bd_l > 12 || s.bd_c < 8 || s.bd_c > 12 || s.max_num_ref_pics < 0 || s.max_num_ref_pics > XGPU_MAX_REFS) return fail("bad SPS");
        // a new geometry / bit depth invalidates every stored picture (their motion fields have the old SCU grid): drop the DPB and take
        // nothing but an IDR picture until one arrives
        if (st.have_sps && (s.width != st.sps.width || s.height != st.sps.height || s.bd_l != st.sps.bd_l || s.bd_c != st.sps.bd_c)) {
            st.dpb.clear(); st.refp[0].clear(); st.refp[1].clear();
            st.need_idr = true;
        }
        st.sps = s;
        st.have_sps = true;
        return XGPU_OK;
    }
    int parse_pps(BitReader &br)
    {
        Pps q = st.pps;                                  // parsed into a copy: a damaged PPS leaves the active one untouched (as parse_sps does)
        br.ue(); br.ue();                                // pps id, sps id
        q.default_active[0] = (int)br.ue() + 1; q.default_active[1] = (int)br.ue() + 1;      // num_ref_idx_default_active_minus1
        br.ue();                                         // additional_lt_poc_lsb_len
        q.rpl1_idx_present = br.get1();
        q.tile_cols = q.tile_rows = q.tile_uniform = 1; q.across_tiles = 0;      // one tile: the flag is not sent and reads as 0
        if (!br.get1()) {                                // single_tile_in_pic_flag == 0 (xevdm_eco.c:2021-2039)
            q.tile_cols = (int)br.ue() + 1; q.tile_rows = (int)br.ue() + 1;
            if (br.overrun || q.tile_cols > XGPU_MAX_TILE_COLS || q.tile_rows > XGPU_MAX_TILE_ROWS) return fail("bad PPS: tile grid");
            q.tile_uniform = br.get1();
            if (!q.tile_uniform) {
                for (int i = 0; i + 1 < q.tile_cols; i++) q.tile_col_w[i] = (int)br.ue() + 1;
                for (int i = 0; i + 1 < q.tile_rows; i++) q.tile_row_h[i] = (int)br.ue() + 1;
            }
            q.across_tiles = br.get1();
            q.offset_bits = (int)br.ue() + 1;
            if (q.offset_bits > 32) return fail("bad PPS: tile_offset_lens_minus1");
        }
        q.id_bits = (int)br.ue() + 1;                    // tile_id_len_minus1
        if (q.id_bits > 15) return fail("bad PPS: tile_id_len_minus1");
        if (br.get1())                                   // explicit_tile_id_flag: tile_id_val[row][col] (xevdm_eco.c:2042-2052).  The reference decoder reads the values and never
            for (int i = 0; i < q.tile_cols * q.tile_rows; i++) br.get(q.id_bits);      // looks at them again - first / last_tile_id stay raster indices (set_tile_info) - so neither do we
        q.dra_on = br.get1();                       // pic_dra_enabled_flag, pic_dra_aps_id (xevdm_eco.c:2054-2060)
        if (q.dra_on) q.dra_aps_id = (int)br.get(5);
        q.arbitrary_slices = br.get1();                  // arbitrary_slice_present_flag
        q.constrained_intra = br.get1();
        q.cu_qp_delta = br.get1();
        q.qp_delta_area = 6;
        if (q.cu_qp_delta) { q.qp_delta_area = (int)br.ue() + 6; if (q.qp_delta_area > 14) return fail("bad PPS: cu_qp_delta_area"); }
        if (br.overrun) return fail("bad PPS");
        st.pps = q;
        st.have_pps = true;
        return XGPU_OK;
    }
    // the tiles of a slice, in the order of their entry points (xevdm_eco.c:2520-2550, set_tile_info src_main/xevdm.c:2185-2236): one tile, the rectangle of
    // tiles between first_tile_id and last_tile_id (it may wrap around the picture's right / bottom border), or an arbitrary ascending list.  br stands behind
    // slice_pic_parameter_set_id
    int slice_tile_list(BitReader &br, std::vector<int> &tl)
    {
        const int n_tiles = st.pps.tile_cols * st.pps.tile_rows;
        tl.assign(1, 0);
        if (n_tiles > 1) {
            const int single = br.get1(), first = (int)br.get(st.pps.id_bits);
            if (first >= n_tiles) return fail("bad slice header: first_tile_id");
            tl[0] = first;
            if (!single) {
                const int arbitrary = st.pps.arbitrary_slices ? br.get1() : 0;
                if (!arbitrary) {
                    const int last = (int)br.get(st.pps.id_bits), wt = st.pps.tile_cols, ht = st.pps.tile_rows;
                    if (last >= n_tiles) return fail("bad slice header: last_tile_id");
                    int delta = last - first;
                    if (last < first) delta += first % wt > last % wt ? n_tiles + wt : n_tiles;
                    else if (first % wt > last % wt) delta += wt;
                    const int ws = delta % wt + 1, hs = delta / wt + 1;
                    if (ws > wt || hs > ht) return fail("bad slice header: tile rectangle");
                    tl.clear();
                    for (int r = 0; r < hs; r++) for (int c2 = 0; c2 < ws; c2++) tl.push_back(((first / wt + r) % ht) * wt + (first % wt + c2) % wt);
                } else {
                    const uint32_t more = br.ue() + 1;             // num_remaining_tiles_in_slice_minus1 + 1
                    if (br.overrun || more >= (uint32_t)n_tiles) return fail("bad slice header: arbitrary slice");
                    for (uint32_t i = 0; i < more; i++) {
                        const uint32_t d = br.ue();
                        if (br.overrun || d >= (uint32_t)n_tiles || tl.back() + (int)d + 1 >= n_tiles) return fail("bad slice header: delta_tile_id");
                        tl.push_back(tl.back() + (int)d + 1);
                    }
                }
            }
        }
        return XGPU_OK;
    }
    int parse_slice(BitReader &br, int nut, int tid, xhost_picture *out)
    {
        if (!st.have_sps || !st.have_pps) return fail("slice before SPS/PPS");
        static const bool tr_on = getenv("XEVD_HOST_TRACE") != NULL;      // phase times of the front end on stderr
        auto tr_t0 = std::chrono::steady_clock::now();
        auto TR = [&](const char *what) { if (tr_on) { const auto t = std::chrono::steady_clock::now(); fprintf(stderr, "  parse: %-14s %.2f ms\n", what, std::chrono::duration<double, std::milli>(t - tr_t0).count()); tr_t0 = t; } };
        if (st.need_idr && nut != NUT_IDR) return fail("the sequence parameters changed: waiting for an IDR picture");
        st.need_idr = false;
        Slice &sh = st.sh;
        br.ue();                                         // slice_pic_parameter_set_id
        const int n_tiles = st.pps.tile_cols * st.pps.tile_rows;
        if (n_tiles > 1 && !st.sps.profile_main) return fail("tiles in a Baseline stream are not supported");
        std::vector<int> tl;
        { const int rc = slice_tile_list(br, tl); if (rc != XGPU_OK) return rc; }
        const int n_slice_tiles = (int)tl.size();
        const bool first_slice = pic_tiles_left == 0;
        if (!first_slice) {
            // The reference decoder takes slice NALs as parts of one picture until every CTU is covered (ctx->num_ctb, src_main/xevdm.c:2995-2999, 3106, 3138); without
            // sps_pocs_flag it derives a NEW picture order count for every slice NAL (xevd_poc_derivation is called per NAL, :3030-3040), so such streams need the flag
            if (!st.sps.tool_pocs) return fail("several slices per picture need sps_pocs_flag (the reference decoder counts a picture per slice NAL otherwise)");
            if (nut != pic_nut || tid != pic_tid) return fail("the slices of a picture differ in NAL unit type or temporal id");
        }
        for (int t : tl) if (!first_slice && tile_done[(size_t)t]) return fail("a tile is coded twice in one picture");
        if (!first_slice && n_slice_tiles > pic_tiles_left) return fail("a tile is coded twice in one picture");
        sh.type = (int)br.ue();
        if (sh.type < 0 || sh.type > 2) return fail("bad slice type");
        if (nut == NUT_IDR) br.get1();                   // no_output_of_prior_pics_flag
        sh.mmvd_group = (st.sps.tool_mmvd && sh.type != XHOST_SLICE_I) ? br.get1() : 0;
        sh.alf_on = sh.alf_chroma_idc = sh.alf_ctb_map = 0;
        if (st.sps.tool_alf) {                           // xevdm_eco.c:2608-2657 (4:2:0)
            sh.alf_on = br.get1();
            if (sh.alf_on) {
                sh.aps_id_y = (int)br.get(5);
                sh.alf_ctb_map = br.get1();
                sh.alf_chroma_idc = (int)br.get(2);
                if (sh.alf_chroma_idc) sh.aps_id_ch = (int)br.get(5);
            }
        }
        sh.poc_lsb = 0; sh.rpl[0] = Rpl(); sh.rpl[1] = Rpl();
        if (nut != NUT_IDR) {                            // xevdm_eco.c:2658-2733
            if (st.sps.tool_pocs) sh.poc_lsb = (int)br.get(st.sps.poc_lsb_bits);
            if (st.sps.tool_rpl) {
                int from_sps[2] = { 0, 0 }, idx[2] = { 0, 0 };
                for (int l = 0; l < 2; l++) {
                    if (l == 0 || st.pps.rpl1_idx_present) from_sps[l] = st.sps.n_rpl[l] > 0 ? br.get1() : 0;
                    else from_sps[1] = from_sps[0];
                    if (from_sps[l]) {
                        if (l == 0 || st.pps.rpl1_idx_present) { if (st.sps.n_rpl[l] > 1) idx[l] = (int)br.ue(); }
                        else idx[1] = idx[0];
                        // (the reference copies list 0 only when the SPS holds more than one candidate - with exactly one it keeps the previous slice's list)
                        if (idx[l] >= st.sps.n_rpl[l] || (l == 0 && st.sps.n_rpl[0] == 1)) return fail("reference picture list index");
                        sh.rpl[l] = st.sps.rpls[l][idx[l]];
                    } else if (!read_rpl(br, sh.rpl[l])) return fail("bad slice header: reference picture list");
                }
            }
        }
        sh.rpl[0].active = st.pps.default_active[0]; sh.rpl[1].active = st.pps.default_active[1];
        if (sh.type != XHOST_SLICE_I && br.get1()) { sh.rpl[0].active = (int)br.ue() + 1; if (sh.type == XHOST_SLICE_B) sh.rpl[1].active = (int)br.ue() + 1; }      // num_ref_idx_active override (only used with tool_rpl)
        sh.tmvp_assigned = sh.col_list = sh.col_src_list = sh.col_ref = 0;
        if (sh.type != XHOST_SLICE_I && st.sps.tool_admvp && (sh.tmvp_assigned = br.get1())) {                // xevdm_eco.c:2748-2760
            if (sh.type == XHOST_SLICE_B) { sh.col_list = br.get1(); sh.col_src_list = br.get1(); }
            sh.col_ref = br.get1();
        }
        sh.deblock = br.get1();
        sh.alpha_off = sh.beta_off = 0;
        if (sh.deblock && st.sps.tool_addb) { sh.alpha_off = br.se(); sh.beta_off = br.se(); }      // xevdm_eco.c:2767-2772
        sh.qp = (int)br.get(6);
        sh.qp_u_offset = br.se(); sh.qp_v_offset = br.se();
        std::vector<size_t> tile_size((size_t)n_slice_tiles, 0);      // entry_point_offset_minus1 + 1: bytes of every tile of the slice but the last (xevdm_eco.c:2789-2795)
        for (int i = 0; i + 1 < n_slice_tiles; i++) tile_size[(size_t)i] = (size_t)br.get(st.pps.offset_bits) + 1;
        while (!br.aligned()) if (br.get1()) return fail("slice header alignment");
        if (br.overrun || sh.qp > 51) return fail("bad slice header");
        st.derive_poc(nut == NUT_IDR, tid);
        if (!first_slice && st.poc != pic_poc) return fail("a slice of another picture before every tile of the picture was coded");
        if (sh.type == XHOST_SLICE_I) st.last_intra_poc = st.poc;
        if (!st.build_ref_lists(nut == NUT_IDR, !first_slice)) return fail("a reference picture list names a picture that is not in the DPB");
        if (sh.type != XHOST_SLICE_I && st.refp[0].empty()) return fail("P/B slice without a reference picture");
        if (sh.type == XHOST_SLICE_B && st.refp[1].empty()) return fail("B slice without a list-1 reference picture");
        for (int l = 0; l < 2; l++)
            for (const RefPic *r : st.refp[l])
                if (r->mv.size() != (size_t)(st.sps.width >> 2) * (st.sps.height >> 2) * 4) return fail("reference picture of another geometry");
        const int W = st.sps.width, H = st.sps.height, w_ctu = (W + 63) >> 6, h_ctu = (H + 63) >> 6;
        if (first_slice) {
            st.pic.size(st.sps.width, st.sps.height, st.sps.host_dmvr());
            {       // the maps' start state, in row bands on the tile threads
                const int bands = std::max(1, std::min(n_threads, st.pic.h_scu / 16));
                parallel_for(bands, [&](int k) { st.pic.clear_rows((int)((long long)st.pic.h_scu * k / bands), (int)((long long)st.pic.h_scu * (k + 1) / bands)); });
            }
            if (!st.setup_tiles()) return fail("the tile grid of the PPS does not fit the picture");
            st.alf_ctb_flag.assign((size_t)w_ctu * h_ctu, 1);
            tile_done.assign((size_t)n_tiles, 0);
            pic_nut = nut; pic_tid = tid; pic_poc = st.poc; pic_inter = 0; pic_qp = sh.qp;
        }
        TR("header + reset");
        // The backend takes ONE pair of reference lists per picture (and the reference decoder deblocks the whole picture with the lists of its last P / B slice,
        // ctx->refp at src_main/xevdm.c:3138-3199): the P / B slices of a picture must name the same pictures; I slices may be mixed in
        if (sh.type != XHOST_SLICE_I) {
            std::vector<int> lp;
            for (int l = 0; l < 2; l++) { lp.push_back(-1 - l); for (const RefPic *r : st.refp[l]) lp.push_back(r->poc); }
            if (pic_inter && lp != pic_lists) return fail("slices of one picture with different reference picture lists are not supported");
            pic_lists = lp; pic_inter = 1;
        }
        if (st.sps.tool_htdf && sh.qp != pic_qp) return fail("slices of one picture with different slice QPs together with HTDF are not supported");

        // ---- tile data (xevdm_dec_slice + xevd_tile_eco, src_main/xevdm.c:2363-2461, 2614-2718): every tile is its own arithmetic-coder
        //      run - contexts, QP predictor and motion history start afresh - at the byte offset the slice header gave; tiles share nothing but
        //      the picture maps (disjoint regions), so they are parsed in parallel when the caller allows threads ----
        while ((int)tiles.size() < n_tiles) tiles.emplace_back(new TileParser(st));
        if (luma_wait) {                                     // planes posted since the last slice go to their DPB entries; the others are waited for where a CU needs them
            st.take_posted_luma(board);
            for (auto &t : tiles) { t->tc.board = &board; t->tc.posted_serial[0] = t->tc.posted_serial[1] = 0; }
        }
        std::vector<size_t> tile_pos((size_t)n_slice_tiles, br.pos);
        for (int t = 1; t < n_slice_tiles; t++) tile_pos[(size_t)t] = tile_pos[(size_t)t - 1] + tile_size[(size_t)t - 1] * 8;
        std::vector<int> tile_rc((size_t)n_slice_tiles, XGPU_OK);
        parallel_for(n_slice_tiles, [&](int i) { const int t = tl[(size_t)i]; tile_rc[(size_t)i] = tiles[(size_t)t]->parse_tile(br, tile_pos[(size_t)i], t % st.grid.n_cols, t / st.grid.n_cols); });
        TR("tiles");
        for (int i = 0; i < n_slice_tiles; i++) if (tile_rc[(size_t)i] != XGPU_OK) { err = tiles[(size_t)tl[(size_t)i]]->err; pic_tiles_left = 0; return tile_rc[(size_t)i]; }
        if (first_slice) pic_tiles_left = n_tiles;
        for (int t : tl) tile_done[(size_t)t] = 1;
        pic_tiles_left -= n_slice_tiles;
        if (pic_tiles_left > 0) return XGPU_OK;                // more slices of this picture follow; the in-loop filters run with the LAST slice's header, as in the
                                                                 // reference decoder (ctx->sh at src_main/xevdm.c:3138-3199: deblocking switch and offsets, chroma QP offsets, ALF)
        merged_arena = nullptr;
        if (n_tiles == 1) { cur = &tiles[0]->batch; n_coef = tiles[0]->n_coef; }
        else {
            // one batch for the backend: the tiles' arrays one after the other, coefficient offsets and CTU starts moved along
            std::vector<size_t> cu0((size_t)n_tiles + 1, 0), cf0((size_t)n_tiles + 1, 0), ct0((size_t)n_tiles + 1, 0);
            for (int t = 0; t < n_tiles; t++) {
                cu0[(size_t)t + 1] = cu0[(size_t)t] + tiles[(size_t)t]->batch.x.size();
                cf0[(size_t)t + 1] = cf0[(size_t)t] + tiles[(size_t)t]->n_coef;
                ct0[(size_t)t + 1] = ct0[(size_t)t] + tiles[(size_t)t]->batch.ctu_start.size();
            }
            if (cf0[(size_t)n_tiles] > 0xFFFFFFFFull) return fail("coefficient arena beyond 32-bit offsets");
            Batch &m = merged;
            const size_t n = cu0[(size_t)n_tiles];
            // the caller's arena (pinned memory of the backend: xgpu_batch_create then sends the coefficients from where they lie) of the slot this picture
            // will be handed out in; without one - or when the allocator has nothing yet - the merged batch's own vector
            Held &slot = held[n_handed % held.size()];
            int16_t *arena = nullptr;
            if (arena_alloc) {
                const size_t need = std::max(cf0[(size_t)n_tiles], (size_t)8);
                if (slot.arena_cap < need) {
                    if (slot.arena && arena_release) arena_release(arena_user, slot.arena);
                    slot.arena_cap = need + need / 4;
                    slot.arena = (int16_t *)arena_alloc(arena_user, slot.arena_cap * sizeof(int16_t));
                    if (!slot.arena) slot.arena_cap = 0;
                }
                arena = slot.arena;
            }
            m.x.resize(n); m.y.resize(n); m.log2w.resize(n); m.log2h.resize(n); m.pred_mode.resize(n); m.qp.resize(n * 3); m.cbf.resize(n); m.ipm.resize(n * 2);
            m.ats.resize(n); m.ats_inter.resize(n); m.dmvr.resize(n); m.tree.resize(n); m.has_tree = false;
            for (int t = 0; t < n_tiles; t++) m.has_tree |= tiles[(size_t)t]->batch.has_tree;
            m.affine.resize(st.sps.tool_affine ? n : 0); m.affine_mv.resize(st.sps.tool_affine ? n * 12 : 0); m.refi.resize(n * 2); m.mv.resize(n * 4); m.coef_off.resize(n);
            if (!arena) m.coef.resize(cf0[(size_t)n_tiles]);
            m.ctu_start.resize(ct0[(size_t)n_tiles]);
            int16_t *coef_dst = arena ? arena : m.coef.data();
            merged_arena = arena;
            parallel_for(n_tiles, [&](int t) {
                const Batch &b = tiles[(size_t)t]->batch;
                const size_t o = cu0[(size_t)t], k = b.x.size();
                auto put = [&](auto &dst, const auto &src, size_t per) { if (k) memcpy(dst.data() + o * per, src.data(), k * per * sizeof(src[0])); };
                put(m.x, b.x, 1); put(m.y, b.y, 1); put(m.log2w, b.log2w, 1); put(m.log2h, b.log2h, 1); put(m.pred_mode, b.pred_mode, 1); put(m.qp, b.qp, 3);
                put(m.cbf, b.cbf, 1); put(m.ipm, b.ipm, 2); put(m.ats, b.ats, 1); put(m.ats_inter, b.ats_inter, 1); put(m.dmvr, b.dmvr, 1); put(m.tree, b.tree, 1); put(m.refi, b.refi, 2); put(m.mv, b.mv, 4);
                if (st.sps.tool_affine) { put(m.affine, b.affine, 1); put(m.affine_mv, b.affine_mv, 12); }
                for (size_t i = 0; i < k; i++) m.coef_off[o + i] = b.coef_off[i] + (uint32_t)cf0[(size_t)t];
                if (tiles[(size_t)t]->n_coef) memcpy(coef_dst + cf0[(size_t)t], b.coef.data(), tiles[(size_t)t]->n_coef * sizeof(int16_t));
                for (size_t i = 0; i < b.ctu_start.size(); i++) m.ctu_start[ct0[(size_t)t] + i] = b.ctu_start[i] + (uint32_t)o;
            });
            cur = &merged; n_coef = cf0[(size_t)n_tiles];
        }
        TR("merge");
        cur->ctu_start.push_back((uint32_t)cur->x.size());
        Held &hd = held[n_handed++ % held.size()];
        std::swap(hd.batch, *cur);                       // the parser's working vectors take over the slot's old storage (cleared / resized at their next use)
        cur = &hd.batch;
        Batch &batch = *cur;

        // ---- hand-over ----
        memset(out, 0, sizeof(*out));
        out->width = W; out->height = H; out->bit_depth_luma = st.sps.bd_l; out->bit_depth_chroma = st.sps.bd_c;
        out->poc = st.poc; out->temporal_id = tid; out->slice_type = sh.type; out->is_idr = nut == NUT_IDR; out->is_ref = st.is_ref_picture();
        for (int l = 0; l < 2; l++) {
            out->num_refp[l] = (int)st.refp[l].size();
            for (size_t i = 0; i < st.refp[l].size(); i++) out->refp_poc[i][l] = st.refp[l][i]->poc;
        }
        out->slice_qp = sh.qp; out->qp_u_offset = sh.qp_u_offset; out->qp_v_offset = sh.qp_v_offset; out->deblock_on = sh.deblock;
        out->profile_main = st.sps.profile_main; out->tool_iqt = st.sps.tool_iqt; out->tool_ats = st.sps.tool_ats; out->tool_addb = st.sps.tool_addb;
        out->deblock_alpha_offset = sh.alpha_off; out->deblock_beta_offset = sh.beta_off;
        out->tool_alf = st.sps.tool_alf; out->alf_on = sh.alf_on; out->tool_eipd = st.sps.tool_eipd; out->tool_admvp = st.sps.tool_admvp;
        out->dra_lut[0] = out->dra_lut[1] = out->dra_lut[2] = nullptr;
        if (st.sps.tool_dra && st.pps.dra_on) {          // what xevd_pull applies to its copy of this picture (xevd_apply_filter, xevdm.c:3305-3349)
            const DraAps &d = st.dra_aps[st.pps.dra_aps_id & 31];
            if (!d.valid) return fail("the PPS names a DRA parameter set that was not sent");
            if (st.sps.bd_l > 10) return fail("DRA tables cover 10 bits");
            const int off = 6 * (st.sps.bd_c - 8);
            int8_t dflt[2][96];
            const int8_t *cq[2];
            for (int c = 0; c < 2; c++) {
                if (st.sps.cqt) cq[c] = st.sps.cq[c] + off;
                else { for (int q = -off; q <= 57; q++) dflt[c][q + off] = q >= 0 ? (st.sps.tool_iqt ? k_chroma_qp_main : k_chroma_qp)[q] : 0; cq[c] = dflt[c] + off; }
            }
            hd.dra.resize(3 * 1024);
            dra_build_luts(d, st.sps.bd_l, cq[0], cq[1], off, hd.dra.data());
            for (int c = 0; c < 3; c++) out->dra_lut[c] = hd.dra.data() + 1024 * c;
        }
        for (int i = 0; i < 4; i++) out->crop[i] = st.sps.crop[i];
        out->chroma_qp_table[0] = st.sps.cqt ? st.sps.cq[0] : nullptr; out->chroma_qp_table[1] = st.sps.cqt ? st.sps.cq[1] : nullptr;
        if (sh.alf_on) {
            if (!st.alf_finalise()) return fail("slice refers to an ALF parameter set that was not sent");
            out->alf.enable[0] = 1; out->alf.enable[1] = sh.alf_chroma_idc & 1; out->alf.enable[2] = (sh.alf_chroma_idc >> 1) & 1;
            memcpy(hd.alf_luma, st.alf_luma_final, sizeof(hd.alf_luma)); memcpy(hd.alf_chroma, st.alf_chroma_final, sizeof(hd.alf_chroma));
            hd.ctb = st.alf_ctb_flag;
            out->alf.luma_coef = &hd.alf_luma[0][0]; out->alf.chroma_coef = hd.alf_chroma;
            out->alf.ctb_flag = hd.ctb.data(); out->alf.across_tiles = st.pps.across_tiles; out->alf.tiles = n_tiles > 1 ? &hd.grid : nullptr;
        }
        std::vector<int> released;
        TR("hand-over");
        st.store_picture(nut == NUT_IDR, released, pic_inter);
        TR("store picture");
        out->n_release = (int)std::min(released.size(), (size_t)32);
        for (int i = 0; i < out->n_release; i++) out->release_poc[i] = released[(size_t)i];
        xgpu_cu_batch &b = out->batch;
        b.n_cu = (int)batch.x.size();
        b.x = batch.x.data(); b.y = batch.y.data(); b.log2w = batch.log2w.data(); b.log2h = batch.log2h.data();
        b.pred_mode = batch.pred_mode.data(); b.refi = batch.refi.data(); b.mv = batch.mv.data(); b.qp = batch.qp.data();
        b.cbf = batch.cbf.data(); b.ipm = batch.ipm.data(); b.coef_off = batch.coef_off.data();
        if (st.sps.tool_ats) { b.ats = batch.ats.data(); b.ats_inter = batch.ats_inter.data(); }
        if (st.sps.tool_affine) { b.affine = batch.affine.data(); b.affine_mv = batch.affine_mv.data(); }
        if (batch.has_tree) b.tree = batch.tree.data();
        if (batch.coef.empty()) batch.coef.push_back(0);
        b.coef = merged_arena ? merged_arena : batch.coef.data(); b.n_coef = batch.x.empty() ? 0 : n_coef;
        b.n_ctu = w_ctu * h_ctu; b.ctu_cu_start = batch.ctu_start.data();
        b.constrained_intra_pred = st.pps.constrained_intra;
        hd.grid = st.grid;
        b.tiles = n_tiles > 1 ? &hd.grid : nullptr;
        b.htdf_slice_qp = st.sps.tool_htdf ? sh.qp : 0;
        // sps->tool_dmvr: the merge-mode flags, and how many 16x16 sub-blocks of candidates (flag, two references, at least 8x8 - the order of
        // xgpu_batch_dmvr_mvs) the backend will report vectors for; they go back in through xhost_parser_set_dmvr_mvs before the next picture
        out->n_dmvr_sub = 0;
        last_poc = st.poc; last_stored = st.is_ref_picture();
        out->needs_ref_luma = st.sps.host_dmvr() && st.is_ref_picture();
        if (out->needs_ref_luma && luma_wait) board.expect(st.pic_serial, st.poc);
        if (st.sps.tool_dmvr) b.dmvr = batch.dmvr.data();
        if (st.sps.tool_dmvr && !st.sps.host_dmvr()) {      // (host-side refinement: nothing comes back from the backend)
            for (int i = 0; i < b.n_cu; i++)
                if (batch.dmvr[(size_t)i] && batch.refi[(size_t)i * 2] >= 0 && batch.refi[(size_t)i * 2 + 1] >= 0 && batch.log2w[(size_t)i] >= 3 && batch.log2h[(size_t)i] >= 3)
                    out->n_dmvr_sub += (batch.log2w[(size_t)i] > 4 ? 1 << (batch.log2w[(size_t)i] - 4) : 1) * (batch.log2h[(size_t)i] > 4 ? 1 << (batch.log2h[(size_t)i] - 4) : 1);
        }
        last_n_dmvr = out->n_dmvr_sub;
        return 1;
    }
    int last_poc = 0, last_n_dmvr = 0;
    bool last_stored = false;
    // the picture being assembled from several slice NALs: tiles still to come, and what its slices must agree on
    int pic_tiles_left = 0, pic_nut = 0, pic_tid = 0, pic_poc = 0, pic_inter = 0, pic_qp = 0;
    std::vector<uint8_t> tile_done;
    std::vector<int> pic_lists;
    // the refined vectors of the picture just handed out: what the reference keeps in map_mv for the temporal candidates of later pictures
    // (dmvr_mv -> map_mv, src_main/xevdm_util.c:4327-4338); the picture's own CUs, the history and the deblocking filter use the unrefined ones
    int set_dmvr_mvs(const int16_t *mv, int n)
    {
        if (n != last_n_dmvr || (n > 0 && !mv)) return fail("xhost_parser_set_dmvr_mvs: not the sub-block count of the last picture");
        if (!last_stored || n == 0 || st.sps.host_dmvr()) return 0;
        RefPic *r = nullptr;
        for (RefPic &q : st.dpb) if (q.poc == last_poc) r = &q;
        if (!r || r->mv.empty()) return 0;
        const int ws = st.sps.width >> 2;
        const Batch &batch = *cur;
        for (size_t i = 0; i < batch.x.size(); i++) {
            if (!(batch.dmvr[i] && batch.refi[i * 2] >= 0 && batch.refi[i * 2 + 1] >= 0 && batch.log2w[i] >= 3 && batch.log2h[i] >= 3)) continue;
            const int w = 1 << batch.log2w[i], h = 1 << batch.log2h[i], dx = std::min(w, 16), dy = std::min(h, 16);
            for (int sy = 0; sy < h; sy += dy) for (int sx = 0; sx < w; sx += dx, mv += 4)
                for (int v = 0; v < dy >> 2; v++) for (int u = 0; u < dx >> 2; u++)
                    memcpy(&r->mv[((size_t)((batch.y[i] + sy) >> 2) + v) * ws * 4 + ((size_t)((batch.x[i] + sx) >> 2) + u) * 4], mv, sizeof(int16_t) * 4);
        }
        return 0;
    }
};

extern "C" xhost_parser *xhost_parser_open(const uint8_t *bytes, size_t size)
{
    if (!bytes) return nullptr;
    xhost_parser *p = new xhost_parser();
    p->data.assign(bytes, bytes + size);
    return p;
}
extern "C" int xhost_parser_rebind(xhost_parser *p, const uint8_t *bytes, size_t size)
{
    if (!p || !bytes) return XGPU_ERR_INVALID_ARGUMENT;
    Stream &st = p->st;
    while (!st.dpb.empty()) st.drop_ref(st.dpb.size() - 1);      // the motion fields go to the pools
    Picture pic = std::move(st.pic);
    std::vector<std::vector<int16_t>> mvp = std::move(st.mv_pool);
    std::vector<std::vector<int8_t>> rfp = std::move(st.refi_pool);
    st.~Stream();                                                // every parameter set, POC / reference-list / ALF / DRA state: as in a new parser
    new (&st) Stream();
    st.pic = std::move(pic); st.mv_pool = std::move(mvp); st.refi_pool = std::move(rfp);
    // the arenas go back to the caller (who may hand the same memory out again, or replace what it was allocated from before the next unit's pictures need one)
    for (xhost_parser::Held &h : p->held) { if (h.arena && p->arena_release) p->arena_release(p->arena_user, h.arena); h.arena = nullptr; h.arena_cap = 0; }
    p->merged_arena = nullptr;
    p->data.assign(bytes, bytes + size);
    p->pos = 0; p->err.clear();
    p->pic_tiles_left = 0; p->last_poc = 0; p->last_n_dmvr = 0; p->last_stored = false;
    p->tile_done.clear(); p->pic_lists.clear();
    p->board.reset();
    return XGPU_OK;
}
// the decoded luma samples of the picture with this POC, for the front end's own refinement search (Sps::host_dmvr)
extern "C" int xhost_parser_set_ref_luma(xhost_parser *p, int poc, const int16_t *plane, int stride)
{
    if (!p) return XGPU_ERR_INVALID_ARGUMENT;
    if (!p->luma_wait) return set_ref_luma(p->st, poc, plane, stride);
    if (!plane || stride <= 0) return XGPU_ERR_INVALID_ARGUMENT;
    return p->board.post(poc, plane, stride);            // possibly from another thread, while xhost_parser_next runs
}
extern "C" int xhost_parser_set_ref_luma_wait(xhost_parser *p, int on)
{
    if (!p || p->n_handed) return XGPU_ERR_INVALID_ARGUMENT;      // before the first picture
    p->luma_wait = on != 0;
    return XGPU_OK;
}
extern "C" void xhost_parser_cancel_wait(xhost_parser *p) { if (p) p->board.cancel(); }
extern "C" int xhost_dmvr_search(int pic_w, int pic_h, int bit_depth, int x, int y, int w, int h, const int16_t mv[4],
                                 const int16_t *ref0, int stride0, const int16_t *ref1, int stride1, int16_t *refined)
{
    if (!mv || !ref0 || !ref1 || !refined || w < 8 || h < 8 || w > 128 || h > 128) return XGPU_ERR_INVALID_ARGUMENT;
    const int16_t m[2][2] = { { mv[0], mv[1] }, { mv[2], mv[3] } };
    const DmvrRefPlane rp[2] = { { ref0, stride0, 0 }, { ref1, stride1, 0 } };
    const int n = (w > 16 ? w / 16 : 1) * (h > 16 ? h / 16 : 1);
    static thread_local std::vector<int16_t> scratch;
    dmvr_search_cu(pic_w, pic_h, bit_depth, x, y, w, h, m, rp, (int16_t (*)[2][2])refined, scratch);
    return n;
}
extern "C" int xhost_parser_set_depth(xhost_parser *p, int depth)
{
    if (!p || depth < 1 || depth > 8 || p->n_handed) return XGPU_ERR_INVALID_ARGUMENT;      // before the first picture
    p->held = std::vector<xhost_parser::Held>((size_t)depth);
    return XGPU_OK;
}
extern "C" int xhost_parser_set_arena(xhost_parser *p, void *(*alloc)(void *, size_t), void (*release)(void *, void *), void *user)
{
    if (!p || !alloc || !release || p->n_handed) return XGPU_ERR_INVALID_ARGUMENT;
    p->arena_alloc = alloc; p->arena_release = release; p->arena_user = user;
    return XGPU_OK;
}
extern "C" int xhost_parser_set_threads(xhost_parser *p, int n) { if (!p || n < 1) return XGPU_ERR_INVALID_ARGUMENT; p->n_threads = std::min(n, 64); return XGPU_OK; }
extern "C" int xhost_parser_set_dmvr_mvs(xhost_parser *p, const int16_t *mv, int n_sub) { return p ? p->set_dmvr_mvs(mv, n_sub) : XHOST_ERR_MALFORMED; }
extern "C" const char *xhost_parser_error(const xhost_parser *p) { return p ? p->err.c_str() : "null parser"; }
extern "C" void xhost_parser_close(xhost_parser *p) { delete p; }

// one NAL unit (2-byte header + payload, no length prefix): 1 = `out` holds a picture, 0 = consumed without a picture, < 0 error
static int parser_nal(xhost_parser *p, const uint8_t *nal, size_t len, xhost_picture *out, bool peek_signature)
{
    if (len < 2) return p->fail("bad NAL length");
    BitReader br;
    br.p = nal; br.size = len;
    if (br.get1()) return p->fail("forbidden_zero_bit");
    const int nut = (int)br.get(6) - 1, tid = (int)br.get(3);
    if (br.get(5) != 0 || br.get1() != 0) return p->fail("reserved NAL header bits");
    // a picture is open between its first and its last slice NAL: the picture maps, the tile grid and the per-tile state were sized by the parameter sets of its
    // first slice, so a sequence or picture parameter set may not change under it (the partial picture is dropped with the error)
    if ((nut == NUT_SPS || nut == NUT_PPS) && p->pic_tiles_left > 0) { p->pic_tiles_left = 0; return p->fail("parameter set between the slices of a picture"); }
    if (nut == NUT_SPS) return p->parse_sps(br);
    if (nut == NUT_PPS) return p->parse_pps(br);
    if (nut == NUT_IDR || nut == NUT_NONIDR) {
        const int rc = p->parse_slice(br, nut, tid, out);
        if (rc == 1) { out->has_md5 = 0; if (peek_signature) p->attach_signature(out); }
        return rc;
    }
    if (nut == 26) {                                     // APS (xevdm_eco_aps_gen, xevdm_eco.c:2082-2135)
        const int id = (int)br.get(5), type = (int)br.get(3);
        if (type == 1) {                                 // DRA parameters (xevdm_eco_dra_aps_param, :2319-2375)
            if (!p->st.have_sps) return p->fail("DRA APS before the SPS");
            DraAps d;
            if (br.get(4) != 4 || br.get(4) != 9) return p->fail("unsupported DRA descriptors");
            d.num_ranges = (int)br.ue() + 1;
            if (d.num_ranges > 32) return p->fail("bad DRA APS");
            const int equal = br.get1(), sh = std::max(0, p->st.sps.bd_l - 10);
            d.in_ranges[0] = (int)br.get(10) << sh;
            int delta[32];
            if (equal) delta[0] = (int)br.get(10);
            else for (int i = 0; i < d.num_ranges; i++) delta[i] = (int)br.get(10);
            for (int i = 0; i < d.num_ranges; i++) d.scale[i] = (int)br.get(13);
            d.cb_scale = (int)br.get(13); d.cr_scale = (int)br.get(13);
            d.table_idx = (int)br.ue();
            for (int i = 1; i <= d.num_ranges; i++) d.in_ranges[i] = d.in_ranges[i - 1] + (delta[equal ? 0 : i - 1] << sh);
            if (br.overrun || d.table_idx > 58) return p->fail("bad DRA APS");
            d.valid = true;
            p->st.dra_aps[id] = d;
            return XGPU_OK;
        }
        if (type != 0) return p->fail("unknown APS type");
        AlfAps a;
        if (!p->st.alf_aps_syntax<false>(&br, nullptr, a)) return p->fail("bad ALF APS");
        a.valid = true;
        p->st.alf_aps[id] = a;
        return XGPU_OK;
    }
    if (nut == NUT_SEI || nut == 27) return XGPU_OK;     // SEI (picture signatures are the caller's to check), filler data
    return p->fail("unsupported NAL unit type");
}

extern "C" int xhost_parser_next(xhost_parser *p, xhost_picture *out)
{
    if (!p || !out) return XGPU_ERR_INVALID_ARGUMENT;
    while (p->pos + 4 <= p->data.size()) {
        const uint8_t *d = p->data.data() + p->pos;
        const size_t len = ((size_t)d[0] << 24) | ((size_t)d[1] << 16) | ((size_t)d[2] << 8) | d[3];
        if (len < 2 || p->pos + 4 + len > p->data.size()) return p->fail("bad NAL length");
        p->pos += 4 + len;
        const int rc = parser_nal(p, d + 4, len, out, true);
        if (rc != XGPU_OK) return rc;
    }
    if (p->pic_tiles_left > 0) { p->pic_tiles_left = 0; return p->fail("the stream ends inside a picture (slices missing)"); }
    return 0;
}

// Picture boundaries without decoding anything (the GOP splitter of the work queue, xwq.cc): a slice NAL belongs to the picture of the slice before it until
// the PPS's tiles are all covered (ctx->num_ctb, src_main/xevdm.c:2995-2999).  The scanner reads PPS NALs (tile grid) and the first fields of slice headers.
struct xhost_scan { xhost_parser p; int tiles_left = 0, last_nut = -1; std::vector<uint8_t> seen; };
extern "C" xhost_scan *xhost_scan_open(void) { return new xhost_scan(); }
extern "C" void xhost_scan_close(xhost_scan *s) { delete s; }
// -> 0: not a slice NAL, 1: the first (or only) slice of a picture, 2: a further slice of the picture, < 0: malformed
extern "C" int xhost_scan_nal(xhost_scan *s, const uint8_t *nal, size_t len)
{
    if (!s || !nal || len < 2) return XGPU_ERR_INVALID_ARGUMENT;
    BitReader br;
    br.p = nal; br.size = len;
    br.get1();
    const int nut = (int)br.get(6) - 1;
    br.get(3); br.get(5); br.get1();
    if (nut == NUT_PPS) { const int rc = s->p.parse_pps(br); return rc < 0 ? rc : 0; }
    if (nut != NUT_IDR && nut != NUT_NONIDR) return 0;
    br.ue();                                             // slice_pic_parameter_set_id
    std::vector<int> tl;
    const int rc = s->p.slice_tile_list(br, tl);
    if (rc != XGPU_OK || br.overrun) return XHOST_ERR_MALFORMED;
    // resynchronise after a lost slice: a slice of the other NAL type, or one that brings a tile the open picture already has, starts a new picture
    const int n_tiles = s->p.st.pps.tile_cols * s->p.st.pps.tile_rows;
    bool first = s->tiles_left <= 0 || nut != s->last_nut || (int)s->seen.size() != n_tiles;
    for (int t : tl) if (!first && (t < 0 || t >= n_tiles || s->seen[(size_t)t])) first = true;
    if (first) { s->tiles_left = n_tiles; s->seen.assign((size_t)n_tiles, 0); }
    for (int t : tl) if (t >= 0 && t < n_tiles) s->seen[(size_t)t] = 1;
    s->tiles_left -= (int)tl.size();
    s->last_nut = nut;
    return first ? 1 : 2;
}

// NAL-at-a-time interface (what xevd_decode takes: one NAL unit without its length prefix, src_base/xevd.c:1786-2024)
extern "C" xhost_parser *xhost_parser_open_nal(void) { return new xhost_parser(); }
extern "C" int xhost_parser_nal(xhost_parser *p, const uint8_t *nal, size_t size, xhost_picture *out)
{
    if (!p || !nal || !out) return XGPU_ERR_INVALID_ARGUMENT;
    return parser_nal(p, nal, size, out, false);
}

