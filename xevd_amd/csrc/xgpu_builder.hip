// xgpu_builder.hip - the host batch builder (xgpu_batch_create): validation, CU / TB records, owner map, the work lists of k_inter, the dependency plan of the
// order-dependent CUs (intra, IBC, HTDF), the pinned staging block and its upload.  Plain C++ (no kernel in this file); xgpu_test_build_batch runs it without a device.
#include <chrono>
#include "xgpu_host.h"
#include "affine_model.h"


// Intra CUs: availability masks, dependency lists and levels.  An SCU map of "CU index in decode order" stands in for the
// reference's COD flags: a neighbouring SCU is reconstructed at CU i's turn iff its CU index is below i (xevd_recon_unit
// sets COD CU by CU, xevd.c:744-754; xevd_get_avail_intra, xevd_util.c:689-745; single tile/slice).  The list is sorted by
// level (1 + the highest level among the intra CUs read), which is a topological order: every dependency sits earlier.
// HTDF (xevdm.c:1381-1392 with xevdm_htdf_skip_condition, xevdm_recon.c:270-297): which CUs are filtered right after their reconstruction, and with which of the five
// tables (-1: not filtered).  Such a CU - inter ones included - reads the final samples of the CUs before it and is read by the ones after it: it is a node of the
// dependency graph next to the intra and IBC CUs
static inline int plan_htdf_idx(const xgpu_cu_batch *b, uint32_t j)
{
    const int hqp = b->htdf_slice_qp;
    const bool intra = b->pred_mode[j] == XGPU_MODE_INTRA;
    if (hqp <= 17 || (b->tree && b->tree[j] == 2) || b->pred_mode[j] == XGPU_MODE_IBC || !((b->cbf[j] & 1) || intra)) return -1;
    const int w = 1 << b->log2w[j], h = 1 << b->log2h[j], mn = std::min(w, h), mx = std::max(w, h);
    if (w * h < 64 || mx >= 128 || (!intra && mn >= 32)) return -1;
    const int qp = hqp - ((intra && w == h && mn >= 32) ? 8 : 0);
    return std::min(std::max((qp - 20 + 4) >> 3, 0), 4);
}
static inline bool plan_is_node(const xgpu_cu_batch *b, uint32_t j) { return b->pred_mode[j] == XGPU_MODE_INTRA || b->pred_mode[j] == XGPU_MODE_IBC || plan_htdf_idx(b, j) >= 0; }

struct IntraPlan { std::vector<IntraRec> recs; std::vector<uint32_t> deps; int n_levels, n_level1, n_level1_small, n_heads; bool has_ibc, has_htdf, has_right; };      // n_heads: level-1 CUs + strand heads = the part of the list the launches range over
static bool build_intra_plan(xgpu_ctx *c, const xgpu_cu_batch *b, IntraPlan &plan, const uint32_t *final_owner, int nthr, WorkPool &pool, const std::vector<uint32_t> &nodes)
{
    const int hqp = b->htdf_slice_qp;
    auto tree_of = [&](uint32_t j) -> int { return b->tree ? b->tree[j] : 0; };      // local dual tree: 1 luma-only, 2 chroma-only CU
    auto htdf_idx = [&](uint32_t j) -> int { return plan_htdf_idx(b, j); };
    auto ordered = [&](uint32_t j) -> bool { return plan_is_node(b, j); };
    (void)hqp;
    const int n = b->n_cu, ws = c->w_scu, hs = c->h_scu;
    const uint32_t NONE = 0xFFFFFFFFu;
    // scratch of the builder thread, kept between pictures: an 8 MB vector per 8K picture allocated and freed every call goes through mmap / munmap, and the
    // munmap's TLB shootdown reaches every thread of the process - the parser's tile threads among them (examples/evc_decode.c runs them next to this one)
    // Batches without local dual trees (everything but BTT + ADMVP streams) take the FINAL SCU -> CU map the caller has already painted (in parallel, for k_inter):
    // "reconstructed before CU i" is then "owner index below i", no painting in step with the loop, and the nodes are independent of each other - built on the
    // builder's threads, levels assigned afterwards.  (At 8K this function was 11 of the builder's 15 ms, whatever the thread count.)
    const bool fast = final_owner != NULL && b->tree == NULL;
    static const bool pt_on = getenv("XEVD_HIP_BUILD_TRACE") != NULL;
    auto pt_t0 = std::chrono::steady_clock::now();
    auto PT = [&](const char *what) { if (pt_on) { const auto t = std::chrono::steady_clock::now(); fprintf(stderr, "    intra plan: %-12s %.2f ms\n", what, std::chrono::duration<double, std::milli>(t - pt_t0).count()); pt_t0 = t; } };
    static thread_local std::vector<uint32_t> owner_own;
    static thread_local std::vector<int> level;
    if (!fast) owner_own.assign((size_t)ws * hs, NONE);
    const uint32_t *const owner = fast ? final_owner : owner_own.data();
    level.assign((size_t)n, 0);
    int *const level_p = level.data();
    // painted CU by CU as the loop below reaches them ("reconstructed before CU i" = painted): inside a local dual tree the node's chroma-only CU follows its
    // luma CUs and covers them again
    // constrained intra prediction inside local dual trees: "is the neighbour intra-coded" is a property of the LUMA CU over the SCU (map_scu is written by the
    // luma CUs only) - an IBC luma CU under a chroma-only intra CU is not an intra neighbour.  The luma owners are kept apart from the repainted map for that test.
    const bool constrained_tree = b->constrained_intra_pred != 0 && b->tree != NULL;
    static thread_local std::vector<uint32_t> luma_owner;
    if (constrained_tree) luma_owner.assign((size_t)ws * hs, NONE);
    auto paint = [&](int i) {
        const int xs = b->x[i] >> 2, ys = b->y[i] >> 2, w = (1 << b->log2w[i]) >> 2, h = (1 << b->log2h[i]) >> 2;
        for (int r = 0; r < h; r++) std::fill_n(owner_own.begin() + (size_t)(ys + r) * ws + xs, w, (uint32_t)i);
        if (constrained_tree && b->tree[i] != 2)
            for (int r = 0; r < h; r++) std::fill_n(luma_owner.begin() + (size_t)(ys + r) * ws + xs, w, (uint32_t)i);
    };
    // tiles: a neighbour in another tile is not available (map_tidx[curr] == map_tidx[neighbour] in xevd_get_avail_intra, xevd_get_nbr_b, xevdm_get_nbr)
    const int ctu_sh = c->sp.log2_ctu - 2;
    std::vector<uint8_t> ctu_tile;
    if (b->tiles) {
        ctu_tile.assign((size_t)c->w_ctu * c->h_ctu, 0);
        for (int tj = 0; tj < b->tiles->n_rows; tj++) for (int ti = 0; ti < b->tiles->n_cols; ti++)
            for (int cy = b->tiles->row_bd[tj]; cy < b->tiles->row_bd[tj + 1]; cy++)
                for (int cx = b->tiles->col_bd[ti]; cx < b->tiles->col_bd[ti + 1]; cx++) ctu_tile[(size_t)cy * c->w_ctu + cx] = (uint8_t)(tj * b->tiles->n_cols + ti);
    }
    auto tile_of = [&](int sx, int sy) -> int { return ctu_tile.empty() ? 0 : ctu_tile[(size_t)(sy >> ctu_sh) * c->w_ctu + (sx >> ctu_sh)]; };
    const bool constrained = b->constrained_intra_pred != 0;
    // (the working arrays of the plan are the builder thread's own, kept between pictures: at 8K they are 6 - 7 MB each, and a fresh vector per call is an mmap, a
    //  page fault per 4 KB and a munmap whose TLB shootdown interrupts the parser's tile threads)
    static thread_local std::vector<IntraRec> recs_tl;      // decode order; dep lists hold CU indices until the sort below
    static thread_local std::vector<uint32_t> deps_tl;
    std::vector<IntraRec> &recs = recs_tl;
    std::vector<uint32_t> &deps = deps_tl;
    deps.clear();      // (recs: cleared by the sequential mode; the parallel mode resizes it - no re-initialisation of records that are overwritten anyway)
    int max_level = 0;
    // one CU: 0 = not a node, 1 = node appended to recs / deps (lv_out = its level when the levels of its dependencies are known: sequential mode), -1 = invalid batch
    // (r = where the node's record goes: a slot of the final list in the parallel mode, a temporary in the sequential one)
    auto make_node = [&](const int i, IntraRec &r, std::vector<uint32_t> &deps, bool &has_ibc, bool &has_htdf, bool &has_right, int &lv_out) -> int {
        if (!ordered((uint32_t)i)) return 0;
        const int xs = b->x[i] >> 2, ys = b->y[i] >> 2, units = ((1 << b->log2w[i]) + (1 << b->log2h[i])) >> 2;
        memset(&r, 0, sizeof(r));
        r.cu = (uint32_t)i; r.dep_first = (uint32_t)deps.size();
        r.x = b->x[i]; r.y = b->y[i]; r.log2w = b->log2w[i]; r.log2h = b->log2h[i]; r.cbf = b->cbf[i] & 7;
        if (b->ipm) { r.ipm[0] = b->ipm[i * 2]; r.ipm[1] = b->ipm[i * 2 + 1]; }
        r.coef_off = b->coef_off[i];
        int lv = 0;
        uint32_t last = NONE;
        const int hidx = htdf_idx((uint32_t)i);
        const bool h_intra = b->pred_mode[i] == XGPU_MODE_INTRA;
        const int my_tile = tile_of(xs, ys);
        // the border samples the filter reads (xevdm_htdf, xevdm_recon.c:299-385) with the availability of xevd_get_avail_intra (xevd_util.c:689-745):
        // "reconstructed" = earlier in decoding order
        auto add_htdf = [&](IntraRec &r) -> bool {
            const int scuw = (1 << b->log2w[i]) >> 2, scuh = (1 << b->log2h[i]) >> 2;
            auto cod = [&](int sx, int sy) -> bool { return owner[(size_t)sy * ws + sx] < (uint32_t)i && tile_of(sx, sy) == my_tile; };
            auto dep = [&](int sx, int sy) {
                const uint32_t j = owner[(size_t)sy * ws + sx];
                if (j >= (uint32_t)i) return;                       // not reconstructed yet: the reference reads what is there, so do we
                if (ordered(j)) {
                    bool seen = false;
                    for (size_t d = r.dep_first; d < deps.size() && !seen; d++) seen = deps[d] == j;
                    if (!seen) deps.push_back(j);
                }
                lv = std::max(lv, level_p[j]);
            };
            uint32_t av = 0;
            if (xs > 0 && cod(xs - 1, ys)) {
                av |= 1u << 1;
                if (ys + scuh + scuw - 1 < hs && cod(xs - 1, ys + scuh + scuw - 1)) av |= 1u << 7;
            }
            if (ys > 0) {
                if (tile_of(xs, ys - 1) == my_tile) av |= 1u << 0;
                if (xs > 0 && cod(xs - 1, ys - 1)) av |= 1u << 5;
                if (xs + scuw < ws && cod(xs + scuw, ys - 1)) av |= 1u << 6;
            }
            if (xs + scuw < ws && cod(xs + scuw, ys)) {
                av |= 1u << 3;
                if (ys + scuh + scuw - 1 < hs && cod(xs + scuw, ys + scuh + scuw - 1)) av |= 1u << 8;
            }
            if (av & 2u)  for (int k = 0; k < scuh; k++) dep(xs - 1, ys + k);
            if (av & 1u)  for (int k = 0; k < scuw; k++) dep(xs + k, ys - 1);
            if (av & 8u)  for (int k = 0; k < scuh; k++) dep(xs + scuw, ys + k);
            if (av & 32u) dep(xs - 1, ys - 1);
            if (av & 64u) dep(xs + scuw, ys - 1);
            if ((av & 128u) && ys + scuh < hs) dep(xs - 1, ys + scuh);
            if ((av & 256u) && ys + scuh < hs) dep(xs + scuw, ys + scuh);
            r.flags |= 4u | (av << 8) | ((uint32_t)hidx << 20) | ((h_intra && constrained) ? 16u : 0u);
            return true;
        };
        if (!h_intra && b->pred_mode[i] != XGPU_MODE_IBC) {
            // an inter CU that is only here for its filter: k_inter / k_affine have reconstructed it, the node filters it in place
            r.cbf = 0; r.ipm[0] = r.ipm[1] = 0;
            r.flags = 8u;
            if (!add_htdf(r)) return -1;
            r.dep_count = (uint32_t)deps.size() - r.dep_first;
            lv_out = lv + 1;
            has_ibc = true;                      // the instantiation with the extra node kinds
            has_htdf = true;
            return 1;
        }
        if (b->pred_mode[i] == XGPU_MODE_IBC) {
            // intra block copy: the CU waits for the intra / IBC CUs under its source block - the luma block at the vector plus, for an odd
            // vector, the sample column / row before it that the halved chroma vector reaches; all of it must precede the CU in decoding order
            const int bvx = b->mv[i * 4], bvy = b->mv[i * 4 + 1], w = 1 << b->log2w[i], h = 1 << b->log2h[i];
            const int x0 = b->x[i] + (bvx & ~1), x1 = b->x[i] + bvx + w - 1, y0 = b->y[i] + (bvy & ~1), y1 = b->y[i] + bvy + h - 1;
            r.ipm[0] = r.ipm[1] = 0;
            r.flags = 2u | (tree_of((uint32_t)i) == 1 ? 64u : 0u); r.le = (uint64_t)(uint16_t)bvx | ((uint64_t)(uint16_t)bvy << 16);
            for (int sy = y0 >> 2; sy <= y1 >> 2; sy++)
                for (int sx = x0 >> 2; sx <= x1 >> 2; sx++) {
                    const uint32_t j = owner[(size_t)sy * ws + sx];
                    if (j >= (uint32_t)i) return -1;
                    if (ordered(j) && j != last) {
                        bool seen = false;
                        for (size_t d = r.dep_first; d < deps.size() && !seen; d++) seen = deps[d] == j;
                        if (!seen) deps.push_back(j);
                        last = j;
                    }
                    lv = std::max(lv, level_p[j]);
                }
            r.dep_count = (uint32_t)deps.size() - r.dep_first;
            lv_out = lv + 1;
            has_ibc = true;
            return 1;
        }
        // which neighbour units the CU's predictors actually read (xevd_ipred.c:96-164,587-622): only those create a dependency;
        // the others are still fetched by the kernel (availability is about COD flags, not about use) but their values are ignored
        const int wu = (1 << b->log2w[i]) >> 2, hu = (1 << b->log2h[i]) >> 2;
        bool need_ul = false;
        int need_up = 0, need_le = 0;                                                      // number of leading units read on each side
        if (c->sp.tool_eipd) { need_up = need_le = units; need_ul = true; }            // EIPD modes: planar / bilinear / angular read both whole sides
        else for (int k = 0; k < 2; k++) {
            const int m = r.ipm[k];
            if (m == 0) { need_up = std::max(need_up, wu); need_le = std::max(need_le, hu); }
            else if (m == 1) need_le = std::max(need_le, hu);
            else if (m == 2) need_up = std::max(need_up, wu);
            else if (m == 3) { need_up = std::max(need_up, wu); need_le = std::max(need_le, hu); need_ul = true; }
            else { need_up = units; need_le = units; }
        }
        if (tree_of((uint32_t)i) == 1) r.flags |= 64u;              // luma only: the chroma samples stay as they are
        if (tree_of((uint32_t)i) == 2) {
            // chroma only: after the luma CUs of its block (the CUs that read this block later wait for this one CU)
            r.flags |= 32u;
            for (int sy = ys; sy < ys + hu; sy++) for (int sx = xs; sx < xs + wu; sx++) {
                const uint32_t j = owner[(size_t)sy * ws + sx];
                if (j >= (uint32_t)i) continue;
                if (ordered(j) && j != last) {
                    bool seen = false;
                    for (size_t d = r.dep_first; d < deps.size() && !seen; d++) seen = deps[d] == j;
                    if (!seen) deps.push_back(j);
                    last = j;
                }
                lv = std::max(lv, level_p[j]);
            }
        }
        bool used = true;
        uint32_t last_used = NONE;                                                         // the neighbour CU the unit before this one was looked up for
        auto ok = [&](int sx, int sy) -> bool {
            const uint32_t j = owner[(size_t)sy * ws + sx];
            if (j >= (uint32_t)i || tile_of(sx, sy) != my_tile) return false;              // not reconstructed yet (or nothing there), or in another tile
            if (constrained) {                                                             // constrained_intra_pred: intra neighbours only
                const uint32_t jl = constrained_tree ? luma_owner[(size_t)sy * ws + sx] : j;
                if (b->pred_mode[jl < (uint32_t)i ? jl : j] != XGPU_MODE_INTRA) return false;
            }
            if (!used) return true;
            if (j != last_used) {                                                          // (a wide neighbour covers several units: looked at once)
                last_used = j;
                if (j != last && ordered(j)) {                                             // inter CUs are complete before the intra kernel starts
                    bool seen = false;
                    for (size_t d = r.dep_first; d < deps.size() && !seen; d++) seen = deps[d] == j;
                    if (!seen) deps.push_back(j);
                    last = j;
                }
                if (!fast) lv = std::max(lv, level_p[j]);                                  // (parallel mode: the levels are assigned afterwards)
            }
            return true;
        };
        used = need_ul;
        if (xs > 0 && ys > 0 && ok(xs - 1, ys - 1)) r.flags |= 1u;
        for (int k = 0; k < units; k++) {
            used = k < need_up;
            if (ys > 0 && xs + k < ws && ok(xs + k, ys - 1)) r.up |= 1ull << k;
        }
        for (int k = 0; k < units; k++) {
            used = k < need_le;
            if (xs > 0 && ys + k < hs && ok(xs - 1, ys + k)) r.le |= 1ull << k;
        }
        // sps_suco_flag: a split coded right to left leaves the CU with its RIGHT neighbours reconstructed.  avail_lr (xevd_check_nev_avail, xevd_util.c:1156-1174: the
        // SCU left of / right of the CU's first row is reconstructed, whatever its mode) goes into flag bits 23 / 24; the units of the right column the predictors may
        // read (xevdm_get_nbr :123-147) into the upper half of `up`: such a CU lies in a node of at most 64x64 that was split vertically, so its masks are short
        {
            const uint32_t jl = xs > 0 ? owner[(size_t)ys * ws + xs - 1] : NONE, jr = xs + wu < ws ? owner[(size_t)ys * ws + xs + wu] : NONE;
            if (jr < (uint32_t)i && tile_of(xs + wu, ys) == my_tile) {
                if (units > 32) return -1;
                r.flags |= 1u << 24;
                has_right = true;
                if (jl < (uint32_t)i && tile_of(xs - 1, ys) == my_tile) r.flags |= 1u << 23;      // (only matters next to bit 24: LR_11 against LR_01)
                uint32_t ri = 0;
                for (int k = 0; k < units; k++) {
                    used = c->sp.tool_eipd != 0;                    // (the Baseline predictors never read the right column; HTDF lists its own dependencies)
                    if (ys + k < hs && ok(xs + wu, ys + k)) ri |= 1u << k;
                }
                r.up |= (uint64_t)ri << 32;
            }
        }
        if (hidx >= 0) { used = true; if (!add_htdf(r)) return -1; has_htdf = true; has_ibc = true; }
        r.dep_count = (uint32_t)deps.size() - r.dep_first;
        lv_out = lv + 1;
        return 1;
    };
    PT("setup");
    if (!fast) {
        recs.clear();
        for (int i = 0; i < n; paint(i), i++) {
            int lv = 0;
            IntraRec r;
            const int rc = make_node(i, r, deps, plan.has_ibc, plan.has_htdf, plan.has_right, lv);
            if (rc < 0) return false;
            if (rc) { recs.push_back(r); level[(size_t)i] = lv; max_level = std::max(max_level, lv); }
        }
    } else {
        // `nodes` = the CUs that are nodes, in decoding order (collected by the caller's validation pass): the ranges of this list go to the threads
        const int nn = (int)nodes.size();
        const int K = std::max(1, std::min(nthr, std::max(1, nn / 2048)));
        // every entry of `nodes` becomes exactly one record: the threads write their ranges of the final list; the dependency lists are per thread and
        // concatenated afterwards (dep_first moved along while the levels are assigned)
        struct Out { std::vector<uint32_t> deps; bool ibc = false, htdf = false, right = false, bad = false; };
        static thread_local std::vector<Out> outs_tl;
        std::vector<Out> &outs = outs_tl;
        if ((int)outs.size() < K) outs.resize((size_t)K);
        for (Out &o : outs) { o.deps.clear(); o.ibc = o.htdf = o.right = o.bad = false; }
        recs.resize((size_t)nn);
        IntraRec *const recs_p = recs.data();
        auto work = [&](int k) {
            Out &o = outs[(size_t)k];
            int lv = 0;
            const int q0 = (int)((long long)nn * k / K), q1 = (int)((long long)nn * (k + 1) / K);
            o.deps.reserve((size_t)(q1 - q0) * 3);
            for (int q = q0; q < q1 && !o.bad; q++) o.bad = make_node((int)nodes[(size_t)q], recs_p[q], o.deps, o.ibc, o.htdf, o.right, lv) != 1;
        };
        pool.run(K, work);
        PT("nodes");
        size_t nd = 0;
        for (int k = 0; k < K; k++) { const Out &o = outs[(size_t)k]; if (o.bad) return false; nd += o.deps.size(); plan.has_ibc |= o.ibc; plan.has_htdf |= o.htdf; plan.has_right |= o.right; }
        deps.reserve(nd);
        // levels, in decoding order: 1 + the highest level among the nodes read (CUs that are no nodes - complete before the intra kernels start - count as level 0)
        for (int k = 0; k < K; k++) {
            const uint32_t base = (uint32_t)deps.size();
            deps.insert(deps.end(), outs[(size_t)k].deps.begin(), outs[(size_t)k].deps.end());
            for (int q = (int)((long long)nn * k / K), q1 = (int)((long long)nn * (k + 1) / K); q < q1; q++) {
                IntraRec &r = recs_p[q];
                r.dep_first += base;
                int lv = 0;
                for (uint32_t d = r.dep_first; d < r.dep_first + r.dep_count; d++) lv = std::max(lv, level[deps[d]]);
                level[r.cu] = lv + 1;
                max_level = std::max(max_level, lv + 1);
            }
        }
    }
    PT("levels");
    // (Rounds 2 - 4 kept a second formulation beside this one - one workgroup per CTU with the CTU's chain in LDS, k_intra_ctu.hip, XEVD_HIP_INTRA_CTU=1 - bit-exact and measured
    //  slower on every all-intra picture (1080p Baseline 3.3 ms against 2.3, Main 8.9 against 8.0: a link of a chain is ~2 - 4 us of single-wave instruction latency either way,
    //  and ordering whole CTUs gives up the overlap of the CU-granular graph).  Removed in round 4; `git show ac993a5:xevd_amd/csrc/k_intra_ctu.hip` has it.)
    // Strands (k_intra.hip): a CU of level 2 and up whose dependency list holds exactly ONE CU of level 2 and up (the others are level-1 CUs, complete before the
    // data-flow launch) is linked to that CU when it has no successor yet; the wave that reconstructs the parent continues with it.
    // Parts (k_intra.hip): a wave takes 64 units (EIPD: rows of four luma samples + a chroma pair) or 64 SCUs (Baseline predictors) of its CU per step, a 64x64 CU is 16 (4) steps
    // of one wave - on the critical path of every chain through it, and the level-1 launch is as long as its largest CUs take.  Such a CU goes into the list as several entries,
    // one per step (at most 16), each with its own done flag: every part stages the neighbours and derives the plan itself and reconstructs its share; whoever reads the CU waits
    // for all parts.  Not for HTDF / IBC nodes (the filter stage works on the whole block).  XEVD_HIP_NO_PARTS=1: A/B measurements.
    static const bool no_parts = getenv("XEVD_HIP_NO_PARTS") != NULL;
    auto parts_of = [&](const IntraRec &r) -> int {
        if (no_parts || (r.flags & (2u | 4u | 8u))) return 1;
        const int nscu = 1 << (r.log2w + r.log2h - 4), steps = (c->sp.tool_eipd ? nscu * 4 : nscu) / 64;
        return std::max(1, std::min(steps, 16));
    };
    static thread_local std::vector<int32_t> rec_of_cu_tl;
    std::vector<int32_t> &rec_of_cu = rec_of_cu_tl;
    rec_of_cu.assign((size_t)n, -1);
    for (size_t ri = 0; ri < recs.size(); ri++) rec_of_cu[recs[ri].cu] = (int32_t)ri;
    static thread_local std::vector<uint32_t> succ_tl;
    static thread_local std::vector<uint8_t> member_tl;
    std::vector<uint32_t> &succ = succ_tl;
    std::vector<uint8_t> &member = member_tl;                // 1: reached through its parent, not through a ticket
    succ.assign(recs.size(), NONE); member.assign(recs.size(), 0);
    for (size_t ri = 0; ri < recs.size(); ri++) {
        const IntraRec &r = recs[ri];
        if (level[r.cu] < 2) continue;
        int cnt = 0;
        uint32_t parent = NONE;
        for (uint32_t d = r.dep_first; d < r.dep_first + r.dep_count; d++) if (level[deps[d]] >= 2) { cnt++; parent = deps[d]; }
        static const bool no_strands = getenv("XEVD_HIP_NO_STRANDS") != NULL;      // A/B measurements
        if (cnt != 1 || no_strands) continue;
        const size_t pr = (size_t)rec_of_cu[parent];
        if (parts_of(r) > 1 || parts_of(recs[pr]) > 1) continue;
        if (succ[pr] == NONE) { succ[pr] = (uint32_t)ri; member[ri] = 1; }
    }
    // sort by level (levels are 1-based; every dependency sits on a lower one), the larger CUs of a level first - a 64x64 CU is four rounds of its wave and should
    // not be the last thing a launch starts -, decode order otherwise, the strand members behind everything else; then dependency CU indices -> list positions
    auto key = [&](size_t ri) { return member[ri] ? ((size_t)max_level + 1) * 16 : (size_t)level[recs[ri].cu] * 16 + (size_t)(14 - (recs[ri].log2w + recs[ri].log2h)); };      // counting sort: log2w + log2h is 4 .. 14
    std::vector<int> first(((size_t)max_level + 3) * 16, 0);
    size_t n_entries = 0;
    for (size_t ri = 0; ri < recs.size(); ri++) { const int np = parts_of(recs[ri]); first[key(ri) + 1] += np; n_entries += (size_t)np; }
    for (size_t l = 1; l < first.size(); l++) first[l] += first[l - 1];
    // the level-1 entries of at most 16 SCUs (log2 w + log2 h <= 8: key 16 + 6 and up) end the level's part of the list: k_intra gives them 16 lanes each
    plan.n_level1_small = max_level >= 1 ? first[2 * 16] - first[16 + 6] : 0;
    static thread_local std::vector<uint32_t> pos;                    // CU index -> list position of its first part
    pos.assign((size_t)n, NONE);
    plan.recs.resize(n_entries);
    plan.n_level1 = 0; plan.n_heads = 0;
    uint32_t *const pos_p = pos.data();
    for (size_t ri = 0; ri < recs.size(); ri++) {                     // positions (serial: a running counter per key) ...
        const int np = parts_of(recs[ri]), k = first[key(ri)];
        first[key(ri)] += np;
        pos_p[recs[ri].cu] = (uint32_t)k;
        if (level[recs[ri].cu] == 1) plan.n_level1 += np;
        if (!member[ri]) plan.n_heads += np;
    }
    plan.n_levels = max_level;
    // dependency CU indices -> list positions, every part of a CU that has parts.  Level-1 CUs are finished by their own launch before the data-flow launch starts: they drop
    // out of the waiting lists; a strand member waits for nobody (its one dependency of the launch is the CU its wave has just finished).  Counted per record first, so that
    // the records and their lists can be written by the builder's threads
    static thread_local std::vector<uint32_t> nfirst;
    nfirst.resize(recs.size() + 1);
    uint32_t *const nf_p = nfirst.data();
    const IntraRec *const rc_p = recs.data();
    const uint32_t *const dp_p = deps.data();
    const int32_t *const roc_p = rec_of_cu.data();
    const uint32_t *const succ_p = succ.data();
    const uint8_t *const mem_p = member.data();
    const uint32_t n_l1 = (uint32_t)plan.n_level1;
    const int KS = std::max(1, std::min(nthr, std::max(1, (int)recs.size() / 2048)));
    auto range = [&](int k, size_t &a0, size_t &a1) { a0 = recs.size() * (size_t)k / KS; a1 = recs.size() * (size_t)(k + 1) / KS; };
    pool.run(KS, [&](int k) {
        size_t a0, a1;
        range(k, a0, a1);
        for (size_t ri = a0; ri < a1; ri++) {
            uint32_t c = 0;
            if (!mem_p[ri])
                for (uint32_t d = rc_p[ri].dep_first; d < rc_p[ri].dep_first + rc_p[ri].dep_count; d++) {
                    const uint32_t j = dp_p[d];
                    if (pos_p[j] >= n_l1) c += (uint32_t)parts_of(rc_p[(size_t)roc_p[j]]);
                }
            nf_p[ri + 1] = c;
        }
    });
    nf_p[0] = 0;
    for (size_t ri = 0; ri < recs.size(); ri++) nf_p[ri + 1] += nf_p[ri];
    PT("sort");
    std::vector<uint32_t> &ndeps = plan.deps;                         // (the caller's plan object keeps its capacity between pictures)
    ndeps.resize((size_t)nf_p[recs.size()]);
    uint32_t *const nd_p = ndeps.data();
    IntraRec *const out_p = plan.recs.data();
    pool.run(KS, [&](int k) {                                         // ... records and lists (parallel: every record knows where it goes)
        size_t a0, a1;
        range(k, a0, a1);
        for (size_t ri = a0; ri < a1; ri++) {
            const IntraRec &r = rc_p[ri];
            uint32_t w = nf_p[ri];
            if (!mem_p[ri])
                for (uint32_t d = r.dep_first; d < r.dep_first + r.dep_count; d++) {
                    const uint32_t j = dp_p[d], pj = pos_p[j];
                    if (pj < n_l1) continue;
                    const int npj = parts_of(rc_p[(size_t)roc_p[j]]);
                    for (int q = 0; q < npj; q++) nd_p[w++] = pj + (uint32_t)q;
                }
            const int np = parts_of(r);
            for (int q = 0; q < np; q++) {
                IntraRec &o = out_p[(size_t)pos_p[r.cu] + q];
                o = r;
                o.pad0 = (uint8_t)q; o.pad1[0] = (uint8_t)np;
                o.dep_first = nf_p[ri]; o.dep_count = nf_p[ri + 1] - nf_p[ri];
                // the device reads the successor's list position where the host kept the CU index
                o.cu = succ_p[ri] == NONE ? NONE : pos_p[rc_p[succ_p[ri]].cu];
            }
        }
    });
    return true;
}

struct StageSeg { size_t off, bytes; };
static int batch_build(xgpu_ctx *c, const xgpu_cu_batch *b, xgpu_dbatch **out, bool host_only, std::vector<StageSeg> *segs)
{
    ARGCHK(c, c != NULL); ARGCHK(c, b != NULL && out != NULL);
    *out = NULL;
    ARGCHK(c, b->n_cu >= 0 && b->n_ctu == c->w_ctu * c->h_ctu);
    ARGCHK(c, b->n_cu == 0 || (b->x && b->y && b->log2w && b->log2h && b->pred_mode && b->refi && b->mv && b->qp && b->cbf && b->coef_off));
    ARGCHK(c, b->ctu_cu_start != NULL && (b->n_coef == 0 || b->coef != NULL));
    ARGCHK(c, b->htdf_slice_qp >= 0 && b->htdf_slice_qp <= 51);
    ARGCHK(c, b->ctu_cu_start[0] == 0 && b->ctu_cu_start[b->n_ctu] == (uint32_t)b->n_cu);      // the kernels index the CU records through it
    for (int k = 0; k < b->n_ctu; k++) ARGCHK(c, b->ctu_cu_start[k] <= b->ctu_cu_start[k + 1]);
    TileMask tmask;
    ARGCHK(c, tile_mask(c, b->tiles, tmask));
    if (!host_only) HIPCHK(c, hipSetDevice(c->sp.device));
    const int n = b->n_cu;
    const int bdoff = 6 * (c->sp.bit_depth_luma - 8);
    static const bool bt_on = getenv("XEVD_HIP_BUILD_TRACE") != NULL;      // phase times of the builder on stderr
    auto bt_t0 = std::chrono::steady_clock::now();
    auto BT = [&](const char *what) { if (bt_on) { const auto t = std::chrono::steady_clock::now(); fprintf(stderr, "  batch build: %-14s %.2f ms\n", what, std::chrono::duration<double, std::milli>(t - bt_t0).count()); bt_t0 = t; } };

    // pass 1: validate + count TBs per size class
    // size class = (log2w, log2h) x (vertical, horizontal) transform kind; ATS kinds only occur for intra luma TBs
    enum { NCLS = 64 * 9 };
    auto ats_inter_of = [&](int i) -> int { return (b->ats_inter && b->pred_mode[i] != XGPU_MODE_INTRA && b->pred_mode[i] != XGPU_MODE_IBC) ? b->ats_inter[i] : 0; };
    auto tr_code = [&](int i, int k) -> int {
        if (k != 0) return 0;
        if (const int ai = ats_inter_of(i)) {
            // xevdm_get_ats_inter_trs (src_main/xevdm_util.c:3636-3668): DST-VII across the split, DCT-VIII along it for the
            // first part / DST-VII for the last; CUs wider or taller than 32 keep DCT-II
            if (b->log2w[i] > 5 || b->log2h[i] > 5) return 0;
            const int idx = ai & 15, pos = ai >> 4, hor = idx == 2 || idx == 4;
            const int tv = hor ? (pos == 0 ? TR_DCT8 : TR_DST7) : TR_DST7, th = hor ? TR_DST7 : (pos == 0 ? TR_DCT8 : TR_DST7);
            return tv * 3 + th;
        }
        if (!b->ats || !(b->ats[i] & 1) || b->pred_mode[i] != XGPU_MODE_INTRA) return 0;
        const int tv = (b->ats[i] >> 1) & 1 ? TR_DCT8 : TR_DST7, th = (b->ats[i] >> 2) & 1 ? TR_DCT8 : TR_DST7;
        return tv * 3 + th;
    };
    // luma log2 size of the CU's coefficient block: the CU, or the ATS-inter TU (xevdm_get_tu_size, xevdm_util.c:3585-3608)
    auto blk_log2 = [&](int i, int &bw, int &bh) {
        bw = b->log2w[i]; bh = b->log2h[i];
        const int idx = ats_inter_of(i) & 15;
        if (idx == 1 || idx == 3) bw -= idx == 3 ? 2 : 1;
        if (idx == 2 || idx == 4) bh -= idx == 4 ? 2 : 1;
    };
    // DMVR candidates the backend can refine: flagged, plain inter, two references, at least 8x8 (the POC test happens on the device)
    auto dmvr_cand = [&](int i) -> bool {
        return b->dmvr && b->dmvr[i] && b->pred_mode[i] != XGPU_MODE_INTRA && b->pred_mode[i] != XGPU_MODE_IBC && !(b->affine && b->affine[i]) &&
               b->refi[i * 2] >= 0 && b->refi[i * 2 + 1] >= 0 && b->log2w[i] >= 3 && b->log2h[i] >= 3;
    };
    // the branch xevdm_affine_mc takes for CU i (EIF when a sub-block would be smaller than 8 samples): the kernels' own code, affine_model.h
    auto affine_is_eif = [&](const xgpu_cu_batch *bb, int i) -> bool {
        const bool use[2] = { bb->refi[i * 2] >= 0, bb->refi[i * 2 + 1] >= 0 };
        AffModel md[2];
        for (int l = 0; l < 2; l++) md[l] = aff_model(bb->affine_mv + (size_t)i * 12 + l * 6, bb->log2w[i], bb->log2h[i], bb->affine[i]);
        int sw, sh; bool mb;
        aff_subblock(md, use, bb->log2w[i], bb->log2h[i], sw, sh, mb);
        return sw < 8 || sh < 8;
    };
    // The builder's per-CU passes run on `builder_threads` host threads (xgpu_set_builder_threads; default 1), each over a contiguous range of CUs: pass 1
    // validates and counts per range, a prefix over the ranges gives every thread its own start in each output list, pass 2 and the owner map then write
    // disjoint parts - the lists come out exactly as the sequential passes build them.
    struct Part { int cls[NCLS]; int n_aff, n_eif, n_sub, n_dmvr; std::vector<uint32_t> nodes; };      // nodes: the CUs of the range that enter the dependency plan (intra, IBC, HTDF)
    const int nthr = std::max(1, std::min(c->builder_threads, std::max(1, n / 4096)));
    std::vector<Part> parts((size_t)nthr);
    for (Part &P : parts) { memset(P.cls, 0, sizeof(P.cls)); P.n_aff = P.n_eif = P.n_sub = P.n_dmvr = 0; }
    static thread_local WorkPool pool;                     // this caller's worker threads, kept between pictures
    auto run_parts = [&](auto fn) {                       // fn(thread, first CU, one past the last)
        pool.run(nthr, [&](int k) { fn(k, (int)((long long)n * k / nthr), (int)((long long)n * (k + 1) / nthr)); });
    };
#define CUCHK(cond) do { if (!(cond)) return #cond; } while (0)
    auto pass1 = [&](int i0, int i1, Part &P) -> const char * {
    for (int i = i0; i < i1; i++) {
        const int lw = b->log2w[i], lh = b->log2h[i];
        CUCHK(lw >= 2 && lw <= 7 && lh >= 2 && lh <= 7 && lw <= c->sp.log2_ctu && lh <= c->sp.log2_ctu);
        CUCHK(b->x[i] + (1 << lw) <= c->sp.width && b->y[i] + (1 << lh) <= c->sp.height && !(b->x[i] & 3) && !(b->y[i] & 3));
        CUCHK(b->pred_mode[i] <= XGPU_MODE_DIR || b->pred_mode[i] == XGPU_MODE_IBC);
        if (b->tree && b->tree[i]) {      // local dual tree: luma-only intra / IBC CUs, chroma-only intra CUs, only the coefficients of the planes they have
            CUCHK(b->tree[i] <= 2 && (b->pred_mode[i] == XGPU_MODE_INTRA || (b->tree[i] == 1 && b->pred_mode[i] == XGPU_MODE_IBC)));
            CUCHK((b->cbf[i] & (b->tree[i] == 1 ? 6 : 1)) == 0);
            if (b->tree[i] == 1) {        // a luma-only CU lies inside the chroma-only CU that closes its tree (checked here: nothing is allocated yet)
                int j = i + 1;
                while (j < n && b->tree[j] != 2) j++;
                CUCHK(j < n && b->x[j] <= b->x[i] && b->y[j] <= b->y[i] && b->x[i] + (1 << lw) <= b->x[j] + (1 << b->log2w[j]) && b->y[i] + (1 << lh) <= b->y[j] + (1 << b->log2h[j]));
            }
        }
        if (b->pred_mode[i] == XGPU_MODE_IBC) {
            // the source block (and the chroma block at the halved vector) inside the active picture; that it is reconstructed before the CU is
            // checked by the dependency plan below
            const int bvx = b->mv[i * 4], bvy = b->mv[i * 4 + 1];
            CUCHK(b->x[i] + (bvx & ~1) >= 0 && b->y[i] + (bvy & ~1) >= 0 && b->x[i] + bvx + (1 << lw) <= c->sp.width && b->y[i] + bvy + (1 << lh) <= c->sp.height);
            CUCHK(!(b->affine && b->affine[i]));
        } else if (b->pred_mode[i] != XGPU_MODE_INTRA) CUCHK(b->refi[i * 2] < XGPU_MAX_REFS && b->refi[i * 2 + 1] < XGPU_MAX_REFS);
        CUCHK(b->qp[i * 3] < 96 && b->qp[i * 3 + 1] < 96 && b->qp[i * 3 + 2] < 96);                     // 0..51 + 6 * (bit depth - 8)
        if (const int ai = ats_inter_of(i)) {
            // availability as xevdm_check_ats_inter_info_coded (xevdm_util.c:3565-3583): CU <= 64, split dimension >= 8 (>= 16 for quarters)
            const int idx = ai & 15, pos = ai >> 4;
            CUCHK(idx >= 1 && idx <= 4 && pos <= 1 && lw <= 6 && lh <= 6);
            CUCHK(((idx == 1 || idx == 3) ? lw : lh) >= (idx >= 3 ? 4 : 3));
        }
        if (b->affine && b->affine[i]) {
            // affine CUs exist from 8x8 (xevdm_eco.c:1529), with 2 or 3 control points and at least one reference
            CUCHK(b->affine_mv != NULL && (b->affine[i] == 2 || b->affine[i] == 3) && b->pred_mode[i] != XGPU_MODE_INTRA);
            CUCHK(lw >= 3 && lh >= 3 && (b->refi[i * 2] >= 0 || b->refi[i * 2 + 1] >= 0));
            P.n_aff++;
            if (affine_is_eif(b, i)) P.n_eif += ((1 << lw) + 15) / 16 * (((1 << lh) + 15) / 16);
            else                     P.n_sub += ((1 << lw) + 31) / 32 * (((1 << lh) + 31) / 32);
        }
        if (dmvr_cand(i)) P.n_dmvr += (lw > 4 ? 1 << (lw - 4) : 1) * (lh > 4 ? 1 << (lh - 4) : 1);
        size_t need = 0;
        int bw, bh;
        blk_log2(i, bw, bh);
        for (int k = 0; k < 3; k++) {
            if (!((b->cbf[i] >> k) & 1)) continue;
            // TBs are at most 64 wide/tall: a larger CU is cut into 64x64 (chroma 32x32) sub-blocks (xevd_itdq.c:544-621)
            const int tw = std::min(bw, 6) - (k ? 1 : 0), th = std::min(bh, 6) - (k ? 1 : 0);
            const int nsx = lw > 6 ? 2 : 1, nsy = lh > 6 ? 2 : 1;
            for (int sb = 0; sb < 4; sb++) {
                if ((sb & 1) >= nsx || (sb >> 1) >= nsy) continue;
                if (nsx * nsy > 1 && b->cbf_sub && !((b->cbf_sub[i] >> (4 * k + sb)) & 1)) continue;
                CUCHK(tr_code(i, k) == 0 || (tw >= 2 && tw <= 5 && th >= 2 && th <= 5));      // ATS exists for 4..32 only (checked before anything is allocated)
                P.cls[tr_code(i, k) * 64 + tw * 8 + th]++;
            }
            need += (size_t)(1 << (bw + bh)) >> (k ? 2 : 0);
        }
        CUCHK((size_t)b->coef_off[i] + need <= b->n_coef);
        if (plan_is_node(b, (uint32_t)i)) P.nodes.push_back((uint32_t)i);
    }
    return nullptr;
    };
#undef CUCHK
    {
        std::vector<const char *> bad((size_t)nthr, nullptr);
        run_parts([&](int k, int i0, int i1) { bad[(size_t)k] = pass1(i0, i1, parts[(size_t)k]); });
        for (const char *m : bad)
            if (m) { snprintf(c->err, sizeof(c->err), "%s: invalid argument: %s", __FILE__, m); return XGPU_ERR_INVALID_ARGUMENT; }
    }
    BT("pass 1");
    int cls_count[NCLS] = { 0 }, n_aff = 0, n_aff_eif = 0, n_aff_sub = 0, n_dmvr = 0;
    for (const Part &P : parts) { for (int k = 0; k < NCLS; k++) cls_count[k] += P.cls[k]; n_aff += P.n_aff; n_aff_eif += P.n_eif; n_aff_sub += P.n_sub; n_dmvr += P.n_dmvr; }
    int cls_first[NCLS], n_tb = 0, n_waves = 0;
    for (int k = 0; k < NCLS; k++) {
        cls_first[k] = n_tb; n_tb += cls_count[k];
        if (cls_count[k]) { const int per = itdq_group_size((k & 63) >> 3, k & 7); n_waves += (cls_count[k] + per - 1) / per; }
    }

    static thread_local IntraPlan plan_tl;                 // (kept between pictures: see build_intra_plan)
    IntraPlan &plan = plan_tl;
    plan.recs.clear(); plan.deps.clear();
    plan.n_levels = 0; plan.n_level1 = 0; plan.n_level1_small = 0; plan.n_heads = 0;
    bool any_intra = false;
    plan.has_ibc = false; plan.has_htdf = false; plan.has_right = false;
    // SCU -> CU map of the picture (k_inter's lanes find their CU through it; the dependency plan reads "reconstructed before" off it); SCUs outside the batch -
    // another tile's - stay unowned.  Painted in ordinary memory (short row fills) and copied into the pinned block in one piece further down
    static thread_local std::vector<uint32_t> own;
    {
        own.resize((size_t)c->w_scu * c->h_scu);
        size_t covered = 0;
        for (int i = 0; i < n; i++) if (!(b->tree && b->tree[i] == 2)) covered += (size_t)1 << (b->log2w[i] + b->log2h[i] - 4);
        if (covered != own.size()) std::fill(own.begin(), own.end(), 0xFFFFFFFFu);
        uint32_t *const own_p = own.data();                     // (`own` is thread_local: inside another thread the NAME would mean that thread's empty vector)
        run_parts([&, own_p](int, int i0, int i1) {             // CUs do not overlap: the ranges paint disjoint SCUs
            for (int i = i0; i < i1; i++) {
                if (b->tree && b->tree[i] == 2) continue;           // the SCU maps of a dual-tree block belong to its luma CUs
                const int ws = (1 << b->log2w[i]) >> 2, hh = (1 << b->log2h[i]) >> 2;
                uint32_t *o = own_p + (size_t)(b->y[i] >> 2) * c->w_scu + (b->x[i] >> 2);
                for (int r = 0; r < hh; r++, o += c->w_scu) std::fill_n(o, ws, (uint32_t)i);
            }
        });
    }
    BT("owner map");
    // Work lists of the three inter launches (k_inter.hip): 64x64 regions inside one CU, 32x32 tiles inside one CU, the other tiles that hold SCUs of the batch.  A tile /
    // region counts as "inside one CU" only when it lies inside the picture as a whole (the kernels' shared-window paths have no partial form).  The CUs mark the tiles
    // (disjoint CUs: disjoint full tiles; `any` is a relaxed flag several CUs of one tile may set), one sequential sweep in the kernels' spatial order - vertical strips
    // XGPU_INTER_STRIP regions wide, row by row inside a strip - emits the lists.
    static thread_local std::vector<uint32_t> tile_cu;
    static thread_local std::vector<uint8_t> tile_any;
    static thread_local std::vector<uint32_t> inter_items;        // [entry * 4 + tile]: the CU a whole tile lies in (the staging copy adds the CU's record), else none
    static thread_local std::vector<uint32_t> inter_work;         // one entry per 64x64 region of the picture (k_inter.hip: InterArgs.work)
    {
        const int tiles_x = (c->sp.width + 31) >> 5, tiles_y = (c->sp.height + 31) >> 5, full_x = c->sp.width >> 5, full_y = c->sp.height >> 5;
        tile_cu.assign((size_t)tiles_x * tiles_y, 0xFFFFFFFFu);
        tile_any.assign((size_t)tiles_x * tiles_y, 0);
        uint32_t *const tcu = tile_cu.data();
        uint8_t *const tany = tile_any.data();
        run_parts([&, tcu, tany](int, int i0, int i1) {
            for (int i = i0; i < i1; i++) {
                if (b->tree && b->tree[i] == 2) continue;
                const int x0 = b->x[i], y0 = b->y[i], x1 = x0 + (1 << b->log2w[i]), y1 = y0 + (1 << b->log2h[i]);
                for (int ty = y0 >> 5; ty <= (y1 - 1) >> 5; ty++)
                    for (int tx = x0 >> 5; tx <= (x1 - 1) >> 5; tx++) {
                        if (tx < full_x && ty < full_y && (tx << 5) >= x0 && (tx << 5) + 32 <= x1 && (ty << 5) >= y0 && (ty << 5) + 32 <= y1) tcu[(size_t)ty * tiles_x + tx] = (uint32_t)i;
                        else __atomic_store_n(&tany[(size_t)ty * tiles_x + tx], (uint8_t)1, __ATOMIC_RELAXED);
                    }
            }
        });
        inter_items.clear(); inter_work.clear();
        const int regions_x = (c->sp.width + 63) >> 6, regions_y = (c->sp.height + 63) >> 6;
        for (int s0 = 0; s0 < regions_x; s0 += XGPU_INTER_STRIP)
            for (int ry = 0; ry < regions_y; ry++)
                for (int rx = s0; rx < std::min(s0 + XGPU_INTER_STRIP, regions_x); rx++) {
                    const int tx = rx * 2, ty = ry * 2;
                    const bool whole = tx + 1 < tiles_x && ty + 1 < tiles_y;
                    const uint32_t o = tcu[(size_t)ty * tiles_x + tx];
                    if (whole && o != 0xFFFFFFFFu && tcu[(size_t)ty * tiles_x + tx + 1] == o && tcu[(size_t)(ty + 1) * tiles_x + tx] == o && tcu[(size_t)(ty + 1) * tiles_x + tx + 1] == o) {
                        inter_work.push_back(XGPU_WORK_REGION);
                        for (int q = 0; q < 4; q++) inter_items.push_back(o);      // every wave of the region role reads the item of its own tile's place
                        continue;
                    }
                    uint32_t kinds = 0;
                    for (int q = 0; q < 4; q++) {
                        const int ux = tx + (q & 1), uy = ty + (q >> 1);
                        uint32_t oq = 0xFFFFFFFFu;
                        if (ux < tiles_x && uy < tiles_y) {
                            oq = tcu[(size_t)uy * tiles_x + ux];
                            if (oq != 0xFFFFFFFFu) kinds |= 1u << (2 * q);
                            else if (tany[(size_t)uy * tiles_x + ux]) kinds |= 2u << (2 * q);
                        }
                        inter_items.push_back(oq);
                    }
                    inter_work.push_back(kinds);
                }
    }
    BT("inter lists");
    // sps_suco_flag: is any CU decoded AFTER its right-hand neighbour?  (All right-hand neighbours of a CU lie in the other part of one vertical split: the first one
    // tells.)  Only the baseline deblocking filter wants to know beforehand - it applies chroma edges 2 samples apart in the order the reference's tree walk reaches
    // them (k_deblock.hip) and takes its left-to-right instantiation otherwise; ADDB is order-free, the intra plan finds its right-hand neighbours itself
    bool order_rl = false;
    if (!c->sp.tool_addb) {
        std::atomic<int> found(0);
        const uint32_t *const own_p = own.data();
        run_parts([&, own_p](int, int i0, int i1) {
            for (int i = i0; i < i1 && !found.load(std::memory_order_relaxed); i++) {
                const int xr = b->x[i] + (1 << b->log2w[i]);
                if (xr < c->sp.width && own_p[(size_t)(b->y[i] >> 2) * c->w_scu + (xr >> 2)] < (uint32_t)i) found.store(1, std::memory_order_relaxed);
            }
        });
        order_rl = found.load() != 0;
    }
    static thread_local std::vector<uint32_t> node_list;
    node_list.clear();
    for (const Part &P : parts) node_list.insert(node_list.end(), P.nodes.begin(), P.nodes.end());
    any_intra = !node_list.empty();
    if (any_intra) ARGCHK(c, build_intra_plan(c, b, plan, own.data(), nthr, pool, node_list));      // false: an IBC source block that is not reconstructed before its CU
    const int n_intra = (int)plan.recs.size(), n_deps = (int)plan.deps.size();
    BT("intra plan");

    xgpu_dbatch *db = new xgpu_dbatch();
    memset(db, 0, sizeof(*db));
    db->n_cu = n; db->n_ctu = b->n_ctu; db->n_tb = n_tb; db->n_waves = n_waves; db->n_coef = b->n_coef; db->n_intra = n_intra; db->n_intra_deps = n_deps; db->n_levels = plan.n_levels; db->n_intra_l1 = plan.n_level1; db->n_intra_l1_small = plan.n_level1_small; db->n_intra_heads = plan.n_heads; db->n_aff_eif = n_aff_eif; db->n_aff_sub = n_aff_sub; db->n_dmvr = n_dmvr; db->has_ibc = plan.has_ibc ? 1 : 0; db->has_htdf = plan.has_htdf ? 1 : 0; db->has_right = plan.has_right ? 1 : 0; db->order_rl = order_rl ? 1 : 0;
    db->tile_starts = tmask; db->tiles_across = b->tiles ? (b->tiles->loop_filter_across_tiles ? 1 : 0) : 1;
    const size_t sz_cus = sizeof(CuRec) * (size_t)std::max(n, 1), sz_ctu = sizeof(uint32_t) * (size_t)(b->n_ctu + 1);
    const size_t sz_tbs = sizeof(TbRec) * (size_t)std::max(n_tb, 1), sz_wv = sizeof(TbWave) * (size_t)std::max(n_waves, 1);
    const size_t sz_coef = sizeof(int16_t) * std::max(b->n_coef, (size_t)8);
    const size_t o_cus = 0, o_ctu = o_cus + align_up((int)sz_cus, 256), o_tbs = o_ctu + align_up((int)sz_ctu, 256);
    const size_t sz_intra = sizeof(IntraRec) * (size_t)std::max(n_intra, 1);
    const size_t o_wv = o_tbs + align_up((int)sz_tbs, 256), o_intra = o_wv + align_up((int)sz_wv, 256);
    const size_t sz_deps = sizeof(uint32_t) * (size_t)std::max(n_deps, 1);
    const size_t sz_aff = sizeof(AffItem) * (size_t)std::max(n_aff_eif + n_aff_sub, 1), sz_cpmv = sizeof(int16_t) * 12 * (size_t)std::max(n_aff, 1);
    const size_t o_deps = o_intra + align_up((int)sz_intra, 256), o_aff = o_deps + align_up((int)sz_deps, 256);
    const size_t sz_dmvr = sizeof(DmvrItem) * (size_t)std::max(n_dmvr, 1);
    const size_t sz_own = sizeof(uint32_t) * (size_t)c->w_scu * c->h_scu;
    const size_t o_cpmv = o_aff + align_up((int)sz_aff, 256), o_dmvr = o_cpmv + align_up((int)sz_cpmv, 256), o_own = o_dmvr + align_up((int)sz_dmvr, 256);
    const size_t sz_iitem = sizeof(InterItem) * std::max(inter_items.size(), (size_t)1), sz_iwork = sizeof(uint32_t) * std::max(inter_work.size(), (size_t)1);
    const size_t o_iitem = o_own + align_up((int)sz_own, 256), o_iwork = o_iitem + align_up((int)sz_iitem, 256);
    const size_t o_coef = o_iwork + align_up((int)sz_iwork, 256);
    db->stage_bytes = o_coef + sz_coef;
    auto fail = [&](int code) { xgpu_batch_destroy(c, db); return code; };
    // device layout: the uploaded arrays at the staging offsets, then the residual arena and the intra done flags
    const size_t sz_done = sizeof(uint32_t) * ((size_t)n_intra + 1);
    const size_t o_resid = align_up((int)(o_coef + sz_coef), 256), o_done = o_resid + align_up((int)sz_coef, 256);
    const size_t o_dmv = o_done + align_up((int)sz_done, 256);
    const size_t d_need = o_dmv + align_up((int)(sizeof(int16_t) * 4 * (size_t)std::max(n_dmvr, 1)), 256);
    // is the coefficient arena inside a range from xgpu_host_alloc?  Then it is sent from where it lies (no staging copy of the largest array)
    bool coef_pinned = host_only && b->n_coef != 0;      // (the builder alone: the coefficient copy - a plain memcpy, skipped for pinned arenas - stays out of the measurement)
    {
        // a pooled block that is large enough (the smallest such), else a new one
        int best = -1;
        {
            std::lock_guard<std::mutex> g(c->pool_mu);
            for (const auto &h : c->pinned)
                if ((const uint8_t *)b->coef >= h.p && (const uint8_t *)(b->coef + b->n_coef) <= h.p + h.n) coef_pinned = b->n_coef != 0;
            for (size_t k = 0; k < c->pool.size(); k++)
                if (c->pool[k].d_cap >= d_need && c->pool[k].h_cap >= db->stage_bytes && (best < 0 || c->pool[k].d_cap < c->pool[best].d_cap)) best = (int)k;
            if (best >= 0) { db->blk = c->pool[best]; c->pool.erase(c->pool.begin() + best); }
        }
        if (host_only) {
            memset(&db->blk, 0, sizeof(db->blk));
            // (kept between calls like a pooled block: a fresh 60 MB malloc per call would time the kernel's page zeroing)
            static thread_local std::vector<uint8_t> host_stage;
            if (host_stage.size() < db->stage_bytes) host_stage.resize(db->stage_bytes);
            db->blk.h_stage = host_stage.data(); db->blk.h_cap = db->stage_bytes;
        } else if (best >= 0) {
            if (hipEventSynchronize(db->blk.uploaded) != hipSuccess) return fail(XGPU_ERR_UNEXPECTED);      // its staging block may still feed an upload
            // ... and its device block the kernels of the batch that had it before: the upload stream waits for them
            if (hipStreamWaitEvent(c->up_stream, db->blk.done, 0) != hipSuccess) return fail(XGPU_ERR_UNEXPECTED);
        } else {
            memset(&db->blk, 0, sizeof(db->blk));
            const size_t d_cap = d_need + d_need / 4, h_cap = db->stage_bytes + db->stage_bytes / 4;             // headroom: pictures of a stream vary
            if (hipMalloc((void **)&db->blk.d_base, d_cap) != hipSuccess) return fail(XGPU_ERR_OUT_OF_MEMORY);
            db->blk.d_cap = d_cap;
            if (hipHostMalloc(&db->blk.h_stage, h_cap, hipHostMallocDefault) != hipSuccess) return fail(XGPU_ERR_OUT_OF_MEMORY);
            db->blk.h_cap = h_cap;
            static const bool blocking = getenv("XEVD_HIP_BLOCKING_SYNC") != NULL && atoi(getenv("XEVD_HIP_BLOCKING_SYNC")) != 0;      // see xgpu_open
            if (hipEventCreateWithFlags(&db->blk.uploaded, hipEventDisableTiming | (blocking ? hipEventBlockingSync : 0)) != hipSuccess) return fail(XGPU_ERR_UNEXPECTED);
            if (hipEventCreateWithFlags(&db->blk.done, hipEventDisableTiming) != hipSuccess) return fail(XGPU_ERR_UNEXPECTED);
            if (hipEventCreateWithFlags(&db->blk.itdq_done, hipEventDisableTiming) != hipSuccess) return fail(XGPU_ERR_UNEXPECTED);
        }
    }
    BT("block");
    db->h_stage = db->blk.h_stage;
    uint8_t *hs = (uint8_t *)db->h_stage;
    CuRec *cus = (CuRec *)(hs + o_cus);
    TbRec *tbs = (TbRec *)(hs + o_tbs);
    TbWave *wv = (TbWave *)(hs + o_wv);

    AffItem *aff_items = (AffItem *)(hs + o_aff);
    int16_t *cpmv = (int16_t *)(hs + o_cpmv);
    DmvrItem *dmvr_items = (DmvrItem *)(hs + o_dmvr);
    {
        const uint8_t *const own_b = (const uint8_t *)own.data();      // (`own` is thread_local: the workers must not name it)
        pool.run(nthr, [&](int k) { const size_t a0 = sz_own * (size_t)k / nthr & ~(size_t)63, a1 = k + 1 == nthr ? sz_own : (sz_own * (size_t)(k + 1) / nthr & ~(size_t)63); memcpy(hs + o_own + a0, own_b + a0, a1 - a0); });
    }

    // pass 2: records + TB scatter into class order; every thread starts where the ranges before it end in each list
    run_parts([&](int part, int i0, int i1) {
    int cls_fill[NCLS];
    memcpy(cls_fill, cls_first, sizeof(cls_fill));
    int aff_fill = 0, eif_fill = 0, sub_fill = n_aff_eif, dmvr_fill = 0;
    for (int q = 0; q < part; q++) {
        const Part &P = parts[(size_t)q];
        for (int k = 0; k < NCLS; k++) cls_fill[k] += P.cls[k];
        aff_fill += P.n_aff; eif_fill += P.n_eif; sub_fill += P.n_sub; dmvr_fill += P.n_dmvr;
    }
    for (int i = i0; i < i1; i++) {
        CuRec &r = cus[i];
        memset(&r, 0, sizeof(r));
        if (b->affine && b->affine[i]) {
            r.affine = b->affine[i];
            memcpy(cpmv + (size_t)aff_fill * 12, b->affine_mv + (size_t)i * 12, sizeof(int16_t) * 12);
            const bool eif = affine_is_eif(b, i);
            const int step = eif ? 16 : 32;
            for (int ty = 0; ty < (1 << b->log2h[i]); ty += step)
                for (int tx = 0; tx < (1 << b->log2w[i]); tx += step) {
                    AffItem &it = aff_items[eif ? eif_fill++ : sub_fill++];
                    it.cu = (uint32_t)i; it.aff = (uint32_t)aff_fill; it.tx = (uint16_t)tx; it.ty = (uint16_t)ty; it.pad = 0;
                }
            aff_fill++;
        }
        r.x = b->x[i]; r.y = b->y[i]; r.log2w = b->log2w[i]; r.log2h = b->log2h[i];
        r.pred_mode = b->pred_mode[i]; r.cbf = b->cbf[i] & 7;
        if (b->tree && b->tree[i] == 1) {
            // a luma-only CU: its left / top edge is an edge of the chroma block (the chroma-only CU that follows) only on that block's border
            int j = i + 1;
            while (j < n && b->tree[j] != 2) j++;
            if (b->x[i] != b->x[j]) r.pred_mode |= CU_NOCH_L;          // (containment was validated in pass 1)
            if (b->y[i] != b->y[j]) r.pred_mode |= CU_NOCH_T;
        }
        r.refi[0] = b->refi[i * 2]; r.refi[1] = b->refi[i * 2 + 1];
        r.qp_map = (uint8_t)((b->qp[i * 3] - bdoff) & 0x7F);
        r.map_cbf = (uint8_t)((r.cbf & 1) && (!(r.log2w > 6 || r.log2h > 6) || !b->cbf_sub || (b->cbf_sub[i] & 1)));
        r.coef_off = b->coef_off[i];
        memcpy(r.mv, &b->mv[i * 4], sizeof(r.mv));
        r.qp[0] = b->qp[i * 3]; r.qp[1] = b->qp[i * 3 + 1]; r.qp[2] = b->qp[i * 3 + 2];
        if (b->ipm) { r.ipm[0] = b->ipm[i * 2]; r.ipm[1] = b->ipm[i * 2 + 1]; }
        r.ats_inter = (uint8_t)ats_inter_of(i);
        if (dmvr_cand(i)) {
            r.dmvr = 1;
            const int dxs = std::min(1 << b->log2w[i], 16), dys = std::min(1 << b->log2h[i], 16);
            for (int sy = 0; sy < (1 << b->log2h[i]); sy += dys)
                for (int sx = 0; sx < (1 << b->log2w[i]); sx += dxs) {
                    DmvrItem &it = dmvr_items[dmvr_fill++];
                    it.cu = (uint32_t)i; it.sx = (uint8_t)(sx >> 2); it.sy = (uint8_t)(sy >> 2); it.pad = 0;
                }
        }
        int bw, bh;
        blk_log2(i, bw, bh);
        uint32_t off = r.coef_off;
        for (int k = 0; k < 3; k++) {
            if (!((r.cbf >> k) & 1)) continue;
            const int cl = k ? bw - 1 : bw, chh = k ? bh - 1 : bh;        // component block of the CU
            const int tw = std::min(bw, 6) - (k ? 1 : 0), th = std::min(bh, 6) - (k ? 1 : 0);
            const int nsx = r.log2w > 6 ? 2 : 1, nsy = r.log2h > 6 ? 2 : 1;
            for (int sb = 0; sb < 4; sb++) {
                const int si = sb & 1, sj = sb >> 1;
                if (si >= nsx || sj >= nsy) continue;
                if (nsx * nsy > 1 && b->cbf_sub && !((b->cbf_sub[i] >> (4 * k + sb)) & 1)) continue;
                TbRec &t = tbs[cls_fill[tr_code(i, k) * 64 + tw * 8 + th]++];
                t.off = off + ((uint32_t)sj << (th + cl)) + ((uint32_t)si << tw);
                t.log2w = (uint8_t)tw; t.log2h = (uint8_t)th; t.qp = r.qp[k]; t.log2s = (uint8_t)cl;
            }
            off += 1u << (cl + chh);
        }
    }
    });
    // work items of the largest blocks first: an item of 64x64 blocks runs longest (two 64-point passes over 4096 samples), and what is launched last is the tail
    int w = 0;
    for (int sz = 12; sz >= 2; sz--)
    for (int k = 0; k < NCLS; k++) {
        if (!cls_count[k] || ((k & 63) >> 3) + (k & 7) != sz) continue;
        const int per = itdq_group_size((k & 63) >> 3, k & 7);
        for (int f = 0; f < cls_count[k]; f += per) {
            wv[w].first = cls_first[k] + f; wv[w].count = (uint16_t)std::min(per, cls_count[k] - f);
            wv[w].log2w = (uint8_t)((k & 63) >> 3); wv[w].log2h = (uint8_t)(k & 7);
            wv[w].tr_v = (uint8_t)((k >> 6) / 3); wv[w].tr_h = (uint8_t)((k >> 6) % 3); wv[w].pad[0] = wv[w].pad[1] = 0; w++;
        }
    }
    memcpy(hs + o_ctu, b->ctu_cu_start, sz_ctu);
    {
        // the items carry their CU's record: the chain of a region- or tile-role wave is item -> reference windows, no CU-record fetch in between
        InterItem *const it = (InterItem *)(hs + o_iitem);
        const uint32_t *const ii = inter_items.data();
        const size_t n_items = inter_items.size();
        pool.run(nthr, [&, it, ii](int part) {
            for (size_t k = n_items * (size_t)part / nthr; k < n_items * (size_t)(part + 1) / nthr; k++) {
                if (ii[k] == 0xFFFFFFFFu) memset(&it[k], 0, sizeof(InterItem));
                else { it[k].pos = 0; it[k].cu = ii[k]; it[k].pad[0] = it[k].pad[1] = 0; it[k].rec = cus[ii[k]]; }
            }
        });
        if (!inter_work.empty()) memcpy(hs + o_iwork, inter_work.data(), sizeof(uint32_t) * inter_work.size());
    }
    if (b->n_coef && !coef_pinned) {                                   // the largest array (45 MB at 8K): in slices on the builder's threads
        const size_t bytes = sizeof(int16_t) * b->n_coef;
        pool.run(nthr, [&](int k) { const size_t a0 = k == 0 ? 0 : (bytes * k / nthr & ~(size_t)63), a1 = k + 1 == nthr ? bytes : (bytes * (k + 1) / nthr & ~(size_t)63);
                                    memcpy(hs + o_coef + a0, (const uint8_t *)b->coef + a0, a1 - a0); });
    }
    if (n_intra) memcpy(hs + o_intra, plan.recs.data(), sizeof(IntraRec) * (size_t)n_intra);
    if (n_deps) memcpy(hs + o_deps, plan.deps.data(), sizeof(uint32_t) * (size_t)n_deps);

    uint8_t *dbase = db->blk.d_base;
    db->d_cus = (CuRec *)(dbase + o_cus); db->d_ctu_start = (uint32_t *)(dbase + o_ctu); db->d_tbs = (TbRec *)(dbase + o_tbs);
    db->d_waves = (TbWave *)(dbase + o_wv); db->d_intra = (IntraRec *)(dbase + o_intra); db->d_intra_deps = (uint32_t *)(dbase + o_deps);
    db->d_aff_items = (AffItem *)(dbase + o_aff); db->d_cpmv = (int16_t *)(dbase + o_cpmv);
    db->d_dmvr_items = (DmvrItem *)(dbase + o_dmvr); db->d_dmvr_mv = (int16_t *)(dbase + o_dmv); db->d_owner = (uint32_t *)(dbase + o_own);
    db->d_inter_items = (InterItem *)(dbase + o_iitem); db->d_inter_work = (uint32_t *)(dbase + o_iwork); db->n_inter_work = (int)inter_work.size();
    db->d_coef = (int16_t *)(dbase + o_coef); db->d_resid = (int16_t *)(dbase + o_resid); db->d_intra_done = (uint32_t *)(dbase + o_done);
    // one copy: the staging block has the device layout (a pinned coefficient arena goes from the caller's buffer).  On the upload stream: the
    // copy overlaps the kernels of the pictures before; xgpu_batch_recon makes the kernel stream wait for `uploaded`
    BT("stage filled");
    if (segs) *segs = { { o_cus, sizeof(CuRec) * (size_t)n }, { o_ctu, sz_ctu }, { o_tbs, sizeof(TbRec) * (size_t)n_tb }, { o_wv, sizeof(TbWave) * (size_t)n_waves }, { o_intra, sizeof(IntraRec) * (size_t)n_intra },
                        { o_deps, sizeof(uint32_t) * (size_t)n_deps }, { o_aff, sizeof(AffItem) * (size_t)(n_aff_eif + n_aff_sub) }, { o_cpmv, sizeof(int16_t) * 12 * (size_t)n_aff },
                        { o_dmvr, sizeof(DmvrItem) * (size_t)n_dmvr }, { o_own, sz_own }, { o_coef, coef_pinned ? 0 : sizeof(int16_t) * b->n_coef },
                        { o_iitem, sizeof(InterItem) * inter_items.size() }, { o_iwork, sizeof(uint32_t) * inter_work.size() } };
    if (host_only) { *out = db; return XGPU_OK; }
    hipError_t e = hipMemcpyAsync(dbase, hs, coef_pinned ? o_coef : db->stage_bytes, hipMemcpyHostToDevice, c->up_stream);
    if (e == hipSuccess && coef_pinned) e = hipMemcpyAsync(dbase + o_coef, b->coef, sizeof(int16_t) * b->n_coef, hipMemcpyHostToDevice, c->up_stream);
    if (e == hipSuccess) e = hipMemsetAsync(db->d_intra_done, 0, sz_done, c->up_stream);
    if (e == hipSuccess) e = hipMemsetAsync(db->d_resid, 0, sz_coef, c->up_stream);
    if (e == hipSuccess) e = hipEventRecord(db->blk.uploaded, c->up_stream);
    if (e != hipSuccess) { snprintf(c->err, sizeof(c->err), "batch upload: %s", hipGetErrorString(e)); return fail(XGPU_ERR_UNEXPECTED); }
    BT("uploads queued");
    *out = db;
    return XGPU_OK;
}

int xgpu_batch_create(xgpu_ctx *c, const xgpu_cu_batch *b, xgpu_dbatch **out) { return batch_build(c, b, out, false, NULL); }

// Test shim (no device, no HIP call): the host batch builder alone on `threads` builder threads -> digest[k] = FNV-1a of array k of the staging block (CU records, CTU
// starts, TB records, work items, intra records, dependency lists, affine tiles, control points, DMVR sub-blocks, owner map, coefficients), info as xgpu_batch_info,
// *ms = the builder's wall time.  What the CPU suite uses to pin the builder (goldens of the digests, independence of the thread count).
int xgpu_test_build_batch(const xgpu_seq_params *sp, const xgpu_cu_batch *b, int threads, uint64_t digest[XGPU_TEST_BUILD_DIGESTS], int info[XGPU_BATCH_INFO_COUNT], double *ms)
{
    if (!sp || !b || !digest || threads < 1 || threads > 64) return XGPU_ERR_INVALID_ARGUMENT;
    if (sp->width <= 0 || sp->height <= 0 || (sp->width & 7) || (sp->height & 7) || sp->log2_ctu < 5 || sp->log2_ctu > 7) return XGPU_ERR_INVALID_ARGUMENT;
    xgpu_ctx *c = new xgpu_ctx();
    c->sp = *sp; c->sp.chroma_qp_table[0] = c->sp.chroma_qp_table[1] = NULL;
    c->builder_threads = threads; c->err[0] = 0;
    c->stream = c->up_stream = c->down_stream = c->side_stream = 0;
    c->w_scu = sp->width >> 2; c->h_scu = sp->height >> 2;
    const int ctu = 1 << sp->log2_ctu;
    c->w_ctu = (sp->width + ctu - 1) / ctu; c->h_ctu = (sp->height + ctu - 1) / ctu;
    xgpu_dbatch *db = NULL;
    std::vector<StageSeg> segs;
    const auto t0 = std::chrono::steady_clock::now();
    const int rc = batch_build(c, b, &db, true, &segs);
    if (ms) *ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    if (rc == XGPU_OK) {
        for (int k = 0; k < XGPU_TEST_BUILD_DIGESTS; k++) {
            uint64_t h = 1469598103934665603ull;
            if (k < (int)segs.size()) { const uint8_t *p = (const uint8_t *)db->h_stage + segs[(size_t)k].off; for (size_t i = 0; i < segs[(size_t)k].bytes; i++) { h ^= p[i]; h *= 1099511628211ull; } }
            digest[k] = h;
        }
        if (info) { info[0] = db->n_cu; info[1] = db->n_tb; info[2] = db->n_waves; info[3] = db->n_intra; info[4] = db->n_intra_l1; info[5] = db->n_levels; info[6] = db->n_dmvr; info[7] = db->n_aff_eif + db->n_aff_sub; }
        delete db;
    }
    delete c;
    return rc;
}

void xgpu_batch_destroy(xgpu_ctx *c, xgpu_dbatch *db)
{
    if (!db) return;
    if (db->blk.h_stage && !db->blk.d_base && !db->blk.uploaded) { delete db; return; }      // a host-only build (xgpu_test_build_batch) that failed half way: the block is the shim's own
    // No synchronisation: kernels still queued on the context's stream keep reading the block; whoever reuses it makes the upload stream wait
    // for the `done` event those kernels signal, and the host waits for `uploaded` before it touches the staging block.
    // The event is recorded here, not behind the batch's kernels: a marker between two kernels of a picture idles the device for ~6 us (profiles/round3_trace_window.txt),
    // and a batch that stays resident (decoded again and again) never needs it.
    if (db->blk.d_base && db->blk.h_stage && db->blk.uploaded && db->blk.done && db->blk.itdq_done && c) {
        if (db->prepared == 1) (void)hipStreamWaitEvent(c->stream, db->blk.itdq_done, 0);      // a residual pass on the side stream that nobody consumed
        if (db->used || db->prepared) (void)hipEventRecord(db->blk.done, c->stream);
        std::lock_guard<std::mutex> g(c->pool_mu); c->pool.push_back(db->blk);
    }
    else {
        if (c && c->stream) (void)hipStreamSynchronize(c->stream);
        if (db->blk.d_base) (void)hipFree(db->blk.d_base);
        if (db->blk.h_stage) (void)hipHostFree(db->blk.h_stage);
        if (db->blk.uploaded) (void)hipEventDestroy(db->blk.uploaded);
        if (db->blk.done) (void)hipEventDestroy(db->blk.done);
        if (db->blk.itdq_done) (void)hipEventDestroy(db->blk.itdq_done);
    }
    delete db;
}


int xgpu_set_builder_threads(xgpu_ctx *c, int n)
{
    ARGCHK(c, c != NULL); ARGCHK(c, n >= 1 && n <= 64);
    c->builder_threads = n;
    return XGPU_OK;
}

int xgpu_batch_info(xgpu_ctx *c, const xgpu_dbatch *db, int info[XGPU_BATCH_INFO_COUNT])
{
    ARGCHK(c, c != NULL); ARGCHK(c, db != NULL && info != NULL);
    info[0] = db->n_cu; info[1] = db->n_tb; info[2] = db->n_waves; info[3] = db->n_intra; info[4] = db->n_intra_l1; info[5] = db->n_levels;
    info[6] = db->n_dmvr; info[7] = db->n_aff_eif + db->n_aff_sub;
    return XGPU_OK;
}

int xgpu_batch_wait_upload(xgpu_ctx *c, xgpu_dbatch *db)
{
    ARGCHK(c, c != NULL); ARGCHK(c, db != NULL);
    HIPCHK(c, hipEventSynchronize(db->blk.uploaded));
    return XGPU_OK;
}

