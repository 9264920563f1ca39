// evc_cu.h - the part of the front end that works inside ONE tile, written once for both directions: the CU-level syntax as a template over the coder (arithmetic
// decoder or encoder), motion derivation (AMVP, merge / skip, MMVD, affine, HMVP, temporal candidates, host-side DMVR search), intra mode derivation, QP
// prediction, and the SCU maps of the tile's CUs.
#pragma once
#include "evc_hls.h"

namespace {
// The part of the front end that works inside ONE tile: CU syntax in both directions, motion derivation, the SCU maps of the tile's CUs.
// Everything a tile changes while it is coded lives here (context models, QP predictor, motion history) or in the tile's own region of the
// picture maps, so the tiles of a picture can be parsed by different threads (xevdm_dec_slice hands tiles to its thread pool the same way,
// src_main/xevdm.c:2640-2690).  The references name the state of the Stream the coder belongs to.
struct TileCoder {
    const Sps &sps;
    const Pps &pps;
    const Slice &sh;
    Picture &pic;
    const std::vector<const RefPic *> (&refp)[2];
    const int &poc;
    const std::vector<uint16_t> (&scan)[6][6], (&scan_inv)[6][6];
    Models models;
    int qp_prev = 0;
    int qp_coded = 0;                // core->cu_qp_delta_is_coded: the current quantisation group has sent its delta
    explicit TileCoder(Stream &s) : sps(s.sps), pps(s.pps), sh(s.sh), pic(s.pic), refp(s.refp), poc(s.poc), scan(s.scan), scan_inv(s.scan_inv) { history_reset(); }

    // motion vector predictor candidates of one list (xevd_get_motion, xevd_util.c:469-515; availability xevd_get_avail_inter :632-687):
    // left, up, up-right neighbour SCU (1,1 when not available) and the co-located list-0 motion of reference 0 of that list
    void mvp_candidates(const Cu &cu, int lidx, int16_t cand[4][2]) const
    {
        const int xs = cu.x >> 2, ys = cu.y >> 2, scuw = (1 << cu.log2w) >> 2, ws = pic.w_scu;
        const int scup = ys * ws + xs;
        auto take = [&](int k, bool ok, int s) {
            cand[k][0] = ok ? pic.mv[(size_t)s * 4 + lidx * 2] : (int16_t)1;
            cand[k][1] = ok ? pic.mv[(size_t)s * 4 + lidx * 2 + 1] : (int16_t)1;
        };
        // an intra-block-copy neighbour does not count on the left and above - but does above-right, where the Main library's availability test
        // only asks "coded and not intra" (xevdm_get_avail_inter, xevdm_util.c:1468-1503): its stored vector is the block vector (list 1: zero)
        take(0, xs > 0 && pic.same_tile(scup, scup - 1) && !pic.intra[scup - 1] && pic.cod[scup - 1] && !pic.ibc[scup - 1], scup - 1);
        take(1, ys > 0 && pic.same_tile(scup, scup - ws) && !pic.intra[scup - ws] && !pic.ibc[scup - ws], scup - ws);
        take(2, ys > 0 && xs + scuw < ws && pic.same_tile(scup, scup - ws + scuw) && pic.cod[scup - ws + scuw] && !pic.intra[scup - ws + scuw], scup - ws + scuw);
        const RefPic *col = refp[lidx].empty() ? nullptr : refp[lidx][0];
        cand[3][0] = col ? col->mv[(size_t)scup * 4] : (int16_t)0;
        cand[3][1] = col ? col->mv[(size_t)scup * 4 + 1] : (int16_t)0;
    }
    // ------------------------------------------------------------------------------------------------------------------------------
    // Main profile, sps->tool_admvp: merge candidates (skip and merge-mode CUs) and the predictor of explicitly coded motion.
    // No SUCO here, so the right-hand neighbours are never decoded before the CU: the avail_lr LR_10 / LR_00 branch of the reference.
    // ------------------------------------------------------------------------------------------------------------------------------
    struct Motion { int8_t refi[2]; int16_t mv[2][2]; };
    // sps->tool_hmvp: the motion of the last 23 inter CUs of the CTU row (XEVD_HISTORY_BUFFER; xevdm_hmvp_init at the start of every CTU row,
    // xevdm.c:553-566, 2499-2503; update_history_buffer_parse_affine after every inter CU, :657-778)
    Motion hist[23];
    int hist_cnt = 0;
    void history_reset() { hist_cnt = 0; for (Motion &m : hist) { m.refi[0] = m.refi[1] = -1; memset(m.mv, 0, sizeof(m.mv)); } }
    void history_push(const Cu &cu)
    {
        if (hist_cnt == 23) { for (int i = 1; i < 23; i++) hist[i - 1] = hist[i]; hist_cnt = 22; }
        Motion &m = hist[hist_cnt++];
        for (int l = 0; l < 2; l++) { m.refi[l] = (int8_t)cu.refi[l]; m.mv[l][0] = cu.mv[l][0]; m.mv[l][1] = cu.mv[l][1]; }
        if (cu.affine) for (int l = 0; l < 2; l++) { m.mv[l][0] = m.mv[l][1] = 0; if (cu.refi[l] >= 0) aff_centre(cu, l, m.mv[l]); }      // an affine CU leaves the vector at its centre
    }
    bool bi_applicable(const Cu &cu) const { return sh.type == XHOST_SLICE_B && (1 << cu.log2w) + (1 << cu.log2h) > 12; }      // xevdm_check_bi_applicability, xevdm_util.c:1083-1096
    // Which sides of the CU are parsed already: bit 0 the SCU left of its first SCU, bit 1 the one right of its first row (xevd_check_eco_nev_avail /
    // xevd_check_nev_avail, xevd_util.c:1139-1174: LR_00 0, LR_10 1, LR_01 2, LR_11 3).  The right side only ever is with sps_suco_flag.
    enum { LR_00 = 0, LR_10 = 1, LR_01 = 2, LR_11 = 3 };
    int avail_lr(const Cu &cu) const
    {
        const int ws = pic.w_scu, xs = cu.x >> 2, ys = cu.y >> 2, scuw = (1 << cu.log2w) >> 2, scup = ys * ws + xs;
        int lr = 0;
        if (xs > 0 && pic.same_tile(scup, scup - 1) && pic.cod[(size_t)scup - 1]) lr |= 1;
        if (sps.suco && xs + scuw < ws && pic.same_tile(scup, scup + scuw) && pic.cod[(size_t)scup + scuw]) lr |= 2;
        return lr;
    }
    // the five spatial neighbours (xevdm_check_motion_availability, xevdm_util.c:594-748): decoded, inter, not IBC.  LR_10 / LR_00: H, D, E, I, A (left of the
    // bottom-left SCU, above the top-right one, above-right, below-left, above-left); LR_01: their mirror images; LR_11: left, right, above, above-right, above-left
    void adm_neighbours(const Cu &cu, int neb[5], bool valid[5]) const
    {
        const int ws = pic.w_scu, hs = pic.h_scu, xs = cu.x >> 2, ys = cu.y >> 2, scuw = (1 << cu.log2w) >> 2, scuh = (1 << cu.log2h) >> 2, scup = ys * ws + xs;
        const int lr = avail_lr(cu);
        const bool le = xs > 0, ri = xs + scuw < ws, up = ys > 0, dn = ys + scuh < hs;
        bool in[5];
        if (lr == LR_11) {
            neb[0] = scup + (scuh - 1) * ws - 1; neb[1] = scup + (scuh - 1) * ws + scuw; neb[2] = scup - ws; neb[3] = scup - ws + scuw; neb[4] = scup - ws - 1;
            in[0] = le; in[1] = ri; in[2] = up; in[3] = up && ri; in[4] = le && up;
        } else if (lr == LR_01) {
            neb[0] = scup + (scuh - 1) * ws + scuw; neb[1] = scup - ws; neb[2] = scup - ws - 1; neb[3] = scup + scuh * ws + scuw; neb[4] = scup - ws + scuw;
            in[0] = ri; in[1] = up; in[2] = up && le; in[3] = dn && ri; in[4] = up && ri;
        } else {
            neb[0] = scup + (scuh - 1) * ws - 1; neb[1] = scup - ws + scuw - 1; neb[2] = scup - ws + scuw; neb[3] = scup + scuh * ws - 1; neb[4] = scup - ws - 1;
            in[0] = le; in[1] = up; in[2] = up && ri; in[3] = dn && le; in[4] = up && le;
        }
        for (int k = 0; k < 5; k++) valid[k] = in[k] && pic.same_tile(scup, neb[k]) && pic.cod[neb[k]] && !pic.intra[neb[k]] && !pic.ibc[neb[k]];      // the tile test first: another tile's maps may be written right now
    }
    static void scale_mv(int ratio, const int16_t in[2], int16_t out[2])      // scaling_mv, xevdm_util.c:180-190 (MVP_SCALING_PRECISION 5)
    {
        for (int d = 0; d < 2; d++) {
            int t = in[d] * ratio;
            t = t == 0 ? 0 : t > 0 ? (t + 16) >> 5 : -((-t + 16) >> 5);
            out[d] = (int16_t)std::min(std::max(t, -32768), 32767);
        }
    }
    // temporal candidate from the collocated picture at SCU `scu_col` (xevdm_get_mv_collocated, xevdm_util.c:3729-3818): 0 none, bit 0 list 0, bit 1 list 1
    int collocated(const Cu &cu, int scu_col, int16_t mvp[2][2]) const
    {
        int list = sh.type == XHOST_SLICE_P ? 0 : 1, ref = 0, src = 0;
        if (sh.tmvp_assigned) { list = sh.col_list; ref = sh.col_ref; src = sh.col_src_list; }
        memset(mvp, 0, sizeof(int16_t) * 4);
        if (ref >= (int)refp[list].size()) return 0;
        const RefPic *col = refp[list][ref];
        if (col->refi.empty()) return 0;
        const int dpoc[2] = { refp[0].empty() ? 0 : poc - refp[0][0]->poc, refp[1].empty() ? 0 : poc - refp[1][0]->poc };
        int have[2] = { 0, 0 };
        if (!sh.tmvp_assigned) {
            for (int l = 0; l < 2; l++) {
                const int r = col->refi[(size_t)scu_col * 2 + l];
                if (r < 0 || r >= 16) continue;
                const int dco = col->poc - col->list_poc[r];
                if (dco == 0) continue;
                have[l] = 1;
                scale_mv((dpoc[l] << 5) / dco, &col->mv[(size_t)scu_col * 4 + l * 2], mvp[l]);
            }
        } else {
            const int r = col->refi[(size_t)scu_col * 2 + src];
            const int dco = (r >= 0 && r < 16) ? col->poc - col->list_poc[r] : 0;
            if (dco != 0) {
                have[0] = have[1] = 1;
                for (int l = 0; l < 2; l++) scale_mv((dpoc[l] << 5) / dco, &col->mv[(size_t)scu_col * 4 + src * 2], mvp[l]);
            }
        }
        // xevdm_clip_mv_pic (xevdm_util.c:1409-1421) at the CU's position - samples and quarter samples mixed exactly like the reference does
        const int x = cu.x, y = cu.y, max_x = 144 + (pic.w_scu << 2) - 1, max_y = 144 + (pic.h_scu << 2) - 1, mn = -144;      // PIC_PAD_SIZE_L = MAX_CU_SIZE + 16
        for (int l = 0; l < 2; l++) { if (x + mvp[l][0] < mn) mvp[l][0] = (int16_t)-(x + mn); }
        for (int l = 0; l < 2; l++) { if (y + mvp[l][1] < mn) mvp[l][1] = (int16_t)-(y + mn); }
        for (int l = 0; l < 2; l++) { if (x + mvp[l][0] > max_x) mvp[l][0] = (int16_t)(max_x - x); }
        for (int l = 0; l < 2; l++) { if (y + mvp[l][1] > max_y) mvp[l][1] = (int16_t)(max_y - y); }
        return have[0] | (have[1] << 1);
    }
    // xevdm_get_motion_merge_main (xevdm_util.c:1169-1391) without the history candidates (sps->tool_hmvp off): up to 6 candidates (4 for CUs of 32 samples)
    // refined: the spatial candidates come from the refined map (ctx->map_mv) - the list xevdm_get_mmvd_mvp_list builds for an MMVD CU (xevdm_util.c:246-247);
    // the ordinary merge list reads the CUs' own vectors (map_unrefined_mv for refined neighbours, :1212-1216)
    void merge_candidates(const Cu &cu, Motion cand[6], bool refined = false) const
    {
        const std::vector<int16_t> &map_mv = (refined && !pic.mv_ref.empty()) ? pic.mv_ref : pic.mv;
        const int ws = pic.w_scu, hs = pic.h_scu, xs = cu.x >> 2, ys = cu.y >> 2, cuw = 1 << cu.log2w, cuh = 1 << cu.log2h, scup = ys * ws + xs;
        const int max_n = cuw * cuh <= 32 ? 4 : 6;
        const bool is_b = sh.type == XHOST_SLICE_B, bi = bi_applicable(cu);
        for (int k = 0; k < 6; k++) { cand[k].refi[0] = cand[k].refi[1] = -1; memset(cand[k].mv, 0, sizeof(cand[k].mv)); }
        int cnt = 0;
        auto insert = [&](const int8_t r[2], const int16_t *mv) {       // xevdm_get_merge_insert_mv + check_redundancy
            Motion &d = cand[cnt];
            d.refi[0] = r[0] >= 0 ? r[0] : (int8_t)-1; d.mv[0][0] = mv[0]; d.mv[0][1] = mv[1];
            if (is_b) {
                if (r[0] >= 0 && !bi) { d.refi[1] = -1; d.mv[1][0] = d.mv[1][1] = 0; }
                else { d.refi[1] = r[1] >= 0 ? r[1] : (int8_t)-1; d.mv[1][0] = mv[2]; d.mv[1][1] = mv[3]; }
            }
            bool dup = false;
            for (int i = cnt - 1; i >= 0 && !dup; i--)
                dup = d.refi[0] == cand[i].refi[0] && d.mv[0][0] == cand[i].mv[0][0] && d.mv[0][1] == cand[i].mv[0][1] &&
                      (!is_b || (d.refi[1] == cand[i].refi[1] && d.mv[1][0] == cand[i].mv[1][0] && d.mv[1][1] == cand[i].mv[1][1]));
            if (!dup) cnt++;
            return !dup;
        };
        int neb[5]; bool valid[5];
        adm_neighbours(cu, neb, valid);
        for (int k = 0; k < 5; k++) {
            if (valid[k]) insert(&pic.refi[(size_t)neb[k] * 2], &map_mv[(size_t)neb[k] * 4]);
            if (cnt == max_n - 1) break;
        }
        // temporal: the centre of the CU on the 8x8 grid, else below, else to the right (inside the CTU row / column)
        bool tmvp_added = false;
        auto temporal = [&](int scu_col) -> bool {      // true: candidate list complete
            int16_t t[2][2];
            const int av = collocated(cu, scu_col, t);
            if (!av) return false;
            const int8_t r[2] = { (int8_t)((av & 1) ? 0 : -1), (int8_t)((av & 2) ? 0 : -1) };
            tmvp_added = insert(r, &t[0][0]);
            return cnt >= max_n;
        };
        if (temporal(((xs + (cuw >> 3)) >> 1 << 1) + ((ys + (cuh >> 3)) >> 1 << 1) * ws)) return;
        const int ye = ys + (cuh >> 2) - 1, ctu = 6;
        if (avail_lr(cu) == LR_01) {        // the right side is there, the left one is not: below the bottom-left SCU, then left of it (xevd_get_right_below_scup_merge_suco, :1030-1057)
            const int xl = xs - 1;
            if (!tmvp_added && ye + 1 < hs && ((ye + 1) << 2 >> ctu) == (ye << 2 >> ctu))
                if (temporal(((ye + 1) >> 1 << 1) * ws + ((xl + 1) >> 1 << 1))) return;
            if (!tmvp_added && xl >= 0 && ((xl + 1) << 2 >> ctu) == (xl << 2 >> ctu))
                if (temporal((ye >> 1 << 1) * ws + (xl >> 1 << 1))) return;
        } else {
            const int xe = xs + (cuw >> 2) - 1;
            if (!tmvp_added && ye + 1 < hs && ((ye + 1) << 2 >> ctu) == (ye << 2 >> ctu))
                if (temporal(((ye + 1) >> 1 << 1) * ws + (xe >> 1 << 1))) return;
            if (!tmvp_added && xe + 1 < ws && ((xe + 1) << 2 >> ctu) == (xe << 2 >> ctu))
                if (temporal((ye >> 1 << 1) * ws + ((xe + 1) >> 1 << 1))) return;
        }
        // every fourth entry of the history, newest first (with tool_hmvp off the buffer is empty)
        for (int k = 3; k <= std::min(hist_cnt, max_n == 4 ? 15 : 23); k += 4) {
            insert(hist[hist_cnt - k].refi, &hist[hist_cnt - k].mv[0][0]);
            if (cnt >= max_n) return;
        }
        if (bi) {       // combinations of the list-0 part of one candidate with the list-1 part of another
            static const int p0[20] = { 0, 1, 0, 2, 1, 2, 0, 3, 1, 3, 2, 3, 0, 4, 1, 4, 2, 4, 3, 4 }, p1[20] = { 1, 0, 2, 0, 2, 1, 3, 0, 3, 1, 3, 2, 4, 0, 4, 1, 4, 2, 4, 3 };
            const int cur = cnt;
            for (int i = 0; i < cur * (cur - 1) && cnt != max_n; i++) {
                const Motion a = cand[p0[i]], b = cand[p1[i]];
                if (a.refi[0] >= 0 && b.refi[1] >= 0) {
                    cand[cnt].refi[0] = a.refi[0]; cand[cnt].mv[0][0] = a.mv[0][0]; cand[cnt].mv[0][1] = a.mv[0][1];
                    cand[cnt].refi[1] = b.refi[1]; cand[cnt].mv[1][0] = b.mv[1][0]; cand[cnt].mv[1][1] = b.mv[1][1];
                    cnt++;
                }
            }
            if (cnt == max_n) return;
        }
        for (int k = cnt; k < max_n; k++) { cand[k].refi[0] = 0; cand[k].refi[1] = bi ? 0 : -1; memset(cand[k].mv, 0, sizeof(cand[k].mv)); }
        (void)scup;
    }
    // mmvd_group_idx (only with the slice's group flag and above 32 samples), mmvd_merge_idx, mmvd_distance_idx, mmvd_direction_idx (xevdm_eco_mmvd_data, xevdm_eco.c:767-812)
    template <class C> void code_mmvd_idx(C &c, Cu &cu)
    {
        int grp = cu.mmvd_idx >> 7, base = (cu.mmvd_idx >> 5) & 3, dist = (cu.mmvd_idx >> 2) & 7, dir = cu.mmvd_idx & 3;
        if (sh.mmvd_group && (1 << (cu.log2w + cu.log2h)) > 32) {
            if (c.bin(grp > 0, models.mmvd_group_idx[0])) grp = 1 + c.bin(grp > 1, models.mmvd_group_idx[1]); else grp = 0;
        } else grp = 0;
        base = sym_trunc_unary(c, base, models.mmvd_merge_idx, 3, 4);
        dist = sym_trunc_unary(c, dist, models.mmvd_dist_idx, 7, 8);
        dir = (c.bin(dir >> 1, models.mmvd_dir_idx[0]) << 1) | c.bin(dir & 1, models.mmvd_dir_idx[1]);
        cu.mmvd_idx = (grp << 7) | (base << 5) | (dist << 2) | dir;
    }
    // merge with vector difference (xevdm_get_mmvd_motion, xevdm_util.c:4682-4716; xevdm_get_mmvd_mvp_list :191-592, for ONE index): base candidate
    // `base` of the merge list, turned into the prediction type of its group (bi-predictive / list 0 / list 1, the missing list mirrored and scaled by
    // POC distance; P slices: the same, another or a third reference), plus an offset of 1 .. 128 quarter samples in one of four directions, scaled
    // between the lists by their POC distances and mirrored when the references lie on either side
    void mmvd_motion(Cu &cu) const
    {
        Motion cand[6];
        merge_candidates(cu, cand, true);
        const int grp = cu.mmvd_idx >> 7, base = (cu.mmvd_idx >> 5) & 3, kk = cu.mmvd_idx & 31;
        const bool is_b = sh.type == XHOST_SLICE_B, small = (1 << (cu.log2w + cu.log2h)) <= 32;
        auto rpoc = [&](int l, int r) -> int { return (r >= 0 && r < (int)refp[l].size()) ? refp[l][(size_t)r]->poc : 0; };      // REF_SET
        // POC distances are non-zero for the lists build_ref_lists accepts; a REF_SET miss (rpoc = 0 at POC 0) must still not trap in the parser
        auto sdiv = [](int a, int b) -> int { return b ? a / b : 0; };
        auto scaled = [&](int w, int v, int sg) -> int { return std::min(std::max(sg * ((abs(w * v) + 16) >> 5), -32768), 32767); };
        // base_mv_t: the candidate (P slices take list 1 of candidate 0: unused), types per group
        int t[2][3] = { { cand[base].mv[0][0], cand[base].mv[0][1], cand[base].refi[0] },
                        { is_b ? cand[base].mv[1][0] : cand[0].mv[1][0], is_b ? cand[base].mv[1][1] : cand[0].mv[1][1], is_b ? cand[base].refi[1] : cand[0].refi[1] } };
        int b[2][3] = { { t[0][0], t[0][1], t[0][2] }, { t[1][0], t[1][1], t[1][2] } };      // base_mv: starts as the candidate
        int pm[3][3] = { { 0 } }, type[3];
        const int n0 = (int)refp[0].size(), n1 = (int)refp[1].size();
        if (t[0][2] >= 0 && t[1][2] >= 0) { type[0] = 0; type[1] = 1; type[2] = 2; }
        else if (t[0][2] >= 0) {
            if (!is_b) {
                type[0] = type[1] = type[2] = 1;
                pm[0][2] = t[0][2];
                pm[1][2] = n0 == 1 ? t[0][2] : !t[0][2];
                pm[2][2] = n0 < 3 ? t[0][2] : (t[0][2] < 2 ? 2 : 1);
                pm[0][0] = t[0][0]; pm[0][1] = t[0][1];
                if (n0 == 1) { pm[1][0] = t[0][0] + 3; pm[1][1] = t[0][1]; pm[2][0] = t[0][0] - 3; pm[2][1] = t[0][1]; }
                else {
                    for (int g = 1; g <= (n0 == 2 ? 1 : 2); g++) {
                        const int w = sdiv((poc - rpoc(0, pm[0][2])) << 5, poc - rpoc(0, pm[g][2]));
                        pm[g][0] = scaled(w, t[0][0], 1); pm[g][1] = scaled(w, t[0][1], 1);
                    }
                    if (n0 == 2) { pm[2][0] = t[0][0] - 3; pm[2][1] = t[0][1]; }
                }
            } else {
                type[0] = 1; type[1] = 0; type[2] = 2;
                const int p0 = rpoc(0, t[0][2]);
                t[1][2] = (n1 > 1 && rpoc(1, 1) - poc == poc - p0) ? 1 : 0;
                const int w = sdiv((poc - rpoc(1, t[1][2])) << 5, poc - p0);
                t[1][0] = scaled(w, t[0][0], w * t[0][0] < 0 ? -1 : 1); t[1][1] = scaled(w, t[0][1], w * t[0][1] < 0 ? -1 : 1);
            }
        } else if (t[1][2] >= 0) {
            type[0] = 2; type[1] = 0; type[2] = 1;
            const int p1 = rpoc(1, t[1][2]);
            t[0][2] = (n0 > 1 && rpoc(0, 1) - poc == poc - p1) ? 1 : 0;
            const int w = sdiv((poc - rpoc(0, t[0][2])) << 5, poc - p1);
            t[0][0] = scaled(w, t[1][0], w * t[1][0] < 0 ? -1 : 1); t[0][1] = scaled(w, t[1][1], w * t[1][1] < 0 ? -1 : 1);
        } else type[0] = type[1] = type[2] = 3;
        if (small) type[0] = 1;
        switch (type[grp]) {
        case 0: for (int l = 0; l < 2; l++) for (int d = 0; d < 3; d++) b[l][d] = t[l][d]; break;
        case 1: if (!is_b) { b[0][0] = pm[grp][0]; b[0][1] = pm[grp][1]; b[0][2] = pm[grp][2]; } else { b[0][0] = t[0][0]; b[0][1] = t[0][1]; b[0][2] = t[0][2]; } b[1][2] = -1; break;
        case 2: b[0][2] = -1; b[1][0] = t[1][0]; b[1][1] = t[1][1]; b[1][2] = t[1][2]; break;
        default: b[0][2] = b[1][2] = -1; break;
        }
        const int r0 = b[0][2], r1 = b[1][2], step = 1 << (kk >> 2);
        int sign = 1, d0 = step, d1 = step;
        if (r0 != -1 && r1 != -1) {
            const int p0 = rpoc(0, r0), p1 = rpoc(1, r1);
            if (is_b && (p0 - poc) * (poc - p1) > 0) sign = -1;
            if (abs(p1 - poc) >= abs(p0 - poc)) d0 = std::min(std::max((sdiv(abs(p0 - poc) << 5, abs(p1 - poc)) * step + 16) >> 5, -32768), 32767);
            else d1 = std::min(std::max((sdiv(abs(p1 - poc) << 5, abs(p0 - poc)) * step + 16) >> 5, -32768), 32767);
        }
        const int dir = kk & 3, s0 = (dir & 1) ? -d0 : d0, s1 = ((dir & 1) ? -d1 : d1) * sign;
        const int real[2][2] = { { b[0][0] + (dir < 2 ? s0 : 0), b[0][1] + (dir < 2 ? 0 : s0) }, { b[1][0] + (dir < 2 ? s1 : 0), b[1][1] + (dir < 2 ? 0 : s1) } };
        cu.refi[0] = r0; cu.mv[0][0] = (int16_t)real[0][0]; cu.mv[0][1] = (int16_t)real[0][1];
        if (is_b) { cu.refi[1] = r1; cu.mv[1][0] = (int16_t)real[1][0]; cu.mv[1][1] = (int16_t)real[1][1]; }
        else { cu.refi[1] = -1; cu.mv[1][0] = cu.mv[1][1] = 0; }
    }
    // skip / merge-mode motion = candidate `idx` (xevd_get_skip_motion / xevd_get_direct_motion, xevdm.c:800-883); entries past the list stay "no reference, zero"
    void merge_motion(Cu &cu, int idx) const
    {
        Motion cand[6];
        merge_candidates(cu, cand);
        const Motion &m = cand[std::min(std::max(idx, 0), 5)];
        cu.refi[0] = m.refi[0]; cu.mv[0][0] = m.mv[0][0]; cu.mv[0][1] = m.mv[0][1];
        if (sh.type == XHOST_SLICE_P) { cu.refi[1] = -1; cu.mv[1][0] = cu.mv[1][1] = 0; }
        else { cu.refi[1] = m.refi[1]; cu.mv[1][0] = m.mv[1][0]; cu.mv[1][1] = m.mv[1][1]; }
    }
    // the fallback motion of a list among the first two neighbours (xevdm_get_default_motion, xevdm_util.c:783-867, no history): the one with `cur_refi`, else any
    void default_motion(const int neb[5], const bool valid[5], int cur_refi, int l, int &refi, int16_t mv[2]) const
    {
        refi = 0; mv[0] = mv[1] = 0;
        for (int pass = 0; pass < 2; pass++)
            for (int k = 0; k < 2; k++) {
                if (!valid[k]) continue;
                const int r = pic.refi[(size_t)neb[k] * 2 + l];
                if (r >= 0 && (pass == 1 || r == cur_refi)) { refi = r; mv[0] = pic.mv[(size_t)neb[k] * 4 + l * 2]; mv[1] = pic.mv[(size_t)neb[k] * 4 + l * 2 + 1]; return; }
            }
        if (!sps.tool_hmvp) return;
        for (int pass = 0; pass < 2; pass++)                       // ... then the four newest history entries the same way
            for (int k = 1; k <= std::min(hist_cnt, 4); k++) {
                const int r = hist[hist_cnt - k].refi[l];
                if (r >= 0 && (pass == 1 || r == cur_refi)) { refi = r; mv[0] = hist[hist_cnt - k].mv[l][0]; mv[1] = hist[hist_cnt - k].mv[l][1]; return; }
            }
    }
    // reference index of a list of a bi-predicted CU that does not code it (bi_idx FL0 / FL1: xevdm_get_first_refi, xevdm_util.c:750-781), resolution index 0
    int first_refi(const Cu &cu, int l, int mvr) const
    {
        int neb[5], dref; bool valid[5]; int16_t dmv[2];
        adm_neighbours(cu, neb, valid);
        default_motion(neb, valid, 0, l, dref, dmv);
        if (valid[mvr] && pic.refi[(size_t)neb[mvr] * 2 + l] >= 0) return pic.refi[(size_t)neb[mvr] * 2 + l];      // the neighbour position is coupled with the resolution index
        return dref;
    }
    // the predictor of explicitly coded motion at resolution index 0 (xevdm_get_motion_from_mvr, xevdm_util.c:869-951): neighbour H, scaled to the
    // CU's reference when it points elsewhere, else the fallback motion
    void mvr_predictor(const Cu &cu, int l, int cur_refi, int mvr, int16_t mvp[2]) const
    {
        int neb[5], dref; bool valid[5]; int16_t dmv[2];
        adm_neighbours(cu, neb, valid);
        default_motion(neb, valid, cur_refi, l, dref, dmv);
        const int n = (int)refp[l].size(), pc = refp[l][std::min(std::max(cur_refi, 0), n - 1)]->poc;
        auto ratio = [&](int r) -> int { const int t0 = poc - refp[l][std::min(std::max(r, 0), n - 1)]->poc; return t0 ? ((poc - pc) * 32) / t0 : 0; };
        int r = valid[mvr] ? (int)pic.refi[(size_t)neb[mvr] * 2 + l] : -1;
        if (r >= 0) {
            const int16_t *m = &pic.mv[(size_t)neb[mvr] * 4 + l * 2];
            if (r == cur_refi) { mvp[0] = m[0]; mvp[1] = m[1]; } else scale_mv(ratio(r), m, mvp);
        } else {
            if (dref == cur_refi) { mvp[0] = dmv[0]; mvp[1] = dmv[1]; } else scale_mv(ratio(dref), dmv, mvp);
        }
        const int rnd = mvr > 0 ? 1 << (mvr - 1) : 0;             // the predictor on the grid of the resolution, rounded away from zero at the half
        for (int d = 0; d < 2; d++) mvp[d] = (int16_t)(mvp[d] >= 0 ? ((mvp[d] + rnd) >> mvr) << mvr : -(((-mvp[d] + rnd) >> mvr) << mvr));
    }

    // temporal direct motion of a B CU (xevd_get_mv_dir, xevd_util.c:540-566; call site xevd.c:713-717): the list-0 motion the
    // co-located picture (reference 0 of list 1) stored at the CU's bottom-right SCU, scaled by POC distances (C division)
    void direct_motion(Cu &cu) const
    {
        const int ws = pic.w_scu, scup = (cu.y >> 2) * ws + (cu.x >> 2);
        const int c_scu = scup + (((1 << cu.log2w) >> 2) - 1) + (((1 << cu.log2h) >> 2) - 1) * ws;
        const RefPic *r0 = refp[0][0], *col = refp[1][0];
        const int mvx = col->mv[(size_t)c_scu * 4], mvy = col->mv[(size_t)c_scu * 4 + 1];
        const int dco = col->poc - col->list0_poc, d0 = poc - r0->poc, d1 = col->poc - poc;
        cu.refi[0] = cu.refi[1] = 0;
        if (dco == 0) { memset(cu.mv, 0, sizeof(cu.mv)); return; }
        cu.mv[0][0] = (int16_t)(d0 * mvx / dco); cu.mv[0][1] = (int16_t)(d0 * mvy / dco);
        cu.mv[1][0] = (int16_t)(-d1 * mvx / dco); cu.mv[1][1] = (int16_t)(-d1 * mvy / dco);
    }
    // code-number table of the luma intra mode (xevd_get_mpm_b, xevd_ipred.c:678-692): neighbours count when intra and already parsed
    const uint8_t *mpm_list(const Cu &cu) const
    {
        const int xs = cu.x >> 2, ys = cu.y >> 2, ws = pic.w_scu, scup = ys * ws + xs;
        int l = 0, u = 0;
        if (xs > 0 && pic.same_tile(scup, scup - 1) && pic.intra[scup - 1] && pic.cod[scup - 1]) l = pic.ipm[scup - 1] + 1;
        if (ys > 0 && pic.same_tile(scup, scup - ws) && pic.intra[scup - ws] && pic.cod[scup - ws]) u = pic.ipm[scup - ws] + 1;
        return k_mpm[l][u];
    }
    // tool_eipd: the two most probable modes, eight "extended" ones and the ordering of all 33 (xevdm_get_mpm, src_main/xevdm_ipred.c:
    // 320-767)
    void eipd_mpm(const Cu &cu, int mpm[2], int ext[8], int pims[33]) const
    {
        enum { DC = 0, PLN = 1, BI = 2, VER = 12, HOR = 24, DIA_R = 18, DIA_L = 6, DIA_U = 30, CNT = 33 };
        const int xs = cu.x >> 2, ys = cu.y >> 2, ws = pic.w_scu, scup = ys * ws + xs;
        int l = DC, u = DC, r = DC;
        bool vl = false, vu = false, vr = false;
        if (xs > 0 && pic.same_tile(scup, scup - 1) && pic.intra[scup - 1] && pic.cod[scup - 1]) { l = pic.ipm[scup - 1]; vl = true; }
        if (ys > 0 && pic.same_tile(scup, scup - ws) && pic.intra[scup - ws] && pic.cod[scup - ws]) { u = pic.ipm[scup - ws]; vu = true; }
        const int scuw = (1 << cu.log2w) >> 2;
        if (sps.suco && xs + scuw < ws && pic.same_tile(scup, scup + scuw) && pic.intra[scup + scuw] && pic.cod[scup + scuw]) {
            // a right-hand neighbour decoded before the CU (SUCO) stands in for a missing / repeated one, or joins as a third mode (:348-377)
            r = pic.ipm[scup + scuw];
            if (vl && vu) { if (l == u) u = r; else vr = true; }
            else if (!vl) l = r;
            else u = r;
            if (vr && (l == r || u == r)) vr = false;
        }
        mpm[0] = std::min(l, u); mpm[1] = std::max(l, u);
        if (mpm[0] == mpm[1]) { mpm[0] = DC; mpm[1] = mpm[1] == DC ? BI : mpm[1]; }
        const int m0 = mpm[0], m1 = mpm[1];
        // candidates of a list in order, those equal to a most probable mode or an earlier entry dropped, until eight are there
        auto fill = [&](int n, const int *list, int cnt) {
            for (int i = 0; i < cnt && n < 8; i++) {
                bool dup = list[i] == m0 || list[i] == m1;
                for (int j = 0; j < n && !dup; j++) dup = list[i] == ext[j];
                if (!dup) ext[n++] = list[i];
            }
        };
        auto below = [](int m) { return (m == 3 || m == 4) ? m + 1 : m - 2; };
        auto above = [](int m) { return (m == CNT - 1 || m == CNT - 2) ? m - 1 : m + 2; };
        if (vr) {                                             // three different modes around the CU (xevdm_ipred.c:388-625)
            if (m1 < 3) {
                ext[0] = m0 == DC ? (m1 == BI ? PLN : BI) : DC;
                if (r < 3) { const int e[7] = { VER, HOR, DIA_R, DIA_L, DIA_U, VER + 4, HOR - 4 }; memcpy(ext + 1, e, sizeof(e)); }
                else {
                    const int list[10] = { VER, HOR, DIA_R, PLN, DIA_L, DIA_U, VER + 4, HOR - 4, VER - 4, HOR + 4 };
                    ext[1] = r; ext[2] = below(r); ext[3] = above(r);
                    fill(4, list, 10);
                }
            } else if (m0 < 3) {
                ext[0] = m0 == PLN ? BI : (m0 == BI ? DC : BI);
                ext[1] = m0 == PLN ? DC : PLN;
                if (r < 3) {
                    if (m1 > CNT - 3)  { const int e[6] = { m1 == CNT - 1 ? CNT - 2 : CNT - 1, CNT - 3, CNT - 4, CNT - 5, HOR, DIA_R }; memcpy(ext + 2, e, sizeof(e)); }
                    else if (m1 < 5)   { const int e[6] = { m1 == 3 ? 4 : 3, 5, 6, 7, VER, DIA_R }; memcpy(ext + 2, e, sizeof(e)); }
                    else {
                        ext[2] = m1 + 2; ext[3] = m1 - 2; ext[4] = m1 + 1; ext[5] = m1 - 1;
                        if (m1 <= 23 && m1 >= 13) { ext[6] = m1 - 5; ext[7] = m1 + 5; }
                        else { ext[6] = m1 > 23 ? m1 - 5 : m1 + 5; ext[7] = m1 > 23 ? m1 - 10 : m1 + 10; }
                    }
                } else {
                    int list[15] = { below(r), above(r), below(m1), above(m1), (r + m1 + 1) >> 1, 0, 0, VER, HOR, DIA_R, PLN, DIA_L, DIA_U, VER + 4, HOR - 4 };
                    list[5] = (list[4] + r + 1) >> 1; list[6] = (list[4] + m1 + 1) >> 1;
                    ext[2] = r;
                    fill(3, list, 15);
                }
            } else if (r < 3) {
                int list[15] = { below(m0), m0 == CNT - 2 ? m0 - 1 : m0 + 2, m1 == 4 ? m1 + 1 : m1 - 2, above(m1), (m0 + m1 + 1) >> 1, 0, 0, VER, HOR, DIA_R, PLN, DIA_L, DIA_U, VER + 4, HOR - 4 };
                list[5] = (list[4] + m0 + 1) >> 1; list[6] = (list[4] + m1 + 1) >> 1;
                ext[0] = r; ext[1] = r == BI ? DC : BI;
                fill(2, list, 15);
            } else {
                const int list[16] = { below(m0), m0 == CNT - 2 ? m0 - 1 : m0 + 2, m1 == 4 ? m1 + 1 : m1 - 2, above(m1), below(r), above(r),
                                       r < m1 ? (m0 + r + 1) >> 1 : (m0 + m1 + 1) >> 1, r < m0 ? (m0 + m1 + 1) >> 1 : (m1 + r + 1) >> 1,
                                       VER, HOR, DIA_R, PLN, DIA_L, DIA_U, VER + 4, HOR - 4 };
                ext[0] = BI; ext[1] = DC; ext[2] = r;
                fill(3, list, 16);
            }
        } else if (m1 < 3) {                                         // two non-angular modes: the third one, then the main directions
            const int e[8] = { m0 == DC ? (m1 == BI ? PLN : BI) : DC, VER, HOR, DIA_R, DIA_L, DIA_U, VER + 4, HOR - 4 };
            memcpy(ext, e, sizeof(e));
        } else if (m0 < 3) {                                  // one angular mode: the other non-angular ones, then its neighbourhood
            ext[0] = m0 == PLN ? BI : (m0 == BI ? DC : BI);
            ext[1] = m0 == PLN ? DC : PLN;
            if (m1 > CNT - 3)  { const int e[6] = { m1 == CNT - 1 ? CNT - 2 : CNT - 1, CNT - 3, CNT - 4, CNT - 5, HOR, DIA_R }; memcpy(ext + 2, e, sizeof(e)); }
            else if (m1 < 5)   { const int e[6] = { m1 == 3 ? 4 : 3, 5, 6, 7, VER, DIA_R }; memcpy(ext + 2, e, sizeof(e)); }
            else {
                ext[2] = m1 + 2; ext[3] = m1 - 2; ext[4] = m1 + 1; ext[5] = m1 - 1;
                if (m1 <= 23 && m1 >= 13) { ext[6] = m1 - 5; ext[7] = m1 + 5; }
                else { ext[6] = m1 > 23 ? m1 - 5 : m1 + 5; ext[7] = m1 > 23 ? m1 - 10 : m1 + 10; }
            }
        } else {                                              // two angular modes: their neighbours and means, then the main directions
            int list[15] = { (m0 == 3 || m0 == 4) ? m0 + 1 : m0 - 2, m0 == CNT - 2 ? m0 - 1 : m0 + 2, m1 == 4 ? m1 + 1 : m1 - 2,
                             (m1 == CNT - 1 || m1 == CNT - 2) ? m1 - 1 : m1 + 2, (m0 + m1 + 1) >> 1, 0, 0,
                             VER, HOR, DIA_R, PLN, DIA_L, DIA_U, VER + 4, HOR - 4 };
            list[5] = (list[4] + m0 + 1) >> 1; list[6] = (list[4] + m1 + 1) >> 1;
            ext[0] = BI; ext[1] = DC;
            int n = 2;
            for (int i = 0; i < 15 && n < 8; i++) {
                bool dup = list[i] == m0 || list[i] == m1;
                for (int j = 0; j < n && !dup; j++) dup = list[i] == ext[j];
                if (!dup) ext[n++] = list[i];
            }
        }
        // the order of all modes: the two, the eight, then a fixed list (intra_mode_list, xevdm_ipred.c:307-318), duplicates dropped
        static const int k_default[33] = { DC, BI, VER, PLN, HOR, VER - 1, VER + 1, VER - 2, VER + 2, VER - 3, VER + 3, HOR - 1, HOR + 1, HOR - 2, HOR + 2,
                                           HOR - 3, HOR + 3, DIA_R, DIA_L, DIA_L - 3, DIA_L - 2, DIA_L - 1, DIA_U, DIA_U + 1, DIA_U + 2, VER + 5, VER + 4,
                                           HOR - 4, HOR - 5, VER - 5, VER - 4, HOR + 5, HOR + 4 };
        bool in[33] = { false };
        int n = 0;
        auto add = [&](int m) { if (m >= 0 && m < 33 && !in[m]) { in[m] = true; pims[n++] = m; } };
        add(m0); add(m1);
        for (int i = 0; i < 8; i++) add(ext[i]);
        for (int i = 0; i < 33; i++) add(k_default[i]);
    }
    void chroma_qps(int qp, int &qp_u, int &qp_v) const      // xevd_eco.c:663-666
    {
        const int off = 6 * (sps.bd_c - 8);
        const int iu = std::min(std::max(qp + sh.qp_u_offset, -off), 57), iv = std::min(std::max(qp + sh.qp_v_offset, -off), 57);
        if (sps.cqt) { qp_u = sps.cq[0][iu + off] + off; qp_v = sps.cq[1][iv + off] + off; return; }
        const int8_t *tbl = sps.tool_iqt ? k_chroma_qp_main : k_chroma_qp;
        qp_u = (iu >= 0 ? tbl[iu] : 0) + off;                 // entries below 0 of the default table are zero-initialised storage
        qp_v = (iv >= 0 ? tbl[iv] : 0) + off;
    }

    // ------------------------------------------------------------------------------------------------------------------------------
    // sps->tool_affine: control-point vectors of affine merge / affine inter CUs (xevd_get_affine_motion, src_main/xevdm.c:937-1013;
    // candidates src_main/xevdm_util.c:2145-3187).  The right-hand neighbours count where SUCO has them decoded before the CU (avail_lr LR_01 / LR_11).
    // ------------------------------------------------------------------------------------------------------------------------------
    static int aff_rnd(int v, int shift) { return (v + (1 << (shift - 1)) - (v >= 0)) >> shift; }      // xevdm_mv_rounding_s32, xevdm_util.c:1856-1861
    static int16_t clip16(int v) { return (int16_t)std::min(std::max(v, -32768), 32767); }
    // neighbour SCU n of the CU at scup: decoded, inter, same tile - and affine (model-based candidates) or not IBC (corner vectors)
    bool aff_nb(int scup, int n, bool inside, bool need_affine) const
    {
        if (!inside || !pic.same_tile(scup, n) || !pic.cod[n] || pic.intra[n]) return false;
        return need_affine ? pic.aff[n] != 0 : !pic.ibc[n];
    }
    // the model of the affine CU that covers SCU scun, evaluated at this CU's corners (xevdm_derive_affine_model_mv, xevdm_util.c:2270-2363)
    void aff_inherit(const Cu &cu, int scun, int l, int cp_num, int16_t mvp[3][2]) const
    {
        const int ws = pic.w_scu, a = pic.aff[(size_t)scun], nlw = (a >> 2) & 7, nlh = (a >> 5) & 7, nw = 1 << nlw, nh = 1 << nlh, tl = (int)pic.aff_tl[(size_t)scun];
        const int addr[4] = { tl, tl + (nw >> 2) - 1, tl + ((nh >> 2) - 1) * ws, tl + ((nh >> 2) - 1) * ws + (nw >> 2) - 1 };
        int nmv[4][2];
        for (int i = 0; i < 4; i++) { nmv[i][0] = pic.mv[(size_t)addr[i] * 4 + l * 2]; nmv[i][1] = pic.mv[(size_t)addr[i] * 4 + l * 2 + 1]; }
        const int nx = (tl % ws) << 2;
        int ny = (tl / ws) << 2;
        bool top = false;
        if ((ny + nh) % 64 == 0 && ny + nh == cu.y) {          // the neighbour sits in the CTU row above: only its bottom row of vectors is used (line buffer)
            top = true; ny += nh;
            nmv[0][0] = nmv[2][0]; nmv[0][1] = nmv[2][1]; nmv[1][0] = nmv[3][0]; nmv[1][1] = nmv[3][1];
        }
        const int dhx = (nmv[1][0] - nmv[0][0]) * (1 << (7 - nlw)), dhy = (nmv[1][1] - nmv[0][1]) * (1 << (7 - nlw));
        int dvx = -dhy, dvy = dhx;
        if (cp_num == 3 && !top) { dvx = (nmv[2][0] - nmv[0][0]) * (1 << (7 - nlh)); dvy = (nmv[2][1] - nmv[0][1]) * (1 << (7 - nlh)); }
        const int hb = nmv[0][0] * 128, vb = nmv[0][1] * 128;
        auto at = [&](int px, int py, int16_t out[2]) {
            out[0] = clip16(aff_rnd(dhx * px + dvx * py + hb, 7)); out[1] = clip16(aff_rnd(dhy * px + dvy * py + vb, 7));
        };
        at(cu.x - nx, cu.y - ny, mvp[0]);
        at(cu.x - nx + (1 << cu.log2w), cu.y - ny, mvp[1]);
        if (cp_num == 3) at(cu.x - nx, cu.y - ny + (1 << cu.log2h), mvp[2]);
    }
    // the two predictor candidates of an affine inter CU for list l / reference cur_refi (xevdm_get_affine_motion_scaling, xevdm_util.c:2367-2761)
    void aff_amvp(const Cu &cu, int l, int cur_refi, int vn, int16_t mvp[2][3][2]) const
    {
        const int ws = pic.w_scu, hs = pic.h_scu, xs = cu.x >> 2, ys = cu.y >> 2, scuw = (1 << cu.log2w) >> 2, scuh = (1 << cu.log2h) >> 2, scup = ys * ws + xs;
        memset(mvp, 0, sizeof(int16_t) * 2 * 3 * 2);
        int cnt = 0;
        auto model = [&](const int *nb, const bool *in, int n) {          // the first neighbour of a group that is affine and uses this reference
            for (int k = 0; k < n; k++)
                if (aff_nb(scup, nb[k], in[k], true) && pic.refi[(size_t)nb[k] * 2 + l] >= 0 && pic.refi[(size_t)nb[k] * 2 + l] == cur_refi) {
                    int16_t t[3][2] = { { 0, 0 }, { 0, 0 }, { 0, 0 } };
                    aff_inherit(cu, nb[k], l, vn, t);
                    memcpy(mvp[cnt++], t, sizeof(t));
                    return;
                }
        };
        { const int nb[2] = { scup + ws * scuh - 1, scup + ws * (scuh - 1) - 1 }; const bool in[2] = { xs > 0 && ys + scuh < hs, xs > 0 }; model(nb, in, 2); }      // A0, A1
        if (cnt >= 2) return;
        { const int nb[3] = { scup - ws + scuw, scup - ws + scuw - 1, scup - ws - 1 }; const bool in[3] = { ys > 0 && xs + scuw < ws, ys > 0, xs > 0 && ys > 0 }; model(nb, in, 3); }      // B0, B1, B2
        if (cnt >= 2) return;
        { const int nb[2] = { scup + ws * scuh + scuw, scup + ws * (scuh - 1) + scuw }; const bool in[2] = { xs + scuw < ws && ys + scuh < hs, xs + scuw < ws }; model(nb, in, 2); }      // C0, C1
        if (cnt >= 2) return;
        // corner vectors: the first neighbour of each corner that uses this reference
        auto corner = [&](const int *nb, const bool *in, int n, int16_t out[2]) -> bool {
            out[0] = out[1] = 0;
            for (int k = 0; k < n; k++)
                if (aff_nb(scup, nb[k], in[k], false) && pic.refi[(size_t)nb[k] * 2 + l] == cur_refi && cur_refi >= 0) {
                    out[0] = pic.mv[(size_t)nb[k] * 4 + l * 2]; out[1] = pic.mv[(size_t)nb[k] * 4 + l * 2 + 1];
                    return true;
                }
            return false;
        };
        int16_t lt[2], rt[2], lb[2], rb[2];
        const int nlt[3] = { scup - ws - 1, scup - ws, scup - 1 }; const bool ilt[3] = { xs > 0 && ys > 0, ys > 0, xs > 0 };
        const int nrt[3] = { scup - ws + scuw, scup - ws + scuw - 1, scup + scuw }; const bool irt[3] = { ys > 0 && xs + scuw < ws, ys > 0, xs + scuw < ws };
        const int nlb[2] = { scup + ws * scuh - 1, scup + ws * (scuh - 1) - 1 }; const bool ilb[2] = { xs > 0 && ys + scuh < hs, xs > 0 };
        const int nrb[2] = { scup + ws * scuh + scuw, scup + ws * (scuh - 1) + scuw }; const bool irb[2] = { xs + scuw < ws && ys + scuh < hs, xs + scuw < ws };
        const bool c_lt = corner(nlt, ilt, 3, lt), c_rt = corner(nrt, irt, 3, rt), c_lb = corner(nlb, ilb, 2, lb), c_rb = corner(nrb, irb, 2, rb);
        auto put = [&](const int16_t a[2], const int16_t b[2], const int16_t c2[2]) { memcpy(mvp[cnt][0], a, 4); memcpy(mvp[cnt][1], b, 4); memcpy(mvp[cnt][2], c2, 4); cnt++; };
        if (c_lt && c_rt && (vn == 2 || c_lb || c_rb)) {
            int16_t third[2] = { lb[0], lb[1] };
            if (!c_lb && c_rb) { third[0] = clip16(rb[0] + lt[0] - rt[0]); third[1] = clip16(rb[1] + lt[1] - rt[1]); }
            put(lt, rt, third);
        }
        if (cnt == 2) return;
        if (c_lb) put(lb, lb, lb); else if (c_rb) put(rb, rb, rb);        // translational candidates: left, (right,) above, above-left
        if (cnt == 2) return;
        if (c_rt) put(rt, rt, rt);
        if (cnt == 2) return;
        if (c_lt) put(lt, lt, lt);
    }
    // one constructed candidate from corner vectors (xevdm_derive_affine_constructed_candidate, xevdm_util.c:2145-2268)
    void aff_constructed(const Cu &cu, const int cp_valid[4], const int16_t cp_mv[2][4][2], const int cp_refi[2][4], const int *idx, int model, int vn,
                         int8_t refi[5][2], int16_t cpmv[5][2][3][2], int cpn[5], int &cnt) const
    {
        if (cnt >= 5) return;
        for (int i = 0; i < vn; i++) if (!cp_valid[idx[i]]) return;
        bool ok[2];
        for (int l = 0; l < 2; l++) {
            ok[l] = true;
            for (int i = 0; i < vn; i++) ok[l] = ok[l] && cp_refi[l][idx[i]] >= 0 && cp_refi[l][idx[i]] == cp_refi[l][idx[0]];
        }
        if (!ok[0] && !ok[1]) return;
        cpn[cnt] = vn;
        const int sh_hw = 7 + cu.log2w - cu.log2h;
        for (int l = 0; l < 2; l++) {
            memset(cpmv[cnt][l], 0, sizeof(cpmv[cnt][l]));
            refi[cnt][l] = -1;
            if (!ok[l]) continue;
            refi[cnt][l] = (int8_t)cp_refi[l][idx[0]];
            int t[4][2] = { { 0, 0 }, { 0, 0 }, { 0, 0 }, { 0, 0 } };
            for (int i = 0; i < vn; i++) { t[idx[i]][0] = cp_mv[l][idx[i]][0]; t[idx[i]][1] = cp_mv[l][idx[i]][1]; }
            switch (model) {      // to top-left, top-right (, bottom-left)
            case 1: t[2][0] = t[3][0] + t[0][0] - t[1][0]; t[2][1] = t[3][1] + t[0][1] - t[1][1]; break;
            case 2: t[1][0] = t[3][0] + t[0][0] - t[2][0]; t[1][1] = t[3][1] + t[0][1] - t[2][1]; break;
            case 3: t[0][0] = t[1][0] + t[2][0] - t[3][0]; t[0][1] = t[1][1] + t[2][1] - t[3][1]; break;
            case 5: {
                const int h = (t[2][1] - t[0][1]) * (1 << sh_hw) + t[0][0] * 128, v = -((t[2][0] - t[0][0]) * (1 << sh_hw)) + t[0][1] * 128;
                t[1][0] = aff_rnd(h, 7); t[1][1] = aff_rnd(v, 7);
                break; }
            default: break;
            }
            for (int i = 0; i < vn; i++) { cpmv[cnt][l][i][0] = clip16(t[i][0]); cpmv[cnt][l][i][1] = clip16(t[i][1]); }
        }
        cnt++;
    }
    // the five affine merge candidates (xevdm_get_affine_merge_candidate, xevdm_util.c:2763-3187)
    void aff_merge(const Cu &cu, int8_t refi[5][2], int16_t cpmv[5][2][3][2], int cpn[5]) const
    {
        const int ws = pic.w_scu, hs = pic.h_scu, xs = cu.x >> 2, ys = cu.y >> 2, scuw = (1 << cu.log2w) >> 2, scuh = (1 << cu.log2h) >> 2, scup = ys * ws + xs;
        const int lr = avail_lr(cu);
        const bool left_avail = (lr & 1) != 0, right_avail = (lr & 2) != 0;
        int cnt = 0;
        memset(cpmv, 0, sizeof(int16_t) * 5 * 2 * 3 * 2);
        {   // model based: A1, B1, B0, A0, B2 - one candidate per distinct affine neighbour CU; with only the right side decoded (LR_01) their mirror images
            const bool mir = lr == LR_01;
            const int nb[5] = { mir ? scup + ws * (scuh - 1) + scuw : scup + ws * (scuh - 1) - 1, mir ? scup - ws : scup - ws + scuw - 1, mir ? scup - ws - 1 : scup - ws + scuw,
                                mir ? scup + ws * scuh + scuw : scup + ws * scuh - 1, mir ? scup - ws + scuw : scup - ws - 1 };
            const bool le = xs > 0, ri = xs + scuw < ws, up = ys > 0, dn = ys + scuh < hs;
            const bool in[5] = { mir ? ri : le, up, up && (mir ? le : ri), dn && (mir ? ri : le), up && (mir ? ri : le) };
            bool valid[5];
            uint32_t tl[5] = { 0, 0, 0, 0, 0 };
            for (int k = 0; k < 5; k++) { valid[k] = aff_nb(scup, nb[k], in[k], true); if (valid[k]) tl[k] = pic.aff_tl[(size_t)nb[k]]; }
            if (valid[2] && valid[1] && tl[1] == tl[2]) valid[2] = false;
            if (valid[3] && valid[0] && tl[0] == tl[3]) valid[3] = false;
            if ((valid[4] && valid[0] && tl[4] == tl[0]) || (valid[4] && valid[1] && tl[4] == tl[1])) valid[4] = false;
            for (int k = 0; k < 5 && cnt < 5; k++) {
                if (!valid[k]) continue;
                cpn[cnt] = (pic.aff[(size_t)nb[k]] & 3) == 1 ? 2 : 3;
                for (int l = 0; l < 2; l++) {
                    refi[cnt][l] = pic.refi[(size_t)nb[k] * 2 + l];
                    if (refi[cnt][l] >= 0) aff_inherit(cu, nb[k], l, cpn[cnt], cpmv[cnt][l]); else refi[cnt][l] = -1;
                }
                cnt++;
            }
        }
        {   // constructed from the vectors at the four corners
            int16_t cp_mv[2][4][2];
            int cp_refi[2][4], cp_valid[4] = { 0, 0, 0, 0 };
            memset(cp_mv, 0, sizeof(cp_mv));
            for (int l = 0; l < 2; l++) for (int i = 0; i < 4; i++) cp_refi[l][i] = -1;
            auto spatial = [&](const int *nb, const bool *in, int n, int v) {
                for (int k = 0; k < n; k++)
                    if (aff_nb(scup, nb[k], in[k], false)) {
                        for (int l = 0; l < 2; l++) {
                            cp_refi[l][v] = pic.refi[(size_t)nb[k] * 2 + l];
                            cp_mv[l][v][0] = pic.mv[(size_t)nb[k] * 4 + l * 2]; cp_mv[l][v][1] = pic.mv[(size_t)nb[k] * 4 + l * 2 + 1];
                        }
                        cp_valid[v] = 1;
                        return;
                    }
            };
            auto temporal = [&](int scu_col, int v) {          // the co-located vectors at an 8x8-aligned position, reference 0 of each list
                int16_t t[2][2];
                const int have = collocated(cu, scu_col, t);
                for (int l = 0; l < 2; l++) {
                    const bool on = (have >> l) & 1 && (l == 0 || sh.type == XHOST_SLICE_B);
                    cp_refi[l][v] = on ? 0 : -1;
                    cp_mv[l][v][0] = on ? t[l][0] : (int16_t)0; cp_mv[l][v][1] = on ? t[l][1] : (int16_t)0;
                }
            };
            const bool same_ctu_row = (((ys + scuh) << 2) >> 6) == ((ys << 2) >> 6);
            { const int nb[3] = { scup - ws - 1, scup - ws, scup - 1 }; const bool in[3] = { xs > 0 && ys > 0, ys > 0, xs > 0 }; spatial(nb, in, 3, 0); }
            { const int nb[3] = { scup - ws + scuw, scup - ws + scuw - 1, scup + scuw }; const bool in[3] = { ys > 0 && xs + scuw < ws, ys > 0, xs + scuw < ws }; spatial(nb, in, 3, 1); }
            if (left_avail) { const int nb[2] = { scup + ws * scuh - 1, scup + ws * (scuh - 1) - 1 }; const bool in[2] = { xs > 0 && ys + scuh < hs, xs > 0 }; spatial(nb, in, 2, 2); }
            else {
                if (xs > 0 && ys + scuh < hs && same_ctu_row && pic.same_tile(scup, scup + ws * scuh - 1) && pic.same_tile(scup, scup - 1))
                    temporal(((xs - 1) >> 1 << 1) + ((ys + scuh) >> 1 << 1) * ws, 2);
                if (cp_refi[0][2] >= 0 || cp_refi[1][2] >= 0) cp_valid[2] = 1;
            }
            if (right_avail) {      // the corner takes the neighbour's vectors; it counts when one of them uses a reference (cp_valid[3] is only set from cp_refi, :3132-3135)
                const int nb[2] = { scup + ws * scuh + scuw, scup + ws * (scuh - 1) + scuw }; const bool in[2] = { xs + scuw < ws && ys + scuh < hs, xs + scuw < ws };
                const int keep = cp_valid[3];
                spatial(nb, in, 2, 3);
                cp_valid[3] = keep;
            } else {
                const int col = ((xs + scuw) >> 1 << 1) + ((ys + scuh) >> 1 << 1) * ws;
                if (xs + scuw < ws && ys + scuh < hs && same_ctu_row && pic.same_tile(scup, col)) temporal(col, 3);
            }
            if (cp_refi[0][3] >= 0 || cp_refi[1][3] >= 0) cp_valid[3] = 1;
            static const int models[6][3] = { { 0, 1, 2 }, { 0, 1, 3 }, { 0, 2, 3 }, { 1, 2, 3 }, { 0, 1, 0 }, { 0, 2, 0 } };
            static const int vns[6] = { 3, 3, 3, 3, 2, 2 };
            for (int m = 0; m < 6; m++) aff_constructed(cu, cp_valid, cp_mv, cp_refi, models[m], m, vns[m], refi, cpmv, cpn, cnt);
        }
        for (; cnt < 5; cnt++) { cpn[cnt] = 2; refi[cnt][0] = 0; refi[cnt][1] = sh.type == XHOST_SLICE_B ? 0 : -1; }      // zero candidates
    }
    // affine merge: candidate idx decides the model and the references
    void aff_merge_motion(Cu &cu, int idx) const
    {
        int8_t refi[5][2]; int16_t cpmv[5][2][3][2]; int cpn[5];
        aff_merge(cu, refi, cpmv, cpn);
        cu.affine = cpn[idx] - 1;
        memset(cu.aff_mv, 0, sizeof(cu.aff_mv));
        for (int l = 0; l < 2; l++) {
            cu.refi[l] = refi[idx][l] >= 0 ? refi[idx][l] : -1;
            cu.mv[l][0] = cu.mv[l][1] = 0;
            if (cu.refi[l] >= 0) for (int v = 0; v < cpn[idx]; v++) { cu.aff_mv[l][v][0] = cpmv[idx][l][v][0]; cu.aff_mv[l][v][1] = cpmv[idx][l][v][1]; }
        }
    }
    // what the history keeps of an affine CU: the model's vector at the CU centre (update_history_buffer_parse_affine, xevdm.c:657-796)
    void aff_centre(const Cu &cu, int l, int16_t out[2]) const
    {
        const int16_t (*m)[2] = cu.aff_mv[l];
        const int dhx = (m[1][0] - m[0][0]) * (1 << (7 - cu.log2w)), dhy = (m[1][1] - m[0][1]) * (1 << (7 - cu.log2w));
        int dvx = -dhy, dvy = dhx;
        if (cu.affine == 2) { dvx = (m[2][0] - m[0][0]) * (1 << (7 - cu.log2h)); dvy = (m[2][1] - m[0][1]) * (1 << (7 - cu.log2h)); }
        const int px = 1 << (cu.log2w - 1), py = 1 << (cu.log2h - 1);
        out[0] = clip16(aff_rnd(m[0][0] * 128 + dhx * px + dvx * py, 7)); out[1] = clip16(aff_rnd(m[0][1] * 128 + dhy * px + dvy * py, 7));
    }
    // the vectors later CUs and pictures see of an affine CU: one per sub-block, the control-point vectors themselves at the corners
    // (xevdm_set_affine_mvf, xevdm_util.c:4095-4195; sub-block size xevdm_derive_affine_subblock_size_bi :1870-1945)
    void aff_store(const Cu &cu)
    {
        const int ws = pic.w_scu, w_cu = (1 << cu.log2w) >> 2, h_cu = (1 << cu.log2h) >> 2, scup = (cu.y >> 2) * ws + (cu.x >> 2), vn = cu.affine + 1;
        AffModel m[2];
        const bool use[2] = { cu.refi[0] >= 0, cu.refi[1] >= 0 };
        for (int l = 0; l < 2; l++) if (use[l]) m[l] = aff_model(&cu.aff_mv[l][0][0], cu.log2w, cu.log2h, vn);
        int sub_w, sub_h; bool mem_band;
        aff_subblock(m, use, cu.log2w, cu.log2h, sub_w, sub_h, mem_band);
        const int sw = sub_w >> 2, shh = sub_h >> 2;
        for (int l = 0; l < 2; l++) {
            if (!use[l]) continue;
            const int16_t (*mv)[2] = cu.aff_mv[l];
            for (int h = 0; h < h_cu; h += shh) for (int w = 0; w < w_cu; w += sw) {
                int vx, vy;
                if (w == 0 && h == 0) { vx = mv[0][0]; vy = mv[0][1]; }
                else if (w + sw == w_cu && h == 0) { vx = mv[1][0]; vy = mv[1][1]; }
                else if (w == 0 && h + shh == h_cu && vn == 3) { vx = mv[2][0]; vy = mv[2][1]; }
                else {
                    const int px = (w << 2) + (sub_w >> 1), py = (h << 2) + (sub_h >> 1);
                    vx = aff_clip18(aff_round(mv[0][0] * 128 + m[l].dh[0] * px + m[l].dv[0] * py, 5)) >> 2;
                    vy = aff_clip18(aff_round(mv[0][1] * 128 + m[l].dh[1] * px + m[l].dv[1] * py, 5)) >> 2;
                }
                for (int y = h; y < h + shh; y++) for (int x = w; x < w + sw; x++) {
                    const size_t k = (size_t)scup + (size_t)y * ws + x;
                    pic.mv[k * 4 + l * 2] = (int16_t)vx; pic.mv[k * 4 + l * 2 + 1] = (int16_t)vy;
                    if (!pic.mv_ref.empty()) { pic.mv_ref[k * 4 + l * 2] = (int16_t)vx; pic.mv_ref[k * 4 + l * 2 + 1] = (int16_t)vy; }      // map_mv and map_unrefined_mv agree outside refined CUs
                }
            }
        }
        const uint8_t tag = (uint8_t)(cu.affine | (cu.log2w << 2) | (cu.log2h << 5));
        for (int y = 0; y < h_cu; y++) for (int x = 0; x < w_cu; x++) { pic.aff[(size_t)scup + (size_t)y * ws + x] = tag; pic.aff_tl[(size_t)scup + (size_t)y * ws + x] = (uint32_t)scup; }
    }
    // SCU maps after a CU (xevd_set_dec_info, xevd_util.c:1574-1660; cod_eco xevd.c:797-803)
    std::vector<int16_t> dmvr_scratch;
    bool missing_ref = false;                            // host-side DMVR needed the samples of a reference picture the caller has not registered
    LumaBoard *board = nullptr;                          // xhost_parser_set_ref_luma_wait: where a plane that is not in the DPB entry yet is waited for
    long posted_serial[2] = { 0, 0 };                    // the last two planes taken from the board (a CU has two references; serials start at 1)
    const int16_t *posted_plane[2] = { nullptr, nullptr };
    int posted_stride[2] = { 0, 0 }, posted_next = 0;
    void commit(const Cu &cu)
    {
        if (cu.tree == 2) return;                        // a chroma-only CU leaves every map as its luma CUs wrote it (xevdm_set_dec_info, xevdm_util.c:4241)
        const int xs = cu.x >> 2, ys = cu.y >> 2, w = (1 << cu.log2w) >> 2, h = (1 << cu.log2h) >> 2;
        // Host-side DMVR (Sps::host_dmvr): the refinement search of a merge-mode bi-predicted CU runs HERE, because the refined vectors are state of this
        // picture's parse - the refined map below (MMVD merge lists, temporal candidates of later pictures) and the history buffer
        int16_t refined[64][2][2];
        bool is_refined = false;
        if (sps.host_dmvr() && cu.dmvr && (cu.mode == MODE_INTER || cu.mode == MODE_SKIP) && !cu.affine && cu.refi[0] >= 0 && cu.refi[1] >= 0 &&
            cu.refi[0] < (int)refp[0].size() && cu.refi[1] < (int)refp[1].size()) {
            const RefPic *r0 = refp[0][(size_t)cu.refi[0]], *r1 = refp[1][(size_t)cu.refi[1]];
            if (dmvr_search_applies(poc, r0->poc, r1->poc, 1 << cu.log2w, 1 << cu.log2h)) {
                const int16_t *plane[2] = { r0->luma, r1->luma };
                int stride[2] = { r0->luma_stride, r1->luma_stride };
                const RefPic *rr[2] = { r0, r1 };
                for (int l = 0; l < 2; l++) if (!plane[l] && board) {      // decoded, but its plane may still be on its way from the device (LumaBoard)
                    int k = 0;
                    while (k < 2 && posted_serial[k] != rr[l]->serial) k++;
                    if (k == 2) {
                        k = posted_next; posted_next ^= 1;
                        posted_serial[k] = board->wait(rr[l]->serial, &posted_plane[k], &posted_stride[k]) ? rr[l]->serial : 0;
                    }
                    if (posted_serial[k]) { plane[l] = posted_plane[k]; stride[l] = posted_stride[k]; }
                }
                if (!plane[0] || !plane[1]) missing_ref = true;
                else {
                    const DmvrRefPlane rp[2] = { { plane[0], stride[0], r0->poc }, { plane[1], stride[1], r1->poc } };
                    dmvr_search_cu(pic.w_scu << 2, pic.h_scu << 2, sps.bd_l, cu.x, cu.y, 1 << cu.log2w, 1 << cu.log2h, cu.mv, rp, refined, dmvr_scratch);
                    is_refined = true;
                }
            }
        }
        if (sps.tool_hmvp && (cu.mode == MODE_INTER || cu.mode == MODE_SKIP)) {
            if (is_refined) {                            // core->mv = map_mv[first SCU] before the history update (xevdm_util.c:4384-4387, xevdm.c:1335-1342)
                Cu first = cu;
                memcpy(first.mv, refined[0], sizeof(first.mv));
                history_push(first);
            } else history_push(cu);
        }
        if (!pic.mv_ref.empty()) {
            const int sbw = std::min(w, 4), sbh = std::min(h, 4), per_row = w / sbw;      // 16x16 sub-blocks in SCUs
            for (int r = 0; r < h; r++) for (int c = 0; c < w; c++) {
                const size_t k = (size_t)(ys + r) * pic.w_scu + xs + c;
                const int16_t (*v)[2] = is_refined ? refined[(r / sbh) * per_row + c / sbw] : cu.mv;
                for (int l = 0; l < 2; l++) { pic.mv_ref[k * 4 + l * 2] = v[l][0]; pic.mv_ref[k * 4 + l * 2 + 1] = v[l][1]; }
            }
        }
        {   // the CU's rows of the SCU maps, row by row as fills (a 64x64 CU is 256 SCUs x 6 maps: element-wise stores were a tenth of the parse time)
            const int8_t rf[2] = { (int8_t)cu.refi[0], (int8_t)cu.refi[1] };
            const int16_t mvq[4] = { cu.mv[0][0], cu.mv[0][1], cu.mv[1][0], cu.mv[1][1] };
            const uint8_t is_intra = cu.mode == MODE_INTRA, is_ibc = cu.mode == MODE_IBC, is_skip = cu.mode == MODE_SKIP, size = (uint8_t)(cu.log2w | (cu.log2h << 4));
            for (int r = 0; r < h; r++) {
                const size_t k = (size_t)(ys + r) * pic.w_scu + xs;
                memset(&pic.cod[k], 1, (size_t)w); memset(&pic.intra[k], is_intra, (size_t)w); memset(&pic.ibc[k], is_ibc, (size_t)w); memset(&pic.ipm[k], (int8_t)cu.ipm, (size_t)w);
                if (!pic.skip.empty()) memset(&pic.skip[k], is_skip, (size_t)w);
                if (!pic.cu_size.empty()) memset(&pic.cu_size[k], size, (size_t)w);
                int8_t *rp = &pic.refi[k * 2];
                int16_t *mp = &pic.mv[k * 4];
                for (int c = 0; c < w; c++) { memcpy(rp + 2 * c, rf, 2); memcpy(mp + 4 * c, mvq, 8); }
            }
        }
        if (cu.affine && cu.mode != MODE_INTRA && cu.mode != MODE_IBC) aff_store(cu);
    }

    // ---- coefficient block, run-length coding in zig-zag order (xevd_eco_run_length_cc, xevd_eco.c:343-395) ----
    template <class C> void code_coefs(C &c, int16_t *coef, int log2w, int log2h, int chroma, bool enc)
    {
        const std::vector<uint16_t> &sc = scan[log2w - 1][log2h - 1];
        const int n = 1 << (log2w + log2h);
        int pos = 0, prev_level = 6;
        for (;;) {
            int run = 0, level = 1, sign = 0, last = 1;
            // context pair of run and level: with sps->tool_cm_init by the level before (xevdm_eco.c:319)
            const int t0 = sps.tool_cm_init ? (std::min(prev_level - 1, 5) << 1) + (chroma ? 12 : 0) : (chroma ? 2 : 0);
            if (enc) {
                while (pos + run < n && coef[sc[pos + run]] == 0) run++;
                const int v = coef[sc[pos + run]];
                level = v < 0 ? -v : v; sign = v < 0;
                for (int q = pos + run + 1; q < n; q++) if (coef[sc[q]]) { last = 0; break; }
            }
            run = sym_unary(c, run, models.run + t0, 2);
            pos += run;
            if (pos >= n) return;                                   // malformed input; the caller checks the reader's overrun flag
            level = sym_unary(c, level - 1, models.level + t0, 2) + 1;
            prev_level = level;
            sign = c.ep(sign);
            if (!enc) coef[sc[pos]] = (int16_t)(sign ? -level : level);
            if (pos >= n - 1) break;
            pos++;
            last = c.bin(last, models.last[chroma ? 1 : 0]);
            if (last) break;
        }
    }


    // ---- coefficient block with sps->tool_adcc (xevdm_eco_adcc, src_main/xevdm_eco.c:482-689): position of the last coefficient in scan order, then per
    //      group of 16 scan positions (last group first): significance flags, greater-than-1 flags of the first 8 coefficients, one greater-than-2 flag,
    //      Golomb-Rice remainders, signs.  Contexts from the five already-coded neighbours to the right and below; the block itself is the working state ----
    static int adcc_nb(const int16_t *coef, int blkpos, int width, int height, int log2w, int what, int base = 0)
    {
        const int16_t *p = coef + blkpos;
        const int py = blkpos >> log2w, px = blkpos - (py << log2w);
        int n = 0;
        auto take = [&](int v) { const int a = v < 0 ? -v : v; n += what == 0 ? v != 0 : what == 1 ? a > 1 : what == 2 ? a > 2 : a; };
        if (px < width - 1) { take(p[1]); if (px < width - 2) take(p[2]); if (py < height - 1) take(p[width + 1]); }
        if (py < height - 1) { take(p[width]); if (py < height - 2) take(p[2 * width]); }
        (void)base;
        return n;
    }
    template <class C> void code_adcc(C &c, int16_t *coef, int log2w, int log2h, int chroma, bool enc)
    {
        static const int group_idx[64] = { 0, 1, 2, 3, 4, 4, 5, 5, 6, 6, 6, 6, 7, 7, 7, 7, 8, 8, 8, 8, 8, 8, 8, 8, 9, 9, 9, 9, 9, 9, 9, 9, 10, 10, 10, 10, 10, 10, 10, 10, 10, 10, 10, 10, 10,
                                           10, 10, 10, 11, 11, 11, 11, 11, 11, 11, 11, 11, 11, 11, 11, 11, 11, 11, 11 };      // g_group_idx, xevdm_tbl.c:390
        static const int min_in_group[14] = { 0, 1, 2, 3, 4, 6, 8, 12, 16, 24, 32, 48, 64, 96 };
        static const int rice_range[10] = { 6, 5, 6, 3, 3, 3, 3, 3, 3, 3 };
        static const int rice_para[32] = { 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 1, 1, 1, 2, 2, 2, 2, 2, 2, 2, 2, 2, 2, 2, 2, 2, 2, 3, 3, 3, 3 };
        const int width = 1 << log2w, height = 1 << log2h, n = width * height;
        const std::vector<uint16_t> &sc = scan[log2w - 1][log2h - 1], &inv = scan_inv[log2w - 1][log2h - 1];
        std::vector<int16_t> want;
        int last_x = 0, last_y = 0;
        if (enc) {
            want.assign(coef, coef + n);
            int last = 0;
            for (int i = 0; i < n; i++) if (want[sc[(size_t)i]]) last = i;
            last_x = sc[(size_t)last] & (width - 1); last_y = sc[(size_t)last] >> log2w;
            memset(coef, 0, sizeof(int16_t) * (size_t)n);
        }
        {   // last_sig_coeff_{x,y}_prefix / _suffix (xevdm_parse_positionLastXY :395-457; context sets by block size, xevd_get_ctx_last_pos_xy_para)
            auto para = [&](int size, int log2s, int &off, int &shift) {
                const int cv = std::max(log2s - 2, 0);
                if (chroma) { off = 0; shift = cv - (size >= 16 ? log2s - 4 : 0); }      // cv - log2(size >> 4), the reference's log2 table maps 0 to 0
                else { off = cv * 3 + ((cv + 1) >> 2); shift = (cv + 3) >> 2; if (cv >= 4) { off += ((size >> 6) << 1) + (size >> 7); shift = 2; } }
            };
            int off[2], shift[2], pos[2] = { last_x, last_y };
            para(width, log2w, off[0], shift[0]); para(height, log2h, off[1], shift[1]);
            Model *cm[2] = { models.last_x + (chroma ? 18 : 0), models.last_y + (chroma ? 18 : 0) };
            int grp[2];
            for (int d = 0; d < 2; d++) {
                const int gmax = group_idx[(d ? height : width) - 1], g = enc ? group_idx[pos[d]] : 0;
                int k = 0;
                for (; k < gmax; k++) if (!c.bin(enc ? k < g : 0, cm[d][off[d] + (k >> shift[d])])) break;
                grp[d] = k;
            }
            for (int d = 0; d < 2; d++) {
                if (grp[d] > 3) { const int cnt = (grp[d] - 2) >> 1; pos[d] = min_in_group[grp[d]] + sym_bits_ep(c, enc ? pos[d] - min_in_group[grp[d]] : 0, cnt); }
                else pos[d] = grp[d];
            }
            last_x = std::min(pos[0], width - 1); last_y = std::min(pos[1], height - 1);
        }
        const int num_coeff = inv[(size_t)(last_x + last_y * width)] + 1, scan_pos_last = num_coeff - 1;
        const int log2_min = std::min(log2w, log2h);
        Model *cm_sig = chroma ? models.sig_coeff + 39 : models.sig_coeff + (log2_min <= 2 ? 0 : 13 << std::min(1, log2_min - 3));
        Model *cm_gt = chroma ? models.gt_ab + 13 : models.gt_ab;
        int ipos = scan_pos_last, pos_last = -1, ctx_gta = 0, ctx_gtb = 0;
        for (int sub = scan_pos_last >> 4; sub >= 0; sub--) {
            int num_nz = 0, pos[16], abs_coef[16];
            for (; ipos >= (sub << 4); ipos--) {
                const int blk = sc[(size_t)ipos];
                int sig = 1;
                if (ipos != scan_pos_last) {
                    const int py = blk >> log2w, px = blk - (py << log2w), diag = px + py;
                    int idx = std::min(adcc_nb(coef, blk, width, height, log2w, 0), 4) + 1;
                    if (diag < 2) idx = std::min(idx, 2);
                    const int ofs = chroma ? (diag < 2 ? 0 : 2) : (diag < 2 ? 0 : diag < 5 ? 2 : 7);
                    sig = c.bin(enc ? want[(size_t)blk] != 0 : 0, cm_sig[ofs + idx]);
                }
                coef[blk] = (int16_t)sig;
                if (sig) { pos[num_nz++] = blk; if (pos_last < 0) pos_last = blk; }
            }
            if (!num_nz) continue;
            auto gt_ctx = [&](int blk, int what) {
                const int py = blk >> log2w, px = blk - (py << log2w), diag = px + py;
                int v = std::min(adcc_nb(coef, blk, width, height, log2w, what), 3) + 1;
                if (!chroma) v += diag < 3 ? 0 : diag < 10 ? 4 : 8;
                return v;
            };
            auto true_abs = [&](int blk) { const int v = want[(size_t)blk]; return v < 0 ? -v : v; };
            bool escape = false;
            int first_c2 = -1;
            for (int i = 0; i < num_nz; i++) abs_coef[i] = 1;
            for (int i = 0; i < std::min(num_nz, 8); i++) {
                if (pos[i] != pos_last) ctx_gta = gt_ctx(pos[i], 1);
                const int f = c.bin(enc ? true_abs(pos[i]) > 1 : 0, cm_gt[ctx_gta]);
                coef[pos[i]] = (int16_t)(coef[pos[i]] + f);
                abs_coef[i] = f + 1;
                if (f) { if (first_c2 < 0) first_c2 = i; else escape = true; }
            }
            if (first_c2 >= 0) {
                if (pos[first_c2] != pos_last) ctx_gtb = gt_ctx(pos[first_c2], 2);
                const int f = c.bin(enc ? true_abs(pos[first_c2]) > 2 : 0, cm_gt[ctx_gtb]);
                coef[pos[first_c2]] = (int16_t)(coef[pos[first_c2]] + f);
                abs_coef[first_c2] = f + 2;
                if (f) escape = true;
            }
            escape = escape || num_nz > 8;
            int first_coeff2 = 1;
            if (escape)
                for (int i = 0; i < num_nz; i++) {
                    const int base = i < 8 ? 2 + first_coeff2 : 1;
                    if (abs_coef[i] >= base) {
                        const int r = rice_para[std::max(std::min(adcc_nb(coef, pos[i], width, height, log2w, 3) - 5 * base, 31), 0)];
                        // xevdm_parse_coef_remain_exgolomb (:458-481): unary prefix, then r bins - or, past the prefix limit of r, an escape with growing suffix
                        int sym = enc ? true_abs(pos[i]) - base : 0, prefix = 0;
                        if (enc) {
                            if (sym < (rice_range[r] << r)) prefix = sym >> r;
                            else { int k = 0; while ((((1 << (k + 1)) + rice_range[r] - 1) << r) <= sym) k++; prefix = rice_range[r] + k; }
                        }
                        int k = 0;
                        while (c.ep(enc ? k < prefix : 0)) { if (++k > 32) break; }
                        prefix = k;
                        if (prefix < rice_range[r]) sym = (prefix << r) + sym_bits_ep(c, sym & ((1 << r) - 1), r);
                        else {
                            const int e = prefix - rice_range[r], b0 = ((1 << e) + rice_range[r] - 1) << r;
                            sym = b0 + sym_bits_ep(c, sym - b0, std::min(e + r, 30));
                        }
                        abs_coef[i] = sym + base;
                        coef[pos[i]] = (int16_t)std::min(abs_coef[i], 32767);
                    }
                    if (abs_coef[i] >= 2) first_coeff2 = 0;
                }
            uint32_t signs = 0;
            if (enc) for (int i = 0; i < num_nz; i++) signs = (signs << 1) | (want[(size_t)pos[i]] < 0);
            signs = (uint32_t)sym_bits_ep(c, (int)signs, num_nz);
            for (int i = 0; i < num_nz; i++) {
                const int neg = (signs >> (num_nz - 1 - i)) & 1, a = std::min(abs_coef[i], 32767);
                coef[pos[i]] = (int16_t)(neg ? -a : a);
            }
        }
    }

    // ---- one CU: syntax (xevd_eco_cu, xevd_eco.c:1048-1176; cbf :260-341; coefficients/QP :593-767) + derivations ----
    // enc: `cu` and `coef` carry the wanted values (mv of an INTER CU is met through mvd, a SKIP CU takes its predictor's motion);
    // dec: they are filled.  coef[c]: w*h (w/2*h/2) values of component c, zero-initialised by the caller when decoding.
    // ---- sps_btt_flag: which splits a node may take (xevdm_check_split_mode, src_main/xevdm_util.c:1575-1680).  Index = split mode: 0 none, 1 binary with a
    //      vertical cut, 2 binary horizontal, 3 ternary vertical (1/4, 1/2, 1/4), 4 ternary horizontal; there is no quad split with BTT.  Shapes are limited
    //      through the SPS table of long sides per aspect ratio; a node crossing the picture border takes a binary split towards it ----
    enum { NO_SPLIT = 0, BI_VER = 1, BI_HOR = 2, TRI_VER = 3, TRI_HOR = 4, QUAD = 5 };
    bool ratio_ok(int long_side, int ratio) const { return ratio <= 2 && long_side <= sps.split_tbl[ratio][1] && long_side >= sps.split_tbl[ratio][0]; }
    bool tri_ok(int long_side) const { return long_side <= sps.split_tbl[3][1] && long_side >= sps.split_tbl[3][0]; }
    static bool small_child_is_4x4(int split, int w, int h) { if (split == BI_HOR) h >>= 1; else if (split == BI_VER) w >>= 1; else if (split == TRI_HOR) h >>= 2; else w >>= 2; return w == 4 && h == 4; }
    static bool chroma_split_ok(int split, int w, int h) { if (split == BI_HOR) h >>= 1; else if (split == BI_VER) w >>= 1; else if (split == TRI_HOR) h >>= 2; else w >>= 2; return w * h >= 64; }
    void split_allowed(int allow[6], int lw, int lh, int x, int y, bool only_inter) const
    {
        const int W = sps.width, H = sps.height, w = 1 << lw, h = 1 << lh;
        const bool boundary = !(x + w <= W && y + h <= H), boundary_r = boundary && x + w > W && !(y + h > H);
        const bool from_boundary_b = y >= H - H % 32 && !(x >= W - W % 32);      // cu_max = half a CTU
        for (int i = 0; i < 6; i++) allow[i] = 0;
        allow[NO_SPLIT] = 1;                     // (the reference leaves this entry unset and never reads it for a node inside the picture)
        const bool tv = tri_ok(lw) && (lw > lh || (lw == lh && ratio_ok(lw, 2))), th = tri_ok(lh) && (lh > lw || (lw == lh && ratio_ok(lh, 2)));
        if (lw == lh) { allow[BI_HOR] = allow[BI_VER] = ratio_ok(lw, 1); }
        else if (lw > lh) {
            allow[BI_HOR] = ratio_ok(lw, lw - lh + 1);
            const int sw = lw - 1, ratio = sw > lh ? sw - lh : lh - sw;
            allow[BI_VER] = ratio_ok(std::max(sw, lh), ratio) || (from_boundary_b && (ratio == 3 || ratio == 4));
        } else {
            const int sh_ = lh - 1, ratio = lw > sh_ ? lw - sh_ : sh_ - lw;
            allow[BI_HOR] = ratio_ok(std::max(lw, sh_), ratio);
            allow[BI_VER] = ratio_ok(lh, lh - lw + 1);
        }
        allow[TRI_VER] = tv; allow[TRI_HOR] = th;
        if (boundary) {
            allow[NO_SPLIT] = allow[TRI_VER] = allow[TRI_HOR] = 0;
            if (boundary_r) allow[BI_HOR] = !allow[BI_VER]; else allow[BI_VER] = !allow[BI_HOR];
        }
        if (only_inter) for (int m = BI_VER; m <= TRI_HOR; m++) allow[m] = allow[m] && !small_child_is_4x4(m, w, h);
    }
    // split syntax of a node inside the picture (xevdm_eco_split_mode, src_main/xevdm_eco.c:1173-1296): btt_split_flag, direction, type - each only when
    // both alternatives are allowed; with tool_cm_init the flag's context counts the neighbours (above / left) that are narrower / lower than the node
    template <class C> int code_split(C &c, int want, int x, int y, int lw, int lh, bool only_inter)
    {
        if (lw < 3 && lh < 3) return NO_SPLIT;
        int allow[6];
        split_allowed(allow, lw, lh, x, y, only_inter);
        if (!(allow[BI_VER] || allow[BI_HOR] || allow[TRI_VER] || allow[TRI_HOR])) return NO_SPLIT;
        int ctx = 0;
        if (sps.tool_cm_init) {
            static const uint8_t shape_ctx[6][6] = { { 255, 4, 4, 14, 15, 15 }, { 4, 4, 3, 3, 2, 2 }, { 4, 3, 3, 2, 2, 1 }, { 14, 3, 2, 2, 1, 1 }, { 15, 2, 2, 1, 1, 0 }, { 15, 2, 1, 1, 0, 0 } };      // xevd_tbl_split_flag_ctx
            const int ws = pic.w_scu, xs = x >> 2, ys = y >> 2, scup = ys * ws + xs;
            int smaller = 0;
            if (ys > 0 && pic.same_tile(scup, scup - ws)) smaller += (1 << (pic.cu_size[(size_t)scup - ws] & 15)) < (1 << lw);                                  // above: parsed whenever it is in the tile
            if (xs > 0 && pic.same_tile(scup, scup - 1) && pic.cod[(size_t)scup - 1]) smaller += (1 << (pic.cu_size[(size_t)scup - 1] >> 4)) < (1 << lh);
            const int scuw = (1 << lw) >> 2;                                                                                                                  // the right-hand neighbour: only SUCO lets it be parsed first
            if (sps.suco && xs + scuw < ws && pic.same_tile(scup, scup + scuw) && pic.cod[(size_t)scup + scuw]) smaller += (1 << (pic.cu_size[(size_t)scup + scuw] >> 4)) < (1 << lh);
            ctx = std::min(smaller, 2) + 3 * shape_ctx[lw - 2][lh - 2];
            if (ctx > 14) ctx = 14;
        }
        if (!c.bin(want != NO_SPLIT, models.btt_split_flag[ctx])) return NO_SPLIT;
        const bool ver_ok = allow[BI_VER] || allow[TRI_VER], hor_ok = allow[BI_HOR] || allow[TRI_HOR];
        int dir = want == BI_VER || want == TRI_VER;
        if (ver_ok && hor_ok) dir = c.bin(dir, models.btt_split_dir[sps.tool_cm_init ? lw - lh + 2 : 0]); else dir = ver_ok;
        int tri = want == TRI_VER || want == TRI_HOR;
        if ((dir && allow[BI_VER] && allow[TRI_VER]) || (!dir && allow[BI_HOR] && allow[TRI_HOR])) tri = c.bin(tri, models.btt_split_type[0]);
        else tri = (dir && allow[TRI_VER]) || (!dir && allow[TRI_HOR]);
        return tri ? (dir ? TRI_VER : TRI_HOR) : (dir ? BI_VER : BI_HOR);
    }
    // suco_flag of a split node (xevdm_eco_suco_flag, src_main/xevdm_eco.c:1303-1331; xevdm_check_suco_cond, xevdm_util.c:1702-1727): sent by a node inside the picture whose
    // split has a vertical cut (quad, or binary / ternary vertical of a node wider than high) and whose sides lie in the SPS range; every other node INHERITS its
    // parent's flag - and a vertical split below it then runs right to left as well (xevdm.c:1805: only the direction of the split is asked there)
    template <class C> int code_suco(C &c, int want, int split, int lw, int lh, bool boundary, int parent)
    {
        if (!sps.suco) return parent;
        const int mx = std::min(6 - sps.suco_raw[0], 6), mn = std::max(mx - sps.suco_raw[1], std::max(4, sps.log2_min_cb));
        if (std::min(lw, lh) < mn || std::max(lw, lh) > mx || boundary) return parent;
        if (split == NO_SPLIT || split == BI_HOR || split == TRI_HOR || (split != QUAD && lw <= lh)) return parent;
        int ctx = 0;
        if (sps.tool_cm_init) { ctx = std::max(lw, lh) - 2; ctx = lw == lh ? ctx * 2 : ctx * 2 + 1; }
        return c.bin(want != 0, models.suco_flag[ctx]);
    }
    // the order a split node's n parts are coded in (xevdm_split_get_suco_order, xevdm_util.c:3482-3512): right to left with the flag - the quad split row by row
    static void part_order(int split, int suco, int n, int order[4])
    {
        const bool rev = suco && split != BI_HOR && split != TRI_HOR;
        for (int i = 0; i < n; i++) order[i] = !rev ? i : split == QUAD ? (i ^ 1) : n - 1 - i;
    }
    // children of a split node: position and size
    static int split_parts(int split, int x, int y, int lw, int lh, int px[3], int py[3], int plw[3], int plh[3])
    {
        const int n = split == BI_VER || split == BI_HOR ? 2 : 3, ver = split == BI_VER || split == TRI_VER;
        int off = 0;
        for (int i = 0; i < n; i++) {
            const int shrink = n == 2 ? 1 : (i == 1 ? 1 : 2);
            plw[i] = ver ? lw - shrink : lw; plh[i] = ver ? lh : lh - shrink;
            px[i] = ver ? x + off : x; py[i] = ver ? y : y + off;
            off += 1 << (ver ? plw[i] : plh[i]);
        }
        return n;
    }
    // the mode constraint a split node hands to its children with sps_btt_flag and tool_admvp (xevd_entropy_decode_tree, src_main/xevdm.c:1775-1802):
    // 0 none, 1 inter only (signalled), -1 intra only = a local dual tree (luma CUs, then one chroma CU), which the batch format cannot express
    template <class C> int code_mode_cons(C &c, int split, int lw, int lh, bool only_inter, int want_only_inter)
    {
        if (!(sps.btt && sps.tool_admvp) || only_inter) return only_inter ? 1 : 0;
        const int w = 1 << lw, h = 1 << lh;
        if (chroma_split_ok(split, w, h)) return 0;
        if (sh.type == XHOST_SLICE_I || small_child_is_4x4(split, w, h)) return -1;
        return c.bin(!want_only_inter, models.mode_cons[0]) ? -1 : 1;      // the flag says "intra only"; its context counts nothing (always 0)
    }

    // sps->tool_cm_init: the contexts of skip_flag / pred_mode_flag / ibc_flag / affine_flag count the neighbours that have the property - above the top-left
    // SCU, left of the bottom-left one and right of the bottom-right one (parsed before the CU only in a part that SUCO runs right to left) - in the same tile and already parsed
    // (xevdm_get_ctx_some_flags, src_main/xevdm_util.c:1729-1853)
    enum { CTX_SKIP, CTX_PRED, CTX_IBC, CTX_AFF };
    int nb_ctx(const Cu &cu, int what) const
    {
        if (!sps.tool_cm_init) return 0;
        const int ws = pic.w_scu, xs = cu.x >> 2, ys = cu.y >> 2, scuw = (1 << cu.log2w) >> 2, scuh = (1 << cu.log2h) >> 2, scup = ys * ws + xs;
        const int nb[3] = { scup - ws, scup - 1 + (scuh - 1) * ws, scup + scuw + (scuh - 1) * ws };
        const bool in[3] = { ys > 0, xs > 0, sps.suco && xs + scuw < ws };
        int n = 0;
        for (int k = 0; k < 3; k++) {
            if (!in[k] || !pic.same_tile(scup, nb[k]) || !pic.cod[(size_t)nb[k]]) continue;
            n += what == CTX_SKIP ? pic.skip[(size_t)nb[k]] : what == CTX_PRED ? pic.intra[(size_t)nb[k]] : what == CTX_IBC ? pic.ibc[(size_t)nb[k]] : (!pic.aff.empty() && pic.aff[(size_t)nb[k]] != 0);
        }
        return std::min(n, what == CTX_PRED ? 2 : 1);
    }
    template <class C> int code_refi(C &c, int want, int nref)      // xevd_eco_refi, xevd_eco.c:409-436
    {
        if (nref <= 1) return 0;
        const int r = std::min(std::max(want, 0), nref - 1);
        int v = 0;
        if (c.bin(r > 0, models.refi[0])) {
            v = 1;
            if (nref > 2 && c.bin(r > 1, models.refi[1])) {
                v = 2;
                for (; v < nref - 1; v++) if (!c.ep(r > v)) break;
            }
        }
        return v;
    }
    template <class C> void code_mvd(C &c, int16_t mvd[2])          // xevd_eco_get_mvd, xevd_eco.c:491-536
    {
        for (int d = 0; d < 2; d++) {
            const int v = mvd[d], a = sym_abs_mvd(c, v < 0 ? -v : v, models.mvd[0]);
            int sg = v < 0;
            if (a) sg = c.ep(sg);
            mvd[d] = (int16_t)(sg ? -a : a);
        }
    }
    // affine_flag + affine merge index of a skip / merge-mode CU of at least 8x8 (xevdm_eco.c:1528-1537, 1622-1632); true: the CU is affine
    template <class C> bool code_affine_merge(C &c, Cu &cu)
    {
        int aff = 0;
        if (sps.tool_affine && cu.log2w >= 3 && cu.log2h >= 3) aff = c.bin(cu.affine != 0, models.affine_flag[nb_ctx(cu, CTX_AFF)]);
        if (!aff) { cu.affine = 0; return false; }
        cu.aff_idx[0] = sym_trunc_unary(c, cu.aff_idx[0], models.affine_mrg, 5, 5);
        aff_merge_motion(cu, cu.aff_idx[0]);
        cu.dmvr = 0;
        return true;
    }
    // the luma intra mode a chroma-only CU refers to: the one stored at the centre of its block (xevd_get_luma_cup, xevd_util.c:1481); Baseline modes: that CU is intra
    // by constraint; EIPD: DC when it is not (an IBC CU)
    int luma_mode_of(const Cu &cu) const
    {
        const size_t k = (size_t)((cu.y >> 2) + ((1 << cu.log2h) >> 3)) * pic.w_scu + (cu.x >> 2) + ((1 << cu.log2w) >> 3);
        return pic.intra[k] ? pic.ipm[k] : 0;
    }
    template <class C> void code_cu(C &c, Cu &cu, int16_t *coef[3], bool enc)
    {
        // mode constraint eOnlyIntra: I slices, and with tool_admvp every 4x4 CU (xevdm.c:1838-1843) - no skip flag, no pred_mode_flag
        // ... and the CUs of a local dual tree (TREE_L / TREE_C)
        const bool inter_slice = sh.type != XHOST_SLICE_I && !(sps.tool_admvp && cu.log2w == 2 && cu.log2h == 2) && cu.tree == 0;
        const int keep_only_inter = cu.only_inter;
        int skip = 0;
        if (inter_slice) skip = c.bin(cu.mode == MODE_SKIP, models.skip[nb_ctx(cu, CTX_SKIP)]);
        if (!enc) { cu.mode = skip ? MODE_SKIP : MODE_INTRA; cu.refi[0] = cu.refi[1] = -1; memset(cu.mv, 0, sizeof(cu.mv)); memset(cu.mvd, 0, sizeof(cu.mvd));
                    cu.mvp_idx[0] = cu.mvp_idx[1] = 0; cu.cbf[0] = cu.cbf[1] = cu.cbf[2] = 0; cu.ipm = cu.ipm_c = 0; cu.ats = cu.ats_inter = 0; cu.dmvr = 0; cu.mmvd = cu.mmvd_idx = 0; cu.affine = 0; memset(cu.aff_mv, 0, sizeof(cu.aff_mv)); memset(cu.aff_mvd, 0, sizeof(cu.aff_mvd)); cu.aff_idx[0] = cu.aff_idx[1] = 0; }
        int16_t cand[4][2];
        const int n_lists = sh.type == XHOST_SLICE_B ? 2 : 1;
        if (skip && sps.tool_admvp) {
            // Main: one merge index, truncated unary over five contexts (xevdm_eco_merge_idx, xevdm_eco.c:731-744; call site :1550-1551)
            if (sps.tool_mmvd) cu.mmvd = c.bin(cu.mmvd, models.mmvd_flag[0]);
            if (cu.mmvd) { cu.affine = 0; code_mmvd_idx(c, cu); mmvd_motion(cu); }
            else if (code_affine_merge(c, cu)) { }
            else {
                cu.mvp_idx[0] = cu.mvp_idx[1] = sym_trunc_unary(c, cu.mvp_idx[0], models.merge_idx, 5, 6);
                merge_motion(cu, cu.mvp_idx[0]);
                cu.dmvr = sps.tool_dmvr;
            }
            cu.cbf[0] = cu.cbf[1] = cu.cbf[2] = 0;
            cu.qp = qp_prev;
            return;
        }
        if (skip) {
            // the motion of candidate mvp_idx of every list, reference 0 (xevd_get_skip_motion, xevd.c:502-531; syntax xevd_eco.c:1079-1085)
            for (int l = 0; l < n_lists; l++) cu.mvp_idx[l] = sym_trunc_unary(c, cu.mvp_idx[l], models.mvp_idx, 3, 4);
            cu.refi[1] = -1; cu.mv[1][0] = cu.mv[1][1] = 0;
            for (int l = 0; l < n_lists; l++) {
                mvp_candidates(cu, l, cand);
                cu.refi[l] = 0; cu.mv[l][0] = cand[cu.mvp_idx[l]][0]; cu.mv[l][1] = cand[cu.mvp_idx[l]][1];
            }
            cu.cbf[0] = cu.cbf[1] = cu.cbf[2] = 0;
            cu.qp = qp_prev;                                         // xevd_eco.c:1091-1115 (cu_qp_delta on: previous QP; off: slice QP = the same)
            return;
        }
        int intra = 1;
        if (inter_slice && keep_only_inter) intra = 0;                 // eOnlyInter: no pred_mode_flag (xevdm_eco_pred_mode, xevdm_eco.c:1401-1438)
        else if (inter_slice) intra = c.bin(cu.mode == MODE_INTRA, models.pred_mode[nb_ctx(cu, CTX_PRED)]);
        // xevdm_eco_pred_mode (xevdm_eco.c:1401-1438): with sps->ibc_flag every CU up to the IBC size limit that is not already known to be
        // intra-predicted carries ibc_flag - in I slices all of them (mode constraint eOnlyIntra: no pred_mode_flag); context 0 without cm_init
        int ibc = 0;
        if (sps.ibc && cu.log2w <= sps.ibc_log_max && cu.log2h <= sps.ibc_log_max && !(inter_slice && intra) && !keep_only_inter && cu.tree != 2)
            ibc = c.bin(cu.mode == MODE_IBC, models.ibc_flag[nb_ctx(cu, CTX_IBC)]);
        if (!enc) { cu.mode = ibc ? MODE_IBC : intra ? MODE_INTRA : MODE_INTER; cu.direct = 0; }
        if (ibc) {
            // the block vector itself is sent as a motion vector difference (xevdm_eco.c:1789-1800); no references, no predictor
            intra = 0;
            cu.refi[0] = cu.refi[1] = -1; cu.mv[1][0] = cu.mv[1][1] = 0; cu.direct = 0;
            for (int d = 0; d < 2; d++) {
                const int v = cu.mv[0][d], a = sym_abs_mvd(c, v < 0 ? -v : v, models.mvd[0]);
                int sg = v < 0;
                if (a) sg = c.ep(sg);
                cu.mv[0][d] = (int16_t)(sg ? -a : a);
            }
        } else if (!intra && sps.tool_admvp) {
            // xevdm_eco.c:1595-1726 with the sub-tools off: merge_mode_flag (the CU takes a merge candidate: pred_mode MODE_DIR), else
            // inter_pred_idc (no bi-prediction for CUs of 4x4 / 4x8 / 8x4), bi_idx of a bi-predicted CU (normal / list 0 / list 1 without a
            // coded difference; the latter two also derive their reference indices), per list reference index and vector difference.
            // The predictor is the resolution-indexed one (index 0 without AMVR): mv = predictor + mvd (xevd_get_inter_motion, xevdm.c:885-932)
            int mvr = 0;                                             // xevdm_eco_mvr_idx (xevdm_eco.c:814-817): quarter, half, 1, 2, 4 samples
            if (sps.tool_amvr) {
                if (enc) mvr = (cu.direct || cu.affine) ? 0 : ((cu.x >> 3) * 5 + (cu.y >> 3) * 3) % 11 % 5 * (((cu.x ^ cu.y) >> 2) & 1);      // about every second coded vector on a coarser grid
                mvr = sym_trunc_unary(c, mvr, models.mvr_idx, 4, 5);
            }
            cu.direct = mvr == 0 ? c.bin(cu.direct, models.merge_mode[0]) : 0;
            if (cu.direct) {
                if (sps.tool_mmvd) cu.mmvd = c.bin(cu.mmvd, models.mmvd_flag[0]);
                if (cu.mmvd) { cu.affine = 0; code_mmvd_idx(c, cu); mmvd_motion(cu); }
                else if (code_affine_merge(c, cu)) { }
                else {
                    cu.mvp_idx[0] = cu.mvp_idx[1] = sym_trunc_unary(c, cu.mvp_idx[0], models.merge_idx, 5, 6);
                    merge_motion(cu, cu.mvp_idx[0]);
                    cu.dmvr = sps.tool_dmvr;
                }
            } else {
                int dir = 0;
                if (n_lists == 2) {
                    if (enc) dir = (cu.refi[0] >= 0 && cu.refi[1] >= 0 && bi_applicable(cu)) ? 2 : ((cu.refi[1] >= 0 && cu.refi[0] < 0) ? 1 : 0);
                    int not_bi = 1;
                    if (bi_applicable(cu)) not_bi = c.bin(dir != 2, models.inter_dir[0]);
                    if (!not_bi) dir = 2;
                    else dir = c.bin(dir == 1, models.inter_dir[1]) ? 1 : 0;
                }
                // affine inter CU (xevdm_eco.c:1649-1682): 16x16 and larger, quarter-sample vectors only; affine_mode picks 2 or 3 control points, then
                // per list the reference, one of two predictors, and the control-point differences (all zero with affine_mvd_flag)
                int aff = 0;
                if (sps.tool_affine && cu.log2w >= 4 && cu.log2h >= 4 && mvr == 0) aff = c.bin(cu.affine != 0, models.affine_flag[nb_ctx(cu, CTX_AFF)]);
                if (!aff) cu.affine = 0;
                else {
                    cu.affine = 1 + c.bin(cu.affine == 2, models.affine_mode[0]);
                    const int vn = cu.affine + 1;
                    for (int l = 0; l < 2; l++) {
                        cu.mv[l][0] = cu.mv[l][1] = 0;
                        if (!(((dir + 1) >> l) & 1)) { cu.refi[l] = -1; memset(cu.aff_mv[l], 0, sizeof(cu.aff_mv[l])); continue; }
                        cu.refi[l] = code_refi(c, cu.refi[l], (int)refp[l].size());
                        cu.aff_idx[l] = c.bin(cu.aff_idx[l] != 0, models.affine_mvp_idx[0]);
                        int16_t mvp[2][3][2];
                        aff_amvp(cu, l, cu.refi[l], vn, mvp);
                        int16_t (*pp)[2] = mvp[cu.aff_idx[l]];
                        if (enc) {      // differences that reproduce the requested control points: the first one also moves the other predictors
                            for (int d = 0; d < 2; d++) {
                                cu.aff_mvd[l][0][d] = (int16_t)(cu.aff_mv[l][0][d] - pp[0][d]);
                                for (int v = 1; v < vn; v++) cu.aff_mvd[l][v][d] = (int16_t)(cu.aff_mv[l][v][d] - (int16_t)(pp[v][d] + cu.aff_mvd[l][0][d]));
                            }
                        }
                        int zero = 1;
                        for (int v = 0; v < vn; v++) zero &= cu.aff_mvd[l][v][0] == 0 && cu.aff_mvd[l][v][1] == 0;
                        zero = c.bin(zero, models.affine_mvd_flag[l]);
                        for (int v = 0; v < vn; v++) {
                            if (zero) cu.aff_mvd[l][v][0] = cu.aff_mvd[l][v][1] = 0; else code_mvd(c, cu.aff_mvd[l][v]);
                            for (int d = 0; d < 2; d++) cu.aff_mv[l][v][d] = (int16_t)(pp[v][d] + cu.aff_mvd[l][v][d]);
                            if (v == 0) for (int d = 0; d < 2; d++) { pp[1][d] = (int16_t)(pp[1][d] + cu.aff_mvd[l][0][d]); pp[2][d] = (int16_t)(pp[2][d] + cu.aff_mvd[l][0][d]); }
                        }
                    }
                }
                int bi_idx = 0;                                      // BI_NON 0, BI_NORMAL 1, BI_FL0 2, BI_FL1 3
                if (dir == 2 && !aff) {
                    if (enc) bi_idx = 1 + ((cu.x >> 2) + (cu.y >> 2) * 3) % 7 % 3;      // a spread of the three kinds over the picture
                    const int v = bi_idx - 1;
                    if (c.bin(v == 0, models.bi_idx[0])) bi_idx = 1;
                    else bi_idx = c.bin(v == 1, models.bi_idx[1]) ? 2 : 3;
                }
                for (int l = 0; l < 2 && !aff; l++) {
                    if (!(((dir + 1) >> l) & 1)) { cu.refi[l] = -1; cu.mv[l][0] = cu.mv[l][1] = 0; continue; }
                    const int nref = (int)refp[l].size();
                    if (bi_idx != 2 && bi_idx != 3) {
                        if (nref > 1) {                              // xevd_eco_refi, xevd_eco.c:409-436
                            const int r = std::min(std::max(cu.refi[l], 0), nref - 1);
                            int v = 0;
                            if (c.bin(r > 0, models.refi[0])) {
                                v = 1;
                                if (nref > 2 && c.bin(r > 1, models.refi[1])) {
                                    v = 2;
                                    for (; v < nref - 1; v++) if (!c.ep(r > v)) break;
                                }
                            }
                            cu.refi[l] = v;
                        } else cu.refi[l] = 0;
                    } else cu.refi[l] = first_refi(cu, l, mvr);
                    int16_t mvp[2];
                    mvr_predictor(cu, l, cu.refi[l], mvr, mvp);
                    if (bi_idx == 2 + l) cu.mvd[l][0] = cu.mvd[l][1] = 0;
                    else {
                        if (enc) { cu.mvd[l][0] = (int16_t)((cu.mv[l][0] - mvp[0]) >> mvr); cu.mvd[l][1] = (int16_t)((cu.mv[l][1] - mvp[1]) >> mvr); }
                        for (int d = 0; d < 2; d++) {                // xevd_eco_get_mvd, xevd_eco.c:491-536
                            const int v = cu.mvd[l][d], a = sym_abs_mvd(c, v < 0 ? -v : v, models.mvd[0]);
                            int sg = v < 0;
                            if (a) sg = c.ep(sg);
                            cu.mvd[l][d] = (int16_t)(sg ? -a : a);
                        }
                    }
                    cu.mv[l][0] = (int16_t)(mvp[0] + cu.mvd[l][0] * (1 << mvr)); cu.mv[l][1] = (int16_t)(mvp[1] + cu.mvd[l][1] * (1 << mvr));
                }
            }
        } else if (!intra) {
            // xevd_eco.c:1120-1148: B: direct_mode_flag, else inter_pred_idc; per list in use: ref index, predictor index, mvd; mv = mvp + mvd (xevd.c:533-556)
            int dir = 0;                                             // PRED_L0 0, PRED_L1 1, PRED_BI 2
            if (n_lists == 2) cu.direct = c.bin(cu.direct, models.direct[0]);
            if (cu.direct) direct_motion(cu);
            else {
                if (n_lists == 2) {
                    if (enc) dir = (cu.refi[0] >= 0 && cu.refi[1] >= 0) ? 2 : (cu.refi[1] >= 0 ? 1 : 0);
                    if (!c.bin(dir != 2, models.inter_dir[0])) dir = 2;
                    else dir = c.bin(dir == 1, models.inter_dir[1]) ? 1 : 0;
                }
                for (int l = 0; l < 2; l++) {
                    if (!(((dir + 1) >> l) & 1)) { cu.refi[l] = -1; cu.mv[l][0] = cu.mv[l][1] = 0; continue; }
                    const int nref = (int)refp[l].size();
                    if (nref > 1) {                                  // xevd_eco_refi, xevd_eco.c:409-436
                        const int r = cu.refi[l];
                        int v = 0;
                        if (c.bin(r > 0, models.refi[0])) {
                            v = 1;
                            if (nref > 2 && c.bin(r > 1, models.refi[1])) {
                                v = 2;
                                for (; v < nref - 1; v++) if (!c.ep(r > v)) break;
                            }
                        }
                        cu.refi[l] = v;
                    } else cu.refi[l] = 0;
                    mvp_candidates(cu, l, cand);
                    if (enc) {                                       // cheapest predictor
                        int best = 0, cost = 1 << 30;
                        for (int k = 0; k < 4; k++) { const int d = abs(cu.mv[l][0] - cand[k][0]) + abs(cu.mv[l][1] - cand[k][1]); if (d < cost) { cost = d; best = k; } }
                        cu.mvp_idx[l] = best;
                        cu.mvd[l][0] = (int16_t)(cu.mv[l][0] - cand[best][0]); cu.mvd[l][1] = (int16_t)(cu.mv[l][1] - cand[best][1]);
                    }
                    cu.mvp_idx[l] = sym_trunc_unary(c, cu.mvp_idx[l], models.mvp_idx, 3, 4);
                    for (int d = 0; d < 2; d++) {                    // xevd_eco_get_mvd, xevd_eco.c:491-536
                        const int v = cu.mvd[l][d], a = sym_abs_mvd(c, v < 0 ? -v : v, models.mvd[0]);
                        int sg = v < 0;
                        if (a) sg = c.ep(sg);
                        cu.mvd[l][d] = (int16_t)(sg ? -a : a);
                    }
                    cu.mv[l][0] = (int16_t)(cand[cu.mvp_idx[l]][0] + cu.mvd[l][0]); cu.mv[l][1] = (int16_t)(cand[cu.mvp_idx[l]][1] + cu.mvd[l][1]);
                }
            }
        } else if (cu.tree == 2 && !sps.tool_eipd) {
            cu.ipm_c = cu.ipm = luma_mode_of(cu);                  // no syntax: the chroma block takes the luma mode at its centre (xevdm_eco.c:1763-1781)
        } else if (sps.tool_eipd) {
            // xevd_eco_intra_dir (xevd_eco.c:842-879): one of the 2 most probable modes, one of the 8 extended ones (bypass), or the index
            // among the remaining 23 in truncated binary (4 or 5 bypass bins)
            int mpm[2], ext[8], pims[33];
            if (cu.tree == 2) cu.ipm = luma_mode_of(cu);           // chroma-only CU: DM refers to the luma mode at the block's centre, DC when that CU is not intra (xevdm_eco.c:1738-1752)
            else {
            eipd_mpm(cu, mpm, ext, pims);
            int pos = 0;
            while (enc && pos < 33 && pims[pos] != cu.ipm) pos++;
            const int in_mpm = enc ? (cu.ipm == mpm[0] || cu.ipm == mpm[1]) : 0;
            if (c.bin(in_mpm, models.ipm_mpm_flag[0])) {
                cu.ipm = mpm[c.bin(cu.ipm == mpm[1], models.ipm_mpm_idx[0])];
            } else {
                int ei = 0;
                while (enc && ei < 8 && ext[ei] != cu.ipm) ei++;
                if (c.ep(enc ? ei < 8 : 0)) {
                    cu.ipm = ext[sym_bits_ep(c, ei & 7, 3)];
                } else {
                    const int rem = pos - 10;                        // 23 symbols: 9 with 4 bins, 14 with 5
                    int v = sym_bits_ep(c, rem < 9 ? rem : (rem + 9) >> 1, 4);
                    if (v >= 9) v = ((v << 1) | c.ep((rem + 9) & 1)) - 9;
                    cu.ipm = pims[10 + std::min(v, 22)];
                }
            }
            }
            if (cu.tree == 1) cu.ipm_c = 0;
            else {
            // xevd_eco_intra_dir_c (:881-910): DM, or one of the other modes - the one DM stands for (a luma DC / BI / HOR / VER) is skipped
            const int lc = cu.ipm == 12 ? 4 : cu.ipm == 24 ? 3 : cu.ipm == 0 ? 2 : cu.ipm == 2 ? 1 : 0;
            if (enc && lc && cu.ipm_c == lc) cu.ipm_c = 0;
            if (c.bin(cu.ipm_c == 0, models.ipm_chroma[0])) cu.ipm_c = 0;
            else {
                int v = sym_unary_ep(c, (lc && cu.ipm_c > lc ? cu.ipm_c - 1 : cu.ipm_c) - 1, 4) + 1;
                if (lc && v >= lc) v++;
                cu.ipm_c = std::min(v, 4);
            }
            }
        } else {
            const uint8_t *mpm = mpm_list(cu);                       // xevd_eco_intra_dir_b, xevd_eco.c:826-846: the code number is sent
            const int code = sym_unary(c, mpm[cu.ipm], models.intra_dir, 2);
            if (!enc) for (int i = 0; i < 5; i++) if (mpm[i] == code) cu.ipm = i;
        }
        // coded block flags (eco_cbf, xevd_eco.c:260-341), CU <= 64: no sub-blocks
        bool all_zero = false;
        if (!intra && cu.tree == 0) {
            // a merge-mode CU (MODE_DIR with tool_admvp) has coefficients by definition - without them it would be a skip CU: no cbf_all (xevdm_eco.c:831)
            const bool merge_cu = sps.tool_admvp && cu.direct && cu.mode == MODE_INTER;
            const int any = merge_cu ? 1 : c.bin((cu.cbf[0] | cu.cbf[1] | cu.cbf[2]) != 0, models.cbf_all[0]);
            if (!any) { cu.cbf[0] = cu.cbf[1] = cu.cbf[2] = 0; all_zero = true; }
            else {
                cu.cbf[1] = c.bin(cu.cbf[1], models.cbf_cb[0]);
                cu.cbf[2] = c.bin(cu.cbf[2], models.cbf_cr[0]);
                if (cu.cbf[1] + cu.cbf[2] == 0) cu.cbf[0] = 1;
                else cu.cbf[0] = c.bin(cu.cbf[0], models.cbf_luma[0]);
            }
        } else {                                                     // intra CUs, and every CU of a dual tree (an IBC luma CU too): one flag per component it has (xevdm_eco_cbf, xevdm_eco.c:266-296)
            cu.cbf[1] = cu.tree == 1 ? 0 : c.bin(cu.cbf[1], models.cbf_cb[0]);
            cu.cbf[2] = cu.tree == 1 ? 0 : c.bin(cu.cbf[2], models.cbf_cr[0]);
            cu.cbf[0] = cu.tree == 2 ? 0 : c.bin(cu.cbf[0], models.cbf_luma[0]);
        }
        // QP (xevd_eco.c:640-668, xevd_eco_dqp :460-479): a delta only when the CU has coefficients
        // sps->dquant_flag (Main; xevdm_eco.c:882-897): one delta per quantisation group - a CU of at least the group size sends it when it has
        // coefficients (code 1), the first CU that gets this far inside a group of smaller CUs sends it in any case (code 2), the others take the predictor
        const bool any_cbf = cu.cbf[0] || cu.cbf[1] || cu.cbf[2];
        const bool qp_here = sps.profile_main && sps.dquant ? ((cu.qp_code == 1 && !qp_coded && any_cbf) || (cu.qp_code == 2 && !qp_coded)) : any_cbf;
        if (!all_zero && pps.cu_qp_delta && qp_here) {
            int dqp = 0;
            if (enc) { dqp = cu.qp - qp_prev; while (dqp > 25) dqp -= 52; while (dqp < -26) dqp += 52; }
            const int a = sym_unary(c, dqp < 0 ? -dqp : dqp, models.dqp, 1);
            int s = dqp < 0;
            if (a) s = c.ep(s);
            dqp = s ? -a : a;
            cu.qp = (qp_prev + dqp + 52) % 52;
            qp_prev = cu.qp;
            qp_coded = 1;
        } else cu.qp = qp_prev;
        if (all_zero) return;
        // Main, tool_ats (xevdm_eco_coef, xevdm_eco.c:902-934): transform selection of intra luma blocks up to 32x32, sub-block
        // transform of inter CUs (the coefficient blocks of all components then have the TU size, xevdm_eco_xcoef :697-703)
        int tlw = cu.log2w, tlh = cu.log2h;
        if (sps.tool_ats) {
            if (intra && cu.cbf[0] && cu.log2w <= 5 && cu.log2h <= 5) {
                int on = c.ep(cu.ats & 1), mh = 0, mv = 0;
                if (on) { mh = c.bin((cu.ats >> 2) & 1, models.ats_mode[0]); mv = c.bin((cu.ats >> 1) & 1, models.ats_mode[0]); }
                cu.ats = on | (mv << 1) | (mh << 2);
            } else cu.ats = 0;
            const int w = 1 << cu.log2w, h = 1 << cu.log2h;
            const int avail = (intra || ibc || w > 64 || h > 64) ? 0 : ((w >= 8) | ((h >= 8) << 1) | ((w >= 16) << 2) | ((h >= 16) << 3));      // xevdm_util.c:3565-3583
            cu.ats_inter = avail ? code_ats_inter(c, cu.ats_inter, avail, cu.log2w, cu.log2h) : 0;
            const int idx = cu.ats_inter & 15;
            if (idx == 1 || idx == 3) tlw -= idx == 3 ? 2 : 1;
            if (idx == 2 || idx == 4) tlh -= idx == 4 ? 2 : 1;
        }
        for (int k = 0; k < 3; k++)
            if (cu.cbf[k]) {
                if (!enc) memset(coef[k], 0, sizeof(int16_t) << (tlw + tlh - (k ? 2 : 0)));      // the decoder writes the coded levels into a cleared block (the arena itself is not: Batch)
                if (sps.tool_adcc) code_adcc(c, coef[k], tlw - (k ? 1 : 0), tlh - (k ? 1 : 0), k != 0, enc); else code_coefs(c, coef[k], tlw - (k ? 1 : 0), tlh - (k ? 1 : 0), k != 0, enc);
            }
    }

    // ats_inter_info syntax (xevdm_eco_ats_inter_info, xevdm_eco.c:128-190; with cm_init the flag's context goes by the CU's area, the direction's by its shape)
    template <class C> int code_ats_inter(C &c, int info, int avail, int log2w, int log2h)
    {
        const int mv_ = avail & 1, mh_ = (avail >> 1) & 1, vq = (avail >> 2) & 1, hq = (avail >> 3) & 1;
        const int ctx_flag = sps.tool_cm_init ? (log2w + log2h >= 8 ? 0 : 1) : 0, ctx_hor = sps.tool_cm_init ? (log2w == log2h ? 0 : log2w < log2h ? 1 : 2) : 0;
        if (!c.bin(info != 0, models.ats_inter_flag[ctx_flag])) return 0;
        const int idx = info & 15;
        int quad = idx >= 3, hor = idx == 2 || idx == 4, pos = (info >> 4) & 1;
        if ((vq || hq) && (mv_ || mh_)) quad = c.bin(quad, models.ats_inter_quad[0]); else quad = 0;
        if ((quad && vq && hq) || (!quad && mv_ && mh_)) hor = c.bin(hor, models.ats_inter_hor[ctx_hor]);
        else hor = (quad && hq) || (!quad && mh_);
        pos = c.bin(pos, models.ats_inter_pos[0]);
        return ((quad ? 2 : 0) + (hor ? 1 : 0) + 1) | (pos << 4);
    }
};

// ------------------------------------------------------------------------------------------------ NAL plumbing

// sps->dquant_flag: where a quantisation group starts in the split tree (xevd_entropy_decode_tree, src_main/xevdm.c:1739-1759) - at a leaf of at least
// pps.cu_qp_delta_area samples (code 1: the delta goes with the CU's coefficients), or at the split node of exactly that size (code 2: the first CU below it
// that reaches its QP syntax sends the delta).  -> the code for the node's children / the leaf
static inline int qp_group(const Stream &st, TileCoder &tc, int split, int lw, int lh, int qp_code)
{
    if (!(st.pps.cu_qp_delta && st.sps.dquant && st.sps.profile_main)) return qp_code;
    if (!split && lw + lh >= st.pps.qp_delta_area && qp_code != 2) { tc.qp_coded = 0; return (lw == 7 || lh == 7) ? 2 : 1; }
    const bool tri = split == TileCoder::TRI_VER || split == TileCoder::TRI_HOR;
    if ((tri && lw + lh == st.pps.qp_delta_area + 1) || (lw + lh == st.pps.qp_delta_area && qp_code != 2)) { tc.qp_coded = 0; return 2; }
    return qp_code;
}

}   // namespace
