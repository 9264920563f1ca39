#!/usr/bin/env python3
"""Generates the golden vectors under tests/golden/ FROM THE REFERENCE ITSELF.

Run in the development container only (needs oracle/_ref, i.e. /root/reference compiled by oracle/Makefile.ref):
    python tests/golden/make_golden.py
Every expected output below is produced by the reference's own code - the exported per-block functions of
libxevd_ref.so (plain-C tables, which are normative) and, at picture level, the reference's
xevd_sub_block_itdq / xevd_mc / xevdm_mc / xevd_recon / xevd_set_dec_info / xevd_deblock_cu_* / picbuf expand
driven by oracle/ref_harness.c.  The files hold inputs AND expected outputs (data only), so the tests that
consume them (tests/test_golden.py on CPU, tests/test_gpu_parity.py on the MI355X) never need the reference.
"""
import ctypes as C
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import cases  # noqa: E402
import oracle_lib as ol  # noqa: E402


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def golden_mc(lib, bds=(8, 10), fname="blocks_mc.npz", seed=2024):
    names = {(0, 0): "00", (1, 0): "n0", (0, 1): "0n", (1, 1): "nn"}
    rng = np.random.default_rng(seed)
    out = {}
    recs, preds = [], []
    for bd in bds:
        plane = rng.integers(0, 1 << bd, (96, 160)).astype(np.int16)
        out[f"plane_bd{bd}"] = plane
        for admvp in (0, 1):
            lp = C.c_void_p.in_dll(lib, "tbl_mc_l_coeff")
            cp = C.c_void_p.in_dll(lib, "tbl_mc_c_coeff")
            lp.value = C.addressof((C.c_int16 * 128).in_dll(lib, "tbl_mc_l_coeff_main" if admvp else "xevd_tbl_mc_l_coeff"))
            cp.value = C.addressof((C.c_int16 * 128).in_dll(lib, "tbl_mc_c_coeff_main" if admvp else "xevd_tbl_mc_c_coeff"))
            for luma in (1, 0):
                for has_dx in (0, 1):
                    for has_dy in (0, 1):
                        for trial in range(6):
                            lo = 2 if luma else 1
                            w = 1 << int(rng.integers(lo, 7))
                            h = 1 << int(rng.integers(lo, 7))
                            prec = 4 if luma else 5
                            step = 1 if (admvp and trial % 2) else 4
                            fx = int(rng.integers(0, (1 << prec) // step)) * step
                            fy = int(rng.integers(0, (1 << prec) // step)) * step
                            ix, iy = int(rng.integers(6, 160 - 12 - w)), int(rng.integers(6, 96 - 12 - h))
                            gx, gy = (ix << prec) + fx, (iy << prec) + fy
                            a = np.zeros((h, w), np.int16)
                            fn = getattr(lib, f"xevd_mc_{'l' if luma else 'c'}_{names[(has_dx, has_dy)]}")
                            fn(_p(plane), gx, gy, plane.shape[1], w, _p(a), w, h, bd)
                            recs.append((bd, admvp, luma, has_dx, has_dy, w, h, gx, gy, sum(p.size for p in preds)))
                            preds.append(a.ravel())
    out["recs"] = np.array(recs, np.int64)
    out["pred"] = np.concatenate(preds)
    np.savez_compressed(os.path.join(HERE, fname), **out)
    print(fname + ":", len(recs), "blocks")


def golden_itdq(lib, bds=(8, 10), fname="blocks_itdq.npz", seed=77):
    rng = np.random.default_rng(seed)
    itxb = (C.c_void_p * 6).in_dll(lib, "xevd_tbl_itxb")
    itx = (C.c_void_p * 6).in_dll(lib, "xevdm_tbl_itx")
    f_itxb = C.CFUNCTYPE(None, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int)
    f_itx = C.CFUNCTYPE(None, C.c_void_p, C.c_void_p, C.c_int, C.c_int)
    lib.xevd_dquant.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int32, C.c_uint8]
    recs, cin, cout = [], [], []
    for iqt in (0, 1):
        for bd in bds:
            for log2w in range(1, 7):
                for log2h in range(1, 7):
                    w, h = 1 << log2w, 1 << log2h
                    for trial in range(3):
                        qp = int(rng.integers(10, 46)) + 6 * (bd - 8)
                        tbl = [40, 45, 51, 57, 64, 72] if iqt else [40, 45, 51, 57, 64, 71]
                        scale = tbl[qp % 6] << (qp // 6)
                        shift = 20 - 14 - (15 - bd - ((log2w + log2h) >> 1)) + (8 if (log2w + log2h) & 1 else 0)
                        offset = 0 if shift == 0 else 1 << (shift - 1)
                        ns = 181 if (log2w + log2h) & 1 else 1
                        cap = max(1, int(2.0 * (1 << bd) / (scale * ns / 2.0 ** shift)))
                        coef = np.zeros((h, w), np.int16)
                        nnz = int(rng.integers(1, max(2, min(w * h // 4, 40))))
                        ys = rng.integers(0, min(h, 32), nnz)          # 64-point dims: first 32 positions only
                        xs = rng.integers(0, min(w, 32), nnz)
                        coef[ys, xs] = rng.integers(-cap, cap + 1, nnz)
                        a = coef.copy()
                        lib.xevd_dquant(_p(a), log2w, log2h, scale, offset, shift)
                        if iqt:
                            t = np.zeros(w * h, np.int16)
                            f_itx(itx[log2h - 1])(_p(a), _p(t), 7, w)
                            f_itx(itx[log2w - 1])(_p(t), _p(a), 12 - (bd - 8), h)
                        else:
                            t = np.zeros(w * h, np.int32)
                            f_itxb(itxb[log2h - 1])(_p(a), _p(t), 0, w, 0)
                            f_itxb(itxb[log2w - 1])(_p(t), _p(a), 7 + 12 - (bd - 8), h, 1)
                        recs.append((iqt, bd, log2w, log2h, qp, sum(c.size for c in cin)))
                        cin.append(coef.ravel())
                        cout.append(a.ravel())
    np.savez_compressed(os.path.join(HERE, fname), recs=np.array(recs, np.int64), coef=np.concatenate(cin),
                        resid=np.concatenate(cout))
    print(fname + ":", len(recs), "blocks")


def golden_pictures(only=None):
    for case in cases.CASES:
        if only and case[0] not in only:
            continue
        cs = cases.build_case(*case)
        final, pre, maps, resid = cases.run_cpu("ref", cs)
        d = {"params": np.array(case[1:6], np.int64), "n_refs": np.array(case[6], np.int64),
             "tools": np.array([cs["addb"], cs["alf"], cs["alpha_off"], cs["beta_off"], cs["no_deblock"], cs["log2_ctu"], cs.get("eipd", 0)], np.int64)}
        if cs["alf_params"] is not None:
            ap = cs["alf_params"]
            d["alf_enable"] = np.array(ap["enable"], np.int64)
            d["alf_luma_coef"], d["alf_chroma_coef"], d["alf_ctb_flag"] = ap["luma_coef"], ap["chroma_coef"], ap["ctb_flag"]
            d["alf_across_tiles"] = np.array(ap["across_tiles"], np.int64)
        seen = {}
        for (i, l), pic in cs["refs"].items():
            if id(pic) in seen:                       # the same picture in another list slot: store an alias
                d[f"refalias_{i}_{l}"] = np.array(seen[id(pic)], np.int64)
                continue
            seen[id(pic)] = (i, l)
            for c in range(3):
                d[f"ref_{i}_{l}_{c}"] = pic.active(c)
            d[f"refpoc_{i}_{l}"] = np.array(pic.poc)
        for k, v in cs["batch"].items():
            if v is not None:
                d["b_" + k] = np.asarray(v)
        for c in range(3):
            d[f"out_{c}"] = final.bufs[c]
            d[f"pre_{c}"] = pre.active(c)
        d["resid"] = resid
        if cs["batch"].get("dmvr") is not None:      # what xevdm_mc leaves for temporal prediction: refined / unrefined vectors of the DMVR candidates
            d["dmvr_mv"] = cases.dmvr_mvs("ref", cs)
        d["map_scu"] = maps.map_scu & 0x7FFFFFFF
        np.savez_compressed(os.path.join(HERE, f"pic_{case[0]}.npz"), **d)
        print(f"pic_{case[0]}.npz")


def golden_streams(only=()):
    import stream_util as su
    for name, (w, h, n, kw) in {"ippp_8b": (208, 120, 5, dict(max_refs=2)), "ippp_10b_offsets": (144, 88, 4, dict(bit_depth=10, qp_offsets=(1, -2))),
                                "idr_period_skip": (72, 136, 6, dict(max_refs=4, skip_frac=0.4, idr_period=4)),
                                "hier_b_gop4": (208, 120, 9, dict(log2_sub_gop=2, max_refs=2)),
                                "hier_b_gop8_10b": (136, 136, 9, dict(log2_sub_gop=3, max_refs=3, bit_depth=10, direct_frac=0.3)),
                                "main_iqt_ats_addb_10b": (144, 136, 9, dict(main=True, iqt=True, ats=True, addb=True, log2_sub_gop=2, max_refs=2, bit_depth=10, addb_offsets=(1, -2))),
                                "main_iqt_addb_8b": (208, 120, 5, dict(main=True, iqt=True, addb=True, max_refs=2)),
                                "main_all_tools_10b": (200, 136, 9, dict(main=True, iqt=True, ats=True, addb=True, alf=True, log2_sub_gop=2, max_refs=2, bit_depth=10)),
                                "main_alf_addb_8b": (264, 136, 6, dict(main=True, alf=True, addb=True)),
                                # every picture followed by a picture-signature SEI (MD5s of the oracle's reconstruction, VERIFIED by the
                                # reference decoder while it produced the pictures below)
                                "cqt_crop_10b": (144, 88, 5, dict(bit_depth=10, chroma_qp_points=(0, [[(30, 0), (10, -4), (15, -6)], [(25, 1), (6, -2), (30, -10)]]), max_refs=2, crop=(2, 4, 0, 6), qp_offsets=(1, -1))),
                                "main_dra_10b": (144, 88, 5, dict(main=True, iqt=True, bit_depth=10, dra="five_ranges_idx40", addb=True, log2_sub_gop=2, max_refs=2)),
                                "main_eipd_i_8b": (136, 120, 2, dict(main=True, eipd=True, idr_period=1, split_prob=0.7)),
                                "main_eipd_all_tools_10b": (200, 136, 9, dict(main=True, iqt=True, ats=True, addb=True, alf=True, eipd=True, inter_frac=0.5, log2_sub_gop=2, max_refs=2, bit_depth=10)),
                                "main_htdf_all_tools_10b": (200, 136, 9, dict(main=True, iqt=True, ats=True, addb=True, alf=True, eipd=True, htdf=True, inter_frac=0.6, log2_sub_gop=2, max_refs=2, bit_depth=10)),
                                "main_htdf_8b": (136, 72, 4, dict(main=True, htdf=True, inter_frac=0.6, max_refs=2)),
                                "signed_hier_b_8b": (136, 120, 5, dict(log2_sub_gop=2, max_refs=2, sign=True)),
                                "signed_main_alf_10b": (136, 72, 4, dict(main=True, iqt=True, addb=True, alf=True, bit_depth=10, sign=True)),
                                # sps->ibc_flag: intra block copy CUs (ibc_flag + block vector syntax) in I, P and B slices next to every other tool
                                "main_admvp_lowdelay_8b": (136, 136, 8, dict(main=True, admvp=True, inter_frac=0.9, max_refs=4)),
                                "main_admvp_all_tools_10b": (200, 136, 9, dict(main=True, admvp=True, iqt=True, ats=True, addb=True, alf=True, eipd=True, htdf=True, ibc_log_max=5, inter_frac=0.7, max_refs=2, log2_sub_gop=2, bit_depth=10)),
                                "main_admvp_amvr_hmvp_10b": (264, 136, 9, dict(main=True, admvp=True, amvr=True, hmvp=True, iqt=True, addb=True, alf=True, inter_frac=0.9, max_refs=3, log2_sub_gop=3, bit_depth=10)),
                                "main_dmvr_b_8b": (200, 136, 9, dict(main=True, admvp=True, dmvr=True, inter_frac=0.9, max_refs=2, log2_sub_gop=2, skip_frac=0.3, direct_frac=0.3)),
                                "main_dmvr_all_tools_10b": (264, 136, 17, dict(main=True, admvp=True, dmvr=True, amvr=True, iqt=True, ats=True, addb=True, alf=True, eipd=True, htdf=True, ibc_log_max=4,
                                                                               inter_frac=0.9, max_refs=3, log2_sub_gop=3, bit_depth=10, skip_frac=0.3, direct_frac=0.3)),
                                # tool_dmvr with tool_hmvp / tool_mmvd: the front end refines vectors itself (reference samples through xhost_parser_set_ref_luma)
                                "main_dmvr_hmvp_mmvd_b_8b": (200, 136, 9, dict(main=True, admvp=True, dmvr=True, hmvp=True, mmvd=True, amvr=True, log2_sub_gop=2, max_refs=4, skip_frac=0.3, direct_frac=0.4)),
                                "main_every_tool_10b": (264, 200, 17, dict(main=True, btt=(2, 0, 0, 0), admvp=True, affine=True, amvr=True, hmvp=True, mmvd=True, dmvr=True, iqt=True, ats=True, addb=True,
                                                                           alf=True, eipd=True, htdf=True, cm_init=True, adcc=True, rpl=True, pocs=True, qp_delta_area=8, max_refs=2, log2_sub_gop=3,
                                                                           split_prob=0.7, bit_depth=10, inter_frac=0.9, skip_frac=0.3, direct_frac=0.3)),
                                "main_every_tool_tiles_8b": (392, 264, 9, dict(main=True, admvp=True, amvr=True, hmvp=True, mmvd=True, dmvr=True, affine=True, iqt=True, addb=True, alf=True, inter_frac=0.95,
                                                                               skip_frac=0.3, direct_frac=0.3, max_refs=2, log2_sub_gop=2, tiles=(2, 2, 0))),
                                "main_mmvd_all_tools_10b": (264, 136, 17, dict(main=True, admvp=True, mmvd=True, amvr=True, hmvp=True, iqt=True, ats=True, addb=True, alf=True, eipd=True, htdf=True,
                                                                               ibc_log_max=5, inter_frac=0.9, skip_frac=0.35, direct_frac=0.3, max_refs=3, log2_sub_gop=3, bit_depth=10)),
                                "main_rpl_pocs_gop8_10b": (264, 136, 17, dict(main=True, rpl=True, pocs=True, admvp=True, amvr=True, hmvp=True, iqt=True, addb=True, alf=True, inter_frac=0.9,
                                                                              skip_frac=0.3, direct_frac=0.3, max_refs=3, log2_sub_gop=3, bit_depth=10)),
                                "main_rpl_in_sps_lowdelay_8b": (200, 136, 12, dict(main=True, rpl=True, pocs=True, admvp=True, max_refs=4, rpl_in_sps=True, idr_period=7)),
                                "main_cm_init_all_tools_10b": (264, 136, 17, dict(main=True, cm_init=True, rpl=True, pocs=True, admvp=True, affine=True, amvr=True, hmvp=True, mmvd=True, iqt=True, ats=True,
                                                                                  addb=True, alf=True, eipd=True, htdf=True, ibc_log_max=5, inter_frac=0.9, skip_frac=0.3, direct_frac=0.3, max_refs=3,
                                                                                  log2_sub_gop=3, bit_depth=10, qp_delta_area=8)),
                                "main_adcc_all_tools_10b": (264, 136, 17, dict(main=True, adcc=True, rpl=True, pocs=True, admvp=True, affine=True, amvr=True, hmvp=True, mmvd=True, iqt=True, ats=True,
                                                                               addb=True, alf=True, eipd=True, htdf=True, ibc_log_max=5, inter_frac=0.9, skip_frac=0.3, direct_frac=0.3, max_refs=3,
                                                                               log2_sub_gop=3, bit_depth=10, qp_delta_area=8)),
                                "main_adcc_low_qp_8b": (200, 136, 5, dict(main=True, adcc=True, iqt=True, max_refs=2, max_level=3000, qp_range=(0, 8))),
                                "main_btt_p_8b": (136, 72, 4, dict(main=True, btt=(2, 0, 0, 0), max_refs=2, split_prob=0.7)),
                                "main_btt_all_tools_10b": (264, 200, 9, dict(main=True, btt=(2, 0, 0, 0), admvp=True, affine=True, amvr=True, hmvp=True, mmvd=True, iqt=True, ats=True, addb=True, alf=True,
                                                                             eipd=True, htdf=True, cm_init=True, adcc=True, rpl=True, pocs=True, qp_delta_area=8, max_refs=2, log2_sub_gop=2, split_prob=0.7,
                                                                             bit_depth=10, inter_frac=0.9, skip_frac=0.3, direct_frac=0.3)),
                                "main_btt_tiles_8b": (392, 264, 5, dict(main=True, btt=(3, 1, 1, 1), iqt=True, addb=True, alf=True, tiles=(2, 2, 0), max_refs=2, split_prob=0.8)),
                                # local dual trees of BTT + ADMVP streams: luma-only intra / IBC CUs followed by their node's chroma-only CU
                                "main_dual_tree_i_8b": (200, 136, 2, dict(main=True, btt=(2, 0, 0, 0), admvp=True, dual_tree=True, split_prob=0.8, idr_period=1)),
                                "main_dual_tree_all_tools_10b": (264, 200, 9, dict(main=True, btt=(2, 0, 0, 0), admvp=True, dual_tree=True, affine=True, amvr=True, hmvp=True, mmvd=True, iqt=True,
                                                                                   ats=True, addb=True, alf=True, eipd=True, htdf=True, ibc_log_max=4, cm_init=True, adcc=True, qp_delta_area=8,
                                                                                   max_refs=2, log2_sub_gop=2, split_prob=0.7, bit_depth=10, inter_frac=0.7, skip_frac=0.3, direct_frac=0.3)),
                                "main_dual_tree_tiles_8b": (392, 264, 4, dict(main=True, btt=(2, 0, 0, 0), admvp=True, dual_tree=True, split_prob=0.7, inter_frac=0.6, tiles=(2, 2, 0), addb=True)),
                                "main_dquant_area8_10b": (264, 200, 9, dict(main=True, admvp=True, iqt=True, ats=True, addb=True, alf=True, eipd=True, inter_frac=0.8, split_prob=0.65, max_refs=2,
                                                                            log2_sub_gop=2, bit_depth=10, qp_delta_area=8)),
                                # sps->tool_affine: affine merge and affine inter CUs from the bitstream
                                "main_affine_p_8b": (392, 264, 6, dict(main=True, admvp=True, affine=True, inter_frac=0.95, split_prob=0.35, skip_frac=0.3, direct_frac=0.3, max_refs=2)),
                                "main_affine_all_tools_10b": (328, 264, 17, dict(main=True, admvp=True, affine=True, amvr=True, hmvp=True, mmvd=True, iqt=True, ats=True, addb=True, alf=True,
                                                                                eipd=True, htdf=True, inter_frac=0.9, split_prob=0.4, skip_frac=0.3, direct_frac=0.3, max_refs=3, log2_sub_gop=3,
                                                                                bit_depth=10)),
                                "main_affine_dmvr_tiles_8b": (392, 264, 9, dict(main=True, admvp=True, affine=True, affine_frac=0.6, dmvr=True, addb=True, alf=True, inter_frac=0.95, split_prob=0.35,
                                                                                skip_frac=0.3, direct_frac=0.3, max_refs=2, log2_sub_gop=2, tiles=(2, 2, 0))),
                                # several tiles per picture: without / with filtering across the tile borders, uniform and explicit grids
                                "main_tiles_2x2_dbk_8b": (256, 192, 4, dict(main=True, tiles=(2, 2, 0), max_refs=2)),
                                "main_tiles_3x2_all_tools_10b": (320, 200, 9, dict(main=True, iqt=True, ats=True, addb=True, alf=True, alf_fixed=True, eipd=True, htdf=True, admvp=True, amvr=True, hmvp=True,
                                                                                   mmvd=True, log2_sub_gop=2, max_refs=2, bit_depth=10, tiles=(3, 2, 0))),
                                "main_tiles_across_dmvr_8b": (392, 264, 9, dict(main=True, iqt=True, addb=True, alf=True, eipd=True, htdf=True, admvp=True, dmvr=True, log2_sub_gop=2, max_refs=2,
                                                                                tiles=(2, 3, 1))),
                                "main_tiles_explicit_10b": (512, 320, 5, dict(main=True, iqt=True, addb=True, alf=True, eipd=True, admvp=True, bit_depth=10, tiles=(3, 3, 0, (1, 5), (2, 1)))),
                                # several slice NAL units per picture (tile rectangles; own slice QPs; the in-loop filters follow the LAST slice's header)
                                "main_slices_4rows_all_tools_10b": (256, 256, 9, dict(main=True, pocs=True, rpl=True, iqt=True, ats=True, addb=True, alf=True, eipd=True, admvp=True, amvr=True,
                                                                                      hmvp=True, mmvd=True, max_refs=2, log2_sub_gop=2, bit_depth=10, tiles=(4, 4, 0),
                                                                                      slices=[(0, 3, 28, 1), (4, 7, 33, 0), (8, 11, -1, 1), (12, 15, 35, 1)])),
                                "main_slices_columns_arbitrary_8b": (384, 256, 5, dict(main=True, pocs=True, iqt=True, addb=True, alf=True, admvp=True, affine=True, dmvr=True, max_refs=2,
                                                                                       tiles=(3, 2, 1), slices=[(0, 3, 31), (1, 5, 25)], arbitrary_slices=True)),
                                # sps_suco_flag: split nodes coded right to left - right-hand neighbours in candidate lists, contexts, most probable modes and intra prediction
                                "main_suco_eipd_i_8b": (136, 120, 2, dict(main=True, suco=(0, 2), eipd=True, idr_period=1, split_prob=0.7)),
                                "main_suco_p_8b": (136, 72, 4, dict(main=True, suco=(0, 2), max_refs=2)),
                                "main_suco_quad_all_tools_10b": (264, 136, 9, dict(main=True, suco=(0, 2), admvp=True, affine=True, amvr=True, hmvp=True, mmvd=True, iqt=True, ats=True, addb=True, alf=True,
                                                                                   eipd=True, htdf=True, ibc_log_max=4, cm_init=True, adcc=True, qp_delta_area=8, max_refs=2, log2_sub_gop=2, split_prob=0.6,
                                                                                   bit_depth=10, inter_frac=0.6, skip_frac=0.3, direct_frac=0.3)),
                                "main_suco_btt_all_tools_10b": (264, 200, 9, dict(main=True, suco=(0, 2), btt=(2, 0, 0, 0), admvp=True, dual_tree=True, affine=True, amvr=True, hmvp=True, mmvd=True, dmvr=True,
                                                                                  iqt=True, ats=True, addb=True, alf=True, eipd=True, htdf=True, ibc_log_max=4, cm_init=True, adcc=True, qp_delta_area=8,
                                                                                  max_refs=2, log2_sub_gop=2, split_prob=0.7, bit_depth=10, inter_frac=0.7, skip_frac=0.3, direct_frac=0.3)),
                                "main_suco_tiles_dbk_8b": (392, 264, 5, dict(main=True, suco=(0, 3), eipd=True, htdf=True, tiles=(2, 2, 0), max_refs=2, split_prob=0.7, inter_frac=0.6)),
                                "main_alf_fixed_8b": (264, 136, 8, dict(main=True, alf=True, addb=True, alf_fixed=True)),
                                # 12 bit (bit_depth_luma_minus8 = bit_depth_chroma_minus8 = 4): Baseline hierarchical B; Main with every pixel tool
                                "hier_b_gop4_12b": (208, 120, 9, dict(log2_sub_gop=2, max_refs=2, bit_depth=12)),
                                "main_all_tools_12b": (264, 200, 9, dict(main=True, admvp=True, affine=True, amvr=True, hmvp=True, mmvd=True, dmvr=True, iqt=True, ats=True, addb=True, alf=True,
                                                                         eipd=True, htdf=True, ibc_log_max=4, max_refs=2, log2_sub_gop=2, bit_depth=12, inter_frac=0.8, skip_frac=0.3, direct_frac=0.3)),
                                "main_ibc_i_8b": (136, 72, 3, dict(main=True, eipd=True, ibc_log_max=4, ibc_frac=0.4, idr_period=1)),
                                "main_ibc_all_tools_10b": (200, 136, 9, dict(main=True, iqt=True, ats=True, addb=True, alf=True, eipd=True, htdf=True, ibc_log_max=5, inter_frac=0.5, log2_sub_gop=2, max_refs=2, bit_depth=10))}.items():
        if only and name not in only:
            continue
        data = su.make_stream(w, h, n, seed=len(name) * 13 + n, **kw)
        ref = su.decode_reference(data, w, h, main=bool(kw.get("main")))
        assert len(ref) == n
        d = {"bytes": np.frombuffer(data, np.uint8), "n": np.array(n), "size": np.array([w, h])}
        for k in range(n):
            for c in range(3):
                d[f"p{k}_{c}"] = ref[k][c]
        np.savez_compressed(os.path.join(HERE, f"stream_{name}.npz"), **d)
        print(f"stream_{name}.npz", len(data), "bytes,", n, "pictures")


def golden_dra():
    """DRA tables from the reference's xevd_init_dra + pictures through the reference's DRA and output conversion"""
    d = {}
    rng = np.random.default_rng(99)
    w, h = 136, 72
    planes = [rng.integers(0, 1024, (h >> (i > 0), w >> (i > 0))).astype(np.int16) for i in range(3)]
    planes[0][0, :4] = [0, 1023, 1, 1022]
    planes[1][0, :4] = [0, 1023, 512, 511]
    for c in range(3):
        d[f"in_{c}"] = planes[c]
    for name in sorted(ol.DRA_SETS):
        luts, out = ol.ref_dra(name, 10, planes)
        d[f"{name}_luts"] = np.stack(luts)
        d[f"{name}_out8"] = ol.ref_output_convert(out, 10, 8)
        d[f"{name}_out10"] = ol.ref_output_convert(out, 10, 10)
    np.savez_compressed(os.path.join(HERE, "dra.npz"), **d)
    print("dra.npz")


if __name__ == "__main__":
    assert ol.have_ref(), "oracle/_ref is not built: run `make -C oracle -f Makefile.ref` in the development container"
    lib = ol.ref()
    import sys
    if len(sys.argv) > 1 and sys.argv[1] == "dra":
        golden_dra()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "blocks12":       # the 12-bit block vectors (added in round 6; the 8 / 10-bit files keep their bytes)
        golden_mc(lib, (12,), "blocks_mc_12b.npz", 2025)
        golden_itdq(lib, (12,), "blocks_itdq_12b.npz", 78)
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "streams":
        golden_streams(sys.argv[2:])
    elif len(sys.argv) > 1:               # python make_golden.py <picture case> ... : only (re)generate those
        golden_pictures(set(sys.argv[1:]))
    else:
        golden_mc(lib)
        golden_itdq(lib)
        golden_mc(lib, (12,), "blocks_mc_12b.npz", 2025)
        golden_itdq(lib, (12,), "blocks_itdq_12b.npz", 78)
        golden_pictures()
        golden_streams()
