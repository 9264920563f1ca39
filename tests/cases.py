"""Shared picture-level test cases: build seeded inputs once, run them through the oracle, the real reference
(oracle/_ref harness) or the HIP backend (C ABI), and return comparable outputs."""
import ctypes as C
import zlib

import numpy as np

import oracle_lib as ol
from xevd_amd import abi, synth

# name, w, h, bd, admvp, iqt, n_refs, bi_frac
CASES = [
    ("base_p_8b", 208, 120, 8, 0, 0, (1, 0), 0.0),
    ("base_b_8b", 136, 72, 8, 0, 0, (2, 2), 0.5),
    ("base_p_10b", 144, 88, 10, 0, 0, (2, 0), 0.0),
    ("main_b_10b", 200, 136, 10, 1, 1, (2, 2), 0.5),
    ("main_admvp_only", 128, 64, 8, 1, 0, (1, 1), 0.4),
    ("main_iqt_only", 128, 72, 10, 0, 1, (1, 1), 0.4),
    # Main-profile in-loop filters: (..., tools)
    ("main_addb_10b", 200, 136, 10, 1, 1, (2, 2), 0.5, {"addb": 1}),
    ("main_addb_8b_shared_refs", 136, 136, 8, 1, 1, (3, 3), 0.6, {"addb": 1}),
    ("main_alf_10b", 200, 136, 10, 1, 1, (2, 2), 0.5, {"addb": 1, "alf": 1, "inter_frac": 1.0}),
    ("main_alf_8b_across_tiles", 136, 72, 8, 1, 1, (1, 1), 0.4, {"addb": 1, "alf": 1, "across_tiles": 1, "inter_frac": 1.0}),
    ("main_alf_only_luma", 72, 136, 10, 1, 1, (1, 0), 0.0, {"alf": 1, "alf_enable": (1, 0, 0), "inter_frac": 1.0, "no_deblock": 1}),
    # CTU 128: CUs up to 128x128 (four 64x64 sub-TBs), deblocking of the inner 64-sample boundaries, ALF CTU rules
    ("main_ctu128_10b", 264, 200, 10, 1, 1, (2, 2), 0.5, {"addb": 1, "alf": 1, "inter_frac": 1.0, "log2_ctu": 7, "split_prob": 0.25}),
    ("main_ctu128_8b_noiqt", 256, 128, 8, 1, 0, (1, 1), 0.4, {"addb": 1, "log2_ctu": 7, "split_prob": 0.2}),
    # ATS: DST-VII / DCT-VIII luma transforms of intra CUs (checked through the residual arena)
    ("main_ats_10b", 136, 136, 10, 1, 1, (1, 1), 0.3, {"addb": 1, "inter_frac": 0.5, "ats_frac": 0.7}),
    ("main_ats_8b_noiqt", 128, 72, 8, 1, 0, (1, 0), 0.0, {"inter_frac": 0.4, "ats_frac": 0.8, "split_prob": 0.7}),
    # ATS-inter: half/quarter-size TU of an inter CU, residual placed at one end, luma cbf and bS on the coded part only
    ("main_atsinter_10b", 200, 136, 10, 1, 1, (2, 2), 0.5, {"addb": 1, "inter_frac": 1.0, "ats_inter_frac": 1.0, "split_prob": 0.4, "coded_frac": 0.85}),
    ("main_atsinter_8b_mixed", 136, 136, 8, 1, 1, (1, 1), 0.3, {"addb": 1, "inter_frac": 0.8, "ats_frac": 0.5, "ats_inter_frac": 0.6, "split_prob": 0.35}),
    ("main_atsinter_noaddb", 128, 72, 10, 1, 0, (1, 0), 0.0, {"inter_frac": 1.0, "ats_inter_frac": 0.8}),
    # BTT: binary / ternary splits -> non-square CUs (4x16 ... 128x32), edges off the 8x8 ADDB grid, all tools together
    ("main_btt_10b", 200, 136, 10, 1, 1, (2, 2), 0.5, {"addb": 1, "alf": 1, "btt_frac": 0.7, "ats_inter_frac": 0.6, "inter_frac": 1.0, "coded_frac": 0.8}),
    ("main_btt_ctu128_8b", 264, 200, 8, 1, 1, (1, 1), 0.4, {"addb": 1, "log2_ctu": 7, "btt_frac": 0.7, "split_prob": 0.45, "ats_inter_frac": 0.5}),
    ("main_btt_noaddb_8b", 136, 72, 8, 1, 0, (1, 0), 0.0, {"btt_frac": 0.8, "split_prob": 0.6}),
    # intra prediction (Baseline modes; Main runs the same ones with tool_eipd = 0): all-intra pictures, constrained intra
    ("base_i_8b", 136, 120, 8, 0, 0, (1, 0), 0.0, {"inter_frac": 0.0}),
    ("base_p_constrained_intra_10b", 144, 88, 10, 0, 0, (1, 0), 0.0, {"inter_frac": 0.5, "constrained_intra": 1}),
    ("main_i_btt_10b", 200, 136, 10, 1, 1, (1, 0), 0.0, {"inter_frac": 0.0, "btt_frac": 0.7, "addb": 1, "alf": 1, "ats_frac": 0.5}),
    ("main_b_ctu128_intra_mix_8b", 264, 200, 8, 1, 1, (1, 1), 0.4, {"inter_frac": 0.6, "log2_ctu": 7, "btt_frac": 0.5, "split_prob": 0.4, "addb": 1, "alf": 1, "ats_frac": 0.4, "ats_inter_frac": 0.4}),
    # sps->tool_eipd: 33 luma / 5 chroma intra modes, neighbour padding by repetition (xevdm_get_nbr)
    ("main_eipd_i_10b", 136, 120, 10, 1, 1, (1, 0), 0.0, {"inter_frac": 0.0, "eipd": 1, "addb": 1}),
    ("main_eipd_i_btt_8b", 200, 136, 8, 1, 1, (1, 0), 0.0, {"inter_frac": 0.0, "eipd": 1, "btt_frac": 0.7, "split_prob": 0.6}),
    ("main_eipd_b_ctu128_constrained_10b", 264, 200, 10, 1, 1, (1, 1), 0.4, {"inter_frac": 0.5, "eipd": 1, "log2_ctu": 7, "btt_frac": 0.5, "split_prob": 0.4,
                                                                             "addb": 1, "alf": 1, "constrained_intra": 1, "ats_frac": 0.4}),
    # affine motion (sps->tool_affine): 2 / 3 control points, sub-block translation and per-sample EIF, vectors past the picture edge, sub-block
    # vectors in the map the deblocking filter reads
    ("main_affine_b_10b", 200, 136, 10, 1, 1, (2, 2), 0.5, {"addb": 1, "inter_frac": 1.0, "affine_frac": 0.7, "split_prob": 0.35}),
    ("main_affine_p_8b_atsinter", 136, 136, 8, 1, 1, (2, 0), 0.0, {"addb": 1, "inter_frac": 0.9, "affine_frac": 0.6, "ats_inter_frac": 0.5, "split_prob": 0.3, "btt_frac": 0.5}),
    ("main_affine_b_ctu128_10b", 264, 200, 10, 1, 1, (1, 1), 0.5, {"addb": 1, "alf": 1, "inter_frac": 0.9, "affine_frac": 0.8, "log2_ctu": 7, "split_prob": 0.3, "btt_frac": 0.4}),
    # intra block copy (sps->ibc_flag): copies out of the current picture's reconstructed part, chained with intra CUs; bS of IBC edges
    ("main_ibc_i_10b", 200, 136, 10, 1, 1, (1, 0), 0.0, {"addb": 1, "inter_frac": 0.0, "ibc_frac": 0.5, "split_prob": 0.5}),
    ("main_ibc_b_8b_noaddb", 136, 136, 8, 1, 1, (1, 1), 0.4, {"inter_frac": 0.5, "ibc_frac": 0.4, "btt_frac": 0.5, "ats_frac": 0.4, "ats_inter_frac": 0.4}),
    ("main_ibc_p_ctu128_eipd_10b", 264, 200, 10, 1, 1, (1, 0), 0.0, {"addb": 1, "alf": 1, "inter_frac": 0.4, "ibc_frac": 0.4, "log2_ctu": 7, "split_prob": 0.4, "eipd": 1,
                                                                    "btt_frac": 0.4, "affine_frac": 0.4, "constrained_intra": 1}),
    # HTDF (sps->tool_htdf): every intra CU and every coded inter CU filtered right after its reconstruction, borders from the CUs before it
    ("main_htdf_b_10b", 200, 136, 10, 1, 1, (1, 1), 0.4, {"addb": 1, "inter_frac": 0.7, "htdf_qp": 32, "coded_frac": 0.8}),
    ("main_htdf_i_8b_constrained", 136, 136, 8, 1, 1, (1, 0), 0.0, {"inter_frac": 0.3, "htdf_qp": 24, "constrained_intra": 1, "btt_frac": 0.5, "split_prob": 0.4, "eipd": 1}),
    ("main_htdf_p_ctu128_10b", 264, 200, 10, 1, 1, (2, 0), 0.0, {"addb": 1, "alf": 1, "inter_frac": 0.8, "htdf_qp": 45, "log2_ctu": 7, "split_prob": 0.4, "btt_frac": 0.4,
                                                                "ats_inter_frac": 0.4, "affine_frac": 0.3, "ibc_frac": 0.15, "coded_frac": 0.8}),
    # DMVR (sps->tool_dmvr): merge-mode bi-predicted CUs refined per 16x16 sub-block when their references are POC-symmetric (lists (0,0) and (1,1) of
    # POCS are, the mixed pairs are not: those CUs take the ordinary path with the flag set)
    ("main_dmvr_b_10b", 200, 136, 10, 1, 1, (2, 2), 0.8, {"addb": 1, "inter_frac": 1.0, "dmvr_frac": 0.8, "split_prob": 0.35}),
    ("main_dmvr_b_8b_ctu128_mixed", 264, 200, 8, 1, 1, (2, 2), 0.7, {"addb": 1, "alf": 1, "inter_frac": 0.85, "dmvr_frac": 0.7, "log2_ctu": 7, "split_prob": 0.3, "btt_frac": 0.4,
                                                                     "ats_inter_frac": 0.3, "affine_frac": 0.2}),
    # CTU 128 without ADDB: the Main library's copy of the Baseline filter, CUs above 64 filtered as two halves
    ("main_ctu128_noaddb_8b", 264, 264, 8, 1, 0, (1, 1), 0.3, {"log2_ctu": 7, "btt_frac": 0.6, "ats_inter_frac": 0.5, "split_prob": 0.3}),
    # 12 bit: the reference is generic in bit depth (src_main/xevdm.c:351-352, src_base/xevd_mc.c:253-286 shift from bit_depth) and xgpu_open accepts 8..12 - Base taps with the
    # baseline filter, Main with every pixel tool (the packed-s16 paths of the kernels are for <= 10 bit; above that the scalar instantiations run)
    ("base_p_12b", 144, 88, 12, 0, 0, (2, 0), 0.0),
    ("base_b_12b", 136, 72, 12, 0, 0, (2, 2), 0.5, {"inter_frac": 0.8}),
    ("main_addb_alf_12b", 200, 136, 12, 1, 1, (2, 2), 0.5, {"addb": 1, "alf": 1, "inter_frac": 1.0}),
    ("main_all_tools_b_12b", 264, 200, 12, 1, 1, (2, 2), 0.6, {"addb": 1, "alf": 1, "inter_frac": 0.7, "eipd": 1, "htdf_qp": 32, "dmvr_frac": 0.6, "ats_frac": 0.4, "ats_inter_frac": 0.4,
                                                              "btt_frac": 0.5, "split_prob": 0.4, "affine_frac": 0.3, "coded_frac": 0.8}),
    ("main_i_eipd_ibc_htdf_ctu128_12b", 264, 136, 12, 1, 1, (1, 0), 0.0, {"addb": 1, "alf": 1, "inter_frac": 0.0, "eipd": 1, "htdf_qp": 28, "ibc_frac": 0.3, "log2_ctu": 7, "ats_frac": 0.5,
                                                                         "btt_frac": 0.5, "split_prob": 0.5}),
]
POCS = [[4, 0, 2], [12, 16, 4]]      # L1 idx 2 has the POC of L0 idx 0 -> identical-motion candidates exist
CUR_POC = 8
QP_OFFSETS = (1, -2)


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def build_case(name, w, h, bd, admvp, iqt, n_refs, bi_frac, tools=None, seed=0, inter_frac=0.9, split_prob=0.5, qp_range=(20, 45),
               amp=2.0, oob_frac=0.1):
    tools = dict(tools or {})
    inter_frac = tools.get("inter_frac", inter_frac)
    split_prob = tools.get("split_prob", split_prob)
    log2_ctu = int(tools.get("log2_ctu", 6))
    rng = np.random.default_rng(zlib.crc32(name.encode()) % 1000 + seed)
    refs = {}
    for l in range(2):
        for i in range(n_refs[l]):
            pic = ol.Picture(w, h, POCS[l][i], synth.gen_picture(rng, w, h, bd))
            pic.pad_numpy()
            refs[(i, l)] = pic
    if (2, 1) in refs and (0, 0) in refs:
        refs[(2, 1)] = refs[(0, 0)]      # the same picture in both lists (same POC 4): ADDB compares pictures, not indices
    batch = synth.gen_frame(rng, w, h, bd, log2_ctu=log2_ctu, ats_frac=float(tools.get("ats_frac", 0.0)), ats_inter_frac=float(tools.get("ats_inter_frac", 0.0)), btt_frac=float(tools.get("btt_frac", 0.0)), inter_frac=inter_frac, bi_frac=bi_frac, n_refs=n_refs, oob_frac=oob_frac,
                            qp_range=qp_range, split_prob=split_prob, amp=amp, coded_frac=float(tools.get("coded_frac", 0.6)), eipd=bool(tools.get("eipd", 0)))
    batch["constrained_intra_pred"] = int(tools.get("constrained_intra", 0))
    if tools.get("htdf_qp"):
        batch["htdf_slice_qp"] = int(tools["htdf_qp"])
    if tools.get("ibc_frac"):
        synth.add_ibc(np.random.default_rng(8000 + seed), batch, w, h, log2_ctu, float(tools["ibc_frac"]))
    if tools.get("affine_frac"):
        synth.add_affine(np.random.default_rng(7000 + seed), batch, float(tools["affine_frac"]))
    if tools.get("dmvr_frac"):
        synth.add_dmvr(np.random.default_rng(9000 + seed), batch, float(tools["dmvr_frac"]))
    if n_refs[0] and n_refs[1]:      # force some identical-motion bi CUs
        sel = (batch["refi"][:, 0] >= 0) & (batch["refi"][:, 1] >= 0)
        idx = np.nonzero(sel)[0][::3]
        batch["mv"][idx, 1] = batch["mv"][idx, 0]
    alf_params = None
    if tools.get("alf"):
        ctu = 1 << log2_ctu
        n_ctu = ((w + ctu - 1) // ctu) * ((h + ctu - 1) // ctu)
        alf_params = synth.gen_alf_params(rng, n_ctu, across_tiles=int(tools.get("across_tiles", 0)),
                                          enable=tools.get("alf_enable", (1, 1, 1)))
    return {"name": name, "w": w, "h": h, "bd": bd, "admvp": admvp, "iqt": iqt, "refs": refs, "batch": batch,
            "alf_params": alf_params, "no_deblock": int(tools.get("no_deblock", 0)), "log2_ctu": log2_ctu,
            "addb": int(tools.get("addb", 0)), "alf": int(tools.get("alf", 0)), "eipd": int(tools.get("eipd", 0)),
            "alpha_off": int(tools.get("alpha_off", 0)), "beta_off": int(tools.get("beta_off", 0))}


def _start_picture(case):
    """Intra CUs are not reconstructed on this path: start from a deterministic mid-grey picture."""
    cur = ol.Picture(case["w"], case["h"], CUR_POC)
    for c in range(3):
        cur.bufs[c][:] = 1 << (case["bd"] - 1)
    return cur


def run_cpu(engine, case, deblock=True, simd=0, pad=True):
    """engine 'oracle' (oracle/liboracle.so) or 'ref' (the real reference through oracle/_ref). -> (final, pre-deblock, maps, resid)"""
    sp = abi.make_seq_params(case["w"], case["h"], case["bd"], log2_ctu=case.get("log2_ctu", 6), iqt=case["iqt"], admvp=case["admvp"],
                             addb=case.get("addb", 0), alf=case.get("alf", 0), eipd=case.get("eipd", 0))
    cb, keep = abi.make_cu_batch(case["batch"])
    cur = _start_picture(case)
    maps = ol.Maps(case["w"], case["h"])
    fr = ol.make_frame(cur, case["refs"], *QP_OFFSETS)
    m = maps.orc()
    resid = np.zeros(max(case["batch"]["n_coef"], 1), np.int16)
    deblock = deblock and not case.get("no_deblock")
    ap = keep_ap = None
    if case.get("alf_params") is not None:
        ap, keep_ap = abi.make_alf_params(case["alf_params"])
    if engine == "oracle":
        o = ol.oracle()
        o.orc_recon_batch(C.byref(sp), C.byref(fr), C.byref(cb), C.byref(m), _p(resid))
        pre = cur.copy()
        if deblock and case.get("addb"):
            o.orc_deblock_addb(C.byref(sp), C.byref(fr), C.byref(cb), C.byref(m), case["alpha_off"], case["beta_off"])
        elif deblock:
            o.orc_deblock_baseline(C.byref(sp), C.byref(fr), C.byref(cb), C.byref(m))
        if ap is not None:
            o.orc_alf(C.byref(sp), C.byref(fr.cur), C.byref(ap))
        if pad:
            o.orc_pad(C.byref(sp), C.byref(fr.cur))
    else:
        hn = ol.harness()
        hn.refh_recon_batch(C.byref(sp), C.byref(fr), C.byref(cb), C.byref(m), _p(resid), simd)
        pre = cur.copy()
        if deblock and case.get("addb"):
            hn.refh_deblock_addb(C.byref(sp), C.byref(fr), C.byref(cb), C.byref(m), case["alpha_off"], case["beta_off"])
        elif deblock:
            hn.refh_deblock_baseline(C.byref(sp), C.byref(fr), C.byref(cb), C.byref(m), simd)
        if ap is not None:
            hn.refh_alf(C.byref(sp), C.byref(fr.cur), C.byref(ap))
        if pad:
            hn.refh_pad(C.byref(sp), C.byref(fr.cur))
    return cur, pre, maps, resid


def dmvr_mvs(engine, case):
    """the vectors kept for temporal prediction of the batch's DMVR candidates ([n_sub_blocks][list][x/y], quarter samples) after the
    reconstruction of `case` by the oracle or the reference (xevdm_mc's dmvr_mv / core->mv)"""
    sp = abi.make_seq_params(case["w"], case["h"], case["bd"], log2_ctu=case.get("log2_ctu", 6), iqt=case["iqt"], admvp=case["admvp"],
                             addb=case.get("addb", 0), alf=case.get("alf", 0), eipd=case.get("eipd", 0))
    cb, keep = abi.make_cu_batch(case["batch"])
    cur = _start_picture(case)
    maps = ol.Maps(case["w"], case["h"])
    fr = ol.make_frame(cur, case["refs"], *QP_OFFSETS)
    m = maps.orc()
    out = np.full((1 << 16, 2, 2), -32768, np.int16)
    if engine == "oracle":
        ol.oracle().orc_recon_batch_ex(C.byref(sp), C.byref(fr), C.byref(cb), C.byref(m), None, _p(out))
    else:
        ol.harness().refh_recon_batch_ex(C.byref(sp), C.byref(fr), C.byref(cb), C.byref(m), None, 0, _p(out))
    return out[:int((out[:, 0, 0] != -32768).sum())].copy()


def run_gpu(case, deblock=True, pad=True, alf=True, resid=False, repeat=1, dmvr=False, ahead=False):
    """The HIP backend through the C ABI. -> list of padded planes (reference buffer geometry).
    ahead: the picture is decoded twice from two batch objects; the second one's residual pass is queued with the first picture's kernels
    (xgpu_batch_recon_ahead) and its picture is the one returned"""
    from xevd_amd.decoder import XgpuDecoder
    with XgpuDecoder(case["w"], case["h"], case["bd"], log2_ctu=case.get("log2_ctu", 6), iqt=case["iqt"], admvp=case["admvp"],
                     addb=case.get("addb", 0), alf=case.get("alf", 0), eipd=case.get("eipd", 0), max_pics=8) as dec:
        slots, by_obj = {}, {}
        for key, pic in case["refs"].items():
            if id(pic) not in by_obj:            # one device picture per distinct picture
                by_obj[id(pic)] = dec.pic_alloc()
                dec.pic_upload_padded(by_obj[id(pic)], pic.bufs)
            slots[key] = (by_obj[id(pic)], pic.poc)
        cur = dec.pic_alloc()
        dec.pic_upload_padded(cur, _start_picture(case).bufs)
        hb = dec.batch_create(case["batch"])
        if ahead:
            hb2 = dec.batch_create(case["batch"])
            dec.decode_picture(cur, CUR_POC, slots, hb, deblock=deblock and not case.get("no_deblock"), pad=pad, qp_u_offset=QP_OFFSETS[0], qp_v_offset=QP_OFFSETS[1],
                               alpha_off=case.get("alpha_off", 0), beta_off=case.get("beta_off", 0), alf=case.get("alf_params") if alf else None, next_batch=hb2)
            dec.sync()
            dec.pic_upload_padded(cur, _start_picture(case).bufs)
            hb = hb2
        for _ in range(repeat - 1):      # the same resident batch decoded again into the same slot (exercises per-batch device state)
            dec.decode_picture(cur, CUR_POC, slots, hb, deblock=deblock and not case.get("no_deblock"), pad=pad,
                               qp_u_offset=QP_OFFSETS[0], qp_v_offset=QP_OFFSETS[1],
                               alpha_off=case.get("alpha_off", 0), beta_off=case.get("beta_off", 0), alf=case.get("alf_params") if alf else None)
        dec.decode_picture(cur, CUR_POC, slots, hb, deblock=deblock and not case.get("no_deblock"), pad=pad,
                           qp_u_offset=QP_OFFSETS[0], qp_v_offset=QP_OFFSETS[1],
                           alpha_off=case.get("alpha_off", 0), beta_off=case.get("beta_off", 0), alf=case.get("alf_params") if alf else None)
        dec.sync()
        if dmvr:
            return dec.pic_download_padded(cur), dec.batch_dmvr_mvs(hb)
        if resid:
            return dec.pic_download_padded(cur), dec.batch_resid(hb, case["batch"]["n_coef"])
        return dec.pic_download_padded(cur)


def bench_case(name, seed=1000):
    """bench.py's workload `name` (same generator, same seed as rank 0 of the benchmark) as a picture-level test case"""
    import bench
    wl = bench.WORKLOADS[name]
    first, batches, alf = bench.make_stream(wl, seed, 1)
    refs = {}
    for l in range(1 + (wl["n_refs"][1] > 0)):
        pic = ol.Picture(wl["w"], wl["h"], POCS[l][0], first[l])
        pic.pad_numpy()
        refs[(0, l)] = pic
    return {"name": name, "w": wl["w"], "h": wl["h"], "bd": wl["bd"], "admvp": wl["admvp"], "iqt": wl["iqt"], "refs": refs, "batch": batches[0],
            "alf_params": alf, "no_deblock": 0, "log2_ctu": 6, "addb": wl["addb"], "alf": wl["alf"], "eipd": 0, "alpha_off": 0, "beta_off": 0}
