"""Frame-level test helpers: synthetic EVC Baseline streams, decoded (a) by the REAL reference decoder through its public API
(oracle/_ref/libref_decode.so), (b) by our parser + the CPU oracle, (c) by our parser + the HIP backend."""
import ctypes as C
import os

import numpy as np

import oracle_lib as ol
from xevd_amd import abi, stream, synth

REF_DECODE = os.path.join(ol.ORACLE_DIR, "_ref", "ref_decode")


def have_ref_decoder():
    return os.path.exists(REF_DECODE)


def gop_tids(log2_sub_gop):
    """temporal ids of one sub-GOP in decoding order: 0, 1, 2, 2, 3, 3, 3, 3, ... (xevd_poc_derivation's expected order)"""
    tids = [0]
    for t in range(1, log2_sub_gop + 1):
        tids += [t] * (1 << (t - 1))
    return tids


def make_stream(width, height, n_pics, bit_depth=8, seed=0, max_refs=1, skip_frac=0.15, inter_frac=0.85, deblock=True, qp_offsets=(0, 0),
                cu_qp_delta=True, split_prob=0.5, idr_period=0, log2_sub_gop=0, bi_frac=0.5, direct_frac=0.1,
                main=False, iqt=False, ats=False, addb=False, addb_offsets=(0, 0), alf=False, sign=False, eipd=False, crop=(0, 0, 0, 0),
                chroma_qp_points=None, dra=None, htdf=False, ibc_log_max=0, ibc_frac=0.25, alf_fixed=False, admvp=False, amvr=False, hmvp=False, dmvr=False, mmvd=False,
                tiles=None, affine=False, affine_frac=0.4, qp_delta_area=0, rpl=False, pocs=False, rpl_in_sps=False, cm_init=False, adcc=False, max_level=6, qp_range=(22, 37), btt=None, dual_tree=False, slices=None, arbitrary_slices=False, suco=None):
    """-> bytes.  Picture 0 is an IDR.  log2_sub_gop = 0: IPPP; n: hierarchical sub-GOPs of 2^n pictures, the layer-0 picture of
    each a P picture, the others B pictures (bi-prediction, temporal direct and two-list skip CUs).
    sign: every picture is followed by a picture-signature SEI with the MD5s of the ORACLE's reconstruction of the stream so far."""
    rng = np.random.default_rng(seed)
    w = stream.StreamWriter(width, height, bit_depth, max_refs, qp_offsets[0], qp_offsets[1], deblock, cu_qp_delta, log2_sub_gop,
                            main=main, iqt=iqt, ats=ats, addb=addb, alpha_off=addb_offsets[0], beta_off=addb_offsets[1], alf=alf, eipd=eipd, crop=crop, chroma_qp_points=chroma_qp_points,
                            dra_aps_id=None if dra is None else 3, htdf=htdf, ibc_log_max=ibc_log_max, admvp=admvp, amvr=amvr, hmvp=hmvp, dmvr=dmvr, mmvd=mmvd, tiles=tiles, affine=affine, qp_delta_area=qp_delta_area, rpl=rpl, pocs=pocs, rpl_in_sps=rpl_in_sps, cm_init=cm_init, adcc=adcc, btt=btt, suco=suco)
    n_ctu = ((width + 63) // 64) * ((height + 63) // 64)
    tids = gop_tids(log2_sub_gop)
    try:
        if slices:      # several slice NAL units per picture: tile rectangles (first, last[, slice_qp[, deblock_on]])
            if arbitrary_slices:
                w.set_arbitrary_slices(True)
            w.set_slices(slices)
        if dra is not None:      # one of oracle_lib.DRA_SETS: the PPS switches DRA on for every picture with parameter set 3
            d = ol.DRA_SETS[dra]
            w.add_dra_aps(3, d["in_ranges"], d["scales"], d["cb"], d["cr"], d["table_idx"])
        since_idr = 0
        for k in range(n_pics):
            idr = k == 0 or (idr_period and k % idr_period == 0)
            if idr:
                since_idr = 0
            tid = 0 if idr else tids[(since_idr - 1) % len(tids)]
            is_b = (not idr) and tid > 0
            only_inter = None if idr else []      # P / B pictures: also splits whose children get the "inter only" mode constraint (sps_btt_flag with tool_admvp)
            part = None if btt is None else synth.gen_partition_tree(rng, width, height, w.split_allowed, split_prob, inter_only=only_inter, dual_tree=dual_tree)      # sps_btt_flag: a legal binary / ternary tree
            b = synth.gen_frame(rng, width, height, bit_depth, partition=part, inter_frac=0.0 if idr else inter_frac, n_refs=(max_refs, max_refs if is_b else 0),
                                bi_frac=bi_frac if is_b else 0.0, split_prob=split_prob, coded_frac=0.6, max_level=max_level, amp=1.0, qp_range=qp_range,
                                ats_frac=0.5 if ats else 0.0, ats_inter_frac=0.5 if ats else 0.0, eipd=eipd)
            if only_inter:      # CUs below a mode-constrained split: inter CUs (an intra one becomes a list-0 CU with a small vector)
                sel = np.array(only_inter)
                fix = sel[b["pred_mode"][sel] == 0]
                b["pred_mode"][fix] = 1
                b["refi"][fix, 0] = 0; b["refi"][fix, 1] = -1
                b["mv"][fix] = 0
                b["mv"][fix, 0, :] = rng.integers(-20, 21, (len(fix), 2))
            if not idr:
                inter = b["pred_mode"] == 1
                r = rng.random(len(inter))
                b["pred_mode"] = np.where(inter & (r < skip_frac), 2, b["pred_mode"]).astype(np.uint8)
                if is_b or admvp:      # temporal direct mode (B), or with tool_admvp merge mode (P and B)
                    # (not the CUs that carry an ATS-inter TU: the writer drops ATS-inter of a direct-mode CU, the TU-sized coefficient
                    #  block would be read as a CU-sized one, and 64-point blocks with coefficients past position 32 hit the reference's
                    #  AVX-vs-C difference)
                    plain = np.ones(len(inter), bool) if b.get("ats_inter") is None else (b["ats_inter"] == 0)
                    b["pred_mode"] = np.where(inter & plain & (r >= skip_frac) & (r < skip_frac + direct_frac), 3, b["pred_mode"]).astype(np.uint8)
            if affine and not idr:      # sps->tool_affine: affine merge CUs (skip / merge mode, 8x8 and up) and affine inter CUs with coded control points (16x16 and up)
                synth.add_affine(rng, b, affine_frac)
            if ibc_log_max:      # sps->ibc_flag: intra block copy CUs in every slice type (I slices included), up to the signalled size
                synth.add_ibc(rng, b, width, height, 6, ibc_frac, max_log2=ibc_log_max)
                if b.get("tree") is not None:      # the chroma block of a local dual tree is always intra-predicted
                    b["pred_mode"][b["tree"] == 2] = 0
                if only_inter:      # ... and no IBC below an "inter only" split
                    sel = np.array(only_inter)
                    fix = sel[b["pred_mode"][sel] == 6]
                    b["pred_mode"][fix] = 1
                    b["refi"][fix, 0] = 0; b["refi"][fix, 1] = -1
                    b["mv"][fix] = 0
                    b["mv"][fix, 0, :] = rng.integers(-20, 21, (len(fix), 2))
            if alf:      # a fresh parameter set every other picture (different shapes of the syntax), per-CTU flags, some pictures without ALF / map
                if k % 2 == 0:
                    nf = int(rng.integers(1, 6))
                    t7 = bool(rng.integers(0, 2))
                    w.add_alf_aps(k % 32, luma=rng.integers(-12, 13, (nf, 12 if t7 else 6)), chroma=rng.integers(-10, 11, 6), type7=t7,
                                  delta_idx=rng.integers(0, nf, 25), coef_delta_flag=int(nf > 1 and rng.random() < 0.4),
                                  pred_mode_flag=int(rng.random() < 0.5), filter_coef_flag=np.maximum(rng.integers(0, 2, 25), np.arange(25) == 0),
                                  fixed_pattern=(k // 2) % 3 if alf_fixed else 0, fixed_usage=rng.integers(0, 2, 25), fixed_idx=rng.integers(0, 16, 25))
                    last_aps = k % 32
                mode = k % 5
                w.set_slice_alf(mode != 3, last_aps, last_aps, chroma_idc=int(rng.integers(0, 4)),
                                ctb_flag=None if mode == 4 else (rng.random(n_ctu) < 0.7).astype(np.uint8))
            w.add_picture(b, stream.SLICE_I if idr else (stream.SLICE_B if is_b else stream.SLICE_P), slice_qp=int(rng.integers(24, 40)), idr=idr,
                          temporal_id=tid)
            if sign:
                w.add_md5_sei(decode_oracle(w.bytes(), order="decoding")[-1])
            if dmvr and (hmvp or mmvd) and k + 1 < n_pics:
                # the writer derives motion like a decoder, refined vectors included: it needs the decoded samples of what it has written so far
                lumas = []
                decode_oracle(w.bytes(), order="decoding", keep_luma=lumas)
                w.set_ref_luma(lumas[-1][0], lumas[-1][1], abi.PAD_L)
            since_idr += 1
        return w.bytes()
    finally:
        w.close()


def decode_reference(data, width, height, max_pics=64, threads=1, main=False):
    """The reference decoder's output pictures (output order) as lists of [Y, U, V] int16 arrays; one process per decode."""
    import subprocess
    import tempfile
    with tempfile.TemporaryDirectory() as td:
        fin, fout = os.path.join(td, "s.evc"), os.path.join(td, "s.raw")
        with open(fin, "wb") as f:
            f.write(data)
        r = subprocess.run([REF_DECODE + ("_main" if main else ""), fin, fout, str(width), str(height), str(threads)],
                           stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, timeout=300)
        if r.returncode != 0:
            raise RuntimeError(f"reference decoder failed ({r.returncode}): {r.stderr.decode()[-300:]}")
        n = int(r.stderr.decode().split()[-2])
        out = np.fromfile(fout, np.int16)
    elems = width * height * 3 // 2
    pics = []
    for k in range(min(n, max_pics, len(out) // elems)):
        p = out[k * elems:(k + 1) * elems]
        pics.append([p[:width * height].reshape(height, width).copy(),
                     p[width * height:width * height * 5 // 4].reshape(height // 2, width // 2).copy(),
                     p[width * height * 5 // 4:].reshape(height // 2, width // 2).copy()])
    return pics


def _output_order(pics):
    """decoding order -> output order: ascending POC inside every IDR period (what xevd_pull's bumping produces)"""
    out, epoch = [], -1
    for p, planes in pics:
        if p["is_idr"]:
            epoch += 1
        out.append(((epoch, p["poc"]), planes))
    return [v for _, v in sorted(out, key=lambda kv: kv[0])]


def decode_oracle(data, order="output", keep_luma=None, keep_params=None):
    """Our parser + the CPU oracle (oracle/liboracle.so). -> pictures in output (or decoding) order.
    keep_luma: a list that receives (poc, padded luma plane) of every picture in decoding order; keep_params: one that receives the parser's picture
    dicts (streams whose parser needs the decoded reference samples cannot be parsed without decoding them)."""
    o = ol.oracle()
    dpb, out = {}, []
    for p in stream.iter_stream(data):      # a generator: with tool_dmvr the parser needs this picture's refined vectors before it parses the next one
        w, h, bd = p["width"], p["height"], p["bit_depth"]
        sp = abi.make_seq_params(w, h, bd, iqt=p["iqt"], admvp=p["admvp"], addb=p["addb"], alf=p["tool_alf"], eipd=p["eipd"])
        if p["chroma_qp_tables"] is not None:
            keep_tables = [np.ascontiguousarray(t, np.int8) for t in p["chroma_qp_tables"]]
            for i in range(2):
                sp.chroma_qp_table[i] = keep_tables[i].ctypes.data_as(C.POINTER(C.c_int8))
        cb, keep = abi.make_cu_batch(p["batch"])
        cur = ol.Picture(w, h, p["poc"])
        refs = {(i, l): dpb[poc] for l in range(2) for i, poc in enumerate(p["refs"][l])}
        fr = ol.make_frame(cur, refs, p["qp_u_offset"], p["qp_v_offset"])
        maps = ol.Maps(w, h)
        m = maps.orc()
        if p["n_dmvr_sub"]:
            dmv = np.zeros((p["n_dmvr_sub"], 2, 2), np.int16)
            o.orc_recon_batch_ex(C.byref(sp), C.byref(fr), C.byref(cb), C.byref(m), None, dmv.ctypes.data)
            p["dmvr_feedback"](dmv)
        else:
            o.orc_recon_batch(C.byref(sp), C.byref(fr), C.byref(cb), C.byref(m), None)
        if p["deblock_on"] and p["addb"]:
            o.orc_deblock_addb(C.byref(sp), C.byref(fr), C.byref(cb), C.byref(m), p["alpha_off"], p["beta_off"])
        elif p["deblock_on"]:
            o.orc_deblock_baseline(C.byref(sp), C.byref(fr), C.byref(cb), C.byref(m))
        if p["alf"] is not None:
            ap, keep_ap = abi.make_alf_params(p["alf"])
            o.orc_alf(C.byref(sp), C.byref(fr.cur), C.byref(ap))
        o.orc_pad(C.byref(sp), C.byref(fr.cur))
        if p["needs_ref_luma"]:      # tool_dmvr with tool_hmvp / tool_mmvd: the parser refines vectors itself on the decoded reference samples
            p["set_ref_luma"](p["poc"], cur.bufs[0], abi.PAD_L)
        if keep_luma is not None:
            keep_luma.append((p["poc"], cur.bufs[0]))
        if keep_params is not None:
            keep_params.append(p)
        if p["is_idr"]:
            dpb.clear()
        for poc in p["release"]:
            dpb.pop(poc, None)
        if p["is_ref"]:
            dpb[p["poc"]] = cur
        planes = [cur.active(c).copy() for c in range(3)]
        if p["dra"] is not None:      # the post-filter runs on the OUTPUT copy only (xevd_pull), references stay unmapped
            planes = ol.dra_apply(planes, p["dra"])
        out.append((p, planes))
    if order == "decoding":
        return [planes for _, planes in out]
    return _output_order(out)


def decode_gpu(data, verify_md5=False, parser_threads=1):
    """Our parser + the HIP backend through the two C ABIs (xevd_amd/player.py). -> pictures in output order."""
    from xevd_amd.player import StreamDecoder
    return [planes for _, planes in StreamDecoder(data, verify_md5=verify_md5, parser_threads=parser_threads).output_order()]
