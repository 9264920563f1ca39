#!/usr/bin/env python3
"""bench.py - frames/s of the MI355X reconstruction backend on BASELINE.json's workload, with the roofline of the dominant kernel and a CPU baseline.

A "step" is one picture through the whole hot path - everything xevdm_dec_nalu does after entropy decoding (src_main/xevdm.c:3136-3219): dequant + inverse
transform of every coded TB (k_itdq), MC + residual add + clip + SCU-map update of every inter CU (k_inter: one launch, a workgroup per 64x64 region), intra CUs in
dependency order (k_intra_l1 + the data-flow launch k_intra, which also carries the NEXT picture's residual pass), ADDB + ALF + border padding in one pass (k_addb_alf).  Pictures chain like a stream: picture k is predicted from pictures k-1 and k-2 (a 3-slot DPB
ring), so steps are serially dependent exactly like a real decode.

What the JSON line's figures are, all at the command line the driver uses (`--steps 20 --warmup 5` included):
  value / kernel_only_fps   the benchmark contract's number: the CU batches (the post-entropy records) are resident in HBM when the timed region starts; W warm-up
                            pictures, then exactly K timed ones between two barrier + synchronize pairs.
  roofline                  the dominant kernel by HIP-event time against SURVEY 8(d)'s bytes of its pass, timed in a second loop of K pictures with every kernel alone
                            on the stream.  `traffic` comes from the committed counter passes (profiles/latest_pmc.json) and is dropped (null) when the kernel's sources
                            no longer hash to what those passes measured.
  end_to_end_fps            host CU batches -> host YUV (builder + H2D + kernels + conversion + D2H inside the timed region).  Its OWN fixed length
  two_contexts              (E2E_PICTURES after E2E_WARMUP; CTX_PICTURES per context after CTX_WARMUP), independent of --steps: a secondary figure that moved with
                            the step count was the round-4 review's finding.
  cpu_baseline              the same batch through the reference's functions (oracle/_ref) or the CPU oracle on a bounded sample, + the reference decoder on a
                            real stream of the workload's shape.

    python bench.py --gpus 1 --steps 200 --warmup 20            # single GPU
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...   # one rank per GPU

With N > 1 the workload is BASELINE.json configs[4]: independent 4K Main streams cut into GOP-sized jobs that the ranks draw from one
host work queue (streams/GOPs shard across GPUs, no collective, no RCCL); N x steps pictures in total, so scaling is "weak"
and value = N x steps / max-over-ranks time; per-rank picture counts and rates are in `per_rank`.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# the files a kernel family is compiled from (xevd_amd/csrc): profiles/latest_pmc.json records their hash at the commit its counter passes ran on
KERNEL_SOURCES = {"inter": ("k_inter.hip", "mc_filters.h"), "alf": ("k_alf.hip", "addb_filter.h"), "itdq": ("k_itdq.hip", "itdq_body.h"),
                  "dbk_v": ("k_addb.hip", "k_deblock.hip", "addb_filter.h"), "dbk_h": ("k_addb.hip", "k_deblock.hip", "addb_filter.h"),
                  "intra": ("k_intra.hip", "intra_pred.h", "itdq_body.h"), "intra_itdq": ("k_intra.hip", "intra_pred.h", "itdq_body.h")}


def kernel_sources_sha256(kernel):
    import hashlib
    h = hashlib.sha256()
    for f in KERNEL_SOURCES.get(kernel, ()):
        h.update(open(os.path.join(ROOT, "xevd_amd", "csrc", f), "rb").read())
    return h.hexdigest()[:16]


WORKLOADS = {
    # BASELINE.json configs[1]: Baseline profile, 1080p, 8 bit, IPPP, one reference
    "cfg2_base_1080p_8b_ippp": dict(w=1920, h=1080, bd=8, admvp=0, iqt=0, addb=0, alf=0, n_refs=(1, 0), bi_frac=0.0),
    "base_8k_10b_ippp": dict(w=7680, h=4320, bd=10, admvp=0, iqt=0, addb=0, alf=0, n_refs=(1, 0), bi_frac=0.0),
    # configs[2]: Main profile 4K 10 bit, two reference lists, 50% bi-prediction, 8-tap MC tables, IQT, ADDB, ALF
    "cfg3_main_4k_10b_ra": dict(w=3840, h=2160, bd=10, admvp=1, iqt=1, addb=1, alf=1, n_refs=(1, 1), bi_frac=0.5),
    # configs[4]: independent 4K Main streams sharded over the GPUs of the node through a host work queue (the default with --gpus N > 1):
    # the same stream shape as configs[2], cut into jobs of GOP_PICTURES pictures that the ranks draw from one queue
    "cfg5_main_4k_10b_ra_streams": dict(w=3840, h=2160, bd=10, admvp=1, iqt=1, addb=1, alf=1, n_refs=(1, 1), bi_frac=0.5),
    # configs[3]: the same at 8K - the configuration the metric (fps + HBM GB/s at 4K/8K) is quoted on
    "cfg4_main_8k_10b_ra": dict(w=7680, h=4320, bd=10, admvp=1, iqt=1, addb=1, alf=1, n_refs=(1, 1), bi_frac=0.5),
    # not a BASELINE config: cfg4 with 30 % of the inter CUs of 8x8 and above affine (2 / 3 control points; sub-block translation and EIF) - k_affine's cost
    # not a BASELINE config either: cfg4 with HTDF on (slice QP 32) - every coded inter CU of a filterable size and every intra CU becomes a node of the
    # data-flow kernel, which is where the time goes (kernels["intra"])
    "main_8k_10b_ra_htdf": dict(w=7680, h=4320, bd=10, admvp=1, iqt=1, addb=1, alf=1, n_refs=(1, 1), bi_frac=0.5, htdf_qp=32),
    "main_8k_10b_ra_affine30": dict(w=7680, h=4320, bd=10, admvp=1, iqt=1, addb=1, alf=1, n_refs=(1, 1), bi_frac=0.5, affine_frac=0.3),
    # nor this one: cfg4 as a hierarchical-B picture (list 1 = the picture AFTER it in output order) with 60 % of the plain inter CUs in merge mode
    # (xgpu_cu_batch.dmvr): the bi-predicted ones of 8x8 and above are refined and predicted by k_dmvr (kernels["dmvr"]), k_inter leaves them out
    "main_8k_10b_ra_dmvr": dict(w=7680, h=4320, bd=10, admvp=1, iqt=1, addb=1, alf=1, n_refs=(1, 1), bi_frac=0.5, dmvr_frac=0.6),
}
DEFAULT_WORKLOAD = "cfg4_main_8k_10b_ra"
DEFAULT_WORKLOAD_MULTI = "cfg5_main_4k_10b_ra_streams"
E2E_PICTURES, E2E_WARMUP = 256, 16     # the end-to-end leg's own length (pictures), whatever --steps says
CTX_PICTURES, CTX_WARMUP = 512, 32     # per context, the several-contexts leg
GOP_PICTURES = 8            # pictures per job of the multi-GPU work queue (one closed GOP)
HBM_PEAK_GBPS = 8000.0     # MI355X_MICROARCH.md: HBM3E 8 TB/s spec


def algorithmic_bytes(batch, w, h, addb=False):
    """SURVEY 8(d) per-sample byte counts applied to the actual batch (2 B/sample, halo re-reads not credited)."""
    cw = (1 << batch["log2w"].astype(np.int64))
    ch = (1 << batch["log2h"].astype(np.int64))
    samples = cw * ch * 3 // 2
    inter = batch["pred_mode"] != 0
    lists = (batch["refi"] >= 0).sum(1)
    coded = np.zeros(len(cw), np.int64)
    for c in range(3):
        coded += ((batch["cbf"] >> c) & 1) * (cw * ch >> (2 if c else 0))
    # the inter CUs k_inter itself predicts: not the affine ones (k_affine) and not the DMVR candidates with two references and 8x8 or more samples (k_dmvr;
    # the benchmark's B pictures lie between their references, so every such candidate is refined)
    aff = batch["affine"] != 0 if batch.get("affine") is not None else np.zeros(len(cw), bool)
    dm = ((batch["dmvr"] != 0) & (lists == 2) & (batch["log2w"] >= 3) & (batch["log2h"] >= 3) & ~aff) if batch.get("dmvr") is not None else np.zeros(len(cw), bool)
    plain = inter & ~aff & ~dm

    def pred_bytes(m):
        return int((samples[m] * (2 * lists[m] + 2)).sum() + 2 * coded[m].sum())
    b_inter, b_aff, b_dmvr = pred_bytes(plain), pred_bytes(inter & aff), pred_bytes(inter & dm)
    b_itdq = int(4 * coded.sum())
    b_intra = int((2 * samples[~inter]).sum() + 2 * coded[~inter].sum())
    s_pic = w * h * 3 // 2
    # deblocking: SURVEY 8(d) prices two passes of 4 B/sample.  ADDB runs as ONE fused kernel (k_addb_fused: one read + one write), timed as
    # "dbk_v" - it is credited with the 4 B/sample it has to move, not with the second pass it no longer makes
    return {"inter": b_inter, "affine": b_aff, "dmvr": b_dmvr, "itdq": b_itdq, "intra": b_intra, "dbk_v": 4 * s_pic, "dbk_h": 0 if addb else 4 * s_pic, "alf": 4 * s_pic}


INTER_CLASS_PICTURES = 6      # timed pictures per CU-size class of inter_by_cu_size_leg (after 2 untimed ones); fixed, like the other secondary legs


def inter_by_cu_size_leg(dec, wl, slots, alf):
    """k_inter's duration (HIP events, every kernel alone on the stream) on pictures of ONE CU size each - 64x64 .. 4x4, all inter, otherwise the workload's statistics
    (vectors N(0, 8 px) per CU, sub-sample classes, share of bi-predicted CUs, 60 % coded) - next to the workload's own mix: which CU sizes the inter pass pays for.
    64 / 32: the region and tile roles (windows shared through LDS); 16 / 8 / 4: the split role, one lane per 4x4 unit with its own 11x11 window."""
    from xevd_amd import synth
    rng = np.random.default_rng(77)
    two_lists = wl["n_refs"][1] > 0
    out = {}
    for name, sp, ml in (("64x64", 0.0, 2), ("32x32", 1.0, 5), ("16x16", 1.0, 4), ("8x8", 1.0, 3), ("4x4", 1.0, 2)):
        b = synth.gen_frame(rng, wl["w"], wl["h"], wl["bd"], inter_frac=1.0, bi_frac=wl["bi_frac"], coded_frac=0.6, n_refs=wl["n_refs"], qp_range=(22, 37), mv_sigma_px=8.0,
                            oob_frac=0.05, split_prob=sp, min_log2=ml, admvp=bool(wl["admvp"]))
        h = dec.batch_create(b)

        def step(k):
            cur, r0, r1 = slots[(k + 2) % 3], slots[(k + 1) % 3], slots[k % 3]
            refs = {(0, 0): (r0, k)}
            if two_lists:
                refs[(0, 1)] = (r1, k - 1)
            dec.decode_picture(cur, k + 1, refs, h, alf=alf)
        for k in range(2):
            step(k)
        dec.sync(); dec.timing_enable(True); dec.timing_reset()
        for k in range(INTER_CLASS_PICTURES):
            step(2 + k)
        tim = dec.timing_get(); dec.timing_enable(False)
        ab = algorithmic_bytes(b, wl["w"], wl["h"], bool(wl["addb"]))["inter"]
        us = 1e3 * tim["inter"][0] / max(tim["inter"][1], 1)
        out[name] = {"avg_us": round(us, 1), "cus": int(len(b["x"])), "algorithmic_gbps": round(ab / us / 1e3, 1)}
        dec.batch_destroy(h)
    return out


def make_stream(wl, seed, n_batches):
    from xevd_amd import synth
    rng = np.random.default_rng(seed)
    first = [synth.gen_picture(rng, wl["w"], wl["h"], wl["bd"]) for _ in range(2)]
    batches = [synth.gen_frame(rng, wl["w"], wl["h"], wl["bd"], inter_frac=0.9, bi_frac=wl["bi_frac"], coded_frac=0.6,
                               n_refs=wl["n_refs"], qp_range=(22, 37), mv_sigma_px=8.0, oob_frac=0.05, admvp=bool(wl["admvp"]))
               for _ in range(n_batches)]
    if wl.get("htdf_qp"):
        for b in batches:
            b["htdf_slice_qp"] = wl["htdf_qp"]
    if wl.get("affine_frac"):
        for b in batches:
            synth.add_affine(rng, b, wl["affine_frac"])
    if wl.get("dmvr_frac"):
        for b in batches:
            synth.add_dmvr(rng, b, wl["dmvr_frac"])
    n_ctu = ((wl["w"] + 63) // 64) * ((wl["h"] + 63) // 64)
    alf = synth.gen_alf_params(rng, n_ctu, ctb_on_frac=1.0) if wl["alf"] else None     # all CTUs on
    return first, batches, alf


def cpu_baseline(wl, first, batch, alf, budget_s=15.0):
    """The CPU side of the comparison on this host: the reference's own functions (AVX2 tables, one thread)
    through oracle/_ref when that was built in the development container, else the plain-C oracle port."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import ctypes as C
    import oracle_lib as ol
    from xevd_amd import abi
    sp = abi.make_seq_params(wl["w"], wl["h"], wl["bd"], iqt=wl["iqt"], admvp=wl["admvp"], addb=wl["addb"], alf=wl["alf"])
    cb, keep = abi.make_cu_batch(batch)
    ref = ol.Picture(wl["w"], wl["h"], 0, first[0])
    ref.pad_numpy()
    ref2 = ol.Picture(wl["w"], wl["h"], 2 if wl.get("dmvr_frac") else -1, first[1])      # DMVR: a B picture between its two references (as the timed steps)
    ref2.pad_numpy()
    cur = ol.Picture(wl["w"], wl["h"], 1)
    ap = keep_ap = None
    if alf is not None:
        ap, keep_ap = abi.make_alf_params(alf)
    maps = ol.Maps(wl["w"], wl["h"])
    m = maps.orc()
    kind = "reference" if ol.have_ref() else "port"
    n, t0 = 0, time.perf_counter()
    while True:
        fr = ol.make_frame(cur, {(0, 0): ref, (0, 1): ref2})
        maps.map_scu[:] = 0          # a new picture starts with no SCU reconstructed (intra availability = COD flags)
        if kind == "reference":
            hn = ol.harness()
            hn.refh_recon_batch(C.byref(sp), C.byref(fr), C.byref(cb), C.byref(m), None, 1)
            if wl["addb"]:
                hn.refh_deblock_addb(C.byref(sp), C.byref(fr), C.byref(cb), C.byref(m), 0, 0)
            else:
                hn.refh_deblock_baseline(C.byref(sp), C.byref(fr), C.byref(cb), C.byref(m), 1)
            if ap is not None:
                hn.refh_alf(C.byref(sp), C.byref(fr.cur), C.byref(ap))
            hn.refh_pad(C.byref(sp), C.byref(fr.cur))
        else:
            o = ol.oracle()
            o.orc_recon_batch(C.byref(sp), C.byref(fr), C.byref(cb), C.byref(m), None)
            if wl["addb"]:
                o.orc_deblock_addb(C.byref(sp), C.byref(fr), C.byref(cb), C.byref(m), 0, 0)
            else:
                o.orc_deblock_baseline(C.byref(sp), C.byref(fr), C.byref(cb), C.byref(m))
            if ap is not None:
                o.orc_alf(C.byref(sp), C.byref(fr.cur), C.byref(ap))
            o.orc_pad(C.byref(sp), C.byref(fr.cur))
        if n == 0:
            first_out = [b.copy() for b in cur.bufs]      # padded planes of the first picture: what the GPU's picture is compared with
        ref2, ref, cur = ref, cur, ref2
        n += 1
        dt = time.perf_counter() - t0
        if dt > budget_s or n >= 64:
            break
    return first_out, {"value": round(n / dt, 3), "unit": "frames/s", "cores": 1, "kind": kind,
            "sample": f"{n} pictures of the same {wl['w']}x{wl['h']} workload (recon + deblock" + (" + ALF" if alf is not None else "")
                      + " + pad), single thread, "
                      + ("the reference's own functions (AVX2/SSE tables where it has them; ADDB and ALF are scalar C there) via oracle/_ref"
                         if kind == "reference" else "plain-C oracle port")}


def write_bench_stream(wl, gop_pictures, repeats, seed=77, refine_tools=False, device=0):
    """A real bitstream of the workload's shape, written by this repository's front end (xevd_amd/host): Main workloads -> random-access coding -
    hierarchical-B sub-GOPs of 8 (temporal layers 0..3), two reference lists with two pictures each, tool_admvp (merge / skip candidates, 8-tap
    interpolation), IQT, ADDB, ALF, a 4x4 tile grid; Baseline workloads -> IPPP with one reference.  ONE closed GOP (an IDR + gop_pictures - 1
    pictures) is generated and its bytes are repeated `repeats` times behind the parameter sets: every IDR period decodes to the same pictures, so
    the reference decoder (1.4 pictures/s at 8K) only has to decode one period for the comparison.  -> (prefix + one GOP, whole stream, description)"""
    from xevd_amd import stream, synth
    w, h, bd = wl["w"], wl["h"], wl["bd"]
    main = bool(wl["addb"] or wl["iqt"] or wl["alf"])
    rng = np.random.default_rng(seed)
    tids = [0, 1, 2, 2, 3, 3, 3, 3]
    wr = stream.StreamWriter(w, h, bd, 2 if main else 1, main=main, iqt=bool(wl["iqt"]), addb=bool(wl["addb"]), alf=bool(wl["alf"]), admvp=bool(wl["admvp"]),
                             log2_sub_gop=3 if main else 0, tiles=(4, 4, 1) if main else None, dmvr=refine_tools, hmvp=refine_tools, mmvd=refine_tools, amvr=refine_tools)
    try:
        if wl["alf"]:       # one parameter set for the whole stream: the repeated IDR periods must not see a later one
            wr.add_alf_aps(0, luma=rng.integers(-12, 13, (5, 12)), chroma=rng.integers(-10, 11, 6), type7=True, delta_idx=rng.integers(0, 5, 25))
        for k in range(gop_pictures):
            idr = k == 0
            tid = 0 if idr or not main else tids[(k - 1) % 8]
            is_b = main and not idr and tid > 0
            b = synth.gen_frame(rng, w, h, bd, inter_frac=0.0 if idr else 0.9, n_refs=(2 if main else 1, 2 if is_b else 0), bi_frac=wl["bi_frac"] if is_b else 0.0,
                                coded_frac=0.6, max_level=6, amp=1.0, admvp=bool(wl["admvp"]))
            if not idr:      # a share of skip and (B pictures / tool_admvp) merge-mode CUs, like the streams of tests/test_stream.py
                inter = b["pred_mode"] == 1
                r = rng.random(len(inter))
                b["pred_mode"] = np.where(inter & (r < 0.15), 2, b["pred_mode"]).astype(np.uint8)
                if is_b or wl["admvp"]:
                    b["pred_mode"] = np.where(inter & (r >= 0.15) & (r < 0.25), 3, b["pred_mode"]).astype(np.uint8)
            if wl["alf"]:
                wr.set_slice_alf(True, 0, 0, chroma_idc=3)
            wr.add_picture(b, stream.SLICE_I if idr else (stream.SLICE_B if is_b else stream.SLICE_P), slice_qp=30, idr=idr, temporal_id=tid)
            if refine_tools and k + 1 < gop_pictures:
                from xevd_amd import abi, player
                last = None
                for p, _ in player.StreamDecoder(wr.bytes(), device=device, parser_threads=8).pictures(download=False):
                    last = p
                if last is not None and last.get("_luma") is not None:      # (a picture that is kept as a reference)
                    wr.set_ref_luma(last["poc"], last["_luma"][1], abi.PAD_L)
        data = wr.bytes()
    finally:
        wr.close()
    # the first IDR slice NAL starts the GOP; everything before it (SPS, PPS, APS) is sent once
    pos = 0
    while pos + 6 <= len(data):
        ln = int.from_bytes(data[pos:pos + 4], "big")
        if (((data[pos + 4] << 8) | data[pos + 5]) >> 9 & 63) - 1 == 1:
            break
        pos += 4 + ln
    prefix, gop = data[:pos], data[pos:]
    what = (f"{w}x{h} {bd}-bit, closed GOPs of {gop_pictures} pictures x {repeats}, " +
            ("Main profile, random access: hierarchical-B sub-GOPs of 8, two lists of two references, tool_admvp (merge / skip, 8-tap tables), IQT, ADDB, ALF, "
             "4x4 tiles" + (", tool_dmvr + tool_hmvp + tool_mmvd + tool_amvr (the parser refines merge vectors itself on decoded reference luma)" if refine_tools else "")
             if main else "Baseline profile, IPPP, one reference") + f"; {len(gop) * 8 / gop_pictures / 1e6:.2f} Mbit per picture; written by xevd_amd/host")
    return prefix + gop, prefix + gop * repeats, what


def host_cpu_quota():
    """CPUs this process may use at once: the cgroup's cpu.max quota (the GPU boxes of this pool: 16 CPUs under 256 hardware threads) or the CPU count"""
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            return max(1, int(round(int(q) / int(per))))
    except Exception:
        pass
    return os.cpu_count() or 1


def run_evc_decode(args, timeout=900):
    """examples/evc_decode --json ... -> its report (dict) or {"error": ...}"""
    import subprocess
    ours = os.path.join(ROOT, "examples", "evc_decode")
    r = subprocess.run([ours, "--json"] + args, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=timeout)
    if r.returncode != 0:
        return {"error": r.stderr.decode()[-300:]}
    return json.loads(r.stdout.decode().strip().splitlines()[-1])


def file_md5s(path, period_bytes, limit=None):
    """one MD5 per `period_bytes` of a file (the output pictures of one IDR period), the first `limit` periods"""
    import hashlib
    out = []
    with open(path, "rb") as f:
        while limit is None or len(out) < limit:
            hh, left = hashlib.md5(), period_bytes
            while left:
                blk = f.read(min(left, 1 << 24))
                if not blk:
                    break
                hh.update(blk)
                left -= len(blk)
            if left == period_bytes:
                break
            out.append(hh.hexdigest())
    return out


def streams_leg(n_gpus, n_streams=8, gop_pictures=17, repeats=6, wl=None):
    """BASELINE.json configs[4], literally: `n_streams` independent 4K Main random-access streams (different seeds; one closed GOP of `gop_pictures` pictures each,
    sent `repeats` times) decoded by examples/evc_decode --gpus N - the C work queue (include/xevd_wq.h) hands the closed GOPs of all streams to one worker set
    per device - with parsing, batch building, kernels and output inside the timed region (the span app/xevd_app.c:492-501,612-624 times).  The first IDR period of
    every output is compared with the reference decoder's pictures (oracle/_ref/ref_decode_main, one thread), where that was built."""
    import subprocess
    import tempfile
    import torch
    wl = wl or WORKLOADS["cfg3_main_4k_10b_ra"]      # (the GPU suite runs the leg on a small picture size)
    w, h, bd = wl["w"], wl["h"], wl["bd"]
    exe = os.path.join(ROOT, "oracle", "_ref", "ref_decode_main")
    n_dev = min(n_gpus, max(torch.cuda.device_count(), 1))
    quota = host_cpu_quota()
    # workers: ~3/4 of the CPUs the host gives (a worker is a parser thread + a builder thread + a device thread: ~1.4 CPUs at 4K), at least one per device; measured on the
    # 16-CPU boxes, 8 streams on one device: 8 / 12 / 16 workers x 1 tile thread = 304 / 400 / 297 pictures/s, 8 x 2 tile threads 384
    total = max(n_dev, min(16, quota * 3 // 4))
    workers = max(1, total // n_dev)
    tile_threads = max(1, min(16, quota // (workers * n_dev)))
    period_bytes = gop_pictures * (w * h * 3 // 2) * 2
    out = {"workload": "cfg5: %d independent %dx%d %d-bit Main random-access streams (seeds differ), closed GOPs of %d pictures x %d" % (n_streams, w, h, bd, gop_pictures, repeats),
           "devices_asked": n_gpus, "devices_used": n_dev, "host_cpu_quota": quota, "host_hardware_threads": os.cpu_count(),
           "workers_per_device": workers, "tile_threads_per_worker": tile_threads, "build_threads_per_worker": 1}
    with tempfile.TemporaryDirectory() as td:
        args, refs, t_ref = [], {}, time.perf_counter()
        for k in range(n_streams):
            one, data, what = write_bench_stream(wl, gop_pictures, repeats, seed=100 + k)
            open(os.path.join(td, f"s{k}.evc"), "wb").write(data)
            open(os.path.join(td, f"s{k}_one.evc"), "wb").write(one)
            args += [os.path.join(td, f"s{k}.evc"), os.path.join(td, f"o{k}.yuv")]
        out["stream"] = what
        procs = []
        if os.path.exists(exe):       # the reference decoder on one IDR period of every stream, all at once (one thread each)
            for k in range(n_streams):
                procs.append(subprocess.Popen([exe, os.path.join(td, f"s{k}_one.evc"), os.path.join(td, f"r{k}.raw"), str(w), str(h), "1"], stdout=subprocess.DEVNULL, stderr=subprocess.PIPE))
            ref_fps = []
            for k, pr in enumerate(procs):
                err = pr.communicate()[1].decode()
                if pr.returncode == 0:
                    pics, secs = err.split()[-2:]
                    ref_fps.append(int(pics) / float(secs))
                    refs[k] = file_md5s(os.path.join(td, f"r{k}.raw"), period_bytes, 1)
            out["reference_decoder_fps_one_thread_each_all_streams_at_once"] = round(float(np.sum(ref_fps)), 2) if ref_fps else None
        rep = run_evc_decode(["--gpus", str(n_dev), "--workers", str(workers), "--tile-threads", str(tile_threads), "--build-threads", "1", "--keep-units", "1"] + args)
        out["evc_decode"] = rep
        if "error" not in rep:
            out["fps_decode_only"] = rep["fps_decode_only"]
            out["fps_wall_incl_device_startup"] = rep["fps_wall"]
            out["pictures"] = rep["pictures"]
            out["pictures_per_device"] = rep["pictures_per_device"]
            out["host_threads"] = n_dev * workers * (tile_threads + 2)      # per worker: the parser with its tile threads, the builder, the device thread
            out["cpu_seconds"] = round(rep["cpu_user_s"] + rep["cpu_sys_s"], 2)
            if refs:
                ok = [file_md5s(os.path.join(td, f"o{k}.yuv"), period_bytes, 1) == refs.get(k) for k in range(n_streams)]
                out["bit_exact"] = bool(all(ok))
                out["bit_exact_what"] = "the first IDR period of every stream's output == the reference decoder's pictures (single-threaded run)"
    return out


def reference_decoder_leg(wl, budget_s=60.0):
    """Real-bitstream decode, .evc -> .yuv (write_bench_stream): the reference DECODER itself (its public API through oracle/_ref/ref_decode, -m 1 and
    -m 8 threads, entropy decoding included - what a user of xevd_app runs on this host) next to examples/evc_decode (plain C on this repository's
    C ABIs) in four shapes: ONE stream on one worker (parser with its tile threads -> batch builder -> device thread, three pictures in flight), the same
    back to back on one thread, and GOP-parallel through the work queue (2 workers x 8 tile threads, 4 x 4).  12 IDR periods per run; the first two of every
    run are written out and compared byte for byte with the reference decoder's output of that period.  The host's CPU quota (cgroup cpu.max) is in the line."""
    import subprocess
    import tempfile
    main_profile = bool(wl["addb"] or wl["iqt"] or wl["alf"])
    exe = os.path.join(ROOT, "oracle", "_ref", "ref_decode_main" if main_profile else "ref_decode")
    ours = os.path.join(ROOT, "examples", "evc_decode")
    if not os.path.exists(exe) or not os.path.exists(ours):
        return None
    w, h, bd = wl["w"], wl["h"], wl["bd"]
    gop_pictures, repeats = 17, 12
    one, data, what = write_bench_stream(wl, gop_pictures, repeats)
    period_bytes = gop_pictures * (w * h * 3 // 2) * (2 if bd > 8 else 1)
    keep = 2                 # IDR periods of every run that are written out and compared (the others are decoded all the same: --keep-units)
    quota = host_cpu_quota()
    fps, ref_sum, t0, gpu = {}, {}, time.perf_counter(), {}
    with tempfile.TemporaryDirectory() as td:
        p_one, p_all = os.path.join(td, "one.evc"), os.path.join(td, "all.evc")
        open(p_one, "wb").write(one)
        open(p_all, "wb").write(data)
        ok = True
        shapes = (("one_stream_pipelined", ["--workers", "1", "--tile-threads", str(min(16, quota)), "--build-threads", "8", "--builders", "2"]),
                  ("one_stream_back_to_back", ["--workers", "1", "--tile-threads", str(min(16, quota)), "--build-threads", "8", "--no-pipeline"]),
                  ("gop_parallel_2x8", ["--workers", "2", "--tile-threads", str(max(1, min(8, quota // 2))), "--build-threads", "4"]),
                  ("gop_parallel_4x4", ["--workers", "4", "--tile-threads", str(max(1, min(4, quota // 4))), "--build-threads", "2"]))
        for name, args in shapes:
            dst = os.path.join(td, "ours.yuv")
            rep = run_evc_decode(args + ["--keep-units", str(keep), "--hash-units", p_all, dst])
            if "error" in rep:
                gpu[name] = rep
                ok = False
                continue
            gpu[name] = {"decode_only_fps": rep["fps_decode_only"], "parse_ms_per_picture": rep["parse_ms_per_picture"], "batch_build_ms_per_picture": rep["build_ms_per_picture"],
                         "host_threads": rep["workers_per_device"] * (rep["tile_threads"] + rep["build_threads"] + 1), "cpu_seconds_per_picture": round((rep["cpu_user_s"] + rep["cpu_sys_s"]) / max(rep["pictures"], 1), 4)}
            # every IDR period of the run repeats the same GOP: all their hashes must equal the first period's, whose pictures are compared with the reference decoder's below
            hashes = (rep.get("unit_hashes") or [[]])[0]
            gpu[name]["all_periods_equal_the_first"] = len(hashes) == repeats and len(set(hashes)) == 1
            if bd > 8:          # 16-bit samples like the reference driver's output
                gpu[name]["periods"] = file_md5s(dst, period_bytes, keep)
        for threads in (1, 8):
            if time.perf_counter() - t0 > budget_s and fps:
                break
            dst = os.path.join(td, "ref.raw")
            r = subprocess.run([exe, p_one, dst, str(w), str(h), str(threads)], stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, timeout=600)
            if r.returncode != 0:
                return {"error": r.stderr.decode()[-200:]}
            pics, secs = r.stderr.decode().split()[-2:]
            fps[str(threads)] = round(int(pics) / float(secs), 2)
            if bd > 8:
                ref_sum[threads] = file_md5s(dst, period_bytes)
        # The same stream shape with tool_dmvr + tool_hmvp + tool_mmvd: the parser needs every reference picture's decoded luma before it parses the next picture (a blocking
        # 66 MB download per 8K reference picture in the decode loop) and runs the refinement search itself - what that configuration costs against the plain stream
        refine = None
        if wl.get("admvp") and wl.get("addb") and not os.environ.get("XEVD_BENCH_NO_REFINE_LEG"):
            try:
                r_one, r_all, r_what = write_bench_stream(wl, gop_pictures, 4, refine_tools=True)
                p_r1, p_ra = os.path.join(td, "r_one.evc"), os.path.join(td, "r_all.evc")
                open(p_r1, "wb").write(r_one)
                open(p_ra, "wb").write(r_all)
                dst = os.path.join(td, "r_ours.yuv")
                rep = run_evc_decode(shapes[0][1] + ["--keep-units", "1", "--hash-units", p_ra, dst])
                if "error" in rep:
                    refine = rep
                else:
                    hashes = (rep.get("unit_hashes") or [[]])[0]
                    refine = {"decode_only_fps": rep["fps_decode_only"], "parse_ms_per_picture": rep["parse_ms_per_picture"], "batch_build_ms_per_picture": rep["build_ms_per_picture"],
                              "all_periods_equal_the_first": len(hashes) == 4 and len(set(hashes)) == 1, "stream": r_what,
                              "ratio_to_the_plain_stream": round(rep["fps_decode_only"] / max(gpu.get("one_stream_pipelined", {}).get("decode_only_fps", 0.0), 1e-9), 3)}
                    if bd > 8:
                        ours = file_md5s(dst, period_bytes, 1)
                        dst_r = os.path.join(td, "r_ref.raw")
                        rr = subprocess.run([exe, p_r1, dst_r, str(w), str(h), "1"], stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, timeout=600)
                        refine["bit_exact"] = rr.returncode == 0 and file_md5s(dst_r, period_bytes) == ours and refine["all_periods_equal_the_first"]
            except Exception as e:
                refine = {"error": repr(e)[:300]}
    # the yardstick is the reference decoder with ONE thread; whether its own threaded run agrees with it is reported, not required (it does not on every
    # tiled stream: DESIGN 5b)
    bit_exact = None
    if bd > 8 and ok and 1 in ref_sum and len(ref_sum[1]) == 1:
        for g in gpu.values():
            g["bit_exact"] = len(g.get("periods", [])) == keep and all(m == ref_sum[1][0] for m in g["periods"]) and bool(g.get("all_periods_equal_the_first"))
        bit_exact = all(g["bit_exact"] for g in gpu.values())
    for g in gpu.values():
        g.pop("periods", None)
    return {"frames_per_s_by_threads": fps, "host_cores": os.cpu_count(), "host_cpu_quota": quota, "evc_decode_on_gpu": gpu, "bit_exact": bit_exact,
            "evc_decode_dmvr_hmvp_mmvd": refine,
            "reference_threads_8_equals_1": (ref_sum[8] == ref_sum[1]) if (1 in ref_sum and 8 in ref_sum) else None, "stream": what,
            "pictures": {"evc_decode": gop_pictures * repeats, "reference_decoder": gop_pictures, "compared_idr_periods_per_run": keep},
            "what": "xevd_create / xevd_decode / xevd_pull of the reference library built in oracle/_ref (entropy decoding + reconstruction), threads = "
                    "XEVD_CDSC.threads, on one IDR period; evc_decode_on_gpu: examples/evc_decode on the whole stream, decode-only rate of the slowest worker "
                    "(parsing + batch build + kernels + output, the span the reference application times, app/xevd_app.c:492-501,612-624); bit_exact: every IDR "
                    "period of every evc_decode run == the reference decoder's pictures (its single-threaded run)"}


def contexts_leg(XgpuDecoder, device, wl, first, batches, alf, steps, warmup, n_ctx=2):
    """What the GPU does when it is fed by SEVERAL decoders at once (the streams of configs[4] on one device; the pictures of one temporal layer of a random-access
    stream): n_ctx contexts on one device, each with its own stream, picture slots and resident batches, cycling them from its own host thread - the resident-batch
    rate of `value` with the idle stretches of one picture (the dependency chain of its intra CUs, the gaps between its kernels) filled by another.  A secondary
    figure: `value` stays one context."""
    import threading
    two_lists = wl["n_refs"][1] > 0
    ctxs = []
    for i in range(n_ctx):
        d = XgpuDecoder(wl["w"], wl["h"], wl["bd"], device=device, iqt=wl["iqt"], admvp=wl["admvp"], addb=wl["addb"], alf=wl["alf"], max_pics=4)
        sl = [d.pic_alloc(), d.pic_alloc(), d.pic_alloc()]
        for k in range(2):
            d.pic_upload(sl[k], first[k])
            d.frame_begin(sl[k], k - 1, {})
            d.pad()
            d.frame_end()
        hs = [d.batch_create(b) for b in batches]
        d.sync()
        ctxs.append((d, sl, hs))

    def run(d, sl, hs, k0, n):
        for k in range(k0, k0 + n):
            refs = {(0, 0): (sl[(k + 1) % 3], k)}
            if two_lists:
                refs[(0, 1)] = (sl[k % 3], k + 2 if wl.get("dmvr_frac") else k - 1)
            d.decode_picture(sl[(k + 2) % 3], k + 1, refs, hs[k % len(hs)], alf=alf, next_batch=hs[(k + 1) % len(hs)] if len(hs) > 1 else None)
        d.sync()

    def all_of(k0, n):
        th = [threading.Thread(target=run, args=(d, sl, hs, k0, n)) for d, sl, hs in ctxs]
        t0 = time.perf_counter()
        for t in th:
            t.start()
        for t in th:
            t.join()
        return time.perf_counter() - t0
    all_of(0, warmup)
    dt = all_of(warmup, steps)
    for d, sl, hs in ctxs:
        for h in hs:
            d.batch_destroy(h)
        d.close()
    return {"contexts": n_ctx, "fps": round(n_ctx * steps / dt, 2), "ms_per_picture": round(dt * 1e3 / (n_ctx * steps), 4),
            "what": "resident batches as in `value`, but from %d decoder contexts at once on this device (own HIP streams, own host threads): the GPU's rate when independent pictures overlap" % n_ctx}


def end_to_end_leg(dec, wl, batches, alf, slots, steps, warmup, builders=int(os.environ.get('XEVD_BENCH_BUILDERS', '4')), depth=int(os.environ.get('XEVD_BENCH_DEPTH', '5'))):
    """CU batches in host memory -> packed YUV pictures in host memory, everything inside the timed region: the host batch builder
    (xgpu_batch_create on `builders` threads, `depth` pictures ahead), the upload of every batch (coefficient arena straight from pinned
    memory, the rest through pinned staging) on the upload stream, the kernels, conversion + packing and the download of every picture on
    the download stream (app/xevd_app.c:492-501 times the same span: bitstream-side input ready -> picture written out).  What stays
    outside: the numpy -> ctypes marshalling of the four distinct batches (a C caller has none) and entropy decoding (this path starts
    after it)."""
    import collections
    from concurrent.futures import ThreadPoolExecutor
    from xevd_amd import abi
    prepared = []
    for b in batches:
        arena = dec.host_alloc(max(b["coef"].nbytes, 16), np.int16)      # the parser's pinned coefficient arena
        arena[:len(b["coef"])] = b["coef"]
        bb = dict(b)
        bb["coef"] = arena[:len(b["coef"])]
        prepared.append(abi.make_cu_batch(bb))
    dec.lib.xgpu_set_builder_threads(dec.ctx, int(os.environ.get("XEVD_BENCH_BUILD_THREADS", "4")))      # every xgpu_batch_create spreads its per-CU passes
    out_size = dec.lib.xgpu_pic_output_size(dec.ctx, wl["bd"], 0, 0, 0, 0)
    outs = [dec.host_alloc(out_size) for _ in range(2)]
    two_lists = wl["n_refs"][1] > 0
    total = warmup + steps
    pool = ThreadPoolExecutor(builders)
    build_s = []

    def build(k):
        t = time.perf_counter()
        h = dec.batch_create_from_struct(prepared[k % len(prepared)][0])
        build_s.append(time.perf_counter() - t)
        return h
    futs = collections.deque(pool.submit(build, k) for k in range(min(depth, total)))
    prev, t0 = None, None
    for k in range(total):
        if k == warmup:
            dec.sync()
            t0 = time.perf_counter()
        hb = futs.popleft().result()
        if k + depth < total:
            futs.append(pool.submit(build, k + depth))
        cur, ref0, ref1 = slots[(k + 2) % 3], slots[(k + 1) % 3], slots[k % 3]
        refs = {(0, 0): (ref0, k)}
        if two_lists:
            refs[(0, 1)] = (ref1, k - 1)
        dec.decode_picture(cur, k + 1, refs, hb, alf=alf)
        ticket = dec.pic_output_async(cur, outs[k & 1], wl["bd"])
        if prev is not None:
            dec.pic_output_wait(prev[0])
            dec.batch_destroy(prev[1])
        prev = (ticket, hb)
    dec.pic_output_wait(prev[0])
    dec.batch_destroy(prev[1])
    dt = time.perf_counter() - t0
    pool.shutdown()
    # the stages alone: one picture's download repeated, and the builder's own time per picture
    t1 = time.perf_counter()
    for _ in range(5):
        dec.pic_output_wait(dec.pic_output_async(slots[0], outs[0], wl["bd"]))
    d2h_ms = (time.perf_counter() - t1) / 5 * 1e3
    h2d = float(np.mean([b["coef"].nbytes for b in batches]))
    return {"fps": round(steps / dt, 2), "ms_per_picture": round(1e3 * dt / steps, 3), "steps": steps,
            "builder_threads": builders, "threads_per_builder": int(os.environ.get("XEVD_BENCH_BUILD_THREADS", "4")), "pictures_in_flight": depth,
            "stage_ms": {"batch_build_per_thread": round(1e3 * float(np.mean(build_s)), 3), "output_kernel_plus_d2h_alone": round(d2h_ms, 3)},
            "h2d_coef_bytes_per_picture": int(h2d), "d2h_bytes_per_picture": int(out_size),
            "what": "host CU batches -> xgpu_batch_create (builder threads, pinned coefficient arena) -> upload stream -> kernels -> "
                    "xgpu_pic_output_async (conversion, packing, download stream) -> host YUV; all of it inside the timed region"}


def spawn_ranks(n):
    """`python bench.py --gpus N` without a launcher: N child processes of this command line, RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set
    (127.0.0.1 and a free port), stdout / stderr shared with the parent; returns the worst exit status."""
    import socket
    import subprocess
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env))
    rc = 0
    for p in procs:
        p.wait()
        rc = rc or p.returncode
    return rc


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--workload", default=None, choices=sorted(WORKLOADS),
                    help=f"default: {DEFAULT_WORKLOAD} on one GPU, {DEFAULT_WORKLOAD_MULTI} with --gpus N > 1")
    ap.add_argument("--batches", type=int, default=4, help="distinct pictures' CU batches kept resident and cycled")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-end-to-end", action="store_true", help="skip the batches-in -> YUV-out leg")
    ap.add_argument("--no-inter-classes", action="store_true", help="skip the k_inter-by-CU-size leg")
    ap.add_argument("--no-prepare", action="store_true", help="every picture's residual pass in front of its own k_inter (no xgpu_batch_prepare of the next picture)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # launched plainly (`python bench.py --gpus N`): this process becomes the launcher of N ranks, one per GPU, each running this
        # file with the rendezvous variables torch.distributed.run would have set; rank 0 prints the JSON line
        sys.exit(spawn_ranks(args.gpus))

    import torch
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    dist = None
    if args.workload is None:
        args.workload = DEFAULT_WORKLOAD_MULTI if world > 1 else DEFAULT_WORKLOAD
    if world > 1:
        # the ranks exchange no picture data (streams / GOPs are independent): the process group only carries the job counter of
        # the work queue (rendezvous store), the barriers and the final accounting - gloo on the host, no RCCL communicator is created
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # gloo reports its connections on the C-level stdout ("[Gloo] Rank 0 is connected to ..."): stdout carries the ONE JSON line and nothing else, so the
        # descriptor points at stderr while the process group comes up
        sys.stdout.flush()
        saved_stdout = os.dup(1)
        os.dup2(2, 1)
        try:
            dist.init_process_group("gloo")
            dist.barrier()
        finally:
            sys.stdout.flush()
            os.dup2(saved_stdout, 1)
            os.close(saved_stdout)
    local_rank %= max(torch.cuda.device_count(), 1)       # more ranks than devices (a one-GPU box exercising the N > 1 path): they share
    # XEVD_BENCH_DECODER names another module with an XgpuDecoder class: tests/stub_decoder.py, with which the CPU suite runs the launcher,
    # the work queue and the accounting of the N > 1 path (the line then says "decoder": "stub_decoder"; no measurement comes out of it)
    dec_mod = os.environ.get("XEVD_BENCH_DECODER", "xevd_amd.decoder")
    if dec_mod == "xevd_amd.decoder":
        torch.cuda.set_device(local_rank)

    import importlib
    XgpuDecoder = importlib.import_module(dec_mod).XgpuDecoder
    wl = WORKLOADS[args.workload]
    first, batches, alf = make_stream(wl, 1000 + rank, args.batches)
    dec = XgpuDecoder(wl["w"], wl["h"], wl["bd"], device=local_rank, iqt=wl["iqt"], admvp=wl["admvp"], addb=wl["addb"],
                      alf=wl["alf"], max_pics=4)
    slots = [dec.pic_alloc(), dec.pic_alloc(), dec.pic_alloc()]
    for i in range(2):
        dec.pic_upload(slots[i], first[i])
        dec.frame_begin(slots[i], i - 1, {})
        dec.pad()
        dec.frame_end()
    # host batch builder + upload per picture, steady state: the first round fills the context's buffer pool, the second one is timed
    handles = [dec.batch_create(b) for b in batches]
    dec.sync()
    for hb in handles:
        dec.batch_destroy(hb)
    t_up = time.perf_counter()
    handles = [dec.batch_create(b) for b in batches]
    dec.sync()
    t_up = (time.perf_counter() - t_up) / len(batches)
    batch_info = dec.batch_info(handles[0])

    two_lists = wl["n_refs"][1] > 0

    # self-check, outside the timed region: the first batch decoded from the two start pictures exactly as the CPU leg does it
    # (src_main/xevdm.c:3136-3219: recon, deblock, ALF, pad); compared plane by plane, padding included, further down
    gpu_check = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        refs = {(0, 0): (slots[0], 0)}
        if two_lists:
            refs[(0, 1)] = (slots[1], 2 if wl.get("dmvr_frac") else -1)
        dec.decode_picture(slots[2], 1, refs, handles[0], alf=alf)
        gpu_check = dec.pic_download_padded(slots[2])

    def step(k, ahead=False):
        # picture k+1 is predicted from picture k (list 0) and, with two lists, picture k-1 (list 1): a 3-slot DPB ring
        cur, ref0, ref1 = slots[(k + 2) % 3], slots[(k + 1) % 3], slots[k % 3]
        refs = {(0, 0): (ref0, k)}
        if two_lists:
            refs[(0, 1)] = (ref1, k + 2 if wl.get("dmvr_frac") else k - 1)      # DMVR: a B picture between its two references
        # ahead: the NEXT picture's residual pass (it depends on nothing but its batch) is queued behind this picture's k_inter (xgpu_batch_prepare) - what a
        # decoder does that has parsed picture k + 1 while picture k is being reconstructed; needs two resident batches
        nxt = handles[(k + 1) % len(handles)] if ahead and len(handles) > 1 else None
        dec.decode_picture(cur, k + 1, refs, handles[k % len(handles)], alf=alf, next_batch=nxt)

    def barrier():
        if dist is not None:
            dist.barrier()
        if torch.cuda.is_available():
            torch.cuda.synchronize()
        dec.sync()

    ahead = not args.no_prepare
    for k in range(args.warmup):
        step(k, ahead)
    barrier()
    per_rank = None
    t0 = time.perf_counter()
    if dist is None:
        for k in range(args.steps):
            step(args.warmup + k, ahead)
        barrier()
        dt = time.perf_counter() - t0
    else:
        # world x steps pictures in jobs of GOP_PICTURES, drawn by the ranks from one queue (xevd_amd/workqueue.py: a counter in the
        # rendezvous store): a rank that is done asks for the next job, nobody waits for a fixed share
        from xevd_amd import workqueue
        import torch.distributed.distributed_c10d as c10d
        n_jobs = (world * args.steps + GOP_PICTURES - 1) // GOP_PICTURES
        q = workqueue.TicketQueue(c10d._get_default_store(), n_jobs, key="xevd_amd/bench_jobs")
        mine, k = 0, args.warmup
        while True:
            j = q.next()
            if j is None:
                break
            n = min(GOP_PICTURES, world * args.steps - j * GOP_PICTURES)
            for _ in range(n):
                step(k, ahead)
                k += 1
            mine += n
        if torch.cuda.is_available():
            torch.cuda.synchronize()
        dec.sync()
        my_dt = time.perf_counter() - t0
        barrier()
        t = torch.tensor([my_dt], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
        g = [torch.zeros(2, dtype=torch.float64) for _ in range(world)]
        dist.all_gather(g, torch.tensor([float(mine), my_dt], dtype=torch.float64))
        per_rank = [{"rank": r, "pictures": int(v[0].item()), "fps": round(float(v[0].item()) / max(float(v[1].item()), 1e-9), 2)} for r, v in enumerate(g)]

    # per-kernel durations with HIP events on the stream the kernels are launched on, same workload and steps - every kernel on the main stream, one after
    # the other (no residual pass queued ahead), so that each duration is the kernel's own
    dec.sync()
    dec.timing_enable(True)
    dec.timing_reset()
    for k in range(args.steps):
        step(args.warmup + args.steps + k)
    tim = dec.timing_get()
    dec.timing_enable(False)

    # batches-in -> YUV-out rate of the same workload, transfers and the host batch builder inside the timed region
    for hb in handles:
        dec.batch_destroy(hb)
    handles = []
    if args.no_end_to_end:
        e2e = {"fps": None, "ms_per_picture": 0.0, "skipped": True}
    else:
        e2e = end_to_end_leg(dec, wl, batches, alf, slots, E2E_PICTURES, E2E_WARMUP)
    inter_classes = None
    if world == 1 and rank == 0 and not args.no_inter_classes and not args.no_end_to_end and dec_mod == "xevd_amd.decoder" and wl["w"] >= 3840:
        try:
            inter_classes = inter_by_cu_size_leg(dec, wl, slots, alf)
        except Exception as e:
            inter_classes = {"error": repr(e)[:200]}
    two_ctx = None
    if world == 1 and not args.no_end_to_end and dec_mod == "xevd_amd.decoder":
        try:
            two_ctx = contexts_leg(XgpuDecoder, local_rank, wl, first, batches, alf, CTX_PICTURES, CTX_WARMUP)
        except Exception as e:      # (a secondary figure must not cost the line)
            two_ctx = {"error": str(e)[:200]}
    if dist is not None and not args.no_end_to_end:
        t = torch.tensor([e2e["ms_per_picture"]], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        e2e["fps"] = round(world * 1e3 / float(t.item()), 2)

    # BASELINE configs[4] literally (N > 1): independent 4K streams -> examples/evc_decode --gpus N, one process with a worker set per device and the C work queue
    # between them.  Rank 0 runs it (the other ranks hold no stream work and wait at the barrier); reported next to `value`, not instead of it.
    streams = None
    if world > 1:
        barrier()
        if rank == 0 and dec_mod == "xevd_amd.decoder":
            try:
                streams = streams_leg(world)
            except Exception as e:
                streams = {"error": repr(e)[:300]}
        barrier()

    if rank == 0:
        ab = [algorithmic_bytes(b, wl["w"], wl["h"], bool(wl["addb"])) for b in batches]
        # ADDB directly followed by ALF runs as ONE kernel (k_addb_alf, timed as "alf"): it is credited with the one read + one write of the picture it has to
        # make (4 B/sample), not with the deblocked picture's round trip through memory that it removes
        addb_alf = bool(wl["addb"] and wl["alf"] and tim["dbk_v"][1] == 0)
        s_pic = wl["w"] * wl["h"] * 3 // 2
        survey_extra = (8 if addb_alf else 4) * s_pic if wl["addb"] else 0      # two deblocking passes and ALF, 4 B/sample each
        if addb_alf:
            for a in ab:
                a["dbk_v"] = 0
        kernels = {}
        for name in ("itdq", "inter", "dmvr", "affine", "intra", "dbk_v", "dbk_h", "alf", "pad"):
            ms, n = tim[name]
            if n:
                kernels["addb_alf" if name == "alf" and addb_alf else name] = {"avg_us": round(1e3 * ms / n, 2), "launches": int(n)}
        # the dominant kernel by time; k_intra is a dependency-chain (latency) kernel in two launches and is reported in
        # `kernels` but not priced against the HBM roofline
        dom = max(("itdq", "inter", "dbk_v", "dbk_h", "alf"), key=lambda k: tim[k][0])
        bytes_per_launch = float(np.mean([a[dom] for a in ab]))
        avg_s = max(tim[dom][0] / max(tim[dom][1], 1) * 1e-3, 1e-12)
        achieved = bytes_per_launch / avg_s / 1e9
        try:
            copy_bw = dec.measure_copy_bw(1 << 30, 10)
        except Exception:
            copy_bw = None
        traffic, pmc_commit, traffic_note = None, None, None
        try:
            pmc = json.load(open(os.path.join(ROOT, "profiles", "latest_pmc.json")))
            if pmc["workload"] == args.workload:
                pmc_commit = pmc.get("commit")
                # the counter passes are a committed record, not part of this run: they only describe this run's kernel while its sources are the ones measured
                if pmc.get("sources_sha256", {}).get(dom) == kernel_sources_sha256(dom):
                    traffic = pmc["kernels"][dom]["traffic_bytes"]       # from the committed rocprofv3 --pmc passes
                else:
                    traffic_note = f"dropped: the sources of the {dom} kernel changed since the counter passes of commit {pmc_commit}"
        except Exception:
            traffic = None
        total_alg = float(np.mean([sum(v for k, v in a.items() if k != "alf" or wl["alf"]) for a in ab]))
        kern_s = max(sum(tim[k][0] for k in ("itdq", "inter", "dmvr", "affine", "intra", "dbk_v", "dbk_h", "alf", "pad")) * 1e-3 / args.steps, 1e-12)
        out = {
            "metric": "frames/sec (bit-exact YUV) + achieved HBM GB/s",
            "value": round(world * args.steps / dt, 2),
            "unit": "frames/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(1e3 * dt / args.steps, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "s16", "data": "synthetic",
            "config": {"workload": args.workload, "width": wl["w"], "height": wl["h"], "bit_depth": wl["bd"],
                       "profile": "Main (admvp 8-tap MC, IQT, ADDB, ALF on every CTU)" if wl["addb"] else "Baseline",
                       "stream": ("2 reference lists, 50% bi-predicted CUs" if two_lists else "IPPP, 1 reference")
                                 + ", 90% inter / 10% intra CUs (5 Baseline modes), 60% coded, deblock on, quad-tree 64..4",
                       "batches_resident": len(batches), "batch": batch_info,
                       "residual_pass_ahead": bool(ahead and len(batches) > 1),
                       "parallelism": (f"{world} ranks, one GPU each, drawing jobs of {GOP_PICTURES} pictures (closed GOPs of independent streams) from one host work "
                                       "queue; no collective on the data path" if world > 1 else "1 stream on 1 GPU")},
            "roofline": {"bound": "hbm", "kernel": "addb_alf" if dom == "alf" and addb_alf else dom, "achieved": round(achieved, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBPS, 4), "traffic": traffic,
                         "algorithmic_bytes_per_launch": int(bytes_per_launch), "avg_launch_us": round(avg_s * 1e6, 2),
                         "measured_copy_bw_gbps": None if copy_bw is None else round(copy_bw, 1),
                         "frac_of_measured_copy_bw": None if not copy_bw else round(achieved / copy_bw, 4),
                         # the guide's measured float4-copy figure for this part (MI355X_MICROARCH.md: 6.29 TB/s); k_copy (one 16-byte element per lane) reaches 6.2
                         "frac_of_guide_copy_bw_6290": round(achieved / 6290.0, 4),
                         "traffic_source": (f"profiles/latest_pmc.json (rocprofv3 --pmc passes of this workload at commit {pmc_commit}, committed; not measured by this run: "
                                            "counter passes cannot run inside a timed benchmark)") if traffic is not None else traffic_note},
            "kernels": kernels,
            "inter_by_cu_size": inter_classes,
            "whole_frame": {"algorithmic_bytes": int(total_alg), "kernel_us": round(kern_s * 1e6, 2),
                            "achieved_gbps": round(total_alg / kern_s / 1e9, 1),
                            # the same kernel time against SURVEY 8(d)'s own accounting (deblocking as two passes of 4 B/sample), for comparison across rounds
                            "algorithmic_bytes_survey_two_pass_deblock": int(total_alg + survey_extra),
                            "achieved_gbps_survey_accounting": round((total_alg + survey_extra) / kern_s / 1e9, 1)},
            # `value` above is the rate with the CU batches resident in HBM (the benchmark contract's definition); the rate of the whole
            # span host batches -> host YUV, transfers and the host batch builder inside the timed region, is end_to_end_fps
            "per_rank": per_rank,
            "streams": streams,
            "decoder": dec_mod,
            "kernel_only_fps": round(world * args.steps / dt, 2),
            "end_to_end_fps": e2e["fps"],
            "end_to_end": e2e,
            "two_contexts": two_ctx,
        }
        if not args.no_cpu_baseline and world == 1:      # the CPU leg runs on rank 0 of the single-GPU run only
            cpu_pic, out["cpu_baseline"] = cpu_baseline(wl, first, batches[0], alf)
            out["bit_exact"] = bool(all(np.array_equal(gpu_check[c], cpu_pic[c]) for c in range(3)))
            out["bit_exact_what"] = (f"picture 1 of the benchmarked stream ({wl['w']}x{wl['h']}, padded planes) on the GPU == the same batch through "
                                     + ("the reference's functions (oracle/_ref)" if out["cpu_baseline"]["kind"] == "reference" else "the CPU oracle"))
            try:
                rd = reference_decoder_leg(wl)
            except Exception as e:                      # the checker's leg must not take the measurement down
                rd = {"error": repr(e)[:200]}
            if rd is not None:
                out["cpu_baseline"]["reference_decoder"] = rd
        print(json.dumps(out))
    barrier()
    dec.close()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
