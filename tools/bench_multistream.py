#!/usr/bin/env python3
"""Many real bitstreams on ONE GPU, through the C ABIs only: examples/evc_decode (plain C: parser thread + device thread per worker, the GOP work queue of
include/xevd_wq.h) decodes K input files on K workers that share the device - K parser threads, K xgpu contexts, K HIP streams.  This is the deployment
shape for stream decoding: one stream's entropy decoding is a serial chain on one host core and a 1080p picture is 0.12 ms of kernels, so a single stream
leaves the MI355X idle more than 95 % of the time; pictures of independent streams fill it (their kernels run concurrently from different HIP streams).
Streams are written by this repository's front end: --profile base = Baseline 1080p 8-bit IPPP (BASELINE.json configs[1] shape), main = random-access Main
(the shape of bench.py's real-bitstream leg, one tile).  No Python in the decode loop.  Prints one JSON line.
usage: python tools/bench_multistream.py [--streams 1,4,16] [--pics 48] [--width 1920 --height 1080] [--profile base|main]"""
import argparse
import json
import os
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from xevd_amd import stream, synth                     # noqa: E402


def write_stream(w, h, n, seed, main):
    rng = np.random.default_rng(seed)
    tids = [0, 1, 2, 2, 3, 3, 3, 3]
    wr = stream.StreamWriter(w, h, 10 if main else 8, 2 if main else 1, main=main, iqt=main, addb=main, alf=main, admvp=main, log2_sub_gop=3 if main else 0)
    try:
        if main:
            wr.add_alf_aps(0, luma=rng.integers(-12, 13, (5, 12)), chroma=rng.integers(-10, 11, 6), type7=True, delta_idx=rng.integers(0, 5, 25))
        for k in range(n):
            idr = k % 24 == 0 if not main else k % 25 == 0          # closed GOPs: one job of the queue each
            j = k % 25
            tid = 0 if idr or not main else tids[(j - 1) % 8]
            is_b = main and not idr and tid > 0
            b = synth.gen_frame(rng, w, h, 10 if main else 8, inter_frac=0.0 if idr else 0.9, n_refs=(2 if main else 1, 2 if is_b else 0), bi_frac=0.5 if is_b else 0.0,
                                coded_frac=0.6, max_level=6, amp=1.0)
            if main:
                wr.set_slice_alf(True, 0, 0, chroma_idc=3)
            wr.add_picture(b, stream.SLICE_I if idr else (stream.SLICE_B if is_b else stream.SLICE_P), slice_qp=30, idr=idr, temporal_id=tid)
        return wr.bytes()
    finally:
        wr.close()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--streams", default="1,4,16")
    ap.add_argument("--pics", type=int, default=48)
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--profile", default="base", choices=["base", "main"])
    ap.add_argument("--extra", default="", help="more evc_decode options, e.g. \"--builders 1\"")
    ap.add_argument("--processes", action="store_true", help="one evc_decode PROCESS per stream instead of one worker thread per stream in one process (own HIP runtime each)")
    args = ap.parse_args()
    exe = os.path.join(ROOT, "examples", "evc_decode")
    is_main = args.profile == "main"
    pics = args.pics if not is_main else (args.pics + 24) // 25 * 25
    data = [write_stream(args.width, args.height, pics, 100 + i, is_main) for i in range(4)]      # 4 distinct streams, reused round-robin
    out = {"stream": f"{'Main random-access (hierarchical B, admvp, IQT, ADDB, ALF) 10-bit' if is_main else 'Baseline 8-bit IPPP'} {args.width}x{args.height}, {pics} pictures per stream, "
                     f"{len(data[0])} bytes", "host_cores": os.cpu_count(), "decoder": "examples/evc_decode (C): one worker = parser thread + device thread + xgpu context per stream" + (", one PROCESS per stream" if args.processes else ", all workers in one process"),
           "pictures_per_s": {}}
    with tempfile.TemporaryDirectory() as td:
        for i, d in enumerate(data):
            open(os.path.join(td, f"s{i}.evc"), "wb").write(d)
        for k in [int(v) for v in args.streams.split(",")]:
            if args.processes:
                import time
                t0 = time.perf_counter()
                ps = [subprocess.Popen([exe, "--workers", "1", os.path.join(td, f"s{i % len(data)}.evc"), os.path.join(td, f"o{i}.yuv")], stderr=subprocess.PIPE) for i in range(k)]
                errs = [p.communicate()[1].decode() for p in ps]
                wall = time.perf_counter() - t0
                if any(p.returncode != 0 for p in ps):
                    out["pictures_per_s"][str(k)] = "error: " + errs[0][-200:]
                    continue
                # every process reports its own decode-only span; the aggregate is K streams over the slowest of them (they run side by side)
                spans = [float(e.split("slowest worker)")[1].split("s,")[0]) for e in errs]
                out["pictures_per_s"][str(k)] = {"decode_only": round(k * pics / max(spans), 2), "wall_incl_start_up": round(k * pics / wall, 2)}
                for i in range(k):
                    os.remove(os.path.join(td, f"o{i}.yuv"))
                continue
            cmd = [exe, "--workers", str(k)] + args.extra.split()
            for i in range(k):
                cmd += [os.path.join(td, f"s{i % len(data)}.evc"), os.path.join(td, f"o{i}.yuv")]
            r = subprocess.run(cmd, stderr=subprocess.PIPE, timeout=900)
            txt = r.stderr.decode()
            if r.returncode != 0:
                out["pictures_per_s"][str(k)] = "error: " + txt[-200:]
                continue
            out["pictures_per_s"][str(k)] = {"decode_only": float(txt.split("slowest worker)")[1].split("s,")[1].split("pictures/s")[0]),
                                             "wall_incl_start_up": float(txt.split("s wall (device start-up included),")[1].split("pictures/s")[0])}
            for i in range(k):
                os.remove(os.path.join(td, f"o{i}.yuv"))
    print(json.dumps(out))


if __name__ == "__main__":
    main()
