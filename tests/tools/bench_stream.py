#!/usr/bin/env python3
"""(test-side tool: it runs the reference decoder from oracle/_ref, which only tests/ may touch)
End-to-end numbers on a REAL bitstream (BASELINE.json configs[1] shape: Baseline 1080p 8-bit IPPP): our writer makes the
stream, then (a) the host parser alone, (b) parser -> batch upload -> HIP kernels, sequentially, one picture at a time
(nothing overlapped yet), (c) the reference decoder itself (oracle/_ref/ref_decode, its public API, -m N threads) on the same
bytes.  Prints one JSON line.  usage: python tests/tools/bench_stream.py [--pics 30] [--width 1920 --height 1080]"""
import argparse
import json
import os
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--pics", type=int, default=30)
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--no-gpu", action="store_true")
    args = ap.parse_args()
    import stream_util as su
    from xevd_amd import stream
    w, h, n = args.width, args.height, args.pics
    t0 = time.perf_counter()
    data = su.make_stream(w, h, n, seed=2, max_refs=1, inter_frac=0.9, skip_frac=0.15)
    t_write = time.perf_counter() - t0
    out = {"stream": f"Baseline {w}x{h} 8-bit IPPP, {n} pictures, {len(data)} bytes ({8 * len(data) / n / 1e6:.2f} Mbit/picture), "
                     "90% inter (15% of them skip) / 10% intra CUs, 60% coded, deblock on, cu_qp_delta on",
           "writer_s": round(t_write, 2)}
    t0 = time.perf_counter()
    pics = stream.parse_stream(data)
    out["parser_fps"] = round(n / (time.perf_counter() - t0), 1)
    if not args.no_gpu:
        from xevd_amd.decoder import XgpuDecoder
        with XgpuDecoder(w, h, 8, max_pics=4) as dec:
            slots = [dec.pic_alloc(), dec.pic_alloc()]
            for rep in range(2):              # second pass is the timed one
                t0 = time.perf_counter()
                for k, p in enumerate(pics):
                    hb = dec.batch_create(p["batch"])
                    refs = {(0, 0): (slots[(k + 1) & 1], p["poc"] - 1)} if p["refs"][0] else {}
                    dec.decode_picture(slots[k & 1], p["poc"], refs, hb, deblock=p["deblock_on"], pad=True)
                    dec.sync()
                    dec.batch_destroy(hb)
                t_gpu = time.perf_counter() - t0
        out["batches_to_gpu_fps"] = round(n / t_gpu, 1)
        out["parse_plus_gpu_sequential_fps"] = round(n / (t_gpu + n / out["parser_fps"]), 1)
        from xevd_amd.player import StreamDecoder
        for rep in range(2):
            t0 = time.perf_counter()
            cnt = sum(1 for _ in StreamDecoder(data).pictures(download=False))
            t_pipe = time.perf_counter() - t0
        out["pipelined_fps"] = round(cnt / t_pipe, 1)      # parser thread one picture ahead of the GPU loop (xevd_amd/player.py)
    ref = os.path.join(ROOT, "oracle", "_ref", "ref_decode")
    if os.path.exists(ref):
        with tempfile.TemporaryDirectory() as td:
            f = os.path.join(td, "s.evc")
            open(f, "wb").write(data)
            out["reference_decoder_fps"] = {}
            for threads in (1, 2, 4, 8):
                r = subprocess.run([ref, f, "-", str(w), str(h), str(threads), "2"], stdout=subprocess.DEVNULL, stderr=subprocess.PIPE)
                if r.returncode == 0:
                    cnt, secs = r.stderr.decode().split()[-2:]
                    out["reference_decoder_fps"][str(threads)] = round(int(cnt) / float(secs), 1)
            out["host_cores"] = os.cpu_count()
    print(json.dumps(out))


if __name__ == "__main__":
    main()
