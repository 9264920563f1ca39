#!/usr/bin/env python3
"""(test-side tool: it runs the reference decoder from oracle/_ref, which only tests/ may touch)
Real-bitstream decode rates at 1080p / 4K / 8K, .evc -> .yuv with no Python in the loop: examples/evc_decode (plain C on the C ABIs; parser
threads x GOP work queue on one GPU) next to the reference decoder (oracle/_ref/ref_decode_main, its public API, -m 1 and -m 8) on the same
bytes, outputs compared byte for byte.  Streams: Main profile (IQT, ADDB, ALF), IDR every `gop` pictures, P pictures with one reference,
written by this repository's front end.  Prints one JSON line.
usage: python tests/tools/bench_stream_sizes.py [--sizes 1080p,4k,8k] [--workers 1,4,8]"""
import argparse
import hashlib
import json
import os
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

SIZES = {"1080p": (1920, 1080, 60, 6), "4k": (3840, 2160, 48, 4), "8k": (7680, 4320, 24, 3)}      # w, h, pictures, pictures per GOP


def md5(path):
    h = hashlib.md5()
    with open(path, "rb") as f:
        for blk in iter(lambda: f.read(1 << 24), b""):
            h.update(blk)
    return h.hexdigest()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--sizes", default="1080p,4k,8k")
    ap.add_argument("--workers", default="1,4,8")
    ap.add_argument("--bit-depth", type=int, default=10)
    ap.add_argument("--tiles", default="", help="CxR: tile grid of every picture (tiles parse on parallel host threads, see --tile-threads)")
    ap.add_argument("--tile-threads", default="1", help="comma list: parser threads per picture; every --workers value is run with each")
    args = ap.parse_args()
    import stream_util as su
    exe = os.path.join(ROOT, "examples", "evc_decode")
    ref = os.path.join(ROOT, "oracle", "_ref", "ref_decode_main")
    out = {"host_cores": os.cpu_count(), "what": "examples/evc_decode (C, .evc -> .yuv; decode_only = parsing + kernels + output of the slowest worker, the span the reference application times; process_wall adds process and device start-up) vs the reference "
           "decoder's public API on the same bytes; Main profile (IQT, ADDB, ALF), P pictures, closed GOPs", "sizes": {}}
    tiles = tuple(int(v) for v in args.tiles.split("x")) + (0,) if args.tiles else None
    if tiles:
        out["tiles"] = args.tiles
    with tempfile.TemporaryDirectory() as td:
        for name in args.sizes.split(","):
            w, h, n, gop = SIZES[name]
            t0 = time.perf_counter()
            data = su.make_stream(w, h, n, bit_depth=args.bit_depth, seed=5, max_refs=1, inter_frac=0.9, skip_frac=0.15, idr_period=gop,
                                  main=True, iqt=True, addb=True, alf=True, tiles=tiles)
            src = os.path.join(td, f"{name}.evc")
            open(src, "wb").write(data)
            r = {"stream": f"{w}x{h} {args.bit_depth}-bit, {n} pictures in GOPs of {gop}, {len(data)} bytes ({8 * len(data) / n / 1e6:.2f} Mbit/picture)",
                 "writer_s": round(time.perf_counter() - t0, 1), "evc_decode_fps": {}, "reference_fps": {}}
            sums = set()
            for wk, tt in [(int(v), int(u)) for v in args.workers.split(",") for u in args.tile_threads.split(",")]:
                dst = os.path.join(td, f"{name}_{wk}.yuv")
                t0 = time.perf_counter()
                p = subprocess.run([exe, "--workers", str(wk), "--tile-threads", str(tt), src, dst], stderr=subprocess.PIPE, timeout=1200)
                wk = f"{wk}x{tt}" if tiles else wk
                dt = time.perf_counter() - t0
                if p.returncode != 0:
                    r["evc_decode_fps"][str(wk)] = "error: " + p.stderr.decode()[-200:]
                    continue
                txt = p.stderr.decode()
                dec_fps = float(txt.split("slowest worker)")[1].split("s,")[1].split("pictures/s")[0])
                r["evc_decode_fps"][str(wk)] = {"decode_only": dec_fps, "process_wall": round(n / dt, 2)}
                sums.add(md5(dst))
                os.remove(dst)
            if os.path.exists(ref):
                # the reference decoder keeps the first 64 pictures for the comparison (its driver's buffer): all of them here
                for threads in (1, 8):
                    dst = os.path.join(td, f"{name}_ref.raw")
                    p = subprocess.run([ref, src, dst, str(w), str(h), str(threads)], stderr=subprocess.PIPE, timeout=1800)
                    if p.returncode != 0:
                        r["reference_fps"][str(threads)] = "error: " + p.stderr.decode()[-200:]
                        continue
                    pics, secs = p.stderr.decode().split()[-2:]
                    r["reference_fps"][str(threads)] = round(int(pics) / float(secs), 2)
                    sums.add(md5(dst))
                    os.remove(dst)
            r["bit_exact"] = len(sums) == 1
            out["sizes"][name] = r
            os.remove(src)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
