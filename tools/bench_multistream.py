#!/usr/bin/env python3
"""Many real bitstreams on ONE GPU: K decoder instances (one host parser thread + one xgpu context / HIP stream each) share the device -
the deployment shape for stream decoding, where one stream's entropy decoding (a serial CABAC chain on the host) cannot keep an MI355X busy.
Streams are written by this repository's front end (Baseline 1080p IPPP, BASELINE.json configs[1] shape).  Prints one JSON line.
usage: python tools/bench_multistream.py [--streams 1,4,16] [--pics 24] [--width 1920 --height 1080]"""
import argparse
import json
import os
import sys
import threading
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from xevd_amd import stream, synth                     # noqa: E402
from xevd_amd.player import StreamDecoder             # noqa: E402


def write_stream(w, h, n, seed):
    rng = np.random.default_rng(seed)
    wr = stream.StreamWriter(w, h, 8, 1)
    try:
        for k in range(n):
            b = synth.gen_frame(rng, w, h, 8, inter_frac=0.0 if k == 0 else 0.9, n_refs=(1, 0), coded_frac=0.6, max_level=6, amp=1.0)
            wr.add_picture(b, stream.SLICE_I if k == 0 else stream.SLICE_P, slice_qp=30, idr=k == 0)
        return wr.bytes()
    finally:
        wr.close()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--streams", default="1,4,16")
    ap.add_argument("--pics", type=int, default=24)
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--device", type=int, default=0)
    args = ap.parse_args()
    data = [write_stream(args.width, args.height, args.pics, 100 + i) for i in range(4)]      # 4 distinct streams, reused round-robin
    out = {"stream": f"Baseline {args.width}x{args.height} 8-bit IPPP, {args.pics} pictures, {len(data[0])} bytes", "host_cores": os.cpu_count(), "fps": {}}

    def run(i, counts):
        n = 0
        for _ in StreamDecoder(data[i % len(data)], device=args.device).pictures(download=False):
            n += 1
        counts[i] = n

    run(0, [0])                                                                             # warm-up: library load, first allocations
    for k in [int(v) for v in args.streams.split(",")]:
        counts = [0] * k
        th = [threading.Thread(target=run, args=(i, counts)) for i in range(k)]
        t0 = time.perf_counter()
        for t in th:
            t.start()
        for t in th:
            t.join()
        dt = time.perf_counter() - t0
        out["fps"][str(k)] = round(sum(counts) / dt, 1)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
