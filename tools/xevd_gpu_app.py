#!/usr/bin/env python3
"""Decode an MPEG-5 EVC Baseline bitstream on the MI355X and write planar YUV - the counterpart of the reference's sample
application (app/xevd_app.c: -i in.evc -o out.yuv --output-bit-depth N).  usage: xevd_gpu_app.py -i in.evc -o out.yuv"""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from xevd_amd.player import StreamDecoder      # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("-i", "--input", required=True)
    ap.add_argument("-o", "--output")
    ap.add_argument("--output-bit-depth", type=int, default=0, help="0 = the stream's bit depth (8 -> bytes, else 16-bit little endian)")
    ap.add_argument("--device", type=int, default=0)
    args = ap.parse_args()
    data = open(args.input, "rb").read()
    t0 = time.perf_counter()
    pics = StreamDecoder(data, device=args.device).output_order()
    dt = time.perf_counter() - t0
    if args.output:
        with open(args.output, "wb") as f:
            for p, planes in pics:
                bd_in = p["bit_depth"]
                bd_out = args.output_bit_depth or bd_in
                for pl in planes:
                    v = pl.astype(np.int32)
                    if bd_out < bd_in:      # rounding down-conversion, as the reference's imgb_cpy_conv_rec (app/xevd_app_util.h) does
                        sh = bd_in - bd_out
                        v = np.clip((v + (1 << (sh - 1))) >> sh, 0, (1 << bd_out) - 1)
                    elif bd_out > bd_in:
                        v = v << (bd_out - bd_in)
                    f.write(v.astype(np.uint8 if bd_out == 8 else "<u2").tobytes())
    print(f"{len(pics)} pictures, {len(pics) / dt:.1f} pictures/s (parse + upload + kernels + download)", file=sys.stderr)


if __name__ == "__main__":
    main()
