#!/usr/bin/env python3
"""Decode an MPEG-5 EVC Baseline bitstream on the MI355X and write planar YUV - the counterpart of the reference's sample
application (app/xevd_app.c: -i in.evc -o out.yuv --output-bit-depth N).  usage: xevd_gpu_app.py -i in.evc -o out.yuv"""
import argparse
import os
import sys
import time

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
    # crop-free output like the reference application; bit-depth conversion and plane packing run on the device (xgpu_pic_output)
    pics = StreamDecoder(data, device=args.device).output_order(output_bit_depth=args.output_bit_depth)
    dt = time.perf_counter() - t0
    if args.output:
        with open(args.output, "wb") as f:
            for _, frame in pics:
                f.write(frame.tobytes())
    print(f"{len(pics)} pictures, {len(pics) / dt:.1f} pictures/s (parse + upload + kernels + download)", file=sys.stderr)


if __name__ == "__main__":
    main()
