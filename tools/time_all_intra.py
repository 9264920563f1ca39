#!/usr/bin/env python3
"""Kernel time of an all-intra picture (the dependency graph at its deepest: thousands of levels): the same resident batch decoded N times.
usage: python tools/time_all_intra.py [--width 1920 --height 1080] [--main] [--reps 20]      (XEVD_HIP_NO_STRANDS=1 for the A/B)"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import cases                                         # noqa: E402
from xevd_amd.decoder import XgpuDecoder            # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--main", action="store_true", help="Main profile: EIPD predictors, IQT, ADDB, ALF")
    ap.add_argument("--reps", type=int, default=20)
    ap.add_argument("--mode", type=int, default=-1, help="every CU with this luma mode (chroma: derived mode) - isolates the cost of a predictor from the cost of the chain")
    a = ap.parse_args()
    tools = {"eipd": 1, "addb": 1, "alf": 1} if a.main else None
    cs = cases.build_case("all_intra", a.width, a.height, 10 if a.main else 8, 1 if a.main else 0, 1 if a.main else 0, (1, 0), 0.0, tools=tools, inter_frac=0.0)
    if a.mode >= 0:
        cs["batch"]["ipm"][:, 0] = a.mode
        cs["batch"]["ipm"][:, 1] = 0
    with XgpuDecoder(cs["w"], cs["h"], cs["bd"], iqt=cs["iqt"], admvp=cs["admvp"], addb=cs.get("addb", 0), alf=cs.get("alf", 0), eipd=cs.get("eipd", 0), max_pics=4) as dec:
        cur = dec.pic_alloc()
        hb = dec.batch_create(cs["batch"])
        info = dec.batch_info(hb)
        kw = dict(deblock=True, pad=True, qp_u_offset=cases.QP_OFFSETS[0], qp_v_offset=cases.QP_OFFSETS[1], alf=cs.get("alf_params"))
        for _ in range(3):
            dec.decode_picture(cur, cases.CUR_POC, {}, hb, **kw)
        dec.sync()
        t0 = time.perf_counter()
        for _ in range(a.reps):
            dec.decode_picture(cur, cases.CUR_POC, {}, hb, **kw)
        dec.sync()
        ms = 1e3 * (time.perf_counter() - t0) / a.reps
    print(json.dumps({"picture": f"{a.width}x{a.height} all intra, {'Main (EIPD)' if a.main else 'Baseline'}", "ms_per_picture": round(ms, 3), "batch": info,
                      "strands": os.environ.get("XEVD_HIP_NO_STRANDS") is None}))


if __name__ == "__main__":
    main()
