"""The host batch builder alone (xgpu_test_build_batch: no device) on the pictures of a .evc stream, or on bench.py's synthetic batches: ms per picture by
thread count, and that the staging block does not depend on the thread count.   usage: build_time.py [in.evc | --bench cfg4_main_8k_10b_ra] [threads ...]"""
import ctypes as C
import os
import sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from xevd_amd import abi, stream

lib = abi.load()
lib.xgpu_test_build_batch.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.POINTER(C.c_uint64), C.POINTER(C.c_int), C.POINTER(C.c_double)]


def build(sp, cb, threads):
    dg, info, ms = (C.c_uint64 * 13)(), (C.c_int * 8)(), C.c_double()
    rc = lib.xgpu_test_build_batch(C.byref(sp), C.byref(cb), threads, dg, info, C.byref(ms))
    if rc:
        raise RuntimeError(f"xgpu_test_build_batch -> {rc}")
    return tuple(dg), tuple(info), ms.value


def main():
    args = sys.argv[1:]
    threads = [int(a) for a in args if a.isdigit()] or [1, 4]
    res = {t: [] for t in threads}
    if args and args[0] == "--bench":
        import bench
        wl = bench.WORKLOADS[args[1]]
        _, batches, _ = bench.make_stream(wl, 1000, 2)
        sp = abi.make_seq_params(wl["w"], wl["h"], wl["bd"], iqt=wl["iqt"], admvp=wl["admvp"], addb=wl["addb"], alf=wl["alf"])
        for b in batches:
            cb, keep = abi.make_cu_batch(b)
            ref = None
            for t in threads:
                for rep in range(3):
                    dg, info, ms = build(sp, cb, t)
                    assert ref is None or dg == ref, "the staging block depends on the thread count"
                    ref = dg
                    res[t].append(ms)
            print("info", info, "ms", {t: round(min(res[t][-3:]), 2) for t in threads}, flush=True)
        return
    data = open(args[0], "rb").read()

    def consume(params, cbs):
        sp = abi.make_seq_params(params["width"], params["height"], params["bit_depth"], iqt=params["iqt"], admvp=params["admvp"], addb=params["addb"], alf=params["tool_alf"], eipd=params["eipd"])
        ref = None
        out = {}
        for t in threads:
            best = 1e9
            for rep in range(2):
                dg, info, ms = build(sp, cbs, t)
                assert ref is None or dg == ref, "the staging block depends on the thread count"
                ref = dg
                best = min(best, ms)
            res[t].append(best); out[t] = round(best, 2)
        print("poc", params["poc"], "info", info, "ms", out, flush=True)
        return None
    for _ in stream.iter_stream(data, consume_batch=consume, threads=8):
        pass
    print({t: round(float(np.mean(v)), 2) for t, v in res.items()})


main()
