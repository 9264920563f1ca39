"""examples/evc_decode on bench.py's 8K Main stream: builder threads per worker (--builders) x threads per build; every run's first two IDR periods must be
the same bytes.   usage: exp_builders.py [workload]"""
import hashlib, json, os, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench

wl = bench.WORKLOADS[sys.argv[1] if len(sys.argv) > 1 else "cfg4_main_8k_10b_ra"]
one, data, what = bench.write_bench_stream(wl, 17, 12)
quota = bench.host_cpu_quota()
ref = None
with tempfile.TemporaryDirectory() as td:
    src, dst = os.path.join(td, "all.evc"), os.path.join(td, "o.yuv")
    open(src, "wb").write(data)
    shapes = []
    for nb in (1, 2, 3):
        for bt in (2, 4, 8):
            shapes.append((f"one stream, builders {nb} x {bt}", ["--workers", "1", "--tile-threads", str(min(16, quota)), "--build-threads", str(bt), "--builders", str(nb)]))
    for nb in (1, 2):
        shapes.append((f"gop 2x8, builders {nb} x 4", ["--workers", "2", "--tile-threads", "8", "--build-threads", "4", "--builders", str(nb)]))
        shapes.append((f"gop 2x8, builders {nb} x 2", ["--workers", "2", "--tile-threads", "8", "--build-threads", "2", "--builders", str(nb)]))
        shapes.append((f"gop 4x4, builders {nb} x 2", ["--workers", "4", "--tile-threads", "4", "--build-threads", "2", "--builders", str(nb)]))
    for name, args in shapes:
        rep = bench.run_evc_decode(args + ["--keep-units", "2", src, dst])
        if "error" in rep:
            print(name, rep, flush=True)
            continue
        md5 = hashlib.md5(open(dst, "rb").read()).hexdigest()
        ref = ref or md5
        print(f"{name:36s} fps {rep['fps_decode_only']:7.2f}  parse {rep['parse_ms_per_picture']:6.2f}  build {rep['build_ms_per_picture']:6.2f}  cpu s/picture "
              f"{(rep['cpu_user_s'] + rep['cpu_sys_s']) / max(rep['pictures'], 1):.4f}  same bytes {md5 == ref}", flush=True)
