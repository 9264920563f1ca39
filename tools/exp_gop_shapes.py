"""examples/evc_decode GOP-parallel on bench.py's 8K Main stream: workers x tile threads x builders x build threads (how much of the host's CPU quota the shapes reach).
usage: exp_gop_shapes.py [workload]"""
import hashlib, os, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench

wl = bench.WORKLOADS[sys.argv[1] if len(sys.argv) > 1 else "cfg4_main_8k_10b_ra"]
one, data, what = bench.write_bench_stream(wl, 17, 12)
ref = None
with tempfile.TemporaryDirectory() as td:
    src, dst = os.path.join(td, "all.evc"), os.path.join(td, "o.yuv")
    open(src, "wb").write(data)
    for w, t, nb, bt in ((2, 8, 2, 4), (2, 16, 2, 4), (3, 8, 1, 4), (3, 8, 2, 2), (4, 8, 1, 2), (4, 8, 2, 2), (3, 16, 2, 4), (6, 4, 1, 2), (4, 16, 2, 4), (2, 8, 3, 4)):
        rep = bench.run_evc_decode(["--workers", str(w), "--tile-threads", str(t), "--builders", str(nb), "--build-threads", str(bt), "--keep-units", "2", src, dst])
        if "error" in rep:
            print(w, t, nb, bt, rep, flush=True)
            continue
        md5 = hashlib.md5(open(dst, "rb").read()).hexdigest()
        ref = ref or md5
        print(f"workers {w} x tile threads {t}, builders {nb} x {bt}: fps {rep['fps_decode_only']:7.2f}  parse {rep['parse_ms_per_picture']:6.2f}  build {rep['build_ms_per_picture']:6.2f}  "
              f"cpu s/picture {(rep['cpu_user_s'] + rep['cpu_sys_s']) / max(rep['pictures'], 1):.4f}  same bytes {md5 == ref}", flush=True)
