"""kernel durations (HIP events on the kernel stream, xgpu_timing) of a Baseline IPPP picture at several sizes: where the fixed part of a kernel's time sits"""
import sys, os, json
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import bench
from xevd_amd.decoder import XgpuDecoder
from xevd_amd import abi
for (w, h) in [(int(a.split("x")[0]), int(a.split("x")[1])) for a in (sys.argv[1:] or ["320x184", "640x360", "1280x720", "1920x1080", "3840x2160"])]:
    wl = dict(bench.WORKLOADS[os.environ.get("EXP_WL", "cfg2_base_1080p_8b_ippp")]); wl["w"], wl["h"] = w, h
    first, batches, alf = bench.make_stream(wl, 1000, 2)
    dec = XgpuDecoder(w, h, wl["bd"], device=0, iqt=wl["iqt"], admvp=wl["admvp"], addb=wl["addb"], alf=wl["alf"], max_pics=4)
    sl = [dec.pic_alloc(), dec.pic_alloc(), dec.pic_alloc()]
    for i in range(2):
        dec.pic_upload(sl[i], first[i]); dec.frame_begin(sl[i], i - 1, {}); dec.pad(); dec.frame_end()
    hs = [dec.batch_create(b) for b in batches]
    dec.sync(); dec.timing_enable(True)
    for rep in range(2):
        dec.timing_reset()
        for k in range(40):
            dec.decode_picture(sl[(k + 2) % 3], k + 1, {(0, 0): (sl[(k + 1) % 3], k)}, hs[k % 2], alf=alf)
        dec.sync()
    t = dec.timing_get()
    print(w, h, {k: round(ms * 1e3 / n, 2) for k, (ms, n) in t.items() if n})
    dec.close()
