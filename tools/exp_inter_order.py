"""k_inter at cfg4 as a function of the region order / occupancy knobs (GPU box).  usage: exp_inter_order.py strip:ldspad[:order] ...   (read per launch from the environment)"""
import sys, os, json
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from xevd_amd.decoder import XgpuDecoder

wl = bench.WORKLOADS[os.environ.get("EXP_WORKLOAD", "cfg4_main_8k_10b_ra")]
first, batches, alf = bench.make_stream(wl, 1000, 2)
dec = XgpuDecoder(wl["w"], wl["h"], wl["bd"], device=0, iqt=wl["iqt"], admvp=wl["admvp"], addb=wl["addb"], alf=wl["alf"], max_pics=4)
slots = [dec.pic_alloc() for _ in range(3)]
for i in range(2):
    dec.pic_upload(slots[i], first[i]); dec.frame_begin(slots[i], i - 1, {}); dec.pad(); dec.frame_end()
handles = [dec.batch_create(b) for b in batches]
def step(k):
    cur, r0, r1 = slots[(k + 2) % 3], slots[(k + 1) % 3], slots[k % 3]
    dec.decode_picture(cur, k + 1, {(0, 0): (r0, k), (0, 1): (r1, k - 1)}, handles[k % 2], alf=alf)
out = {}
steps = int(os.environ.get("EXP_STEPS", "20"))
for cfg in sys.argv[1:]:
    f = cfg.split(":")
    os.environ["XEVD_HIP_INTER_STRIP"] = f[0]
    os.environ["XEVD_HIP_INTER_LDSPAD"] = f[1] if len(f) > 1 else "0"
    os.environ["XEVD_HIP_INTER_ORDER"] = f[2] if len(f) > 2 else "0"
    for k in range(3):
        step(k)
    dec.sync(); dec.timing_enable(True); dec.timing_reset()
    for k in range(steps):
        step(3 + k)
    tim = dec.timing_get(); dec.timing_enable(False)
    out[cfg] = {k: round(1e3 * v[0] / v[1], 1) for k, v in tim.items() if v[1]}
    print(cfg, "inter_us", out[cfg].get("inter"), out[cfg], flush=True)
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump(out, open(os.path.join(ROOT, "gpurun_out", "exp_inter_order.json"), "w"))
