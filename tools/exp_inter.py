"""k_inter time at 8K as a function of the CU size mix (GPU box): which waves cost what."""
import sys, os, json
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from xevd_amd import synth
from xevd_amd.decoder import XgpuDecoder

W, H, BD = 7680, 4320, 10
cases = [("all64", 0.0, 2), ("all32", 1.0, 5), ("all16", 1.0, 4), ("all8", 1.0, 3), ("all4", 1.0, 2), ("mix", 0.5, 2)]
sel = sys.argv[1:] or [c[0] for c in cases]
rng = np.random.default_rng(5)
first = [synth.gen_picture(rng, W, H, BD) for _ in range(2)]
dec = XgpuDecoder(W, H, BD, device=0, iqt=1, admvp=1, addb=1, alf=0, max_pics=4)
slots = [dec.pic_alloc() for _ in range(3)]
for i in range(2):
    dec.pic_upload(slots[i], first[i]); dec.frame_begin(slots[i], i - 1, {}); dec.pad(); dec.frame_end()
out = {}
for name, sp, ml in cases:
    if name not in sel:
        continue
    for tag, kw in (("bi50", dict(bi_frac=0.5)), ("uni", dict(bi_frac=0.0)), ("zero_mv", dict(bi_frac=0.0, mv_sigma_px=0.0, oob_frac=0.0))):
        if os.environ.get("EXP_TAGS") and tag not in os.environ["EXP_TAGS"].split(","):
            continue
        b = synth.gen_frame(rng, W, H, BD, inter_frac=1.0, coded_frac=0.6, n_refs=(1, 1), split_prob=sp, min_log2=ml, **kw)
        h = dec.batch_create(b)
        def step(k):
            cur, r0, r1 = slots[(k + 2) % 3], slots[(k + 1) % 3], slots[k % 3]
            dec.decode_picture(cur, k + 1, {(0, 0): (r0, k), (0, 1): (r1, k - 1)}, h, alf=None)
        for k in range(3):
            step(k)
        dec.sync(); dec.timing_enable(True); dec.timing_reset()
        for k in range(10):
            step(3 + k)
        tim = dec.timing_get(); dec.timing_enable(False)
        out[f"{name}_{tag}"] = round(1e3 * tim["inter"][0] / tim["inter"][1], 1)
        print(name, tag, "n_cu", len(b["x"]), "inter_us", out[f"{name}_{tag}"], flush=True)
        dec.batch_destroy(h)
os.makedirs(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out"), exist_ok=True)
json.dump(out, open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out", "exp_inter.json"), "w"))
