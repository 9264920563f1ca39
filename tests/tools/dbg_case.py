"""debug helper: run one picture case on the GPU stage by stage against the oracle and print the first differences"""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import numpy as np
import cases, golden_io
from xevd_amd import abi

name = sys.argv[1]
case, exp = golden_io.load_picture_case(name)
b = case["batch"]
for stage, kw in (("recon", dict(deblock=False, pad=False, alf=False)), ("deblock", dict(pad=False, alf=False)), ("alf", dict(pad=False))):
    out = cases.run_gpu(case, **kw)
    c2 = dict(case)
    if stage != "alf":
        c2["alf_params"] = None
    ref, pre, maps, _ = cases.run_cpu("oracle", c2, deblock=stage != "recon", pad=False)
    for c in range(3):
        pad = abi.PAD_L if c == 0 else abi.PAD_C
        g = out[c][pad:-pad, pad:-pad]; r = ref.active(c)
        d = np.argwhere(g != r)
        print(stage, "plane", c, "ndiff", len(d), d[:6].tolist())
        if len(d):
            yy, xx = d[0]
            sc = 1 if c == 0 else 2
            for i in range(len(b["x"])):
                if b["x"][i] <= xx * sc < b["x"][i] + (1 << b["log2w"][i]) and b["y"][i] <= yy * sc < b["y"][i] + (1 << b["log2h"][i]):
                    print("  CU", i, b["x"][i], b["y"][i], 1 << b["log2w"][i], 1 << b["log2h"][i], "mode", b["pred_mode"][i], "cbf", b["cbf"][i],
                          "ats", None if b.get("ats") is None else b["ats"][i], "ats_inter", None if b.get("ats_inter") is None else b["ats_inter"][i])
            print("  gpu", g[yy, xx:xx + 8].tolist(), "ref", r[yy, xx:xx + 8].tolist())
