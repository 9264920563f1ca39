"""debug aid: a synthetic picture whose CUs are reordered inside their CTUs, GPU against oracle - where the first differences sit"""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests")); sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import numpy as np, cases
import test_gpu_parity as t
cfgs = {"base": ("perm_base_modes_b", 200, 136, 8, 0, 0, (1, 1), 0.4, {"inter_frac": 0.6, "split_prob": 0.7}),
        "eipd_i": ("perm_eipd_noaddb_i", 136, 136, 8, 1, 1, (1, 0), 0.0, {"eipd": 1, "inter_frac": 0.0, "split_prob": 0.7})}
for key in sys.argv[1:]:
    name, w, h, bd, admvp, iqt, n_refs, bi_frac, tools = cfgs[key]
    cs = cases.build_case(name, w, h, bd, admvp, iqt, n_refs, bi_frac, tools, seed=1)
    cs["batch"] = t._permute_inside_ctus(cs["batch"], 41)
    b = cs["batch"]
    for dbk in (False, True):
        o = cases.run_cpu("oracle", cs, deblock=dbk, pad=False)[0]
        g = cases.run_gpu(cs, deblock=dbk, pad=False)
        for c in range(3):
            d = np.argwhere(g[c] != o.bufs[c])
            print(key, "deblock", dbk, "plane", c, len(d), "diffs")
            seen = set()
            for (yy, xx) in d[:300]:
                sh = 1 if c else 0; pad = 144 >> sh
                X, Y = (xx - pad) << sh, (yy - pad) << sh
                for i in range(len(b["x"])):
                    if b["x"][i] <= X < b["x"][i] + (1 << b["log2w"][i]) and b["y"][i] <= Y < b["y"][i] + (1 << b["log2h"][i]):
                        if i not in seen:
                            seen.add(i)
                            print(f"    CU {i} at ({b['x'][i]},{b['y'][i]}) {1 << b['log2w'][i]}x{1 << b['log2h'][i]} mode {b['pred_mode'][i]} ipm {b['ipm'][i].tolist() if b.get('ipm') is not None else None} cbf {b['cbf'][i]} first ({X},{Y}) gpu {g[c][yy, xx]} want {o.bufs[c][yy, xx]}")
                        break
                if len(seen) >= 6: break
        if dbk is False and any(len(np.argwhere(g[c] != o.bufs[c])) for c in range(3)): break
