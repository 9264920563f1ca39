"""debug aid: a golden stream through the GPU path against its golden pictures - the CUs that hold the first differing samples"""
import sys, os
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests")); sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import stream_util as su, golden_io
name = sys.argv[1]
d = np.load(os.path.join(golden_io.GOLDEN, f"stream_{name}.npz"))
data = d["bytes"].tobytes()
pics = []
su.decode_oracle(data, keep_params=pics)
ours = su.decode_gpu(data)
for k in range(len(ours)):
    b = pics[k]["batch"] if k < len(pics) else None
    for c in range(3):
        diff = np.argwhere(ours[k][c] != d[f"p{k}_{c}"])
        if len(diff) == 0: continue
        print(f"picture {k} plane {c}: {len(diff)} samples differ")
        seen = set()
        for (y, x) in diff[:400]:
            X, Y = (x << 1, y << 1) if c else (x, y)
            for i in range(len(b["x"])):
                if b["x"][i] <= X < b["x"][i] + (1 << b["log2w"][i]) and b["y"][i] <= Y < b["y"][i] + (1 << b["log2h"][i]):
                    if i not in seen:
                        seen.add(i)
                        print(f"   CU {i} at ({b['x'][i]},{b['y'][i]}) {1 << b['log2w'][i]}x{1 << b['log2h'][i]} mode {b['pred_mode'][i]} ipm {b['ipm'][i].tolist()} first diff at ({x},{y}) ours {ours[k][c][y, x]} want {d[f'p{k}_{c}'][y, x]}")
                    break
            if len(seen) >= 8: break
    if any(not np.array_equal(ours[k][c], d[f"p{k}_{c}"]) for c in range(3)): break
