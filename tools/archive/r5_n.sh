# where the life of a k_inter wave goes: the library built with -DXGPU_INTER_TRACE (make EXTRA=-DXGPU_INTER_TRACE in xevd_amd/csrc), bench.py's resident steps, then
# the per-role sums of shader cycles between the marks (xgpu_test_inter_trace)
cd $GRAFT_REPO_ROOT
timeout -k 5 600 python - <<PY
import sys, ctypes as C, io, contextlib
sys.argv = ["bench.py", "--steps", "20", "--warmup", "5", "--no-cpu-baseline", "--no-end-to-end"]
import bench
buf = io.StringIO()
with contextlib.redirect_stdout(buf):
    bench.main()
from xevd_amd import abi
lib = abi.load()
import json
d = json.loads(buf.getvalue().strip().splitlines()[-1])
print("value", d["value"], "kernel us", d["roofline"].get("kernel_us"), "frac", d["roofline"]["frac"])
out = (C.c_ulonglong * 48)()
rc = lib.xgpu_test_inter_trace(out, 1)
names = ["work entry", "item / tables+barrier", "records, map", "requests issued", "wait luma 0", "luma passes", "sync / chroma wait", "chroma passes", "wait luma 1", "split: lists", "stores issued"]
for r, role in enumerate(("region", "tile", "split")):
    n = out[r * 16 + 15]
    if not n: continue
    tot = sum(out[r * 16 + k] for k in range(11))
    print(f"{role}: {n} waves traced, {tot / n:.0f} cycles per wave")
    for k in range(11):
        print(f"   {names[k]:24s} {out[r * 16 + k] / n:9.0f}")
PY
