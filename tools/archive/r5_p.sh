# where the life of a k_addb_alf wave goes: the library built with -DXGPU_ALF_TRACE (make EXTRA=-DXGPU_ALF_TRACE in xevd_amd/csrc), bench.py's resident steps, then the
# sums of shader cycles between the marks (xgpu_test_alf_trace)
cd $GRAFT_REPO_ROOT
timeout -k 5 600 python - <<PY
import sys, ctypes as C, io, contextlib, json
sys.argv = ["bench.py", "--steps", "20", "--warmup", "5", "--no-cpu-baseline", "--no-end-to-end"]
import bench
buf = io.StringIO()
with contextlib.redirect_stdout(buf):
    bench.main()
d = json.loads(buf.getvalue().strip().splitlines()[-1])
print("value", d["value"])
from xevd_amd import abi
lib = abi.load()
out = (C.c_ulonglong * 16)()
lib.xgpu_test_alf_trace(out, 1)
names = ["-", "setup, window + record loads issued, tables", "barrier 1", "records + windows to LDS, vertical strengths, list", "barrier 2", "vertical edges filtered, horizontal strengths, list", "barrier 3",
         "horizontal edges filtered, chroma", "barrier 4", "ALF classification sums", "barrier 5", "ALF filters + stores"]
n = out[15]
tot = sum(out[k] for k in range(1, 12))
print(n, "waves traced,", round(tot / n), "cycles per wave")
for k in range(1, 12):
    print(f"   {names[k]:58s} {out[k] / n:8.0f}  {100 * out[k] / tot:5.1f} %")
PY
