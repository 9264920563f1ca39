#!/bin/bash
# where the life of a k_addb_alf wave goes: a library built with -DXGPU_ALF_TRACE (tools/build_variant.sh trace -DXGPU_ALF_TRACE) under bench.py's resident steps, then the sums
# of shader cycles between the kernel's marks (xgpu_test_alf_trace).  usage (on the GPU box): tools/alf_trace.sh [variant name, default trace]
cd ${GRAFT_REPO_ROOT:-/root/repo}
V=${1:-trace}
cp xevd_amd/libxevd_hip.so /tmp/libxevd_hip_product.so
cp tools/ab/libxevd_hip_$V.so xevd_amd/libxevd_hip.so
timeout -k 5 600 python - <<PY
import sys, ctypes as C, io, contextlib, json
sys.argv = ["bench.py", "--steps", "20", "--warmup", "5", "--no-cpu-baseline", "--no-end-to-end"]
import bench
buf = io.StringIO()
with contextlib.redirect_stdout(buf):
    bench.main()
d = json.loads(buf.getvalue().strip().splitlines()[-1])
print("value", d["value"], {k: v["avg_us"] for k, v in d["kernels"].items()})
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
cp /tmp/libxevd_hip_product.so xevd_amd/libxevd_hip.so
