cd $GRAFT_REPO_ROOT
run() {
  timeout -k 5 200 python bench.py --steps 60 --warmup 10 --no-cpu-baseline --no-end-to-end > gpurun_out/r5k.json 2> gpurun_out/r5k.err
  python - "$1" <<PY
import json, sys
d=json.load(open("gpurun_out/r5k.json"))
print(sys.argv[1], d["value"], d["ms_per_step"], {k: v["avg_us"] for k, v in d["kernels"].items()})
PY
}
for s in 16 4 8 12 24 32 60 120 16; do XEVD_HIP_INTER_STRIP=$s run strip_$s; done
XEVD_HIP_INTER_ALL_FIRST=1 run all_first
