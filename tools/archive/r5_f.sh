cd $GRAFT_REPO_ROOT
run() {
  timeout -k 5 200 python bench.py --steps 60 --warmup 10 --no-cpu-baseline --no-end-to-end > gpurun_out/r5f.json 2> gpurun_out/r5f.err
  python - "$1" <<PY
import json, sys
d=json.load(open("gpurun_out/r5f.json"))
print(sys.argv[1], d["value"], d["ms_per_step"], {k: v["avg_us"] for k, v in d["kernels"].items()})
PY
}
for rep in 1 2; do
run anyorder
XEVD_HIP_INTER_IN_ORDER=1 run inorder
XEVD_HIP_INTER_STREAMS=1 run streams_q4
GPU_MAX_HW_QUEUES=8 XEVD_HIP_INTER_STREAMS=1 run streams_q8
done
