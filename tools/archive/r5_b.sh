# round 5: launch modes of the three inter kernels (XEVD_HIP_INTER_MODE 0 one stream / 1 three streams / 2 any-order launches; XEVD_HIP_INTER_PRIO: high-priority side streams)
cd $GRAFT_REPO_ROOT
run() {
  timeout -k 5 200 python bench.py --steps 60 --warmup 10 --no-cpu-baseline --no-end-to-end > gpurun_out/r5b.json 2> gpurun_out/r5b.err
  python - "$1" <<PY
import json, sys
d=json.load(open("gpurun_out/r5b.json"))
print(sys.argv[1], d["value"], d["ms_per_step"], {k: v["avg_us"] for k, v in d["kernels"].items()})
PY
}
for rep in 1 2; do
XEVD_HIP_INTER_MODE=0 run mode0_serial
XEVD_HIP_INTER_MODE=1 run mode1_streams
XEVD_HIP_INTER_MODE=1 XEVD_HIP_INTER_PRIO=1 run mode1_streams_hiprio
XEVD_HIP_INTER_MODE=2 run mode2_anyorder
done
timeout -k 5 300 python -m pytest tests -m gpu -x -q -k "golden or workload" 2>&1 | tail -2
