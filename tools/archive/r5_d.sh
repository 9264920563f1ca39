cd $GRAFT_REPO_ROOT
timeout -k 5 400 python -m pytest tests -m gpu -x -q -k "golden or workload or all_intra" 2>&1 | tail -3
run() {
  timeout -k 5 200 python bench.py --steps 60 --warmup 10 --no-cpu-baseline --no-end-to-end > gpurun_out/r5d.json 2> gpurun_out/r5d.err
  python - "$1" <<PY
import json, sys
d=json.load(open("gpurun_out/r5d.json"))
print(sys.argv[1], d["value"], d["ms_per_step"], {k: v["avg_us"] for k, v in d["kernels"].items()})
PY
}
for rep in 1 2; do
run split
XEVD_HIP_SPLIT_W4=1 run split_w4
done
bash tools/kernel_stats.sh cfg4_main_8k_10b_ra r5d > /dev/null 2>&1
head -6 gpurun_out/r5d_cfg4_main_8k_10b_ra_kernel_stats.csv
