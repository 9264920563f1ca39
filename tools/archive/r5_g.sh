cd $GRAFT_REPO_ROOT
timeout -k 5 600 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
run() {
  timeout -k 5 200 python bench.py --steps 60 --warmup 10 --no-cpu-baseline --no-end-to-end > gpurun_out/r5g.json 2> gpurun_out/r5g.err
  python - "$1" <<PY
import json, sys
d=json.load(open("gpurun_out/r5g.json"))
print(sys.argv[1], d["value"], d["ms_per_step"], {k: v["avg_us"] for k, v in d["kernels"].items()})
PY
}
for rep in 1 2; do
run quad
XEVD_HIP_INTER_NO_QUAD=1 run noquad
done
bash tools/kernel_stats.sh cfg4_main_8k_10b_ra r5g > /dev/null 2>&1
head -7 gpurun_out/r5g_cfg4_main_8k_10b_ra_kernel_stats.csv | cut -c1-100
