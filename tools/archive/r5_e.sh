cd $GRAFT_REPO_ROOT
timeout -k 5 600 python -m pytest tests -m gpu -x -q 2>&1 | tail -4
timeout -k 5 200 python bench.py --steps 60 --warmup 10 --no-cpu-baseline --no-end-to-end > gpurun_out/r5e.json 2> gpurun_out/r5e.err
python - <<PY
import json
d=json.load(open("gpurun_out/r5e.json"))
print(d["value"], d["ms_per_step"], {k: v["avg_us"] for k, v in d["kernels"].items()})
PY
bash tools/kernel_stats.sh cfg4_main_8k_10b_ra r5e > /dev/null 2>&1
head -8 gpurun_out/r5e_cfg4_main_8k_10b_ra_kernel_stats.csv
