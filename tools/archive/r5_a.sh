# round 5, first GPU pass of the three-kernel k_inter: suite, bench in both launch modes, per-kernel rocprof stats
set -x
cd $GRAFT_REPO_ROOT
timeout -k 5 500 python -m pytest tests -m gpu -x -q 2>&1 | tail -5
for mode in fork serial; do
  if [ $mode = serial ]; then export XEVD_HIP_INTER_SERIAL=1; else unset XEVD_HIP_INTER_SERIAL; fi
  timeout -k 5 200 python bench.py --steps 60 --warmup 10 --no-cpu-baseline --no-end-to-end > gpurun_out/r5a_$mode.json 2> gpurun_out/r5a_$mode.err
  python - <<PY
import json
d=json.load(open("gpurun_out/r5a_$mode.json"))
print("$mode", d["value"], d["ms_per_step"], d["kernels"], d["roofline"]["frac"])
PY
done
unset XEVD_HIP_INTER_SERIAL
bash tools/kernel_stats.sh cfg4_main_8k_10b_ra r5a
cat gpurun_out/r5a_cfg4_main_8k_10b_ra_kernel_stats.csv
