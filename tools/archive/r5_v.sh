# round 5: level-1 intra launch with 16 lanes per small CU (k_intra_l1) against a wave per CU (XEVD_HIP_INTRA_SMALL_MIN=1000000000): parity, then both intra launches by HIP events
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 5 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "pictures_golden or vs_oracle_random or bench_workload or full_size or all_intra or golden_streams" 2>&1 | tail -3
for w in cfg4_main_8k_10b_ra cfg2_base_1080p_8b_ippp cfg3_main_4k_10b_ra; do
for ns in 1 0 1 0; do
  if [ $ns = 1 ]; then export XEVD_HIP_INTRA_SMALL_MIN=1000000000; else unset XEVD_HIP_INTRA_SMALL_MIN; fi
  timeout -k 5 300 python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-end-to-end --workload $w > gpurun_out/r5v.json 2> gpurun_out/r5v.err
  python - <<PY
import json
try:
    d=json.load(open("gpurun_out/r5v.json"))
    print("$w no_small=$ns", d["value"], d["ms_per_step"], "intra", d["kernels"]["intra"]["avg_us"], d["config"]["batch"])
except Exception as e:
    print("$w no_small=$ns FAILED", e)
PY
done
done
