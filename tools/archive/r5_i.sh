cd $GRAFT_REPO_ROOT
timeout -k 5 300 python -m pytest tests -m gpu -x -q -k "golden or intra or workload" 2>&1 | tail -2
run() {
  timeout -k 5 200 python bench.py --steps 60 --warmup 10 --no-cpu-baseline --no-end-to-end $2 > gpurun_out/r5i.json 2> gpurun_out/r5i.err
  python - "$1" <<PY
import json, sys
d=json.load(open("gpurun_out/r5i.json"))
print(sys.argv[1], d["value"], d["ms_per_step"], {k: v["avg_us"] for k, v in d["kernels"].items()})
PY
}
for rep in 1 2; do
run l1_w8
XEVD_HIP_INTRA_L1_W6=1 run l1_w6
done
run l1_w8_1080p "--workload cfg2_base_1080p_8b_ippp"
XEVD_HIP_INTRA_L1_W6=1 run l1_w6_1080p "--workload cfg2_base_1080p_8b_ippp"
bash tools/kernel_stats.sh cfg4_main_8k_10b_ra r5i > /dev/null 2>&1
grep -i "intra" gpurun_out/r5i_cfg4_main_8k_10b_ra_kernel_stats.csv | cut -c1-110
