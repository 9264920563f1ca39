cd $GRAFT_REPO_ROOT
XEVD_HIP_INTER_FUSED=1 timeout -k 5 600 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
XEVD_HIP_INTER_FUSED=2 timeout -k 5 600 python -m pytest tests -m gpu -x -q -k "golden or workload" 2>&1 | tail -2
run() {
  timeout -k 5 200 python bench.py --steps 60 --warmup 10 --no-cpu-baseline --no-end-to-end > gpurun_out/r5h.json 2> gpurun_out/r5h.err
  python - "$1" <<PY
import json, sys
d=json.load(open("gpurun_out/r5h.json"))
print(sys.argv[1], d["value"], d["ms_per_step"], {k: v["avg_us"] for k, v in d["kernels"].items()})
PY
}
for rep in 1 2; do
run three_launches
XEVD_HIP_INTER_FUSED=1 run fused
XEVD_HIP_INTER_FUSED=2 run fused_all_first
done
export XEVD_HIP_INTER_FUSED=1
bash tools/prof_pmc.sh r5h_rd cfg4_main_8k_10b_ra "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum" | grep -i inter
bash tools/prof_pmc.sh r5h_wr cfg4_main_8k_10b_ra "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum" | grep -i inter
