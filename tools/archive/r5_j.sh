cd $GRAFT_REPO_ROOT
run() {
  timeout -k 5 200 python bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-end-to-end > gpurun_out/r5j.json 2> gpurun_out/r5j.err
  python - "$1" <<PY
import json, sys
d=json.load(open("gpurun_out/r5j.json"))
print(sys.argv[1], d["kernels"]["addb_alf"]["avg_us"])
PY
}
run full
XEVD_HIP_ALF_ABLATE=15 run skeleton
XEVD_HIP_ALF_ABLATE=31 run skeleton_no_loads
XEVD_HIP_ALF_ABLATE=47 run skeleton_no_stores
XEVD_HIP_ALF_ABLATE=63 run skeleton_no_loads_no_stores
XEVD_HIP_ALF_LDSPAD=8000 XEVD_HIP_ALF_ABLATE=15 run skeleton_4wg_per_cu
XEVD_HIP_ALF_LDSPAD=8000 run full_4wg_per_cu
