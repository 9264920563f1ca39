# round 5: the split role's requests - lanes sit out of what their identity taps multiply by zero, chroma rows as one 12-byte request, fenced instalments.
# Libraries built from the variants' macros (tools/ab/lib_*.so); parity first, then k_inter by HIP events under bench.py on ONE box.
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for lib in $PARITY_LIBS; do
cp tools/ab/lib_$lib.so xevd_amd/libxevd_hip.so
timeout -k 5 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "mc_blocks or pictures_golden or inter_all_first or vs_oracle_random or bench_workload or full_size" 2>&1 | tail -2
done
run() {
  cp tools/ab/lib_$1.so xevd_amd/libxevd_hip.so
  if [ "$2" = 1 ]; then export XEVD_HIP_INTER_ALL_FIRST=1; else unset XEVD_HIP_INTER_ALL_FIRST; fi
  timeout -k 5 300 python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-end-to-end > gpurun_out/r5u.json 2> gpurun_out/r5u.err
  python - <<PY
import json
try:
    d=json.load(open("gpurun_out/r5u.json"))
    print("$1 af=$2", d["value"], d["ms_per_step"], "inter", d["kernels"]["inter"]["avg_us"])
except Exception as e:
    print("$1 af=$2 FAILED", e)
PY
}
for v in $RUNS; do run $v 0; done
