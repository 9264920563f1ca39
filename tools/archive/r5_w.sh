# round 5: A/B of library variants under the pipelined bench loop (value = frames/s with the next picture's residual pass riding in the data-flow intra launch)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for rep in 1 2 3; do
for v in $RUNS; do
  cp tools/ab/lib_$v.so xevd_amd/libxevd_hip.so
  for w in $WLS; do
  timeout -k 5 300 python bench.py --steps 60 --warmup 10 --no-cpu-baseline --no-end-to-end --workload $w > gpurun_out/r5w.json 2> gpurun_out/r5w.err
  python - <<PY
import json
try:
    d=json.load(open("gpurun_out/r5w.json"))
    print("$v $w", d["value"], d["ms_per_step"], {k:x["avg_us"] for k,x in d["kernels"].items()})
except Exception as e:
    print("$v $w FAILED", e)
PY
  done
done
done
