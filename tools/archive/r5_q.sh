cd $GRAFT_REPO_ROOT
for nt in 0 1 0 1; do
XEVD_HIP_INTER_NT=$nt timeout -k 5 300 python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-end-to-end > gpurun_out/r5q.json 2> gpurun_out/r5q.err
python - <<PY
import json
d=json.load(open("gpurun_out/r5q.json"))
print("nt $nt", d["value"], d["ms_per_step"], d["roofline"]["avg_launch_us"])
PY
done
