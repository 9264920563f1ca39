cd $GRAFT_REPO_ROOT
timeout -k 5 600 python -m pytest tests -m gpu -x -q 2>&1 | tail -6
timeout -k 5 200 python bench.py --steps 60 --warmup 10 --no-cpu-baseline --no-end-to-end > gpurun_out/r5t.json 2> gpurun_out/r5t.err
python - <<PY
import json
d=json.load(open("gpurun_out/r5t.json"))
print(d["value"], d["ms_per_step"], {k: v["avg_us"] for k, v in d["kernels"].items()})
PY
