cd $GRAFT_REPO_ROOT
timeout -k 5 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -4
timeout -k 5 600 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-end-to-end > gpurun_out/r5o.json 2> gpurun_out/r5o.err
python - <<PY
import json
d=json.load(open("gpurun_out/r5o.json"))
print(d["value"], d["ms_per_step"], d["roofline"])
PY
