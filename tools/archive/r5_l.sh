cd $GRAFT_REPO_ROOT
timeout -k 5 500 python -m pytest tests -m gpu -x -q -k "evc_decode or streams_leg or golden_streams or plain_c" 2>&1 | tail -3
timeout -k 5 600 python bench.py --steps 20 --warmup 5 > gpurun_out/r5l.json 2> gpurun_out/r5l.err
python - <<PY
import json
d=json.load(open("gpurun_out/r5l.json"))
print(d["value"], d["ms_per_step"], d["end_to_end_fps"], (d.get("two_contexts") or {}).get("fps"), d["roofline"]["frac"], d["roofline"]["traffic"])
rd=d["cpu_baseline"]["reference_decoder"]
print(rd["frames_per_s_by_threads"], rd["bit_exact"], rd["reference_threads_8_equals_1"])
for k,v in rd["evc_decode_on_gpu"].items(): print(k, v)
PY
