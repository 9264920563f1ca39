R=$GRAFT_REPO_ROOT
cd $R
timeout -k 5 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
gcc -O2 -I include -o /tmp/parse_time tools/parse_time.c -L xevd_amd -lxevd_host -Wl,-rpath,$R/xevd_amd
python - <<'PY'
import bench
wl = bench.WORKLOADS["cfg4_main_8k_10b_ra"]
one, data, _ = bench.write_bench_stream(wl, 17, 1)
open("/tmp/s8k1.evc", "wb").write(one)
PY
for t in 1 16; do /tmp/parse_time /tmp/s8k1.evc $t 2 | tail -1; done
timeout -k 5 600 python bench.py > gpurun_out/r4h_bench_cfg4.json 2> gpurun_out/r4h_bench_cfg4.err; python - <<'PY'
import json
d=json.loads(open("gpurun_out/r4h_bench_cfg4.json").read().strip().splitlines()[-1])
print(json.dumps({k:d[k] for k in ("value","ms_per_step","kernels","roofline","end_to_end_fps","bit_exact")}))
print(json.dumps(d["end_to_end"]))
print(json.dumps(d["cpu_baseline"]["reference_decoder"],indent=0)[:2500])
PY
timeout -k 5 600 python bench.py --gpus 2 --steps 40 > gpurun_out/r4h_bench_gpus2.json 2> gpurun_out/r4h_bench_gpus2.err; tail -c 3000 gpurun_out/r4h_bench_gpus2.json; tail -5 gpurun_out/r4h_bench_gpus2.err
