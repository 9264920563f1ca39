# round 3, GPU call e: full GPU suite (host-side DMVR streams, bench-stream test), then the 4K bench line with the diagnostic stream leg
set -x
mkdir -p gpurun_out
timeout -k 5 1200 python -m pytest tests -m gpu -q 2>&1 | tail -15
P='import json,sys;d=json.load(open(sys.argv[1]));print(d["value"],d["kernels"],d["whole_frame"],d.get("bit_exact"));print(json.dumps(d["cpu_baseline"].get("reference_decoder"),indent=1))'
( time timeout -k 5 600 python bench.py --steps 50 --warmup 5 --workload cfg3_main_4k_10b_ra > gpurun_out/r3e_b4k.json 2> gpurun_out/r3e_b4k.err ) 2>&1 | tail -3; python -c "$P" gpurun_out/r3e_b4k.json; tail -3 gpurun_out/r3e_b4k.err
