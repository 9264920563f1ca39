# round 3, GPU call i: padding written by k_alf's border tiles, residual pass ahead (xgpu_batch_prepare), copy kernel
set -x
mkdir -p gpurun_out
timeout -k 5 1500 python -m pytest tests -m gpu -q -x 2>&1 | tail -6
P='import json,sys;d=json.load(open(sys.argv[1]));print(d["value"],d["kernels"],d["whole_frame"],d.get("bit_exact"),d["roofline"]["measured_copy_bw_gbps"])'
timeout -k 5 300 python bench.py --steps 100 --warmup 10 --no-end-to-end > gpurun_out/r3i_b8k.json 2> gpurun_out/r3i_b8k.err; python -c "$P" gpurun_out/r3i_b8k.json
timeout -k 5 300 python bench.py --steps 100 --warmup 10 --no-end-to-end --no-cpu-baseline --no-prepare > gpurun_out/r3i_b8k_noprep.json 2> gpurun_out/r3i_b8k_noprep.err; python -c "$P" gpurun_out/r3i_b8k_noprep.json
timeout -k 5 300 python bench.py --steps 100 --warmup 10 --no-end-to-end --no-cpu-baseline --workload cfg3_main_4k_10b_ra > gpurun_out/r3i_b4k.json 2> gpurun_out/r3i_b4k.err; python -c "$P" gpurun_out/r3i_b4k.json
timeout -k 5 300 python bench.py --steps 100 --warmup 10 --no-end-to-end --no-cpu-baseline --workload cfg3_main_4k_10b_ra --no-prepare > gpurun_out/r3i_b4k_noprep.json 2> gpurun_out/r3i_b4k_noprep.err; python -c "$P" gpurun_out/r3i_b4k_noprep.json
