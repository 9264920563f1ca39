set -x
timeout -k 5 300 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
timeout -k 5 200 python bench.py --steps 100 --warmup 10 --no-cpu-baseline > gpurun_out/b8k.json 2> gpurun_out/b8k.err; tail -c 3000 gpurun_out/b8k.json
timeout -k 5 200 python bench.py --steps 200 --warmup 10 --workload cfg2_base_1080p_8b_ippp --no-cpu-baseline > gpurun_out/b1080.json 2> gpurun_out/b1080.err; tail -c 3000 gpurun_out/b1080.json
