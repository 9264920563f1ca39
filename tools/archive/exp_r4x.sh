#!/bin/bash
# SUCO: the GPU suite (new golden streams among it) and the default bench line
cd /root/repo
mkdir -p gpurun_out
timeout 2400 python -m pytest tests -m gpu -x -q > gpurun_out/r4x_gpu_tests.log 2>&1; echo "pytest exit $?" >> gpurun_out/r4x_gpu_tests.log
tail -5 gpurun_out/r4x_gpu_tests.log
timeout 600 python bench.py --steps 60 > gpurun_out/r4x_bench.json 2> gpurun_out/r4x_bench.err; tail -c 1500 gpurun_out/r4x_bench.json
