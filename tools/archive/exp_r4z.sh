#!/bin/bash
cd /root/repo
mkdir -p gpurun_out
bash tools/exp_ab.sh cfg4_main_8k_10b_ra
bash tools/exp_ab.sh cfg2_base_1080p_8b_ippp
timeout 1200 python -m pytest tests/test_gpu_parity.py -x -q -k "golden_streams or reference_parser" 2>&1 | tail -4
