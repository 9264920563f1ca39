#!/bin/bash
cd /root/repo
bash tools/collect_profiles.sh round4_c c5fa5d5
python bench.py --gpus 2 --steps 40 > gpurun_out/round4_c_bench_gpus2_on_one_device.json 2> /dev/null
bash tools/exp_multistream.sh round4_c 2>/dev/null || true
