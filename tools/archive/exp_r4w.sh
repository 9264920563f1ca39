#!/bin/bash
cd /root/repo
bash tools/collect_profiles.sh round4_d e87e7b8
python bench.py --gpus 2 --steps 40 > gpurun_out/round4_d_bench_gpus2_on_one_device.json 2> /dev/null
