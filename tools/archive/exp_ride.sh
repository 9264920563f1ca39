#!/bin/bash
# where the next picture's residual pass runs (XEVD_HIP_RIDE 0 / 1 / 2), alternating on one box
cd /root/repo
mkdir -p gpurun_out
for rep in 1 2; do
  for WL in cfg4_main_8k_10b_ra cfg3_main_4k_10b_ra cfg2_base_1080p_8b_ippp; do
    for m in 0 1 2; do
      XEVD_HIP_RIDE=$m timeout 600 python bench.py --steps 60 --workload $WL --no-cpu-baseline --no-end-to-end 2>/dev/null | python -c "
import sys, json
d = json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('ride $m $WL', d['value'], d['ms_per_step'], d.get('bit_exact'), d.get('two_contexts'))"
    done
  done
done > gpurun_out/ride.log 2>&1
cat gpurun_out/ride.log
