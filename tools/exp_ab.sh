#!/bin/bash
# A/B on one box: tools/ab/libxevd_hip_old.so against tools/ab/libxevd_hip_new.so, the resident-batch legs of bench.py, alternating
cd /root/repo
mkdir -p gpurun_out
WL=${1:-cfg4_main_8k_10b_ra}
for rep in 1 2 3; do
  for v in old new; do
    cp tools/ab/libxevd_hip_$v.so xevd_amd/libxevd_hip.so
    timeout 600 python bench.py --steps 60 --workload $WL --no-cpu-baseline --no-end-to-end 2>/dev/null | python -c "
import sys, json
d = json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('$v', d['value'], d['ms_per_step'], {k: v['avg_us'] for k, v in d['kernels'].items()})"
  done
done > gpurun_out/ab_$WL.log 2>&1
cp tools/ab/libxevd_hip_new.so xevd_amd/libxevd_hip.so
cat gpurun_out/ab_$WL.log
