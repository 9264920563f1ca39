#!/bin/bash
cd /root/repo
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_parity.py -x -q -k "golden_streams or workload_vs_oracle or golden_pic or pictures" 2>&1 | tail -4
WL=cfg4_main_8k_10b_ra
for rep in 1 2 3; do
  for v in old new noq; do
    if [ $v = old ]; then cp tools/ab/libxevd_hip_old.so xevd_amd/libxevd_hip.so; else cp tools/ab/libxevd_hip_new.so xevd_amd/libxevd_hip.so; fi
    if [ $v = noq ]; then export XEVD_HIP_INTER_NO_QUAD=1; else unset XEVD_HIP_INTER_NO_QUAD; fi
    timeout 600 python bench.py --steps 60 --workload $WL --no-cpu-baseline --no-end-to-end 2>/dev/null | python -c "
import sys, json
d = json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('$v', d['value'], d['ms_per_step'], {k: v['avg_us'] for k, v in d['kernels'].items()})"
  done
done > gpurun_out/ab_quad.log 2>&1
unset XEVD_HIP_INTER_NO_QUAD
cp tools/ab/libxevd_hip_new.so xevd_amd/libxevd_hip.so
cat gpurun_out/ab_quad.log
