#!/bin/bash
# persistent chain workgroups of the fused launch: parity first, then old/new A/B at 8K and 4K, then the cap swept
cd /root/repo
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -x -q -k "stream or golden or dataflow or intra" > gpurun_out/chain_tests.log 2>&1; tail -3 gpurun_out/chain_tests.log
bash tools/exp_ab.sh cfg4_main_8k_10b_ra
bash tools/exp_ab.sh cfg3_main_4k_10b_ra
for n in 64 128 512 1024 100000; do
  for WL in cfg4_main_8k_10b_ra cfg3_main_4k_10b_ra cfg2_main_1080p_10b_ra; do
  XEVD_HIP_CHAIN_WGS=$n timeout 600 python bench.py --steps 60 --workload $WL --no-cpu-baseline --no-end-to-end 2>/dev/null | python -c "
import sys, json
d = json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('$n $WL', d['value'], d['ms_per_step'], {k: v['avg_us'] for k, v in d['kernels'].items() if 'intra' in k})"
  done
done > gpurun_out/chain_sweep.log 2>&1
cat gpurun_out/chain_sweep.log
