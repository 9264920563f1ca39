#!/bin/bash
# several builds of the library on one box, alternating: tools/exp_variants.sh "<variant> <variant> ..." [workload] [reps]   (tools/ab/libxevd_hip_<variant>.so, see tools/build_variant.sh)
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
WL=${2:-cfg4_main_8k_10b_ra}
cp xevd_amd/libxevd_hip.so /tmp/libxevd_hip_product.so
for rep in $(seq 1 ${3:-2}); do
  for v in $1; do
    cp tools/ab/libxevd_hip_$v.so xevd_amd/libxevd_hip.so
    timeout 600 python bench.py --steps 60 --workload $WL --no-cpu-baseline --no-end-to-end 2>/dev/null | python -c "
import sys, json
d = json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('%-10s' % '$v', d['value'], d['ms_per_step'], {k: v['avg_us'] for k, v in d['kernels'].items()})"
  done
done 2>&1 | tee gpurun_out/variants_$WL.log
cp /tmp/libxevd_hip_product.so xevd_amd/libxevd_hip.so
