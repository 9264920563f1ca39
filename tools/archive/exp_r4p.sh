#!/bin/bash
# how much of the frame time is GPU idle that a second, independent picture could fill: two processes on one device, each cycling its own resident batches
cd /root/repo
mkdir -p gpurun_out
one() { timeout 600 python bench.py --steps 6000 --warmup 50 --workload $1 --no-cpu-baseline --no-end-to-end 2>/dev/null | python -c "
import sys, json
d = json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('$2', d['value'], d['ms_per_step'])"; }
for WL in cfg4_main_8k_10b_ra cfg3_main_4k_10b_ra; do
  echo "== $WL"
  one $WL alone
  one $WL A & one $WL B & wait
  one $WL A & one $WL B & one $WL C & wait
done > gpurun_out/two_procs.log 2>&1
cat gpurun_out/two_procs.log
