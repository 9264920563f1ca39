#!/bin/bash
# usage: tools/gpu.sh <timeout-seconds> '<command>'   - gpurun with retries while all GPU slots of the pod are busy (nothing is charged for those)
for i in $(seq 1 20); do
  out=$(/usr/local/graft/bin/gpurun --timeout "$1" -- "$2" 2>&1)
  if echo "$out" | grep -q "status=transient"; then sleep 45; else echo "$out"; exit 0; fi
done
echo "$out"; exit 3
