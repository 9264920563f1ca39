#!/bin/bash
cd /root/repo
cp xevd_amd/libxevd_hip.so /tmp/keep.so
for v in base dbk6; do
  if [ $v = base ]; then cp /tmp/keep.so xevd_amd/libxevd_hip.so; else cp tools/ab/libxevd_hip_$v.so xevd_amd/libxevd_hip.so; fi
  echo "== $v"; timeout 300 python tools/exp_small.py 320x184 1920x1080 2>&1 | tail -2
done
cp /tmp/keep.so xevd_amd/libxevd_hip.so
