#!/bin/bash
# usage: tools/build_variant.sh <name> [extra compiler flags]   -> tools/ab/libxevd_hip_<name>.so built from the working tree with the flags (objects under /tmp), for
# A/B runs (tools/exp_ab.sh) and measurement builds (-DXGPU_ALF_TRACE, -DXGPU_INTER_TRACE) that must not replace the product library
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
N=$1; shift
B=/tmp/build_$N; mkdir -p $B $R/tools/ab
cd $R/xevd_amd/csrc
SRCS=$(sed -n 's/^SRCS := //p' Makefile)
for f in $SRCS; do
  ( /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -Wno-unused-result "$@" -c $f -o $B/${f%.hip}.o ) &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $R/tools/ab/libxevd_hip_$N.so $(for f in $SRCS; do echo $B/${f%.hip}.o; done)
ls -la $R/tools/ab/libxevd_hip_$N.so
