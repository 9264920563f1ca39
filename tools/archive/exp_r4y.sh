#!/bin/bash
cd /root/repo
mkdir -p gpurun_out
for s in main_suco_eipd_i_8b main_suco_p_8b main_suco_tiles_dbk_8b main_suco_quad_all_tools_10b main_suco_btt_all_tools_10b; do
  echo "=== $s"
  timeout 300 python tools/dbg_suco.py $s 2>&1 | grep -v "^  File\|^Extension\|^$" | tail -12
done > gpurun_out/r4y.log 2>&1
cat gpurun_out/r4y.log | cut -c1-300
