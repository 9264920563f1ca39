# what k_dbk does with its time at 1080p: instruction and wait counters, the shipped kernel against the ablation without chroma chains (tools/ab/libxevd_hip_dbk4.so)
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
cp $R/xevd_amd/libxevd_hip.so /tmp/keep.so
for v in base dbk4; do
  if [ $v = base ]; then cp /tmp/keep.so $R/xevd_amd/libxevd_hip.so; else cp $R/tools/ab/libxevd_hip_$v.so $R/xevd_amd/libxevd_hip.so; fi
  echo "== $v"
  for ctrs in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD" "SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_ACTIVE_INST_ANY" "SQ_INST_CYCLES_VMEM_RD SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA"; do
    rm -rf $R/gpurun_out/pmc_x
    timeout -k 5 200 rocprofv3 --pmc $ctrs -d $R/gpurun_out/pmc_x -o p -- python $R/tools/exp_small.py 1920x1080 > /dev/null 2>&1
    python $R/tools/rocpd_pmc.py $(find $R/gpurun_out/pmc_x -name "*.db" | head -1) /dev/null | grep -E "k_dbk" | sed 's/(DbkArgs.*)"/"/' | cut -c1-110
  done
done
cp /tmp/keep.so $R/xevd_amd/libxevd_hip.so
rm -rf $R/gpurun_out/pmc_x
