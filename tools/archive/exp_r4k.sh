R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for ctrs in "TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_READ_sum" "TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TA_TCP_STATE_READ_sum"; do
  rm -rf $R/gpurun_out/pmc_x
  timeout -k 5 120 rocprofv3 --pmc $ctrs -d $R/gpurun_out/pmc_x -o p -- $R/tools/ubench/win_bw > /dev/null 2>&1
  python $R/tools/rocpd_pmc.py $(find $R/gpurun_out/pmc_x -name "*.db" | head -1) /dev/null | grep -E "<32, 32, 2, true, 2>|<64, 32, 2, true, 2>|k_seq<2, true>" | sed 's/(short const.*)"//' | cut -c1-110
  rm -rf $R/gpurun_out/pmc_x
  EXP_STEPS=4 timeout -k 5 200 rocprofv3 --pmc $ctrs -d $R/gpurun_out/pmc_x -o p -- python $R/tools/exp_inter_order.py 16:0 > /dev/null 2>&1
  python $R/tools/rocpd_pmc.py $(find $R/gpurun_out/pmc_x -name "*.db" | head -1) /dev/null | grep -E "k_inter|k_addb_alf" | cut -c1-110
done
rm -rf $R/gpurun_out/pmc_x
