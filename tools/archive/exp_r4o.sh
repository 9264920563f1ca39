R=$GRAFT_REPO_ROOT
timeout -k 5 60 $R/tools/ubench/split_pat
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/pmc_x
timeout -k 5 120 rocprofv3 --pmc TA_TA_BUSY_sum TD_TD_BUSY_sum TCP_TOTAL_ACCESSES_sum TCP_TCC_READ_REQ_sum -d $R/gpurun_out/pmc_x -o p -- $R/tools/ubench/split_pat > /dev/null 2>&1
python $R/tools/rocpd_pmc.py $(find $R/gpurun_out/pmc_x -name "*.db" | head -1) /dev/null | grep -v "^kernel" | sed 's/(short const.*)"//' | cut -c1-90
rm -rf $R/gpurun_out/pmc_x
