R=$GRAFT_REPO_ROOT
timeout -k 5 120 $R/tools/ubench/win_bw > $R/gpurun_out/r4b_win_bw2.txt 2>&1; cat $R/gpurun_out/r4b_win_bw2.txt
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/pmc_x
timeout -k 5 200 rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_128B_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum -d $R/gpurun_out/pmc_x -o p -- $R/tools/ubench/win_bw > /dev/null 2>&1
python $R/tools/rocpd_pmc.py $(find $R/gpurun_out/pmc_x -name "*.db" | head -1) $R/gpurun_out/r4b_win_pmc.csv | grep -v "^kernel" | sed 's/void k_win//' | cut -c1-110
rm -rf $R/gpurun_out/pmc_x
