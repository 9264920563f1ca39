# where k_inter's re-fetched reference lines come from: read / write requests per CU-size class (tools/exp_inter.py cases, 50 % bi-predicted)
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for c in all64 all32 all16 all8 all4 mix; do
  rm -rf $R/gpurun_out/pmc_x
  EXP_TAGS=bi50 timeout -k 5 200 rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_128B_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum -d $R/gpurun_out/pmc_x -o p -- python $R/tools/exp_inter.py $c > $R/gpurun_out/r4i_$c.log 2>&1
  echo "== $c $(grep inter_us $R/gpurun_out/r4i_$c.log)"; python $R/tools/rocpd_pmc.py $(find $R/gpurun_out/pmc_x -name "*.db" | head -1) $R/gpurun_out/r4i_pmc_$c.csv | grep k_inter | cut -c1-100
done
rm -rf $R/gpurun_out/pmc_x
