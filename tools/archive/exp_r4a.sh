# round 4, first GPU call: the copy yardstick, the suite, the baseline bench line, k_inter's region-order / occupancy sweep (+ read-request counters for a few points)
set -x
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
timeout -k 5 120 $R/tools/ubench/copy_bw 1024 > $R/gpurun_out/r4a_copy_bw.txt 2>&1; cat $R/gpurun_out/r4a_copy_bw.txt
timeout -k 5 600 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
timeout -k 5 300 python tools/exp_inter_order.py 16:0 8:0 4:0 12:0 24:0 32:0 16:13000 16:27000 8:13000 8:0:1 4:0:1 9:0:1 17:0:1 2>&1 | grep inter_us | cut -c1-200
cd /tmp && export TMPDIR=/tmp
for cfg in 16:0 8:0 16:13000 8:0:1; do
  rm -rf $R/gpurun_out/pmc_x
  EXP_STEPS=4 timeout -k 5 200 rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_128B_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum -d $R/gpurun_out/pmc_x -o p -- python $R/tools/exp_inter_order.py $cfg > /dev/null 2>&1
  echo "== $cfg"; python $R/tools/rocpd_pmc.py $(find $R/gpurun_out/pmc_x -name "*.db" | head -1) $R/gpurun_out/r4a_pmc_$(echo $cfg | tr : _).csv | grep k_inter | cut -c1-120
done
rm -rf $R/gpurun_out/pmc_x
cd $R
timeout -k 5 400 python bench.py > gpurun_out/r4a_bench_cfg4.json 2> gpurun_out/r4a_bench_cfg4.err; tail -c 2500 gpurun_out/r4a_bench_cfg4.json
