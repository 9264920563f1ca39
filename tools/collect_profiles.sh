# Round evidence on the GPU box: bench lines (with cpu_baseline), rocprofv3 kernel stats, HBM traffic counters (separate --pmc passes).
# usage (through gpurun): bash tools/collect_profiles.sh <tag>      -> gpurun_out/<tag>_*
R=$GRAFT_REPO_ROOT; T=$1
cd /tmp && export TMPDIR=/tmp
for wl in cfg4_main_8k_10b_ra cfg2_base_1080p_8b_ippp cfg3_main_4k_10b_ra; do
  timeout -k 5 240 python $R/bench.py --workload $wl > $R/gpurun_out/${T}_bench_$wl.json 2> $R/gpurun_out/${T}_bench_$wl.err
done
for wl in main_8k_10b_ra_affine30 main_8k_10b_ra_htdf; do      # the Main tools outside BASELINE's configs: their kernels' cost
  timeout -k 5 200 python $R/bench.py --workload $wl --no-cpu-baseline --steps 30 > $R/gpurun_out/${T}_bench_$wl.json 2> /dev/null
done
for wl in cfg4_main_8k_10b_ra cfg2_base_1080p_8b_ippp; do
  rm -rf $R/gpurun_out/kt_$wl
  timeout -k 5 200 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/kt_$wl -o p -- python $R/bench.py --steps 30 --warmup 5 --workload $wl --no-cpu-baseline > $R/gpurun_out/kt_$wl.log 2>&1
  python $R/tools/rocpd_stats.py $(find $R/gpurun_out/kt_$wl -name "*.db" | head -1) $R/gpurun_out/${T}_${wl}_kernel_stats.csv > /dev/null
  rm -rf $R/gpurun_out/kt_$wl
done
for ctr in FETCH_SIZE WRITE_SIZE; do
  bash $R/tools/prof_pmc.sh ${T}_main8k_pmc_$ctr cfg4_main_8k_10b_ra $ctr > /dev/null
done
ls -la $R/gpurun_out | grep $T
