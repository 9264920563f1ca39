# rocprofv3 kernel stats of the two Main-tool workloads outside BASELINE's configs (affine, HTDF).  usage (through gpurun): bash tools/prof_tools_workloads.sh <tag>
R=$GRAFT_REPO_ROOT; T=$1
cd /tmp && export TMPDIR=/tmp
for wl in main_8k_10b_ra_affine30 main_8k_10b_ra_htdf; do
  rm -rf $R/gpurun_out/kt_$wl
  timeout -k 5 200 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/kt_$wl -o p -- python $R/bench.py --steps 20 --warmup 5 --workload $wl --no-cpu-baseline > $R/gpurun_out/kt_$wl.log 2>&1
  python $R/tools/rocpd_stats.py $(find $R/gpurun_out/kt_$wl -name "*.db" | head -1) $R/gpurun_out/${T}_${wl}_kernel_stats.csv > /dev/null
  rm -rf $R/gpurun_out/kt_$wl
done
ls -la $R/gpurun_out | grep ${T}_main_8k
