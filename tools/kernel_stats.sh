# rocprofv3 kernel statistics of one bench workload.  usage (through gpurun): bash tools/kernel_stats.sh <workload> <tag>   -> gpurun_out/<tag>_<workload>_kernel_stats.csv
R=$GRAFT_REPO_ROOT; wl=$1; T=$2
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/kt_$wl
timeout -k 5 200 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/kt_$wl -o p -- python $R/bench.py --steps 30 --warmup 5 --workload $wl --no-cpu-baseline --no-end-to-end > $R/gpurun_out/kt_$wl.log 2>&1
python $R/tools/rocpd_stats.py $(find $R/gpurun_out/kt_$wl -name "*.db" | head -1) $R/gpurun_out/${T}_${wl}_kernel_stats.csv
rm -rf $R/gpurun_out/kt_$wl
