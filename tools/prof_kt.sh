# usage: prof_kt.sh <workload>  -> per-kernel stats of a short bench run (rocprofv3 --kernel-trace)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf $R/gpurun_out/kt
timeout -k 5 200 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/kt -o p -- python $R/bench.py --steps 30 --warmup 5 --workload $1 --no-cpu-baseline > $R/gpurun_out/kt.log 2>&1
python $R/tools/rocpd_stats.py $(find $R/gpurun_out/kt -name "*.db" | head -1) | cut -c1-110
rm -rf $R/gpurun_out/kt
