cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_intra -o p -- python $R/bench.py --steps 30 --warmup 5 --workload cfg2_base_1080p_8b_ippp --no-cpu-baseline > $R/gpurun_out/prof_intra.log 2>&1
find $R/gpurun_out/prof_intra -name "*.db" | head -3
python $R/tools/rocpd_stats.py $(find $R/gpurun_out/prof_intra -name "*.db" | head -1) 2>&1 | cut -c1-160 | head -30
