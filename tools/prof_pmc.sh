# usage: prof_pmc.sh <tag> <workload> "<counters>"   -> gpurun_out/<tag>.csv  (one --pmc pass, no other tracing)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf $R/gpurun_out/pmc_$1
timeout -k 5 150 rocprofv3 --pmc $3 -d $R/gpurun_out/pmc_$1 -o p -- python $R/bench.py --steps 6 --warmup 2 --workload $2 --no-cpu-baseline --no-end-to-end > $R/gpurun_out/pmc_$1.log 2>&1
python $R/tools/rocpd_pmc.py $(find $R/gpurun_out/pmc_$1 -name "*.db" | head -1) $R/gpurun_out/$1.csv | cut -c1-150 | head -40
rm -rf $R/gpurun_out/pmc_$1
