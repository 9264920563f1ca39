# usage: prof_exp.sh <case> "<counters pass 1>" "<counters pass 2>" ...  -> gpurun_out/exp_<case>_<n>.csv (one --pmc pass each, on tools/exp_inter.py)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
CASE=$1; shift
n=0
for ctrs in "$@"; do
  n=$((n+1))
  rm -rf $R/gpurun_out/pmc_x
  timeout -k 5 200 rocprofv3 --pmc $ctrs -d $R/gpurun_out/pmc_x -o p -- python $R/tools/exp_inter.py $CASE > $R/gpurun_out/pmc_x.log 2>&1
  python $R/tools/rocpd_pmc.py $(find $R/gpurun_out/pmc_x -name "*.db" | head -1) $R/gpurun_out/exp_${CASE}_$n.csv | grep -E "k_inter|k_copy" | cut -c1-150
  rm -rf $R/gpurun_out/pmc_x
done
