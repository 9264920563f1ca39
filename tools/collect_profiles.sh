# Round evidence on the GPU box: bench lines (with cpu_baseline), rocprofv3 kernel stats, HBM traffic counters (separate --pmc passes).
# usage (through gpurun): XEVD_COMMIT=$(git rev-parse --short HEAD) ... bash tools/collect_profiles.sh <tag> [commit]     -> gpurun_out/<tag>_*
[ -n "$2" ] && export XEVD_COMMIT=$2
R=$GRAFT_REPO_ROOT; T=$1
mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
for wl in cfg4_main_8k_10b_ra cfg2_base_1080p_8b_ippp cfg3_main_4k_10b_ra; do
  timeout -k 5 400 python $R/bench.py --workload $wl > $R/gpurun_out/${T}_bench_$wl.json 2> $R/gpurun_out/${T}_bench_$wl.err
done
for wl in main_8k_10b_ra_affine30 main_8k_10b_ra_htdf main_8k_10b_ra_dmvr; do      # the Main tools outside BASELINE's configs: their kernels' cost
  timeout -k 5 300 python $R/bench.py --workload $wl --steps 30 --no-end-to-end > $R/gpurun_out/${T}_bench_$wl.json 2> /dev/null
done
for wl in cfg4_main_8k_10b_ra cfg2_base_1080p_8b_ippp; do
  rm -rf $R/gpurun_out/kt_$wl
  timeout -k 5 200 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/kt_$wl -o p -- python $R/bench.py --steps 30 --warmup 5 --workload $wl --no-cpu-baseline --no-end-to-end > $R/gpurun_out/kt_$wl.log 2>&1
  python $R/tools/rocpd_stats.py $(find $R/gpurun_out/kt_$wl -name "*.db" | head -1) $R/gpurun_out/${T}_${wl}_kernel_stats.csv > /dev/null
  rm -rf $R/gpurun_out/kt_$wl
done
for ctr in FETCH_SIZE WRITE_SIZE; do
  bash $R/tools/prof_pmc.sh ${T}_main8k_pmc_$ctr cfg4_main_8k_10b_ra $ctr > /dev/null
done
# exact HBM traffic: the request counters carry their size (tools/make_pmc_json.py)
bash $R/tools/prof_pmc.sh ${T}_main8k_pmc_rdreq cfg4_main_8k_10b_ra "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum" > /dev/null
bash $R/tools/prof_pmc.sh ${T}_main8k_pmc_wrreq cfg4_main_8k_10b_ra "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum" > /dev/null
bash $R/tools/prof_pmc.sh ${T}_main8k_pmc_sq cfg4_main_8k_10b_ra "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" > /dev/null
bash $R/tools/prof_pmc.sh ${T}_main8k_pmc_sq2 cfg4_main_8k_10b_ra "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" > /dev/null
python $R/tools/make_pmc_json.py cfg4_main_8k_10b_ra $R/gpurun_out/${T}_main8k_pmc_rdreq.csv $R/gpurun_out/${T}_main8k_pmc_wrreq.csv $R/gpurun_out/${T}_pmc.json \
       $R/gpurun_out/${T}_main8k_pmc_FETCH_SIZE.csv $R/gpurun_out/${T}_main8k_pmc_WRITE_SIZE.csv > /dev/null
ls -la $R/gpurun_out | grep $T
