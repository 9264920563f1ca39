cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for e in 0 1; do
rm -rf $R/gpurun_out/pmc_ai$e
XEVD_HIP_INTRA_CTU=$e timeout -k 5 150 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES -d $R/gpurun_out/pmc_ai$e -o p -- python $R/tools/time_all_intra.py --main --reps 2 > $R/gpurun_out/pmc_ai$e.log 2>&1
python $R/tools/rocpd_pmc.py $(find $R/gpurun_out/pmc_ai$e -name "*.db" | head -1) $R/gpurun_out/ai$e.csv | grep -i intra | cut -c1-220 | head -8
tail -1 $R/gpurun_out/pmc_ai$e.log | cut -c60-330
rm -rf $R/gpurun_out/pmc_ai$e
done
