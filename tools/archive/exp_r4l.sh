R=$GRAFT_REPO_ROOT
cd $R
echo "== build B: 128 VGPRs + scratch, region path on / off"
timeout -k 5 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -2
timeout -k 5 300 python tools/exp_inter_order.py 16:0 2>&1 | grep inter_us | cut -c1-150
XEVD_HIP_INTER_NO_REGION=1 timeout -k 5 300 python tools/exp_inter_order.py 16:0 2>&1 | grep inter_us | cut -c1-150
(cd xevd_amd/csrc && sed -i 's/ __attribute__((amdgpu_waves_per_eu(4, 4)))//' k_inter.hip && make >/dev/null 2>&1)
echo "== build A: 143 VGPRs (3 waves per SIMD), region path on / off"
timeout -k 5 300 python tools/exp_inter_order.py 16:0 2>&1 | grep inter_us | cut -c1-150
XEVD_HIP_INTER_NO_REGION=1 timeout -k 5 300 python tools/exp_inter_order.py 16:0 2>&1 | grep inter_us | cut -c1-150
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/pmc_x
EXP_STEPS=4 timeout -k 5 200 rocprofv3 --pmc TCC_REQ_sum TCP_TCC_READ_REQ_sum TCC_EA0_RDREQ_sum SQ_WAVES -d $R/gpurun_out/pmc_x -o p -- python $R/tools/exp_inter_order.py 16:0 > /dev/null 2>&1
python $R/tools/rocpd_pmc.py $(find $R/gpurun_out/pmc_x -name "*.db" | head -1) /dev/null | grep -E "k_inter" | cut -c1-110
rm -rf $R/gpurun_out/pmc_x
