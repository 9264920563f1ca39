R=$GRAFT_REPO_ROOT
cd $R
echo "== aligned region fetch"; timeout -k 5 300 python tools/exp_inter_order.py 16:0 16:0 16:0 2>&1 | grep inter_us | cut -c1-30
cp tools/tmp_k_unaligned.hip xevd_amd/csrc/k_inter.hip; (cd xevd_amd/csrc && make >/dev/null 2>&1)
echo "== unaligned (previous commit)"; timeout -k 5 300 python tools/exp_inter_order.py 16:0 16:0 16:0 2>&1 | grep inter_us | cut -c1-30
