R=$GRAFT_REPO_ROOT
cd $R
timeout -k 5 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -2
echo "== aligned in region and wave-uniform paths"; timeout -k 5 300 python tools/exp_inter_order.py 16:0 16:0 16:0 2>&1 | grep inter_us | cut -c1-30
cp tools/tmp_k_prev.hip xevd_amd/csrc/k_inter.hip; (cd xevd_amd/csrc && make >/dev/null 2>&1)
echo "== previous commit (region path only)"; timeout -k 5 300 python tools/exp_inter_order.py 16:0 16:0 16:0 2>&1 | grep inter_us | cut -c1-30
