R=$GRAFT_REPO_ROOT
cd $R
echo "== current (143 VGPRs, residual early)"; timeout -k 5 300 python tools/exp_inter_order.py 16:0 16:0 2>&1 | grep inter_us | cut -c1-60
for v in tmp_k_late tmp_k_late_f; do
  cp tools/$v.hip xevd_amd/csrc/k_inter.hip; (cd xevd_amd/csrc && make >/dev/null 2>&1)
  echo "== $v"; timeout -k 5 300 python tools/exp_inter_order.py 16:0 16:0 2>&1 | grep inter_us | cut -c1-60
done
timeout -k 5 600 python -m pytest tests -m gpu -x -q -k "golden or bench_workload" 2>&1 | tail -2
