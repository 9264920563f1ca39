R=$GRAFT_REPO_ROOT
cd $R
timeout -k 5 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -2
echo "== packed"; timeout -k 5 300 python tools/exp_inter_order.py 16:0 16:0 2>&1 | grep inter_us | cut -c50-120
echo "== scalar"; XEVD_HIP_ADDB_SCALAR=1 timeout -k 5 300 python tools/exp_inter_order.py 16:0 16:0 2>&1 | grep inter_us | cut -c50-120
