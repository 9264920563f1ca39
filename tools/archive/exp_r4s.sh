R=$GRAFT_REPO_ROOT
cd $R
python - <<'PY'
import bench, os
wl = bench.WORKLOADS["cfg3_main_4k_10b_ra"]
for k in range(8):
    one, data, _ = bench.write_bench_stream(wl, 17, 6, seed=100 + k)
    open(f"/tmp/s4k_{k}.evc", "wb").write(data)
PY
A=""; for k in 0 1 2 3 4 5 6 7; do A="$A /tmp/s4k_$k.evc /tmp/o$k.yuv"; done
run() { echo "== $*"; "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print({k:d[k] for k in ('fps_decode_only','fps_wall','parse_ms_per_picture','build_ms_per_picture','cpu_user_s','cpu_sys_s','pictures')})"; }
E="$R/examples/evc_decode --json --keep-units 1"
run $E --workers 8 --tile-threads 1 --build-threads 1 $A
run $E --workers 12 --tile-threads 1 --build-threads 1 $A
run $E --workers 16 --tile-threads 1 --build-threads 1 $A
run $E --workers 8 --tile-threads 2 --build-threads 1 $A
run env XEVD_HIP_BLOCKING_SYNC=1 $E --workers 12 --tile-threads 1 --build-threads 1 $A
run env XEVD_HIP_BLOCKING_SYNC=1 $E --workers 16 --tile-threads 1 --build-threads 1 $A
