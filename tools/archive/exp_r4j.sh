R=$GRAFT_REPO_ROOT
cd $R
timeout -k 5 60 $R/tools/ubench/win_bw 2>&1 | grep -E "^seq|wg 2x2 tile  32x32|wg 2x2 tile  64x32"
python - <<'PY'
import bench
wl = bench.WORKLOADS["cfg4_main_8k_10b_ra"]
one, data, _ = bench.write_bench_stream(wl, 17, 12)
open("/tmp/s8k.evc", "wb").write(data); open("/tmp/s8k1.evc", "wb").write(one)
PY
echo "== builder alone, cfg4 synthetic batches"; python tools/build_time.py --bench cfg4_main_8k_10b_ra 1 4 8 2>&1 | tail -1
echo "== builder alone, 8K stream pictures"; python tools/build_time.py /tmp/s8k1.evc 1 4 8 2>&1 | tail -1
XEVD_HIP_BUILD_TRACE=1 python tools/build_time.py /tmp/s8k1.evc 4 2>&1 | grep -E "batch build|intra plan" | tail -9
run() { echo "== $*"; "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print({k:d[k] for k in ('fps_decode_only','parse_ms_per_picture','build_ms_per_picture','cpu_user_s','pictures')})"; }
E="$R/examples/evc_decode --json --keep-units 1"
run $E --workers 1 --tile-threads 16 --build-threads 4 /tmp/s8k.evc /tmp/o.yuv
run $E --workers 1 --tile-threads 16 --build-threads 8 /tmp/s8k.evc /tmp/o.yuv
run $E --workers 1 --tile-threads 12 --build-threads 6 /tmp/s8k.evc /tmp/o.yuv
run $E --workers 2 --tile-threads 8 --build-threads 2 /tmp/s8k.evc /tmp/o.yuv
run $E --workers 2 --tile-threads 8 --build-threads 4 /tmp/s8k.evc /tmp/o.yuv
run env XEVD_HIP_BLOCKING_SYNC=1 $E --workers 2 --tile-threads 8 --build-threads 4 /tmp/s8k.evc /tmp/o.yuv
run env XEVD_HIP_BLOCKING_SYNC=1 $E --workers 3 --tile-threads 5 --build-threads 2 /tmp/s8k.evc /tmp/o.yuv
run env XEVD_HIP_BLOCKING_SYNC=1 $E --workers 4 --tile-threads 4 --build-threads 1 /tmp/s8k.evc /tmp/o.yuv
bash $R/tools/exp_r4i.sh
