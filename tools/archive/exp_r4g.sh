R=$GRAFT_REPO_ROOT
cd $R
timeout -k 5 600 python -m pytest tests -m gpu -x -q -k "decoder or stream or golden_streams or plain_c or compat or app or work_queue" 2>&1 | tail -3
python - <<'PY'
import bench
wl = bench.WORKLOADS["cfg4_main_8k_10b_ra"]
one, data, _ = bench.write_bench_stream(wl, 17, 16)
open("/tmp/s8k.evc", "wb").write(data)
PY
run() { echo "== $*"; ( time env "$@" ) 2>&1 | grep -E "pictures/s|stages per picture|^real|^user|^sys" | sed 's/(device start-up included)//; s/decoding alone (parsing + kernels + output, the span xevd_app times: app\/xevd_app.c:492-501,612-624; slowest worker)/decode-only/' | cut -c1-220; }
E="$R/examples/evc_decode --keep-units 1"
run $E --workers 1 --tile-threads 16 --build-threads 4 /tmp/s8k.evc /tmp/o.yuv
run $E --workers 1 --tile-threads 12 --build-threads 4 /tmp/s8k.evc /tmp/o.yuv
run $E --workers 2 --tile-threads 8 --build-threads 2 /tmp/s8k.evc /tmp/o.yuv
run $E --workers 4 --tile-threads 4 --build-threads 2 /tmp/s8k.evc /tmp/o.yuv
run $E --workers 4 --tile-threads 3 --build-threads 1 /tmp/s8k.evc /tmp/o.yuv
run $E --workers 8 --tile-threads 2 --build-threads 1 /tmp/s8k.evc /tmp/o.yuv
run $E --workers 1 --tile-threads 16 --build-threads 4 --no-pipeline /tmp/s8k.evc /tmp/o.yuv
grep -E "nr_throttled|throttled_usec" /sys/fs/cgroup/cpu.stat
