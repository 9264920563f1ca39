# host side on the GPU box's CPUs (cgroup quota: 16 CPUs): the builder alone, the parser alone, evc_decode in a few shapes
R=$GRAFT_REPO_ROOT
cd $R
gcc -O2 -I include -o /tmp/parse_time tools/parse_time.c -L xevd_amd -lxevd_host -Wl,-rpath,$R/xevd_amd
python - <<'PY'
import bench
wl = bench.WORKLOADS["cfg4_main_8k_10b_ra"]
one, data, _ = bench.write_bench_stream(wl, 17, 4)
open("/tmp/s8k.evc", "wb").write(data); open("/tmp/s8k1.evc", "wb").write(one)
PY
echo "== builder alone, cfg4 synthetic batches"; python tools/build_time.py --bench cfg4_main_8k_10b_ra 1 2 4 8 2>&1 | tail -2
echo "== builder alone, 8K stream pictures"; python tools/build_time.py /tmp/s8k1.evc 1 4 8 2>&1 | tail -4
XEVD_HIP_BUILD_TRACE=1 python tools/build_time.py /tmp/s8k1.evc 4 2>&1 | grep -E "batch build|intra plan" | tail -11
for t in 1 4 16; do /tmp/parse_time /tmp/s8k1.evc $t 2 | tail -1; done
XEVD_HOST_TRACE=1 /tmp/parse_time /tmp/s8k1.evc 16 2>&1 | grep "parse:" | tail -10
run() { echo "== $*"; "$@" 2>&1 | grep -E "pictures/s|stages per picture" | sed 's/(device start-up included)//; s/decoding alone (parsing + kernels + output, the span xevd_app times: app\/xevd_app.c:492-501,612-624; slowest worker)/decode-only/' | cut -c1-260; }
E=$R/examples/evc_decode
run $E --workers 1 --tile-threads 16 --build-threads 4 /tmp/s8k.evc /tmp/o.yuv
run $E --workers 1 --tile-threads 8 --build-threads 4 /tmp/s8k.evc /tmp/o.yuv
run $E --workers 2 --tile-threads 6 --build-threads 2 /tmp/s8k.evc /tmp/o.yuv
run $E --workers 4 --tile-threads 3 --build-threads 1 /tmp/s8k.evc /tmp/o.yuv
run $E --workers 4 --tile-threads 8 --build-threads 4 /tmp/s8k.evc /tmp/o.yuv
