R=$GRAFT_REPO_ROOT
cd $R && python - <<'PY'
import bench
wl = bench.WORKLOADS["cfg4_main_8k_10b_ra"]
one, data, _ = bench.write_bench_stream(wl, 17, 1, seed=9)
open("/tmp/sweep.evc", "wb").write(data)
PY
XEVD_HIP_BUILD_TRACE=1 $R/examples/evc_decode --workers 1 --tile-threads 16 --build-threads 4 /tmp/sweep.evc /tmp/sweep.yuv 2>&1 | tail -60 > $R/gpurun_out/build_trace.txt
