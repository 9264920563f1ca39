# host-side contention experiment: one 8K Main stream (4 closed GOPs), evc_decode in several shapes and malloc settings, the builder's phase trace
R=$GRAFT_REPO_ROOT
cd $R && python - <<'PY'
import bench
wl = bench.WORKLOADS["cfg4_main_8k_10b_ra"]
one, data, _ = bench.write_bench_stream(wl, 17, 4)
open("/tmp/s8k.evc", "wb").write(data)
PY
ls -la /tmp/s8k.evc
run() { echo "== $*"; env "$@" 2>&1 | grep -E "pictures/s|stages per picture" | cut -c1-300; }
E=$R/examples/evc_decode
run $E --workers 1 --tile-threads 16 /tmp/s8k.evc /tmp/o.yuv
run $E --workers 4 --tile-threads 8 /tmp/s8k.evc /tmp/o.yuv
run MALLOC_MMAP_THRESHOLD_=4294967295 MALLOC_TRIM_THRESHOLD_=4294967295 MALLOC_TOP_PAD_=268435456 $E --workers 4 --tile-threads 8 /tmp/s8k.evc /tmp/o.yuv
run MALLOC_ARENA_MAX=1 $E --workers 4 --tile-threads 8 /tmp/s8k.evc /tmp/o.yuv
run MALLOC_MMAP_THRESHOLD_=4294967295 MALLOC_TRIM_THRESHOLD_=4294967295 MALLOC_TOP_PAD_=268435456 $E --workers 1 --tile-threads 16 /tmp/s8k.evc /tmp/o.yuv
run $E --workers 2 --tile-threads 16 /tmp/s8k.evc /tmp/o.yuv
run $E --workers 4 --tile-threads 8 --build-threads 1 /tmp/s8k.evc /tmp/o.yuv
run $E --workers 4 --tile-threads 2 --build-threads 2 /tmp/s8k.evc /tmp/o.yuv
# four processes, one GOP-set each (the same stream four times): contention inside one process or on the host?
( for i in 1 2 3 4; do $E --workers 1 --tile-threads 8 /tmp/s8k.evc /tmp/o$i.yuv 2>&1 | grep -E "pictures/s" | cut -c1-200 & done; wait )
XEVD_HIP_BUILD_TRACE=1 $E --workers 1 --tile-threads 16 --build-threads 4 /tmp/s8k.evc /tmp/o.yuv 2>&1 | grep -E "batch build|intra plan" | tail -40 > $R/gpurun_out/r4c_build_trace.txt
tail -28 $R/gpurun_out/r4c_build_trace.txt
lscpu | grep -E "Model name|Socket|NUMA|^CPU\(s\)" 
