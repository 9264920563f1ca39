# Single-stream decode rate of examples/evc_decode on bench.py's real-bitstream leg (random-access Main, 4x4 tiles) against the number of batch-builder threads.
# usage (through gpurun): bash tools/stream_fps_sweep.sh <workload> "<build-thread counts>"     -> gpurun_out/stream_fps_sweep.txt
R=$GRAFT_REPO_ROOT; WL=${1:-cfg4_main_8k_10b_ra}; BT=${2:-"4 8 16"}
cd $R && python - "$WL" <<'PY'
import sys
import bench
wl = bench.WORKLOADS[sys.argv[1]]
one, data, _ = bench.write_bench_stream(wl, 17, 2, seed=9)
open("/tmp/sweep.evc", "wb").write(data)
PY
: > $R/gpurun_out/stream_fps_sweep.txt
for b in $BT; do
  echo "build threads $b" >> $R/gpurun_out/stream_fps_sweep.txt
  $R/examples/evc_decode --workers 1 --tile-threads 16 --build-threads $b /tmp/sweep.evc /tmp/sweep.yuv 2>&1 | grep -E "pictures/s|stages per picture" >> $R/gpurun_out/stream_fps_sweep.txt
done
cat $R/gpurun_out/stream_fps_sweep.txt
