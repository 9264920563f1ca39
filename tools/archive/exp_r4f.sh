R=$GRAFT_REPO_ROOT
cd $R
gcc -O2 -I include -o /tmp/parse_time tools/parse_time.c -L xevd_amd -lxevd_host -Wl,-rpath,$R/xevd_amd
python - <<'PY'
import bench
wl = bench.WORKLOADS["cfg4_main_8k_10b_ra"]
one, data, _ = bench.write_bench_stream(wl, 17, 1)
open("/tmp/s8k1.evc", "wb").write(one)
PY
echo "== one parser, 3 threads"; /tmp/parse_time /tmp/s8k1.evc 3 3 | grep pass
echo "== four parsers at once, 3 threads each"; for i in 1 2 3 4; do /tmp/parse_time /tmp/s8k1.evc 3 3 | grep pass | tr '\n' ' ' & done; wait; echo
echo "== eight parsers at once, 2 threads each"; for i in 1 2 3 4 5 6 7 8; do /tmp/parse_time /tmp/s8k1.evc 2 2 | grep "pass 1" | tr '\n' ' ' & done; wait; echo
echo "== sixteen parsers at once, 1 thread each"; for i in $(seq 16); do /tmp/parse_time /tmp/s8k1.evc 1 2 | grep "pass 1" | tr '\n' ' ' & done; wait; echo
echo "== four parsers pinned to distinct cores (taskset), 3 threads each"; for i in 0 1 2 3; do taskset -c $((i*4))-$((i*4+3)) /tmp/parse_time /tmp/s8k1.evc 3 3 | grep "pass 2" | tr '\n' ' ' & done; wait; echo
cat /sys/fs/cgroup/cpu.stat | grep -E "nr_throttled|throttled_usec"
