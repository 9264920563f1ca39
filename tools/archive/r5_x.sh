# round 5: the host front end under different compiler options, on the GPU box's host CPU (no GPU work): ms per 4K / 8K picture inside xhost_parser_next, 1 and 16 tile threads
cd $GRAFT_REPO_ROOT
lscpu | grep -E "Model name|^CPU\(s\)" ; grep -m1 flags /proc/cpuinfo | tr ' ' '\n' | grep -E "^(avx2|bmi2|abm|avx512f)$" | tr '\n' ' '; echo
python - <<'PY'
import bench
for name, n in (("cfg3_main_4k_10b_ra", 9), ("cfg4_main_8k_10b_ra", 5)):
    one, whole, what = bench.write_bench_stream(bench.WORKLOADS[name], n, 1)
    open(f"/tmp/{name}.evc", "wb").write(whole)
    print(name, len(whole), what[-60:])
PY
cd xevd_amd/host
for v in "A:-O2" "B:-O3 -march=x86-64-v3" "C:-O2 -march=x86-64-v3" "D:-O3"; do
  n=${v%%:*}; fl=${v#*:}; mkdir -p /tmp/lib$n
  g++ $fl -std=c++17 -fPIC -pthread -shared -o /tmp/lib$n/libxevd_host.so evc_parser.cc evc_writer.cc xwq.cc
done
cd ../..
gcc -O2 -I include -o /tmp/parse_time tools/parse_time.c -L xevd_amd -lxevd_host
for s in cfg3_main_4k_10b_ra cfg4_main_8k_10b_ra; do
for th in 1 16; do
for r in 1 2 3; do for n in A B C D; do echo -n "$s threads $th $n: "; LD_LIBRARY_PATH=/tmp/lib$n /tmp/parse_time /tmp/$s.evc $th 3 | tail -1; done; done
done
done
