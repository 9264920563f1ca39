R=$GRAFT_REPO_ROOT
cd $R
bash tools/collect_profiles.sh round4_a bb3828d
cd $R
timeout -k 5 600 python tools/bench_multistream.py --streams 1,4,16,32 --pics 48 --profile base > gpurun_out/round4_a_multistream_base1080p.json 2> /dev/null; tail -c 1200 gpurun_out/round4_a_multistream_base1080p.json
timeout -k 5 600 python tools/bench_multistream.py --streams 1,4,16,32 --pics 50 --profile main > gpurun_out/round4_a_multistream_main1080p.json 2> /dev/null; tail -c 1200 gpurun_out/round4_a_multistream_main1080p.json
gcc -O2 -I include -o /tmp/parse_time tools/parse_time.c -L xevd_amd -lxevd_host -Wl,-rpath,$R/xevd_amd
python - <<'PY'
import bench
wl = bench.WORKLOADS["cfg4_main_8k_10b_ra"]
one, data, _ = bench.write_bench_stream(wl, 17, 1)
open("/tmp/s8k1.evc", "wb").write(one)
PY
for t in 1 16; do /tmp/parse_time /tmp/s8k1.evc $t 2 | tail -1; done
