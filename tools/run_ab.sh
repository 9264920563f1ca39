# A/B of k_inter build variants on the GPU box: tests, then the 8K bench per variant
set -x
timeout -k 5 400 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
timeout -k 5 200 python bench.py --steps 100 --warmup 10 --no-cpu-baseline > gpurun_out/ab_a.json 2> gpurun_out/ab_a.err; tail -c 1500 gpurun_out/ab_a.json
for v in "$@"; do
  (cd xevd_amd/csrc && touch k_inter.hip && make CXXFLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-result $v" >/dev/null 2>&1)
  timeout -k 5 200 python bench.py --steps 100 --warmup 10 --no-cpu-baseline > "gpurun_out/ab_$v.json" 2> gpurun_out/ab_v.err; tail -c 1500 "gpurun_out/ab_$v.json"
done
