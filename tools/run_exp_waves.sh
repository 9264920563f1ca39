# k_inter at a forced occupancy: rebuild with -DINTER_WAVES_PER_SIMD=n on the GPU box and time the 8K workload (usage through gpurun: bash tools/run_exp_waves.sh 5 6)
set -x
P='import json,sys;d=json.load(open(sys.argv[1]));print(sys.argv[1],d["value"],d["kernels"]["inter"],d.get("bit_exact"))'
for n in "$@"; do
  (cd xevd_amd/csrc && touch k_inter.hip && make CXXFLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-result -DINTER_WAVES_PER_SIMD=$n" > /dev/null 2>&1)
  timeout -k 5 300 python bench.py --steps 60 --warmup 10 --no-end-to-end > gpurun_out/exp_inter_waves$n.json 2> /dev/null; python -c "$P" gpurun_out/exp_inter_waves$n.json
done
