# tests, then tools/exp_inter.py for the default build and for each extra flag set given as arguments
set -x
timeout -k 5 400 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
timeout -k 5 300 python tools/exp_inter.py $EXP_CASES 2>&1 | grep inter_us
for v in "$@"; do
  (cd xevd_amd/csrc && touch k_inter.hip && make CXXFLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-result $v" >/dev/null 2>&1)
  echo "== $v"
  timeout -k 5 300 python tools/exp_inter.py $EXP_CASES 2>&1 | grep inter_us
done
