#!/bin/bash
# usage: tools/kstat.sh k_addb   -> VGPRs / SGPRs / LDS / scratch of every kernel in xevd_amd/csrc/<name>.hip (device asm in /tmp)
R=$(cd "$(dirname "$0")/.." && pwd)
mkdir -p /tmp/kstat && cd /tmp/kstat
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-function -Wno-unused-result --cuda-device-only -S -I$R/xevd_amd/csrc $R/xevd_amd/csrc/$1.hip -o $1.s || exit 1
python3 - "$1.s" <<'PY'
import re, sys
txt = open(sys.argv[1]).read()
for m in re.finditer(r"\.amdhsa_kernel (\S+)(.*?)\.end_amdhsa_kernel", txt, re.S):
    body = m.group(2)
    g = lambda k: (re.search(r"\.amdhsa_%s (\S+)" % k, body) or [None, "?"])[1]
    print(f"{m.group(1)[:60]:60s} vgpr={g('next_free_vgpr')} sgpr={g('next_free_sgpr')} lds={g('group_segment_fixed_size')} scratch={g('private_segment_fixed_size')} accum_off={g('accum_offset')}")
PY
