cd $GRAFT_REPO_ROOT
AMD_LOG_LEVEL=1 timeout 300 python - <<PY 2>&1 | grep -v "^$" | tail -25
import torch
torch.cuda.set_device(0)
x = torch.zeros(4, device="cuda")
from xevd_amd.decoder import XgpuDecoder
for sz in ((256, 256), (7680, 4320)):
    try:
        d = XgpuDecoder(sz[0], sz[1], 10, device=0, iqt=True, admvp=True, addb=True, alf=True)
        print("open ok", sz)
    except Exception as e:
        print("ERR", sz, e)
PY
