import time, hashlib, numpy as np, sys
sys.path.insert(0, '.')
from xevd_amd.decoder import XgpuDecoder
for w, h, bd in ((1920, 1080, 10), (3840, 2160, 10), (7680, 4320, 10)):
    rng = np.random.default_rng(1)
    planes = [rng.integers(0, 1 << bd, (h >> (1 if c else 0), w >> (1 if c else 0)), dtype=np.int16) for c in range(3)]
    with XgpuDecoder(w, h, bd, device=0) as dec:
        pic = dec.pic_alloc(); dec.pic_upload(pic, planes)
        dec.pic_md5(pic)
        t = time.perf_counter(); got = dec.pic_md5(pic); dt = time.perf_counter() - t
    t = time.perf_counter(); ref = [hashlib.md5(p.tobytes()).digest() for p in planes]; ht = time.perf_counter() - t
    print(f"{w}x{h}: device {dt*1e3:.1f} ms ({planes[0].nbytes/dt/1e6:.0f} MB/s on the luma chain), host (hashlib, one thread, three planes) {ht*1e3:.1f} ms, equal {got == ref}")
