"""Golden picture signatures: the reference's own xevd_md5_imgb (src_base/xevd_util.c:985-1002, from oracle/_ref/libxevd_ref.so built out of /root/reference by
oracle/Makefile.ref) on seeded pictures -> tests/golden/md5_pictures.json.  Run in the development container (the reference does not travel):
    python tests/golden/make_md5_golden.py
The pictures are regenerated from (seed, width, height, bit depth) by md5_picture() below - which the tests import - so the fixture holds digests only."""
import ctypes as C
import json
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
CASES = [(1, 8, 8, 8), (2, 200, 136, 10), (3, 328, 200, 10), (4, 72, 40, 8), (5, 1920, 1080, 10), (6, 24, 16, 10), (7, 136, 72, 10)]      # seed, w, h, bit depth


def md5_picture(seed, w, h, bd):
    """[Y, U, V] int16 planes, 4:2:0, values over the whole range of the bit depth"""
    rng = np.random.default_rng(9000 + seed)
    return [rng.integers(0, 1 << bd, (h >> (1 if c else 0), w >> (1 if c else 0)), dtype=np.int16) for c in range(3)]


def reference_digests(planes):
    lib = C.CDLL(os.path.join(HERE, "..", "..", "oracle", "_ref", "libxevd_ref.so"))
    imgb = (C.c_uint8 * 408)()                        # XEVD_IMGB by the offsets of tests/golden/api_layout.txt: np 4, w 8, h 24, x 40, y 56, s 72, a 104
    I32 = lambda off, i, v: C.c_int32.from_buffer(imgb, off + 4 * i).__setattr__("value", v)
    keep = [np.ascontiguousarray(p) for p in planes]
    C.c_int32.from_buffer(imgb, 4).value = 3
    for i, p in enumerate(keep):
        I32(8, i, p.shape[1]); I32(24, i, p.shape[0]); I32(40, i, 0); I32(56, i, 0); I32(72, i, p.shape[1] * 2)
        C.c_void_p.from_buffer(imgb, 104 + 8 * i).value = p.ctypes.data
    dig = (C.c_uint8 * 64)()
    lib.xevd_md5_imgb.restype = C.c_int
    rc = lib.xevd_md5_imgb(C.byref(imgb), C.byref(dig))
    assert rc == 0
    return [bytes(dig[16 * i:16 * i + 16]).hex() for i in range(3)]


if __name__ == "__main__":
    out = {}
    for seed, w, h, bd in CASES:
        out[f"{seed}_{w}x{h}_{bd}b"] = reference_digests(md5_picture(seed, w, h, bd))
    with open(os.path.join(HERE, "md5_pictures.json"), "w") as f:
        json.dump(out, f, indent=1)
    print(json.dumps(out, indent=1))
