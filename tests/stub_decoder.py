"""Test infrastructure: a decoder object with XgpuDecoder's surface that does nothing.  Only tests/test_workqueue.py uses it (through the
XEVD_BENCH_DECODER hook of bench.py) to run bench.py's launcher, work queue and accounting with several ranks on a host without a GPU.
Nothing is measured or decoded with it; the bench line it produces says "decoder": "stub_decoder"."""
import time

import numpy as np

K_NAMES = ("itdq", "inter", "dmvr", "affine", "intra", "dbk_v", "dbk_h", "alf", "pad")


class XgpuDecoder:
    def __init__(self, width, height, bit_depth=8, **kw):
        self.width, self.height, self.bit_depth = width, height, bit_depth
        self._slots = 0
        self.pictures = 0

    def pic_alloc(self):
        self._slots += 1
        return self._slots - 1

    def batch_create(self, batch):
        return object()

    def decode_picture(self, *a, **kw):
        self.pictures += 1
        time.sleep(0.002)

    def timing_get(self):
        return {k: (1.0, 1) for k in K_NAMES}

    def measure_copy_bw(self, *a):
        raise RuntimeError("stub")

    def pic_download_padded(self, pic):
        return [np.zeros((1, 1), np.int16)] * 3

    def __getattr__(self, name):          # every other call of the surface: accepted, no effect
        return lambda *a, **kw: None
