"""Mutated CU batches through the host batch builder (xgpu_test_build_batch: no device): every call must answer XGPU_OK or XGPU_ERR_INVALID_ARGUMENT - a binding
hands the backend arrays it made itself, the validation pass is what stands between them and the kernels.
usage: fuzz_builder.py <seed> [iterations] [picture golden ...]      env FZ_MAXMUT: most mutated elements per batch + 1 (default 4), FZ_LIB: the library to load
Under AddressSanitizer (round 4: 1600 batches with up to 3 and 800 with up to 23 mutated elements, clean):
  hipcc --offload-arch=gfx950 -O1 -g -std=c++17 -fPIC -fsanitize=address -fno-gpu-sanitize -Ixevd_amd/csrc -c xevd_amd/csrc/xgpu_builder.hip -o /tmp/builder_asan.o
  hipcc --offload-arch=gfx950 -shared -fPIC -fsanitize=address -shared-libsan -o /tmp/libxevd_hip_asan.so /tmp/builder_asan.o xevd_amd/csrc/xgpu_api.o xevd_amd/csrc/xgpu_launch.o xevd_amd/csrc/xgpu_shims.o xevd_amd/csrc/k_*.o
  (round 5, with the work lists of k_inter in the builder: 600 batches with up to 3 and 300 with up to 23 mutated elements, clean)
  FZ_LIB=/tmp/libxevd_hip_asan.so LD_PRELOAD=$(find /opt/rocm/lib/llvm -name 'libclang_rt.asan-x86_64.so') ASAN_OPTIONS=detect_leaks=0 python tests/tools/fuzz_builder.py 1"""
import ctypes as C
import os
import sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import golden_io
from xevd_amd import abi


def main():
    abi.LIB_PATH = os.environ.get("FZ_LIB", abi.LIB_PATH)
    lib = abi.load()
    lib.xgpu_test_build_batch.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.POINTER(C.c_uint64), C.POINTER(C.c_int), C.POINTER(C.c_double)]
    rng = np.random.default_rng(int(sys.argv[1]))
    iters = int(sys.argv[2]) if len(sys.argv) > 2 else 400
    names = sys.argv[3:] or list(golden_io.PICTURE_CASES)
    max_mut = int(os.environ.get("FZ_MAXMUT", "4"))
    n_ok = n_rej = 0
    for it in range(iters):
        name = names[rng.integers(len(names))]
        case, _ = golden_io.load_picture_case(name)
        b = {k: (v.copy() if isinstance(v, np.ndarray) else v) for k, v in case["batch"].items()}
        fields = [k for k, v in b.items() if isinstance(v, np.ndarray) and v.size and k != "coef"]
        for _ in range(int(rng.integers(1, max_mut))):
            flat = b[fields[rng.integers(len(fields))]].reshape(-1)
            idx, lim, kind = int(rng.integers(flat.size)), np.iinfo(flat.dtype), rng.integers(4)
            if kind == 0:
                flat[idx] = lim.max
            elif kind == 1:
                flat[idx] = lim.min
            elif kind == 2:
                flat[idx] = rng.integers(lim.min, lim.max, endpoint=True)
            else:
                flat[idx] = np.clip(int(flat[idx]) + int(rng.integers(-8, 9)), lim.min, lim.max)
        sp = abi.make_seq_params(case["w"], case["h"], case["bd"], **{k: case[k] for k in ("iqt", "admvp", "addb", "alf", "eipd") if k in case})
        try:
            cb, keep = abi.make_cu_batch(b)
        except Exception:      # noqa: BLE001 - the Python plumbing refused the arrays before the library saw them
            continue
        dg, info, ms = (C.c_uint64 * 13)(), (C.c_int * 8)(), C.c_double()
        rc = lib.xgpu_test_build_batch(C.byref(sp), C.byref(cb), int(rng.integers(1, 4)), dg, info, C.byref(ms))
        if rc == 0:
            n_ok += 1
        elif rc == -101:
            n_rej += 1
        else:
            print(f"iteration {it} ({name}): xgpu_test_build_batch -> {rc}")
            return 1
    print(f"done: {n_ok} built, {n_rej} refused")
    return 0


if __name__ == "__main__":
    sys.exit(main())
