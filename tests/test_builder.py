"""The host batch builder (xgpu_batch_create's host half) without a device: xgpu_test_build_batch builds the staging block of a batch in host memory and returns a digest
per array (CU records, CTU starts, TB records, itdq work items, intra records, dependency lists, affine tiles, control points, DMVR sub-blocks, owner map).
Pinned here: the arrays do not depend on the number of builder threads, and they equal the committed digests (tests/golden/builder_digests.json, written by this
file's --write with the builder whose staging blocks the GPU suite decodes bit-exactly: a refactoring of the builder must reproduce them byte for byte)."""
import ctypes as C
import glob
import json
import os
import sys

import numpy as np
import pytest

import golden_io
from xevd_amd import abi, stream

GOLDEN_FILE = os.path.join(golden_io.GOLDEN, "builder_digests.json")


def _lib():
    lib = abi.load()
    lib.xgpu_test_build_batch.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.POINTER(C.c_uint64), C.POINTER(C.c_int), C.POINTER(C.c_double)]
    lib.xgpu_test_build_batch.restype = C.c_int
    return lib


def _build(lib, sp, cb, threads):
    dg, info, ms = (C.c_uint64 * 13)(), (C.c_int * 8)(), C.c_double()
    rc = lib.xgpu_test_build_batch(C.byref(sp), C.byref(cb), threads, dg, info, C.byref(ms))
    assert rc == 0, rc
    # (the coefficient array is the caller's, sent from where it lies: digest 10 is empty; 11..13: the work lists of the three inter launches)
    h = [f"{int(v):016x}" for v in dg]
    return h[:10] + [int(v) for v in info] + h[11:13]


def picture_digests(lib, name, threads):
    case, _ = golden_io.load_picture_case(name)
    sp = abi.make_seq_params(case["w"], case["h"], case["bd"], log2_ctu=case["log2_ctu"], iqt=case["iqt"], admvp=case["admvp"], addb=case["addb"], alf=case["alf"], eipd=case["eipd"])
    cb, keep = abi.make_cu_batch(case["batch"])
    return _build(lib, sp, cb, threads)


def stream_digests(lib, path, threads):
    data = np.load(path)["bytes"].tobytes()
    out = []

    def consume(params, cbs):
        sp = abi.make_seq_params(params["width"], params["height"], params["bit_depth"], iqt=params["iqt"], admvp=params["admvp"], addb=params["addb"], alf=params["tool_alf"],
                                 eipd=params["eipd"], bit_depth_chroma=params["bit_depth_chroma"])
        out.append(_build(lib, sp, cbs, threads))
    for p in stream.iter_stream(data, consume_batch=consume):
        # streams with tool_dmvr expect the backend's refined vectors / decoded luma before the next picture: there is no backend here - fixed stand-ins (the
        # vectors of later pictures then differ from a real decode, deterministically: the builder's input is still a valid batch)
        if p["n_dmvr_sub"]:
            p["dmvr_feedback"](np.zeros((p["n_dmvr_sub"], 2, 2), np.int16))
        if p["needs_ref_luma"]:
            p["set_ref_luma"](p["poc"], np.full((p["height"] + 288, p["width"] + 288), 1 << (p["bit_depth"] - 1), np.int16), 144)
    return out


STREAMS = sorted(glob.glob(os.path.join(golden_io.GOLDEN, "stream_*.npz")))


@pytest.mark.parametrize("name", golden_io.PICTURE_CASES)
def test_builder_picture_cases(name):
    lib = _lib()
    want = json.load(open(GOLDEN_FILE))["pic_" + name]
    for threads in (1, 3):
        assert picture_digests(lib, name, threads) == want, f"{threads} builder thread(s)"


@pytest.mark.parametrize("path", STREAMS, ids=os.path.basename)
def test_builder_stream_pictures(path):
    lib = _lib()
    want = json.load(open(GOLDEN_FILE))[os.path.basename(path)]
    for threads in (1, 4):
        assert stream_digests(lib, path, threads) == want, f"{threads} builder thread(s)"


LARGE = (("intra1080", 1920, 1080, dict(inter_frac=0.0), {}), ("mix1080", 1920, 1080, dict(inter_frac=0.5, bi_frac=0.3, n_refs=(1, 1)), dict(admvp=1)),
         ("eipd4k", 3840, 2160, dict(inter_frac=0.2, eipd=True), dict(eipd=1)), ("htdf1080", 1920, 1080, dict(inter_frac=0.7, n_refs=(1, 0)), {}))


def large_digests(lib, case, threads):
    from xevd_amd import synth
    name, w, h, kw, spkw = case
    b = synth.gen_frame(np.random.default_rng(11), w, h, 10, coded_frac=0.6, qp_range=(22, 37), **kw)
    if name.startswith("htdf"):
        b["htdf_slice_qp"] = 32
    cb, keep = abi.make_cu_batch(b)
    return _build(lib, abi.make_seq_params(w, h, 10, **spkw), cb, threads)


@pytest.mark.parametrize("case", LARGE, ids=[c[0] for c in LARGE])
def test_builder_large_pictures_on_several_threads(case):
    """pictures with enough CUs and dependency nodes (14 k - 58 k) that every parallel section of the builder really runs on several threads: validation and counting,
    owner map, node construction, record scatter and dependency rewrite - same arrays whatever the thread count, equal to the committed digests"""
    lib = _lib()
    want = json.load(open(GOLDEN_FILE))["large_" + case[0]]
    for threads in (1, 4, 7):
        assert large_digests(lib, case, threads) == want, f"{threads} builder thread(s)"


def test_builder_rejects_invalid_batches_without_a_device():
    """the validation pass of the builder (geometry, reference indices, coefficient extents) answers before anything is allocated - also in the host-only shim"""
    lib = _lib()
    case, _ = golden_io.load_picture_case("base_p_8b")
    sp = abi.make_seq_params(case["w"], case["h"], case["bd"])
    for field, value in (("x", 60000), ("log2w", 9), ("coef_off", 1 << 30)):
        b = dict(case["batch"])
        b[field] = b[field].copy()
        b[field][len(b[field]) // 2] = value
        cb, keep = abi.make_cu_batch(b)
        dg, info, ms = (C.c_uint64 * 13)(), (C.c_int * 8)(), C.c_double()
        assert lib.xgpu_test_build_batch(C.byref(sp), C.byref(cb), 2, dg, info, C.byref(ms)) == -101, field


if __name__ == "__main__" and "--write" in sys.argv:
    lib = _lib()
    out = {"pic_" + n: picture_digests(lib, n, 1) for n in golden_io.PICTURE_CASES}
    for p in STREAMS:
        out[os.path.basename(p)] = stream_digests(lib, p, 1)
    for c in LARGE:
        out["large_" + c[0]] = large_digests(lib, c, 1)
    json.dump(out, open(GOLDEN_FILE, "w"), indent=0)
    print(len(out), "entries")


@pytest.mark.parametrize("name", ["base_i_8b", "main_eipd_b_ctu128_constrained_10b", "main_btt_10b", "main_b_ctu128_intra_mix_8b"])
def test_builder_arbitrary_cu_order_inside_ctus(name):
    """The ORDER of a batch carries meaning (a neighbour counts as reconstructed when it comes earlier; with sps_suco_flag also right-hand neighbours): the CUs of every
    CTU in a random order - orders no split tree produces - must build the same arrays on 1 and 4 threads or be refused alike (a right-hand neighbour next to a CU with more
    than 32 units has no place in the record, an intra block copy source may come later), and never crash."""
    lib = _lib()
    case, _ = golden_io.load_picture_case(name)
    b = dict(case["batch"])
    rng = np.random.default_rng(5)
    start = np.asarray(b["ctu_cu_start"])
    perm = np.concatenate([rng.permutation(np.arange(start[k], start[k + 1])) for k in range(len(start) - 1)]).astype(np.int64)
    per_cu = {"x": 1, "y": 1, "log2w": 1, "log2h": 1, "pred_mode": 1, "refi": 2, "mv": 4, "qp": 3, "cbf": 1, "coef_off": 1, "ipm": 2, "ats": 1, "ats_inter": 1, "affine": 1,
              "affine_mv": 12, "dmvr": 1, "tree": 1, "cbf_sub": 1}
    n = len(b["x"])
    for k, width in per_cu.items():
        if b.get(k) is not None and hasattr(b[k], "shape") and b[k].size == n * width:
            b[k] = np.ascontiguousarray(np.asarray(b[k]).reshape(n, -1)[perm].reshape(np.asarray(b[k]).shape))
    sp = abi.make_seq_params(case["w"], case["h"], case["bd"], log2_ctu=case["log2_ctu"], iqt=case["iqt"], admvp=case["admvp"], addb=case["addb"], alf=case["alf"], eipd=case["eipd"])
    cb, keep = abi.make_cu_batch(b)
    out = []
    for threads in (1, 4):
        dg, info, ms = (C.c_uint64 * 13)(), (C.c_int * 8)(), C.c_double()
        rc = lib.xgpu_test_build_batch(C.byref(sp), C.byref(cb), threads, dg, info, C.byref(ms))
        out.append((rc, [int(v) for v in dg][:10] if rc == 0 else None))
    assert out[0] == out[1] and out[0][0] in (0, -101)


def test_builder_several_callers_at_once():
    """pictures built side by side (examples/evc_decode --builders, bench.py's end-to-end leg): four caller threads, each with its own pool of builder threads, inside
    the builder at the same time - every call returns the committed digests of its picture"""
    import threading
    from xevd_amd import synth
    lib = _lib()
    goldens = json.load(open(GOLDEN_FILE))
    jobs = []
    for case in LARGE[:2]:
        name, w, h, kw, spkw = case
        b = synth.gen_frame(np.random.default_rng(11), w, h, 10, coded_frac=0.6, qp_range=(22, 37), **kw)
        cb, keep = abi.make_cu_batch(b)
        jobs.append((goldens["large_" + name], abi.make_seq_params(w, h, 10, **spkw), cb, keep))
    errors = []

    def caller(k):
        try:
            for rep in range(4):
                want, sp, cb, _ = jobs[(k + rep) % len(jobs)]
                if _build(lib, sp, cb, 1 + k % 3) != want:
                    errors.append(f"caller {k}, call {rep}")
        except Exception as e:      # noqa: BLE001 - reported below
            errors.append(f"caller {k}: {e!r}")
    threads = [threading.Thread(target=caller, args=(k,)) for k in range(4)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors


def test_builder_survives_mutated_batches():
    """tests/tools/fuzz_builder.py (a process of its own: a crash must not take the suite along): picture goldens with a few elements of their arrays overwritten -
    out-of-range geometry, modes, reference indices, offsets - are built or refused, nothing else"""
    import subprocess
    r = subprocess.run([sys.executable, os.path.join(os.path.dirname(os.path.abspath(__file__)), "tools", "fuzz_builder.py"), "5", "150"], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=600)
    assert r.returncode == 0 and b"done:" in r.stdout, r.stdout.decode()[-800:]
