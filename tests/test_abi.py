"""CPU suite: the product library loads and exports every symbol include/xevd_hip.h declares; argument checks
that do not need a GPU behave like the reference's (negative XEVD_ERR_* codes, nothing thrown)."""
import ctypes as C
import os
import re

import pytest

from xevd_amd import abi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    src = open(os.path.join(ROOT, "include", "xevd_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(xgpu_[a-z0-9_]+)\s*\(", src)))


def test_header_and_binding_agree():
    assert _declared() == abi.exported_names()


def test_library_exports_every_declared_symbol():
    if not os.path.exists(abi.LIB_PATH):
        pytest.fail(f"{abi.LIB_PATH} missing - run __graft_entry__.build()")
    lib = C.CDLL(abi.LIB_PATH)
    for name in _declared():
        assert hasattr(lib, name), name


def test_host_library_exports_every_declared_symbol():
    """include/xevd_host.h (parser / writer) and include/xevd_wq.h (multi-GPU work queue) -> libxevd_host.so"""
    lib = C.CDLL(os.path.join(ROOT, "xevd_amd", "libxevd_host.so"))
    for header, prefix in (("xevd_host.h", "xhost_"), ("xevd_wq.h", "xwq_")):
        src = re.sub(r"/\*.*?\*/", "", open(os.path.join(ROOT, "include", header)).read(), flags=re.S)
        names = sorted(set(re.findall(r"\b(%s[a-z0-9_]+)\s*\(" % prefix, src)))
        assert len(names) >= 5, header
        for name in names:
            if name.endswith("_fn"):
                continue                      # callback types
            assert hasattr(lib, name), name


def test_struct_sizes_match_header():
    # computed by hand from include/xevd_hip.h (LP64)
    assert C.sizeof(abi.SeqParams) == 12 * 4 + 2 * 8 + 8          # ... + tool_eipd + tail padding
    assert C.sizeof(abi.FrameParams) == (4 + 2 * 17 * 2 + 4 + 2) * 4
    assert C.sizeof(abi.CuBatch) == 8 + 15 * 8 + 8 + 8 + 8 + 8 + 16 + 8 + 8 + 8 + 8  # ... + dmvr, htdf_slice_qp (padded), tiles, tree
    assert C.sizeof(abi.TileGrid) == (2 + 21 + 23 + 1) * 4
    assert C.sizeof(abi.AlfParams) == 3 * 4 + 4 + 3 * 8 + 8 + 8                      # enable, pad, three pointers, across_tiles (padded), tiles


def _ref_codes():
    """XEVD_* constants as the reference's own header defines them: tests/golden/api_layout.txt, printed by tests/tools/api_layout_probe.c compiled against
    /root/reference/inc/xevd.h (tests/golden/make_golden.py) - NOT this repository's restatement of them"""
    out = {}
    for ln in open(os.path.join(ROOT, "tests", "golden", "api_layout.txt")):
        f = ln.split()
        if len(f) == 2 and f[0].startswith("XEVD_") and f[1].lstrip("-").isdigit():
            out[f[0]] = int(f[1])
    return out


def _header_defines(path, prefix):
    import re
    txt = open(path).read()
    return {m.group(1): int(m.group(2)) for m in re.finditer(r"#define\s+(" + prefix + r"\w*)\s+\(?(-?\d+)\)?", txt)}


def test_error_codes_equal_the_reference_headers():
    """Every XGPU_ERR_* / XHOST_ERR_* / XWQ_ERR_* code of the C ABI is the reference's XEVD_ERR_* code of the same name (inc/xevd.h:48-77): a binding forwards them
    unchanged.  Pinned against the values printed from the reference's header, and xevd_api.h's own enum against the same list."""
    ref = _ref_codes()
    assert ref["XEVD_ERR_UNSUPPORTED"] == -104 and ref["XEVD_ERR_UNEXPECTED"] == -105      # (what the reference header says today)
    seen = 0
    for hdr, prefix in (("xevd_hip.h", "XGPU_"), ("xevd_host.h", "XHOST_"), ("xevd_wq.h", "XWQ_")):
        for name, val in _header_defines(os.path.join(ROOT, "include", hdr), prefix + "ERR").items():
            suffix = name[len(prefix):]
            want = {"ERR_MALFORMED": "XEVD_ERR_MALFORMED_BITSTREAM"}.get(suffix, "XEVD_" + suffix)
            assert ref[want] == val, (name, val, want, ref[want])
            seen += 1
    assert seen >= 7      # XGPU_ERR, _INVALID_ARGUMENT, _OUT_OF_MEMORY, _UNSUPPORTED, _UNEXPECTED; XHOST_ERR_MALFORMED; XWQ_ERR_INVALID_ARGUMENT, _UNEXPECTED
    assert _header_defines(os.path.join(ROOT, "include", "xevd_hip.h"), "XGPU_OK")["XGPU_OK"] == ref["XEVD_OK"]
    import re
    api = open(os.path.join(ROOT, "include", "xevd_api.h")).read()
    for name, val in re.findall(r"(XEVD_(?:ERR|OK|WARN)\w*)\s*=\s*(-?\d+)", api):
        assert ref[name] == int(val), name


def test_argument_errors_without_gpu():
    lib = abi.load()
    assert lib.xgpu_open(None, None) == -101
    out = C.c_void_p()
    sp = abi.make_seq_params(100, 64)             # width not a multiple of 8
    assert lib.xgpu_open(C.byref(sp), C.byref(out)) == -101 and not out.value
    sp = abi.make_seq_params(128, 64)
    sp.chroma_format_idc = 3
    assert lib.xgpu_open(C.byref(sp), C.byref(out)) == _ref_codes()["XEVD_ERR_UNSUPPORTED"]
    assert lib.xgpu_version().startswith(b"xevd_amd")


def test_no_cpu_fallback_in_product_package():
    """The product never imports the oracle (parity would be void otherwise)."""
    pkg = os.path.join(ROOT, "xevd_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".c", ".cpp")):
                txt = open(os.path.join(dirpath, f), errors="ignore").read()
                assert "liboracle" not in txt and "xevd_oracle" not in txt and "oracle_lib" not in txt, f


def test_plain_c_example_is_built():
    """examples/evc_decode (plain C over include/xevd_host.h + include/xevd_hip.h) links against both libraries; without arguments it only
    prints its usage (no GPU needed)"""
    import subprocess
    exe = os.path.join(ROOT, "examples", "evc_decode")
    if not os.path.exists(exe):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "examples")])
    r = subprocess.run([exe], stderr=subprocess.PIPE, timeout=30)
    assert r.returncode == 2 and b"usage" in r.stderr


def _build_api_lib():
    import subprocess
    lib_path = os.path.join(ROOT, "xevd_amd", "libxevd_amd_api.so")
    if not os.path.exists(lib_path):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "xevd_amd", "compat")])
    return lib_path


def test_public_api_library_exports_the_reference_entry_points():
    """libxevd_amd_api.so (xevd_amd/compat, compiled against include/xevd_api.h - no reference tree needed) exports the six functions of inc/xevd.h:369-374"""
    lib = C.CDLL(_build_api_lib())
    for name in ("xevd_create", "xevd_delete", "xevd_decode", "xevd_pull", "xevd_config", "xevd_info"):
        assert hasattr(lib, name), name


def _probe(header_flag, includes, tmp_path, tag):
    import subprocess
    exe = str(tmp_path / f"probe_{tag}")
    subprocess.check_call(["gcc", "-Wall", "-Werror", f"-DPROBE_HEADER={header_flag}"] + [f"-I{i}" for i in includes]
                          + [os.path.join(ROOT, "tests", "tools", "api_layout_probe.c"), "-L" + os.path.join(ROOT, "xevd_amd"), "-lxevd_amd_api",
                             "-Wl,-rpath," + os.path.join(ROOT, "xevd_amd"), "-o", exe])
    return subprocess.run([exe], stdout=subprocess.PIPE, check=True).stdout.decode()


def test_public_api_header_matches_reference_layout(tmp_path):
    """include/xevd_api.h is this repository's restatement of the reference's public ABI (inc/xevd.h:48-374): every constant's value, every struct's size and
    every field's offset and size - printed by one probe program - equal the committed fixture that the same probe produced from the reference's header
    (tests/golden/api_layout.txt), and, where the reference tree is present, the reference's header itself; the probe also assigns the six entry points to
    pointers of the reference's prototypes"""
    _build_api_lib()
    ours = _probe('"xevd_api.h"', [os.path.join(ROOT, "include")], tmp_path, "ours")
    golden = open(os.path.join(ROOT, "tests", "golden", "api_layout.txt")).read()
    assert ours == golden
    assert ours.count("\n") >= 150 and "prototypes 1" in ours
    gen = os.path.join(ROOT, "oracle", "_ref", "gen")
    if os.path.isdir("/root/reference/inc") and os.path.exists(os.path.join(gen, "xevd_exports.h")):
        assert _probe("<xevd.h>", ["/root/reference/inc", gen], tmp_path, "ref") == ours
