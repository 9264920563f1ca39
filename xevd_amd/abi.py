"""ctypes view of include/xevd_hip.h - struct layouts and the loader for the product library.

Plumbing only: the product is xevd_amd/libxevd_hip.so (hand-written HIP for gfx950 behind a C ABI).  This
module never falls back to a CPU path: if the shared library is missing, `load()` raises.
"""
import ctypes as C
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "libxevd_hip.so")

XGPU_MAX_REFS = 17
PAD_L, PAD_C = 144, 72
MODE_INTRA, MODE_INTER, MODE_SKIP, MODE_DIR = 0, 1, 2, 3
K_NAMES = ["itdq", "inter", "dbk_v", "dbk_h", "pad", "intra", "alf", "affine", "dmvr"]
K_COUNT = 9


class SeqParams(C.Structure):
    _fields_ = [
        ("device", C.c_int), ("width", C.c_int), ("height", C.c_int),
        ("bit_depth_luma", C.c_int), ("bit_depth_chroma", C.c_int), ("chroma_format_idc", C.c_int),
        ("log2_ctu", C.c_int), ("tool_iqt", C.c_int), ("tool_admvp", C.c_int), ("tool_addb", C.c_int),
        ("tool_alf", C.c_int), ("max_pics", C.c_int),
        ("chroma_qp_table", C.POINTER(C.c_int8) * 2),
        ("tool_eipd", C.c_int),
    ]


class DraLuts(C.Structure):
    _fields_ = [("luma_inv_scale_lut", C.c_void_p), ("chroma_inv_scale_lut", C.c_void_p * 2)]


class FrameParams(C.Structure):
    _fields_ = [
        ("pic", C.c_int), ("poc", C.c_int), ("num_refp", C.c_int * 2),
        ("refp_pic", (C.c_int * 2) * XGPU_MAX_REFS), ("refp_poc", (C.c_int * 2) * XGPU_MAX_REFS),
        ("qp_u_offset", C.c_int), ("qp_v_offset", C.c_int),
        ("deblock_alpha_offset", C.c_int), ("deblock_beta_offset", C.c_int),
        ("deblock_on", C.c_int), ("alf_on", C.c_int),
    ]


XGPU_MAX_TILE_COLS, XGPU_MAX_TILE_ROWS = 20, 22


class TileGrid(C.Structure):
    _fields_ = [("n_cols", C.c_int), ("n_rows", C.c_int), ("col_bd", C.c_int * (XGPU_MAX_TILE_COLS + 1)), ("row_bd", C.c_int * (XGPU_MAX_TILE_ROWS + 1)),
                ("loop_filter_across_tiles", C.c_int)]


def make_tile_grid(d):
    """{'col_bd': [...], 'row_bd': [...], 'across': 0/1} (borders in CTUs, first 0, last = CTUs per row / column) -> TileGrid"""
    g = TileGrid()
    g.n_cols, g.n_rows = len(d["col_bd"]) - 1, len(d["row_bd"]) - 1
    for i, v in enumerate(d["col_bd"]):
        g.col_bd[i] = int(v)
    for i, v in enumerate(d["row_bd"]):
        g.row_bd[i] = int(v)
    g.loop_filter_across_tiles = int(d.get("across", 0))
    return g


def tile_grid_dict(g):
    return {"col_bd": [g.col_bd[i] for i in range(g.n_cols + 1)], "row_bd": [g.row_bd[i] for i in range(g.n_rows + 1)], "across": int(g.loop_filter_across_tiles)}


class CuBatch(C.Structure):
    _fields_ = [
        ("n_cu", C.c_int),
        ("x", C.POINTER(C.c_uint16)), ("y", C.POINTER(C.c_uint16)),
        ("log2w", C.POINTER(C.c_uint8)), ("log2h", C.POINTER(C.c_uint8)),
        ("pred_mode", C.POINTER(C.c_uint8)),
        ("refi", C.POINTER(C.c_int8)), ("mv", C.POINTER(C.c_int16)),
        ("qp", C.POINTER(C.c_uint8)), ("cbf", C.POINTER(C.c_uint8)), ("cbf_sub", C.POINTER(C.c_uint16)), ("ipm", C.POINTER(C.c_uint8)), ("ats", C.POINTER(C.c_uint8)), ("ats_inter", C.POINTER(C.c_uint8)),
        ("coef_off", C.POINTER(C.c_uint32)), ("coef", C.POINTER(C.c_int16)), ("n_coef", C.c_size_t),
        ("n_ctu", C.c_int), ("ctu_cu_start", C.POINTER(C.c_uint32)), ("constrained_intra_pred", C.c_int),
        ("affine", C.POINTER(C.c_uint8)), ("affine_mv", C.POINTER(C.c_int16)), ("dmvr", C.POINTER(C.c_uint8)), ("htdf_slice_qp", C.c_int),
        ("tiles", C.POINTER(TileGrid)), ("tree", C.POINTER(C.c_uint8)),
    ]


class AlfParams(C.Structure):
    _fields_ = [("enable", C.c_int * 3), ("luma_coef", C.POINTER(C.c_int16)), ("chroma_coef", C.POINTER(C.c_int16)),
                ("ctb_flag", C.POINTER(C.c_uint8)), ("across_tiles", C.c_int), ("tiles", C.POINTER(TileGrid))]


def make_alf_params(d):
    """{'enable': (y,u,v), 'luma_coef': [25][13], 'chroma_coef': [7], 'ctb_flag': [n_ctu] or None, 'across_tiles': 0/1}"""
    keep = {"luma": np.ascontiguousarray(d["luma_coef"], np.int16).reshape(25 * 13),
            "chroma": np.ascontiguousarray(d["chroma_coef"], np.int16).reshape(7),
            "flag": None if d.get("ctb_flag") is None else np.ascontiguousarray(d["ctb_flag"], np.uint8)}
    ap = AlfParams()
    for i in range(3):
        ap.enable[i] = int(d["enable"][i])
    ap.luma_coef = keep["luma"].ctypes.data_as(C.POINTER(C.c_int16))
    ap.chroma_coef = keep["chroma"].ctypes.data_as(C.POINTER(C.c_int16))
    if keep["flag"] is not None:
        ap.ctb_flag = keep["flag"].ctypes.data_as(C.POINTER(C.c_uint8))
    ap.across_tiles = int(d.get("across_tiles", 0))
    if d.get("tiles") is not None:
        keep["tiles"] = make_tile_grid(d["tiles"])
        ap.tiles = C.pointer(keep["tiles"])
    return ap, keep


def _ptr(a, ctype):
    return a.ctypes.data_as(C.POINTER(ctype))


def make_cu_batch(b):
    """dict of numpy arrays (see synth.gen_frame) -> (CuBatch, keepalive)."""
    keep = {
        "x": np.ascontiguousarray(b["x"], np.uint16), "y": np.ascontiguousarray(b["y"], np.uint16),
        "log2w": np.ascontiguousarray(b["log2w"], np.uint8), "log2h": np.ascontiguousarray(b["log2h"], np.uint8),
        "pred_mode": np.ascontiguousarray(b["pred_mode"], np.uint8),
        "refi": np.ascontiguousarray(b["refi"], np.int8), "mv": np.ascontiguousarray(b["mv"], np.int16),
        "qp": np.ascontiguousarray(b["qp"], np.uint8), "cbf": np.ascontiguousarray(b["cbf"], np.uint8),
        "ipm": np.ascontiguousarray(b["ipm"], np.uint8),
        "cbf_sub": None if b.get("cbf_sub") is None else np.ascontiguousarray(b["cbf_sub"], np.uint16),
        "ats": None if b.get("ats") is None else np.ascontiguousarray(b["ats"], np.uint8),
        "ats_inter": None if b.get("ats_inter") is None else np.ascontiguousarray(b["ats_inter"], np.uint8),
        "affine": None if b.get("affine") is None else np.ascontiguousarray(b["affine"], np.uint8),
        "affine_mv": None if b.get("affine") is None else np.ascontiguousarray(b["affine_mv"], np.int16),
        "dmvr": None if b.get("dmvr") is None else np.ascontiguousarray(b["dmvr"], np.uint8),
        "tree": None if b.get("tree") is None else np.ascontiguousarray(b["tree"], np.uint8),
        "coef_off": np.ascontiguousarray(b["coef_off"], np.uint32),
        "coef": np.ascontiguousarray(b["coef"], np.int16),
        "ctu_cu_start": np.ascontiguousarray(b["ctu_cu_start"], np.uint32),
    }
    cb = CuBatch()
    cb.n_cu = len(keep["x"])
    cb.x, cb.y = _ptr(keep["x"], C.c_uint16), _ptr(keep["y"], C.c_uint16)
    cb.log2w, cb.log2h = _ptr(keep["log2w"], C.c_uint8), _ptr(keep["log2h"], C.c_uint8)
    cb.pred_mode = _ptr(keep["pred_mode"], C.c_uint8)
    cb.refi, cb.mv = _ptr(keep["refi"], C.c_int8), _ptr(keep["mv"], C.c_int16)
    cb.qp, cb.cbf, cb.ipm = _ptr(keep["qp"], C.c_uint8), _ptr(keep["cbf"], C.c_uint8), _ptr(keep["ipm"], C.c_uint8)
    if keep["cbf_sub"] is not None:
        cb.cbf_sub = _ptr(keep["cbf_sub"], C.c_uint16)
    if keep["ats"] is not None:
        cb.ats = _ptr(keep["ats"], C.c_uint8)
    if keep["ats_inter"] is not None:
        cb.ats_inter = _ptr(keep["ats_inter"], C.c_uint8)
    if keep["affine"] is not None:
        cb.affine, cb.affine_mv = _ptr(keep["affine"], C.c_uint8), _ptr(keep["affine_mv"], C.c_int16)
    if keep["dmvr"] is not None:
        cb.dmvr = _ptr(keep["dmvr"], C.c_uint8)
    if keep["tree"] is not None:
        cb.tree = _ptr(keep["tree"], C.c_uint8)
    cb.coef_off, cb.coef = _ptr(keep["coef_off"], C.c_uint32), _ptr(keep["coef"], C.c_int16)
    cb.n_coef = len(keep["coef"])
    cb.n_ctu = len(keep["ctu_cu_start"]) - 1
    cb.ctu_cu_start = _ptr(keep["ctu_cu_start"], C.c_uint32)
    cb.constrained_intra_pred = int(b.get("constrained_intra_pred", 0) or 0)
    cb.htdf_slice_qp = int(b.get("htdf_slice_qp", 0) or 0)
    if b.get("tiles") is not None:
        keep["tiles"] = make_tile_grid(b["tiles"])
        cb.tiles = C.pointer(keep["tiles"])
    return cb, keep


def make_seq_params(width, height, bit_depth=8, log2_ctu=6, device=0, iqt=0, admvp=0, addb=0, alf=0, max_pics=4,
                    bit_depth_chroma=None, eipd=0):
    sp = SeqParams()
    sp.device, sp.width, sp.height = device, width, height
    sp.bit_depth_luma = bit_depth
    sp.bit_depth_chroma = bit_depth if bit_depth_chroma is None else bit_depth_chroma
    sp.chroma_format_idc = 1
    sp.log2_ctu = log2_ctu
    sp.tool_iqt, sp.tool_admvp, sp.tool_addb, sp.tool_alf = iqt, admvp, addb, alf
    sp.max_pics = max_pics
    sp.tool_eipd = eipd
    return sp


_EXPORTS = {
    # name: (restype, argtypes)
    "xgpu_open": (C.c_int, [C.POINTER(SeqParams), C.POINTER(C.c_void_p)]),
    "xgpu_close": (None, [C.c_void_p]),
    "xgpu_sync": (C.c_int, [C.c_void_p]),
    "xgpu_last_error": (C.c_char_p, [C.c_void_p]),
    "xgpu_version": (C.c_char_p, []),
    "xgpu_pic_alloc": (C.c_int, [C.c_void_p]),
    "xgpu_pic_free": (C.c_int, [C.c_void_p, C.c_int]),
    "xgpu_pic_upload": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int]),
    "xgpu_pic_download": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int]),
    "xgpu_pic_output_size": (C.c_size_t, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]),
    "xgpu_pic_output": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_size_t]),
    "xgpu_pic_output_async": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_size_t, C.POINTER(C.c_int)]),
    "xgpu_pic_output_wait": (C.c_int, [C.c_void_p, C.c_int]),
    "xgpu_pic_md5": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]),
    "xgpu_host_alloc": (C.c_int, [C.c_void_p, C.c_size_t, C.POINTER(C.c_void_p)]),
    "xgpu_host_free": (None, [C.c_void_p, C.c_void_p]),
    "xgpu_batch_wait_upload": (C.c_int, [C.c_void_p, C.c_void_p]),
    "xgpu_set_builder_threads": (C.c_int, [C.c_void_p, C.c_int]),
    "xgpu_batch_prepare": (C.c_int, [C.c_void_p, C.c_void_p]),
    "xgpu_batch_info": (C.c_int, [C.c_void_p, C.c_void_p, C.POINTER(C.c_int)]),
    "xgpu_batch_dmvr_mvs": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]),
    "xgpu_pic_download_padded": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]),
    "xgpu_pic_upload_padded": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]),
    "xgpu_frame_begin": (C.c_int, [C.c_void_p, C.POINTER(FrameParams)]),
    "xgpu_batch_create": (C.c_int, [C.c_void_p, C.POINTER(CuBatch), C.POINTER(C.c_void_p)]),
    "xgpu_batch_destroy": (None, [C.c_void_p, C.c_void_p]),
    "xgpu_batch_recon": (C.c_int, [C.c_void_p, C.c_void_p]),
    "xgpu_batch_recon_ahead": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p]),
    "xgpu_deblock": (C.c_int, [C.c_void_p]),
    "xgpu_alf": (C.c_int, [C.c_void_p, C.POINTER(AlfParams)]),
    "xgpu_pad": (C.c_int, [C.c_void_p]),
    "xgpu_frame_end": (C.c_int, [C.c_void_p]),
    "xgpu_timing_enable": (C.c_int, [C.c_void_p, C.c_int]),
    "xgpu_timing_reset": (C.c_int, [C.c_void_p]),
    "xgpu_timing_get": (C.c_int, [C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_longlong)]),
    "xgpu_measure_copy_bw": (C.c_int, [C.c_void_p, C.c_size_t, C.c_int, C.POINTER(C.c_double)]),
    "xgpu_test_mc_l": (C.c_int, [C.c_void_p, C.c_void_p] + [C.c_int] * 8 + [C.c_void_p] + [C.c_int] * 3),
    "xgpu_test_mc_c": (C.c_int, [C.c_void_p, C.c_void_p] + [C.c_int] * 8 + [C.c_void_p] + [C.c_int] * 3),
    "xgpu_test_batch_resid": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p]),
    "xgpu_test_build_batch": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.POINTER(C.c_uint64), C.POINTER(C.c_int), C.POINTER(C.c_double)]),
    "xgpu_test_recon": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p] + [C.c_int] * 4 + [C.c_void_p, C.c_int]),
    "xgpu_test_dbk": (C.c_int, [C.c_void_p, C.c_void_p] + [C.c_int] * 7),
    "xgpu_test_dbk_chroma": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p] + [C.c_int] * 8),
    "xgpu_test_itdq": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int]),
}

_lib = None


def exported_names():
    return sorted(_EXPORTS)


def load():
    """Load the product library.  Fails loudly when it has not been built - there is no CPU fallback."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(hipcc --offload-arch=gfx950).  xevd_amd has no CPU fallback.")
        lib = C.CDLL(LIB_PATH)
        for name, (res, args) in _EXPORTS.items():
            fn = getattr(lib, name)
            fn.restype, fn.argtypes = res, args
        _lib = lib
    return _lib
