"""Test-side loader for the CPU oracle (oracle/liboracle.so) and, when present, the reference harness
(oracle/_ref/libref_harness.so = OUR driver around the real reference functions).  Test infrastructure only."""
import ctypes as C
import os
import subprocess

import numpy as np

from xevd_amd import abi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
ORACLE_SO = os.path.join(ORACLE_DIR, "liboracle.so")
REF_SO = os.path.join(ORACLE_DIR, "_ref", "libxevd_ref.so")
HARNESS_SO = os.path.join(ORACLE_DIR, "_ref", "libref_harness.so")


class OrcPic(C.Structure):
    _fields_ = [("y", C.c_void_p), ("u", C.c_void_p), ("v", C.c_void_p), ("s_l", C.c_int), ("s_c", C.c_int), ("poc", C.c_int)]


class OrcMaps(C.Structure):
    _fields_ = [("map_scu", C.c_void_p), ("map_refi", C.c_void_p), ("map_mv", C.c_void_p), ("map_ats", C.c_void_p), ("w_scu", C.c_int), ("h_scu", C.c_int), ("map_tidx", C.c_void_p)]


class OrcFrame(C.Structure):
    _fields_ = [("cur", OrcPic), ("refp", (OrcPic * 2) * abi.XGPU_MAX_REFS), ("qp_u_offset", C.c_int), ("qp_v_offset", C.c_int)]


def build_oracle():
    if (not os.path.exists(ORACLE_SO) or
            os.path.getmtime(ORACLE_SO) < max(os.path.getmtime(os.path.join(ORACLE_DIR, f)) for f in ("xevd_oracle.c", "xevd_oracle.h"))):
        subprocess.check_call(["make", "-C", ORACLE_DIR, "-s"], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)


_oracle = None
_harness = None
_ref = None


def oracle():
    global _oracle
    if _oracle is None:
        build_oracle()
        lib = C.CDLL(ORACLE_SO)
        lib.orc_mc_l.argtypes = [C.c_void_p] + [C.c_int] * 4 + [C.c_void_p] + [C.c_int] * 6
        lib.orc_mc_c.argtypes = lib.orc_mc_l.argtypes
        lib.orc_itdq.argtypes = [C.c_void_p] + [C.c_int] * 5
        lib.orc_tm.restype = C.POINTER(C.c_int8)
        lib.orc_tm.argtypes = [C.c_int]
        lib.orc_recon.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int]
        lib.orc_dbk_luma.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int]
        lib.orc_dbk_chroma.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]
        lib.orc_recon_batch.argtypes = [C.POINTER(abi.SeqParams), C.POINTER(OrcFrame), C.POINTER(abi.CuBatch), C.POINTER(OrcMaps), C.c_void_p]
        lib.orc_recon_batch_ex.argtypes = [C.POINTER(abi.SeqParams), C.POINTER(OrcFrame), C.POINTER(abi.CuBatch), C.POINTER(OrcMaps), C.c_void_p, C.c_void_p]
        lib.orc_deblock_baseline.argtypes = [C.POINTER(abi.SeqParams), C.POINTER(OrcFrame), C.POINTER(abi.CuBatch), C.POINTER(OrcMaps)]
        lib.orc_deblock_addb.argtypes = [C.POINTER(abi.SeqParams), C.POINTER(OrcFrame), C.POINTER(abi.CuBatch), C.POINTER(OrcMaps), C.c_int, C.c_int]
        lib.orc_alf.argtypes = [C.POINTER(abi.SeqParams), C.POINTER(OrcPic), C.POINTER(abi.AlfParams)]
        lib.orc_pad.argtypes = [C.POINTER(abi.SeqParams), C.POINTER(OrcPic)]
        lib.orc_output_convert.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]
        lib.orc_output_convert.restype = None
        lib.orc_dra_apply.argtypes = [C.c_void_p] * 3 + [C.c_int] * 2 + [C.c_void_p] * 3
        lib.orc_dra_apply.restype = None
        _oracle = lib
    return _oracle


REF_OUTPUT_SO = os.path.join(ORACLE_DIR, "_ref", "libref_output.so")
_ref_output = None


def output_convert(planes, src_bd, dst_bd, crop=(0, 0, 0, 0)):
    """Oracle: cropped + converted picture as the bytes of one .yuv frame (orc_output_convert per plane)."""
    cl, cr, ct, cb = crop
    out = []
    for i, p in enumerate(planes):
        sh = 1 if i else 0
        p = np.ascontiguousarray(p, np.int16)
        h, w = p.shape
        x0, y0, w2, h2 = cl >> sh, ct >> sh, w - ((cl + cr) >> sh), h - ((ct + cb) >> sh)
        dst = np.zeros(w2 * h2 * (1 if dst_bd == 8 else 2), np.uint8)
        oracle().orc_output_convert(p.ctypes.data + 2 * (y0 * w + x0), w, w2, h2, src_bd, dst_bd, dst.ctypes.data)
        out.append(dst)
    return np.concatenate(out)


def ref_output_convert(planes, src_bd, dst_bd):
    """The reference application's imgb_cpy_codec_to_out through oracle/ref_output.c (no crop: the app does not crop)."""
    global _ref_output
    if _ref_output is None:
        _ref_output = C.CDLL(REF_OUTPUT_SO)
        _ref_output.refh_output_convert.argtypes = [C.c_void_p] * 3 + [C.c_int] * 4 + [C.c_void_p]
    y, u, v = (np.ascontiguousarray(p, np.int16) for p in planes)
    h, w = y.shape
    dst = np.zeros((w * h + 2 * (w // 2) * (h // 2)) * (1 if dst_bd == 8 else 2), np.uint8)
    _ref_output.refh_output_convert(y.ctypes.data, u.ctypes.data, v.ctypes.data, w, h, src_bd, dst_bd, dst.ctypes.data)
    return dst


# DRA parameter sets (what a DRA APS signals: xevdm_eco_dra_aps_param, src_main/xevdm_eco.c:2319-2375) used by the DRA tests;
# table index 58 takes the direct chroma-scale branch of xevd_correct_local_chroma_scale, the others the QP-table one
DRA_SETS = {
    "three_ranges_idx58": dict(table_idx=58, in_ranges=[64, 300, 600, 940], scales=[600, 512, 400], cb=520, cr=500),
    "five_ranges_idx40": dict(table_idx=40, in_ranges=[16, 200, 420, 610, 800, 1000], scales=[700, 560, 512, 470, 380], cb=512, cr=540),
    "one_range_idx30": dict(table_idx=30, in_ranges=[1, 1023], scales=[512], cb=480, cr=512),
}


def ref_dra(name, bit_depth=10, planes=None):
    """The reference's xevd_init_dra tables for a parameter set -> (luma_inv, cb_inv, cr_inv); with planes: also the pictures after
    the real xevd_apply_dra_chroma_plane x2 + xevd_apply_dra_luma_plane (tight int16 planes, modified copies returned)."""
    d = DRA_SETS[name]
    luts = np.zeros((3, 1024), np.int32)
    ir, sc = np.array(d["in_ranges"], np.int32), np.array(d["scales"], np.int32)
    out = None
    if planes is None:
        harness().refh_dra(bit_depth, d["table_idx"], len(sc), ir.ctypes.data, sc.ctypes.data, d["cb"], d["cr"], None, None, None, 0, 0, luts.ctypes.data)
    else:
        out = [np.ascontiguousarray(p, np.int16).copy() for p in planes]
        h, w = out[0].shape
        harness().refh_dra(bit_depth, d["table_idx"], len(sc), ir.ctypes.data, sc.ctypes.data, d["cb"], d["cr"],
                           out[0].ctypes.data, out[1].ctypes.data, out[2].ctypes.data, w, h, luts.ctypes.data)
    return (luts[0].copy(), luts[1].copy(), luts[2].copy()), out


def dra_apply(planes, luts):
    """Oracle: DRA sample processing on copies of tight planes."""
    out = [np.ascontiguousarray(p, np.int16).copy() for p in planes]
    h, w = out[0].shape
    l = [np.ascontiguousarray(t, np.int32) for t in luts]
    oracle().orc_dra_apply(out[0].ctypes.data, out[1].ctypes.data, out[2].ctypes.data, w, h, l[0].ctypes.data, l[1].ctypes.data, l[2].ctypes.data)
    return out


def have_ref():
    return os.path.exists(REF_SO) and os.path.exists(HARNESS_SO)


def ref():
    """The real reference library (exported per-block functions)."""
    global _ref
    if _ref is None:
        _ref = C.CDLL(REF_SO, mode=C.RTLD_GLOBAL)
    return _ref


def harness():
    global _harness
    if _harness is None:
        ref()
        lib = C.CDLL(HARNESS_SO)
        lib.refh_recon_batch.argtypes = [C.POINTER(abi.SeqParams), C.POINTER(OrcFrame), C.POINTER(abi.CuBatch), C.POINTER(OrcMaps), C.c_void_p, C.c_int]
        lib.refh_recon_batch_ex.argtypes = [C.POINTER(abi.SeqParams), C.POINTER(OrcFrame), C.POINTER(abi.CuBatch), C.POINTER(OrcMaps), C.c_void_p, C.c_int, C.c_void_p]
        lib.refh_deblock_baseline.argtypes = [C.POINTER(abi.SeqParams), C.POINTER(OrcFrame), C.POINTER(abi.CuBatch), C.POINTER(OrcMaps), C.c_int]
        lib.refh_deblock_addb.argtypes = [C.POINTER(abi.SeqParams), C.POINTER(OrcFrame), C.POINTER(abi.CuBatch), C.POINTER(OrcMaps), C.c_int, C.c_int]
        lib.refh_alf.argtypes = [C.POINTER(abi.SeqParams), C.POINTER(OrcPic), C.POINTER(abi.AlfParams)]
        lib.refh_pad.argtypes = [C.POINTER(abi.SeqParams), C.POINTER(OrcPic)]
        lib.refh_dra.argtypes = [C.c_int] * 3 + [C.c_void_p] * 2 + [C.c_int] * 2 + [C.c_void_p] * 3 + [C.c_int] * 2 + [C.c_void_p]
        _harness = lib
    return _harness


class Picture:
    """Padded planar 4:2:0 s16 picture with the reference's buffer geometry (stride = w + 2*pad)."""

    def __init__(self, width, height, poc=0, planes=None, fill=None):
        self.w, self.h, self.poc = width, height, poc
        self.bufs = []
        for c in range(3):
            pw, ph, pad = (width, height, abi.PAD_L) if c == 0 else (width // 2, height // 2, abi.PAD_C)
            b = np.zeros((ph + 2 * pad, pw + 2 * pad), np.int16)
            if fill is not None:
                b[:] = fill
            if planes is not None:
                b[pad:pad + ph, pad:pad + pw] = planes[c]
            self.bufs.append(b)

    def pad_of(self, c):
        return abi.PAD_L if c == 0 else abi.PAD_C

    def active(self, c):
        pad = self.pad_of(c)
        ph, pw = (self.h, self.w) if c == 0 else (self.h // 2, self.w // 2)
        return self.bufs[c][pad:pad + ph, pad:pad + pw]

    def stride(self, c):
        return self.bufs[c].shape[1]

    def origin_ptr(self, c):
        pad = self.pad_of(c)
        return self.bufs[c].ctypes.data + 2 * (pad * self.stride(c) + pad)

    def orc(self):
        p = OrcPic()
        p.y, p.u, p.v = self.origin_ptr(0), self.origin_ptr(1), self.origin_ptr(2)
        p.s_l, p.s_c, p.poc = self.stride(0), self.stride(1), self.poc
        return p

    def copy(self):
        q = Picture(self.w, self.h, self.poc)
        for c in range(3):
            q.bufs[c][:] = self.bufs[c]
        return q

    def pad_numpy(self):
        """border replication (numpy restatement, used to prepare reference pictures in tests)."""
        for c in range(3):
            pad = self.pad_of(c)
            a = self.active(c)
            self.bufs[c][:] = np.pad(a, pad, mode="edge")


class Maps:
    def __init__(self, width, height):
        self.w_scu, self.h_scu = (width + 3) // 4, (height + 3) // 4
        n = self.w_scu * self.h_scu
        self.map_scu = np.zeros(n, np.uint32)
        self.map_refi = np.full((n, 2), -1, np.int8)
        self.map_mv = np.zeros((n, 2, 2), np.int16)
        self.map_ats = np.zeros(n, np.uint8)

    def orc(self):
        m = OrcMaps()
        m.map_scu, m.map_refi, m.map_mv = self.map_scu.ctypes.data, self.map_refi.ctypes.data, self.map_mv.ctypes.data
        m.map_ats = self.map_ats.ctypes.data
        m.w_scu, m.h_scu = self.w_scu, self.h_scu
        return m


def make_frame(cur, refs, qp_u_offset=0, qp_v_offset=0):
    """refs: {(idx, list): Picture}"""
    fr = OrcFrame()
    fr.cur = cur.orc()
    dummy = cur.orc()
    for i in range(abi.XGPU_MAX_REFS):
        for l in range(2):
            fr.refp[i][l] = refs[(i, l)].orc() if (i, l) in refs else dummy
    fr.qp_u_offset, fr.qp_v_offset = qp_u_offset, qp_v_offset
    return fr
