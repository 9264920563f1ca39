"""ctypes plumbing for the host front end (xevd_amd/libxevd_host.so, include/xevd_host.h): an EVC Baseline bitstream
writer for synthetic CU batches and the parser that turns a .evc byte string back into CU batches + picture parameters.
No arithmetic here."""
import ctypes as C
import os

import numpy as np

from . import abi

_HERE = os.path.dirname(os.path.abspath(__file__))
HOST_SO = os.path.join(_HERE, "libxevd_host.so")
SLICE_B, SLICE_P, SLICE_I = 0, 1, 2


class StreamParams(C.Structure):
    _fields_ = [("width", C.c_int), ("height", C.c_int), ("bit_depth", C.c_int), ("max_num_ref_pics", C.c_int), ("log2_sub_gop_length", C.c_int),
                ("qp_u_offset", C.c_int), ("qp_v_offset", C.c_int), ("deblock_on", C.c_int), ("cu_qp_delta", C.c_int),
                ("profile_main", C.c_int), ("tool_iqt", C.c_int), ("tool_ats", C.c_int), ("tool_addb", C.c_int),
                ("deblock_alpha_offset", C.c_int), ("deblock_beta_offset", C.c_int), ("tool_alf", C.c_int), ("tool_eipd", C.c_int),
                ("crop", C.c_int * 4), ("tool_dra", C.c_int), ("dra_aps_id", C.c_int), ("cqt_present", C.c_int), ("cqt_same", C.c_int), ("cqt_global_offset", C.c_int),
                ("cqt_num_points", C.c_int * 2), ("cqt_delta_in", (C.c_int * 16) * 2), ("cqt_delta_out", (C.c_int * 16) * 2), ("tool_htdf", C.c_int), ("tool_admvp", C.c_int), ("tool_mmvd", C.c_int), ("tool_dmvr", C.c_int), ("tool_amvr", C.c_int), ("tool_hmvp", C.c_int), ("ibc_log_max_size", C.c_int),
                ("tile_cols", C.c_int), ("tile_rows", C.c_int), ("tile_col_w", C.c_int * abi.XGPU_MAX_TILE_COLS), ("tile_row_h", C.c_int * abi.XGPU_MAX_TILE_ROWS),
                ("loop_filter_across_tiles", C.c_int), ("tool_affine", C.c_int), ("cu_qp_delta_area", C.c_int), ("tool_rpl", C.c_int), ("tool_pocs", C.c_int), ("tool_cm_init", C.c_int), ("tool_adcc", C.c_int),
                ("btt", C.c_int), ("btt_log2_min_cb", C.c_int), ("btt_diff_max_14", C.c_int), ("btt_diff_max_tt", C.c_int), ("btt_diff_min_tt", C.c_int), ("rpl_in_sps", C.c_int),
                ("suco", C.c_int), ("suco_diff_max", C.c_int), ("suco_diff_min", C.c_int)]


class AlfAps(C.Structure):
    _fields_ = [("aps_id", C.c_int), ("luma_present", C.c_int), ("chroma_present", C.c_int), ("luma_type_7x7", C.c_int),
                ("num_luma_filters", C.c_int), ("delta_idx", C.c_uint8 * 25), ("coef_delta_flag", C.c_int), ("pred_mode_flag", C.c_int),
                ("filter_coef_flag", C.c_uint8 * 25), ("luma_coef", (C.c_int16 * 12) * 25), ("chroma_coef", C.c_int16 * 6),
                ("fixed_filter_pattern", C.c_int), ("fixed_filter_usage", C.c_uint8 * 25), ("fixed_filter_idx", C.c_uint8 * 25)]


class SliceAlf(C.Structure):
    _fields_ = [("alf_on", C.c_int), ("aps_id_y", C.c_int), ("aps_id_ch", C.c_int), ("chroma_idc", C.c_int), ("ctb_map", C.c_int),
                ("ctb_flag", C.POINTER(C.c_uint8))]


class SliceDesc(C.Structure):
    _fields_ = [("first_tile", C.c_int), ("last_tile", C.c_int), ("slice_qp", C.c_int), ("deblock_on", C.c_int)]


class DraAps(C.Structure):
    _fields_ = [("aps_id", C.c_int), ("num_ranges", C.c_int), ("in_ranges", C.c_int * 33), ("scale", C.c_int * 32), ("cb_scale", C.c_int),
                ("cr_scale", C.c_int), ("table_idx", C.c_int)]


class HostPicture(C.Structure):
    _fields_ = [("width", C.c_int), ("height", C.c_int), ("bit_depth_luma", C.c_int), ("bit_depth_chroma", C.c_int),
                ("poc", C.c_int), ("temporal_id", C.c_int), ("slice_type", C.c_int), ("is_idr", C.c_int), ("is_ref", C.c_int),
                ("num_refp", C.c_int * 2), ("refp_poc", (C.c_int * 2) * abi.XGPU_MAX_REFS),
                ("slice_qp", C.c_int), ("qp_u_offset", C.c_int), ("qp_v_offset", C.c_int), ("deblock_on", C.c_int),
                ("profile_main", C.c_int), ("tool_iqt", C.c_int), ("tool_ats", C.c_int), ("tool_addb", C.c_int),
                ("deblock_alpha_offset", C.c_int), ("deblock_beta_offset", C.c_int),
                ("tool_alf", C.c_int), ("tool_eipd", C.c_int), ("tool_admvp", C.c_int), ("crop", C.c_int * 4), ("chroma_qp_table", C.POINTER(C.c_int8) * 2),
                ("dra_lut", C.POINTER(C.c_int32) * 3),
                ("alf_on", C.c_int), ("alf", abi.AlfParams),
                ("has_md5", C.c_int), ("md5", (C.c_uint8 * 16) * 3),
                ("n_dmvr_sub", C.c_int), ("needs_ref_luma", C.c_int), ("n_release", C.c_int), ("release_poc", C.c_int * 32), ("batch", abi.CuBatch)]


_lib = None


def load():
    global _lib
    if _lib is None:
        if not os.path.exists(HOST_SO):
            raise RuntimeError(f"{HOST_SO} is missing: build it with `make -C xevd_amd/host` (or __graft_entry__.build())")
        lib = C.CDLL(HOST_SO)
        lib.xhost_parser_open.restype = C.c_void_p
        lib.xhost_parser_open.argtypes = [C.c_void_p, C.c_size_t]
        lib.xhost_parser_next.argtypes = [C.c_void_p, C.POINTER(HostPicture)]
        lib.xhost_parser_error.restype = C.c_char_p
        lib.xhost_parser_error.argtypes = [C.c_void_p]
        lib.xhost_parser_close.argtypes = [C.c_void_p]
        lib.xhost_parser_set_dmvr_mvs.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
        lib.xhost_parser_set_ref_luma.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int]
        lib.xhost_writer_set_ref_luma.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int]
        lib.xhost_parser_set_ref_luma_wait.argtypes = [C.c_void_p, C.c_int]
        lib.xhost_parser_cancel_wait.argtypes = [C.c_void_p]
        lib.xhost_parser_cancel_wait.restype = None
        lib.xhost_parser_set_threads.argtypes = [C.c_void_p, C.c_int]
        lib.xhost_writer_split_allowed.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int)]
        lib.xhost_writer_open.restype = C.c_void_p
        lib.xhost_writer_open.argtypes = [C.POINTER(StreamParams)]
        lib.xhost_writer_add_picture.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(abi.CuBatch)]
        lib.xhost_writer_bytes.argtypes = [C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_size_t)]
        lib.xhost_writer_close.argtypes = [C.c_void_p]
        lib.xhost_writer_add_alf_aps.argtypes = [C.c_void_p, C.POINTER(AlfAps)]
        lib.xhost_writer_add_dra_aps.argtypes = [C.c_void_p, C.POINTER(DraAps)]
        lib.xhost_writer_add_md5_sei.argtypes = [C.c_void_p, C.c_void_p]
        lib.xhost_writer_set_slice_alf.argtypes = [C.c_void_p, C.POINTER(SliceAlf)]
        lib.xhost_writer_set_slices.argtypes = [C.c_void_p, C.c_int, C.POINTER(SliceDesc)]
        lib.xhost_writer_set_arbitrary_slices.argtypes = [C.c_void_p, C.c_int]
        _lib = lib
    return _lib


class StreamWriter:
    def __init__(self, width, height, bit_depth=8, max_num_ref_pics=1, qp_u_offset=0, qp_v_offset=0, deblock=True, cu_qp_delta=True,
                 log2_sub_gop=0, main=False, iqt=False, ats=False, addb=False, alpha_off=0, beta_off=0, alf=False, eipd=False, crop=(0, 0, 0, 0),
                 chroma_qp_points=None, dra_aps_id=None, htdf=False, ibc_log_max=0, admvp=False, amvr=False, hmvp=False, dmvr=False, mmvd=False,
                 tiles=None, affine=False, qp_delta_area=0, rpl=False, pocs=False, rpl_in_sps=False, cm_init=False, adcc=False, btt=None, suco=None):
        """chroma_qp_points: None, or (global_offset_flag, [table, ...]) with 1 (same for Cb and Cr) or 2 tables of (delta_in_minus1, delta_out) pairs"""
        self.lib = load()
        sp = StreamParams(width, height, bit_depth, max_num_ref_pics, log2_sub_gop, qp_u_offset, qp_v_offset, int(deblock), int(cu_qp_delta),
                          int(main), int(iqt), int(ats), int(addb), alpha_off, beta_off, int(alf), int(eipd))
        for i in range(4):
            sp.crop[i] = int(crop[i])
        sp.tool_htdf = int(htdf)
        sp.ibc_log_max_size = int(ibc_log_max)
        if tiles is not None:      # (cols, rows, across) or (cols, rows, across, col widths, row heights) - sizes in CTUs of all but the last column / row
            sp.tile_cols, sp.tile_rows, sp.loop_filter_across_tiles = int(tiles[0]), int(tiles[1]), int(tiles[2])
            if len(tiles) > 3:
                for i, v in enumerate(tiles[3]):
                    sp.tile_col_w[i] = int(v)
                for i, v in enumerate(tiles[4]):
                    sp.tile_row_h[i] = int(v)
        sp.tool_admvp = int(admvp)
        sp.tool_amvr, sp.tool_hmvp, sp.tool_dmvr, sp.tool_mmvd = int(amvr), int(hmvp), int(dmvr), int(mmvd)
        sp.tool_affine = int(affine)
        sp.cu_qp_delta_area = int(qp_delta_area)
        sp.tool_rpl, sp.tool_pocs, sp.rpl_in_sps = int(rpl), int(pocs), int(rpl_in_sps)
        sp.tool_cm_init, sp.tool_adcc = int(cm_init), int(adcc)
        if btt is not None:      # (log2 of the smallest CU side, log2_diff_ctu_max_14_cb_size, log2_diff_ctu_max_tt_cb_size, log2_diff_min_cb_min_tt_cb_size_minus2)
            sp.btt, sp.btt_log2_min_cb, sp.btt_diff_max_14, sp.btt_diff_max_tt, sp.btt_diff_min_tt = 1, int(btt[0]), int(btt[1]), int(btt[2]), int(btt[3])
        if suco is not None:     # (log2_diff_ctu_size_max_suco_cb_size, log2_diff_max_suco_min_suco_cb_size)
            sp.suco, sp.suco_diff_max, sp.suco_diff_min = 1, int(suco[0]), int(suco[1])
        if dra_aps_id is not None:
            sp.tool_dra, sp.dra_aps_id = 1, int(dra_aps_id)
        if chroma_qp_points is not None:
            sp.cqt_present, sp.cqt_global_offset, sp.cqt_same = 1, int(chroma_qp_points[0]), int(len(chroma_qp_points[1]) == 1)
            for c, tbl in enumerate(chroma_qp_points[1]):
                sp.cqt_num_points[c] = len(tbl)
                for j, (di, do) in enumerate(tbl):
                    sp.cqt_delta_in[c][j], sp.cqt_delta_out[c][j] = int(di), int(do)
        self.h = self.lib.xhost_writer_open(C.byref(sp))
        if not self.h:
            raise ValueError("xhost_writer_open: bad stream parameters")

    def split_allowed(self, x, y, log2w, log2h):
        """sps_btt_flag: [none, binary vertical cut, binary horizontal, ternary vertical, ternary horizontal] - 0 no, 1 yes, 2 only over inter CUs of a P / B picture"""
        a = (C.c_int * 5)()
        self.lib.xhost_writer_split_allowed(self.h, int(x), int(y), int(log2w), int(log2h), a)
        return list(a)

    def add_picture(self, batch, slice_type=SLICE_P, slice_qp=32, idr=False, temporal_id=0):
        cb, keep = abi.make_cu_batch(batch)
        rc = self.lib.xhost_writer_add_picture(self.h, int(idr), slice_type, slice_qp, temporal_id, C.byref(cb))
        if rc != 0:
            raise RuntimeError(f"xhost_writer_add_picture -> {rc}")

    def add_dra_aps(self, aps_id, in_ranges, scales, cb_scale, cr_scale, table_idx):
        """DRA parameter set (APS type 1): range borders, 4.9 fixed-point scales per range, chroma scales, dra_table_idx"""
        a = DraAps()
        a.aps_id, a.num_ranges, a.cb_scale, a.cr_scale, a.table_idx = aps_id, len(scales), int(cb_scale), int(cr_scale), int(table_idx)
        for i, v in enumerate(in_ranges):
            a.in_ranges[i] = int(v)
        for i, v in enumerate(scales):
            a.scale[i] = int(v)
        rc = self.lib.xhost_writer_add_dra_aps(self.h, C.byref(a))
        if rc != 0:
            raise RuntimeError(f"xhost_writer_add_dra_aps -> {rc}")

    def add_alf_aps(self, aps_id, luma=None, chroma=None, type7=True, delta_idx=None, coef_delta_flag=0, pred_mode_flag=0, filter_coef_flag=None,
                    fixed_pattern=0, fixed_usage=None, fixed_idx=None):
        """luma: [n_filters][12 or 6] coded coefficient values or None; chroma: [6] or None; fixed_pattern 0 / 1 / 2 with per-class usage flags and set indices 0..15"""
        a = AlfAps()
        a.aps_id, a.luma_present, a.chroma_present, a.luma_type_7x7 = aps_id, int(luma is not None), int(chroma is not None), int(type7)
        a.num_luma_filters = 1 if luma is None else len(luma)
        a.coef_delta_flag, a.pred_mode_flag = int(coef_delta_flag), int(pred_mode_flag)
        for c in range(25):
            a.delta_idx[c] = 0 if delta_idx is None else int(delta_idx[c])
            a.filter_coef_flag[c] = 1 if filter_coef_flag is None else int(filter_coef_flag[c])
        a.fixed_filter_pattern = int(fixed_pattern)
        for c in range(25):
            a.fixed_filter_usage[c] = 0 if fixed_usage is None else int(fixed_usage[c])
            a.fixed_filter_idx[c] = 0 if fixed_idx is None else int(fixed_idx[c])
        if luma is not None:
            for f, row in enumerate(luma):
                for i, v in enumerate(row):
                    a.luma_coef[f][i] = int(v)
        if chroma is not None:
            for i, v in enumerate(chroma):
                a.chroma_coef[i] = int(v)
        rc = self.lib.xhost_writer_add_alf_aps(self.h, C.byref(a))
        if rc != 0:
            raise RuntimeError(f"xhost_writer_add_alf_aps -> {rc}")

    def set_slice_alf(self, alf_on, aps_id_y=0, aps_id_ch=0, chroma_idc=0, ctb_flag=None):
        sa = SliceAlf(int(alf_on), aps_id_y, aps_id_ch, chroma_idc, int(ctb_flag is not None), None)
        keep = None
        if ctb_flag is not None:
            keep = np.ascontiguousarray(ctb_flag, np.uint8)
            sa.ctb_flag = keep.ctypes.data_as(C.POINTER(C.c_uint8))
        rc = self.lib.xhost_writer_set_slice_alf(self.h, C.byref(sa))
        if rc != 0:
            raise RuntimeError(f"xhost_writer_set_slice_alf -> {rc}")

    def set_arbitrary_slices(self, on=True):
        rc = self.lib.xhost_writer_set_arbitrary_slices(self.h, int(on))
        if rc != 0:
            raise RuntimeError(f"xhost_writer_set_arbitrary_slices -> {rc}")

    def set_slices(self, slices):
        """the following pictures as len(slices) slice NAL units: (first_tile, last_tile[, slice_qp[, deblock_on]]) per slice, tile rectangles that
        partition the PPS's grid; -1 / missing = what add_picture / the stream parameters say.  [] = one slice with every tile"""
        arr = (SliceDesc * max(len(slices), 1))()
        for i, sd in enumerate(slices):
            sd = tuple(sd) + (-1,) * (4 - len(sd))
            arr[i] = SliceDesc(int(sd[0]), int(sd[1]), int(sd[2]), int(sd[3]))
        rc = self.lib.xhost_writer_set_slices(self.h, len(slices), arr)
        if rc != 0:
            raise RuntimeError(f"xhost_writer_set_slices -> {rc}")

    def add_md5_sei(self, planes):
        """picture-signature SEI for the picture added last; planes = its decoded [Y, U, V] (the MD5 runs over 16-bit LE samples)"""
        import hashlib
        md5 = ((C.c_uint8 * 16) * 3)()
        for c, pl in enumerate(planes):
            for i, v in enumerate(hashlib.md5(np.ascontiguousarray(pl, "<i2").tobytes()).digest()):
                md5[c][i] = v
        rc = self.lib.xhost_writer_add_md5_sei(self.h, md5)
        if rc != 0:
            raise RuntimeError(f"xhost_writer_add_md5_sei -> {rc}")

    def set_ref_luma(self, poc, padded_plane, pad):
        """tool_dmvr with tool_hmvp / tool_mmvd: the decoded luma of a picture already written (xhost_writer_set_ref_luma)"""
        if not hasattr(self, "_luma_keep"):
            self._luma_keep = {}
        _ref_luma(self.lib.xhost_writer_set_ref_luma, self.h, self._luma_keep)(poc, padded_plane, pad)

    def bytes(self):
        p, n = C.c_void_p(), C.c_size_t()
        self.lib.xhost_writer_bytes(self.h, C.byref(p), C.byref(n))
        return C.string_at(p, n.value)

    def close(self):
        if self.h:
            self.lib.xhost_writer_close(self.h)
            self.h = None


def _arr(ptr, n, dtype):
    if n == 0:
        return np.zeros(0, dtype)
    return np.ctypeslib.as_array(ptr, shape=(n,)).astype(dtype, copy=True)


def _feedback(lib, h):
    def feed(mv):
        mv = np.ascontiguousarray(mv, np.int16)
        rc = lib.xhost_parser_set_dmvr_mvs(h, mv.ctypes.data, int(mv.size // 4))
        if rc < 0:
            raise RuntimeError(f"xhost_parser_set_dmvr_mvs -> {rc}: {lib.xhost_parser_error(h).decode()}")
    return feed


def _ref_luma(fn, h, keep):
    def give(poc, padded_plane, pad):
        """padded_plane: int16 array [h + 2 pad][w + 2 pad] of the decoded picture with this POC (pad >= 144); kept alive with the parser / writer"""
        a = np.ascontiguousarray(padded_plane, np.int16)
        keep[poc] = a
        rc = fn(h, int(poc), C.c_void_p(a.ctypes.data + 2 * (pad * a.shape[1] + pad)), int(a.shape[1]))
        if rc < 0:
            raise RuntimeError(f"set_ref_luma -> {rc}")
    return give


def iter_stream(data, consume_batch=None, threads=1, luma_wait=False):
    """generator over the pictures of a .evc byte string in decoding order: dict(params..., batch=dict of numpy arrays in the
    layout of synth.gen_frame).  The C parser runs inside each next() with the GIL released (ctypes).
    consume_batch(params, cu_batch_struct): zero-copy hand-over - called while the parser's arrays are valid (before the next picture
    is parsed) with the xgpu_cu_batch the parser filled; its return value becomes params["batch"] (e.g. a device batch handle).
    luma_wait: xhost_parser_set_ref_luma_wait - params["set_ref_luma"] may then be called from another thread while this generator is inside a later
    picture (which waits where it needs the plane); params["cancel_wait"]() lets a waiting parser fail instead."""
    lib = load()
    buf = (C.c_uint8 * len(data)).from_buffer_copy(data)
    h = lib.xhost_parser_open(buf, len(data))
    if threads > 1:      # the tiles of a picture on parallel host threads (xhost_parser_set_threads)
        lib.xhost_parser_set_threads(h, int(threads))
    if luma_wait:
        lib.xhost_parser_set_ref_luma_wait(h, 1)
    luma_keep = {}
    try:
        while True:
            hp = HostPicture()
            rc = lib.xhost_parser_next(h, C.byref(hp))
            if rc == 0:
                break
            if rc < 0:
                raise RuntimeError(f"xhost_parser_next -> {rc}: {lib.xhost_parser_error(h).decode()}")
            b, n = hp.batch, hp.batch.n_cu
            batch = None if consume_batch is not None else {
                "x": _arr(b.x, n, np.uint16), "y": _arr(b.y, n, np.uint16), "log2w": _arr(b.log2w, n, np.uint8), "log2h": _arr(b.log2h, n, np.uint8),
                "pred_mode": _arr(b.pred_mode, n, np.uint8), "refi": _arr(b.refi, n * 2, np.int8).reshape(n, 2),
                "mv": _arr(b.mv, n * 4, np.int16).reshape(n, 2, 2), "qp": _arr(b.qp, n * 3, np.uint8).reshape(n, 3),
                "cbf": _arr(b.cbf, n, np.uint8), "cbf_sub": None, "ats": _arr(b.ats, n, np.uint8) if b.ats else None,
                "ats_inter": _arr(b.ats_inter, n, np.uint8) if b.ats_inter else None, "ipm": _arr(b.ipm, n * 2, np.uint8).reshape(n, 2),
                "coef_off": _arr(b.coef_off, n, np.uint32), "coef": _arr(b.coef, max(b.n_coef, 1), np.int16), "n_coef": int(b.n_coef),
                "ctu_cu_start": _arr(b.ctu_cu_start, b.n_ctu + 1, np.uint32), "constrained_intra_pred": int(b.constrained_intra_pred), "htdf_slice_qp": int(b.htdf_slice_qp),
                "dmvr": _arr(b.dmvr, n, np.uint8) if b.dmvr else None,
                "tree": _arr(b.tree, n, np.uint8) if b.tree else None,
                "affine": _arr(b.affine, n, np.uint8) if b.affine else None,
                "affine_mv": _arr(b.affine_mv, n * 12, np.int16).reshape(n, 2, 3, 2) if b.affine else None,
                "tiles": abi.tile_grid_dict(b.tiles.contents) if b.tiles else None,
            }
            params = {
                "width": hp.width, "height": hp.height, "bit_depth": hp.bit_depth_luma, "bit_depth_chroma": hp.bit_depth_chroma, "poc": hp.poc, "temporal_id": hp.temporal_id, "slice_type": hp.slice_type,
                "is_idr": bool(hp.is_idr), "is_ref": bool(hp.is_ref),
                "refs": [[hp.refp_poc[i][l] for i in range(hp.num_refp[l])] for l in range(2)],
                "slice_qp": hp.slice_qp, "qp_u_offset": hp.qp_u_offset, "qp_v_offset": hp.qp_v_offset, "deblock_on": bool(hp.deblock_on),
                "main": bool(hp.profile_main), "iqt": hp.tool_iqt, "ats": hp.tool_ats, "addb": hp.tool_addb,
                "alpha_off": hp.deblock_alpha_offset, "beta_off": hp.deblock_beta_offset, "tool_alf": hp.tool_alf, "eipd": hp.tool_eipd, "admvp": hp.tool_admvp, "crop": tuple(hp.crop[i] for i in range(4)),
                "dra": None if not hp.dra_lut[0] else [np.ctypeslib.as_array(hp.dra_lut[c], (1024,)).copy() for c in range(3)],
                "chroma_qp_tables": None if not hp.chroma_qp_table[0] else [np.ctypeslib.as_array(hp.chroma_qp_table[c], (58 + 6 * (hp.bit_depth_chroma - 8),)).copy() for c in range(2)],
                "alf": None if not hp.alf_on else {
                    "enable": tuple(hp.alf.enable[i] for i in range(3)), "luma_coef": _arr(hp.alf.luma_coef, 25 * 13, np.int16).reshape(25, 13),
                    "chroma_coef": _arr(hp.alf.chroma_coef, 7, np.int16), "ctb_flag": _arr(hp.alf.ctb_flag, b.n_ctu, np.uint8), "across_tiles": int(hp.alf.across_tiles),
                    "tiles": abi.tile_grid_dict(hp.alf.tiles.contents) if hp.alf.tiles else None},
                "md5": [bytes(hp.md5[c]) for c in range(3)] if hp.has_md5 else None,
                "release": [hp.release_poc[i] for i in range(hp.n_release)], "batch": batch,
                # sps->tool_dmvr: the number of sub-blocks whose vectors (xgpu_batch_dmvr_mvs / the oracle's dmvr_mv_out) must be handed to
                # dmvr_feedback() before the generator is advanced - the temporal candidates of later pictures read them
                "n_dmvr_sub": int(hp.n_dmvr_sub), "dmvr_feedback": _feedback(lib, h),
                # tool_dmvr with tool_hmvp / tool_mmvd: the parser refines vectors itself while it parses later pictures and needs this picture's decoded
                # luma (padded) for it: set_ref_luma(poc, padded_plane, pad) before the generator is advanced
                "needs_ref_luma": bool(hp.needs_ref_luma), "set_ref_luma": _ref_luma(lib.xhost_parser_set_ref_luma, h, luma_keep),
                "cancel_wait": (lambda: lib.xhost_parser_cancel_wait(h)),
            }
            for gone in params["release"]:      # no longer in the parser's DPB: their registered luma planes can go
                luma_keep.pop(gone, None)
            if consume_batch is not None:
                params["batch"] = consume_batch(params, hp.batch)
            yield params
    finally:
        lib.xhost_parser_close(h)


def parse_stream(data, threads=1):
    """-> list of the pictures of iter_stream(data)"""
    return list(iter_stream(data, threads=threads))
