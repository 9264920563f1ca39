"""Host-side mirror of the reference's coarse decode sequence for the reconstruction path, over the C ABI.

Mirrors, per picture, what xevd_dec_nalu does after entropy decoding (src_base/xevd.c:1890-1983):
    slice/refp set-up -> recon of every CU -> deblock (vertical edges, then horizontal) -> picbuf_expand -> DPB
Plumbing only (ctypes + numpy); all arithmetic happens in xevd_amd/libxevd_hip.so.  There is no CPU fallback.
"""
import ctypes as C

import numpy as np

from . import abi


class XgpuError(RuntimeError):
    pass


class XgpuDecoder:
    def __init__(self, width, height, bit_depth=8, device=0, log2_ctu=6, iqt=0, admvp=0, addb=0, alf=0, max_pics=6,
                 bit_depth_chroma=None, chroma_qp_tables=None, eipd=0):
        self.lib = abi.load()
        self.sp = abi.make_seq_params(width, height, bit_depth, log2_ctu, device, iqt, admvp, addb, alf, max_pics,
                                      bit_depth_chroma, eipd)
        self._tables = None
        if chroma_qp_tables is not None:
            self._tables = [np.ascontiguousarray(t, np.int8) for t in chroma_qp_tables]
            for i in range(2):
                self.sp.chroma_qp_table[i] = self._tables[i].ctypes.data_as(C.POINTER(C.c_int8))
        self.ctx = C.c_void_p()
        rc = self.lib.xgpu_open(C.byref(self.sp), C.byref(self.ctx))
        if rc != 0:
            raise XgpuError(f"xgpu_open failed: {rc}")
        self.width, self.height, self.bit_depth = width, height, bit_depth
        self._batches = []

    # -- helpers -------------------------------------------------------------------------------------
    def _chk(self, rc, what):
        if rc < 0:
            raise XgpuError(f"{what} failed: {rc}: {self.lib.xgpu_last_error(self.ctx).decode()}")
        return rc

    def close(self):
        if self.ctx:
            for b in self._batches:
                self.lib.xgpu_batch_destroy(self.ctx, b)
            self._batches = []
            self.lib.xgpu_close(self.ctx)
            self.ctx = C.c_void_p()

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def sync(self):
        self._chk(self.lib.xgpu_sync(self.ctx), "xgpu_sync")

    # -- pictures ------------------------------------------------------------------------------------
    def pic_alloc(self):
        return self._chk(self.lib.xgpu_pic_alloc(self.ctx), "xgpu_pic_alloc")

    def pic_free(self, pic):
        self._chk(self.lib.xgpu_pic_free(self.ctx, pic), "xgpu_pic_free")

    def pic_upload(self, pic, planes):
        y, u, v = (np.ascontiguousarray(p, np.int16) for p in planes)
        self._chk(self.lib.xgpu_pic_upload(self.ctx, pic, y.ctypes.data, y.shape[1], u.ctypes.data, v.ctypes.data, u.shape[1]),
                  "xgpu_pic_upload")

    def pic_download(self, pic):
        y = np.zeros((self.height, self.width), np.int16)
        u = np.zeros((self.height // 2, self.width // 2), np.int16)
        v = np.zeros_like(u)
        self._chk(self.lib.xgpu_pic_download(self.ctx, pic, y.ctypes.data, self.width, u.ctypes.data, v.ctypes.data, self.width // 2),
                  "xgpu_pic_download")
        return [y, u, v]

    def pic_output(self, pic, out_bit_depth=0, crop=(0, 0, 0, 0), dra=None):
        """The picture as the bytes of a .yuv file frame (Y, U, V planes, tight rows): converted to `out_bit_depth` (0 = the coding
        depth; 8 -> one byte per sample) and cropped by (left, right, top, bottom) luma samples on the device (xgpu_pic_output).
        dra: (luma_inv_scale_lut, cb_inv_scale_lut, cr_inv_scale_lut), 1024 int32 each - the DRA post-filter's tables."""
        bd = out_bit_depth or self.bit_depth
        n = self.lib.xgpu_pic_output_size(self.ctx, bd, *crop)
        if n == 0:
            raise ValueError(f"invalid output format: bit depth {bd}, crop {crop}")
        out = np.empty(n, np.uint8)
        dl, keep = None, None
        if dra is not None:
            keep = [np.ascontiguousarray(t, np.int32) for t in dra]
            assert all(t.size == 1024 for t in keep)
            d = abi.DraLuts()
            d.luma_inv_scale_lut = keep[0].ctypes.data
            d.chroma_inv_scale_lut[0], d.chroma_inv_scale_lut[1] = keep[1].ctypes.data, keep[2].ctypes.data
            dl = C.byref(d)
        self._chk(self.lib.xgpu_pic_output(self.ctx, pic, dl, bd, *crop, out.ctypes.data, n), "xgpu_pic_output")
        return out

    def pic_md5(self, pic, dra=None):
        """the picture signature made on the device (xgpu_pic_md5): [Y, U, V] digests of 16 bytes - the MD5 of every plane's 16-bit samples as the reference's
        xevd_md5_imgb makes it; with `dra` tables (as pic_output takes them) of the DRA-mapped picture"""
        dl, keep = None, None
        if dra is not None:
            keep = [np.ascontiguousarray(t, np.int32) for t in dra]
            assert all(t.size == 1024 for t in keep)
            d = abi.DraLuts()
            d.luma_inv_scale_lut = keep[0].ctypes.data
            d.chroma_inv_scale_lut[0], d.chroma_inv_scale_lut[1] = keep[1].ctypes.data, keep[2].ctypes.data
            dl = C.byref(d)
        out = np.zeros((3, 16), np.uint8)
        self._chk(self.lib.xgpu_pic_md5(self.ctx, pic, dl, out.ctypes.data), "xgpu_pic_md5")
        return [bytes(out[c]) for c in range(3)]

    def host_alloc(self, nbytes, dtype=np.uint8):
        """pinned host memory from the backend as a numpy array (freed with the decoder): for coefficient arenas and output buffers"""
        p = C.c_void_p()
        self._chk(self.lib.xgpu_host_alloc(self.ctx, nbytes, C.byref(p)), "xgpu_host_alloc")
        return np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_uint8)), (nbytes,)).view(dtype)

    def pic_output_async(self, pic, out, out_bit_depth=0, crop=(0, 0, 0, 0)):
        """queue conversion + packing + the copy into `out` (uint8 array, pinned for a truly asynchronous copy) -> ticket for pic_output_wait"""
        t = C.c_int()
        self._chk(self.lib.xgpu_pic_output_async(self.ctx, pic, None, out_bit_depth or self.bit_depth, *crop, out.ctypes.data, out.nbytes, C.byref(t)),
                  "xgpu_pic_output_async")
        return t.value

    def pic_output_wait(self, ticket):
        self._chk(self.lib.xgpu_pic_output_wait(self.ctx, ticket), "xgpu_pic_output_wait")

    def batch_dmvr_mvs(self, h):
        """the vectors kept for temporal prediction of the batch's DMVR candidates after batch_recon: [n_sub_blocks][list][x/y], quarter samples"""
        n = self._chk(self.lib.xgpu_batch_dmvr_mvs(self.ctx, h, None, 0), "xgpu_batch_dmvr_mvs")
        out = np.zeros((max(n, 1), 2, 2), np.int16)
        self._chk(self.lib.xgpu_batch_dmvr_mvs(self.ctx, h, out.ctypes.data, n), "xgpu_batch_dmvr_mvs")
        return out[:n]

    def batch_info(self, h):
        """what the batch builder made of the batch: CU / TB / work-item counts, nodes and depth of the dependency graph (xgpu_batch_info)"""
        v = (C.c_int * 8)()
        self._chk(self.lib.xgpu_batch_info(self.ctx, h, v), "xgpu_batch_info")
        return dict(zip(("n_cu", "n_tb", "itdq_items", "dep_nodes", "dep_nodes_level1", "dep_levels", "dmvr_sub_blocks", "affine_tiles"), list(v)))

    def batch_wait_upload(self, h):
        self._chk(self.lib.xgpu_batch_wait_upload(self.ctx, h), "xgpu_batch_wait_upload")

    def pic_upload_padded(self, pic, bufs):
        y, u, v = (np.ascontiguousarray(p, np.int16) for p in bufs)
        self._chk(self.lib.xgpu_pic_upload_padded(self.ctx, pic, y.ctypes.data, u.ctypes.data, v.ctypes.data), "xgpu_pic_upload_padded")

    def pic_download_padded(self, pic):
        y = np.zeros((self.height + 2 * abi.PAD_L, self.width + 2 * abi.PAD_L), np.int16)
        u = np.zeros((self.height // 2 + 2 * abi.PAD_C, self.width // 2 + 2 * abi.PAD_C), np.int16)
        v = np.zeros_like(u)
        self._chk(self.lib.xgpu_pic_download_padded(self.ctx, pic, y.ctypes.data, u.ctypes.data, v.ctypes.data), "xgpu_pic_download_padded")
        return [y, u, v]

    def pic_download_padded_luma(self, pic):
        """the padded luma plane alone (what a front end that refines vectors itself registers with xhost_parser_set_ref_luma)"""
        y = np.zeros((self.height + 2 * abi.PAD_L, self.width + 2 * abi.PAD_L), np.int16)
        self._chk(self.lib.xgpu_pic_download_padded(self.ctx, pic, y.ctypes.data, None, None), "xgpu_pic_download_padded")
        return y

    # -- per picture ---------------------------------------------------------------------------------
    def frame_begin(self, pic, poc, refs, qp_u_offset=0, qp_v_offset=0, deblock_on=0, alf_on=0, alpha_off=0, beta_off=0):
        """refs: {(idx, list): (pic_slot, poc)}"""
        fp = abi.FrameParams()
        fp.pic, fp.poc = pic, poc
        for l in range(2):
            idxs = [i for (i, ll) in refs if ll == l]
            fp.num_refp[l] = (max(idxs) + 1) if idxs else 0
        for (i, l), (slot, rpoc) in refs.items():
            fp.refp_pic[i][l] = slot
            fp.refp_poc[i][l] = rpoc
        fp.qp_u_offset, fp.qp_v_offset = qp_u_offset, qp_v_offset
        fp.deblock_alpha_offset, fp.deblock_beta_offset = alpha_off, beta_off
        fp.deblock_on, fp.alf_on = int(deblock_on), int(alf_on)
        self._chk(self.lib.xgpu_frame_begin(self.ctx, C.byref(fp)), "xgpu_frame_begin")

    def batch_create(self, batch):
        cb, keep = abi.make_cu_batch(batch)
        h = C.c_void_p()
        self._chk(self.lib.xgpu_batch_create(self.ctx, C.byref(cb), C.byref(h)), "xgpu_batch_create")
        self._batches.append(h)
        return h

    def batch_create_from_struct(self, cu_batch):
        """xgpu_batch_create on a filled abi.CuBatch (e.g. the one inside the host parser's xhost_picture): no numpy round trip"""
        h = C.c_void_p()
        self._chk(self.lib.xgpu_batch_create(self.ctx, C.byref(cu_batch), C.byref(h)), "xgpu_batch_create")
        self._batches.append(h)
        return h

    def batch_resid(self, h, n_coef):
        out = np.zeros(max(n_coef, 1), np.int16)
        self._chk(self.lib.xgpu_test_batch_resid(self.ctx, h, out.ctypes.data), "xgpu_test_batch_resid")
        return out

    def batch_destroy(self, h):
        self._batches = [b for b in self._batches if b.value != h.value]
        self.lib.xgpu_batch_destroy(self.ctx, h)

    def batch_prepare(self, h):
        """queue the batch's residual pass ahead of its picture (xgpu_batch_prepare)"""
        self._chk(self.lib.xgpu_batch_prepare(self.ctx, h), "xgpu_batch_prepare")

    def batch_recon(self, h, next_batch=None):
        """next_batch: the NEXT picture's batch - its residual pass is queued with this picture's kernels (xgpu_batch_recon_ahead)"""
        if next_batch is None:
            self._chk(self.lib.xgpu_batch_recon(self.ctx, h), "xgpu_batch_recon")
        else:
            self._chk(self.lib.xgpu_batch_recon_ahead(self.ctx, h, next_batch), "xgpu_batch_recon_ahead")

    def deblock(self):
        self._chk(self.lib.xgpu_deblock(self.ctx), "xgpu_deblock")

    def alf(self, params):
        ap, keep = abi.make_alf_params(params)
        self._chk(self.lib.xgpu_alf(self.ctx, C.byref(ap)), "xgpu_alf")

    def pad(self):
        self._chk(self.lib.xgpu_pad(self.ctx), "xgpu_pad")

    def frame_end(self):
        self._chk(self.lib.xgpu_frame_end(self.ctx), "xgpu_frame_end")

    def decode_picture(self, pic, poc, refs, batch_handle, deblock=True, pad=True, qp_u_offset=0, qp_v_offset=0,
                       alpha_off=0, beta_off=0, alf=None, next_batch=None):
        """The coarse sequence of xevd_dec_nalu for one picture (src_base/xevd.c:1905-1983, src_main/xevdm.c:3136-3219)."""
        self.frame_begin(pic, poc, refs, qp_u_offset, qp_v_offset, deblock_on=deblock, alf_on=alf is not None,
                         alpha_off=alpha_off, beta_off=beta_off)
        self.batch_recon(batch_handle, next_batch)      # with the next picture's residual pass under this picture's dependency kernel
        if deblock:
            self.deblock()
        if alf is not None:
            self.alf(alf)
        if pad:
            self.pad()
        self.frame_end()

    # -- measurement ---------------------------------------------------------------------------------
    def timing_enable(self, on=True):
        self._chk(self.lib.xgpu_timing_enable(self.ctx, 1 if on else 0), "xgpu_timing_enable")

    def timing_reset(self):
        self._chk(self.lib.xgpu_timing_reset(self.ctx), "xgpu_timing_reset")

    def timing_get(self):
        ms = (C.c_double * abi.K_COUNT)()
        n = (C.c_longlong * abi.K_COUNT)()
        self._chk(self.lib.xgpu_timing_get(self.ctx, ms, n), "xgpu_timing_get")
        return {abi.K_NAMES[i]: (ms[i], n[i]) for i in range(abi.K_COUNT)}

    def measure_copy_bw(self, nbytes=1 << 30, iters=10):
        g = C.c_double()
        self._chk(self.lib.xgpu_measure_copy_bw(self.ctx, nbytes, iters, C.byref(g)), "xgpu_measure_copy_bw")
        return g.value

    # -- fine-grained shims --------------------------------------------------------------------------
    def test_mc(self, plane, ref_x, ref_y, has_dx, has_dy, gmv_x, gmv_y, w, h, bit_depth, luma=True):
        plane = np.ascontiguousarray(plane, np.int16)
        pred = np.zeros((h, w), np.int16)
        fn = self.lib.xgpu_test_mc_l if luma else self.lib.xgpu_test_mc_c
        self._chk(fn(self.ctx, plane.ctypes.data, plane.shape[1], plane.shape[0], ref_x, ref_y, has_dx, has_dy, gmv_x, gmv_y,
                     pred.ctypes.data, w, h, bit_depth), "xgpu_test_mc")
        return pred

    def test_recon(self, coef, pred, is_coef, rec, bit_depth):
        """fn_recon's shape: coef / pred [cuh][cuw], rec [cuh][s_rec] (only its first cuw columns are written) -> rec"""
        coef, pred = np.ascontiguousarray(coef, np.int16), np.ascontiguousarray(pred, np.int16)
        rec = np.ascontiguousarray(rec, np.int16).copy()
        self._chk(self.lib.xgpu_test_recon(self.ctx, coef.ctypes.data, pred.ctypes.data, int(is_coef), pred.shape[1], pred.shape[0], rec.shape[1],
                                           rec.ctypes.data, bit_depth), "xgpu_test_recon")
        return rec

    def test_dbk(self, plane, x, y, st, hor, bit_depth, plane_v=None, st_v=0):
        """fn_dbk / fn_dbk_chroma's shape on one edge segment of a small plane (both chroma planes when plane_v is given) -> filtered plane(s)"""
        plane = np.ascontiguousarray(plane, np.int16).copy()
        if plane_v is None:
            self._chk(self.lib.xgpu_test_dbk(self.ctx, plane.ctypes.data, plane.shape[1], plane.shape[0], x, y, st, int(hor), bit_depth), "xgpu_test_dbk")
            return plane
        plane_v = np.ascontiguousarray(plane_v, np.int16).copy()
        self._chk(self.lib.xgpu_test_dbk_chroma(self.ctx, plane.ctypes.data, plane_v.ctypes.data, plane.shape[1], plane.shape[0], x, y, st, st_v, int(hor),
                                                bit_depth), "xgpu_test_dbk_chroma")
        return plane, plane_v

    def test_itdq(self, coef, log2w, log2h, qp, bit_depth):
        coef = np.ascontiguousarray(coef, np.int16).copy()
        qp = np.ascontiguousarray(qp, np.uint8)
        self._chk(self.lib.xgpu_test_itdq(self.ctx, coef.ctypes.data, len(qp), log2w, log2h, qp.ctypes.data, bit_depth), "xgpu_test_itdq")
        return coef
