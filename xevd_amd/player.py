"""Bitstream -> pictures on the GPU: the decoder loop a user of the reference's xevd_decode / xevd_pull pair would write on top of
the two C ABIs (include/xevd_host.h parser, include/xevd_hip.h backend).  The host parser runs in its own thread, one picture
ahead (entropy decoding of picture k+1 overlaps the upload + kernels of picture k); DPB slots are managed by POC exactly as the
parser reports them (references by POC, released POCs).  Plumbing only - no sample arithmetic here."""
import queue
import threading

from . import abi, stream
from .decoder import XgpuDecoder


class StreamDecoder:
    def __init__(self, data, device=0, prefetch=2, verify_md5=False, apply_crop=False, parser_threads=1):
        """verify_md5: check downloaded pictures against the stream's picture-signature SEIs (the reference's
        XEVD_CFG_SET_USE_PIC_SIGNATURE), raising on a mismatch"""
        self.data, self.device, self.prefetch, self.verify_md5 = data, device, prefetch, verify_md5
        self._lock, self._dec, self._abort = threading.Lock(), None, False
        self.parser_threads = parser_threads      # host threads for the tiles of one picture (xhost_parser_set_threads)
        self.apply_crop = apply_crop      # packed output: cut the SPS conformance window (the reference application writes uncropped pictures)

    N_SLOTS = 33      # the parser keeps at most 32 reference pictures (+ the current one): a slot is always free

    def _producer(self, q):
        """parser thread: entropy decoding, and - zero-copy, while the parser's arrays are valid - the hand-over of every picture's CU batch to
        the backend (builder + upload into pooled buffers).  The backend context is not thread-safe: calls on it are serialised by self._lock."""
        def to_device(p, cu_batch):
            with self._lock:
                if self._dec is None:
                    self._dec = XgpuDecoder(p["width"], p["height"], p["bit_depth"], device=self.device, iqt=p["iqt"], admvp=p["admvp"], addb=p["addb"], alf=p["tool_alf"],
                                            eipd=p["eipd"], max_pics=34, chroma_qp_tables=p["chroma_qp_tables"], bit_depth_chroma=p["bit_depth_chroma"])
                return self._dec.batch_create_from_struct(cu_batch)
        try:
            for p in stream.iter_stream(self.data, consume_batch=to_device, threads=self.parser_threads):
                if p["n_dmvr_sub"]:
                    p["_dmvr"] = [threading.Event(), None]
                if p["needs_ref_luma"]:
                    p["_luma"] = [threading.Event(), None]
                q.put(p)
                if p["needs_ref_luma"]:
                    # tool_dmvr with tool_hmvp / tool_mmvd: the parser refines merge vectors itself while it parses the NEXT pictures and reads this
                    # picture's decoded luma for it - the consumer downloads it (padded) right behind the picture's kernels
                    while not p["_luma"][0].wait(0.05):
                        if self._abort:
                            return
                    p["set_ref_luma"](p["poc"], p["_luma"][1], abi.PAD_L)
                if p["n_dmvr_sub"]:
                    # sps->tool_dmvr: the temporal candidates of later pictures read this picture's REFINED vectors - the parser waits for the
                    # backend's (xgpu_batch_dmvr_mvs, fetched by the consumer right after the picture's kernels were queued)
                    while not p["_dmvr"][0].wait(0.05):
                        if self._abort:
                            return
                    p["dmvr_feedback"](p["_dmvr"][1])
            q.put(None)
        except Exception as e:      # surfaced in the consumer thread
            q.put(e)

    @staticmethod
    def signature_ok(p, planes):
        """the stream's picture-signature SEI (MD5 per plane over 16-bit samples, xevd_picbuf_check_signature) against decoded planes"""
        import hashlib
        import numpy as np
        return all(hashlib.md5(np.ascontiguousarray(pl, "<i2").tobytes()).digest() == p["md5"][c] for c, pl in enumerate(planes))

    def pictures(self, download=True, output_bit_depth=None):
        """generator of (params, planes or None) in DECODING order; planes = [Y, U, V] int16 arrays of the active area, or - with
        output_bit_depth (0 = the coding depth) - the bytes of one .yuv frame, converted and packed on the device"""
        q = queue.Queue(maxsize=self.prefetch)
        th = threading.Thread(target=self._producer, args=(q,), daemon=True)
        th.start()
        slots, free = {}, []

        def decode(p):
            dec, hb = self._dec, p["batch"]
            if p["is_idr"]:
                free.extend(slots.values()); slots.clear()
            cur = free.pop()
            refs = {(i, l): (slots[poc], poc) for l in range(2) for i, poc in enumerate(p["refs"][l])}
            with self._lock:
                dec.decode_picture(cur, p["poc"], refs, hb, deblock=p["deblock_on"], pad=True, qp_u_offset=p["qp_u_offset"], qp_v_offset=p["qp_v_offset"],
                                   alpha_off=p["alpha_off"], beta_off=p["beta_off"], alf=p["alf"])
                if p["n_dmvr_sub"]:
                    p["_dmvr"][1] = dec.batch_dmvr_mvs(hb)
                    p["_dmvr"][0].set()
                if p["needs_ref_luma"]:
                    p["_luma"][1] = dec.pic_download_padded_luma(cur)
                    p["_luma"][0].set()
                dec.batch_destroy(hb)          # back to the pool; queued kernels keep reading it (same HIP stream)
            planes = None
            if download and output_bit_depth is not None:
                planes = dec.pic_output(cur, output_bit_depth, p["crop"] if self.apply_crop else (0, 0, 0, 0), dra=p["dra"])
            elif download and p["dra"] is not None:      # the DRA post-filter belongs to the output: planes through the output kernel
                import numpy as np
                w, h = p["width"], p["height"]
                flat = dec.pic_output(cur, max(p["bit_depth"], 9), dra=p["dra"]).view("<u2").astype(np.int16) if p["bit_depth"] > 8 else None
                if flat is None:
                    raise RuntimeError("DRA on 8-bit pictures: use output_bit_depth")
                planes = [flat[:w * h].reshape(h, w), flat[w * h:w * h * 5 // 4].reshape(h // 2, w // 2), flat[w * h * 5 // 4:].reshape(h // 2, w // 2)]
            elif download:
                planes = dec.pic_download(cur)
                if self.verify_md5 and p["md5"] is not None and not self.signature_ok(p, planes):
                    raise RuntimeError(f"picture signature mismatch at POC {p['poc']} (XEVD_ERR_BAD_CRC)")
            for poc in p["release"]:      # unmarked when THIS picture is stored (it may still have referenced them)
                if poc in slots:
                    free.append(slots.pop(poc))
            if p["is_ref"]:
                slots[p["poc"]] = cur
            else:
                free.append(cur)
            return planes

        try:
            while True:
                p = q.get()
                if p is None:
                    break
                if isinstance(p, Exception):
                    raise p
                if not free and not slots:
                    with self._lock:
                        free = [self._dec.pic_alloc() for _ in range(self.N_SLOTS)]
                yield p, decode(p)
            if self._dec is not None and not download:
                self._dec.sync()
        finally:
            # let the parser thread finish (it may be inside a backend call), then release the device
            self._abort = True
            while th.is_alive():
                try:
                    q.get(timeout=0.05)
                except queue.Empty:
                    pass
            if self._dec is not None:
                self._dec.close()
                self._dec = None

    def output_order(self, output_bit_depth=None):
        """all pictures in output order (ascending POC inside every IDR period), as xevd_pull's bumping delivers them"""
        out, epoch = [], -1
        for p, planes in self.pictures(output_bit_depth=output_bit_depth):
            if p["is_idr"]:
                epoch += 1
            out.append(((epoch, p["poc"]), p, planes))
        return [(p, planes) for _, p, planes in sorted(out, key=lambda t: t[0])]
