"""Bitstream -> pictures on the GPU: the decoder loop a user of the reference's xevd_decode / xevd_pull pair would write on top of
the two C ABIs (include/xevd_host.h parser, include/xevd_hip.h backend).  The host parser runs in its own thread, one picture
ahead (entropy decoding of picture k+1 overlaps the upload + kernels of picture k); DPB slots are managed by POC exactly as the
parser reports them (references by POC, released POCs).  Plumbing only - no sample arithmetic here."""
import queue
import threading

from . import abi, stream
from .decoder import XgpuDecoder


class StreamDecoder:
    def __init__(self, data, device=0, prefetch=2):
        self.data, self.device, self.prefetch = data, device, prefetch

    def _producer(self, q):
        try:
            for p in stream.iter_stream(self.data):
                q.put(p)
            q.put(None)
        except Exception as e:      # surfaced in the consumer thread
            q.put(e)

    def pictures(self, download=True):
        """generator of (params, planes or None) in DECODING order; planes = [Y, U, V] int16 arrays of the active area"""
        q = queue.Queue(maxsize=self.prefetch)
        th = threading.Thread(target=self._producer, args=(q,), daemon=True)
        th.start()
        dec, slots, free = None, {}, []
        try:
            while True:
                p = q.get()
                if p is None:
                    break
                if isinstance(p, Exception):
                    raise p
                if dec is None:
                    dec = XgpuDecoder(p["width"], p["height"], p["bit_depth"], device=self.device, iqt=p["iqt"], addb=p["addb"], alf=p["tool_alf"], max_pics=12)
                    free = [dec.pic_alloc() for _ in range(10)]
                if p["is_idr"]:
                    free.extend(slots.values()); slots.clear()
                cur = free.pop()
                refs = {(i, l): (slots[poc], poc) for l in range(2) for i, poc in enumerate(p["refs"][l])}
                hb = dec.batch_create(p["batch"])
                dec.decode_picture(cur, p["poc"], refs, hb, deblock=p["deblock_on"], pad=True, qp_u_offset=p["qp_u_offset"], qp_v_offset=p["qp_v_offset"],
                                   alpha_off=p["alpha_off"], beta_off=p["beta_off"], alf=p["alf"])
                planes = None
                if download:
                    planes = dec.pic_download(cur)
                else:
                    dec.sync()
                dec.batch_destroy(hb)
                for poc in p["release"]:      # unmarked when THIS picture is stored (it may still have referenced them)
                    if poc in slots:
                        free.append(slots.pop(poc))
                if p["is_ref"]:
                    slots[p["poc"]] = cur
                else:
                    free.append(cur)
                yield p, planes
        finally:
            if dec is not None:
                dec.close()

    def output_order(self):
        """all pictures in output order (ascending POC inside every IDR period), as xevd_pull's bumping delivers them"""
        out, epoch = [], -1
        for p, planes in self.pictures():
            if p["is_idr"]:
                epoch += 1
            out.append(((epoch, p["poc"]), p, planes))
        return [(p, planes) for _, p, planes in sorted(out, key=lambda t: t[0])]
