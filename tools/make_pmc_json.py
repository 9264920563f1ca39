#!/usr/bin/env python3
"""profiles/latest_pmc.json from rocprofv3 --pmc passes (one pass per counter group, no tracing).
usage: make_pmc_json.py <workload> <rdreq.csv> <wrreq.csv> <out.json> [<fetch_size.csv> <write_size.csv>]      (XEVD_COMMIT in the environment: recorded as the code's commit)

HBM traffic per dispatch from the L2 <-> fabric REQUEST counters, which carry their size:
    read bytes  = 128 * TCC_EA0_RDREQ_128B + 64 * TCC_EA0_RDREQ_64B + 32 * TCC_EA0_RDREQ_32B      (the three classes partition TCC_EA0_RDREQ)
    write bytes =  64 * TCC_EA0_WRREQ_64B + 32 * (TCC_EA0_WRREQ - TCC_EA0_WRREQ_64B)
Checked on the runtime's own copy kernel in the same passes: one 8K 10-bit picture (99 532 800 B) counts 777 600 x 128 B read, 1 555 200 x 64 B
written - exact.  FETCH_SIZE / WRITE_SIZE (the guide's method: KB per dispatch, FETCH_SIZE doubled on gfx950 because 128-byte requests are
tallied at 64) are kept next to it when their passes are given: FETCH_SIZE x 2 over-counts kernels that issue 64-byte requests (it doubles those
too), which is where round 1's "2.16x" ALF traffic came from."""
import csv
import json
import os
import sys

NAMES = {"k_inter(": "inter", "k_inter_af(": "inter", "k_inter_region(": "inter", "k_inter_tile(": "inter", "k_inter_split(": "inter",      # (the three inter launches are one pass: summed)
          "k_alf(": "alf", "k_addb<0>(": "dbk_v", "k_addb<1>(": "dbk_h", "k_addb_fused<": "dbk_v", "k_dbk<0>(": "dbk_v", "k_dbk<1>(": "dbk_h",
         "k_itdq": "itdq", "k_intra<": "intra", "k_intra_itdq<": "intra_itdq", "k_addb_alf(": "alf", "k_addb_alf<": "alf", "k_affine_": "affine", "k_pad(": "pad", "k_dmvr(": "dmvr"}


def load(path):
    out = {}
    for r in csv.DictReader(open(path)):
        for k, v in NAMES.items():
            if k in r["kernel"]:
                d = out.setdefault(v, {})
                d[r["counter"]] = d.get(r["counter"], 0.0) + float(r["avg"])
    return out


def main():
    wl, rpath, wpath, opath = sys.argv[1:5]
    rd, wr = load(rpath), load(wpath)
    fs = load(sys.argv[5]) if len(sys.argv) > 6 else {}
    ws = load(sys.argv[6]) if len(sys.argv) > 6 else {}
    kern = {}
    for k in sorted(set(rd) | set(wr)):
        r, w = rd.get(k, {}), wr.get(k, {})
        rb = 128 * r.get("TCC_EA0_RDREQ_128B_sum", 0) + 64 * r.get("TCC_EA0_RDREQ_64B_sum", 0) + 32 * r.get("TCC_EA0_RDREQ_32B_sum", 0)
        wb = 64 * w.get("TCC_EA0_WRREQ_64B_sum", 0) + 32 * (w.get("TCC_EA0_WRREQ_sum", 0) - w.get("TCC_EA0_WRREQ_64B_sum", 0))
        kern[k] = {"read_bytes": int(rb), "write_bytes": int(wb), "traffic_bytes": int(rb + wb)}
        if k in fs or k in ws:
            f, wv = fs.get(k, {}).get("FETCH_SIZE", 0.0), ws.get(k, {}).get("WRITE_SIZE", 0.0)
            kern[k]["fetch_size_kb"] = round(f, 1)
            kern[k]["write_size_kb"] = round(wv, 1)
            kern[k]["traffic_bytes_fetch_size_method"] = int((2.0 * f + wv) * 1024)
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
    import bench
    json.dump({"workload": wl, "commit": os.environ.get("XEVD_COMMIT"), "sources_sha256": {k: bench.kernel_sources_sha256(k) for k in kern if k in bench.KERNEL_SOURCES}, "source": f"{rpath} + {wpath} (rocprofv3 --pmc, separate passes, no tracing)",
               "method": "read = 128*RDREQ_128B + 64*RDREQ_64B + 32*RDREQ_32B, write = 64*WRREQ_64B + 32*(WRREQ - WRREQ_64B); traffic_bytes_fetch_size_method = "
                         "(2*FETCH_SIZE + WRITE_SIZE) KB, the guide's gfx950 correction, for comparison",
               "kernels": kern}, open(opath, "w"), indent=1)
    print(json.dumps(kern, indent=1))


if __name__ == "__main__":
    main()
