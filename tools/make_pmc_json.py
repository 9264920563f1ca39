#!/usr/bin/env python3
"""profiles/latest_pmc.json from the two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; KB per dispatch).
usage: make_pmc_json.py <workload> <fetch.csv> <write.csv> <out.json>
gfx950 correction (MI355X_MICROARCH.md, HBM/rocprofv3 section; confirmed here on k_copy: 524300 KB counted for a 1 GiB read):
FETCH_SIZE tallies 16-B/lane reads at half -> doubled; WRITE_SIZE is exact."""
import csv
import json
import sys

NAMES = {"k_inter(": "inter", "k_alf(": "alf", "k_addb<0>(": "dbk_v", "k_addb<1>(": "dbk_h", "k_dbk<0>(": "dbk_v", "k_dbk<1>(": "dbk_h",
         "k_itdq(": "itdq", "k_intra<": "intra", "k_affine_": "affine", "k_pad(": "pad"}


def load(path):
    out = {}
    for r in csv.DictReader(open(path)):
        for k, v in NAMES.items():
            if k in r["kernel"]:
                out[v] = out.get(v, 0.0) + float(r["avg"])
    return out


def main():
    wl, fpath, wpath, opath = sys.argv[1:5]
    f, w = load(fpath), load(wpath)
    kern = {k: {"fetch_kb": round(f.get(k, 0.0), 1), "write_kb": round(w.get(k, 0.0), 1),
                "traffic_bytes": int((2.0 * f.get(k, 0.0) + w.get(k, 0.0)) * 1024)} for k in sorted(set(f) | set(w))}
    json.dump({"workload": wl, "source": f"{fpath} + {wpath} (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes)",
               "correction": "FETCH_SIZE doubled (gfx950: 16-B/lane reads tallied at half, confirmed on k_copy: 524300 KB for 1 GiB); WRITE_SIZE as is",
               "kernels": kern}, open(opath, "w"), indent=1)
    print(json.dumps(kern, indent=1))


if __name__ == "__main__":
    main()
