#!/usr/bin/env python3
"""Summarise a rocprofv3 results.db (rocpd sqlite) per kernel: calls, total/avg/min/max duration.
usage: python tools/rocpd_stats.py <results.db> [out.csv]"""
import sqlite3
import sys


def main():
    db = sqlite3.connect(sys.argv[1])
    cols = [r[1] for r in db.execute("pragma table_info(kernels)")]
    name = "name" if "name" in cols else "kernel_name"
    rows = db.execute(f"select {name}, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) "
                      f"from kernels group by {name} order by 3 desc").fetchall()
    tot = sum(r[2] for r in rows) or 1
    lines = ["kernel,calls,total_us,avg_us,min_us,max_us,percent"]
    for n, c, s, a, mn, mx in rows:
        lines.append(f"\"{n}\",{c},{s / 1e3:.1f},{a / 1e3:.2f},{mn / 1e3:.2f},{mx / 1e3:.2f},{100 * s / tot:.1f}")
    txt = "\n".join(lines)
    if len(sys.argv) > 2:
        open(sys.argv[2], "w").write(txt + "\n")
    print(txt)


if __name__ == "__main__":
    main()
