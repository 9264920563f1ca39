#!/usr/bin/env python3
"""Per-kernel average of one PMC counter from a rocprofv3 results.db.  usage: rocpd_pmc.py <results.db> [out.csv]"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
rows = db.execute("select kernel_name, counter_name, count(*), avg(value), min(value), max(value) from counters_collection "
                  "group by kernel_name, counter_name order by 4 desc").fetchall()
lines = ["kernel,counter,dispatches,avg,min,max"]
for r in rows:
    lines.append(f"\"{r[0]}\",{r[1]},{r[2]},{r[3]:.1f},{r[4]:.1f},{r[5]:.1f}")
txt = "\n".join(lines)
if len(sys.argv) > 2:
    open(sys.argv[2], "w").write(txt + "\n")
print(txt)
