R=$GRAFT_REPO_ROOT
timeout -k 5 60 $R/tools/ubench/win_bw 2>&1 | head -8
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/pmc_x
timeout -k 5 120 rocprofv3 --pmc TA_TA_BUSY_sum TD_TD_BUSY_sum TCP_TOTAL_ACCESSES_sum TCP_TOTAL_CACHE_ACCESSES_sum -d $R/gpurun_out/pmc_x -o p -- $R/tools/ubench/win_bw > /dev/null 2>&1
python - <<PY
import sqlite3,glob
db=sqlite3.connect(glob.glob("$R/gpurun_out/pmc_x/**/*.db",recursive=True)[0])
rows=db.execute("select dispatch_id, kernel_name, counter_name, value from counters_collection order by dispatch_id").fetchall()
import collections
d=collections.OrderedDict()
for did,k,c,v in rows:
    d.setdefault((did,k[:40]),{})[c]=v
seen=0
for (did,k),v in d.items():
    if 'k_win' in k:
        seen+=1
        if seen in (3,13,23,33): print(did,k,{a:round(b/1e6,2) for a,b in v.items()})
PY
rm -rf $R/gpurun_out/pmc_x
