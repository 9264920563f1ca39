R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/kt_tr
timeout -k 5 200 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/kt_tr -o p -- python $R/bench.py --steps 12 --warmup 3 --workload cfg4_main_8k_10b_ra --no-cpu-baseline --no-end-to-end > $R/gpurun_out/kt_tr.log 2>&1
f=$(find $R/gpurun_out/kt_tr -name "*kernel_trace.csv" | head -1)
python - "$f" <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
# print a window in the middle of the timed region (with prepare): rows 200..260
t0=int(rows[0]["Start_Timestamp"])
out=open("/root/repo/gpurun_out/trace_window.txt","w")
for r in rows:
    out.write("%10.1f %10.1f %8.1f q=%s %s\n"%((int(r["Start_Timestamp"])-t0)/1e3,(int(r["End_Timestamp"])-t0)/1e3,(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3,r.get("Queue_Id","?"),r["Kernel_Name"][:40]))
PY
rm -rf $R/gpurun_out/kt_tr
