cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python tools/rainbow_update_trace.py 300 2>&1 | tail -1
rocprofv3 --kernel-trace -d gpurun_out/tru -o t --output-format csv -- python tools/rainbow_update_trace.py 60 > /dev/null 2>&1
f=$(find gpurun_out/tru -name "*kernel_trace.csv" | head -1)
python - $f <<'PY'
import csv,sys,re
rows=list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
marks=[i for i,r in enumerate(rows) if "k_sample_gather" in r["Kernel_Name"] or "k_sample" in r["Kernel_Name"]]
print(len(marks))
a,b=marks[-3],marks[-2]
t0=int(rows[a]["Start_Timestamp"])
for r in rows[a:b]:
    s,e=int(r["Start_Timestamp"]),int(r["End_Timestamp"])
    nm=(re.findall(r"k_\w+(?:<[^>]*>)?",r["Kernel_Name"]) or [r["Kernel_Name"][:40]])[0]
    print("%8.1f %7.1f q%s %s grid=%s wg=%s"%((s-t0)/1e3,(e-s)/1e3,r["Queue_Id"],nm,r.get("Grid_Size_X","?"),r.get("Workgroup_Size_X","?")))
PY
rm -rf gpurun_out/tru
