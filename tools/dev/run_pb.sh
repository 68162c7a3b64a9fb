set -u
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r02d
export PB_SPP=1
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/r02d/pb -o pb -- python3 tools/dev/per_bounce.py > gpurun_out/r02d/per_bounce.log 2>&1
python3 tools/dev/per_bounce_join.py gpurun_out/r02d/per_bounce.log $(find gpurun_out/r02d/pb -name "*kernel_trace.csv" | head -1) | tee gpurun_out/r02d/per_bounce.txt
python3 - <<'PY'
import csv,glob
rows=list(csv.DictReader(open(glob.glob("gpurun_out/r02d/pb/**/*kernel_trace.csv",recursive=True)[0])))
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
# last frame: from last k_raygen to the end
idx=[i for i,r in enumerate(rows) if "k_raygen" in r["Kernel_Name"]][-1]
t0=int(rows[idx]["Start_Timestamp"]); prev_end=t0
tot_k=0; tot_gap=0
for r in rows[idx:]:
    s,e=int(r["Start_Timestamp"]),int(r["End_Timestamp"])
    name=r["Kernel_Name"].split("(")[0][-40:]
    print("%-42s start %8.1f us dur %7.1f us gap %5.1f us" % (name,(s-t0)/1e3,(e-s)/1e3,(s-prev_end)/1e3))
    tot_k+=e-s; tot_gap+=max(0,s-prev_end); prev_end=e
print("kernels %.1f us gaps %.1f us total %.1f us" % (tot_k/1e3,tot_gap/1e3,(prev_end-t0)/1e3))
PY
