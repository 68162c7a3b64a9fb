"""durations of the k_shade launches by their position inside a wavefront, from a rocprofv3 kernel trace: tools/dev/r06_first_shade.py <kernel_trace.csv>"""
import csv, sys, statistics as st
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
per_q = {}
for r in rows:
    n = r["Kernel_Name"]
    q = r["Queue_Id"]
    if "k_raygen" in n: per_q[q] = 0
    elif "k_shade" in n and q in per_q:
        per_q.setdefault(("d", per_q[q]), []).append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1000.0)
        per_q[q] += 1
for k in sorted(x for x in per_q if isinstance(x, tuple)):
    d = per_q[k]
    print("k_shade #%d of a wavefront: %4d launches, median %.1f us, min %.1f" % (k[1], len(d), st.median(d), min(d)))
