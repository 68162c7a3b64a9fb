#!/usr/bin/env python3
"""usage: per_bounce_join.py counts.json kernel_trace.csv — last frame's k_trace / k_shade launches with their ray counts"""
import csv
import json
import sys

counts = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
rows = list(csv.DictReader(open(sys.argv[2])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
tr = [r for r in rows if "k_trace" in r["Kernel_Name"]][-9:]
sh = [r for r in rows if "k_shade" in r["Kernel_Name"]][-8:]
c, s = counts["closest"], counts["shadow"]
for i, r in enumerate(tr):
    ms = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
    nc = c[i] if i < 8 else 0
    ns = s[i - 1] if i >= 1 else 0
    print("k_trace %d: closest %8d shadow %8d  %.3f ms  %.0f Mrays/s" % (i, nc, ns, ms, (nc + ns) / ms / 1e3))
for i, r in enumerate(sh):
    ms = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
    print("k_shade %d: rays %8d  %.3f ms  %.0f Mrays/s" % (i, c[i], ms, c[i] / ms / 1e3))
