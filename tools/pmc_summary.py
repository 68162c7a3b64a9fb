#!/usr/bin/env python3
"""Aggregate rocprofv3 --pmc counter_collection CSVs per kernel: sum and per-launch mean of every counter."""
import csv
import re
import sys
from collections import defaultdict


def short(name):
    # k_trace<STATS, PIPE, TAIL>: keyed by the first template argument (the STATS variant is a different kernel for the averages)
    m = re.search(r"lptd::(k_[a-z_]+)(<(true|false)(, (true|false))*>)?", name)
    if m:
        return m.group(1) + ("<%s>" % m.group(3) if m.group(3) else "")
    return name[:40]


def main(paths):
    agg = defaultdict(lambda: defaultdict(float))
    launches = defaultdict(set)
    for path in paths:
        with open(path) as f:
            for row in csv.DictReader(f):
                k = short(row["Kernel_Name"])
                agg[k][row["Counter_Name"]] += float(row["Counter_Value"])
                launches[(k, row["Counter_Name"])].add(row["Dispatch_Id"])
    for k in sorted(agg):
        print("== %s" % k)
        for c in sorted(agg[k]):
            n = len(launches[(k, c)])
            print("   %-28s total %.6g   per-launch %.6g   (launches %d)" % (c, agg[k][c], agg[k][c] / max(n, 1), n))


if __name__ == "__main__":
    main(sys.argv[1:])
