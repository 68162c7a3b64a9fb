#!/usr/bin/env python3
"""Parse `hipcc -Rpass-analysis=kernel-resource-usage` stderr into one line per kernel."""
import re
import sys
cur = None
rows = {}
for line in open(sys.argv[1]):
    m = re.search(r"Function Name: (\S+)", line)
    if m:
        cur = re.sub(r"^_ZN4lptd\d+", "", m.group(1))[:22]
        rows[cur] = {}
    for k in ["VGPRs", "TotalSGPRs", "Occupancy [waves/SIMD]", "LDS Size [bytes/block]", "ScratchSize [bytes/lane]"]:
        m = re.search(re.escape(k) + r": (\d+)", line)
        if m and cur:
            rows[cur][k.split(" ")[0]] = m.group(1)
for k, v in rows.items():
    print(k.ljust(24), v)
