#!/bin/bash
# on the GPU box: kernel + memory-copy timeline of a few span frames; prints per-frame gaps (where does the time outside kernels go?)
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/timeline
rm -rf $OUT; mkdir -p $OUT
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $OUT -o t -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-extras --frames-per-step 4 $@ > $OUT/log.txt 2>&1
python3 - <<'PY'
import csv
ev = []
for r in csv.DictReader(open("gpurun_out/timeline/t_kernel_trace.csv")):
    n = r["Kernel_Name"]
    short = "k_" + n.split("k_")[1].split("(")[0].split("<")[0] if "k_" in n else n[:30]
    ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short, r.get("Stream_Id", r.get("Queue_Id", "?"))))
ev.sort()
# frames: split at k_raygen that follows a D2H copy of >= 1 MB duration... simply: a frame starts at each first k_raygen after a COPY
frames, cur = [], []
for e in ev:
    if e[2].startswith("k_raygen") and cur and any(x[2].startswith("k_resolve") for x in cur):
        frames.append(cur); cur = []
    cur.append(e)
frames.append(cur)
for f in frames[-4:-1]:
    t0 = f[0][0]
    busy_end = t0
    gaps = []
    for s, e, n, q in f:
        if s > busy_end + 20000: gaps.append((round((busy_end - t0) / 1e6, 3), round((s - busy_end) / 1e6, 3), n))
        busy_end = max(busy_end, e)
    print("frame: %d events, span %.3f ms; idle gaps > 20 us (at ms, length ms, next): %s" % (len(f), (busy_end - t0) / 1e6, gaps))
    print("   tail:", [(n, round((s - t0) / 1e6, 3), round((e - s) / 1e6, 3)) for s, e, n, q in f[-6:]])
PY
