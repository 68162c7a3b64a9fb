#!/bin/bash
# on the GPU box: every kernel of ONE span frame of rank 0's 1/N tile shard with its start offset, duration and the idle time before it
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
N=${1:-8}; shift
OUT=gpurun_out/sh_timeline_$N
rm -rf $OUT; mkdir -p $OUT
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $OUT -o t -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras --emulate-shard $N $@ > $OUT/log.txt 2>&1
python3 - $OUT <<'PY'
import csv, sys, glob
out = sys.argv[1]
f = glob.glob(out + "/**/t_kernel_trace.csv", recursive=True)[0]
ev = []
for r in csv.DictReader(open(f)):
    n = r["Kernel_Name"]
    short = "k_" + n.split("k_")[1].split("(")[0] if "k_" in n else n[:30]
    ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short))
ev.sort()
frames, cur = [], []
for e in ev:
    if e[2].startswith("k_raygen") and cur:
        frames.append(cur); cur = []
    cur.append(e)
frames.append(cur)
fr = frames[-3]
t0 = fr[0][0]; end = t0; lines = []
tot_busy = 0; tot_gap = 0
for s, e, n in fr:
    gap = s - end
    lines.append("%8.1f us  +%6.1f gap  %7.1f us  %s" % ((s - t0) / 1e3, gap / 1e3, (e - s) / 1e3, n))
    tot_busy += e - s; tot_gap += max(gap, 0); end = max(end, e)
open(out + "/timeline.txt", "w").write("\n".join(lines) + "\nkernel time %.1f us, gaps %.1f us, span %.1f us\n" % (tot_busy / 1e3, tot_gap / 1e3, (end - t0) / 1e3))
print(open(out + "/timeline.txt").read())
PY
