#!/bin/bash
# on the GPU box: the kernels of a span frame of rank 0's 1/N tile shard, each launch's duration averaged over the frames of the timed region
# tools/dev/r05_launch_table.sh <tag> <N> [bench args]
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
TAG=$1; N=$2; shift; shift
OUT=gpurun_out/launches_$TAG
rm -rf $OUT; mkdir -p $OUT
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $OUT -o t -- python3 bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-extras --emulate-shard $N "$@" > $OUT/log.txt 2>&1
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
def gaps(fr):
    end = fr[0][0]; g = 0
    for s, e, n in fr:
        if n.startswith("__amd"): break
        g += max(0, s - end); end = e
    return g
good = [fr for fr in frames[3:] if [x[2] for x in fr if not x[2].startswith("__amd")] == [x[2] for x in frames[3] if not x[2].startswith("__amd")] and gaps(fr) < 60000]
print("%d frames of the timed region (no stage events)" % len(good))
names = [x[2] for x in good[0] if not x[2].startswith("__amd")]
tot = 0
for i, n in enumerate(names):
    d = sorted((fr[i][1] - fr[i][0]) / 1e3 for fr in good)
    med = d[len(d) // 2]; tot += med
    print("%2d %-34s median %7.1f us  min %7.1f" % (i, n, med, d[0]))
print("sum of medians %.1f us" % tot)
PY
