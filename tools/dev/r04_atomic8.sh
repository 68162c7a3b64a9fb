#!/bin/bash
# timing probe: k_shade's compaction atomic on eight words (by block) instead of one; only the FIRST k_shade launch of a frame is comparable
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
for v in atomic8; do
  for sh in 0 8; do
    OUT=gpurun_out/atomic8/${v}_$sh
    rm -rf $OUT; mkdir -p $OUT
    if [ $v = atomic8 ]; then export LPT_LIB_PATH=$GRAFT_REPO_ROOT/loupiote_amd/libloupiote_hip_atomic8.so; else unset LPT_LIB_PATH; fi
    A=""; [ $sh != 0 ] && A="--emulate-shard $sh"
    timeout 120 rocprofv3 --kernel-trace --output-format csv -d $OUT -o t -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-extras --lanes 1 --max-fused 4 $A > $OUT/log.txt 2>&1
    python3 - $OUT $v $sh <<'PY'
import csv, sys, glob
out, v, sh = sys.argv[1:4]
f = glob.glob(out + "/**/t_kernel_trace.csv", recursive=True)[0]
ev = []
for r in csv.DictReader(open(f)):
    n = r["Kernel_Name"]
    if "k_" not in n: continue
    ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "k_" + n.split("k_")[1].split("(")[0].split("<")[0]))
ev.sort()
first, prev = [], None
for s, e, n in ev:
    if n == "k_shade" and prev in ("k_trace_packet", "k_trace") and first is not None:
        pass
    prev = n
# the first k_shade after each k_raygen
seen_raygen = False
for s, e, n in ev:
    if n == "k_raygen": seen_raygen = True
    elif n == "k_shade" and seen_raygen:
        first.append((e - s) / 1e3); seen_raygen = False
first.sort()
print("%s shard %s: first k_shade of a wavefront: n %d, median %.1f us, min %.1f" % (v, sh, len(first), first[len(first) // 2], first[0]))
PY
  done
done
