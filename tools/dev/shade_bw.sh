#!/bin/bash
# on the GPU box: fabric bytes (FETCH_SIZE x2 + WRITE_SIZE) and duration of k_shade at several frame sizes — does the
# shading pass run faster per byte when a frame's streaming ray queues fit the 256 MB Infinity Cache beside the textures?
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
for wh in "1920 1080" "960 540" "480 270"; do
  set -- $wh
  OUT=gpurun_out/shbw_$1
  rm -rf $OUT; mkdir -p $OUT
  timeout 150 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT -o p -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-extras --frames-per-step 3 --width $1 --height $2 $EXTRA > $OUT/log.txt 2>&1
  python3 - $OUT $1 $2 <<'PY'
import csv, sys, collections
out, w, h = sys.argv[1:4]
dur = collections.defaultdict(list)
for r in csv.DictReader(open(out + "/p_kernel_trace.csv")):
    n = r["Kernel_Name"]
    k = "k_shade" if "k_shade" in n else ("k_trace" if "k_trace<false>" in n.replace("(lptd", "<") or "k_traceILb0" in n or "k_trace<false" in n else None)
    if k: dur[k].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
cnt = collections.defaultdict(lambda: collections.defaultdict(float)); nl = collections.defaultdict(set)
for r in csv.DictReader(open(out + "/p_counter_collection.csv")):
    n = r["Kernel_Name"]
    k = "k_shade" if "k_shade" in n else ("k_trace" if "k_trace<false" in n else None)
    if k:
        cnt[k][r["Counter_Name"]] += float(r["Counter_Value"]); nl[k].add(r["Dispatch_Id"])
for k in ("k_shade", "k_trace"):
    if not dur[k]: continue
    n = len(nl[k]); c = cnt[k]
    byts = (c["FETCH_SIZE"] * 2) * 1024 / n   # reads only (one counter per pass: a set the hardware cannot collect hangs rocprofv3)
    ms = sum(dur[k]) / len(dur[k]) / 1e6
    print("%sx%s %-8s launches %3d  avg %.4f ms  fabric %.1f MB/launch  -> %.2f TB/s (fetch only)" % (w, h, k, n, ms, byts / 1e6, byts / (ms * 1e-3) / 1e12))
PY
done
