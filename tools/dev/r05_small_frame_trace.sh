cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/small_trace; rm -rf $OUT; mkdir -p $OUT
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $OUT -o t -- python3 bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-extras --width 64 --height 36 > $OUT/log.txt 2>&1
python3 - $OUT <<'PY'
import csv, sys, glob
f = glob.glob(sys.argv[1] + "/**/t_kernel_trace.csv", recursive=True)[0]
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
fr = frames[len(frames) // 2]
t0 = fr[0][0]; end = t0
for s, e, n in fr:
    print("%8.1f us  +%6.1f gap  %7.1f us  %s" % ((s - t0) / 1e3, (s - end) / 1e3, (e - s) / 1e3, n)); end = e
nxt = frames[len(frames) // 2 + 1][0][0]
print("frame period %.1f us; kernels+copies end at %.1f" % ((nxt - t0) / 1e3, (end - t0) / 1e3))
PY
grep -o '"ms_per_frame": [0-9.]*' $OUT/log.txt | head -2
