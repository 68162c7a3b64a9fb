"""per-launch durations of the traversal kernels of a shard frame, from a rocprofv3 --kernel-trace CSV: tools/dev/r06_trace_launches.py <kernel_trace.csv> [frames]
prints, per position of the launch inside a frame (9 k_trace launches per 4-spp depth-8 frame), the median duration in us"""
import csv, sys, statistics
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
names = [r["Kernel_Name"] for r in rows]
# a frame starts at k_raygen
frames, cur = [], None
for r in rows:
    n = r["Kernel_Name"]
    if "k_raygen" in n:
        cur = []
        frames.append(cur)
    if cur is not None:
        cur.append((n.split("(")[0].replace("void lptd::", "")[:40], (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1000.0, int(r["Start_Timestamp"]), int(r["End_Timestamp"])))
frames = [f for f in frames if len(f) == len(frames[-2])][-int(sys.argv[2]) if len(sys.argv) > 2 else -8:]
print("%d frames of %d launches" % (len(frames), len(frames[0])))
for i in range(len(frames[0])):
    d = [f[i][1] for f in frames]
    gap = [(f[i][2] - f[i - 1][3]) / 1000.0 for f in frames] if i else [0.0]
    print("%2d %-42s %8.1f us   gap before %6.1f us" % (i, frames[0][i][0], statistics.median(d), statistics.median(gap)))
print("sum of durations %.1f us, span %.1f us" % (statistics.median([sum(x[1] for x in f) for f in frames]), statistics.median([(f[-1][3] - f[0][2]) / 1000.0 for f in frames])))
