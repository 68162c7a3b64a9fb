"""summarise the printf probe of the tail path (kernels.h LPT_PROBE_TAIL_PRINT): tools/dev/r06_tail_probe.py <stdout of a bench run with the probe library>"""
import re, sys, statistics as st
waves, rays = [], []
for line in open(sys.argv[1], errors="replace"):
    m = re.match(r"TAILWAVE blk (\d+) cb (-?\d+) live (\d+) loop_us ([\d.]+) tail_us ([\d.]+)", line)
    if m: waves.append((int(m[1]), int(m[2]), int(m[3]), float(m[4]), float(m[5]))); continue
    m = re.match(r"TAILRAY blk (\d+) r (\d+) of (\d+) shadow (\d) frontier (\d+) nodes (\d+) leaves (\d+) us ([\d.]+)", line)
    if m: rays.append((int(m[1]), int(m[2]), int(m[3]), int(m[4]), int(m[5]), int(m[6]), int(m[7]), float(m[8])))
print("%d wave records, %d ray records" % (len(waves), len(rays)))
for cb in sorted(set(w[1] for w in waves)):
    ws = [w for w in waves if w[1] == cb]
    lo = [w[3] for w in ws]; ta = [w[4] for w in ws]; tot = [w[3] + w[4] for w in ws]
    print("cb %2d: %4d waves  live mean %.2f  loop us median %.1f max %.1f   tail us median %.1f p90 %.1f max %.1f   loop+tail max %.1f" % (
        cb, len(ws), st.mean(w[2] for w in ws), st.median(lo), max(lo), st.median(ta), sorted(ta)[int(0.9 * len(ta))], max(ta), max(tot)))
if rays:
    us = sorted(r[7] for r in rays)
    print("rays: us median %.2f p90 %.2f p99 %.2f max %.2f; nodes median %d p90 %d max %d; frontier median %d max %d; us per node (sum/sum) %.3f" % (
        st.median(us), us[int(0.9 * len(us))], us[int(0.99 * len(us))], us[-1], st.median(r[5] for r in rays), sorted(r[5] for r in rays)[int(0.9 * len(rays))], max(r[5] for r in rays),
        st.median(r[4] for r in rays), max(r[4] for r in rays), sum(us) / max(1, sum(r[5] for r in rays))))
    worst = sorted(rays, key=lambda r: -r[7])[:8]
    for r in worst: print("   worst: blk %d r %d/%d shadow %d frontier %d nodes %d leaves %d us %.2f" % r)
