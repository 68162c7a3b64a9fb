#!/usr/bin/env python3
"""usage: limits_from_pmc.py <tag> <prof dir>   (run by tools/profile.sh on the GPU box)

Reduces the round's rocprofv3 passes to the two small files bench.py attaches to its JSON line:
  traffic.json : fabric-side bytes per k_trace / k_shade launch (FETCH_SIZE x 2 + WRITE_SIZE, the gfx950 correction of
                 /opt/skills/guides/MI355X_MICROARCH.md §HBM: FETCH_SIZE tallies 128-B requests at 64 B; both in KiB)
  limits.json  : what bounds k_trace, each as a fraction of its own ceiling:
     valu_issue   wave-level VALU instructions (SQ_INSTS_VALU) x issue cycles per instruction / (SIMDs x kernel cycles);
                  issue cycles from tools/microbench/valu_rate.hip: 1/0.39 for v_fma/mul/add/sub/and/add_u32/mov,
                  1/0.23 for everything else; the static split of the k_trace loop (ISA of the node test) is 36 % / 64 %
     gather_path  bytes gathered per launch (nodes + triangles, the algorithmic figure) / launch time, against the
                  9.7-13.5 TB/s a fully divergent dwordx4 gather sustains (tools/microbench/gather_rate.hip)
     fabric       traffic.json bytes / launch time / 8 TB/s
     tcc_hit      TCC_HIT / (TCC_HIT + TCC_MISS)
"""
import csv
import json
import os
import re
import sys
from collections import defaultdict

import hashlib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def kernel_source_hash():
    """sha256 (16 hex digits) over the device code the counters were measured on: bench.py compares it with the tree it runs from and marks the replayed
    figures `stale` when they differ (the same function lives in bench.py)"""
    h = hashlib.sha256()
    for f in ("kernels.h", "device_math.h"):
        h.update(open(os.path.join(ROOT, "loupiote_amd", "csrc", f), "rb").read())
    return h.hexdigest()[:16]


CLOCK_HZ = 2.4e9      # MI355X peak engine clock
SIMDS = 256 * 4
FAST_RATE, SLOW_RATE = 0.40, 0.234  # wave-instructions per REAL shader cycle per SIMD at 8 waves/SIMD (valu_rate under rocprofv3: profiles/r03_valu_rate_real_clock.txt)
FAST_SHARE = 0.36                   # of k_trace's VALU instructions (85 fast + 150 slow per node visit)
GATHER_TBS = (9.7, 13.5)


def short(name):
    # k_trace<STATS, PIPE, TAIL>: the first template argument (the STATS variant is a different kernel for the averages), not the second
    m = re.search(r"lptd::(k_[a-z_]+)(<(true|false)(, (true|false))*>)?", name)
    return (m.group(1) + ("<%s>" % m.group(3) if m.group(3) else "")) if m else name[:40]


def main(tag, out):
    agg = defaultdict(lambda: defaultdict(float))
    launches = defaultdict(set)
    for dirpath, _, files in os.walk(out):
        for f in files:
            if f.endswith("counter_collection.csv"):
                for row in csv.DictReader(open(os.path.join(dirpath, f))):
                    k = short(row["Kernel_Name"])
                    agg[k][row["Counter_Name"]] += float(row["Counter_Value"])
                    launches[(k, row["Counter_Name"])].add(row["Dispatch_Id"])

    def per_launch(k, c):
        n = len(launches[(k, c)])
        return agg[k][c] / n if n else None

    solo = {}
    p = os.path.join(out, tag + "_solo_kernel_stats.csv")
    if os.path.exists(p):
        for row in csv.DictReader(open(p)):
            solo[short(row["Name"])] = (float(row["AverageNs"]), int(row["Calls"]))
    bench = {}
    p = os.path.join(out, tag + "_solo_bench.json")
    if os.path.exists(p):
        try:
            bench = json.loads(open(p).read().strip())
        except Exception:
            bench = {}
    traffic = {"_comment": "fabric-side bytes per launch from rocprofv3 --pmc (separate passes, tools/profile.sh): FETCH_SIZE (KiB; gfx950 tallies 128-B requests at "
                           "64 B, so doubled, MI355X_MICROARCH.md §HBM) + WRITE_SIZE (KiB).  The BVH is L2 / Infinity-Cache resident: this is fabric traffic, not "
                           "necessarily DRAM traffic.", "round": tag, "source": "profiles/traffic.json@" + tag, "kernel_source_hash": kernel_source_hash()}
    for k, key in (("k_trace<false>", "k_trace"), ("k_shade<false>", "k_shade")):
        f, w = per_launch(k, "FETCH_SIZE"), per_launch(k, "WRITE_SIZE")
        if f is not None and w is not None:
            traffic[key + "_fetch_kib_per_launch"] = f
            traffic[key + "_write_kib_per_launch"] = w
            traffic[key + "_bytes_per_launch"] = int((2.0 * f + w) * 1024)
            traffic[key + "_launches_profiled"] = len(launches[(k, "FETCH_SIZE")])
    json.dump(traffic, open(os.path.join(out, "traffic.json"), "w"), indent=1)

    lim = {"round": tag, "source": "profiles/limits.json@" + tag, "kernel": "k_trace<false>", "clock_hz": CLOCK_HZ, "kernel_source_hash": kernel_source_hash()}
    k = "k_trace<false>"
    avg_ns = solo.get(k, (None, 0))[0]
    if avg_ns:
        lim["solo_avg_launch_ms_rocprof"] = avg_ns / 1e6
        cycles = avg_ns * 1e-9 * CLOCK_HZ
        g = per_launch(k, "GRBM_GUI_ACTIVE")   # summed over the 8 XCDs: busy cycles of the launch at the clock it actually ran at
        v = per_launch(k, "SQ_INSTS_VALU")
        if v:
            issue = v * (FAST_SHARE / FAST_RATE + (1 - FAST_SHARE) / SLOW_RATE)
            lim["valu_issue"] = {"frac": issue / (SIMDS * (g / 8.0 if g else cycles)), "frac_at_nominal_clock": issue / (SIMDS * cycles),
                                 "valu_wave_insts_per_launch": v, "insts_per_clk_per_simd": v / (SIMDS * (g / 8.0 if g else cycles)),
                                 "busy_cycles_per_launch": (g / 8.0 if g else None), "fast_share": FAST_SHARE,
                                 "rates_inst_per_clk_per_simd": [FAST_RATE, SLOW_RATE],
                                 "ceiling_inst_per_clk_per_simd": 1.0 / (FAST_SHARE / FAST_RATE + (1 - FAST_SHARE) / SLOW_RATE)}
        r = bench.get("roofline", {})
        if r.get("bytes_per_launch"):
            gb = r["bytes_per_launch"] - r.get("rays_per_launch", 0) * 48.0   # nodes + triangles only
            tbs = gb / (avg_ns * 1e-9) / 1e12
            lim["gather_path"] = {"tb_per_s": tbs, "frac_of_13.5": tbs / GATHER_TBS[1], "frac_of_9.7": tbs / GATHER_TBS[0], "ceiling_tb_per_s": list(GATHER_TBS)}
        if "k_trace_bytes_per_launch" in traffic:
            lim["fabric"] = {"bytes_per_launch": traffic["k_trace_bytes_per_launch"], "frac": traffic["k_trace_bytes_per_launch"] / (avg_ns * 1e-9) / 8e12}
    # lane efficiency: of the lane slots of the VALU instructions issued, how many carried a live lane (tracked)
    tc, vi = per_launch(k, "SQ_THREAD_CYCLES_VALU"), per_launch(k, "SQ_INSTS_VALU")
    if tc and vi:
        lim["lane_efficiency"] = {"k_trace": tc / (64.0 * vi), "what": "SQ_THREAD_CYCLES_VALU / (64 x SQ_INSTS_VALU)"}
        tcs, vis = per_launch("k_shade<false>", "SQ_THREAD_CYCLES_VALU"), per_launch("k_shade<false>", "SQ_INSTS_VALU")
        if tcs and vis:
            lim["lane_efficiency"]["k_shade"] = tcs / (64.0 * vis)
    h, m = per_launch(k, "TCC_HIT_sum"), per_launch(k, "TCC_MISS_sum")
    if h is not None and m is not None and h + m > 0:
        lim["tcc_hit"] = h / (h + m)
    # k_shade against the HBM roof: it is the kernel that IS bound by fabric bytes
    ks = solo.get("k_shade<false>", (None, 0))[0]
    if ks and "k_shade_bytes_per_launch" in traffic:
        tb = traffic["k_shade_bytes_per_launch"] / (ks * 1e-9) / 1e12
        lim["k_shade_fabric"] = {"bytes_per_launch": traffic["k_shade_bytes_per_launch"], "avg_launch_ms": ks / 1e6, "tb_per_s": tb, "frac_of_8": tb / 8.0, "frac_of_6.29_achievable": tb / 6.29}
    k2 = "k_shade<false>"
    h, m = per_launch(k2, "TCC_HIT_sum"), per_launch(k2, "TCC_MISS_sum")
    if h is not None and m is not None and h + m > 0:
        lim["k_shade_tcc_hit"] = h / (h + m)
    # the whole frame against the VALU issue bound: a frame = 8 k_trace + 8 k_shade launches + the packet launch of bounce 0 (depth 8), the other kernels are < 1 %
    vt, vs = per_launch(k, "SQ_INSTS_VALU"), per_launch(k2, "SQ_INSTS_VALU")
    gt = per_launch(k, "GRBM_GUI_ACTIVE")
    b2 = {}
    # the frame time of an UN-profiled run when there is one (rocprofv3's kernel trace slows the frame by ~15 %): <tag>_bench_full.json
    # beside the pass directories or under profiles/; else the bench line of the trace pass
    for p2 in (os.path.join(os.path.dirname(out.rstrip("/")), tag + "_bench_full.json"), os.path.join("profiles", tag + "_bench_full.json"), os.path.join(out, tag + "_bench.json")):
        if os.path.exists(p2):
            try:
                b2 = json.loads([l for l in open(p2) if l.startswith("{")][-1])
                b2["_source"] = p2
                break
            except Exception:
                b2 = {}
    if vt and vs and gt and avg_ns and b2.get("ms_per_frame"):
        clock = (gt / 8.0) / (avg_ns * 1e-9)                      # Hz the PMC pass actually ran at
        ceiling = 1.0 / (FAST_SHARE / FAST_RATE + (1 - FAST_SHARE) / SLOW_RATE)
        vp = per_launch("k_trace_packet<false>", "SQ_INSTS_VALU") or 0.0
        insts = 8.0 * vt + 8.0 * vs + vp
        bound_ms = insts / SIMDS / ceiling / clock * 1e3
        lim["frame_valu"] = {"valu_wave_insts_per_frame": insts, "k_trace_share": 8.0 * vt / insts, "k_trace_packet_share": vp / insts, "bound_ms_per_frame": bound_ms,
                             "measured_ms_per_frame": b2["ms_per_frame"], "measured_from": b2.get("_source"), "frac": bound_ms / b2["ms_per_frame"], "clock_hz": clock,
                             "note": "all VALU wave-instructions of a frame (8 k_trace + 8 k_shade launches + the packet launch of bounce 0) at the issue ceiling of k_trace's mix vs the "
                                     "measured frame (the SURVEY 8d span: one renderer, read-back included)"}
    wv, wc, bc = per_launch(k, "SQ_WAVES"), per_launch(k, "SQ_WAVE_CYCLES"), per_launch(k, "SQ_BUSY_CYCLES")
    if wv:
        lim["waves_per_launch"] = wv
    # wave occupancy (north star: "wave occupancy counters"): SQ_WAVE_CYCLES counts quad-cycles of resident waves; against the launch's busy cycles
    # (GRBM_GUI_ACTIVE / 8 XCDs) on 1024 SIMDs that is the average number of waves resident per SIMD (6 = full for k_trace, 4 for k_shade)
    occ = {}
    for kk, cap in ((k, 6), (k2, 4)):   # resident waves per SIMD the kernels' VGPR counts allow: k_trace<., PIPE> 78 -> 6, k_shade 127 -> 4
        wcy, gg = per_launch(kk, "SQ_WAVE_CYCLES"), per_launch(kk, "GRBM_GUI_ACTIVE")
        if wcy and gg:
            occ[kk] = {"waves_per_simd": 4.0 * wcy / ((gg / 8.0) * SIMDS), "max_waves_per_simd": cap,
                       "wait_any_share": (per_launch(kk, "SQ_WAIT_ANY") or 0.0) / wcy, "issue_stall_share": (per_launch(kk, "SQ_WAIT_INST_ANY") or 0.0) / wcy,
                       "valu_active_share": (per_launch(kk, "SQ_ACTIVE_INST_VALU") or 0.0) / wcy}
    if occ:
        occ["what"] = "waves_per_simd = 4 x SQ_WAVE_CYCLES / (GRBM_GUI_ACTIVE / 8 x 1024 SIMDs); shares = of a resident wave's cycles: parked on s_waitcnt (memory), issue-stalled, issuing VALU"
        lim["occupancy"] = occ
    json.dump(lim, open(os.path.join(out, "limits.json"), "w"), indent=1)
    print(json.dumps(lim, indent=1))
    print(json.dumps({k: v for k, v in traffic.items() if not k.startswith("_")}, indent=1))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
