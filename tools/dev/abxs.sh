#!/bin/bash
# usage (on the GPU box): tools/dev/abxs.sh "<name>|<env assignments>|<bench args>" ...
# Like abx.sh with the bench's extras on: the span (4 steps after 2 warm-up steps), the un-overlapped k_trace launch time and
# the traversal statistics of the STATS kernel.
for spec in "$@"; do
  IFS='|' read -r name envs args <<< "$spec"
  ( for e in $envs; do export "$e"; done
    python bench.py --steps 4 --warmup 2 --no-cpu-baseline $args > gpurun_out/abxs_$name.json 2> gpurun_out/abxs_$name.err || { echo "$name FAILED"; tail -5 gpurun_out/abxs_$name.err; } )
  python - "$name" <<'PY'
import json, sys
v = sys.argv[1]
try:
    j = json.loads([l for l in open("gpurun_out/abxs_%s.json" % v) if l.startswith("{")][-1])
    r = j["roofline"]; w = r["wave"]; a = j["accel"]
    print("%-10s ms/frame %7.3f  k_trace solo %.4f ms  nodes/ray %.3f tris/ray %.3f  sh nodes %.3f tris %.3f | live %.1f node %.1f tri %.1f  slots/ray %.2f | %d nodes depth %d build %.0f ms  sum %.3f" % (
        v, j["ms_per_frame"], r["avg_launch_ms"], r["nodes_per_ray"], r["tris_per_ray"], r["shadow_nodes_per_ray"], r["shadow_tris_per_ray"],
        w["live_lanes_per_step"], w["node_lanes_per_step"], w["tri_lanes_per_step"], w["lane_slots_per_ray"], a["nodes"], a["depth"], a["build_ms"], j["config"]["frame_checksum"]))
except Exception as e:
    print(v, "no result", e)
PY
done
