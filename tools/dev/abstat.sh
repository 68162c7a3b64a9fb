#!/bin/bash
# usage (on the GPU box): tools/dev/abstat.sh <variant> ...   — `base` = the shipped library
# Like ab.sh, plus the traversal statistics of the bench line (the STATS kernel on one frame): nodes / triangles per ray,
# lanes per wave-step, solo k_trace launch time.
for v in "$@"; do
  if [ "$v" = base ]; then unset LPT_LIB_PATH; else export LPT_LIB_PATH=$PWD/loupiote_amd/libloupiote_hip_$v.so; fi
  python bench.py --steps 4 --warmup 2 --no-cpu-baseline > gpurun_out/abs_$v.json 2> gpurun_out/abs_$v.err || { echo "$v FAILED"; tail -5 gpurun_out/abs_$v.err; continue; }
  python - "$v" <<'PY'
import json, sys
v = sys.argv[1]
j = json.loads([l for l in open("gpurun_out/abs_%s.json" % v) if l.startswith("{")][-1])
r = j["roofline"]; w = r["wave"]
print("%-10s ms/frame %7.3f  k_trace solo %.4f ms  nodes/ray %.3f tris/ray %.3f  sh nodes %.3f tris %.3f | live %.1f node %.1f tri %.1f  slots/ray %.2f  sum %.3f" % (
    v, j["ms_per_frame"], r["avg_launch_ms"], r["nodes_per_ray"], r["tris_per_ray"], r["shadow_nodes_per_ray"], r["shadow_tris_per_ray"],
    w["live_lanes_per_step"], w["node_lanes_per_step"], w["tri_lanes_per_step"], w["lane_slots_per_ray"], j["config"]["frame_checksum"]))
PY
done
