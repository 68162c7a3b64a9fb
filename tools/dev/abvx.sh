#!/bin/bash
# usage (on the GPU box): tools/dev/abvx.sh "<name>|<variant or base>|<env assignments>|<bench args>" ...
# abxs.sh for a library variant built by `make variant`: span, un-overlapped k_trace launch time, traversal statistics.
for spec in "$@"; do
  IFS='|' read -r name variant envs args <<< "$spec"
  ( for e in $envs; do export "$e"; done
    if [ "$variant" != base ]; then export LPT_LIB_PATH=$PWD/loupiote_amd/libloupiote_hip_$variant.so; fi
    python bench.py --steps 4 --warmup 2 --no-cpu-baseline $args > gpurun_out/abvx_$name.json 2> gpurun_out/abvx_$name.err || { echo "$name FAILED"; tail -5 gpurun_out/abvx_$name.err; } )
  python - "$name" <<'PY'
import json, sys
v = sys.argv[1]
try:
    j = json.loads([l for l in open("gpurun_out/abvx_%s.json" % v) if l.startswith("{")][-1])
    r = j["roofline"]; w = r["wave"]
    print("%-10s ms/frame %7.3f  k_trace solo %.4f ms  nodes/ray %.3f tris/ray %.3f | slots/ray %.2f  sum %.3f" % (
        v, j["ms_per_frame"], r["avg_launch_ms"], r["nodes_per_ray"], r["tris_per_ray"], w["lane_slots_per_ray"], j["config"]["frame_checksum"]))
except Exception as e:
    print(v, "no result", e)
PY
done
