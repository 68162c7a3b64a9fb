#!/bin/bash
# usage (on the GPU box): tools/dev/abx.sh "<name>|<env assignments>|<bench args>" ...
# One bench timed region (no extras, no CPU baseline) per spec; prints ms/frame and stage times.
for spec in "$@"; do
  IFS='|' read -r name envs args <<< "$spec"
  ( for e in $envs; do export "$e"; done
    python bench.py --steps 3 --warmup 1 --no-extras --no-cpu-baseline $args > gpurun_out/abx_$name.json 2> gpurun_out/abx_$name.err || { echo "$name FAILED"; tail -5 gpurun_out/abx_$name.err; } )
  python - "$name" <<'PY'
import json, sys
v = sys.argv[1]
try:
    j = json.loads([l for l in open("gpurun_out/abx_%s.json" % v) if l.startswith("{")][-1])
    st = j["stage_ms_per_frame"]
    print("%-28s value %7.1f  ms/frame %7.3f  trace %6.3f  shade %6.3f  sum %.4f  rays %d" % (v, j["value"], j["ms_per_frame"], st["intersection"] + st["shadow"], st["shading"], j["config"]["frame_checksum"], j["config"]["rays_per_frame"]))
except Exception as e:
    print(v, "no result", e)
PY
done
