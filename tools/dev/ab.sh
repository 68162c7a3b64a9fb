#!/bin/bash
# usage (on the GPU box): tools/dev/ab.sh <variant> [<variant> ...]   — `base` = the shipped library
# Runs the bench's timed region (no extras, no CPU baseline) for every library variant built by `make variant` and prints
# ms/frame and the per-stage times; one JSON line per variant in gpurun_out/ab_<variant>.json.
for v in "$@"; do
  if [ "$v" = base ]; then unset LPT_LIB_PATH; else export LPT_LIB_PATH=$PWD/loupiote_amd/libloupiote_hip_$v.so; fi
  python bench.py --steps 3 --warmup 1 --no-extras --no-cpu-baseline > gpurun_out/ab_$v.json 2> gpurun_out/ab_$v.err || { echo "$v FAILED"; tail -5 gpurun_out/ab_$v.err; continue; }
  python - "$v" <<'PY'
import json, sys
v = sys.argv[1]
j = json.loads([l for l in open("gpurun_out/ab_%s.json" % v) if l.startswith("{")][-1])
st = j["stage_ms_per_frame"]
print("%-12s value %7.1f  ms/frame %7.3f  trace %6.3f  shade %6.3f  checksum %.6f  rays %d" % (v, j["value"], j["ms_per_frame"], st["intersection"] + st["shadow"], st["shading"], j["config"]["frame_checksum"], j["config"]["rays_per_frame"]))
PY
done
