#!/bin/bash
# tiny frames (fewer rays than resident waves): every ray traced by a whole wave (per-bounce launches, step budget 1 -> k_trace_coop) against k_path / the per-lane launches
OUT=gpurun_out/$1; mkdir -p $OUT
run() {
  local name=$1; shift
  timeout 400 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras "$@" > $OUT/$name.json 2> $OUT/$name.err
  python - <<PY
import json
try:
    j = json.loads(open("$OUT/$name.json").read().strip().splitlines()[-1])
    print("%-28s %.3f ms/frame  rays/frame %d  checksum %r" % ("$name", j["ms_per_frame"], j["config"]["rays_per_frame"], j["config"]["frame_checksum"]))
except Exception as e:
    print("$name: FAILED", e); print(open("$OUT/$name.err").read()[-800:])
PY
}
for sz in 32x18 64x36 96x54 128x72 160x90; do
  w=${sz%x*}; h=${sz#*x}
  run ${sz}_default --width $w --height $h
  run ${sz}_perbounce --width $w --height $h --opt path_rays=0
  run ${sz}_coop_all --width $w --height $h --opt path_rays=0 --opt tail_lanes=0 --opt step_budget=1
  run ${sz}_coop_after4 --width $w --height $h --opt path_rays=0 --opt tail_lanes=0 --opt step_budget=4
done
