#!/bin/bash
# where packets of 4 samples x 4x4 pixels stop paying: dense frames (multiples of the 32x8 tile) of falling resolution, packets forced / per ray
OUT=gpurun_out/${1:-r04q2}
mkdir -p $OUT
run() {
  timeout 300 python bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-extras $2 > $OUT/$1.json 2> $OUT/$1.err
  python - <<PY
import json
try:
    j = json.loads(open("$OUT/$1.json").read().strip().splitlines()[-1])
    print("$1: %.3f ms/frame  stages %s" % (j["ms_per_frame"], {k: round(x, 3) for k, x in j["stage_ms_per_frame"].items() if x}))
except Exception as e:
    print("$1: FAILED", e); print(open("$OUT/$1.err").read()[-800:])
PY
}
for sz in "256 144" "384 216" "512 288" "640 360" "896 504"; do
  set -- $sz
  run quads_$1 "--width $1 --height $2 --opt packet_primary=1"
  run patch_$1 "--width $1 --height $2 --opt packet_primary=1 --opt packet_quads=0"
  run perray_$1 "--width $1 --height $2 --opt packet_primary=0"
done
