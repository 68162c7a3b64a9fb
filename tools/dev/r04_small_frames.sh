#!/bin/bash
OUT=gpurun_out/${1:-r04k}
mkdir -p $OUT
run() {
  timeout 300 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-extras $2 > $OUT/$1.json 2> $OUT/$1.err
  python - <<PY
import json
try:
    j = json.loads(open("$OUT/$1.json").read().strip().splitlines()[-1])
    print("$1: %.3f ms/frame  stages %s" % (j["ms_per_frame"], {k: round(x, 3) for k, x in j["stage_ms_per_frame"].items() if x}))
except Exception as e:
    print("$1: FAILED", e); print(open("$OUT/$1.err").read()[-800:])
PY
}
for sz in "240 135" "480 270" "720 405" "960 540" "1280 720"; do
  set -- $sz
  run auto_$1 "--width $1 --height $2"
  run packet_$1 "--width $1 --height $2 --opt packet_primary=1"
  run perray_$1 "--width $1 --height $2 --opt packet_primary=0"
done
