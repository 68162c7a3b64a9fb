#!/bin/bash
# where a wave per ray stops paying: whole frames, LPT_OPT_COOP_RAYS off / forced on.  tools/dev/r05_coop_threshold.sh <out>
OUT=gpurun_out/$1; mkdir -p $OUT
run() {
  local name=$1; shift
  timeout 400 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras "$@" > $OUT/$name.json 2> $OUT/$name.err
  python - <<PY
import json
try:
    j = json.loads(open("$OUT/$name.json").read().strip().splitlines()[-1])
    print("%-24s %.3f ms/frame  rays/frame %d  checksum %r" % ("$name", j["ms_per_frame"], j["config"]["rays_per_frame"], j["config"]["frame_checksum"]))
except Exception as e:
    print("$name: FAILED", e); print(open("$OUT/$name.err").read()[-800:])
PY
}
for sz in 32x18 64x36 80x45 96x54 112x63 128x72 144x81; do
  w=${sz%x*}; h=${sz#*x}
  for rep in 1 2; do
    run ${sz}_off_$rep --width $w --height $h --opt coop_rays=0
    run ${sz}_on_$rep --width $w --height $h --opt coop_rays=2000000000
  done
done
