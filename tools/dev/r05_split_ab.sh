#!/bin/bash
# one solo wavefront with its tails in place against two pieces on two lanes, around LPT_EXP_SPLIT_RAYS: tools/dev/r05_split_ab.sh <out>
OUT=gpurun_out/$1
mkdir -p $OUT
run() {
  local name=$1; shift
  timeout 400 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras "$@" > $OUT/$name.json 2> $OUT/$name.err
  python - <<PY
import json
try:
    j = json.loads(open("$OUT/$name.json").read().strip().splitlines()[-1])
    print("%-30s %.3f ms/frame  rays/frame %d  wavefronts %.1f  checksum %r" % ("$name", j["ms_per_frame"], j["config"]["rays_per_frame"], j["config"]["wavefronts_per_frame"], j["config"]["frame_checksum"]))
except Exception as e:
    print("$name: FAILED", e); print(open("$OUT/$name.err").read()[-800:])
PY
}
BIG=2000000000
for rep in 1 2; do
  for cfg in "sh2|--emulate-shard 2" "sh3|--emulate-shard 3" "f1280|--width 1280 --height 720" "f1600|--width 1600 --height 900"; do
    n=${cfg%%|*}; a=${cfg#*|}
    run ${n}_default_$rep $a
    run ${n}_pieces_tail_$rep $a --opt budget_split=1 --opt budget_rays=$BIG
    run ${n}_one_tail_$rep $a --opt split_rays=$BIG --opt budget_rays=$BIG --opt wavefront_rays=6000000
  done
done
