#!/bin/bash
# the tail finished in place (LPT_OPT_TAIL_LANES) on the 1/8 shard, span form: tools/dev/r05_tail_ab.sh <out>
OUT=gpurun_out/$1
mkdir -p $OUT
run() {  # name, bench args...
  local name=$1; shift
  timeout 400 python bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-extras "$@" > $OUT/$name.json 2> $OUT/$name.err
  python - <<PY
import json
try:
    j = json.loads(open("$OUT/$name.json").read().strip().splitlines()[-1])
    print("$name: %.3f ms/frame  stages %s  checksum %r" % (j["ms_per_frame"], {k: round(x, 3) for k, x in j["stage_ms_per_frame"].items() if x}, j["config"]["frame_checksum"]))
except Exception as e:
    print("$name: FAILED", e); print(open("$OUT/$name.err").read()[-800:])
PY
}
for rep in 1 2; do
  run sh8_base_$rep --emulate-shard 8
  run sh8_nobudget_$rep --emulate-shard 8 --opt step_budget=0
  for t in 1 2 3 4 6 8; do run sh8_tail${t}_$rep --emulate-shard 8 --opt tail_lanes=$t; done
done
run sh4_base --emulate-shard 4
run sh4_tail2 --emulate-shard 4 --opt tail_lanes=2
run sh4_tail4 --emulate-shard 4 --opt tail_lanes=4
run full_base
run full_tail4 --opt tail_lanes=4 --opt budget_rays=2000000000 --opt budget_split=1
