#!/bin/bash
# finer A/B of LPT_OPT_TAIL_LANES: 1/8 and 1/4 shards, and whole small frames (path_rays 0: the per-bounce launches): tools/dev/r05_tail_ab.sh <out>
OUT=gpurun_out/$1
mkdir -p $OUT
run() {
  local name=$1; shift
  timeout 400 python bench.py --steps 12 --warmup 3 --no-cpu-baseline --no-extras "$@" > $OUT/$name.json 2> $OUT/$name.err
  python - <<PY
import json
try:
    j = json.loads(open("$OUT/$name.json").read().strip().splitlines()[-1])
    print("$name: %.3f ms/frame  trace %.3f shadow %.3f shade %.3f  checksum %r" % (j["ms_per_frame"], j["stage_ms_per_frame"]["intersection"], j["stage_ms_per_frame"]["shadow"], j["stage_ms_per_frame"]["shading"], j["config"]["frame_checksum"]))
except Exception as e:
    print("$name: FAILED", e); print(open("$OUT/$name.err").read()[-800:])
PY
}
for rep in 1 2 3; do
  for t in 0 2 3 4 5; do run sh8_t${t}_$rep --emulate-shard 8 --opt tail_lanes=$t; done
done
for t in 0 3 4; do run sh4_t${t} --emulate-shard 4 --opt tail_lanes=$t; done
for t in 0 3 4; do run sh16_t${t} --emulate-shard 16 --opt tail_lanes=$t; done
for t in 0 3 4; do run f480_t${t} --width 480 --height 270 --opt path_rays=0 --opt tail_lanes=$t; done
for t in 0 3 4; do run f960_t${t} --width 960 --height 540 --opt path_rays=0 --opt tail_lanes=$t; done
