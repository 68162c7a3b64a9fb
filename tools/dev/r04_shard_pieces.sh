#!/bin/bash
OUT=gpurun_out/${1:-r04j}
mkdir -p $OUT
run() {
  timeout 300 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-extras $2 > $OUT/$1.json 2> $OUT/$1.err
  python - <<PY
import json
try:
    j = json.loads(open("$OUT/$1.json").read().strip().splitlines()[-1])
    print("$1: %.3f ms/frame  wavefronts/frame %.1f" % (j["ms_per_frame"], j["config"]["wavefronts_per_frame"]))
except Exception as e:
    print("$1: FAILED", e); print(open("$OUT/$1.err").read()[-800:])
PY
}
run sh2_1piece "--emulate-shard 2"
run sh2_2pieces "--emulate-shard 2 --opt wavefront_rays=2200000"
run sh2_3pieces "--emulate-shard 2 --opt wavefront_rays=1500000"
run sh4_1piece "--emulate-shard 4"
run sh4_2pieces "--emulate-shard 4 --opt wavefront_rays=1100000"
run sh8_1piece "--emulate-shard 8"
run sh8_2pieces "--emulate-shard 8 --opt wavefront_rays=600000 --opt path_rays=0"
run sh8_2pieces_b "--emulate-shard 8 --opt wavefront_rays=600000 --opt path_rays=0"
run sh8_1piece_b "--emulate-shard 8"
run full_2pieces "--steps 10"
run full_3pieces "--steps 10 --opt wavefront_rays=3000000"
run full_4pieces "--steps 10 --opt wavefront_rays=2100000"
