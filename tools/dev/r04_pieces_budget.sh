#!/bin/bash
OUT=gpurun_out/${1:-r04t}
mkdir -p $OUT
run() {
  timeout 300 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-extras $2 > $OUT/$1.json 2> $OUT/$1.err
  python - <<PY
import json
try:
    j = json.loads(open("$OUT/$1.json").read().strip().splitlines()[-1])
    print("$1: %.3f ms/frame  wavefronts %.1f" % (j["ms_per_frame"], j["config"]["wavefronts_per_frame"]))
except Exception as e:
    print("$1: FAILED", e); print(open("$OUT/$1.err").read()[-800:])
PY
}
run sh8_solo_b48 "--emulate-shard 8"
run sh8_2p_b0 "--emulate-shard 8 --opt wavefront_rays=600000 --opt path_rays=0"
run sh8_2p_b48 "--emulate-shard 8 --opt wavefront_rays=600000 --opt path_rays=0 --opt budget_rays=100000000"
run sh4_solo_b48 "--emulate-shard 4"
run sh4_2p_b48 "--emulate-shard 4 --opt wavefront_rays=1100000 --opt budget_rays=100000000"
run sh2_2p_b0 "--emulate-shard 2"
run sh2_2p_b48 "--emulate-shard 2 --opt budget_rays=100000000"
run sh2_2p_b64 "--emulate-shard 2 --opt budget_rays=100000000 --opt step_budget=64"
run full_2p_b64 "--opt budget_rays=100000000 --opt step_budget=64"
run full_2p_b96 "--opt budget_rays=100000000 --opt step_budget=96"
