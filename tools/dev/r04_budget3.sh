#!/bin/bash
OUT=gpurun_out/${1:-r04p}
mkdir -p $OUT
run() {
  timeout 300 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-extras $2 > $OUT/$1.json 2> $OUT/$1.err
  python - <<PY
import json
try:
    j = json.loads(open("$OUT/$1.json").read().strip().splitlines()[-1])
    st = j["stage_ms_per_frame"]
    print("$1: %.3f ms/frame  %s" % (j["ms_per_frame"], {k: round(v, 3) for k, v in st.items() if v}))
except Exception as e:
    print("$1: FAILED", e); print(open("$OUT/$1.err").read()[-800:])
PY
}
run sh32_path "--emulate-shard 32"
run sh32_pb48 "--emulate-shard 32 --opt path_rays=0 --opt step_budget=48"
run sh32_pb40 "--emulate-shard 32 --opt path_rays=0 --opt step_budget=40"
run sh64_path "--emulate-shard 64"
run sh64_pb48 "--emulate-shard 64 --opt path_rays=0 --opt step_budget=48"
run f480_path "--width 480 --height 270"
run f480_pb48 "--width 480 --height 270 --opt path_rays=0 --opt step_budget=48"
run f240_path "--width 240 --height 135"
run f240_pb48 "--width 240 --height 135 --opt path_rays=0 --opt step_budget=48"
run f720_pb0 "--width 720 --height 405"
run f720_pb48 "--width 720 --height 405 --opt step_budget=48"
