#!/bin/bash
OUT=gpurun_out/${1:-r04n}
mkdir -p $OUT
run() {
  timeout 300 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-extras $2 > $OUT/$1.json 2> $OUT/$1.err
  python - <<PY
import json
try:
    j = json.loads(open("$OUT/$1.json").read().strip().splitlines()[-1])
    st = j["stage_ms_per_frame"]
    print("$1: %.3f ms/frame  intersection %.3f shadow %.3f shading %.3f" % (j["ms_per_frame"], st.get("intersection", 0), st.get("shadow", 0), st.get("shading", 0)))
except Exception as e:
    print("$1: FAILED", e); print(open("$OUT/$1.err").read()[-800:])
PY
}
for b in 0 24 32 48 64 96; do run sh8_b$b "--emulate-shard 8 --opt path_rays=0 --opt step_budget=$b"; done
for b in 0 32 48; do run sh4_b$b "--emulate-shard 4 --opt step_budget=$b"; done
for b in 0 32 48; do run sh16_b$b "--emulate-shard 16 --opt path_rays=0 --opt step_budget=$b"; done
for b in 0 32 48 64; do run full_b$b "--steps 8 --opt step_budget=$b --opt budget_rays=100000000"; done
