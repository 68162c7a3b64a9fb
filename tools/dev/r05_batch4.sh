#!/bin/bash
# round 5, batch 4: the 1/8 shard as pieces on lanes AGAIN, now that k_shade's per-launch floor (one atomic per wave on one word) is gone
OUT=gpurun_out/${1:-r05f}
mkdir -p $OUT
run() {
  timeout 300 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-extras $2 > $OUT/$1.json 2> $OUT/$1.err
  python - <<PY
import json
try:
    j = json.loads(open("$OUT/$1.json").read().strip().splitlines()[-1])
    st = j["stage_ms_per_frame"]
    print("$1: %.3f ms/frame  wf %.1f  stages %s  checksum %r" % (j["ms_per_frame"], j["config"]["wavefronts_per_frame"], {k: round(x, 3) for k, x in st.items() if x}, j["config"]["frame_checksum"]))
except Exception as e:
    print("$1: FAILED", e); print(open("$OUT/$1.err").read()[-800:])
PY
}
run sh8_base "--emulate-shard 8"
run sh8_b0 "--emulate-shard 8 --opt step_budget=0"
run sh8_b32 "--emulate-shard 8 --opt step_budget=32"
run sh8_b64 "--emulate-shard 8 --opt step_budget=64"
run sh8_2p "--emulate-shard 8 --opt wavefront_rays=600000 --opt path_rays=0 --opt budget_split=1"
run sh8_2p_b0 "--emulate-shard 8 --opt wavefront_rays=600000 --opt path_rays=0"
run sh8_2p_full "--emulate-shard 8 --opt wavefront_rays=600000 --opt path_rays=0 --opt budget_split=1 --opt trace_waves_per_cu=24"
run sh8_3p "--emulate-shard 8 --opt wavefront_rays=400000 --opt path_rays=0 --opt budget_split=1 --lanes 3"
run sh8_4p "--emulate-shard 8 --opt wavefront_rays=300000 --opt path_rays=0 --opt budget_split=1 --lanes 4"
run sh8_w16 "--emulate-shard 8 --opt trace_waves_per_cu=16"
run sh8_w32 "--emulate-shard 8 --opt trace_waves_per_cu=32"
run sh8_path "--emulate-shard 8 --opt path_rays=2147483647"
run sh4_base "--emulate-shard 4"
run sh4_2p "--emulate-shard 4 --opt wavefront_rays=1100000 --opt budget_split=1"
run sh2_base "--emulate-shard 2"
run full "--steps 10"
