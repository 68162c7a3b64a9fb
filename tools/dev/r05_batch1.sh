#!/bin/bash
# round 5, first GPU batch (no code changes): the 1/8 shard as two pieces that TIME-SHARE the chip (full-size persistent grids, so that one piece's
# blocks fill the slots the other's tail frees), k_shade's grid at shard size, one wavefront lane against two at whole-frame size.
OUT=gpurun_out/${1:-r05a}
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
run sh8_path "--emulate-shard 8 --opt path_rays=2147483647"
run sh8_2p_full_b0 "--emulate-shard 8 --opt wavefront_rays=600000 --opt path_rays=0 --opt trace_waves_per_cu=24"
run sh8_2p_full_b48 "--emulate-shard 8 --opt wavefront_rays=600000 --opt path_rays=0 --opt trace_waves_per_cu=24 --opt budget_rays=100000000"
run sh8_2p_half_b48 "--emulate-shard 8 --opt wavefront_rays=600000 --opt path_rays=0 --opt budget_rays=100000000"
run sh8_4p_full_b48 "--emulate-shard 8 --opt wavefront_rays=300000 --opt path_rays=0 --opt trace_waves_per_cu=24 --opt budget_rays=100000000 --lanes 4"
run sh8_3p_full_b48 "--emulate-shard 8 --opt wavefront_rays=400000 --opt path_rays=0 --opt trace_waves_per_cu=24 --opt budget_rays=100000000 --lanes 3"
for sb in 2 3 6 8 16; do
  run sh8_sb$sb "--emulate-shard 8 --opt shade_blocks_per_cu=$sb"
done
run sh8_base2 "--emulate-shard 8"
run full_lanes2 "--steps 10"
run full_lanes1 "--steps 10 --lanes 1"
run full_lanes2b "--steps 10"
run full_lanes1b "--steps 10 --lanes 1"
