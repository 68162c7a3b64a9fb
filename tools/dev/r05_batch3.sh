#!/bin/bash
# round 5, batch 3: k_pool v2 (TRACE payload ring in LDS, one-line records): bit-identity, then a sweep of block shapes on the 1/8 shard
OUT=gpurun_out/${1:-r05e}
mkdir -p $OUT
timeout 900 python tools/dev/r05_pool_check.py 9 2>&1 | tail -12
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
P="--emulate-shard 8 --opt pool_rays=2147483647"
for w in 16 8 4; do
  for s in 1 2 3 4 6; do
    if [ $s -lt $w ]; then run sh8_pool_w${w}_s$s "$P --opt pool_waves=$w --opt pool_shaders=$s"; fi
  done
done
run sh8_pool_w8_s2_r32 "$P --opt pool_waves=8 --opt pool_shaders=2 --opt pool_refill=32"
run sh8_pool_w8_s2_r52 "$P --opt pool_waves=8 --opt pool_shaders=2 --opt pool_refill=52"
run sh8_pool_w8_s0 "$P --opt pool_waves=8 --opt pool_shaders=0"
run sh4_base "--emulate-shard 4"
run sh4_pool "--emulate-shard 4 --opt pool_rays=2147483647"
run sh2_pool "--emulate-shard 2 --opt pool_rays=2147483647"
