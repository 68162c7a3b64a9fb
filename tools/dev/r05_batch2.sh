#!/bin/bash
# round 5, batch 2: first runs of k_pool (bit-identity against the per-bounce launches, small frames first), the k_shade surface-count fix at shard and frame size
OUT=gpurun_out/${1:-r05b}
mkdir -p $OUT
timeout 600 python tools/dev/r05_pool_check.py 5 2>&1 | tail -20
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
run sh8_sb2 "--emulate-shard 8 --opt shade_blocks_per_cu=2"
run sh8_sb8 "--emulate-shard 8 --opt shade_blocks_per_cu=8"
run sh8_pool "--emulate-shard 8 --opt pool_rays=2147483647"
run sh8_pool_s1 "--emulate-shard 8 --opt pool_rays=2147483647 --opt pool_shaders=1"
run sh8_pool_s3 "--emulate-shard 8 --opt pool_rays=2147483647 --opt pool_shaders=3"
run sh8_pool_w8 "--emulate-shard 8 --opt pool_rays=2147483647 --opt pool_waves=8 --opt pool_shaders=1"
run sh4_base "--emulate-shard 4"
run sh4_pool "--emulate-shard 4 --opt pool_rays=2147483647"
run full "--steps 10"
run full_sb8 "--steps 10 --opt shade_blocks_per_cu=8"
