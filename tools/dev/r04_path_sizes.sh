#!/bin/bash
# round 4: where does the path kernel win?  whole (unsharded) frames of several sizes, span form, both pipelines; w4 = k_path forced to 4 waves/SIMD
OUT=gpurun_out/${1:-r04c}
mkdir -p $OUT
run() {  # name, lib, args
  if [ -n "$2" ]; then export LPT_LIB_PATH=$PWD/loupiote_amd/libloupiote_hip_$2.so; else unset LPT_LIB_PATH; fi
  timeout 300 python bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-extras $3 > $OUT/$1.json 2> $OUT/$1.err
  python - <<PY
import json
try:
    j = json.loads(open("$OUT/$1.json").read().strip().splitlines()[-1])
    print("$1: %.3f ms/frame  stages %s" % (j["ms_per_frame"], {k: round(x, 3) for k, x in j["stage_ms_per_frame"].items() if x}))
except Exception as e:
    print("$1: FAILED", e); print(open("$OUT/$1.err").read()[-800:])
PY
}
for sz in "240 135" "480 270" "960 540"; do
  set -- $sz
  run pb_$1 "" "--width $1 --height $2 --opt path_rays=0"
  run path12_$1 "" "--width $1 --height $2 --opt path_rays=2147483647"
  run path16_$1 w4 "--width $1 --height $2 --opt path_rays=2147483647 --opt path_waves_per_cu=16 --opt path_refill=32"
done
for s in 16 32; do
  run pb_sh$s "" "--emulate-shard $s --opt path_rays=0"
  run path16_sh$s w4 "--emulate-shard $s --opt path_rays=2147483647 --opt path_waves_per_cu=16 --opt path_refill=32"
done
run path16_sh8_r24 w4 "--emulate-shard 8 --opt path_rays=2147483647 --opt path_waves_per_cu=16 --opt path_refill=24"
run path16_sh8_r16 w4 "--emulate-shard 8 --opt path_rays=2147483647 --opt path_waves_per_cu=16 --opt path_refill=16"
