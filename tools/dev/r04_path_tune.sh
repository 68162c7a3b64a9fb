#!/bin/bash
# round 4: k_path tuning on rank 0's 1/8 tile shard: library variant x waves per CU x refill threshold
OUT=gpurun_out/${1:-r04b}
mkdir -p $OUT
B="python bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-extras --emulate-shard 8 --opt path_rays=2147483647"
run() {  # name, lib, extra opts
  if [ -n "$2" ]; then export LPT_LIB_PATH=$PWD/loupiote_amd/libloupiote_hip_$2.so; else unset LPT_LIB_PATH; fi
  timeout 300 $B $3 > $OUT/$1.json 2> $OUT/$1.err
  python - <<PY
import json
try:
    j = json.loads(open("$OUT/$1.json").read().strip().splitlines()[-1])
    print("$1: %.3f ms/frame  path %.3f" % (j["ms_per_frame"], j["stage_ms_per_frame"].get("path", 0)))
except Exception as e:
    print("$1: FAILED", e); print(open("$OUT/$1.err").read()[-800:])
PY
}
run base_w12_r44 "" "--opt path_waves_per_cu=12"
run base_w12_r32 "" "--opt path_waves_per_cu=12 --opt path_refill=32"
run base_w12_r52 "" "--opt path_waves_per_cu=12 --opt path_refill=52"
run base_w12_r58 "" "--opt path_waves_per_cu=12 --opt path_refill=58"
run base_w8_r44 "" "--opt path_waves_per_cu=8"
run w4_w16_r44 w4 "--opt path_waves_per_cu=16"
run w4_w16_r32 w4 "--opt path_waves_per_cu=16 --opt path_refill=32"
run w4_w16_r52 w4 "--opt path_waves_per_cu=16 --opt path_refill=52"
run w4_w12_r44 w4 "--opt path_waves_per_cu=12"
