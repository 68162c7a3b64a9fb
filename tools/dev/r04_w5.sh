#!/bin/bash
OUT=gpurun_out/${1:-r04m}
mkdir -p $OUT
run() {
  if [ -n "$2" ]; then export LPT_LIB_PATH=$PWD/loupiote_amd/libloupiote_hip_$2.so; else unset LPT_LIB_PATH; fi
  timeout 300 python bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-extras $3 > $OUT/$1.json 2> $OUT/$1.err
  python - <<PY
import json
try:
    j = json.loads(open("$OUT/$1.json").read().strip().splitlines()[-1])
    print("$1: %.3f ms/frame  path %.3f" % (j["ms_per_frame"], j["stage_ms_per_frame"].get("path", 0)))
except Exception as e:
    print("$1: FAILED", e); print(open("$OUT/$1.err").read()[-800:])
PY
}
for sh in 8 16 32; do
  run w4_sh$sh "" "--emulate-shard $sh --opt path_rays=2147483647"
  run w5_sh$sh w5 "--emulate-shard $sh --opt path_rays=2147483647 --opt path_waves_per_cu=20"
done
