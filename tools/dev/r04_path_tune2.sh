#!/bin/bash
OUT=gpurun_out/${1:-r04e}
mkdir -p $OUT
B="python bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-extras --emulate-shard 8 --opt path_rays=2147483647"
run() {
  timeout 300 $B $2 > $OUT/$1.json 2> $OUT/$1.err
  python - <<PY
import json
try:
    j = json.loads(open("$OUT/$1.json").read().strip().splitlines()[-1])
    print("$1: %.3f ms/frame  path %.3f" % (j["ms_per_frame"], j["stage_ms_per_frame"].get("path", 0)))
except Exception as e:
    print("$1: FAILED", e); print(open("$OUT/$1.err").read()[-800:])
PY
}
for r in 0 4 8 12 16 24; do run r$r "--opt path_refill=$r"; done
