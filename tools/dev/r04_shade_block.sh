#!/bin/bash
OUT=gpurun_out/${1:-r04v}
mkdir -p $OUT
run() {
  timeout 300 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-extras $2 > $OUT/$1.json 2> $OUT/$1.err
  python - <<PY
import json
try:
    j = json.loads(open("$OUT/$1.json").read().strip().splitlines()[-1])
    st = j["stage_ms_per_frame"]; so = j["stage_ms_per_frame_solo"]
    print("$1: %.3f ms/frame  shading %.3f (solo %.3f)  checksum %r" % (j["ms_per_frame"], st.get("shading", 0), so.get("shading", 0), j["config"]["frame_checksum"]))
except Exception as e:
    print("$1: FAILED", e); print(open("$OUT/$1.err").read()[-800:])
PY
}
for bs in 256 512 1024; do
  run sh8_bs$bs "--emulate-shard 8 --opt path_rays=0 --opt shade_block=$bs"
  run full_bs$bs "--steps 8 --opt shade_block=$bs"
done
