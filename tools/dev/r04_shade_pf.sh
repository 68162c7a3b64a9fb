#!/bin/bash
OUT=gpurun_out/${1:-r04u}
mkdir -p $OUT
run() {
  timeout 300 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-extras $2 > $OUT/$1.json 2> $OUT/$1.err
  python - <<PY
import json
try:
    j = json.loads(open("$OUT/$1.json").read().strip().splitlines()[-1])
    st = j["stage_ms_per_frame"]
    print("$1: %.3f ms/frame  intersection %.3f shading %.3f  checksum %r" % (j["ms_per_frame"], st.get("intersection", 0), st.get("shading", 0), j["config"]["frame_checksum"]))
except Exception as e:
    print("$1: FAILED", e); print(open("$OUT/$1.err").read()[-800:])
PY
}
for sh in 8 4 16; do
  run sh${sh}_pf0 "--emulate-shard $sh --opt path_rays=0"
  run sh${sh}_pf1 "--emulate-shard $sh --opt path_rays=0 --opt shade_prefetch_rays=100000000"
done
run f720_pf0 "--width 720 --height 405"
run f720_pf1 "--width 720 --height 405 --opt shade_prefetch_rays=100000000"
