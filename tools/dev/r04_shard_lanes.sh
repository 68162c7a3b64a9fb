#!/bin/bash
# a 1/8 shard cut into pieces on several lanes (streams): do the drains of one piece hide under the other's work?
OUT=gpurun_out/${1:-r04i}
mkdir -p $OUT
run() {
  timeout 300 python bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-extras --emulate-shard 8 $2 > $OUT/$1.json 2> $OUT/$1.err
  python - <<PY
import json
try:
    j = json.loads(open("$OUT/$1.json").read().strip().splitlines()[-1])
    print("$1: %.3f ms/frame  wavefronts/frame %.1f  stages %s" % (j["ms_per_frame"], j["config"]["wavefronts_per_frame"], {k: round(x, 3) for k, x in j["stage_ms_per_frame"].items() if x}))
except Exception as e:
    print("$1: FAILED", e); print(open("$OUT/$1.err").read()[-800:])
PY
}
run pb_1piece "--opt path_rays=0"
for l in 2 3 4; do
  run pb_600k_l$l "--opt path_rays=0 --opt wavefront_rays=600000 --lanes $l"
  run pb_300k_l$l "--opt path_rays=0 --opt wavefront_rays=300000 --lanes $l"
  run path_600k_l$l "--opt wavefront_rays=600000 --lanes $l"
  run path_300k_l$l "--opt wavefront_rays=300000 --lanes $l"
done
