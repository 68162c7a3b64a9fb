#!/bin/bash
# round 4: the path kernel against the per-bounce launches on rank 0's 1/N tile shard (span form, one GPU)
OUT=gpurun_out/${1:-r04a}
mkdir -p $OUT
B="python bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-extras"
for n in 8 4 2; do
  for v in "path_rays=0" "path_rays=2147483647"; do
    timeout 300 $B --emulate-shard $n --opt $v > $OUT/sh${n}_${v}.json 2> $OUT/sh${n}_${v}.err
    python - <<PY
import json
try:
    j = json.loads(open("$OUT/sh${n}_${v}.json").read().strip().splitlines()[-1])
    print("shard $n $v: %.3f ms/frame  stages %s" % (j["ms_per_frame"], {k: round(x, 3) for k, x in j["stage_ms_per_frame"].items() if x}))
except Exception as e:
    print("shard $n $v: FAILED", e); print(open("$OUT/sh${n}_${v}.err").read()[-1500:])
PY
  done
done
