#!/bin/bash
# A/B on one box: the queues behind bounce 0 in eight segments with a counter each (the tree) against one counter per queue (variant library `preseg`, built from the commit before)
OUT=gpurun_out/${1:-r04seg_ab}
mkdir -p $OUT
run() {
  timeout 300 python bench.py --steps ${STEPS:-10} --warmup 3 --no-cpu-baseline --no-extras $2 > $OUT/$1.json 2> $OUT/$1.err
  python - <<PY
import json
try:
    j = json.loads(open("$OUT/$1.json").read().strip().splitlines()[-1])
    print("$1: %.3f ms/frame  stages %s" % (j["ms_per_frame"], {k: round(x, 3) for k, x in j["stage_ms_per_frame"].items() if x}))
except Exception as e:
    print("$1: FAILED", e); print(open("$OUT/$1.err").read()[-800:])
PY
}
for rep in 1 2; do
  for v in seg preseg; do
    if [ $v = preseg ]; then export LPT_LIB_PATH=$GRAFT_REPO_ROOT/loupiote_amd/libloupiote_hip_preseg.so; else unset LPT_LIB_PATH; fi
    run ${v}_full_$rep ""
    run ${v}_solo_$rep "--lanes 1 --max-fused 4"
    run ${v}_sh8_$rep "--emulate-shard 8"
  done
done
