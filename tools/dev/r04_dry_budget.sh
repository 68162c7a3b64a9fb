#!/bin/bash
# the step budget drops to n once a wave has nothing left to refill from (variants dry16 / dry24 / dry32) against the constant budget (48): 1/8, 1/4 shard and a 720p frame
OUT=gpurun_out/${1:-r04dry}
mkdir -p $OUT
run() {
  timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras $2 > $OUT/$1.json 2> $OUT/$1.err
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
  for v in base dry16 dry24 dry32; do
    if [ $v = base ]; then unset LPT_LIB_PATH; else export LPT_LIB_PATH=$GRAFT_REPO_ROOT/loupiote_amd/libloupiote_hip_$v.so; fi
    run ${v}_sh8_$rep "--emulate-shard 8"
  done
done
for v in base dry16 dry24 dry32; do
  if [ $v = base ]; then unset LPT_LIB_PATH; else export LPT_LIB_PATH=$GRAFT_REPO_ROOT/loupiote_amd/libloupiote_hip_$v.so; fi
  run ${v}_sh4 "--emulate-shard 4"
  run ${v}_sh16 "--emulate-shard 16"
done
