#!/bin/bash
# k_trace_packet: node and triangle fetches as asynchronous vector loads (one per node, issued before the leaf tests; one per leaf slot) against waited scalar loads (variant `prepk`)
OUT=gpurun_out/${1:-r04pk}
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
  for v in new prepk; do
    if [ $v = prepk ]; then export LPT_LIB_PATH=$GRAFT_REPO_ROOT/loupiote_amd/libloupiote_hip_prepk.so; else unset LPT_LIB_PATH; fi
    run ${v}_full_$rep ""
    run ${v}_solo_$rep "--lanes 1 --max-fused 4"
  done
done
for v in new prepk; do
  if [ $v = prepk ]; then export LPT_LIB_PATH=$GRAFT_REPO_ROOT/loupiote_amd/libloupiote_hip_prepk.so; else unset LPT_LIB_PATH; fi
  run ${v}_sh8 "--emulate-shard 8"
  run ${v}_720 "--width 1280 --height 720"
  run ${v}_4k "--width 3840 --height 2160 --steps 4"
done
