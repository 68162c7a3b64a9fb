#!/bin/bash
# A/B of library variants on the 1/8 shard (span form): tools/dev/r05_ab_shard.sh <out> <variant names...>   ("base" = the default library)
OUT=gpurun_out/$1; shift
mkdir -p $OUT
for rep in 1 2; do
for v in "$@"; do
  if [ $v = base ]; then unset LPT_LIB_PATH; else export LPT_LIB_PATH=$PWD/loupiote_amd/libloupiote_hip_$v.so; fi
  timeout 400 python bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-extras --emulate-shard 8 > $OUT/${v}_$rep.json 2> $OUT/${v}_$rep.err
  python - <<PY
import json
try:
    j = json.loads(open("$OUT/${v}_$rep.json").read().strip().splitlines()[-1])
    print("$v $rep: %.3f ms/frame  stages %s  checksum %r" % (j["ms_per_frame"], {k: round(x, 3) for k, x in j["stage_ms_per_frame"].items() if x}, j["config"]["frame_checksum"]))
except Exception as e:
    print("$v $rep: FAILED", e); print(open("$OUT/${v}_$rep.err").read()[-800:])
PY
done
done
