#!/bin/bash
# A/B of library variants on TINY frames (every ray traced by a whole wave: k_trace_coop): tools/dev/r06_ab_tiny.sh <out> <variant names...>   ("base" = the default library)
OUT=gpurun_out/$1; shift
mkdir -p $OUT
for size in "64 36" "128 72" "256 144"; do
set -- $size "${@:1}"
W=$1; H=$2; shift 2
for rep in 1 2; do
for v in "$@"; do
  if [ $v = base ]; then unset LPT_LIB_PATH; else export LPT_LIB_PATH=$PWD/loupiote_amd/libloupiote_hip_$v.so; fi
  timeout 300 python bench.py --width $W --height $H --steps 30 --warmup 5 --no-cpu-baseline --no-extras > $OUT/${v}_${W}_$rep.json 2> $OUT/${v}_${W}_$rep.err
  python - <<PY
import json
try:
    j = json.loads(open("$OUT/${v}_${W}_$rep.json").read().strip().splitlines()[-1])
    print("${W}x${H} $v $rep: %.4f ms/frame  checksum %r" % (j["ms_per_frame"], j["config"]["frame_checksum"]))
except Exception as e:
    print("$v $rep: FAILED", e); print(open("$OUT/${v}_${W}_$rep.err").read()[-800:])
PY
done
done
done
