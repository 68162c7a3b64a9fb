#!/bin/bash
# the bench frame under renderer options: tools/dev/r06_opts_ab.sh <out> "<opt>=<v>[,<opt>=<v>]" ...   ("none" = the defaults)
OUT=gpurun_out/$1; shift
mkdir -p $OUT
for rep in 1 2; do
for o in "$@"; do
  args=""
  if [ "$o" != none ]; then for kv in ${o//,/ }; do args="$args --opt $kv"; done; fi
  tag=${o//[=,]/_}
  timeout 400 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras $args > $OUT/${tag}_$rep.json 2> $OUT/${tag}_$rep.err
  python - <<PY
import json
try:
    j = json.loads(open("$OUT/${tag}_$rep.json").read().strip().splitlines()[-1])
    print("$o $rep: %.3f ms/frame  solo %s  checksum %r" % (j["ms_per_frame"], {k: round(x, 3) for k, x in j["stage_ms_per_frame_solo"].items() if x}, j["config"]["frame_checksum"]))
except Exception as e:
    print("$o $rep: FAILED", e); print(open("$OUT/${tag}_$rep.err").read()[-800:])
PY
done
done
