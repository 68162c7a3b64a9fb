#!/bin/bash
# the bench frame cut into more pieces on more lanes: tools/dev/r06_lanes_ab.sh <out>
OUT=gpurun_out/$1; mkdir -p $OUT
for rep in 1 2; do
for cfg in "2 0" "3 2800000" "4 2100000" "3 2100000" "4 1400000" "2 2100000"; do
  set -- $cfg; L=$1; WR=$2
  args="--lanes $L"; [ $WR != 0 ] && args="$args --opt wavefront_rays=$WR"
  timeout 400 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras $args > $OUT/l${L}_${WR}_$rep.json 2> $OUT/l${L}_${WR}_$rep.err
  python - <<PY
import json
try:
    j = json.loads(open("$OUT/l${L}_${WR}_$rep.json").read().strip().splitlines()[-1])
    print("lanes $L wavefront_rays $WR rep $rep: %.3f ms/frame  checksum %r" % (j["ms_per_frame"], j["config"]["frame_checksum"]))
except Exception as e:
    print("lanes $L $WR: FAILED", e); print(open("$OUT/l${L}_${WR}_$rep.err").read()[-600:])
PY
done
done
