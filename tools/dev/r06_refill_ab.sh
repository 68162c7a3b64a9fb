#!/bin/bash
# round 6: the refill threshold of the persistent traversal waves (lanes live at or below which a wave retires finished rays and takes new ones) with the 72-VGPR kernel: tools/dev/r06_refill_ab.sh <out>
OUT=gpurun_out/$1; mkdir -p $OUT
for rep in 1 2; do
for w in 36 44 50 56 60; do
  timeout 400 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras --opt refill=$w > $OUT/f${w}_$rep.json 2> $OUT/f${w}_$rep.err
  python - <<PY
import json
try:
    j = json.loads(open("$OUT/f${w}_$rep.json").read().strip().splitlines()[-1])
    print("refill $w rep $rep: %.3f ms/frame  solo intersection %.3f  lanes live/node/tri %.1f %.1f %.1f" % (j["ms_per_frame"], j["stage_ms_per_frame_solo"]["intersection"], j["roofline"]["wave"]["live_lanes_per_step"], j["roofline"]["wave"]["node_lanes_per_step"], j["roofline"]["wave"]["tri_lanes_per_step"]))
except Exception as e:
    print("f$w $rep: FAILED", e)
PY
done
done
