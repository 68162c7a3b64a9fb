#!/bin/bash
# round 6: persistent k_trace waves per CU with the 72-VGPR kernel (a 124-VGPR k_shade wave fits beside FIVE of its waves on a SIMD, not beside six): tools/dev/r06_waves_ab.sh <out>
OUT=gpurun_out/$1; mkdir -p $OUT
for rep in 1 2; do
for w in 0 16 20 24 28; do
  timeout 400 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras --opt trace_waves_per_cu=$w > $OUT/w${w}_$rep.json 2> $OUT/w${w}_$rep.err
  python - <<PY
import json
try:
    j = json.loads(open("$OUT/w${w}_$rep.json").read().strip().splitlines()[-1])
    print("waves/CU $w rep $rep: %.3f ms/frame  stages %s" % (j["ms_per_frame"], {k: round(x, 2) for k, x in j["stage_ms_per_frame"].items() if x}))
except Exception as e:
    print("w$w $rep: FAILED", e)
PY
done
done
