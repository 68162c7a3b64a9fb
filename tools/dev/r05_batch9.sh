#!/bin/bash
# round 5: a last sweep of the whole-frame knobs on the final kernels (k_shade's grid now that its per-wave atomic is gone, refill, trace waves)
OUT=gpurun_out/r05v
mkdir -p $OUT
run() {
  timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras $2 > $OUT/$1.json 2> $OUT/$1.err
  python - <<PY
import json
try:
    j = json.loads(open("$OUT/$1.json").read().strip().splitlines()[-1])
    so = j["stage_ms_per_frame_solo"]
    print("$1: %.3f ms/frame  solo %s" % (j["ms_per_frame"], {k: round(x, 3) for k, x in so.items() if x}))
except Exception as e:
    print("$1: FAILED", e); print(open("$OUT/$1.err").read()[-800:])
PY
}
run base ""
for sb in 3 6 8 12; do run sb$sb "--opt shade_blocks_per_cu=$sb"; done
for rf in 40 48 52; do run refill$rf "--opt refill=$rf"; done
for tw in 20 28 32; do run tw$tw "--opt trace_waves_per_cu=$tw"; done
run base2 ""
