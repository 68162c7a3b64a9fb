#!/bin/bash
# packet = 4 samples of a 4x4-pixel quarter (packet_quads=1) against one sample of an 8x8 patch (0): full frame, the 1/8 shard, 4K
OUT=gpurun_out/${1:-r04q}
mkdir -p $OUT
run() {
  timeout 300 python bench.py --steps ${STEPS:-12} --warmup 3 --no-cpu-baseline --no-extras $2 > $OUT/$1.json 2> $OUT/$1.err
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
  run q1_$rep "--opt packet_quads=1"
  run q0_$rep "--opt packet_quads=0"
done
run q1_4k "--width 3840 --height 2160 --opt packet_quads=1"
run q0_4k "--width 3840 --height 2160 --opt packet_quads=0"
run q1_720 "--width 1280 --height 720 --opt packet_quads=1"
run q0_720 "--width 1280 --height 720 --opt packet_quads=0"
