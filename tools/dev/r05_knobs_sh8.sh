#!/bin/bash
# knob sweep on the 1/8 shard with the tail in place: tools/dev/r05_knobs_sh8.sh <out>
OUT=gpurun_out/$1
mkdir -p $OUT
run() {
  local name=$1; shift
  timeout 400 python bench.py --steps 12 --warmup 3 --no-cpu-baseline --no-extras --emulate-shard 8 "$@" > $OUT/$name.json 2> $OUT/$name.err
  python - <<PY
import json
try:
    j = json.loads(open("$OUT/$name.json").read().strip().splitlines()[-1])
    st = j["stage_ms_per_frame"]
    print("%-22s %.3f ms/frame  trace %.3f shadow %.3f shade %.3f primary %.3f  checksum %r" % ("$name", j["ms_per_frame"], st["intersection"], st["shadow"], st["shading"], st["primary intersection"], j["config"]["frame_checksum"]))
except Exception as e:
    print("$name: FAILED", e); print(open("$OUT/$name.err").read()[-800:])
PY
}
for rep in 1 2; do
  run base_$rep
  for b in 2 3 6 8; do run shadeblocks${b}_$rep --opt shade_blocks_per_cu=$b; done
  for f in 36 40 48 52; do run refill${f}_$rep --opt refill=$f; done
  for w in 20 28 32; do run waves${w}_$rep --opt trace_waves_per_cu=$w; done
done
