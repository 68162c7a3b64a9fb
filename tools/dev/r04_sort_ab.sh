#!/bin/bash
OUT=gpurun_out/${1:-r04g}
mkdir -p $OUT
for rep in 1 2; do
for v in 0 4; do
  timeout 400 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras --sort $v > $OUT/sort${v}_$rep.json 2> $OUT/sort${v}_$rep.err
  python - <<PY
import json
try:
    j = json.loads(open("$OUT/sort${v}_$rep.json").read().strip().splitlines()[-1])
    print("sort $v $rep: %.3f ms/frame  solo %s  checksum %r" % (j["ms_per_frame"], {k: round(x, 3) for k, x in j["stage_ms_per_frame_solo"].items() if x}, j["config"]["frame_checksum"]))
except Exception as e:
    print("sort $v $rep: FAILED", e); print(open("$OUT/sort${v}_$rep.err").read()[-800:])
PY
done
done
