#!/bin/bash
# the bench frame with k_shade's outgoing queues ordered by direction octant within a block (lpt_renderer_set_sort_queues 0 / 1 / 2 / 3): tools/dev/r06_sort_ab.sh <out>
OUT=gpurun_out/$1; mkdir -p $OUT
for rep in 1 2; do
for sv in 0 1 2 3; do
  timeout 400 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras --sort $sv > $OUT/s${sv}_$rep.json 2> $OUT/s${sv}_$rep.err
  python - <<PY
import json
try:
    j = json.loads(open("$OUT/s${sv}_$rep.json").read().strip().splitlines()[-1])
    print("sort $sv rep $rep: %.3f ms/frame  solo %s  checksum %r" % (j["ms_per_frame"], {k: round(x, 3) for k, x in j["stage_ms_per_frame_solo"].items() if x}, j["config"]["frame_checksum"]))
except Exception as e:
    print("sort $sv: FAILED", e); print(open("$OUT/s${sv}_$rep.err").read()[-600:])
PY
done
done
