#!/bin/bash
OUT=gpurun_out/${1:-r04h}
mkdir -p $OUT
for c in 60 250 1000 4000; do
  timeout 400 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras --opt occ_cell_milli=$c > $OUT/occ_$c.json 2> $OUT/occ_$c.err
  python - <<PY
import json
try:
    j = json.loads(open("$OUT/occ_$c.json").read().strip().splitlines()[-1])
    print("cell $c:", {k: (round(v, 4) if isinstance(v, float) else v) for k, v in j["roofline"]["occluder_cache_probe"].items() if k != "what"}, "shadow nodes/ray", round(j["roofline"]["shadow_nodes_per_ray"], 2))
except Exception as e:
    print("cell $c: FAILED", e); print(open("$OUT/occ_$c.err").read()[-800:])
PY
done
