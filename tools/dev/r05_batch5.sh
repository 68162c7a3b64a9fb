#!/bin/bash
# round 5, batch 5: the fixed traversal counts of config 4 (two-round-trip step), then the round's profiles: whole frame (tools/profile.sh), the 1/8 shard
# per-bounce and pool (tools/profile_shard.sh)
OUT=gpurun_out/r05j
mkdir -p $OUT
timeout 300 python bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-extras --opt pipe_rays=0 > $OUT/pipe0.json 2> $OUT/pipe0.err
python - <<'PY'
import json
j = json.loads(open("gpurun_out/r05j/pipe0.json").read().strip().splitlines()[-1])
r = j["roofline"]
print("pipe_rays=0 counts:", {k: r[k] for k in ("nodes_per_ray", "tris_per_ray", "shadow_nodes_per_ray", "shadow_tris_per_ray")}, "ms/frame", j["ms_per_frame"])
PY
bash tools/profile.sh r05a > $OUT/profile_r05a.log 2>&1
tail -5 $OUT/profile_r05a.log
bash tools/profile_shard.sh r05_sh8_pb > $OUT/profile_sh8_pb.log 2>&1
bash tools/profile_shard.sh r05_sh8_pool --opt pool_rays=2147483647 > $OUT/profile_sh8_pool.log 2>&1
tail -40 $OUT/profile_sh8_pool.log | cut -c1-220
