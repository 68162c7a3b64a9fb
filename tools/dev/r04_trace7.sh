#!/bin/bash
# k_trace forced to 7 waves per SIMD (72 VGPRs + 28 B scratch) against 6 (78 VGPRs): trace_waves_per_cu 24 (default cap) and 28
OUT=gpurun_out/${1:-r04tr7}
mkdir -p $OUT
run() {
  timeout 300 python bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-extras $2 > $OUT/$1.json 2> $OUT/$1.err
  python - <<PY
import json
try:
    j = json.loads(open("$OUT/$1.json").read().strip().splitlines()[-1])
    print("$1: %.3f ms/frame  stages %s" % (j["ms_per_frame"], {k: round(x, 3) for k, x in j["stage_ms_per_frame"].items() if x}))
except Exception as e:
    print("$1: FAILED", e); print(open("$OUT/$1.err").read()[-800:])
PY
}
for v in base tr7; do
  if [ $v = base ]; then unset LPT_LIB_PATH; else export LPT_LIB_PATH=$GRAFT_REPO_ROOT/loupiote_amd/libloupiote_hip_$v.so; fi
  run ${v}_solo "--lanes 1 --max-fused 4"
  run ${v}_solo28 "--lanes 1 --max-fused 4 --opt trace_waves_per_cu=28"
  run ${v}_full ""
  run ${v}_full28 "--opt trace_waves_per_cu=28"
done
