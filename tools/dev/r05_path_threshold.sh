#!/bin/bash
# where does k_path stop paying against the per-bounce launches with their tails in place?  whole frames of several sizes, both pipelines: tools/dev/r05_path_threshold.sh <out>
OUT=gpurun_out/$1
mkdir -p $OUT
run() {
  local name=$1; shift
  timeout 400 python bench.py --steps 12 --warmup 3 --no-cpu-baseline --no-extras "$@" > $OUT/$name.json 2> $OUT/$name.err
  python - <<PY
import json
try:
    j = json.loads(open("$OUT/$name.json").read().strip().splitlines()[-1])
    print("%-26s %.3f ms/frame  rays/frame %d  wavefronts %.1f  checksum %r" % ("$name", j["ms_per_frame"], j["config"]["rays_per_frame"], j["config"]["wavefronts_per_frame"], j["config"]["frame_checksum"]))
except Exception as e:
    print("$name: FAILED", e); print(open("$OUT/$name.err").read()[-800:])
PY
}
for sz in ${SIZES:-192x108 256x144 320x180 384x216 448x252 512x288 640x360}; do
  w=${sz%x*}; h=${sz#*x}
  for rep in 1 2; do
    run path_${sz}_$rep --width $w --height $h --opt path_rays=2147483647
    run perbounce_${sz}_$rep --width $w --height $h --opt path_rays=0
  done
done
