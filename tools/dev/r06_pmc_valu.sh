#!/bin/bash
# VALU / SALU wave-instructions per un-overlapped k_trace launch for library variants: tools/dev/r06_pmc_valu.sh <out> <variant names...>   ("base" = the default library)
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/$1; shift; mkdir -p $OUT
for v in "$@"; do
  if [ $v = base ]; then unset LPT_LIB_PATH; else export LPT_LIB_PATH=$PWD/loupiote_amd/libloupiote_hip_$v.so; fi
  timeout 400 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VMEM_RD SQ_INSTS_LDS --output-format csv -d $OUT/$v -o p -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-extras --frames-per-step 3 --max-fused 4 --lanes 1 > $OUT/$v.log 2>&1
  python3 - <<PY
import csv, collections
f = [l for l in open("$OUT/$v.log")]
import glob
rows = list(csv.DictReader(open(glob.glob("$OUT/$v/*counter_collection.csv")[0])))
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for r in rows:
    k = r["Kernel_Name"]
    if "k_trace<" in k or "k_shade<" in k:
        k = k.split("(")[0].replace("void lptd::", "")
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[(k, r["Counter_Name"])] += 1
for k in acc:
    print("$v", k, {c: round(v / n[(k, c)] / 1e6, 2) for c, v in acc[k].items()}, "launches", n[(k, "SQ_INSTS_VALU")])
PY
done
