#!/bin/bash
# quick SQ-counter pass for the bench workload (run on the GPU box): tools/pmc_quick.sh <tag>
set -u
TAG=${1:-q}
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY --output-format csv -d $OUT/pmc1 -o p -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline > $OUT/pmc1.log 2>&1
python3 tools/pmc_summary.py $OUT/pmc1/p_counter_collection.csv > $OUT/summary.txt
grep -A9 -E "== k_trace<false>|== k_shade" $OUT/summary.txt
