#!/bin/bash
# Run ON THE GPU BOX (via gpurun): counters of the kernels of rank 0's 1/8 tile shard, path kernel and per-bounce launches.
# usage: tools/profile_shard.sh <tag> [extra bench args]  -> gpurun_out/prof_<tag>/{trace,pmc1,pmc2}
set -u
TAG=${1:-r04_sh8}
shift
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
ARGS="bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-extras --emulate-shard 8 --frames-per-step 3 $*"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o t -- python3 $ARGS > $OUT/trace.log 2>&1
timeout 300 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc1 -o p -- python3 $ARGS > $OUT/pmc1.log 2>&1
timeout 300 rocprofv3 --pmc FETCH_SIZE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INST_LEVEL_VMEM SQ_ACTIVE_INST_ANY SQ_INSTS_SALU SQ_ACTIVE_INST_VMEM --output-format csv -d $OUT/pmc2 -o p -- python3 $ARGS > $OUT/pmc2.log 2>&1
# (a third pass with TCP_* / TA_BUSY counters exceeds what the hardware collects in one pass on this pool: rocprofv3 aborts — left out)
python3 tools/pmc_summary.py $(find $OUT/pmc* -name "*counter_collection.csv") > $OUT/pmc_summary.txt
cp $(find $OUT/trace -name "*kernel_stats.csv" | head -1) $OUT/${TAG}_kernel_stats.csv
grep -E "k_path|k_trace|k_shade" $OUT/${TAG}_kernel_stats.csv | sed "s/(lptd::DScene[^\"]*\"/\"/" | cut -c1-160
grep -A14 "k_path\|k_trace<\|k_shade<" $OUT/pmc_summary.txt | head -120
for f in $OUT/*.log; do grep -E 'rror|abort' $f | head -2; done
