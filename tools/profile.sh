#!/bin/bash
# Run ON THE GPU BOX (via gpurun): kernel-trace stats + separate PMC passes for the bench workload.
# usage: tools/profile.sh <tag>     -> gpurun_out/prof_<tag>/{trace,solo,pmc1..4}
#   trace : rocprofv3 --kernel-trace --stats of the bench command (the two 2-sample wavefronts of a frame overlap on the renderer's lanes)
#   solo  : the same with --max-fused 4 --lanes 1 (one 4-sample wavefront per frame, nothing overlaps): what roofline.frac is computed from
#   pmc*  : counter passes, collected on their own (never combined with --sys-trace etc.)
# The program after `--` is python3 itself (no env/bash hop); every pass is bounded by `timeout` (a counter set the hardware
# cannot collect makes rocprofv3 abort and then hang in its finaliser).  --no-extras keeps the latency / throughput legs (launches of
# other sizes) out of the per-kernel averages.
set -u
TAG=${1:-r02}
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
ARGS="bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras"
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o t -- python3 $ARGS > $OUT/trace.log 2>&1
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/solo -o t -- python3 $ARGS --max-fused 4 --lanes 1 > $OUT/solo.log 2>&1
PARGS="bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-extras --frames-per-step 3 --max-fused 4 --lanes 1"   # counters per UN-OVERLAPPED launch
timeout 400 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc1 -o p -- python3 $PARGS > $OUT/pmc1.log 2>&1
timeout 400 rocprofv3 --pmc FETCH_SIZE SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INST_LEVEL_VMEM SQ_ACTIVE_INST_ANY SQ_INSTS_SALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS --output-format csv -d $OUT/pmc2 -o p -- python3 $PARGS > $OUT/pmc2.log 2>&1
timeout 400 rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/pmc3 -o p -- python3 $PARGS > $OUT/pmc3.log 2>&1
timeout 400 rocprofv3 --pmc TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum --output-format csv -d $OUT/pmc4 -o p -- python3 $PARGS > $OUT/pmc4.log 2>&1
find $OUT -name "*.csv" | xargs ls -la
for f in $OUT/*.log; do echo "== $f"; grep -E '"value"|rror' $f | cut -c1-200 | head -3; done
# reduce: summaries that get committed under profiles/
python3 tools/pmc_summary.py $(find $OUT/pmc* -name "*counter_collection.csv") > $OUT/pmc_summary.txt
cp $(find $OUT/trace -name "*kernel_stats.csv" | head -1) $OUT/${TAG}_kernel_stats.csv
cp $(find $OUT/solo -name "*kernel_stats.csv" | head -1) $OUT/${TAG}_solo_kernel_stats.csv
grep '"value"' $OUT/trace.log | tail -1 > $OUT/${TAG}_bench.json
grep '"value"' $OUT/solo.log | tail -1 > $OUT/${TAG}_solo_bench.json
python3 tools/limits_from_pmc.py $TAG $OUT
