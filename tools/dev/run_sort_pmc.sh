set -u
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r02p
for m in 0 3; do
  export LPT_SORT=$m
  timeout 150 rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --output-format csv -d gpurun_out/r02p/pmc_$m -o p -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-extras --frames-per-step 3 > gpurun_out/r02p/log_$m.txt 2>&1
  timeout 150 rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/r02p/pmcf_$m -o p -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-extras --frames-per-step 3 > gpurun_out/r02p/logf_$m.txt 2>&1
  python3 tools/pmc_summary.py $(find gpurun_out/r02p/pmc_$m gpurun_out/r02p/pmcf_$m -name "*counter_collection.csv") > gpurun_out/r02p/summary_$m.txt
  echo "== LPT_SORT=$m"; grep -A5 "== k_shade<false>" gpurun_out/r02p/summary_$m.txt; grep -A5 "== k_trace<false>" gpurun_out/r02p/summary_$m.txt
done
