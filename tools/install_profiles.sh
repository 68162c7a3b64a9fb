#!/bin/bash
# after `gpurun -- tools/profile.sh <tag>` (+ bench.py > gpurun_out/<tag>_bench_full.json, tools/profile_shard.sh <tag>_sh8): copy the summaries that get judged
# from gpurun_out/ (scratch) into profiles/ (tracked), drop the previous build's copies and re-point the docs.   usage: tools/install_profiles.sh <tag> <old tag>
set -eu
TAG=$1; OLD=$2
cd "$(dirname "$0")/.."
P=gpurun_out/prof_$TAG
for f in bench.json kernel_stats.csv solo_bench.json solo_kernel_stats.csv; do cp $P/${TAG}_$f profiles/; done
cp $P/pmc_summary.txt profiles/${TAG}_pmc_summary.txt
cp $P/limits.json $P/traffic.json profiles/
[ -f gpurun_out/${TAG}_bench_full.json ] && cp gpurun_out/${TAG}_bench_full.json profiles/
if [ -d gpurun_out/prof_${TAG}_sh8 ]; then
  cp gpurun_out/prof_${TAG}_sh8/${TAG}_sh8_kernel_stats.csv profiles/${TAG}_sh8_tail_kernel_stats.csv
  cp gpurun_out/prof_${TAG}_sh8/pmc_summary.txt profiles/${TAG}_sh8_tail_pmc_summary.txt
fi
for f in profiles/${OLD}_bench.json profiles/${OLD}_kernel_stats.csv profiles/${OLD}_solo_bench.json profiles/${OLD}_solo_kernel_stats.csv profiles/${OLD}_pmc_summary.txt \
         profiles/${OLD}_bench_full.json profiles/${OLD}_sh8_tail_kernel_stats.csv profiles/${OLD}_sh8_tail_pmc_summary.txt; do
  [ -f $f ] && git rm -q --cached $f 2>/dev/null; rm -f $f
done
for f in profiles/${OLD}_configs_timing.jsonl profiles/${OLD}_two_proc_standin_gather.json profiles/${OLD}_two_proc_standin_host.json; do
  [ -f $f ] && git mv $f ${f/${OLD}_/${TAG}_}
done
sed -i "s|${OLD}_|${TAG}_|g; s|\`${OLD}\`|\`${TAG}\`|g" profiles/README.md
sed -i "s|profiles/${OLD}_|profiles/${TAG}_|g" README.md DESIGN.md docs/ROUNDS.md profiles/r05_experiments_ab.txt
git add -A profiles README.md DESIGN.md docs/ROUNDS.md
python3 - <<PY
import json, sys
sys.path.insert(0, ".")
import bench
j = json.loads(open("profiles/${TAG}_bench_full.json").read().strip().splitlines()[-1])
r = j["roofline"]
print("value %.1f  ms/frame %.3f  frac %.3f  fetched %.3f  Grays/s %.2f  traffic %s  stale %s  hash %s (tree %s)" % (j["value"], j["ms_per_frame"], r["frac"], r["frac_fetched"], r["Grays_per_s"], r["traffic"], r["from_profiles"]["stale"], r["from_profiles"]["kernel_source_hash"]["profiles"], bench.kernel_source_hash()))
for k, v in j["shard_emulation"].items():
    if isinstance(v, dict):
        print(k, round(v["ms_per_frame"], 3), round(v["ms_per_frame_host_gather"], 3), v.get("speedup_vs_1"), v.get("speedup_vs_1_host_gather"))
PY
