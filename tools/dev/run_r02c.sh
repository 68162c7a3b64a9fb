set -u
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r02c
timeout 600 python bench.py --steps 20 --warmup 5 > gpurun_out/r02c/bench.json 2> gpurun_out/r02c/bench.err; echo "rc $?"
tail -c 6000 gpurun_out/r02c/bench.json
timeout 300 python bench.py --steps 3 --warmup 1 --force-dist --no-cpu-baseline > gpurun_out/r02c/bench_dist.json 2> gpurun_out/r02c/bench_dist.err; echo "rc $?"
tail -3 gpurun_out/r02c/bench_dist.err
python - <<PY
import json
for f in ("bench_dist",):
    j=json.loads(open("gpurun_out/r02c/%s.json"%f).read().strip().splitlines()[-1])
    print(f, j["value"], j["ms_per_frame"], j["latency_ms"], j["config"]["exchange"])
PY
timeout 300 python bench.py --steps 3 --warmup 1 --force-dist --exchange reduce --no-cpu-baseline --no-extras > gpurun_out/r02c/bench_dist_r.json 2> gpurun_out/r02c/bench_dist_r.err; echo "rc $?"
tail -c 300 gpurun_out/r02c/bench_dist_r.json
