set -u
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r02k
timeout 600 python tools/configs_timing.py > gpurun_out/r02k/configs_timing.jsonl 2> gpurun_out/r02k/configs.err; cat gpurun_out/r02k/configs_timing.jsonl | cut -c1-400; tail -3 gpurun_out/r02k/configs.err
timeout 600 python bench.py --steps 20 --warmup 5 > gpurun_out/r02k/bench.json 2> gpurun_out/r02k/bench.err; echo rc $?
python - <<PY
import json
j=json.loads(open("gpurun_out/r02k/bench.json").read().strip().splitlines()[-1])
print(j["value"], j["ms_per_frame"], j["latency_ms"]["median"], j["drop_in"]["ms_per_frame"], j["cpu_baseline"])
PY
