set -u
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r02j
timeout 900 python bench.py --steps 20 --warmup 5 > gpurun_out/r02j/bench.json 2> gpurun_out/r02j/bench.err; echo rc $?
python - <<PY
import json
j=json.loads(open("gpurun_out/r02j/bench.json").read().strip().splitlines()[-1])
print(j["value"], j["ms_per_frame"], j["latency_ms"]["median"], j["drop_in"]["ms_per_frame"], j["cpu_baseline"])
PY
for n in 2 4 8; do
timeout 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extras --emulate-shard $n > gpurun_out/r02j/shard_$n.json 2> gpurun_out/r02j/shard_$n.err
python - <<PY
import json
j=json.loads(open("gpurun_out/r02j/shard_$n.json").read().strip().splitlines()[-1])
print("1/$n shard: %.3f ms/frame (3 in flight), rays/frame %.3g" % (j["ms_per_frame"], j["config"]["rays_per_frame"]))
PY
done
timeout 600 python tools/configs_timing.py > gpurun_out/r02j/configs_timing.jsonl 2> gpurun_out/r02j/configs.err; cat gpurun_out/r02j/configs_timing.jsonl | cut -c1-300
