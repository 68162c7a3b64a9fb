set -u
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r02g
timeout 1500 python -m pytest tests -x -q -m gpu --durations=8 > gpurun_out/r02g/pytest.log 2>&1; echo "pytest exit $?" >> gpurun_out/r02g/pytest.log
tail -25 gpurun_out/r02g/pytest.log
timeout 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/r02g/bench.json 2> gpurun_out/r02g/bench.err
python - <<PY
import json
j=json.loads(open("gpurun_out/r02g/bench.json").read().strip().splitlines()[-1])
r=j["roofline"]
print("%.0f Mrays/s %.2f ms/frame  solo launch %.3f ms  shade solo %.2f ms/frame trace solo %.2f latency %s drop_in %.2f" % (j["value"], j["ms_per_frame"], r["avg_launch_ms"], j["stage_ms_per_frame_solo"]["shading"], j["stage_ms_per_frame_solo"]["intersection"], j["latency_ms"]["median"], j["drop_in"]["ms_per_frame"]))
PY
