set -u
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r02o
timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -4
for l in 1 2 3; do
  LPT_LANES=$l timeout 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/r02o/lanes_$l.json 2> gpurun_out/r02o/lanes_$l.err
  python - <<PY
import json
j=json.loads(open("gpurun_out/r02o/lanes_$l.json").read().strip().splitlines()[-1])
r=j["roofline"]
print("lanes $l: %.0f Mrays/s %.2f ms/frame  solo launch %.3f ms  latency %.2f drop_in %.2f ms (%.0f Mrays/s)" % (j["value"], j["ms_per_frame"], r["avg_launch_ms"], j["latency_ms"]["median"], j["drop_in"]["ms_per_frame"], j["drop_in"]["value"]))
PY
done
