set -u
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r02e
timeout 600 python -m pytest tests/test_gpu_sorted.py -x -q -m gpu 2>&1 | tail -5
for s in 0 1 2 3; do
  LPT_SORT=$s timeout 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extras > gpurun_out/r02e/sort_$s.json 2> gpurun_out/r02e/sort_$s.err
  python - <<PY
import json
j=json.loads(open("gpurun_out/r02e/sort_$s.json").read().strip().splitlines()[-1])
r=j["roofline"]
print("sort $s: %.0f Mrays/s %.2f ms/frame  solo launch %.3f ms  shade solo %.2f ms/frame trace solo %.2f nodes/ray %.2f lanes %s" % (j["value"], j["ms_per_frame"], r["avg_launch_ms"], j["stage_ms_per_frame_solo"]["shading"], j["stage_ms_per_frame_solo"]["intersection"], r["nodes_per_ray"], {k: round(v,1) for k,v in r["wave"].items()}))
PY
done
