set -u
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r02b
for pc in 0.1 0.2 0.3 0.45 0.7 1.0; do
  LPT_BVH_PRIM_COST=$pc timeout 300 python bench.py --steps 12 --warmup 3 --no-cpu-baseline > gpurun_out/r02b/pc_$pc.json 2> gpurun_out/r02b/pc_$pc.err
  python - <<PY
import json
j=json.loads(open("gpurun_out/r02b/pc_$pc.json").read().strip().splitlines()[-1])
r=j["roofline"]
print("prim_cost $pc: %.0f Mrays/s %.2f ms  nodes/ray %.2f tris/ray %.2f  nodes %d solo %.3f ms lanes %s" % (j["value"], j["ms_per_step"], r["nodes_per_ray"], r["tris_per_ray"], j["accel"]["nodes"], r["solo"]["avg_launch_ms"], {k: round(v,1) for k,v in r["wave"].items()}))
PY
done
