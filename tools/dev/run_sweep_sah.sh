set -u
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r02q
for sw in 0 16 64 256 1024; do
  LPT_BVH_SWEEP=$sw timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras > gpurun_out/r02q/sw_$sw.json 2> gpurun_out/r02q/sw_$sw.err
  python - <<PY
import json
j=json.loads(open("gpurun_out/r02q/sw_$sw.json").read().strip().splitlines()[-1])
r=j["roofline"]
print("sweep<=$sw: %.0f Mrays/s %.2f ms/frame nodes/ray %.2f tris/ray %.2f shadow %.2f/%.2f nodes %d build %.0f ms" % (j["value"], j["ms_per_frame"], r["nodes_per_ray"], r["tris_per_ray"], r["shadow_nodes_per_ray"], r["shadow_tris_per_ray"], j["accel"]["nodes"], j["accel"]["build_ms"]))
PY
done
