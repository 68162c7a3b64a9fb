set -u
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r02m
for cfg in "32 4 3" "28 4 3" "24 4 3" "24 2 3" "24 1 3" "20 4 3" "24 4 4" "28 2 4" "16 4 4"; do
  set -- $cfg
  LPT_WAVES_PER_CU=$1 LPT_SHADE_BLOCKS_PER_CU=$2 timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras --pipeline $3 > gpurun_out/r02m/c_$1_$2_$3.json 2> gpurun_out/r02m/c_$1_$2_$3.err
  python - <<PY
import json
j=json.loads(open("gpurun_out/r02m/c_$1_$2_$3.json").read().strip().splitlines()[-1])
r=j["roofline"]
print("trace waves/CU $1 shade blocks/CU $2 pipeline $3: %.0f Mrays/s %.2f ms/frame  solo launch %.3f ms" % (j["value"], j["ms_per_frame"], r["avg_launch_ms"]))
PY
done
