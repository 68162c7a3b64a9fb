set -u
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r02f
for s in 32 44 52 58 62; do
  LPT_REFILL=$s timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras > gpurun_out/r02f/refill_$s.json 2> gpurun_out/r02f/refill_$s.err
  python - <<PY
import json
j=json.loads(open("gpurun_out/r02f/refill_$s.json").read().strip().splitlines()[-1])
r=j["roofline"]
print("refill $s: %.0f Mrays/s %.2f ms/frame  solo launch %.3f ms  trace solo %.2f lanes %s" % (j["value"], j["ms_per_frame"], r["avg_launch_ms"], j["stage_ms_per_frame_solo"]["intersection"], {k: round(v,1) for k,v in r["wave"].items()}))
PY
done
