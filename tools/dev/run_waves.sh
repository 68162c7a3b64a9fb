set -u
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r02i
for s in 0 8 12 16 24 32; do
  if [ $s = 0 ]; then unset LPT_WAVES_PER_CU; else export LPT_WAVES_PER_CU=$s; fi
  timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras --no-batch --pipeline 1 > gpurun_out/r02i/w_$s.json 2> gpurun_out/r02i/w_$s.err
  python - <<PY
import json
j=json.loads(open("gpurun_out/r02i/w_$s.json").read().strip().splitlines()[-1])
r=j["roofline"]
print("waves/CU $s (unbatched, 1 in flight): %.0f Mrays/s %.2f ms/frame  solo launch %.3f ms lanes %s" % (j["value"], j["ms_per_frame"], r["avg_launch_ms"], {k: round(v,1) for k,v in r["wave"].items()}))
PY
done
