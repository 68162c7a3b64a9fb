set -u
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r02l
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_atrium.py -x -q -m gpu 2>&1 | tail -2
LPT_TRI2=8 timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_atrium.py tests/test_gpu_golden.py -x -q -m gpu 2>&1 | tail -2
for s in 65 40 24 16 8 1; do
  LPT_TRI2=$s timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras > gpurun_out/r02l/tri2_$s.json 2> gpurun_out/r02l/tri2_$s.err
  python - <<PY
import json
j=json.loads(open("gpurun_out/r02l/tri2_$s.json").read().strip().splitlines()[-1])
r=j["roofline"]
print("tri2 $s: %.0f Mrays/s %.2f ms/frame  solo launch %.3f ms lanes %s" % (j["value"], j["ms_per_frame"], r["avg_launch_ms"], {k: round(v,1) for k,v in r["wave"].items()}))
PY
done
