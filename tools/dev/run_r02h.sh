set -u
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r02h
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_atrium.py tests/test_gpu_golden.py tests/test_gpu_denoiser.py -x -q -m gpu 2>&1 | tail -4
for s in ${SWEEP:-44}; do
LPT_REFILL=$s timeout 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/r02h/bench_$s.json 2> gpurun_out/r02h/bench_$s.err
python - <<PY
import json
j=json.loads(open("gpurun_out/r02h/bench_$s.json").read().strip().splitlines()[-1])
r=j["roofline"]
print("refill $s: %.0f Mrays/s %.2f ms/frame  solo launch %.3f ms  shade solo %.2f ms/frame trace solo %.2f latency %.2f drop_in %.2f lanes %s" % (j["value"], j["ms_per_frame"], r["avg_launch_ms"], j["stage_ms_per_frame_solo"]["shading"], j["stage_ms_per_frame_solo"]["intersection"]+j["stage_ms_per_frame_solo"]["shadow"], j["latency_ms"]["median"], j["drop_in"]["ms_per_frame"], {k: round(v,1) for k,v in r["wave"].items()}))
PY
done
