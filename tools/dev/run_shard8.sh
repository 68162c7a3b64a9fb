set -u
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r02n
for cfg in "0 3" "0 4" "0 6" "0 8" "8 4" "12 4" "24 4" "8 8" "12 8" ; do
  set -- $cfg
  if [ $1 = 0 ]; then unset LPT_WAVES_PER_CU; else export LPT_WAVES_PER_CU=$1; fi
  GPU_MAX_HW_QUEUES=8 timeout 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extras --emulate-shard 8 --pipeline $2 > gpurun_out/r02n/s_$1_$2.json 2> gpurun_out/r02n/s_$1_$2.err
  python - <<PY
import json
j=json.loads(open("gpurun_out/r02n/s_$1_$2.json").read().strip().splitlines()[-1])
print("1/8 shard, trace waves/CU $1, pipeline $2: %.3f ms/frame  (%.0f Mrays/s on this GPU)" % (j["ms_per_frame"], j["value"]))
PY
done
for n in 2 4; do for p in 3 4 6; do
  unset LPT_WAVES_PER_CU
  timeout 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extras --emulate-shard $n --pipeline $p > gpurun_out/r02n/n_${n}_$p.json 2> /dev/null
  python - <<PY
import json
j=json.loads(open("gpurun_out/r02n/n_${n}_$p.json").read().strip().splitlines()[-1])
print("1/$n shard, pipeline $p: %.3f ms/frame" % (j["ms_per_frame"]))
PY
done; done
