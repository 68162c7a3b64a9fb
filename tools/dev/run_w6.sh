set -u
cd "$GRAFT_REPO_ROOT"
for w in 24 0; do
  if [ $w = 0 ]; then unset LPT_WAVES_PER_CU; else export LPT_WAVES_PER_CU=$w; fi
  timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras | python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=j['roofline']
print('waves/CU $w: %.0f Mrays/s %.2f ms/frame solo launch %.3f ms' % (j['value'], j['ms_per_frame'], r['avg_launch_ms']))"
done
