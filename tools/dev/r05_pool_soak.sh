#!/bin/bash
# round 5: k_pool under repetition — the pool tests and the pool arm of the atrium / deferred / edge parity tests, N times; any flake (a race in the hand-over
# protocol would show as a differing frame, a bounded wait that ran out as LPT_ERR_HIP) fails the loop
N=${1:-12}
fail=0
for i in $(seq 1 $N); do
  timeout 600 python -m pytest tests/test_gpu_pool.py tests/test_gpu_atrium.py tests/test_gpu_deferred.py tests/test_gpu_edge.py tests/test_gpu_denoiser.py -m gpu -x -q -k "pool" > gpurun_out/r05s_soak_$i.log 2>&1 || { fail=$((fail+1)); tail -20 gpurun_out/r05s_soak_$i.log; }
  tail -1 gpurun_out/r05s_soak_$i.log
done
timeout 900 python tools/dev/r05_pool_check.py 9 2>&1 | tail -11
echo "soak: $fail of $N iterations failed"
