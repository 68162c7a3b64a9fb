#!/bin/bash
# soak of the tail-in-place path: which lanes end in a cooperative walk depends on the scheduling of the waves; the frames must not.  tools/dev/r05_tail_soak.sh <n>
N=${1:-12}
fail=0
for i in $(seq 1 $N); do
  timeout 600 python -m pytest tests/test_gpu_tail.py tests/test_gpu_hall.py -x -q -p no:cacheprovider > gpurun_out/tail_soak_$i.txt 2>&1 || { fail=$((fail+1)); echo "run $i FAILED"; tail -20 gpurun_out/tail_soak_$i.txt; }
  tail -1 gpurun_out/tail_soak_$i.txt
done
echo "soak: $N runs, $fail failed"
