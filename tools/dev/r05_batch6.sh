#!/bin/bash
# round 5, final batch: the profiles of the final kernels (whole frame; the 1/8 shard, per-bounce and pool), the other configs' timings, test durations
OUT=gpurun_out/r05p
mkdir -p $OUT
bash tools/profile.sh r05b > $OUT/profile_r05b.log 2>&1
tail -3 $OUT/profile_r05b.log
bash tools/profile_shard.sh r05b_sh8_pb > $OUT/profile_sh8_pb.log 2>&1
bash tools/profile_shard.sh r05b_sh8_pool --opt pool_rays=2147483647 > $OUT/profile_sh8_pool.log 2>&1
timeout 900 python tools/configs_timing.py > $OUT/configs_timing.jsonl 2> $OUT/configs_timing.err
tail -3 $OUT/configs_timing.jsonl | cut -c1-300
timeout 2000 python -m pytest tests -m gpu -x -q --durations=30 > $OUT/tests.log 2>&1
tail -45 $OUT/tests.log
