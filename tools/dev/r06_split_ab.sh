#!/bin/bash
# round 6: the host builder's triangle pre-splitting (bvh.cpp presplit) on / off / other thresholds on the Sponza-shaped scenes: tools/dev/r06_split_ab.sh <out>
# needs loupiote_amd/libloupiote_hip_exp.so = the library with bvh.cpp compiled -DLPT_EXPERIMENTS (reads LPT_BVH_SPLIT="ratio,budget")
OUT=gpurun_out/$1; mkdir -p $OUT
export LPT_LIB_PATH=$PWD/loupiote_amd/libloupiote_hip_exp.so
for sp in "1e30,0" "16,0.3" "4,0.3" "1,0.5"; do
  echo "== LPT_BVH_SPLIT=$sp"
  LPT_BVH_SPLIT=$sp PYTHONPATH=. python tools/configs_timing.py "hall" "4: synthetic_atrium" 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    j = json.loads(l)
    print('  %-28s %7.2f ms/frame  nodes/ray %.2f tris/ray %.2f  shadow %.2f / %.2f  tree nodes %d' % (j['config'][:28], j['ms_per_frame'], j['nodes_per_ray'], j['tris_per_ray'], j['shadow_nodes_per_ray'], j['shadow_tris_per_ray'], j['nodes']))"
done
