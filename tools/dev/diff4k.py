#!/usr/bin/env python3
"""On the GPU box: config 5 (3840x2160, depth 8) sample by sample, HIP against the oracle: the first samples / pixels that differ."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import loupiote_amd as lp
from loupiote_amd import scenes, testing as T
from oracle import harness, orc

W, H = 3840, 2160
K0, K1 = int(sys.argv[1]), int(sys.argv[2])
dev = lp.Device(0)
desc = scenes.synthetic_atrium()
sg = lp.SceneGPU.new_from_scene(scenes.to_product(desc), dev)
pr = lp.ProbeGPU(dev, desc["probe"], desc["probe"].shape[1], desc["probe"].shape[0])
view = T.look(desc["camera"]["origin"], desc["camera"]["direction"])
osc = orc.OracleScene.from_scene(harness.to_oracle(desc), probe=desc["probe"])
r = lp.Renderer(dev, (W, H)); r.downsample_factor = 1.0; r.resize(dev, sg, pr, (W, H)); r.set_max_bounces(8); r.set_vfov(T.VFOV)
r.reset_accumulation(); r.accumulate = False
found = 0
for k in range(K1):
    r.raytrace(view)                       # accumulate == false: the target holds sample k alone
    if k < K0:
        continue
    img = r.read_radiance()
    t0 = time.time()
    acc, cnt = osc.render(W, H, view, T.VFOV, 8, frames=1, seed_counter=k * 8, threads=16, want_counters=True)
    ref = orc.resolve(acc)
    bad = np.argwhere(np.any(img != ref, axis=-1))
    print("sample %d: %d differing pixels (oracle %.1f s)" % (k, len(bad), time.time() - t0), flush=True)
    for y, x in bad[:8]:
        print("    pixel", (int(x), int(y)), "hip", img[y, x], "oracle", ref[y, x], flush=True)
    found += len(bad)
    if found > 8:
        break
