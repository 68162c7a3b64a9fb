#!/usr/bin/env python3
"""On the GPU box: one pixel of one sample of config 5, depth-limited 1..8 with the seeds of the full-depth frame: at which bounce do HIP and the oracle part?"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import loupiote_amd as lp
from loupiote_amd import scenes, testing as T
from oracle import harness, orc

W, H = 3840, 2160
K, PX, PY = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
dev = lp.Device(0)
desc = scenes.synthetic_atrium()
sg = lp.SceneGPU.new_from_scene(scenes.to_product(desc), dev)
pr = lp.ProbeGPU(dev, desc["probe"], desc["probe"].shape[1], desc["probe"].shape[0])
view = T.look(desc["camera"]["origin"], desc["camera"]["direction"])
osc = orc.OracleScene.from_scene(harness.to_oracle(desc), probe=desc["probe"])
for d in range(1, 9):
    r = lp.Renderer(dev, (W, H)); r.downsample_factor = 1.0; r.resize(dev, sg, pr, (W, H)); r.set_max_bounces(8); r.set_vfov(T.VFOV)
    r.reset_accumulation(); r.accumulate = False
    for _ in range(K):
        r.raytrace(view)
    r.set_max_bounces(d)
    r.raytrace(view)
    img = r.read_radiance()
    c = r.ray_counts()
    r.close()
    acc, cnt = osc.render(W, H, view, T.VFOV, d, frames=1, seed_counter=K * 8, threads=16, want_counters=True)
    ref = orc.resolve(acc)
    bad = np.argwhere(np.any(img != ref, axis=-1))
    print("depth %d: pixel hip %s oracle %s | %d differing pixels in the frame %s" % (d, img[PY, PX, :3], ref[PY, PX, :3], len(bad), [tuple(b[::-1]) for b in bad[:4]]), flush=True)
