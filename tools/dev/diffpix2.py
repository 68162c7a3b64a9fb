#!/usr/bin/env python3
"""On the GPU box: the one tile that holds a pixel where HIP and the oracle differ — oracle with its BVH, oracle by brute force over all triangles, HIP."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import loupiote_amd as lp
from loupiote_amd import scenes, testing as T
from oracle import harness, orc

W, H = 3840, 2160
K, PX, PY = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
tile = (PY // 8) * ((W + 31) // 32) + PX // 32
world = ((W + 31) // 32) * ((H + 7) // 8)
dev = lp.Device(0)
desc = scenes.synthetic_atrium()
sg = lp.SceneGPU.new_from_scene(scenes.to_product(desc), dev)
pr = lp.ProbeGPU(dev, desc["probe"], desc["probe"].shape[1], desc["probe"].shape[0])
view = T.look(desc["camera"]["origin"], desc["camera"]["direction"])
osc = orc.OracleScene.from_scene(harness.to_oracle(desc), probe=desc["probe"])
for d in (4, 5, 8):
    a, _ = osc.render(W, H, view, T.VFOV, d, frames=1, seed_counter=K * 8, rank=tile, world_size=world, threads=16, want_counters=True)
    b, _ = osc.render(W, H, view, T.VFOV, d, frames=1, seed_counter=K * 8, rank=tile, world_size=world, brute_force=True, threads=16, want_counters=True)
    r = lp.Renderer(dev, (W, H)); r.downsample_factor = 1.0; r.resize(dev, sg, pr, (W, H)); r.set_max_bounces(8); r.set_vfov(T.VFOV)
    r.reset_accumulation(); r.accumulate = False
    for _ in range(K):
        r.raytrace(view)
    r.set_max_bounces(d)
    r.raytrace(view)
    img = r.read_radiance()
    r.close()
    a, b = orc.resolve(a), orc.resolve(b)
    print("depth %d: oracle BVH %s | oracle brute force %s | HIP %s" % (d, a[PY, PX, :3], b[PY, PX, :3], img[PY, PX, :3]), flush=True)
    y0, x0 = (PY // 8) * 8, (PX // 32) * 32
    print("   tile: oracle BVH == brute force: %s ; HIP == brute force: %s" % (np.array_equal(a[y0:y0 + 8, x0:x0 + 32], b[y0:y0 + 8, x0:x0 + 32]), np.array_equal(img[y0:y0 + 8, x0:x0 + 32], b[y0:y0 + 8, x0:x0 + 32])))
