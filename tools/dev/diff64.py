#!/usr/bin/env python3
"""On the GPU box: HIP vs the oracle after many progressive samples at a reduced size: where do they start to differ?"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import loupiote_amd as lp
from loupiote_amd import scenes, testing as T
from oracle import harness, orc

W, H, N = 640, 360, int(sys.argv[1]) if len(sys.argv) > 1 else 64
dev = lp.Device(0)
desc = scenes.synthetic_atrium(texture_size=256)
sg = lp.SceneGPU.new_from_scene(scenes.to_product(desc), dev)
pr = lp.ProbeGPU(dev, desc["probe"], desc["probe"].shape[1], desc["probe"].shape[0])
view = T.look(desc["camera"]["origin"], desc["camera"]["direction"])
osc = orc.OracleScene.from_scene(harness.to_oracle(desc), probe=desc["probe"])
r = lp.Renderer(dev, (W, H)); r.downsample_factor = 1.0; r.resize(dev, sg, pr, (W, H)); r.set_max_bounces(8); r.set_vfov(T.VFOV)
if len(sys.argv) > 2: r.set_max_fused(int(sys.argv[2]))
r.reset_accumulation(); r.accumulate = True
prev_bad = 0
for n in (1, 2, 4, 8, 16, 32, N):
    acc, cnt = osc.render(W, H, view, T.VFOV, 8, frames=n, want_counters=True)
    ref = orc.resolve(acc)
    r.reset_accumulation(); r.accumulate = True
    # the seed is never reset: use a fresh renderer state instead
    rr = lp.Renderer(dev, (W, H)); rr.downsample_factor = 1.0; rr.resize(dev, sg, pr, (W, H)); rr.set_max_bounces(8); rr.set_vfov(T.VFOV)
    if len(sys.argv) > 2: rr.set_max_fused(int(sys.argv[2]))
    rr.reset_accumulation(); rr.accumulate = True
    for _ in range(n):
        rr.raytrace(view); rr.accumulate = True
    img = rr.read_radiance()
    bad = np.argwhere(np.any(img != ref, axis=-1))
    print("n=%d: %d differing pixels" % (n, len(bad)), flush=True)
    for y, x in bad[:5]:
        print("   ", (x, y), img[y, x], ref[y, x], (img[y, x] - ref[y, x]))
    rr.close()
