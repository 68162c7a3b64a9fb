#!/usr/bin/env python3
"""Differential fuzz of the tail in place: random frame sizes, depths, sample counts, refill thresholds, waves per CU and tail widths on three scenes; every frame
must equal the one the plain per-lane launches give (LPT_OPT_TAIL_LANES 0, no step budget), bit for bit, with equal ray counts.  GPU box: python tools/dev/r05_tail_fuzz.py [cases] [seed]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: F401,E402

import loupiote_amd as lp  # noqa: E402
from loupiote_amd import scenes, testing as T  # noqa: E402

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 120
mode = sys.argv[3] if len(sys.argv) > 3 else "tail"   # "coop": small frames, a wave per ray (forced / as shipped) against the plain per-lane launches
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 5)
dev = lp.Device(0)
world = []
for desc in (scenes.synthetic_hall(), scenes.synthetic_atrium(texture_size=64), scenes.synthetic_helmet(texture_size=64)):
    sg = lp.SceneGPU.new_from_scene(scenes.to_product(desc), dev)
    pr = lp.ProbeGPU(dev, desc["probe"], desc["probe"].shape[1], desc["probe"].shape[0])
    world.append((desc, sg, pr))


def render(desc, sg, pr, size, depth, spp, frames, opts, shard):
    r = lp.Renderer(dev, size)
    r.downsample_factor = 1.0
    r.resize(dev, sg, pr, size)
    r.set_max_bounces(depth)
    r.set_vfov(T.VFOV)
    for k, v in opts.items():
        r.set_option(k, v)
    if shard:
        r.set_shard(*shard)
        r.set_resources(dev, sg, pr)
    r.reset_accumulation(); r.accumulate = True; r.reset_ray_counts()
    view = T.look(desc["camera"]["origin"], desc["camera"]["direction"])
    for _ in range(frames):
        r.raytrace_n(view, spp)
    img, c = r.read_radiance(), r.ray_counts()
    r.close()
    return img.tobytes(), (c.closest, c.shadow, c.shaded)


bad = 0
for k in range(cases):
    desc, sg, pr = world[k % len(world)]
    size = (int(rng.integers(17, 420)), int(rng.integers(9, 260)))
    depth, spp, frames = int(rng.integers(1, 9)), int(rng.integers(1, 6)), int(rng.integers(1, 3))
    base = {"path_rays": 0, "step_budget": 0, "tail_lanes": 0, "coop_rays": 0}
    opts = dict(base, tail_lanes=int(rng.integers(1, 9)), refill=int(rng.integers(0, 64)), trace_waves_per_cu=int(rng.choice([0, 1, 2, 5, 24, 32])),
                pipe_rays=int(rng.choice([0, 0x7FFFFFFF])), packet_primary=int(rng.integers(0, 3)), wavefront_rays=int(rng.choice([4194304, 20000, 70000])))
    if mode == "coop":
        size = (int(rng.integers(9, 200)), int(rng.integers(5, 120)))
        opts = dict({} if rng.random() < 0.5 else {"coop_rays": 0x7FFFFFFF}, packet_primary=int(rng.integers(0, 3)), wavefront_rays=int(rng.choice([4194304, 5000, 30000])))
    shard = None if rng.random() < 0.6 else (int(rng.integers(0, 3)), 3)
    ref = render(desc, sg, pr, size, depth, spp, frames, base, shard)
    got = render(desc, sg, pr, size, depth, spp, frames, opts, shard)
    ok = ref == got
    bad += 0 if ok else 1
    if not ok or k % 20 == 0:
        print("%3d %-28s %4dx%-4d depth %d spp %d frames %d shard %s %s -> %s %s" % (k, desc["name"][:28], size[0], size[1], depth, spp, frames, shard, {a: b for a, b in opts.items() if a not in base or a in ("tail_lanes", "coop_rays")}, "identical" if ok else "DIFFERENT", got[1]), flush=True)
print("%s fuzz: %d cases, %d different" % (mode, cases, bad))
sys.exit(1 if bad else 0)
