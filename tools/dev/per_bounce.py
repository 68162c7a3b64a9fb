#!/usr/bin/env python3
"""Per-bounce ray counts of one bench frame (run under `rocprofv3 --kernel-trace` to get the matching per-launch
k_trace / k_shade durations; tools/dev/per_bounce_join.py joins the two)."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import loupiote_amd as lp  # noqa: E402
from loupiote_amd import scenes, testing as T  # noqa: E402

W, H, SPP, DEPTH = 1920, 1080, int(os.environ.get("PB_SPP", "4")), 8
dev = lp.Device(0)
desc = scenes.synthetic_atrium()
sg = lp.SceneGPU.new_from_scene(scenes.to_product(desc), dev)
probe = lp.ProbeGPU(dev, desc["probe"], desc["probe"].shape[1], desc["probe"].shape[0])
view = T.look(desc["camera"]["origin"], desc["camera"]["direction"])
r = lp.Renderer(dev, (W, H))
r.downsample_factor = 1.0
r.resize(dev, sg, probe, (W, H))
r.set_max_bounces(DEPTH)
r.set_vfov(T.VFOV)
for k in range(3):
    r.reset_accumulation()
    r.accumulate = True
    r.raytrace_n(view, SPP)
    r.synchronize()
c, s = r.queue_counts(DEPTH)
print(json.dumps({"closest": c.tolist(), "shadow": s.tolist()}))
