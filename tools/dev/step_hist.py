"""dev: traversal steps per ray (stats kernels) for a 1/N shard of the bench frame: usage step_hist.py [shard]"""
import sys
import numpy as np
import loupiote_amd as lp
from loupiote_amd import scenes, testing as T
shard = int(sys.argv[1]) if len(sys.argv) > 1 else 8
dev = lp.Device(0)
desc = scenes.synthetic_atrium()
sg = lp.SceneGPU.new_from_scene(scenes.to_product(desc), dev)
pr = lp.ProbeGPU(dev, desc["probe"], desc["probe"].shape[1], desc["probe"].shape[0])
r = lp.Renderer(dev, (1920, 1080))
r.downsample_factor = 1.0
r.resize(dev, sg, pr, (1920, 1080))
r.set_max_bounces(8)
r.set_vfov(T.VFOV)
r.set_option("path_rays", 0)
r.set_option("step_budget", 0)   # a ray dropped at the budget would be missing from the histogram (the stats kernels run without it anyway since round 5)
if shard > 1:
    r.set_shard(0, shard, 32, 8)
    r.set_resources(dev, sg, pr)
view = T.look(desc["camera"]["origin"], desc["camera"]["direction"])
r.enable_stats(True)
r.reset_accumulation(); r.accumulate = True
r.raytrace_n(view, 4)
r.synchronize()
mx, h = r.step_histogram()
c = r.ray_counts()
tot = int(h.sum())
print("shard", shard, "rays traced by k_trace", tot, "closest+shadow-primary", c.closest + c.shadow - c.primary, "max steps", mx)
cum = 0
for k in range(12):
    cum += int(h[k])
    print("  %4d..%4d steps: %9d  (%.4f %%, cumulative %.5f %%)" % (2 ** k, 2 ** (k + 1) - 1, h[k], 100.0 * h[k] / max(tot, 1), 100.0 * cum / max(tot, 1)))
print("mean steps", sum((1.5 * 2 ** k) * int(h[k]) for k in range(12)) / max(tot, 1), "(bucket midpoints)")
