import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import loupiote_amd as lp
from loupiote_amd import scenes, testing as T
W, H = 1920, 1080
dev = lp.Device(0)
desc = scenes.synthetic_atrium()
sg = lp.SceneGPU.new_from_scene(scenes.to_product(desc), dev)
probe = lp.ProbeGPU(dev, desc["probe"], desc["probe"].shape[1], desc["probe"].shape[0])
view = T.look(desc["camera"]["origin"], desc["camera"]["direction"])
for lanes in (1, 2, 3, 4):
    r = lp.Renderer(dev, (W, H)); r.set_lanes(lanes); r.downsample_factor = 1.0; r.resize(dev, sg, probe, (W, H)); r.set_max_bounces(8); r.set_vfov(T.VFOV)
    def one():
        r.reset_accumulation(); r.accumulate = True
        r.raytrace_n(view, 4)
        r.synchronize()
    for _ in range(3): one()
    ts = []
    for _ in range(9):
        t0 = time.perf_counter(); one(); ts.append((time.perf_counter() - t0) * 1e3)
    ts.sort()
    print("lanes %d: raytrace_n(4) alone: min %.2f median %.2f ms" % (lanes, ts[0], ts[len(ts)//2]))
    r.close()
