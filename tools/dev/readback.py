import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import loupiote_amd as lp
from loupiote_amd import scenes, testing as T
W, H = 1920, 1080
dev = lp.Device(0)
desc = scenes.synthetic_atrium()
sg = lp.SceneGPU.new_from_scene(scenes.to_product(desc), dev)
probe = lp.ProbeGPU(dev, desc["probe"], desc["probe"].shape[1], desc["probe"].shape[0])
view = T.look(desc["camera"]["origin"], desc["camera"]["direction"])
r = lp.Renderer(dev, (W, H)); r.downsample_factor = 1.0; r.resize(dev, sg, probe, (W, H)); r.set_max_bounces(8); r.set_vfov(T.VFOV)
def t(f, n=10):
    f(); r.synchronize()
    t0 = time.perf_counter()
    for _ in range(n): f()
    r.synchronize()
    return (time.perf_counter() - t0) / n * 1e3
def unb():
    r.reset_accumulation(); r.accumulate = True
    for _ in range(4): r.raytrace(view)
    r.synchronize()
def bat():
    r.reset_accumulation(); r.accumulate = True
    r.raytrace_n(view, 4)
    r.synchronize()
print("4 x raytrace + sync: %.2f ms" % t(unb))
print("raytrace_n(4) + sync: %.2f ms" % t(bat))
print("read_radiance: %.2f ms" % t(lambda: r.read_radiance()))
print("read_pixels: %.2f ms" % t(lambda: r.read_pixels()))
