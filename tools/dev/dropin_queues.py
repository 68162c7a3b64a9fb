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
r = lp.Renderer(dev, (W, H)); r.downsample_factor = 1.0; r.resize(dev, sg, probe, (W, H)); r.set_max_bounces(8); r.set_vfov(T.VFOV)
def frame():
    r.reset_accumulation(); r.accumulate = True
    for _ in range(4): r.raytrace(view)
    return r.read_radiance()
for _ in range(3): frame()
t0 = time.perf_counter()
for _ in range(10): frame()
print("GPU_MAX_HW_QUEUES=%s: 4 x raytrace + read_radiance: %.2f ms per frame" % (os.environ.get("GPU_MAX_HW_QUEUES", "(unset)"), (time.perf_counter() - t0) * 100))
