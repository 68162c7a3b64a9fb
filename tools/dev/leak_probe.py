import torch, numpy as np, loupiote_amd as lp
from loupiote_amd import testing as T
dev = lp.Device(0)
glb = open("tests/golden/cornell-box.glb", "rb").read()
scene = lp.Scene(); lp.loaders.load_gltf(glb, scene)
view = T.look(T.CORNELL_EYE, T.CORNELL_DIR)
def free(): dev.synchronize(); return torch.cuda.mem_get_info(0)[0]
def measure(name, fn, n=10):
    fn(); a = free()
    for _ in range(n): fn()
    b = free(); print("%-28s %8.2f KiB per cycle" % (name, (a - b) / n / 1024))
def scene_host(): lp.SceneGPU.new_from_scene(scene, dev).close()
def scene_gpu(): lp.SceneGPU.new_from_scene(scene, dev, gpu_build=True).close()
sg = lp.SceneGPU.new_from_scene(scene, dev); pr = lp.ProbeGPU(dev, T.CORNELL_PROBE, 1, 1)
def probe(): lp.ProbeGPU(dev, T.CORNELL_PROBE, 1, 1).close()
def rend_create(): lp.Renderer(dev, (64, 64)).close()
def rend_resize():
    r = lp.Renderer(dev, (64, 64)); r.downsample_factor = 1.0; r.resize(dev, sg, pr, (200, 120)); r.close()
def rend_trace():
    r = lp.Renderer(dev, (64, 64)); r.downsample_factor = 1.0; r.resize(dev, sg, pr, (200, 120)); r.accumulate = True; r.raytrace_n(view, 3); r.read_pixels(); r.close()
def rend_resize2():
    r = lp.Renderer(dev, (64, 64)); r.downsample_factor = 1.0
    for s in [(64, 48), (208, 120), (33, 9)]: r.resize(dev, sg, pr, s); r.raytrace(view)
    r.close()
def rend_den():
    r = lp.Renderer(dev, (64, 64)); r.downsample_factor = 1.0; r.resize(dev, sg, pr, (200, 120)); r.set_blit_mode(lp.BlitMode.DenoisedPathrace); r.raytrace(view); r.read_pixels(); r.close()
def upd(): sg.update_instances(scene)
for name, fn in [("scene host", scene_host), ("scene gpu-built", scene_gpu), ("probe", probe), ("renderer create", rend_create), ("renderer resize", rend_resize),
                 ("renderer trace n=3", rend_trace), ("renderer 3 resizes", rend_resize2), ("renderer denoise", rend_den), ("update_instances", upd)]:
    measure(name, fn)
