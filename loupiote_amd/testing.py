"""Shared scene/camera setup for tests, smoke() and bench.py (host-side plumbing only; the oracle-side
counterpart lives in oracle/harness.py — nothing in this package imports oracle/)."""
import numpy as np

from . import api
from ._abi import LIGHT_DT

VFOV = 0.78539816339744830962

# build-defined Cornell setup: the asset has no camera and no emitter
CORNELL_EYE = (0.0, 0.6, 13.5)
CORNELL_DIR = (0.0, 0.0, -1.0)
CORNELL_PROBE = np.array([[[64, 64, 64, 128]]], np.uint8)  # RGBE: constant 0.25 grey


def cornell_light():
    l = np.zeros(1, LIGHT_DT)
    l["normal"] = (0, -1, 0, 0)
    l["tangent"] = (1, 0, 0, 0.8)
    l["bitangent"] = (0, 0, 1, 0.8)
    l["origin"] = (0, 3.55, 0.4, 12.0)
    return l


def look(origin, direction):
    return api.CameraController.from_origin_dir(origin, direction).update(0.0)


def render_hip(dev, glb, width, height, bounces, frames, seed=0, rank=0, world=1, light=None, probe=CORNELL_PROBE,
               eye=CORNELL_EYE, direction=CORNELL_DIR, raw_accum=False, options=None):
    """reset_accumulation(); accumulate = true; frames x raytrace(); read back.  Returns (radiance, ray counts)."""
    scene = api.Scene()
    api.loaders.load_gltf(glb, scene)
    scene.set_light(0, cornell_light() if light is None else light)
    sg = api.SceneGPU.new_from_scene(scene, dev)
    pr = api.ProbeGPU(dev, probe, probe.shape[1], probe.shape[0]) if probe is not None else None
    r = api.Renderer(dev, (width, height))
    r.downsample_factor = 1.0
    r.resize(dev, sg, pr, (width, height))
    r.set_max_bounces(bounces)
    r.set_seed(seed)
    r.set_vfov(VFOV)
    for k, v in (options or {}).items():
        r.set_option(k, v)
    if world > 1:
        r.set_shard(rank, world)
        r.set_resources(dev, sg, pr)
    view = look(eye, direction)
    r.reset_accumulation()
    r.accumulate = True
    r.reset_ray_counts()
    for _ in range(frames):
        r.raytrace(view)
    img = r.read_radiance()
    counts = r.ray_counts()
    r.close()
    if pr is not None:
        pr.close()
    sg.close()
    return img, counts
