"""Test-side glue between scene descriptions and the oracle (TEST INFRASTRUCTURE — used by tests/, by
__graft_entry__.smoke() and by bench.py's cpu_baseline leg only; the product package never imports it).

`render_oracle` renders the Cornell fixture the way `loupiote_amd.testing.render_hip` drives the product;
`to_oracle` feeds a procedural scene description (loupiote_amd.scenes) into the oracle's numpy Scene."""
import numpy as np

from . import gltf_oracle as G
from . import orc


def render_oracle(glb, width, height, bounces, frames, seed=0, rank=0, world=1, light=None, probe=None,
                  eye=None, direction=None, brute_force=False, threads=None):
    from loupiote_amd import testing as T  # camera / light conventions shared with the product-side helper
    probe = T.CORNELL_PROBE if probe is None else probe
    eye = T.CORNELL_EYE if eye is None else eye
    direction = T.CORNELL_DIR if direction is None else direction
    s = G.Scene()
    G.load_gltf(glb, s)
    s.lights[0] = (T.cornell_light() if light is None else light)[0]
    sc = orc.OracleScene.from_scene(s, probe=probe)
    acc, cnt = sc.render(width, height, T.look(eye, direction), T.VFOV, bounces, frames=frames, user_seed=seed,
                         rank=rank, world_size=world, brute_force=brute_force, threads=threads, want_counters=True)
    return orc.resolve(acc), cnt


def to_oracle(desc):
    """a loupiote_amd.scenes description -> the oracle's numpy Scene (same arrays, same order as scenes.to_product)"""
    s = G.Scene()
    for m in desc["meshes"]:
        s.add_mesh(m["positions"], m["normals"], m["uvs"], m["indices"])
    for color, rough, metal, at, mt in desc["materials"]:
        s.add_material(color, rough, metal, at, mt)
    for img in desc["images"]:
        s.images.append(img)
    for blas, mat16, material in desc["instances"]:
        s.add_instance(blas, mat16, material)
    for i, l in enumerate(desc["lights"]):
        if i == 0:
            s.lights[0] = np.asarray(l, G.LIGHT_DT)[0]
        else:
            s.lights = np.concatenate([s.lights, np.asarray(l, G.LIGHT_DT)])
    return s
