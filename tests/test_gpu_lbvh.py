"""-m gpu: the GPU BVH builder (Morton radix tree -> 8-wide collapse on the device).

Hits do not depend on the tree (SPEC §7), so a scene built on the GPU must trace and render exactly like the same
scene built by the host SAH builder — and like the oracle; the refit must work on a GPU-built tree as well."""
import numpy as np
import pytest

import loupiote_amd as lp
from loupiote_amd import scenes, testing as T
from oracle import harness

pytestmark = [pytest.mark.gpu, pytest.mark.usefixtures("pipeline")]   # every test body over the three arms of tests/conftest.py PIPELINES: k_path, the per-bounce launches, the shipped defaults


def _rays(n, lo, hi, seed):
    rng = np.random.default_rng(seed)
    o = np.zeros((n, 4), np.float32); d = np.zeros((n, 4), np.float32)
    o[:, :3] = rng.uniform(lo, hi, (n, 3))
    v = rng.normal(size=(n, 3)); d[:, :3] = v / np.linalg.norm(v, axis=1, keepdims=True)
    return o, d


def test_gpu_built_atrium_traces_like_the_host_built_one(device):
    desc = scenes.synthetic_atrium(textures=False)
    scene = scenes.to_product(desc)
    host = lp.SceneGPU.new_from_scene(scene, device)
    gpu = lp.SceneGPU.new_from_scene(scene, device, gpu_build=True)
    hs, gs = host.stats(), gpu.stats()
    assert gs.triangles == hs.triangles == 262144 and 0 < gs.nodes < 262144 and 4 <= gs.max_depth <= 40
    print("host: %d nodes depth %d build %.1f ms upload %.1f ms | gpu: %d nodes depth %d build %.1f ms upload %.1f ms"
          % (hs.nodes, hs.max_depth, hs.build_ms, hs.upload_ms, gs.nodes, gs.max_depth, gs.build_ms, gs.upload_ms))
    # the GPU build never touches the host baker: instances are baked by k_bake_instance on the device
    assert gs.host_baked_triangles == 0 and hs.host_baked_triangles == 262144
    assert 0 < gs.upload_ms and gs.build_ms <= gs.upload_ms
    o, d = _rays(200000, (-12, 0.2, -6), (12, 9, 6), 3)
    a, b = host.trace_closest(o, d), gpu.trace_closest(o, d)
    assert a.tobytes() == b.tobytes() and (a["prim"] != 0xFFFFFFFF).mean() > 0.5
    tmax = np.full(o.shape[0], 7.5, np.float32)
    assert host.trace_occluded(o, d, tmax).tobytes() == gpu.trace_occluded(o, d, tmax).tobytes()
    # a refit on the GPU-built tree
    idx = scene.counts().instances - 2
    m = scene.instances[idx]["model_to_world"].reshape(-1).copy()
    m[12] += 1.0; m[14] -= 1.5
    scene.set_instance_transform(idx, m)
    assert gpu.update_instances(scene) == 1 and host.update_instances(scene) == 1
    assert host.trace_closest(o, d).tobytes() == gpu.trace_closest(o, d).tobytes()
    host.close(); gpu.close()


def test_gpu_built_scenes_render_like_the_oracle(device, cornell_glb):
    # Cornell (34 triangles): bit-exact vs the oracle
    scene = lp.Scene()
    lp.loaders.load_gltf(cornell_glb, scene)
    scene.set_light(0, T.cornell_light())
    sg = lp.SceneGPU.new_from_scene(scene, device, gpu_build=True)
    pr = lp.ProbeGPU(device, T.CORNELL_PROBE, 1, 1)
    r = lp.Renderer(device, (128, 128))
    r.downsample_factor = 1.0
    r.resize(device, sg, pr, (128, 128))
    r.set_max_bounces(4)
    r.set_vfov(T.VFOV)
    r.reset_accumulation(); r.accumulate = True; r.reset_ray_counts()
    view = T.look(T.CORNELL_EYE, T.CORNELL_DIR)
    r.raytrace(view); r.raytrace(view)
    img, c = r.read_radiance(), r.ray_counts()
    ref, oc = harness.render_oracle(cornell_glb, 128, 128, 4, 2)
    assert img.tobytes() == ref.tobytes() and (c.closest, c.shadow, c.shaded) == (oc.closest, oc.shadow, oc.shaded)
    r.close(); pr.close(); sg.close()


def test_gpu_built_helmet_frame_equals_host_built(device):
    desc = scenes.synthetic_helmet()
    imgs = []
    for gpu_build in (False, True):
        sg = lp.SceneGPU.new_from_scene(scenes.to_product(desc), device, gpu_build=gpu_build)
        pr = lp.ProbeGPU(device, desc["probe"], desc["probe"].shape[1], desc["probe"].shape[0])
        r = lp.Renderer(device, (480, 270))
        r.downsample_factor = 1.0
        r.resize(device, sg, pr, (480, 270))
        r.set_max_bounces(6)
        r.set_vfov(T.VFOV)
        r.reset_accumulation(); r.accumulate = True
        r.raytrace_n(T.look(desc["camera"]["origin"], desc["camera"]["direction"]), 4)
        imgs.append(r.read_radiance())
        r.close(); pr.close(); sg.close()
    assert imgs[0].tobytes() == imgs[1].tobytes()


@pytest.mark.parametrize("kind", ["coincident", "coplanar", "n16", "n17", "random", "clusters"])
def test_gpu_builder_degenerate_soups(device, kind):
    """duplicate Morton codes, zero-extent axes and the smallest sizes the radix tree handles"""
    rng = np.random.default_rng(21)
    if kind == "coincident":
        tri = np.repeat(rng.normal(0, 1, (1, 3, 3)), 300, axis=0)
    elif kind == "coplanar":
        tri = rng.uniform(-4, 4, (500, 1, 3)) + rng.normal(0, 0.3, (500, 3, 3)); tri[..., 1] = 0.5
    elif kind in ("n16", "n17"):
        n = int(kind[1:]); tri = rng.uniform(-2, 2, (n, 1, 3)) + rng.normal(0, 0.4, (n, 3, 3))
    elif kind == "random":
        tri = rng.uniform(-5, 5, (5000, 1, 3)) + rng.normal(0, 0.25, (5000, 3, 3))
    else:
        c = rng.uniform(-50, 50, (8, 3))
        tri = c[rng.integers(0, 8, 4000)][:, None, :] + rng.normal(0, 0.02, (4000, 3, 3))
    pos = np.zeros((tri.shape[0] * 3, 4), np.float32); pos[:, :3] = tri.reshape(-1, 3)
    scene = lp.Scene()
    b = scene.add_mesh(pos, None, None, None)
    scene.add_instance(b, np.eye(4, dtype=np.float32).reshape(-1), 0)
    host = lp.SceneGPU.new_from_scene(scene, device)
    gpu = lp.SceneGPU.new_from_scene(scene, device, gpu_build=True)
    assert gpu.stats().triangles == tri.shape[0] and gpu.stats().nodes >= 1
    lo, hi = tri.min() - 1, tri.max() + 1
    o, d = _rays(50000, (lo, lo, lo), (hi, hi, hi), 5)
    # half of the rays aim at triangle centroids so that there are hits even in sparse soups
    cen = tri.mean(axis=1)[rng.integers(0, tri.shape[0], 25000)]
    v = cen - o[:25000, :3]; d[:25000, :3] = v / np.maximum(np.linalg.norm(v, axis=1, keepdims=True), 1e-9)
    a, bq = host.trace_closest(o, d), gpu.trace_closest(o, d)
    assert a.tobytes() == bq.tobytes() and (a["prim"] != 0xFFFFFFFF).any()
    host.close(); gpu.close()


def test_rebuild_after_large_edits_equals_fresh_upload(device):
    """lpt_scene_gpu_rebuild: all instances re-baked on the device + a new GPU-built tree, in place"""
    import time
    desc = scenes.synthetic_atrium(textures=False)
    scene = scenes.to_product(desc)
    sg = lp.SceneGPU.new_from_scene(scene, device)            # host-built to begin with
    inst = scene.instances
    rng = np.random.default_rng(4)
    for idx in range(1, len(inst), 3):                         # move every third instance, some of them far
        m = inst[idx]["model_to_world"].reshape(-1).copy()
        m[12:15] += rng.uniform(-3, 3, 3).astype(np.float32)
        scene.set_instance_transform(idx, m)
    t0 = time.perf_counter(); sg.rebuild(scene); dt = time.perf_counter() - t0
    fresh = lp.SceneGPU.new_from_scene(scene, device)
    print("rebuild %.2f ms (nodes %d, depth %d); fresh host upload builds in %.1f ms" % (dt * 1e3, sg.stats().nodes, sg.stats().max_depth, fresh.stats().build_ms))
    o, d = _rays(150000, (-14, 0.2, -8), (14, 9, 8), 8)
    a, b = sg.trace_closest(o, d), fresh.trace_closest(o, d)
    assert a.tobytes() == b.tobytes() and (a["prim"] != 0xFFFFFFFF).mean() > 0.4
    # a bound renderer keeps working across the rebuild and sees the new scene
    pr = lp.ProbeGPU(device, desc["probe"], desc["probe"].shape[1], desc["probe"].shape[0])
    view = T.look(desc["camera"]["origin"], desc["camera"]["direction"])
    imgs = []
    for g in (sg, fresh):
        r = lp.Renderer(device, (320, 180))
        r.downsample_factor = 1.0
        r.resize(device, g, pr, (320, 180))
        r.set_max_bounces(5); r.set_vfov(T.VFOV)
        r.reset_accumulation(); r.accumulate = True
        r.raytrace_n(view, 2)
        imgs.append(r.read_radiance())
        if g is sg:                                            # edit again under the live renderer
            sg.rebuild(scene)
            r.reset_accumulation(); r.accumulate = True; r.set_seed(0)
        r.close()
    assert imgs[0].tobytes() == imgs[1].tobytes()
    assert sg.update_instances(scene) == 0                     # the rebuild recorded the new transforms
    pr.close(); fresh.close(); sg.close()
