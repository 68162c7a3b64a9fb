"""-m gpu: interactive instance edits (reference Instance::set_transform, standalone/src/lib.rs:118-121).

`SceneGPU.update_instances` re-bakes only the moved instances and refits the wide BVH on the GPU.  Hits do not
depend on the tree (SPEC §7), so the refitted scene must give exactly what a fresh upload of the edited scene —
and the oracle, which rebuilds from scratch — gives."""
import numpy as np
import pytest

import loupiote_amd as lp
from loupiote_amd import scenes, testing as T

pytestmark = [pytest.mark.gpu, pytest.mark.usefixtures("pipeline")]   # every test body over the three arms of tests/conftest.py PIPELINES: k_path, the per-bounce launches, the shipped defaults


def _trs(angle, t, s=1.0):
    c, sn = np.cos(angle), np.sin(angle)
    m = np.eye(4, dtype=np.float32)
    m[:3, :3] = np.array([[c, 0, sn], [0, 1, 0], [-sn, 0, c]], np.float32) * np.float32(s)
    m[:3, 3] = t
    return np.ascontiguousarray(m.T).reshape(-1)      # column-major


def _render(device, sg, w, h, bounces, frames):
    pr = lp.ProbeGPU(device, T.CORNELL_PROBE, 1, 1)
    r = lp.Renderer(device, (w, h))
    r.downsample_factor = 1.0
    r.resize(device, sg, pr, (w, h))
    r.set_max_bounces(bounces)
    r.set_vfov(T.VFOV)
    r.reset_accumulation()
    r.accumulate = True
    r.reset_ray_counts()
    view = T.look(T.CORNELL_EYE, T.CORNELL_DIR)
    for _ in range(frames):
        r.raytrace(view)
    img, counts = r.read_radiance(), r.ray_counts()
    r.close(); pr.close()
    return img, counts


def test_moved_instances_refit_equals_fresh_upload_and_oracle(device, cornell_glb):
    from oracle import gltf_oracle as G, orc
    scene = lp.Scene()
    lp.loaders.load_gltf(cornell_glb, scene)
    scene.set_light(0, T.cornell_light())
    sg = lp.SceneGPU.new_from_scene(scene, device)
    before, _ = _render(device, sg, 128, 128, 4, 2)
    assert sg.update_instances(scene) == 0                      # nothing moved: nothing to do
    n_inst = scene.counts().instances
    moves = {n_inst - 1: _trs(0.5, (0.8, 0.9, 0.6), 0.8), n_inst - 2: _trs(-0.3, (-0.9, 0.0, -0.5))}
    osc = G.Scene()
    G.load_gltf(cornell_glb, osc)
    osc.lights[0] = T.cornell_light()[0]
    for idx, m in moves.items():
        base = scene.instances[idx]["model_to_world"].reshape(-1).copy()
        new = (m.reshape(4, 4).T @ base.reshape(4, 4).T).T.astype(np.float32).reshape(-1)   # move in world space
        scene.set_instance_transform(idx, new)
        osc.instances[idx]["model_to_world"] = new.reshape(osc.instances[idx]["model_to_world"].shape)
    assert sg.update_instances(scene) == 2
    refit, c_refit = _render(device, sg, 128, 128, 4, 2)
    assert refit.tobytes() != before.tobytes()
    fresh_sg = lp.SceneGPU.new_from_scene(scene, device)
    fresh, c_fresh = _render(device, fresh_sg, 128, 128, 4, 2)
    assert refit.tobytes() == fresh.tobytes()
    assert (c_refit.closest, c_refit.shadow, c_refit.shaded) == (c_fresh.closest, c_fresh.shadow, c_fresh.shaded)
    sc = orc.OracleScene.from_scene(osc, probe=T.CORNELL_PROBE)
    acc, oc = sc.render(128, 128, T.look(T.CORNELL_EYE, T.CORNELL_DIR), T.VFOV, 4, frames=2, want_counters=True)
    assert refit.tobytes() == orc.resolve(acc).tobytes()
    assert (c_refit.closest, c_refit.shadow, c_refit.shaded) == (oc.closest, oc.shadow, oc.shaded)
    # random rays: closest hits of the refitted tree == those of the rebuilt tree
    rng = np.random.default_rng(5)
    o = np.zeros((20000, 4), np.float32); d = np.zeros((20000, 4), np.float32)
    o[:, :3] = rng.uniform(-2.5, 2.5, (20000, 3))
    v = rng.normal(size=(20000, 3)); d[:, :3] = v / np.linalg.norm(v, axis=1, keepdims=True)
    assert sg.trace_closest(o, d).tobytes() == fresh_sg.trace_closest(o, d).tobytes()
    # moving back restores the original picture (the refit is not cumulative)
    lp.loaders  # noqa
    scene2 = lp.Scene()
    lp.loaders.load_gltf(cornell_glb, scene2)
    scene2.set_light(0, T.cornell_light())
    for idx in moves:
        scene.set_instance_transform(idx, scene2.instances[idx]["model_to_world"].reshape(-1))
    assert sg.update_instances(scene) == 2
    again, _ = _render(device, sg, 128, 128, 4, 2)
    assert again.tobytes() == before.tobytes()
    fresh_sg.close(); sg.close()


def test_refit_rejects_a_changed_instance_list(device, cornell_glb):
    scene = lp.Scene()
    lp.loaders.load_gltf(cornell_glb, scene)
    sg = lp.SceneGPU.new_from_scene(scene, device)
    scene.add_instance(1, np.eye(4, dtype=np.float32).reshape(-1), 0)
    with pytest.raises(lp.Error) as e:
        sg.update_instances(scene)
    assert e.value.kind == "InvalidArg"
    sg.close()


def test_refit_large_scene_statue_moves(device):
    """atrium: move one ~7k-triangle instance; refit == fresh upload on 100k random rays and a 480x270 frame"""
    desc = scenes.synthetic_atrium(textures=False)
    scene = scenes.to_product(desc)
    sg = lp.SceneGPU.new_from_scene(scene, device)
    counts = scene.counts()
    idx = counts.instances - 2
    m = scene.instances[idx]["model_to_world"].reshape(-1).copy()
    m[12] += 1.5; m[13] += 0.25; m[14] -= 2.0
    scene.set_instance_transform(idx, m)
    assert sg.update_instances(scene) == 1
    fresh = lp.SceneGPU.new_from_scene(scene, device)
    rng = np.random.default_rng(9)
    n = 100000
    o = np.zeros((n, 4), np.float32); d = np.zeros((n, 4), np.float32)
    o[:, :3] = rng.uniform((-12, 0.2, -6), (12, 9, 6), (n, 3))
    v = rng.normal(size=(n, 3)); d[:, :3] = v / np.linalg.norm(v, axis=1, keepdims=True)
    a, b = sg.trace_closest(o, d), fresh.trace_closest(o, d)
    assert a.tobytes() == b.tobytes() and (a["prim"] != 0xFFFFFFFF).mean() > 0.5
    fresh.close(); sg.close()
