"""-m gpu: HIP path vs the CPU oracle on a MIXED-SCALE scene (scenes.synthetic_hall: walls of two triangles each, 12 000 centimetre-sized triangles in one corner,
slivers through the whole volume, 120 coincident triangles, nested triangles) — what the uniformly tessellated stand-ins of the bench do not exercise: leaves that
overlap at very different scales, equal hit distances (the tie goes to the lowest primitive id), long thin boxes."""
import numpy as np
import pytest

import loupiote_amd as lp
from loupiote_amd import scenes, testing as T
from oracle import harness

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def hall():
    from oracle import orc
    desc = scenes.synthetic_hall()
    return desc, orc.OracleScene.from_scene(harness.to_oracle(desc), probe=desc["probe"])


def _render(device, desc, size, depth, frames, options=None, shard=None):
    sg = lp.SceneGPU.new_from_scene(scenes.to_product(desc), device)
    pr = lp.ProbeGPU(device, desc["probe"], desc["probe"].shape[1], desc["probe"].shape[0])
    r = lp.Renderer(device, size)
    r.downsample_factor = 1.0
    r.resize(device, sg, pr, size)
    r.set_max_bounces(depth)
    r.set_vfov(T.VFOV)
    for k, v in (options or {}).items():
        r.set_option(k, v)
    if shard:
        r.set_shard(*shard)
        r.set_resources(device, sg, pr)
    r.reset_accumulation()
    r.accumulate = True
    r.reset_ray_counts()
    view = T.look(desc["camera"]["origin"], desc["camera"]["direction"])
    for _ in range(frames):
        r.raytrace(view)
    img, c = r.read_radiance(), r.ray_counts()
    r.close(); pr.close(); sg.close()
    return img, (c.closest, c.shadow, c.shaded)


@pytest.mark.usefixtures("pipeline")   # over the arms of tests/conftest.py PIPELINES
@pytest.mark.parametrize("size,depth,frames", [((160, 90), 6, 2), ((384, 216), 8, 1)])
def test_hall_frames_equal_the_oracle(device, hall, size, depth, frames):
    from oracle import orc
    desc, osc = hall
    view = T.look(desc["camera"]["origin"], desc["camera"]["direction"])
    acc, oc = osc.render(size[0], size[1], view, T.VFOV, depth, frames=frames, want_counters=True)
    img, counts = _render(device, desc, size, depth, frames)
    assert counts == (oc.closest, oc.shadow, oc.shaded)
    assert img.tobytes() == orc.resolve(acc).tobytes()


@pytest.mark.usefixtures("pipeline")
def test_hall_shards_add_up_to_the_whole_frame(device, hall):
    desc, _ = hall
    whole, counts = _render(device, desc, (320, 184), 6, 2)
    acc = np.zeros_like(whole)
    tot = np.zeros(3, np.int64)
    for rank in range(3):
        part, c = _render(device, desc, (320, 184), 6, 2, shard=(rank, 3))
        acc += part
        tot += np.asarray(c, np.int64)
    assert acc.tobytes() == whole.tobytes() and tuple(int(x) for x in tot) == counts


def test_hall_rays_from_inside_the_gravel_and_along_the_slivers(device, hall):
    """closest-hit and any-hit queries (lpt_trace_closest / lpt_trace_occluded) against the oracle's brute force: origins inside the pile of tiny triangles, directions along the walls"""
    desc, osc = hall
    rng = np.random.default_rng(11)
    n = 6000
    o = np.concatenate([rng.uniform((-7.5, 0.05, -7.5), (-4.5, 1.5, -4.5), (n // 2, 3)), rng.uniform((-7.9, 0.1, -7.9), (7.9, 8.9, 7.9), (n - n // 2, 3))]).astype(np.float32)
    d = rng.normal(size=(n, 3)).astype(np.float32)
    d[: n // 4, 1] *= 0.02                                   # grazing along the floor
    d /= np.linalg.norm(d, axis=1, keepdims=True).astype(np.float32)
    sg = lp.SceneGPU.new_from_scene(scenes.to_product(desc), device)
    got, want = sg.trace_closest(o, d), osc.trace_closest(o, d, brute_force=True)
    assert np.array_equal(got["prim"], want["prim"]) and (got["prim"] != 0xFFFFFFFF).mean() > 0.9
    for k in ("t", "u", "v"):
        assert got[k].tobytes() == want[k].tobytes()
    tmax = rng.uniform(0.05, 12.0, n).astype(np.float32)
    assert np.array_equal(sg.trace_occluded(o, d, tmax), osc.trace_occluded(o, d, tmax, brute_force=True))
    sg.close()


def test_grazing_rays_at_a_scale_ratio_of_a_million_match_brute_force(device):
    """SPEC §7's margin on the GPU (scenes.origin_dust): 3 000 millimetre-sized triangles around the world origin in a scene 2 000 units across; rays from up to a thousand
    units away aimed at vertices and edge points of random triangles (where a box test that is not conservative against the Woop test's rounding would lose the hit):
    lpt_trace_closest / lpt_trace_occluded equal the oracle's BRUTE FORCE bit for bit (the CPU twin of this test walks the same tree: tests/test_bvh_builder.py)"""
    from oracle import orc
    desc = scenes.origin_dust()
    osc = orc.OracleScene.from_scene(harness.to_oracle(desc), probe=desc["probe"])
    for gpu_build in (False, True):          # the host builder's tree and the GPU builder's (LBVH + refit: the same padding rule on the device)
        sg = lp.SceneGPU.new_from_scene(scenes.to_product(desc), device, gpu_build=gpu_build)
        for seed in (5, 6):
            o, d, dist = scenes.grazing_rays(desc["dust"], 60000, seed=seed)
            got, want = sg.trace_closest(o, d), osc.trace_closest(o, d, brute_force=True)
            assert (want["prim"] != 0xFFFFFFFF).mean() > 0.3
            assert np.array_equal(got["prim"], want["prim"]), (gpu_build, seed, int((got["prim"] != want["prim"]).sum()))
            for k in ("t", "u", "v"):
                assert got[k].tobytes() == want[k].tobytes()
            tmax = (dist * np.random.default_rng(6).uniform(0.5, 1.5, dist.shape[0])).astype(np.float32)
            assert np.array_equal(sg.trace_occluded(o, d, tmax), osc.trace_occluded(o, d, tmax, brute_force=True))
        sg.close()


def test_split_triangles_survive_an_instance_edit(device, hall):
    """round 6: the host builder splits the hall's slivers (bvh.cpp presplit: a triangle then has several places in the tree).  Moving the instance they belong to
    re-bakes them on the device and a PLACE-driven scatter takes the new Woop maps to every place; the refit recomputes the leaf boxes from whole triangles.  The
    edited scene answers closest-hit / any-hit queries exactly like a fresh upload of the edited scene and like the oracle's brute force."""
    from oracle import orc
    desc, _ = hall
    scene = scenes.to_product(desc)
    sg = lp.SceneGPU.new_from_scene(scene, device)
    assert sg.stats().triangles == 12382
    moved = dict(desc)
    m = np.eye(4, dtype=np.float32)
    m[:3, 3] = (0.35, 0.2, -0.45)
    m16 = np.ascontiguousarray(m.T).reshape(-1)
    inst = list(desc["instances"])
    inst[2] = (inst[2][0], m16, inst[2][2])            # the slivers' instance (scenes.synthetic_hall: walls, gravel, NEEDLES, stack, telescope)
    moved["instances"] = inst
    scene.set_instance_transform(3, m16)                # (instance 0 is the reference's dummy)
    assert sg.update_instances(scene) == 1
    fresh = lp.SceneGPU.new_from_scene(scenes.to_product(moved), device)
    osc = orc.OracleScene.from_scene(harness.to_oracle(moved), probe=desc["probe"])
    rng = np.random.default_rng(5)
    n = 20000
    o = rng.uniform((-7.9, 0.1, -7.9), (7.9, 8.9, 7.9), (n, 3)).astype(np.float32)
    d = rng.normal(size=(n, 3)).astype(np.float32)
    d /= np.linalg.norm(d, axis=1, keepdims=True).astype(np.float32)
    a, b, want = sg.trace_closest(o, d), fresh.trace_closest(o, d), osc.trace_closest(o, d, brute_force=True)
    for k in ("prim", "t", "u", "v"):
        assert a[k].tobytes() == b[k].tobytes() == want[k].tobytes(), k
    tmax = rng.uniform(0.05, 12.0, n).astype(np.float32)
    assert np.array_equal(sg.trace_occluded(o, d, tmax), osc.trace_occluded(o, d, tmax, brute_force=True))
    fresh.close(); sg.close()


def test_a_speck_that_flies_in_from_far_away_keeps_every_hit(device):
    """round 6: the node origins live on a SCENE GRID (common.h scene_grid) that every refit re-chooses from the triangles' bounds, while the scene-wide part of the
    triangle padding never shrinks (device.hip scene_bounds_and_grid).  A decimetre of dust 40 000 units from the origin, then moved TO the origin: its padding (from the
    old place, 0.08) is now wider than the dust itself, and the grid must still reach below every padded node box — the bounds are widened by the padding before the grid is
    chosen.  Both builders; the moved scene answers like a fresh upload of it and like the oracle's brute force."""
    from oracle import orc
    dust = scenes.origin_dust()["dust"]
    base = scenes.origin_dust()
    mesh = scenes._mesh(dust.reshape(-1, 3), np.arange(dust.shape[0] * 3, dtype=np.uint32))
    far = dict(base, meshes=[mesh], instances=[(1, scenes._translate(40000.0, 0.0, 0.0), 1)], triangles=int(dust.shape[0]))
    near = dict(far, instances=[(1, scenes._translate(0.0, 0.0, 0.0), 1)])
    osc = orc.OracleScene.from_scene(harness.to_oracle(near), probe=base["probe"])
    o, d, dist = scenes.grazing_rays(dust, 40000, seed=11, extent=2.0)
    want = osc.trace_closest(o, d, brute_force=True)
    assert (want["prim"] != 0xFFFFFFFF).sum() > 10000
    tmax = (dist * np.random.default_rng(6).uniform(0.5, 1.5, dist.shape[0])).astype(np.float32)
    want_occ = osc.trace_occluded(o, d, tmax, brute_force=True)
    for gpu_build in (False, True):
        scene = scenes.to_product(far)
        sg = lp.SceneGPU.new_from_scene(scene, device, gpu_build=gpu_build)
        scene.set_instance_transform(1, scenes._translate(0.0, 0.0, 0.0))     # (instance 0 is the reference's dummy)
        assert sg.update_instances(scene) == 1
        got = sg.trace_closest(o, d)
        for k in ("prim", "t", "u", "v"):
            assert got[k].tobytes() == want[k].tobytes(), (gpu_build, k, int((got["prim"] != want["prim"]).sum()))
        assert np.array_equal(sg.trace_occluded(o, d, tmax), want_occ), gpu_build
        sg.close()
