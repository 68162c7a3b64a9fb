"""-m gpu: HIP path vs the CPU oracle on the Sponza STAND-IN (synthetic_atrium: 262,144 baked
triangles, ~100 instances, textured materials, RGBE sky probe) — the scene bench.py measures."""
import numpy as np
import pytest

import loupiote_amd as lp
from loupiote_amd import scenes, testing as T
from oracle import harness

pytestmark = [pytest.mark.gpu, pytest.mark.usefixtures("pipeline")]   # every test body over the three arms of tests/conftest.py PIPELINES: k_path, the per-bounce launches, the shipped defaults
TOL = 1e-5


@pytest.fixture(scope="module")
def atrium():
    from oracle import orc
    desc = scenes.synthetic_atrium()
    osc = orc.OracleScene.from_scene(harness.to_oracle(desc), probe=desc["probe"])
    return desc, osc


def render_desc(device, desc, w, h, bounces, frames, rank=0, world=1, noise=None, options=None):
    scene = scenes.to_product(desc)
    sg = lp.SceneGPU.new_from_scene(scene, device)
    pr = lp.ProbeGPU(device, desc["probe"], desc["probe"].shape[1], desc["probe"].shape[0])
    r = lp.Renderer(device, (w, h))
    r.downsample_factor = 1.0
    if noise is not None:
        r.upload_noise_texture(noise, noise.shape[1], noise.shape[0], noise.shape[1] * 4)
        r.use_noise_texture(True)
    r.resize(device, sg, pr, (w, h))
    r.set_max_bounces(bounces)
    r.set_vfov(T.VFOV)
    for k, v in (options or {}).items():
        r.set_option(k, v)
    if world > 1:
        r.set_shard(rank, world)
        r.set_resources(device, sg, pr)
    view = T.look(desc["camera"]["origin"], desc["camera"]["direction"])
    r.reset_accumulation()
    r.accumulate = True
    r.reset_ray_counts()
    for _ in range(frames):
        r.raytrace(view)
    img, counts = r.read_radiance(), r.ray_counts()
    srgb = r.read_pixels()
    r.close()
    pr.close()
    sg.close()
    return img, counts, srgb


def _rays(n, seed):
    rng = np.random.default_rng(seed)
    o = rng.uniform(-5.5, 5.5, (n, 3)).astype(np.float32)
    o[:, 0] *= 2.4
    o[:, 1] = np.abs(o[:, 1]) * 1.9
    d = rng.normal(size=(n, 3)).astype(np.float32)
    d /= np.linalg.norm(d, axis=1, keepdims=True).astype(np.float32)
    return o, d.astype(np.float32)


def test_atrium_closest_hit_matches_oracle(device, atrium):
    """product BVH (binned SAH, Node2) vs the oracle's own median-split BVH: same hits, bit for bit"""
    desc, osc = atrium
    sg = lp.SceneGPU.new_from_scene(scenes.to_product(desc), device)
    assert sg.stats().triangles == 262144
    o, d = _rays(200000, 7)
    got = sg.trace_closest(o, d)
    want = osc.trace_closest(o, d)
    assert np.array_equal(got["prim"], want["prim"])
    for k in ("t", "u", "v"):
        assert got[k].tobytes() == want[k].tobytes()
    tmax = np.full(o.shape[0], 6.0, np.float32)
    assert np.array_equal(sg.trace_occluded(o, d, tmax), osc.trace_occluded(o, d, tmax))
    sg.close()


def test_atrium_radiance_matches_oracle(device, atrium):
    from oracle import orc
    desc, osc = atrium
    w, h, bounces, frames = 320, 180, 8, 2
    img, counts, srgb = render_desc(device, desc, w, h, bounces, frames)
    view = T.look(desc["camera"]["origin"], desc["camera"]["direction"])
    acc, oc = osc.render(w, h, view, T.VFOV, bounces, frames=frames, want_counters=True)
    ref = orc.resolve(acc)
    assert (counts.closest, counts.shadow, counts.shaded) == (oc.closest, oc.shadow, oc.shaded)
    print("max|err| = %g, mismatching pixels %.4f%%" % (np.max(np.abs(img - ref)), 100 * np.mean(np.any(img != ref, axis=2))))
    assert np.max(np.abs(img - ref)) <= TOL * max(1.0, float(ref.max()))
    assert img.tobytes() == ref.tobytes()
    # read_pixels (sRGB8): exact (threshold-table OETF, SPEC §13.2)
    want8 = orc.tonemap(acc)
    assert srgb.tobytes() == want8.tobytes()


def test_atrium_full_size_properties(device, atrium):
    """BASELINE size (1920x1080, 4 spp, depth 8): size-independent properties instead of the oracle —
    determinism, finiteness, sample count, ray-count bounds, and tile shards summing to the frame."""
    desc, _ = atrium
    w, h, bounces, frames = 1920, 1080, 8, 4
    a, ca, _ = render_desc(device, desc, w, h, bounces, frames)
    b, cb, _ = render_desc(device, desc, w, h, bounces, frames)
    assert a.tobytes() == b.tobytes()
    assert (ca.closest, ca.shadow) == (cb.closest, cb.shadow)
    assert np.all(np.isfinite(a)) and np.all(a[..., :3] >= 0) and np.all(a[..., 3] == 1.0)
    assert w * h * frames <= ca.closest <= w * h * frames * bounces
    assert ca.shadow <= ca.shaded <= ca.closest
    acc = np.zeros_like(a)
    for rank in range(2):
        part, _, _ = render_desc(device, desc, w, h, bounces, frames, rank=rank, world=2)
        acc += part
    assert acc.tobytes() == a.tobytes()


def test_blue_noise_mode_matches_oracle(device, atrium):
    from oracle import orc
    desc, _ = atrium
    noise = np.random.default_rng(5).integers(0, 256, (64, 64, 4), dtype=np.uint8)
    osc = orc.OracleScene.from_scene(harness.to_oracle(desc), probe=desc["probe"], noise=noise)
    w, h, bounces, frames = 160, 96, 4, 2
    img, counts, _ = render_desc(device, desc, w, h, bounces, frames, noise=noise)
    view = T.look(desc["camera"]["origin"], desc["camera"]["direction"])
    acc, oc = osc.render(w, h, view, T.VFOV, bounces, frames=frames, use_noise=True, want_counters=True)
    assert (counts.closest, counts.shadow) == (oc.closest, oc.shadow)
    assert img.tobytes() == orc.resolve(acc).tobytes()
    plain, _, _ = render_desc(device, desc, w, h, bounces, frames)
    assert plain.tobytes() != img.tobytes()


def test_raytrace_n_equals_n_sequential_calls(device, atrium):
    """lpt_renderer_raytrace_n(view, n) is bit-identical to n x { raytrace(view); accumulate = true }, from a
    reset state and when continuing an accumulation, and leaves the same frame_count / seed behind."""
    desc, _ = atrium
    sg = lp.SceneGPU.new_from_scene(scenes.to_product(desc), device)
    pr = lp.ProbeGPU(device, desc["probe"], desc["probe"].shape[1], desc["probe"].shape[0])
    view = T.look(desc["camera"]["origin"], desc["camera"]["direction"])
    out = []
    for batched in (False, True):
        r = lp.Renderer(device, (200, 120))      # 120 rows: not a whole number of 8-row tiles
        r.downsample_factor = 1.0
        r.resize(device, sg, pr, (200, 120))
        r.set_max_bounces(5)
        r.set_vfov(T.VFOV)
        r.reset_accumulation()                   # accumulate == false: the first call overwrites, no increment
        if batched:
            r.raytrace_n(view, 3)
            r.raytrace_n(view, 2)
        else:
            for _ in range(5):
                r.raytrace(view)
                r.accumulate = True              # app.rs:318
        out.append((r.read_radiance(), r.frame_state(), r.ray_counts().closest, r.accumulate))
        r.close()
    pr.close()
    sg.close()
    assert out[0][0].tobytes() == out[1][0].tobytes()
    assert out[0][1:] == out[1][1:]
    assert np.all(out[0][0][..., 3] == 1.0)


@pytest.mark.parametrize("budget", [1, 9, 24])
def test_step_budget_and_the_cooperative_kernel_give_the_same_frame(device, atrium, budget):
    """per-bounce launches with a step budget (LPT_EXP_STEP_BUDGET): a ray that is not finished after `budget` traversal steps is dropped by the per-lane kernel
    and traced again by a whole wave (k_trace_coop: eight lanes per node, up to eight pending nodes per round).  budget 1 sends every ray of bounces 1.. that way,
    9 about half of them, 24 the long ones — the frame and the ray counts are the oracle's on the 262 144-triangle scene"""
    desc, osc = atrium
    from oracle import orc
    view = T.look(desc["camera"]["origin"], desc["camera"]["direction"])
    acc, oc = osc.render(160, 90, view, T.VFOV, 6, frames=2, want_counters=True)
    img, c, _ = render_desc(device, desc, 160, 90, 6, 2, options={"path_rays": 0, "step_budget": budget, "tail_lanes": 0, "coop_rays": 0})
    assert (c.closest, c.shadow, c.shaded) == (oc.closest, oc.shadow, oc.shaded)
    assert img.tobytes() == orc.resolve(acc).tobytes()
