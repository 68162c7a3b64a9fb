"""-m gpu: the HIP path (through the C ABI) against the CPU oracle on identical inputs.

Bar: integer outputs (hit ids, ray counts) exact; fp32 radiance compared bit-for-bit, with the
stated fall-back tolerance 1e-5 absolute on mean radiance if a libm/ULP difference ever shows
up (none is expected: both sides execute SPEC.md's operation order without contraction)."""
import numpy as np
import pytest

import loupiote_amd as lp
from loupiote_amd import testing as T
from oracle import harness

pytestmark = pytest.mark.gpu

TOL = 1e-5


def _oracle_scene(glb):
    from oracle import gltf_oracle as G, orc
    s = G.Scene()
    G.load_gltf(glb, s)
    s.lights[0] = T.cornell_light()[0]
    return s, orc.OracleScene.from_scene(s, probe=T.CORNELL_PROBE)


def _random_rays(n, seed, lo, hi):
    rng = np.random.default_rng(seed)
    o = rng.uniform(lo, hi, (n, 3)).astype(np.float32)
    d = rng.normal(size=(n, 3)).astype(np.float32)
    d /= np.linalg.norm(d, axis=1, keepdims=True).astype(np.float32)
    return o, d.astype(np.float32)


def test_closest_hit_matches_oracle_cornell(device, cornell_glb):
    scene = lp.Scene()
    lp.loaders.load_gltf(cornell_glb, scene)
    scene.set_light(0, T.cornell_light())
    sg = lp.SceneGPU.new_from_scene(scene, device)
    _, osc = _oracle_scene(cornell_glb)
    o, d = _random_rays(100000, 1, -2.9, 3.5)
    got = sg.trace_closest(o, d)
    want = osc.trace_closest(o, d, brute_force=True)
    assert np.array_equal(got["prim"], want["prim"])
    assert got["t"].tobytes() == want["t"].tobytes()
    assert got["u"].tobytes() == want["u"].tobytes()
    assert got["v"].tobytes() == want["v"].tobytes()
    tmax = np.full(o.shape[0], 4.0, np.float32)
    occ = sg.trace_occluded(o, d, tmax)
    assert np.array_equal(occ, osc.trace_occluded(o, d, tmax, brute_force=True))
    sg.close()


@pytest.mark.parametrize("size,bounces,frames", [(256, 4, 1), (128, 8, 3)])
def test_cornell_radiance_matches_oracle(device, cornell_glb, size, bounces, frames):
    img, counts = T.render_hip(device, cornell_glb, size, size, bounces, frames)
    ref, oc = harness.render_oracle(cornell_glb, size, size, bounces, frames)
    assert (counts.closest, counts.shadow, counts.shaded) == (oc.closest, oc.shadow, oc.shaded)
    mism = np.mean(np.any(img != ref, axis=2))
    print("bit-mismatching pixels: %.4f%%  max|err| = %g" % (100 * mism, np.max(np.abs(img - ref))))
    assert np.max(np.abs(img - ref)) <= TOL
    assert img.tobytes() == ref.tobytes()


def test_run_to_run_determinism(device, cornell_glb):
    a, _ = T.render_hip(device, cornell_glb, 192, 160, 5, 2)
    b, _ = T.render_hip(device, cornell_glb, 192, 160, 5, 2)
    assert a.tobytes() == b.tobytes()


def test_tile_shards_sum_to_full_frame(device, cornell_glb):
    full, fc = T.render_hip(device, cornell_glb, 200, 120, 4, 2)
    acc = np.zeros_like(full)
    closest = 0
    for rank in range(3):
        part, c = T.render_hip(device, cornell_glb, 200, 120, 4, 2, rank=rank, world=3)
        assert np.all(acc[part[..., 3] > 0] == 0)  # disjoint ownership
        acc += part
        closest += c.closest
    assert acc.tobytes() == full.tobytes()
    assert closest == fc.closest


def test_per_bounce_launches_and_path_kernel_agree(device, cornell_glb):
    """the per-bounce launches (k_trace: closest-hit and shadow rays in one persistent launch), the same with the two-round-trip step, and the path
    kernel (every bounce in one launch) give the same bits and the same ray counts"""
    merged, cm = T.render_hip(device, cornell_glb, 160, 96, 6, 3, options={"path_rays": 0, "coop_rays": 0})
    plain, cs = T.render_hip(device, cornell_glb, 160, 96, 6, 3, options={"path_rays": 0, "coop_rays": 0, "pipe_rays": 0})
    path, cp = T.render_hip(device, cornell_glb, 160, 96, 6, 3, options={"coop_rays": 0})
    assert merged.tobytes() == plain.tobytes() == path.tobytes()
    assert (cm.closest, cm.shadow, cm.shaded) == (cs.closest, cs.shadow, cs.shaded) == (cp.closest, cp.shadow, cp.shaded)


def test_stats_kernels_report_steps_per_ray_and_the_occluder_probe(device, cornell_glb):
    """the stats variants of the per-bounce traversal launches (lpt_renderer_enable_stats): the steps-per-ray histogram covers every ray k_trace traced, its
    maximum lies in the last non-empty bucket; the shadow rays that found an occluder and the occluder-cache probe's counts are consistent; the frame is unchanged"""
    scene = lp.Scene()
    lp.loaders.load_gltf(cornell_glb, scene)
    scene.set_light(0, T.cornell_light())
    sg = lp.SceneGPU.new_from_scene(scene, device)
    pr = lp.ProbeGPU(device, T.CORNELL_PROBE, 1, 1)
    view = T.look(T.CORNELL_EYE, T.CORNELL_DIR)
    frames = []
    for stats in (False, True):
        r = lp.Renderer(device, (160, 96))
        r.downsample_factor = 1.0
        r.resize(device, sg, pr, (160, 96))
        r.set_max_bounces(6)
        r.set_vfov(T.VFOV)
        r.set_option("path_rays", 0)          # the per-bounce launches (k_trace): the path kernel has no per-ray step counter
        r.set_option("step_budget", 0)
        r.enable_stats(stats)
        r.reset_accumulation()
        r.accumulate = True
        r.reset_ray_counts()
        r.raytrace_n(view, 3)
        frames.append(r.read_radiance())
        if stats:
            c = r.ray_counts()
            mx, hist = r.step_histogram()
            assert int(hist.sum()) == c.closest + c.shadow - c.primary     # every ray k_trace carried (bounce 0 here is traced per ray as well: primary == 0)
            last = max(k for k in range(12) if hist[k])
            assert 2 ** last <= mx < 2 ** (last + 1) and mx >= 2
            assert 0 < c.shadow_occluded < c.shadow
            assert c.occluder_cache_hits <= c.occluder_cache_found <= c.shadow and c.occluder_cache_hits <= c.shadow_occluded
            assert c.nodes > 0 and c.shadow_nodes > 0
        r.close()
    assert frames[0].tobytes() == frames[1].tobytes()
    pr.close()
    sg.close()
