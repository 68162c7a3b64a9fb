"""-m gpu: the sorted shade / next-event stage (lpt_renderer_set_sort_queues): queue order is a scheduling freedom —
radiance, ray counts and per-bounce queue sizes are bit-identical with and without it, and equal to the oracle's."""
import numpy as np
import pytest

import loupiote_amd as lp
from loupiote_amd import scenes, testing as T
from oracle import harness

pytestmark = pytest.mark.gpu


def _render(device, sg, pr, view, w, h, bounces, frames, sort, noise=None):
    r = lp.Renderer(device, (w, h))
    r.downsample_factor = 1.0
    r.resize(device, sg, pr, (w, h))
    r.set_max_bounces(bounces)
    r.set_vfov(T.VFOV)
    r.set_sort_queues(int(sort) if sort is not True else 3)
    r.set_option("path_rays", 0)      # the sorted stages belong to the per-bounce launches (k_shade); the path kernel has no queues to sort
    r.reset_accumulation()
    r.accumulate = True
    r.reset_ray_counts()
    r.raytrace_n(view, frames)
    img = r.read_radiance()
    c = r.ray_counts()
    qc, qs = r.queue_counts(bounces)
    r.close()
    return img, (c.closest, c.shadow, c.shaded), qc.tolist(), qs.tolist()


def test_sorted_queues_do_not_change_a_bit_cornell(device, cornell_glb):
    scene = lp.Scene()
    lp.loaders.load_gltf(cornell_glb, scene)
    scene.set_light(0, T.cornell_light())
    sg = lp.SceneGPU.new_from_scene(scene, device)
    pr = lp.ProbeGPU(device, T.CORNELL_PROBE, 1, 1)
    view = T.look(T.CORNELL_EYE, T.CORNELL_DIR)
    for (w, h, b, f) in [(256, 256, 4, 1), (203, 117, 8, 3), (64, 48, 17, 2)]:
        a = _render(device, sg, pr, view, w, h, b, f, False)
        for flag in (True, 4, 7):     # both outgoing queues by octant; the shading INPUT regrouped by kind inside each block; all three
            s = _render(device, sg, pr, view, w, h, b, f, flag)
            assert a[0].tobytes() == s[0].tobytes(), flag
            assert a[1:] == s[1:], flag
    ref, oc = harness.render_oracle(cornell_glb, 256, 256, 4, 1)
    s = _render(device, sg, pr, view, 256, 256, 4, 1, True)
    assert s[0].tobytes() == ref.tobytes()
    assert s[1] == (oc.closest, oc.shadow, oc.shaded)
    pr.close()
    sg.close()


def test_sorted_queues_atrium_textured(device):
    desc = scenes.synthetic_atrium(texture_size=128)
    sg = lp.SceneGPU.new_from_scene(scenes.to_product(desc), device)
    pr = lp.ProbeGPU(device, desc["probe"], desc["probe"].shape[1], desc["probe"].shape[0])
    view = T.look(desc["camera"]["origin"], desc["camera"]["direction"])
    a = _render(device, sg, pr, view, 480, 270, 8, 2, False)
    for flag in (True, 4):
        s = _render(device, sg, pr, view, 480, 270, 8, 2, flag)
        assert a[0].tobytes() == s[0].tobytes(), flag
        assert a[1:] == s[1:], flag
    pr.close()
    sg.close()


@pytest.mark.parametrize("mode", [lp.BlitMode.Pahtrace, lp.BlitMode.DenoisedPathrace])
def test_wavefront_lanes_do_not_change_a_bit(device, cornell_glb, mode):
    """lpt_renderer_set_lanes: consecutive raytrace() calls overlap on 1 / 2 / 3 / 4 lanes (own streams, own ray queues);
    accumulation stays in call order, so every intermediate and final frame is bit-identical — also when reads, batched
    calls and a resize are interleaved"""
    scene = lp.Scene()
    lp.loaders.load_gltf(cornell_glb, scene)
    scene.set_light(0, T.cornell_light())
    sg = lp.SceneGPU.new_from_scene(scene, device)
    pr = lp.ProbeGPU(device, T.CORNELL_PROBE, 1, 1)
    view = T.look(T.CORNELL_EYE, T.CORNELL_DIR)

    def run(lanes):
        r = lp.Renderer(device, (203, 117))
        r.downsample_factor = 1.0
        r.resize(device, sg, pr, (203, 117))
        r.set_max_bounces(5)
        r.set_vfov(T.VFOV)
        r.set_lanes(lanes)
        r.set_blit_mode(mode)
        r.reset_accumulation()
        r.accumulate = True
        r.reset_ray_counts()
        outs = []
        for k in range(7):
            r.raytrace(view)
            if k in (2, 5):
                outs.append(r.read_radiance())
        r.raytrace_n(view, 3)
        outs.append(r.read_radiance())
        r.resize(device, sg, pr, (96, 64))
        r.set_max_bounces(5)
        for _ in range(5):
            r.raytrace(view)
        outs.append(r.read_radiance())
        c = r.ray_counts()
        outs.append(np.array([c.closest, c.shadow, c.shaded]))
        r.close()
        return outs

    base = run(1)
    for lanes in (2, 3, 4):
        for a, b in zip(base, run(lanes)):
            assert a.tobytes() == b.tobytes(), lanes
    pr.close()
    sg.close()
