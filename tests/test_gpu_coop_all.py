"""-m gpu: tiny wavefronts — fewer rays than the chip has wave slots — take the per-bounce launches with EVERY ray traced by a whole wave (LPT_OPT_COOP_RAYS; k_trace_coop
over the two queues of a launch, eight lanes per node).  The order of the tests differs from the per-lane kernel's, the hits do not (only the Woop test decides, ties go to
the lower primitive id): frames and ray counts equal the oracle's bit for bit, at the sizes the rule picks by itself and at sizes it is forced on."""
import numpy as np
import pytest

import loupiote_amd as lp
from loupiote_amd import scenes, testing as T
from oracle import harness

pytestmark = pytest.mark.gpu
FORCE = {"coop_rays": 0x7FFFFFFF}


def _render(device, sg, pr, size, depth, spp, frames, options, view, shard=None, mode=None):
    r = lp.Renderer(device, size)
    r.downsample_factor = 1.0
    r.resize(device, sg, pr, size)
    r.set_max_bounces(depth)
    r.set_vfov(T.VFOV)
    for k, v in options.items():
        r.set_option(k, v)
    if mode is not None:
        r.set_blit_mode(mode)
    if shard:
        r.set_shard(*shard)
        r.set_resources(device, sg, pr)
    r.reset_accumulation()
    r.accumulate = True
    r.reset_ray_counts()
    for _ in range(frames):
        r.raytrace_n(view, spp)
    img, c = r.read_radiance(), r.ray_counts()
    r.close()
    return img, (c.closest, c.shadow, c.shaded)


def test_the_option_round_trips_and_has_its_default(device):
    r = lp.Renderer(device, (64, 64))
    assert r.get_option("coop_rays") == 32000
    for v in (0, 1, 5000, 0x7FFFFFFF):
        r.set_option("coop_rays", v)
        assert r.get_option("coop_rays") == v
    r.close()


@pytest.mark.parametrize("size,spp", [((32, 18), 4), ((64, 36), 4), ((96, 54), 1), ((77, 61), 3), ((160, 90), 2)])
def test_tiny_frames_as_shipped_equal_the_oracle(device, cornell_glb, size, spp):
    """the sizes the rule picks by itself (at most 32 000 rays per wavefront): nothing forced"""
    assert size[0] * size[1] * spp <= 32000
    img, c = T.render_hip(device, cornell_glb, size[0], size[1], 6, spp)
    ref, oc = harness.render_oracle(cornell_glb, size[0], size[1], 6, spp)
    assert (c.closest, c.shadow, c.shaded) == (oc.closest, oc.shadow, oc.shaded)
    assert img.tobytes() == ref.tobytes()


@pytest.mark.parametrize("which", ["hall", "atrium", "helmet"])
def test_a_wave_per_ray_at_every_size_equals_the_oracle(device, which):
    """forced on at sizes far above the rule's (correct at any size, just not fast there): the three stand-in scenes, with and without packets for bounce 0, sharded"""
    from oracle import orc
    desc = {"hall": scenes.synthetic_hall, "atrium": lambda: scenes.synthetic_atrium(texture_size=64), "helmet": lambda: scenes.synthetic_helmet(texture_size=64)}[which]()
    osc = orc.OracleScene.from_scene(harness.to_oracle(desc), probe=desc["probe"])
    view = T.look(desc["camera"]["origin"], desc["camera"]["direction"])
    sg = lp.SceneGPU.new_from_scene(scenes.to_product(desc), device)
    pr = lp.ProbeGPU(device, desc["probe"], desc["probe"].shape[1], desc["probe"].shape[0])
    for size, depth, frames in (((48, 27), 8, 3), ((160, 90), 6, 2), ((256, 144), 5, 1)):
        acc, oc = osc.render(size[0], size[1], view, T.VFOV, depth, frames=frames, want_counters=True)
        for extra in ({}, {"packet_primary": 1}, {"packet_primary": 0}):
            img, counts = _render(device, sg, pr, size, depth, 1, frames, dict(FORCE, **extra), view)
            assert counts == (oc.closest, oc.shadow, oc.shaded), (size, extra)
            assert img.tobytes() == orc.resolve(acc).tobytes(), (size, extra)
    whole, counts = _render(device, sg, pr, (160, 96), 6, 2, 2, FORCE, view)
    acc = np.zeros_like(whole)
    tot = np.zeros(3, np.int64)
    for rank in range(3):
        part, c = _render(device, sg, pr, (160, 96), 6, 2, 2, FORCE, view, shard=(rank, 3))
        acc += part
        tot += np.asarray(c, np.int64)
    assert acc.tobytes() == whole.tobytes() and tuple(int(x) for x in tot) == counts
    pr.close(); sg.close()


def test_the_denoising_modes_on_a_tiny_frame_are_the_per_lane_frames(device):
    """G-buffer / motion vectors come from bounce 0's shading pass, whichever kernel traced it: temporal and a-trous outputs equal those of the per-lane launches"""
    desc = scenes.synthetic_atrium(texture_size=64)
    view = T.look(desc["camera"]["origin"], desc["camera"]["direction"])
    sg = lp.SceneGPU.new_from_scene(scenes.to_product(desc), device)
    pr = lp.ProbeGPU(device, desc["probe"], desc["probe"].shape[1], desc["probe"].shape[0])
    for mode in (lp.BlitMode.Temporal, lp.BlitMode.DenoisedPathrace):
        a = _render(device, sg, pr, (96, 56), 5, 1, 3, {"coop_rays": 0, "path_rays": 0}, view, mode=mode)
        b = _render(device, sg, pr, (96, 56), 5, 1, 3, {}, view, mode=mode)
        assert a[1] == b[1] and a[0].tobytes() == b[0].tobytes(), mode
    pr.close(); sg.close()
