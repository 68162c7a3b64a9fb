"""-m gpu: the tail of a traversal launch finished IN PLACE (LPT_OPT_TAIL_LANES; kernels.h tail_park / tail_walk).  A wave of k_trace whose queues are
dry and that is down to `tail_lanes` live rays stops stepping them lane by lane and finishes them one after the other with all 64 lanes, from where each stands:
the lane's stack and the group in hand become the cooperative walk's node column, the triangle groups it still holds are tested first, its best hit carries over.
Which lanes meet that condition depends on the scheduling of the waves — the frame and the ray counts must not: every test compares with the oracle or with
the committed vectors, bit for bit."""
import os

import numpy as np
import pytest

import loupiote_amd as lp
from loupiote_amd import scenes, testing as T
from oracle import harness

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
PER_BOUNCE = {"path_rays": 0, "coop_rays": 0}   # coop_rays 0: the tiny-wavefront rule (a wave per ray) would take the small frames here


@pytest.fixture(scope="module")
def cornell(device, cornell_glb):
    scene = lp.Scene()
    lp.loaders.load_gltf(cornell_glb, scene)
    scene.set_light(0, T.cornell_light())
    sg = lp.SceneGPU.new_from_scene(scene, device)
    pr = lp.ProbeGPU(device, T.CORNELL_PROBE, 1, 1)
    yield scene, sg, pr
    pr.close()
    sg.close()


def _render(device, sg, pr, size, depth, frames, options, view, shard=None):
    r = lp.Renderer(device, size)
    r.downsample_factor = 1.0
    r.resize(device, sg, pr, size)
    r.set_max_bounces(depth)
    r.set_vfov(T.VFOV)
    for k, v in options.items():
        r.set_option(k, v)
    if shard:
        r.set_shard(*shard)
        r.set_resources(device, sg, pr)
    r.reset_accumulation()
    r.accumulate = True
    r.reset_ray_counts()
    r.raytrace_n(view, frames)
    img, c = r.read_radiance(), r.ray_counts()
    r.close()
    _render.wave_rays = c.wave_rays    # rays a whole wave finished (tail in place / step budget): lets a test see that the path ran
    return img, (c.closest, c.shadow, c.shaded)


@pytest.mark.parametrize("size", [(64, 64), (203, 117), (256, 136)])
def test_cornell_frames_with_every_tail_width_equal_the_oracle(device, cornell, cornell_glb, size):
    """tail widths 1..8 over both traversal steps, with and without packets for bounce 0, a refill threshold below the tail width (the host clamps), few waves per CU
    (long tails: most rays end in a cooperative walk) and many"""
    _, sg, pr = cornell
    view = T.look(T.CORNELL_EYE, T.CORNELL_DIR)
    ref, oc = harness.render_oracle(cornell_glb, size[0], size[1], 5, 4)
    variants = [dict(PER_BOUNCE, tail_lanes=t) for t in (1, 2, 3, 4, 8)]
    variants += [dict(PER_BOUNCE, tail_lanes=4, pipe_rays=0), dict(PER_BOUNCE, tail_lanes=8, pipe_rays=0, packet_primary=0), dict(PER_BOUNCE, tail_lanes=8, refill=3),
                 dict(PER_BOUNCE, tail_lanes=8, refill=63, trace_waves_per_cu=1), dict(PER_BOUNCE, tail_lanes=6, trace_waves_per_cu=32, packet_primary=1),
                 dict(PER_BOUNCE, tail_lanes=8, refill=8, trace_waves_per_cu=2, pipe_rays=0)]
    for opts in variants:
        img, counts = _render(device, sg, pr, size, 5, 4, opts, view)
        assert img.tobytes() == ref.tobytes(), opts
        assert counts == (oc.closest, oc.shadow, oc.shaded), opts


def test_the_option_round_trips_and_is_clamped(device, cornell):
    _, sg, pr = cornell
    r = lp.Renderer(device, (64, 64))
    for want, got in ((0, 0), (1, 1), (8, 8), (9, 8), (1000, 8)):
        r.set_option("tail_lanes", want)
        assert r.get_option("tail_lanes") == got
    r.close()


@pytest.mark.parametrize("tail", [1, 4, 8])
def test_atrium_frames_with_the_tail_in_place_equal_the_oracle(device, tail):
    """the 262 144-triangle scene (deep tree: long per-lane stacks to expand, long walks), depth 6, two frames; one wave per CU so that whole queues end in tails"""
    from oracle import orc
    desc = scenes.synthetic_atrium()
    osc = orc.OracleScene.from_scene(harness.to_oracle(desc), probe=desc["probe"])
    view = T.look(desc["camera"]["origin"], desc["camera"]["direction"])
    acc, oc = osc.render(160, 90, view, T.VFOV, 6, frames=2, want_counters=True)
    sg = lp.SceneGPU.new_from_scene(scenes.to_product(desc), device)
    pr = lp.ProbeGPU(device, desc["probe"], desc["probe"].shape[1], desc["probe"].shape[0])
    for extra in ({}, {"trace_waves_per_cu": 1}, {"pipe_rays": 0}):
        img, counts = _render(device, sg, pr, (160, 90), 6, 2, dict(PER_BOUNCE, tail_lanes=tail, **extra), view)
        assert counts == (oc.closest, oc.shadow, oc.shaded), extra
        assert img.tobytes() == orc.resolve(acc).tobytes(), extra
    pr.close(); sg.close()


def test_config4_at_full_size_and_its_eighth_shard_with_the_tail_in_place(device):
    """the bench frame as ONE 8.3 M-ray wavefront with the tail forced on at that size (budget_rays: the size limit the step budget and the tail share) equals the
    committed vectors; rank 3 of 8 of the same frame (1 M rays per launch: the size the tail is for) equals that rank's pixels of it, with and without the tail"""
    import hashlib
    g = np.load(os.path.join(GOLD, "full_configs.npz"))
    desc = scenes.synthetic_atrium()
    sg = lp.SceneGPU.new_from_scene(scenes.to_product(desc), device)
    pr = lp.ProbeGPU(device, desc["probe"], desc["probe"].shape[1], desc["probe"].shape[0])
    view = T.look(desc["camera"]["origin"], desc["camera"]["direction"])
    # wavefront_rays / split_rays: really ONE wavefront (the automatic cut into two pieces would switch the tail off — ADVICE r05)
    img, counts = _render(device, sg, pr, (1920, 1080), 8, 4, dict(PER_BOUNCE, tail_lanes=4, budget_rays=0x7FFFFFFF, wavefront_rays=1 << 24, split_rays=0), view)
    assert _render.wave_rays > 1000, _render.wave_rays   # waves did finish their last rays cooperatively
    assert list(counts) == g["cfg4_counts"].tolist()
    assert hashlib.sha256(np.ascontiguousarray(img).tobytes()).hexdigest() == str(g["cfg4_sha256"])
    shard = {}
    for tail in (0, 2, 8):
        shard[tail] = _render(device, sg, pr, (1920, 1080), 8, 4, dict(PER_BOUNCE, tail_lanes=tail), view, shard=(3, 8))
    assert shard[0][1] == shard[2][1] == shard[8][1]
    assert shard[0][0].tobytes() == shard[2][0].tobytes() == shard[8][0].tobytes()
    own = shard[0][0][..., 3] != 0.0      # the pixels rank 3 owns (the others stay zero in its buffer)
    assert own.any() and np.array_equal(shard[0][0][own], img[own])
    pr.close(); sg.close()


def test_random_frames_with_and_without_the_tail_are_the_same_frame(device):
    """differential: 36 random frame sizes / depths / sample counts / refill thresholds / waves per CU / tail widths / shards on the mixed-scale hall and the atrium;
    every frame equals the one the plain per-lane launches give (no tail, no step budget), bit for bit, with equal ray counts (tools/dev/r05_tail_fuzz.py runs more)"""
    rng = np.random.default_rng(17)
    base = dict(PER_BOUNCE, step_budget=0, tail_lanes=0)
    for desc in (scenes.synthetic_hall(), scenes.synthetic_atrium(texture_size=64)):
        sg = lp.SceneGPU.new_from_scene(scenes.to_product(desc), device)
        pr = lp.ProbeGPU(device, desc["probe"], desc["probe"].shape[1], desc["probe"].shape[0])
        view = T.look(desc["camera"]["origin"], desc["camera"]["direction"])
        for _ in range(18):
            size = (int(rng.integers(17, 360)), int(rng.integers(9, 220)))
            depth, spp = int(rng.integers(1, 9)), int(rng.integers(1, 6))
            opts = dict(base, tail_lanes=int(rng.integers(1, 9)), refill=int(rng.integers(0, 64)), trace_waves_per_cu=int(rng.choice([0, 1, 2, 5, 24, 32])),
                        pipe_rays=int(rng.choice([0, 0x7FFFFFFF])), packet_primary=int(rng.integers(0, 3)))
            shard = None if rng.random() < 0.6 else (int(rng.integers(0, 3)), 3)
            ref = _render(device, sg, pr, size, depth, spp, base, view, shard=shard)
            got = _render(device, sg, pr, size, depth, spp, opts, view, shard=shard)
            assert got[1] == ref[1] and got[0].tobytes() == ref[0].tobytes(), (desc["name"], size, depth, spp, shard, opts)
        pr.close(); sg.close()


def test_tail_on_the_trees_of_the_gpu_builder(device):
    """the LBVH builder's trees (Morton order, collapsed to 8-wide on the GPU; depth 8 / 11 here against the host builder's 8 / 10): other stacks to expand into the walk's
    node column, other leaves.  Frames with the tail equal the frames without it — and the host-built tree's frames: only the Woop test decides hits"""
    for desc in (scenes.synthetic_hall(), scenes.synthetic_atrium(texture_size=64)):
        view = T.look(desc["camera"]["origin"], desc["camera"]["direction"])
        pr = lp.ProbeGPU(device, desc["probe"], desc["probe"].shape[1], desc["probe"].shape[0])
        frames = {}
        for gpu_build in (False, True):
            sg = lp.SceneGPU.new_from_scene(scenes.to_product(desc), device, gpu_build=gpu_build)
            depth = sg.stats().max_depth
            for tail in (0, 1, 4, 8):
                frames[(gpu_build, tail)] = _render(device, sg, pr, (200, 120), 6, 2, dict(PER_BOUNCE, step_budget=0, tail_lanes=tail, trace_waves_per_cu=2), view)
            print("%s: %s tree depth %d" % (desc["name"], "GPU-built" if gpu_build else "host-built", depth))
            sg.close()
        ref = frames[(False, 0)]
        for key, got in frames.items():
            assert got[1] == ref[1] and got[0].tobytes() == ref[0].tobytes(), (desc["name"], key)
        pr.close()


@pytest.mark.parametrize("size,spp", [((64, 64), 1), ((203, 117), 4), ((256, 136), 3), ((320, 200), 4)])
def test_shading_grids_of_every_shape_park_and_reserve_to_the_same_frame(device, cornell, cornell_glb, size, spp):
    """round 6: k_shade reserves queue slots once per TWO iterations — every other iteration only parks its rows in LDS (kernels.h k_shade).  Small grids
    (LPT_EXP_SHADE_BLOCKS_PER_CU 1 / 2 / 3: 256 .. 768 blocks) make the blocks of these frames loop 1 .. 5 times, some of them once more than others: a lone iteration
    (nothing parked), park + reserve, and a last iteration that reserves alone behind a pair — each shape must give the oracle's frame and ray counts."""
    _, sg, pr = cornell
    view = T.look(T.CORNELL_EYE, T.CORNELL_DIR)
    ref, oc = harness.render_oracle(cornell_glb, size[0], size[1], 5, spp)
    for blocks in (0, 1, 2, 3):
        for extra in ({}, {"wavefront_rays": 40000}):      # (and the same frame cut into pieces: grids of 3 blocks per CU by default)
            img, counts = _render(device, sg, pr, size, 5, spp, dict(PER_BOUNCE, shade_blocks_per_cu=blocks, **extra), view)
            assert counts == (oc.closest, oc.shadow, oc.shaded), (blocks, extra)
            assert img.tobytes() == ref.tobytes(), (blocks, extra)
