"""-m gpu: record-then-submit at the boundary (lpt_renderer_raytrace records, the next submission point launches; reference:
the passes of a frame go into one encoder, renderer.rs:392-549, submitted once, app.rs:335-337).  Consecutive recorded calls
that continue one accumulation from one view are traced as ONE wavefront — every frame below must equal, bit for bit, what the
same calls give when each launches at once (lpt_renderer_set_max_fused(1), the round-2 behaviour), what raytrace_n gives, and
the oracle."""
import numpy as np
import pytest

import loupiote_amd as lp
from loupiote_amd import scenes, testing as T
from oracle import harness

pytestmark = [pytest.mark.gpu, pytest.mark.usefixtures("pipeline")]   # every test body over the three arms of tests/conftest.py PIPELINES: k_path, the per-bounce launches, the shipped defaults
W, H, DEPTH = 203, 117, 5


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


def _renderer(device, sg, pr, max_fused, lanes=None, size=(W, H), options=None):
    r = lp.Renderer(device, size)
    r.downsample_factor = 1.0
    r.resize(device, sg, pr, size)
    r.set_max_bounces(DEPTH)
    r.set_vfov(T.VFOV)
    r.set_max_fused(max_fused)
    for k, v in (options or {}).items():
        r.set_option(k, v)
    if lanes:
        r.set_lanes(lanes)
    r.reset_accumulation()
    r.accumulate = True
    r.reset_ray_counts()
    return r


def _state(r):
    c = r.ray_counts()
    return (r.frame_state(), r.accumulate, (c.closest, c.shadow, c.shaded))


def test_four_calls_equal_one_batch_equal_eager_equal_oracle(device, cornell, cornell_glb):
    _, sg, pr = cornell
    view = T.look(T.CORNELL_EYE, T.CORNELL_DIR)
    outs = []
    for how in ("deferred", "eager", "batch", "deferred+submit"):
        r = _renderer(device, sg, pr, 1 if how == "eager" else 0)
        if how == "batch":
            r.raytrace_n(view, 4)
        else:
            for k in range(4):
                r.raytrace(view)
                r.accumulate = True          # app.rs:318
                if how == "deferred+submit" and k == 1:
                    r.submit()               # a submission in the middle of the frame splits the wavefront, nothing else
        outs.append((r.read_radiance(), _state(r)))
        r.close()
    ref, oc = harness.render_oracle(cornell_glb, W, H, DEPTH, 4)
    for img, st in outs:
        assert img.tobytes() == ref.tobytes()
        assert st == outs[0][1]
        assert st[2] == (oc.closest, oc.shadow, oc.shaded)


def _script(r, views, sg, scene):
    """a caller that changes the view in the middle of a frame, resets, toggles accumulate, edits a setter and the scene"""
    v0, v1 = views
    outs = []
    for v in (v0, v0, v1, v1, v1):               # view change mid-frame: the first two samples stay on the old view
        r.raytrace(v)
    outs.append(r.read_radiance())
    outs.append(_state(r))
    r.raytrace(v1); r.raytrace(v1)
    r.reset_accumulation()                       # frame_count = 1, accumulate = false: the two calls above are dropped from the mean
    r.raytrace(v0)                               # accumulate == false: overwrites
    r.raytrace(v0)                               # still false: overwrites again
    r.accumulate = True
    r.raytrace(v0); r.raytrace(v0); r.raytrace(v0)
    outs.append(_state(r))                       # reading the counters is a submission point
    r.raytrace(v0)
    r.set_max_bounces(2)                         # a setter the passes read: what is recorded keeps the old depth
    r.raytrace(v0); r.raytrace(v0)
    r.set_max_bounces(DEPTH)
    outs.append(r.read_radiance())
    outs.append(r.read_pixels())
    # scene edit between recorded calls: the first call must see the scene as it was
    r.reset_accumulation(); r.accumulate = True
    r.raytrace(v0)
    m = np.eye(4, dtype=np.float32); m[3, :3] = (0.4, 0.2, -0.3)
    scene.set_instance_transform(2, m.reshape(16))
    sg.update_instances(scene)
    r.raytrace(v0)
    outs.append(r.read_radiance())
    scene.set_instance_transform(2, np.eye(4, dtype=np.float32).reshape(16))
    sg.update_instances(scene)
    outs.append(_state(r))
    return outs


@pytest.mark.parametrize("lanes", [1, 2])
def test_any_call_sequence_equals_the_eager_launches(device, cornell, lanes):
    scene, sg, pr = cornell
    views = (T.look(T.CORNELL_EYE, T.CORNELL_DIR), T.look((0.5, 0.4, 12.0), (-0.05, 0.02, -1.0)))
    results = []
    for max_fused in (1, 0, 3):
        r = _renderer(device, sg, pr, max_fused, lanes)
        results.append(_script(r, views, sg, scene))
        r.close()
    for other in results[1:]:
        for a, b in zip(results[0], other):
            if isinstance(a, np.ndarray):
                assert a.tobytes() == b.tobytes()
            else:
                assert a == b


def test_batches_are_cut_at_the_cap_and_nothing_is_lost_on_destroy(device, cornell):
    _, sg, pr = cornell
    view = T.look(T.CORNELL_EYE, T.CORNELL_DIR)
    r = _renderer(device, sg, pr, 3)
    for _ in range(7):                           # 3 + 3 launch when full, 1 stays recorded
        r.raytrace(view)
    assert r.frame_state() == (8, 7 * DEPTH)     # the protocol state moves at record time
    assert r.submission_stats() == (7, 2, 1)     # seven calls recorded, two full batches launched, one call still recorded
    img = r.read_radiance()
    assert r.submission_stats() == (7, 3, 0)     # the read submitted it
    r.raytrace(view)                             # recorded, never submitted
    r.close()
    e = _renderer(device, sg, pr, 1)
    for _ in range(7):
        e.raytrace(view)
    assert img.tobytes() == e.read_radiance().tobytes()
    e.close()


def test_atrium_frame_of_the_unchanged_caller_equals_the_batched_frame(device):
    desc = scenes.synthetic_atrium(texture_size=128)
    sg = lp.SceneGPU.new_from_scene(scenes.to_product(desc), device)
    pr = lp.ProbeGPU(device, desc["probe"], desc["probe"].shape[1], desc["probe"].shape[0])
    view = T.look(desc["camera"]["origin"], desc["camera"]["direction"])
    global DEPTH
    keep, DEPTH = DEPTH, 8
    try:
        a = _renderer(device, sg, pr, 0, size=(480, 270))
        for _ in range(4):
            a.raytrace(view)
        dst = lp.pinned_array((270, 480, 4))     # page-locked read-back destination (lpt_host_alloc)
        assert a.submission_stats() == (4, 0, 4)         # 480x270: the four calls of the frame wait for the read as ONE wavefront
        ia = a.read_radiance(out=dst).copy()
        assert a.submission_stats() == (4, 1, 0)
        b = _renderer(device, sg, pr, 0, size=(480, 270))
        b.raytrace_n(view, 4)
        assert ia.tobytes() == b.read_radiance().tobytes()
        assert _state(a) == _state(b)
        a.close(); b.close()
    finally:
        DEPTH = keep
    pr.close()
    sg.close()


def _cut_renderer(device, sg, pr, monkeypatch, rays, size, lanes):
    """an automatic renderer whose submissions are cut into wavefronts of about `rays` rays (lpt_renderer_set_option)"""
    return _renderer(device, sg, pr, 0, lanes, size=size, options={"wavefront_rays": rays})


@pytest.mark.parametrize("size,lanes", [((203, 117), 2), ((203, 117), 1), ((256, 128), 2), ((97, 301), 3)])
def test_a_large_batch_is_cut_into_runs_of_tile_rows_and_read_back_piece_by_piece(device, cornell, monkeypatch, size, lanes):
    """a recorded batch of more than ~4 M rays (here: of more than LPT_OPT_WAVEFRONT_RAYS) leaves as several wavefronts, each a run of tile
    rows with ALL the samples; read_radiance has each run copied to the host behind its own accumulation.  Bit for bit the
    frame of one wavefront, of eager launches, and of the ordinary read path; frames that continue an accumulation too."""
    _, sg, pr = cornell
    view = T.look(T.CORNELL_EYE, T.CORNELL_DIR)
    w, h = size
    tile_rows = -(-h // 8)
    ref = _renderer(device, sg, pr, 1, size=size)            # eager: one launch per call
    cut = _cut_renderer(device, sg, pr, monkeypatch, 3 * 256 * -(-w // 32) * 4, size, lanes)   # about three tile rows of 4 samples
    for _ in range(4):
        ref.raytrace(view); cut.raytrace(view)
    assert cut.submission_stats() == (4, 0, 4)
    dst = lp.pinned_array((h, w, 4))
    a = cut.read_radiance(out=dst).copy()                    # submits: the pieces carry their own read-back
    rec, wavefronts, pending = cut.submission_stats()
    assert (rec, pending) == (4, 0) and wavefronts == -(-tile_rows // 3)
    assert a.tobytes() == ref.read_radiance().tobytes()
    assert a.tobytes() == cut.read_radiance().tobytes()      # nothing recorded: the ordinary path, same frame
    # three more calls continue the accumulation (another cut: 3 samples per piece), then a view change resets nothing by itself
    for _ in range(3):
        ref.raytrace(view); cut.raytrace(view)
    v2 = T.look((0.5, 0.4, 12.0), (-0.05, 0.02, -1.0))
    ref.raytrace(v2); cut.raytrace(v2)                       # another view: a new batch (the first is submitted, cut, without a read)
    b = cut.read_radiance()
    assert b.tobytes() == ref.read_radiance().tobytes()
    assert _state(cut) == _state(ref)
    ref.close(); cut.close()


def test_the_cut_frame_equals_the_oracle_and_explicit_batches_stay_whole(device, cornell, cornell_glb, monkeypatch):
    _, sg, pr = cornell
    view = T.look(T.CORNELL_EYE, T.CORNELL_DIR)
    cut = _cut_renderer(device, sg, pr, monkeypatch, 20000, (W, H), 2)
    for _ in range(4):
        cut.raytrace(view)
    img = cut.read_radiance()
    ref, oc = harness.render_oracle(cornell_glb, W, H, DEPTH, 4)
    assert img.tobytes() == ref.tobytes()
    c = cut.ray_counts()
    assert (c.closest, c.shadow, c.shaded) == (oc.closest, oc.shadow, oc.shaded)
    assert cut.submission_stats()[1] > 4
    cut.close()
    whole = _cut_renderer(device, sg, pr, monkeypatch, 20000, (W, H), 2)
    whole.raytrace_n(view, 4)                                 # the explicit batch: ONE wavefront whatever its size
    assert whole.submission_stats() == (4, 1, 0)
    assert whole.read_radiance().tobytes() == ref.tobytes()
    whole.close()


@pytest.mark.parametrize("size,lanes", [((203, 117), 2), ((256, 128), 1)])
def test_read_pixels_and_blit_of_a_recorded_frame_travel_piece_by_piece_too(device, cornell, monkeypatch, size, lanes):
    """read_pixels (Renderer::read_pixels, renderer.rs:727-811) / blit on a still recorded frame: every wavefront's rows are tonemapped
    and copied behind its own accumulation, also into a destination with padded rows — byte for byte the ordinary path's image"""
    _, sg, pr = cornell
    view = T.look(T.CORNELL_EYE, T.CORNELL_DIR)
    w, h = size
    ref = _renderer(device, sg, pr, 1, size=size)
    for _ in range(4):
        ref.raytrace(view)
    want = ref.read_pixels()
    for how in ("read_pixels", "blit", "blit_padded"):
        cut = _cut_renderer(device, sg, pr, monkeypatch, 3 * 256 * -(-w // 32) * 4, size, lanes)
        for _ in range(4):
            cut.raytrace(view)
        assert cut.submission_stats() == (4, 0, 4)
        if how == "read_pixels":
            got = cut.read_pixels()
        elif how == "blit":
            got = cut.blit()
        else:
            pitch = w * 4 + 52
            buf = cut.blit(row_bytes=pitch)
            assert not buf[:, w * 4:].any()                  # the padding of every row is untouched
            got = buf[:, :w * 4].reshape(h, w, 4)
        assert cut.submission_stats()[1] > 1 and cut.submission_stats()[2] == 0
        assert got.tobytes() == want.tobytes(), how
        assert cut.read_pixels().tobytes() == want.tobytes()   # nothing recorded: the ordinary path
        cut.close()
    ref.close()


def test_every_kernel_variant_gives_the_oracle_frame(device, cornell, cornell_glb):
    """The default for a frame this small: bounce 0 per ray (packet traversal, k_trace_packet: one tree walk per 8x8-pixel patch, is chosen by pixel
    footprint — packet_primary = 1 forces it), then every later bounce in ONE launch (k_path); path_rays = 0 selects the per-bounce launches of renderer.rs:484-509 (k_shade + k_trace, one
    memory round trip per traversal step: ray_step_pipe), pipe_rays = 0 their two-round-trip step, packet_primary = 0 per-ray traversal
    for bounce 0 too, packet_quads = 0 packets of one sample over 8x8 pixels instead of four samples over 4x4.  The order of the tests differs, the frame and the ray
    counts do not (lpt_renderer_set_option: no environment variable selects a kernel)."""
    _, sg, pr = cornell
    view = T.look(T.CORNELL_EYE, T.CORNELL_DIR)
    ref, oc = harness.render_oracle(cornell_glb, W, H, DEPTH, 4)
    variants = ({}, {"packet_primary": 1}, {"packet_primary": 1, "packet_quads": 0}, {"path_rays": 0}, {"path_rays": 0, "packet_primary": 1}, {"path_rays": 0, "pipe_rays": 0, "packet_primary": 1},
                {"path_rays": 0, "pipe_rays": 30000}, {"packet_primary": 0}, {"packet_primary": 0, "pipe_rays": 0},
                {"path_waves_per_cu": 3, "path_refill": 20, "packet_primary": 1}, {"path_refill": 63}, {"path_refill": 0},
                # the step budget: rays not finished after n steps are dropped by the per-lane kernel and traced again by a whole wave (k_trace_coop);
                # n = 1: every ray of the per-bounce launches goes that way
                {"path_rays": 0, "step_budget": 0, "tail_lanes": 0}, {"path_rays": 0, "step_budget": 1, "tail_lanes": 0}, {"path_rays": 0, "step_budget": 7, "packet_primary": 1, "tail_lanes": 0},
                {"path_rays": 0, "step_budget": 12, "pipe_rays": 0, "tail_lanes": 0},
                {"path_rays": 0, "step_budget": 3, "budget_rays": 10, "tail_lanes": 0},
                # the shipped form of the same launches: tails finished in place (tests/test_gpu_tail.py)
                {"path_rays": 0, "tail_lanes": 4}, {"path_rays": 0, "tail_lanes": 8, "pipe_rays": 0},
                # a wave per ray for every ray of the wavefront (the form tiny wavefronts take by themselves: tests/test_gpu_coop_all.py)
                {"coop_rays": 0x7FFFFFFF}, {"coop_rays": 0x7FFFFFFF, "packet_primary": 1}, {"coop_rays": 0})
    for opts in variants:
        r = _renderer(device, sg, pr, 0, options=opts)
        assert all(r.get_option(k) == v for k, v in opts.items())
        packet = r.get_option("packet_primary")       # the suite's `pipeline` fixture forces 1 in its "path" arm
        for _ in range(4):
            r.raytrace(view)
        assert r.read_radiance().tobytes() == ref.tobytes(), opts
        c = r.ray_counts()
        assert (c.closest, c.shadow, c.shaded) == (oc.closest, oc.shadow, oc.shaded), opts
        # which kernel traced bounce 0: packets only where asked for — the default (2) picks by pixel footprint, and a frame this small is traced per ray
        assert c.primary == (4 * W * H if packet == 1 else 0), opts
        r.close()


@pytest.mark.parametrize("samples", [4, 8, 12, 6])
def test_packets_of_four_samples_on_a_dense_frame(device, cornell, cornell_glb, samples):
    """k_trace_packet on a frame of whole 32x8 tiles (the 203x117 frames above are not): with a multiple of four samples in the wavefront a packet is the
    four samples of a 4x4-pixel quarter of an 8x8 patch (LPT_EXP_PACKET_QUADS, default), else — 6 samples, or the option off — one sample of the whole patch.
    Which 64 rays share a tree walk changes nothing: the frame is the oracle's, the primary rays are all traced by packets."""
    _, sg, pr = cornell
    size = (256, 136)
    view = T.look(T.CORNELL_EYE, T.CORNELL_DIR)
    ref, oc = harness.render_oracle(cornell_glb, size[0], size[1], DEPTH, samples)
    for quads in (1, 0):
        r = _renderer(device, sg, pr, 0, size=size, options={"packet_primary": 1, "packet_quads": quads})
        r.raytrace_n(view, samples)
        assert r.read_radiance().tobytes() == ref.tobytes(), (samples, quads)
        c = r.ray_counts()
        assert (c.closest, c.shadow, c.shaded, c.primary) == (oc.closest, oc.shadow, oc.shaded, samples * size[0] * size[1]), (samples, quads)
        r.close()
