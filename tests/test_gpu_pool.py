"""-m gpu: k_pool — every bounce behind the primary hits in ONE launch as a CU-local pool of trace and shade work (loupiote_amd/csrc/pool_kernels.h; the
loop it replaces: reference crates/lib/src/renderer.rs:484-509).  Whatever the shape of the pool — waves per block, waves that prefer shading, retire
threshold, records per block — the frame, the ray counts and the per-bounce queue sizes are the ORACLE's bit for bit: a path's radiance is summed in the
order of the per-bounce launches (the light sample of bounce b before anything of bounce b + 1).  The `pool` arm of the suite's `pipeline` fixture runs
every other parity test through the default shape; the protocol's model under ThreadSanitizer is tests/test_pool_model.py (CPU)."""
import numpy as np
import pytest

import loupiote_amd as lp
from loupiote_amd import scenes, testing as T
from oracle import harness

pytestmark = pytest.mark.gpu
W, H, DEPTH = 203, 117, 5
POOL = {"path_rays": 0, "pool_rays": 0x7FFFFFFF, "coop_rays": 0}   # coop_rays 0: small wavefronts would otherwise be traced a wave per ray


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


def _renderer(device, sg, pr, size, depth, options):
    r = lp.Renderer(device, size)
    r.downsample_factor = 1.0
    r.resize(device, sg, pr, size)
    r.set_max_bounces(depth)
    r.set_vfov(T.VFOV)
    for k, v in options.items():
        r.set_option(k, v)
    r.reset_accumulation()
    r.accumulate = True
    r.reset_ray_counts()
    return r


SHAPES = [{}, {"pool_waves": 16}, {"pool_waves": 16, "pool_shaders": 6}, {"pool_waves": 16, "pool_shaders": 0}, {"pool_waves": 4, "pool_shaders": 1}, {"pool_waves": 4, "pool_shaders": 3},
          {"pool_waves": 8, "pool_shaders": 8}, {"pool_refill": 0}, {"pool_refill": 63}, {"pool_entries": 256}, {"pool_entries": 256, "pool_waves": 16, "pool_refill": 20},
          {"packet_primary": 1}, {"packet_primary": 0, "pool_waves": 4}]


def test_every_pool_shape_gives_the_oracles_frame_counts_and_queue_sizes(device, cornell, cornell_glb):
    _, sg, pr = cornell
    view = T.look(T.CORNELL_EYE, T.CORNELL_DIR)
    ref, oc = harness.render_oracle(cornell_glb, W, H, DEPTH, 4)
    base = _renderer(device, sg, pr, (W, H), DEPTH, {"path_rays": 0, "pool_rays": 0})
    for _ in range(4):
        base.raytrace(view)
    assert base.read_radiance().tobytes() == ref.tobytes()
    q_ref = base.queue_counts(DEPTH)
    base.close()
    for shape in SHAPES:
        r = _renderer(device, sg, pr, (W, H), DEPTH, dict(POOL, **shape))
        assert all(r.get_option(k) == v for k, v in shape.items())
        for _ in range(4):
            r.raytrace(view)
        assert r.read_radiance().tobytes() == ref.tobytes(), shape
        c = r.ray_counts()
        assert (c.closest, c.shadow, c.shaded) == (oc.closest, oc.shadow, oc.shaded), shape
        q = r.queue_counts(DEPTH)
        assert [list(map(int, x)) for x in q] == [list(map(int, x)) for x in q_ref], shape
        r.close()


def test_the_pool_kernel_is_the_one_that_ran(device, cornell):
    """stage timers: a pool wavefront has a "path" stage (the one launch) and no "shading" / "intersection" stage; the option off gives the opposite"""
    _, sg, pr = cornell
    view = T.look(T.CORNELL_EYE, T.CORNELL_DIR)
    for opts, want_path in ((dict(POOL, packet_primary=1), True), ({"path_rays": 0, "pool_rays": 0}, False)):
        r = _renderer(device, sg, pr, (256, 136), DEPTH, opts)
        r.enable_timings(True)
        r.raytrace_n(view, 4)
        r.synchronize()
        tm = r.timings()
        assert (tm.get("path", (0, 0))[1] > 0) is want_path, (opts, tm)
        assert (tm.get("shading", (0, 0))[1] > 0) is (not want_path), (opts, tm)
        r.close()


def test_pool_on_the_textured_stand_in_with_stats_and_a_tile_shard(device):
    """the bench scene at a small size: textured materials (paired texels), misses into the RGBE sky, emitter hits; the stats variant of the kernel (nodes /
    triangles per ray equal the per-bounce stats kernels'); rank 1 of a 3-way tile shard"""
    desc = scenes.synthetic_atrium(texture_size=64)
    scene = scenes.to_product(desc)
    sg = lp.SceneGPU.new_from_scene(scene, device)
    pr = lp.ProbeGPU(device, desc["probe"], desc["probe"].shape[1], desc["probe"].shape[0])
    view = T.look(desc["camera"]["origin"], desc["camera"]["direction"])
    outs = {}
    for name, opts in (("per_bounce", {"path_rays": 0, "pool_rays": 0, "step_budget": 0}), ("pool", POOL), ("pool16", dict(POOL, pool_waves=16, pool_shaders=3))):
        for shard in (None, (1, 3)):
            r = _renderer(device, sg, pr, (320, 176), 8, opts)
            if shard:
                r.set_shard(*shard)
                r.set_resources(device, sg, pr)
                r.reset_accumulation()
                r.accumulate = True
            r.enable_stats(True)
            r.raytrace_n(view, 4)
            img = r.read_radiance()
            c = r.ray_counts()
            outs[(name, shard)] = (img.tobytes(), (c.closest, c.shadow, c.shaded, c.nodes, c.tris, c.shadow_nodes, c.shadow_tris))
            r.close()
    for shard in (None, (1, 3)):
        assert outs[("pool", shard)] == outs[("per_bounce", shard)], shard
        assert outs[("pool16", shard)] == outs[("per_bounce", shard)], shard
    pr.close()
    sg.close()


def test_pool_in_the_denoising_modes_writes_the_same_g_buffer(device, cornell):
    """BlitMode::Temporal: the primary pass of a pool wavefront (bounce-0 shading inside k_pool<GBUF>) writes G-buffer and motion as k_shade<GBUF> does"""
    _, sg, pr = cornell
    view = T.look(T.CORNELL_EYE, T.CORNELL_DIR)
    outs = []
    for opts in ({"path_rays": 0, "pool_rays": 0}, dict(POOL, packet_primary=1), dict(POOL, packet_primary=0, pool_waves=4)):
        r = _renderer(device, sg, pr, (256, 136), DEPTH, opts)
        r.set_blit_mode(lp.BlitMode.Temporal)
        for _ in range(3):
            r.raytrace(view)
        g, m, rad, hist = r.read_denoiser()
        outs.append((g.tobytes(), m.tobytes(), rad.tobytes(), hist.tobytes(), r.read_pixels().tobytes()))
        r.close()
    assert outs[1] == outs[0] and outs[2] == outs[0]


def test_stats_kernels_of_every_traversal_variant_run_clean(tmp_path):
    """round 5 regression: with stats on, the occluder-cache probe of k_trace reads the occluder's leaf slot out of an any-hit ray's `v` — which only the
    one-round-trip step wrote.  `pipe_rays = 0` + stats (what `bench.py --opt pipe_rays=0` does in its stats frame) indexed the triangle array with a
    barycentric's bit pattern: a GPU memory fault at bench size.  Every variant in its own process (a fault kills the process, not the suite); the stats
    frame must equal the plain frame."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = r'''
import sys, json
sys.path.insert(0, %r)
import numpy as np
import loupiote_amd as lp
from loupiote_amd import scenes, testing as T
opts = json.loads(sys.argv[1])
dev = lp.Device(0)
desc = scenes.synthetic_atrium(texture_size=64)
sg = lp.SceneGPU.new_from_scene(scenes.to_product(desc), dev)
pr = lp.ProbeGPU(dev, desc["probe"], desc["probe"].shape[1], desc["probe"].shape[0])
view = T.look(desc["camera"]["origin"], desc["camera"]["direction"])
sums = []
for stats in (0, 1):
    r = lp.Renderer(dev, (960, 544)); r.downsample_factor = 1.0; r.resize(dev, sg, pr, (960, 544)); r.set_max_bounces(8); r.set_vfov(T.VFOV)
    for k, v in opts.items(): r.set_option(k, v)
    r.enable_stats(bool(stats))
    r.reset_accumulation(); r.accumulate = True
    r.raytrace_n(view, 4)
    img = r.read_radiance()
    c = r.ray_counts()
    sums.append((float(np.float64(img[..., :3].sum())), int(c.closest), int(c.shadow), int(c.nodes), int(c.shadow_nodes)))
    r.close()
print(json.dumps(sums))
''' % root
    for opts in ({"pipe_rays": 0}, {"pipe_rays": 0, "merge_trace": 0}, {}, {"path_rays": 0x7FFFFFFF}, {"path_rays": 0, "pool_rays": 0x7FFFFFFF}, {"pipe_rays": 0, "step_budget": 8, "budget_rays": 0x7FFFFFFF, "tail_lanes": 0}, {"path_rays": 0, "tail_lanes": 8, "budget_rays": 0x7FFFFFFF}):
        p = subprocess.run([sys.executable, "-c", code, json.dumps(opts)], capture_output=True, text=True, timeout=600)
        assert p.returncode == 0, (opts, p.stderr[-1500:])
        plain, stats = json.loads(p.stdout.strip().splitlines()[-1])
        assert plain[:3] == stats[:3], opts                # the same frame and ray counts
        assert plain[3] == 0 and stats[3] > 0 and (stats[4] > 0 or opts.get("merge_trace") == 0), (opts, plain, stats)   # nodes are counted by the stats kernels only
