"""-m gpu: the ASVGF path (BlitMode::DenoisedPathrace / Temporal; reference render/asvgf.rs, renderer.rs:466-522)
against the oracle's restatement (SPEC §15), frame by frame with a moving camera: G-buffer, motion vectors,
temporally accumulated radiance + variance, history length and the composited main target — all bit-exact."""
import numpy as np
import pytest

import loupiote_amd as lp
from loupiote_amd import dist as D, testing as T

pytestmark = [pytest.mark.gpu, pytest.mark.usefixtures("pipeline")]   # every test body over the three arms of tests/conftest.py PIPELINES: k_path, the per-bounce launches, the shipped defaults


def _setup(device, glb, w, h, bounces):
    from oracle import gltf_oracle as G, orc
    scene = lp.Scene()
    lp.loaders.load_gltf(glb, scene)
    scene.set_light(0, T.cornell_light())
    sg = lp.SceneGPU.new_from_scene(scene, device)
    pr = lp.ProbeGPU(device, T.CORNELL_PROBE, 1, 1)
    r = lp.Renderer(device, (w, h))
    r.downsample_factor = 1.0
    r.resize(device, sg, pr, (w, h))
    r.set_max_bounces(bounces)
    r.set_vfov(T.VFOV)
    s = G.Scene()
    G.load_gltf(glb, s)
    s.lights[0] = T.cornell_light()[0]
    osc = orc.OracleScene.from_scene(s, probe=T.CORNELL_PROBE)
    return sg, pr, r, osc


@pytest.mark.parametrize("mode", [lp.BlitMode.DenoisedPathrace, lp.BlitMode.Temporal])
def test_denoiser_frames_match_oracle(device, cornell_glb, mode):
    from oracle import orc
    w, h, bounces = 160, 96, 3
    sg, pr, r, osc = _setup(device, cornell_glb, w, h, bounces)
    r.set_blit_mode(mode)
    den = orc.Denoiser(osc, w, h, T.VFOV, bounces)
    eyes = [(0.0, 0.6, 13.5), (0.0, 0.6, 13.5), (0.15, 0.62, 13.4), (0.3, 0.65, 13.3), (0.3, 0.65, 13.3), (0.3, 0.65, 13.3)]
    for k, eye in enumerate(eyes):
        view = T.look(eye, (0.0, 0.0, -1.0))
        r.reset_accumulation() if k in (0, 2, 3) else None      # the app resets while the camera moves (app.rs:308-310)
        r.raytrace(view)
        r.accumulate = True
        want = den.frame(view, int(mode))
        got = r.read_radiance()
        g, m, rad, hist = r.read_denoiser()
        og, om, orad, ohist = den.read()
        assert np.array_equal(g, og), "G-buffer, frame %d" % k
        assert m.tobytes() == om.tobytes(), "motion, frame %d" % k
        assert np.array_equal(hist, ohist), "history, frame %d" % k
        assert rad.tobytes() == orad.tobytes(), "temporal radiance, frame %d" % k
        assert got.tobytes() == want.tobytes(), "main target, frame %d" % k
    # static camera: zero motion, history grows; moving camera: most pixels still reproject
    assert np.all(m == 0) and hist.max() >= 3
    assert r.frame_state()[0] == 1          # frame_count only moves in the Pahtrace arm (renderer.rs:523-538)
    srgb = r.read_pixels()
    assert srgb.shape == (h, w, 4) and srgb[..., 3].min() == 255
    r.close(); pr.close(); sg.close()


def test_denoiser_reduces_noise_and_debug_views(device, cornell_glb):
    w, h, bounces = 256, 256, 4
    sg, pr, r, _ = _setup(device, cornell_glb, w, h, bounces)
    view = T.look(T.CORNELL_EYE, T.CORNELL_DIR)
    r.reset_accumulation()
    r.accumulate = True
    r.raytrace_n(view, 64)
    ref = r.read_radiance()[..., :3]                 # 64-spp reference
    r.reset_accumulation()
    r.raytrace(view)
    noisy = r.read_radiance()[..., :3]               # 1 spp
    r.set_blit_mode(lp.BlitMode.DenoisedPathrace)
    for _ in range(8):
        r.raytrace(view)
    den = r.read_radiance()[..., :3]
    e_noisy = float(np.mean((noisy - ref) ** 2))
    e_den = float(np.mean((den - ref) ** 2))
    print("MSE vs 64 spp: 1 spp %.5f, denoised (8 frames) %.5f" % (e_noisy, e_den))
    assert e_den < 0.35 * e_noisy     # measured: 0.068 -> 0.017
    r.set_blit_mode(lp.BlitMode.GBuffer)
    r.raytrace(view)
    nrm = r.blit()
    assert nrm.shape == (h, w, 4) and nrm[..., :3].std() > 5     # normals vary over the box
    r.set_blit_mode(lp.BlitMode.MotionVector)
    r.raytrace(view)
    assert r.blit()[..., :2].max() == 0                          # static camera
    r.close(); pr.close(); sg.close()


def test_sharded_denoising_equals_single_gpu(device, cornell_glb):
    """Multi-GPU denoising (config 5): two tile shards (emulated on this GPU by two renderers) trace their tiles,
    their filter inputs are summed (what `dist.exchange_denoiser_inputs` does with RCCL reduces) into rank 0's
    buffers, rank 0 filters the whole frame — bit-identical to the single-GPU denoiser, frame after frame, with a
    moving camera so that reprojection crosses tile borders."""
    import torch
    from loupiote_amd.dist import DevView
    scene = lp.Scene()
    lp.loaders.load_gltf(cornell_glb, scene)
    scene.set_light(0, T.cornell_light())
    sg = lp.SceneGPU.new_from_scene(scene, device)
    pr = lp.ProbeGPU(device, T.CORNELL_PROBE, 1, 1)
    W, H = 160, 96

    def make(rank, world):
        r = lp.Renderer(device, (W, H))
        r.downsample_factor = 1.0
        r.resize(device, sg, pr, (W, H))
        r.set_max_bounces(3)
        r.set_vfov(T.VFOV)
        if world > 1:
            r.set_shard(rank, world)
            r.set_resources(device, sg, pr)
        r.set_blit_mode(lp.BlitMode.DenoisedPathrace)
        return r

    single, a, b = make(0, 1), make(0, 2), make(1, 2)
    dev = torch.device("cuda", 0)

    def views(r):
        noisy, gbuf, motion, n = r.denoiser_inputs()
        return [torch.as_tensor(DevView(noisy, 4 * n, "<f4"), device=dev), torch.as_tensor(DevView(gbuf, 4 * n, "<i4"), device=dev),
                torch.as_tensor(DevView(motion, 2 * n, "<f4"), device=dev)]

    for f in range(4):
        eye = (0.15 * f, 0.6 + 0.05 * f, 13.5 - 0.2 * f)
        view = T.look(eye, T.CORNELL_DIR)
        for r in (single, a, b):
            r.raytrace(view)
            r.synchronize()
        va, vb = views(a), views(b)
        own = torch.from_numpy(D.owned_mask(W, H, 0, 2)).to(dev)
        assert bool((va[0].view(H, W, 4)[~own] == 0).all()) and bool((vb[0].view(H, W, 4)[own] == 0).all())
        for x, y in zip(va, vb):
            x += y                                    # the reduce (sum over ranks) onto rank 0
        torch.cuda.synchronize()
        a.denoise_filter()
        assert a.read_radiance().tobytes() == single.read_radiance().tobytes(), "frame %d" % f
        ga, ma, ra, ha = a.read_denoiser()
        gs, ms, rs, hs = single.read_denoiser()
        assert ga.tobytes() == gs.tobytes() and ma.tobytes() == ms.tobytes() and ra.tobytes() == rs.tobytes() and ha.tobytes() == hs.tobytes()
    assert ha.max() == 4
    for r in (single, a, b):
        r.close()
    pr.close(); sg.close()


def test_exchange_plumbing_on_device_pointers():
    """one-rank RCCL group: exchange_denoiser_inputs / reduce_radiance / OwnedTileGather on real device pointers
    (tests/tools/dist_den_check.py, in a subprocess so that the process group does not leak into this session)"""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    p = subprocess.run([sys.executable, os.path.join(root, "tests", "tools", "dist_den_check.py")], capture_output=True, text=True, timeout=600)
    assert p.returncode == 0 and "dist-den-check ok" in p.stdout, (p.stdout[-2000:], p.stderr[-2000:])
