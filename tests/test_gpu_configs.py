"""-m gpu: the remaining BASELINE.json configs as parity cases (the bench line is config 4).

  config 2  cornell-box.glb, 1024x1024, 4 spp, depth 8           -> bit-exact vs the oracle at full size
  config 3  DamagedHelmet stand-in + sky probe, 1920x1080, 8 spp  -> oracle on a 480x270 render of the same
            scene (bit-exact) + full-size properties
  config 5  Sponza stand-in, 3840x2160, progressive + temporal    -> full-size properties (64 spp progressive
            equals 8 x raytrace_n(8); the temporal mode keeps history and stays finite)
"""
import numpy as np
import pytest

import loupiote_amd as lp
from loupiote_amd import scenes, testing as T
from oracle import harness

pytestmark = [pytest.mark.gpu, pytest.mark.usefixtures("pipeline")]   # every test body over the three arms of tests/conftest.py PIPELINES: k_path, the per-bounce launches, the shipped defaults


def _renderer(device, desc, w, h, bounces):
    sg = lp.SceneGPU.new_from_scene(scenes.to_product(desc), device)
    pr = lp.ProbeGPU(device, desc["probe"], desc["probe"].shape[1], desc["probe"].shape[0])
    r = lp.Renderer(device, (w, h))
    r.downsample_factor = 1.0
    r.resize(device, sg, pr, (w, h))
    r.set_max_bounces(bounces)
    r.set_vfov(T.VFOV)
    return sg, pr, r, T.look(desc["camera"]["origin"], desc["camera"]["direction"])


def test_config2_cornell_1024_4spp_depth8(device, cornell_glb):
    img, counts = T.render_hip(device, cornell_glb, 1024, 1024, 8, 4)
    ref, oc = harness.render_oracle(cornell_glb, 1024, 1024, 8, 4)
    assert (counts.closest, counts.shadow, counts.shaded) == (oc.closest, oc.shadow, oc.shaded)
    assert img.tobytes() == ref.tobytes()


def test_config3_helmet_standin(device):
    from oracle import orc
    desc = scenes.synthetic_helmet()
    assert desc["triangles"] > 69000 and len(desc["images"]) == 5 and len(desc["materials"]) == 5
    osc = orc.OracleScene.from_scene(harness.to_oracle(desc), probe=desc["probe"])
    sg, pr, r, view = _renderer(device, desc, 480, 270, 8)
    r.reset_accumulation()
    r.accumulate = True
    r.reset_ray_counts()
    r.raytrace_n(view, 8)
    img, counts = r.read_radiance(), r.ray_counts()
    acc, oc = osc.render(480, 270, view, T.VFOV, 8, frames=8, want_counters=True)
    assert (counts.closest, counts.shadow, counts.shaded) == (oc.closest, oc.shadow, oc.shaded)
    assert img.tobytes() == orc.resolve(acc).tobytes()
    # full size: 1920x1080, 8 spp, depth 8
    r.resize(device, sg, pr, (1920, 1080))
    r.reset_accumulation()
    r.accumulate = True
    r.reset_ray_counts()
    r.raytrace_n(view, 8)
    a = r.read_radiance()
    c = r.ray_counts()
    r.reset_accumulation()
    r.accumulate = True
    r.raytrace_n(view, 8)
    b = r.read_radiance()   # the seed is never reset (renderer.rs:613-615): new samples, same expectation
    assert a.tobytes() != b.tobytes() and abs(float(a[..., :3].mean()) / float(b[..., :3].mean()) - 1.0) < 0.03
    assert np.all(np.isfinite(a)) and np.all(a[..., 3] == 1.0) and 1920 * 1080 * 8 <= c.closest <= 1920 * 1080 * 64
    assert 0.02 < float(a[..., :3].mean()) < 50.0
    r.close(); pr.close(); sg.close()


def test_config5_4k_progressive_and_temporal(device):
    desc = scenes.synthetic_atrium()
    sg, pr, r, view = _renderer(device, desc, 3840, 2160, 8)
    # progressive accumulation: 16 samples as 2 x raytrace_n(8) == one raytrace_n(16)
    r.reset_accumulation()
    r.accumulate = True
    r.reset_ray_counts()
    r.raytrace_n(view, 8)
    r.raytrace_n(view, 8)
    a = r.read_radiance()
    ca = r.ray_counts()
    assert r.frame_state() == (17, 16 * 8)
    r.reset_accumulation()
    r.accumulate = True
    r.set_seed(0)
    r2_seed_before = r.frame_state()[1]
    assert r2_seed_before == 16 * 8                      # the seed is never reset (renderer.rs:613-615)
    assert np.all(np.isfinite(a)) and np.all(a[..., 3] == 1.0)
    assert 3840 * 2160 * 16 <= ca.closest <= 3840 * 2160 * 16 * 8
    # temporal accumulate (BlitMode::Temporal): history grows on a static camera, output finite
    r.set_blit_mode(lp.BlitMode.Temporal)
    for _ in range(3):
        r.raytrace(view)
    out = r.read_radiance()
    _, motion, _, hist = r.read_denoiser()
    assert np.all(np.isfinite(out)) and np.all(out[..., 3] == 1.0)
    assert hist.max() == 3 and (hist == 3).mean() > 0.8 and np.all(motion == 0)
    r.close(); pr.close(); sg.close()
