"""-m gpu: the HIP path against ANALYTIC values (not against the oracle): the furnace of lights, the light-tight closed box and the
small-light closed form of tests/kat_scenes.py at 256 x 256.  'Parity unpinned' means neither the oracle
nor the product can be compared with the reference's WGSL; both can be compared with physics."""
import numpy as np
import pytest

import kat_scenes as K
import loupiote_amd as lp
from loupiote_amd import scenes, testing as T

pytestmark = [pytest.mark.gpu, pytest.mark.usefixtures("pipeline")]


def _render(device, desc, view, vfov, bounces, frames, size=(256, 256)):
    sg = lp.SceneGPU.new_from_scene(scenes.to_product(desc), device)
    pr = lp.ProbeGPU(device, desc["probe"], desc["probe"].shape[1], desc["probe"].shape[0])
    r = lp.Renderer(device, size)
    r.downsample_factor = 1.0
    r.resize(device, sg, pr, size)
    r.set_max_bounces(bounces)
    r.set_vfov(vfov)
    r.reset_accumulation()
    r.accumulate = True
    for _ in range(frames):
        r.raytrace(view)
    img = r.read_radiance()
    c = r.ray_counts()
    r.close(); pr.close(); sg.close()
    return img, c


@pytest.mark.parametrize("base,rough,metal", [((0.8, 0.6, 0.4), 0.5, 0.0), ((0.95, 0.9, 0.8), 0.25, 1.0)])
def test_furnace_of_lights_gives_the_directional_albedo(device, base, rough, metal):
    Le = 2.0
    desc = K.light_box_furnace(base, rough, metal, radiance=Le)
    eye = np.array([0.9, 1.3, 2.2])
    img, _ = _render(device, desc, T.look(eye, -eye), 0.02, 3, 8)     # 256^2 pixels x 8 samples of (nearly) one shading point
    assert np.all(img[..., 3] == 1.0)
    rgb = img[..., :3].astype(np.float64).reshape(-1, 3)
    got, err = rgb.mean(axis=0), rgb.std(axis=0, ddof=1) / np.sqrt(rgb.shape[0])
    want = Le * K.directional_albedo(base, rough, metal, eye)
    assert np.all(np.abs(got - want) < np.maximum(4.0 * err, 0.004 * want)), (got, want, err)
    wall, _ = _render(device, desc, T.look((0.0, 2.0, 0.0), (0.3, 1.0, 0.2)), 0.5, 3, 2, size=(64, 64))
    assert np.all(wall[..., :3] == Le)      # an emitter seen directly: exactly Le, every sample


def test_a_closed_box_is_light_tight(device):
    desc = K.closed_box()
    for eye, d in (((0.3, -0.2, 0.1), (1.0, 0.2, 0.3)), ((-1.9, 1.9, 1.9), (1.0, -1.0, -1.0)), ((0.0, 0.0, 0.0), (-1.0, -1.0, -1.0))):
        img, c = _render(device, desc, T.look(eye, d), 1.2, 8, 3)
        assert np.all(img[..., :3] == 0.0) and np.all(img[..., 3] == 1.0)
        assert c.closest > 256 * 256 * 3 * 4 and c.shadow == 0      # the only emitter has radiance 0: no shadow ray is ever cast
    outside, _ = _render(device, desc, T.look((0.0, 0.0, 9.0), (0.0, 0.0, 1.0)), 0.3, 2, 1, size=(32, 32))
    assert np.all(outside[..., :3] == 100.0)


def test_small_light_closed_form(device):
    base, rough, metal = (0.7, 0.5, 0.3), 0.6, 0.0
    desc = K.small_light(base, rough, metal)
    eye, P = np.array([0.0, 1.0, 3.0]), np.array([0.2, 0.0, 0.3])
    img, _ = _render(device, desc, T.look(eye, P - eye), 0.004, 2, 4)
    rgb = img[..., :3].astype(np.float64).reshape(-1, 3)
    got, err = rgb.mean(axis=0), rgb.std(axis=0, ddof=1) / np.sqrt(rgb.shape[0])
    w = np.array(desc["light_pos"]) - P
    d2 = float(w @ w)
    wi = w / np.sqrt(d2)
    V = (eye - P) / np.linalg.norm(eye - P)
    f = K.bsdf(base, rough, metal, np.array([[0.0, 1.0, 0.0]]), V[None], wi[None])[0]
    want = f * wi[1] * 4.0e4 * desc["light_area"] * wi[1] / d2
    assert np.all(np.abs(got - want) < np.maximum(4.0 * err, 0.004 * want)), (got, want, err)
