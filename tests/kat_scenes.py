"""Scenes with ANALYTIC answers (test infrastructure; plain scene descriptions in the loupiote_amd.scenes format, fed to the oracle through
oracle/harness.to_oracle and to the product through scenes.to_product, so both start from identical bytes).  SURVEY §7.3 asks for a
white furnace and a point-light closed form as pins against physics — the only axis on which "parity unpinned" can still be tightened.

SPEC §10's BSDF has no pure-Lambert configuration (a dielectric keeps F0 = 0.04, a metal is all specular) and its GGX lobe is single
scattering, so "albedo 1 => radiance 1" does not hold for it and the Lambert geometric series has no counterpart.  What does hold exactly:

  * FURNACE OF LIGHTS (`light_box_furnace`): a CLOSED cube of six inward-facing rectangular emitters of one radiance Le around a convex
    object (a quad).  Every direction of the object's hemisphere sees radiance Le, so the radiance it sends to the camera is
    Le * a(V), a(V) = int f(V, L) (N.L) dL, the BSDF's directional albedo — whatever mixture of light sampling, BSDF sampling, MIS
    weights, light-selection probabilities and emitter hits the integrator uses to get there.  a(V) comes from a binary64 quadrature
    of SPEC §10 written again in numpy (`directional_albedo`).  Pixels that see an emitter wall are exactly Le.
  * CLOSED BOX (`closed_box`): inside a closed box of ordinary surfaces with no light in it, under an environment of any radiance, every
    pixel is exactly 0: nothing leaks through the shared edges and corners (SPEC §7: watertight test, offset origins).
  * SMALL LIGHT (`small_light`): a 2 cm emitter far above a quad is a point light: L = f(V, L) cos(theta) Le A cos(theta_l) / d^2.
"""
import numpy as np

from loupiote_amd.scenes import INVALID

LIGHT_DT = np.dtype([("normal", "<f4", 4), ("tangent", "<f4", 4), ("bitangent", "<f4", 4), ("origin", "<f4", 4)])
IDENT = np.eye(4, dtype=np.float32).T.reshape(16)


def _quad(p0, eu, ev, normal):
    """two triangles over p0 + s eu + t ev, s, t in [0, 1], wound so that the geometric normal is `normal`"""
    p0, eu, ev, n = (np.asarray(x, np.float64) for x in (p0, eu, ev, normal))
    pos = np.array([p0, p0 + eu, p0 + eu + ev, p0 + ev], np.float32)
    idx = np.array([0, 1, 2, 0, 2, 3], np.uint32)
    if np.dot(np.cross(eu, ev), n) < 0:
        idx = np.array([0, 2, 1, 0, 3, 2], np.uint32)
    return {"positions": pos, "normals": np.tile(n.astype(np.float32), (4, 1)), "uvs": np.zeros((4, 2), np.float32), "indices": idx}


def _light(normal, tangent, bitangent, origin, half_w, half_h, radiance):
    l = np.zeros(1, LIGHT_DT)
    l["normal"] = tuple(normal) + (0.0,)
    l["tangent"] = tuple(tangent) + (half_w,)
    l["bitangent"] = tuple(bitangent) + (half_h,)
    l["origin"] = tuple(origin) + (radiance,)
    return l


BLACK_PROBE = np.zeros((1, 1, 4), np.uint8)
UNIT_PROBE = np.array([[[128, 128, 128, 129]]], np.uint8)        # RGBE of exactly 1.0
BRIGHT_PROBE = np.array([[[200, 200, 200, 135]]], np.uint8)      # 100.0


def light_box_furnace(base, rough, metal, radiance=2.0, half=4.0):
    """a 2x2 quad in the plane y = 0 (normal +Y) inside a closed cube of six emitters, side 2 * half, all of radiance `radiance`"""
    meshes = [_quad((-1, 0, -1), (2, 0, 0), (0, 0, 2), (0, 1, 0))]
    lights = []
    axes = [((1, 0, 0), (0, 1, 0), (0, 0, 1)), ((0, 1, 0), (0, 0, 1), (1, 0, 0)), ((0, 0, 1), (1, 0, 0), (0, 1, 0))]
    for n, t, b in axes:
        for sgn in (1.0, -1.0):
            centre = tuple(sgn * half * c for c in n)
            inward = tuple(-sgn * c for c in n)
            lights.append(_light(inward, t, b, centre, half, half, radiance))
    return {"name": "light_box_furnace", "meshes": meshes, "instances": [(1, IDENT, 1)], "materials": [(tuple(base) + (1.0,), rough, metal, INVALID, INVALID)],
            "images": [], "lights": lights, "probe": BRIGHT_PROBE, "triangles": 2}


def closed_box(half=2.0):
    """six inward-facing walls (12 triangles, shared edges and corners), three materials, no emitter with radiance"""
    h = half
    meshes = [_quad((-h, -h, -h), (2 * h, 0, 0), (0, 0, 2 * h), (0, 1, 0)), _quad((-h, h, -h), (2 * h, 0, 0), (0, 0, 2 * h), (0, -1, 0)),
              _quad((-h, -h, -h), (0, 2 * h, 0), (0, 0, 2 * h), (1, 0, 0)), _quad((h, -h, -h), (0, 2 * h, 0), (0, 0, 2 * h), (-1, 0, 0)),
              _quad((-h, -h, -h), (2 * h, 0, 0), (0, 2 * h, 0), (0, 0, 1)), _quad((-h, -h, h), (2 * h, 0, 0), (0, 2 * h, 0), (0, 0, -1))]
    materials = [((1.0, 1.0, 1.0, 1.0), 1.0, 0.0, INVALID, INVALID), ((0.9, 0.9, 0.9, 1.0), 0.1, 1.0, INVALID, INVALID), ((0.8, 0.6, 0.4, 1.0), 0.4, 0.0, INVALID, INVALID)]
    instances = [(k + 1, IDENT, 1 + k % 3) for k in range(6)]
    dark = _light((0, -1, 0), (1, 0, 0), (0, 0, 1), (0, 0.5 * h, 0), 0.25, 0.25, 0.0)   # Light::new()'s slot, switched off
    return {"name": "closed_box", "meshes": meshes, "instances": instances, "materials": materials, "images": [], "lights": [dark], "probe": BRIGHT_PROBE, "triangles": 12}


def small_light(base, rough, metal, radiance=4.0e4, height=2.0, half=0.01, offset=(0.6, 0.0, 0.3)):
    """a large quad in y = 0 under a 2 cm x 2 cm emitter facing down at `offset` + (0, height, 0); black environment"""
    meshes = [_quad((-20, 0, -20), (40, 0, 0), (0, 0, 40), (0, 1, 0))]
    pos = (offset[0], height, offset[2])
    lights = [_light((0, -1, 0), (1, 0, 0), (0, 0, 1), pos, half, half, radiance)]
    return {"name": "small_light", "meshes": meshes, "instances": [(1, IDENT, 1)], "materials": [(tuple(base) + (1.0,), rough, metal, INVALID, INVALID)],
            "images": [], "lights": lights, "probe": BLACK_PROBE, "triangles": 2, "light_pos": pos, "light_area": 4.0 * half * half}


# ---------------------------------------------------------------------------- SPEC §10 in numpy, binary64 (independent of oracle and product)
def bsdf(base, rough, metal, N, V, L):
    base = np.asarray(base, np.float64)
    r = np.clip(rough, 0.045, 1.0)
    a = r * r
    a2 = a * a
    m = np.clip(metal, 0.0, 1.0)
    diff, F0 = base * (1 - m), 0.04 * (1 - m) + base * m
    H = V + L
    H = H / np.linalg.norm(H, axis=-1, keepdims=True)
    NoL, NoV = np.sum(N * L, -1), np.maximum(np.sum(N * V, -1), 1e-4)
    NoH, VoH = np.maximum(np.sum(N * H, -1), 0.0), np.maximum(np.sum(V * H, -1), 0.0)
    F = F0 + (1 - F0) * ((1 - VoH) ** 5)[..., None]
    D = a2 / (np.pi * (NoH * NoH * (a2 - 1) + 1) ** 2)
    k = a / 2
    vis = 1.0 / (4 * (NoL * (1 - k) + k) * (NoV * (1 - k) + k))
    f = diff / np.pi * (1 - F) + (D * vis)[..., None] * F
    return np.where((NoL > 0)[..., None], f, 0.0)


def directional_albedo(base, rough, metal, V, n_theta=2048, n_phi=4096):
    """a(V) = int_hemisphere f(V, L) (N.L) dL for N = +Y: midpoint rule in (cos theta, phi); the GGX peak of the roughest-to-smooth test
    materials is resolved by the grid (alpha >= 0.04: lobe width ~ 0.04 rad against a 0.0015 rad step)"""
    N = np.array([0.0, 1.0, 0.0])
    V = np.asarray(V, np.float64) / np.linalg.norm(V)
    ct = (np.arange(n_theta) + 0.5) / n_theta
    ph = (np.arange(n_phi) + 0.5) / n_phi * 2 * np.pi
    total = np.zeros(3)
    st = np.sqrt(1 - ct * ct)
    for i in range(n_theta):     # one ring of directions at a time: bounded memory
        L = np.stack([st[i] * np.cos(ph), np.full_like(ph, ct[i]), st[i] * np.sin(ph)], axis=-1)
        total += (bsdf(base, rough, metal, N[None], V[None], L) * ct[i]).sum(axis=0)
    return total * (1.0 / n_theta) * (2 * np.pi / n_phi)
