"""Deterministic procedural stand-in scenes (input data only; no integrator arithmetic).

The reference's default scene is DamagedHelmet + Sponza + uffizi-large.hdr, loaded from
hard-coded paths (reference crates/standalone/src/lib.rs:107-126).  None of those assets ship
with the reference or exist on this machine, so BASELINE configs 3-5 run on the stand-ins below
 and every result is labelled as such:

  synthetic_atrium(seed=2) — Sponza stand-in: a two-storey colonnaded hall with arches, drapes,
      a statue and an open roof; exactly 262,144 triangles, 25 materials, ~100 instances, 20 x 1024^2 textures
      (80 MB); camera = the reference's start pose (app.rs:64-67): origin (-10,1,0), dir (1,0.35,0).
  synthetic_helmet(seed=1) — DamagedHelmet stand-in: a displaced sphere (~81k tris with its attachments and the ground),
      5 x 2048^2 textures (84 MB), 5 materials (roughness 0.05-1, metallic 0 and 1), 1024x512 RGBE sky.

A scene description is a plain dict of numpy arrays; `to_product` feeds it through the C ABI
(`Scene.add_mesh / add_instance / ...`); the tests feed the same arrays to the oracle's numpy Scene
(`oracle/harness.py: to_oracle`), so both sides start from identical bytes.
"""
import numpy as np

INVALID = 0xFFFFFFFF


def _grid(nu, nv):
    """(nu+1)*(nv+1) parameter grid in [0,1]^2 and its 2*nu*nv triangles."""
    u, v = np.meshgrid(np.linspace(0, 1, nu + 1, dtype=np.float32), np.linspace(0, 1, nv + 1, dtype=np.float32), indexing="xy")
    i = np.arange(nu * nv, dtype=np.uint32)
    x, y = i % nu, i // nu
    a = y * (nu + 1) + x
    b, c, d = a + 1, a + (nu + 1), a + (nu + 1) + 1
    idx = np.stack([a, b, d, a, d, c], axis=1).reshape(-1).astype(np.uint32)
    return u.reshape(-1), v.reshape(-1), idx


def _normals_from(pos, idx):
    tri = idx.reshape(-1, 3)
    fn = np.cross(pos[tri[:, 1]] - pos[tri[:, 0]], pos[tri[:, 2]] - pos[tri[:, 0]])
    n = np.zeros_like(pos)
    for k in range(3):
        np.add.at(n, tri[:, k], fn)
    l = np.linalg.norm(n, axis=1, keepdims=True)
    l[l == 0] = 1
    return (n / l).astype(np.float32)


def _mesh(pos, idx, uv=None, nrm=None):
    pos = np.ascontiguousarray(pos, np.float32)
    idx = np.ascontiguousarray(idx, np.uint32)
    if nrm is None:
        nrm = _normals_from(pos.astype(np.float64), idx).astype(np.float32)
    if uv is None:
        uv = np.zeros((pos.shape[0], 2), np.float32)
    return {"positions": pos, "normals": np.ascontiguousarray(nrm, np.float32), "uvs": np.ascontiguousarray(uv, np.float32), "indices": idx}


def _plane(nu, nv, origin, eu, ev, uv_scale=1.0):
    u, v, idx = _grid(nu, nv)
    pos = np.asarray(origin, np.float32)[None] + u[:, None] * np.asarray(eu, np.float32)[None] + v[:, None] * np.asarray(ev, np.float32)[None]
    return _mesh(pos, idx, np.stack([u * uv_scale, v * uv_scale], axis=1))


def _column(segs, rings, radius=0.35, height=4.4, flutes=12):
    u, v, idx = _grid(segs, rings)
    ang = u * (2 * np.pi)
    r = radius * (1.0 - 0.12 * v) * (1.0 + 0.035 * np.cos(flutes * ang))
    r = r * (1.0 + 0.25 * np.exp(-((v - 0.0) / 0.04) ** 2) + 0.3 * np.exp(-((v - 1.0) / 0.05) ** 2))
    pos = np.stack([r * np.cos(ang), v * height, r * np.sin(ang)], axis=1)
    return _mesh(pos, idx, np.stack([u * 4, v * 4], axis=1))


def _arch(segs, depth_segs, span=3.0, rise=1.4, thick=0.5, depth=0.8):
    """underside + two faces of a semi-elliptical arch spanning `span` along x"""
    u, v, idx = _grid(segs, depth_segs)
    ang = np.pi * (1.0 - u)
    x = 0.5 * span * np.cos(ang)
    y = rise * np.sin(ang)
    under = np.stack([x, y, (v - 0.5) * depth], axis=1)
    parts_p, parts_i, parts_uv = [under], [idx], [np.stack([u * 3, v], axis=1)]
    off = under.shape[0]
    for sgn in (-1.0, 1.0):
        yy = y + v * (thick + (rise - y) * 1.0)
        face = np.stack([x, yy, np.full_like(x, sgn * 0.5 * depth)], axis=1)
        parts_p.append(face)
        parts_i.append((idx if sgn > 0 else idx.reshape(-1, 3)[:, ::-1].reshape(-1)) + off)
        parts_uv.append(np.stack([u * 3, v], axis=1))
        off += face.shape[0]
    return _mesh(np.concatenate(parts_p), np.concatenate(parts_i), np.concatenate(parts_uv))


def _drape(nu, nv, width, height, rng):
    u, v, idx = _grid(nu, nv)
    ph = rng.uniform(0, 2 * np.pi, 3)
    sag = 0.35 * np.sin(np.pi * u) * (1 - v) * 0.0
    z = 0.18 * np.sin(u * 9 * np.pi + ph[0]) * (0.3 + 0.7 * (1 - v)) + 0.07 * np.sin(u * 23 * np.pi + ph[1] + 3 * v) + 0.05 * np.sin(v * 7 * np.pi + ph[2])
    pos = np.stack([(u - 0.5) * width, v * height - sag, z], axis=1)
    return _mesh(pos, idx, np.stack([u * 2, v * 2], axis=1))


def _displaced_sphere(n_theta, n_phi, radius, rng, amp=0.18):
    u, v, idx = _grid(n_theta, n_phi)
    th, ph = u * 2 * np.pi, v * np.pi
    d = np.stack([np.sin(ph) * np.cos(th), np.cos(ph), np.sin(ph) * np.sin(th)], axis=1)
    k = rng.normal(size=(6, 3)) * 3.0
    phase = rng.uniform(0, 2 * np.pi, 6)
    disp = sum(np.sin(d @ k[i] + phase[i]) / (i + 1) for i in range(6))
    r = radius * (1.0 + amp * disp / 2.5)
    return _mesh(d * r[:, None], idx, np.stack([u * 2, v], axis=1))


def _translate(x, y, z):
    m = np.eye(4, dtype=np.float32)
    m[:3, 3] = (x, y, z)
    return m.T.reshape(16).copy()


def _rot_y_translate(angle, x, y, z, scale=(1, 1, 1)):
    c, s = np.float32(np.cos(angle)), np.float32(np.sin(angle))
    m = np.eye(4, dtype=np.float32)
    m[:3, :3] = np.array([[c, 0, s], [0, 1, 0], [-s, 0, c]], np.float32) * np.asarray(scale, np.float32)[None, :]
    m[:3, 3] = (x, y, z)
    return m.T.reshape(16).copy()


def checker_texture(size, a, b, cells, rng, noise=18):
    y, x = np.mgrid[0:size, 0:size]
    m = (((x * cells) // size + (y * cells) // size) & 1).astype(bool)
    img = np.where(m[..., None], np.asarray(a, np.int32)[None, None], np.asarray(b, np.int32)[None, None]).astype(np.int32)
    img = img + rng.integers(-noise, noise + 1, (size, size, 1))
    out = np.zeros((size, size, 4), np.uint8)
    out[..., :3] = np.clip(img, 0, 255)
    out[..., 3] = 255
    return out


def mra_texture(size, rng, rough_lo=60, rough_hi=230, metal=0):
    y, x = np.mgrid[0:size, 0:size]
    f = 0.5 + 0.5 * np.sin(x * (2 * np.pi * 6 / size)) * np.sin(y * (2 * np.pi * 5 / size))
    out = np.zeros((size, size, 4), np.uint8)
    out[..., 0] = 255
    out[..., 1] = np.clip(rough_lo + (rough_hi - rough_lo) * f + rng.integers(-8, 9, (size, size)), 0, 255)
    out[..., 2] = metal
    out[..., 3] = 255
    return out


def sky_probe(width=512, height=256, sun_dir=(0.35, 0.8, 0.25), sun_power=60.0):
    """procedural RGBE equirect sky: gradient + soft sun (stand-in for uffizi-large.hdr)"""
    v, u = np.mgrid[0:height, 0:width]
    phi = ((u + 0.5) / width - 0.5) * 2 * np.pi
    th = (v + 0.5) / height * np.pi
    d = np.stack([np.sin(th) * np.cos(phi), np.cos(th), np.sin(th) * np.sin(phi)], axis=-1)
    s = np.asarray(sun_dir, np.float64)
    s /= np.linalg.norm(s)
    up = np.clip(d[..., 1], 0, 1)
    sky = np.stack([0.35 + 0.25 * (1 - up), 0.5 + 0.25 * (1 - up), 0.95 - 0.1 * (1 - up)], axis=-1) * (0.6 + 0.8 * up[..., None])
    ground = np.array([0.18, 0.16, 0.14])
    col = np.where(d[..., 1:2] > 0, sky, ground[None, None])
    c = np.clip((d @ s), 0, 1)
    col = col + sun_power * np.exp((c - 1.0) * 900.0)[..., None] * np.array([1.0, 0.93, 0.8])
    m = col.max(axis=-1)
    e = np.ceil(np.log2(np.maximum(m, 1e-30))).astype(np.int32)
    scale = np.exp2(-e.astype(np.float64)) * 256.0
    rgb = np.clip(col * scale[..., None], 0, 255).astype(np.uint8)
    out = np.zeros((height, width, 4), np.uint8)
    out[..., :3] = rgb
    out[..., 3] = np.clip(e + 128, 0, 255).astype(np.uint8)
    out[m < 1e-20] = 0
    return out


def synthetic_atrium(seed=2, target_tris=262144, textures=True, texture_size=1024, shell_quads=False):
    """262,144 triangles, 24 materials (+ the dummy = 25), ~100 instances, 20 procedural
    `texture_size`^2 RGBA8 textures (1024: 80 MB of texels, the size class of Sponza's texture set).
    shell_quads (round 6, VERDICT r05 #6: "hall_large"): floor, walls, roof strips and gallery floors as TWO triangles each instead of fine grids — the triangle-size
    distribution of a real building model (Sponza's walls are a handful of large polygons around finely tessellated ornaments); the budget goes to the statue."""
    rng = np.random.default_rng(seed)
    meshes, instances, materials, images = [], [], [], []

    def add_material(color, rough, metal, albedo=INVALID, mra=INVALID):
        materials.append((tuple(color) + (1.0,), rough, metal, albedo, mra))
        return len(materials)  # index in the final scene (dummy material 0 precedes)

    def add_mesh(m):
        meshes.append(m)
        return len(meshes)  # BLAS index in the final scene (dummy entry 0 precedes)

    tex = {}
    if textures:
        def T(img):
            images.append(img)
            return len(images) - 1
        tex["floor"] = T(checker_texture(texture_size, (200, 190, 170), (120, 100, 90), 16, rng))
        tex["floor_mra"] = T(mra_texture(texture_size, rng, 40, 160))
        tex["wall"] = T(checker_texture(texture_size, (190, 170, 140), (170, 150, 125), 8, rng, 25))
        tex["wall_mra"] = T(mra_texture(texture_size, rng, 150, 250))
        for k, (a, b, cells) in enumerate([((205, 200, 190), (185, 180, 170), 4), ((198, 190, 176), (170, 164, 150), 6),
                                           ((188, 182, 170), (160, 152, 140), 10), ((178, 170, 160), (150, 140, 128), 3)]):
            tex["stone%d" % k] = T(checker_texture(texture_size, a, b, cells, rng, 30))
        tex["stone_mra"] = T(mra_texture(texture_size, rng, 120, 240))
        for k, (a, b) in enumerate([((170, 30, 30), (120, 15, 20)), ((30, 60, 150), (20, 35, 100)), ((30, 120, 50), (15, 80, 35)),
                                    ((190, 150, 40), (150, 110, 20))]):
            tex["drape%d" % k] = T(checker_texture(texture_size, a, b, 24, rng, 12))
        tex["drape_mra"] = T(mra_texture(texture_size, rng, 190, 255))
        for k, (a, b) in enumerate([((160, 120, 90), (120, 85, 60)), ((110, 130, 150), (80, 95, 115)), ((150, 150, 120), (105, 110, 85))]):
            tex["trim%d" % k] = T(checker_texture(texture_size, a, b, 12, rng, 20))
        tex["trim_mra"] = T(mra_texture(texture_size, rng, 80, 220))
        tex["metal_mra"] = T(mra_texture(texture_size, rng, 25, 120, metal=255))
        tex["gloss_mra"] = T(mra_texture(texture_size, rng, 10, 70))
        assert len(images) == 20
    g = lambda k: tex.get(k, INVALID)

    m_floor = add_material((1, 1, 1) if textures else (0.7, 0.65, 0.6), 0.6, 0.0, g("floor"), g("floor_mra"))
    m_wall = add_material((1, 1, 1) if textures else (0.72, 0.65, 0.55), 0.95, 0.0, g("wall"), g("wall_mra"))
    m_stone = [add_material(tuple(np.array((0.8, 0.78, 0.72)) * s) if not textures else (s, s, s), 0.8, 0.0, g("stone%d" % k), g("stone_mra"))
               for k, s in enumerate((1.0, 0.92, 0.85, 0.78))]
    m_drape = [add_material((1, 1, 1) if textures else c, 0.9, 0.0, g("drape%d" % k), g("drape_mra"))
               for k, c in enumerate([(0.65, 0.1, 0.1), (0.1, 0.2, 0.6), (0.1, 0.45, 0.18), (0.7, 0.55, 0.12)])]
    m_metal = [add_material(c, r, 1.0, INVALID, g("metal_mra") if k % 2 == 0 else INVALID)
               for k, (c, r) in enumerate([((0.95, 0.78, 0.35), 0.25), ((0.9, 0.9, 0.92), 0.12), ((0.72, 0.45, 0.2), 0.4), ((0.56, 0.57, 0.58), 0.05)])]
    m_gloss = [add_material(c, r, 0.0, INVALID, g("gloss_mra") if k % 2 == 1 else INVALID)
               for k, (c, r) in enumerate([((0.1, 0.1, 0.12), 0.08), ((0.6, 0.08, 0.06), 0.15), ((0.9, 0.9, 0.88), 0.2), ((0.05, 0.25, 0.3), 0.05)])]
    m_trim = [add_material((1, 1, 1) if textures else tuple(rng.uniform(0.25, 0.85, 3)), float(rng.uniform(0.3, 1.0)), 0.0, g("trim%d" % (k % 3)), g("trim_mra") if k < 3 else INVALID)
              for k in range(6)]

    X0, X1, Z0, Z1, HY = -14.0, 14.0, -6.0, 6.0, 11.0
    ident = _translate(0, 0, 0)
    _fine_plane = globals()["_plane"]

    def _plane(nu, nv, *a):   # the shell's planes below: one quad each on request (shadows the module's _plane inside this function)
        return _fine_plane(1, 1, *a) if shell_quads else _fine_plane(nu, nv, *a)
    # shell: floor, two long walls, two end walls, roof rim (open to the sky in the middle)
    b = add_mesh(_plane(96, 48, (X0, 0, Z0), (X1 - X0, 0, 0), (0, 0, Z1 - Z0), 14))
    instances.append((b, ident, m_floor))
    b = add_mesh(_plane(96, 40, (X0, 0, Z0), (0, HY, 0), (X1 - X0, 0, 0), 10))
    instances.append((b, ident, m_wall))
    b = add_mesh(_plane(96, 40, (X0, 0, Z1), (X1 - X0, 0, 0), (0, HY, 0), 10))
    instances.append((b, ident, m_wall))
    b = add_mesh(_plane(40, 40, (X0, 0, Z0), (0, 0, Z1 - Z0), (0, HY, 0), 5))
    instances.append((b, ident, m_wall))
    b = add_mesh(_plane(40, 40, (X1, 0, Z0), (0, HY, 0), (0, 0, Z1 - Z0), 5))
    instances.append((b, ident, m_wall))
    for z0, z1 in ((Z0, -3.2), (3.2, Z1)):  # roof strips over the side aisles
        b = add_mesh(_plane(96, 12, (X0, HY, z0), (0, 0, z1 - z0), (X1 - X0, 0, 0), 8))
        instances.append((b, ident, m_stone[1]))
    # gallery floors (second storey) over the aisles
    for z0, z1 in ((Z0, -3.2), (3.2, Z1)):
        b = add_mesh(_plane(96, 12, (X0, 5.4, z0), (X1 - X0, 0, 0), (0, 0, z1 - z0), 8))
        instances.append((b, ident, m_stone[2]))
        b = add_mesh(_plane(96, 12, (X0, 5.2, z0), (0, 0, z1 - z0), (X1 - X0, 0, 0), 8))
        instances.append((b, ident, m_stone[2]))
    # columns: one BLAS, instanced along both aisles on both storeys
    col = add_mesh(_column(40, 24))
    col_hi = add_mesh(_column(40, 24, radius=0.28, height=4.2, flutes=16))
    arch = add_mesh(_arch(28, 6))
    xs = np.linspace(X0 + 2.0, X1 - 2.0, 9)
    for zi, z in enumerate((-3.2, 3.2)):
        for xi, x in enumerate(xs):
            instances.append((col, _rot_y_translate(0.37 * xi + zi, x, 0.0, z), m_stone[(xi + zi) % 4]))
            instances.append((col_hi, _rot_y_translate(0.21 * xi + zi, x, 5.4, z), m_stone[(xi + zi + 1) % 4]))
        for xi in range(len(xs) - 1):
            xm = 0.5 * (xs[xi] + xs[xi + 1])
            span = (xs[xi + 1] - xs[xi]) / 3.0
            instances.append((arch, _rot_y_translate(0.0, xm, 3.8, z, (span, 1.0, 1.0)), m_trim[xi % 6]))
            instances.append((arch, _rot_y_translate(0.0, xm, 9.0, z, (span, 0.8, 1.0)), m_trim[(xi + 3) % 6]))
    # drapes hanging from the gallery, one BLAS each (different folds)
    for k in range(6):
        d = add_mesh(_drape(96, 56, 2.6, 3.6, rng))
        x = X0 + 4.0 + 4.2 * k
        z = -2.7 if k % 2 == 0 else 2.7
        instances.append((d, _rot_y_translate(0.0 if k % 2 == 0 else np.pi, x, 1.5, z), m_drape[k % 4]))
    # objects on the floor: metal / glossy spheres and a central statue
    orb = add_mesh(_displaced_sphere(48, 24, 0.6, rng, amp=0.0))
    for k in range(8):
        x = X0 + 3.0 + 3.3 * k
        z = (-1.2, 1.3, -0.4, 0.9)[k % 4]
        mat = (m_metal + m_gloss)[k % 8]
        instances.append((orb, _rot_y_translate(0.3 * k, x, 0.6, z), mat))
    used = sum(meshes[bi - 1]["indices"].size // 3 for bi, _, _ in instances)
    remaining = target_tris - used
    if remaining < 2000:
        raise ValueError("triangle budget exceeded: %d used" % used)
    # statue: a displaced sphere sized to land EXACTLY on the triangle budget
    nt = int(np.sqrt(remaining))
    while nt > 8 and (remaining // 2) % nt != 0:
        nt -= 1
    if shell_quads:     # (the divisor search above may end at a handful of rings of needle triangles for this budget: keep the statue's triangles well shaped, the plinth takes the rest)
        nt = int(np.sqrt(remaining // 2))
    nphi = (remaining // 2) // nt
    filler = remaining - 2 * nt * nphi
    statue = add_mesh(_displaced_sphere(nt, nphi, 1.3, rng, amp=0.35))
    instances.append((statue, _rot_y_translate(0.6, 2.0, 1.9, 0.0, (1.0, 1.45, 1.0)), m_metal[0]))
    if filler:
        u, v, idx = _grid(filler // 2 if filler > 1 else 1, 1)
        plinth = _fine_plane(max(filler // 2, 1), 1, (1.0, 0.02, -1.0), (2.0, 0, 0), (0, 0, 2.0))
        plinth["indices"] = plinth["indices"][: 3 * filler]
        pb = add_mesh(plinth)
        instances.append((pb, ident, m_stone[0]))
    total = sum(meshes[bi - 1]["indices"].size // 3 for bi, _, _ in instances)
    assert total == target_tris, (total, target_tris)

    light = np.zeros(1, dtype=[("normal", "<f4", 4), ("tangent", "<f4", 4), ("bitangent", "<f4", 4), ("origin", "<f4", 4)])
    light["normal"] = (0, -1, 0, 0)
    light["tangent"] = (1, 0, 0, 5.0)
    light["bitangent"] = (0, 0, 1, 1.5)
    light["origin"] = (0.0, 10.9, 0.0, 18.0)
    return {"name": "synthetic_atrium(seed=%d)" % seed, "meshes": meshes, "instances": instances, "materials": materials,
            "images": images, "lights": [light], "probe": sky_probe(1024, 512), "triangles": total,
            "camera": {"origin": (-10.0, 1.0, 0.0), "direction": (1.0, 0.35, 0.0)}}


def synthetic_helmet(seed=1, textures=True, texture_size=2048):
    """DamagedHelmet stand-in — a displaced sphere (69,696 triangles) with three attachments on a ground
    plane, 5 procedural `texture_size`^2 RGBA8 textures (2048: 84 MB), 5 materials spanning roughness 0.05-1 and both
    metallic values, a 1024x512 RGBE sky"""
    rng = np.random.default_rng(seed)
    sphere = _displaced_sphere(264, 132, 1.0, rng, amp=0.3)  # 69,696 triangles
    images, materials = [], []
    t = [INVALID] * 5
    if textures:
        images.append(checker_texture(texture_size, (200, 200, 205), (60, 70, 90), 32, rng, 20))       # shell: base colour
        images.append(mra_texture(texture_size, rng, 20, 255, metal=255))                               # shell: roughness 0.08-1, metal
        images.append(checker_texture(texture_size, (150, 140, 125), (95, 90, 80), 48, rng, 25))       # ground: base colour
        images.append(mra_texture(texture_size, rng, 200, 255))                                         # ground: rough dielectric
        images.append(checker_texture(texture_size, (230, 120, 40), (40, 40, 45), 8, rng, 10))         # visor / attachments
        t = [0, 1, 2, 3, 4]
    materials.append(((1, 1, 1, 1), 1.0, 1.0, t[0], t[1]))                    # 1 shell: textured metal
    materials.append(((1, 1, 1, 1) if textures else (0.55, 0.5, 0.45, 1), 1.0, 0.0, t[2], t[3]))   # 2 ground: rough dielectric
    materials.append(((1, 1, 1, 1) if textures else (0.8, 0.4, 0.15, 1), 0.05, 0.0, t[4], INVALID))  # 3 visor: smooth dielectric
    materials.append(((0.95, 0.93, 0.88, 1), 0.3, 1.0, INVALID, INVALID))      # 4 polished metal
    materials.append(((0.8, 0.3, 0.2, 1), 0.6, 0.0, INVALID, INVALID))         # 5 painted dielectric
    plane = _plane(64, 64, (-6, -1.4, -6), (12, 0, 0), (0, 0, 12), 6)
    knob = _displaced_sphere(32, 16, 0.28, rng, amp=0.1)                       # 1,024 triangles, instanced three times
    light = np.zeros(1, dtype=[("normal", "<f4", 4), ("tangent", "<f4", 4), ("bitangent", "<f4", 4), ("origin", "<f4", 4)])
    light["normal"] = (0, -1, 0, 0)
    light["tangent"] = (1, 0, 0, 1.0)
    light["bitangent"] = (0, 0, 1, 1.0)
    light["origin"] = (0.0, 4.0, 0.0, 25.0)
    instances = [(1, _translate(0, 0, 0), 1), (2, _translate(0, 0, 0), 2),
                 (3, _rot_y_translate(0.0, 0.0, 0.1, 1.02, (1.6, 0.9, 0.5)), 3),       # visor, towards the camera
                 (3, _rot_y_translate(0.7, 1.12, 0.05, 0.1), 4), (3, _rot_y_translate(-0.7, -1.12, 0.05, 0.1), 5)]
    meshes = [sphere, plane, knob]
    total = sum(meshes[b - 1]["indices"].size // 3 for b, _, _ in instances)
    return {"name": "synthetic_helmet(seed=%d)" % seed, "meshes": meshes, "instances": instances, "materials": materials, "images": images,
            "lights": [light], "probe": sky_probe(1024, 512), "triangles": total,
            "camera": {"origin": (0.0, 0.6, 4.2), "direction": (0.0, -0.12, -1.0)}}


def synthetic_hall(seed=3, gravel=12000, needles=160, stack=40, telescope=90):
    """A MIXED-SCALE scene (VERDICT r04 "missing" 7: the stand-ins have uniformly small triangles): a hall of six walls of TWO triangles each (16 units across),
    a pile of `gravel` triangles 2-3 cm across in one corner, `needles` slivers 6-14 units long and a centimetre wide through the whole volume, the same quad
    `stack` times in the same place (coincident triangles: equal hit distances, the tie goes to the lowest primitive id; and a subtree no split can separate),
    and a "telescope" of `telescope` triangles sharing a corner at scales 0.93^k (a tree the builder can only peel one triangle at a time: depth).  Untextured
    materials, one emitter, the sky probe through an open roof panel."""
    rng = np.random.default_rng(seed)
    def soup(tris):   # (n, 3, 3) -> mesh with unshared vertices
        pos = np.asarray(tris, np.float32).reshape(-1, 3)
        return _mesh(pos, np.arange(pos.shape[0], dtype=np.uint32))
    def quad(o, eu, ev):
        o, eu, ev = (np.asarray(x, np.float32) for x in (o, eu, ev))
        return [[o, o + eu, o + eu + ev], [o, o + eu + ev, o + ev]]
    H = 8.0
    walls = (quad((-H, 0, -H), (0, 0, 2 * H), (2 * H, 0, 0)) + quad((-H, 0, -H), (2 * H, 0, 0), (0, 9, 0)) + quad((H, 0, H), (-2 * H, 0, 0), (0, 9, 0))
             + quad((-H, 0, H), (0, 0, -2 * H), (0, 9, 0)) + quad((H, 0, -H), (0, 0, 2 * H), (0, 9, 0)) + quad((-H, 9, -H), (2 * H, 0, 0), (0, 0, 1.2 * H)))   # the roof leaves a strip open
    c = rng.uniform((-7.5, 0.0, -7.5), (-4.5, 1.6, -4.5), (gravel, 1, 3)).astype(np.float32)
    c[:, 0, 1] *= rng.uniform(0, 1, gravel).astype(np.float32)   # a pile: denser near the floor
    grav = c + rng.normal(0, 0.012, (gravel, 3, 3)).astype(np.float32)
    a0 = rng.uniform((-H, 0.2, -H), (H, 8.5, H), (needles, 3)).astype(np.float32)
    dirn = rng.normal(size=(needles, 3)).astype(np.float32)
    dirn /= np.linalg.norm(dirn, axis=1, keepdims=True)
    side = np.cross(dirn, rng.normal(size=(needles, 3))).astype(np.float32)
    side /= np.linalg.norm(side, axis=1, keepdims=True)
    length = rng.uniform(6, 14, (needles, 1)).astype(np.float32)
    ndl = np.stack([a0, a0 + dirn * length, a0 + dirn * length * 0.5 + side * 0.01], axis=1)
    one = np.asarray(quad((1.0, 0.8, -2.0), (2.5, 0, 0.4), (0, 2.0, 0.3)), np.float32)
    stk = np.concatenate([one] * stack + [one[:1] * np.float32(1.0)] * stack, axis=0)         # 3 x stack coincident triangles
    corner = np.asarray((5.5, 0.05, 5.5), np.float32)
    tele = np.stack([np.stack([corner, corner + np.float32(0.93 ** k) * np.asarray((-4.0, 0.3, 0.2), np.float32),
                               corner + np.float32(0.93 ** k) * np.asarray((0.2, 3.5 + 0.01 * k, -4.0), np.float32)]) for k in range(telescope)])
    meshes = [soup(walls), soup(grav), soup(ndl), soup(stk), soup(tele)]
    materials = [((0.72, 0.7, 0.66, 1), 0.9, 0.0, INVALID, INVALID), ((0.5, 0.42, 0.3, 1), 0.7, 0.0, INVALID, INVALID), ((0.9, 0.9, 0.92, 1), 0.15, 1.0, INVALID, INVALID),
                 ((0.8, 0.25, 0.2, 1), 0.4, 0.0, INVALID, INVALID), ((0.3, 0.5, 0.8, 1), 0.25, 0.0, INVALID, INVALID)]
    ident = _translate(0, 0, 0)
    instances = [(k + 1, ident, k + 1) for k in range(5)]
    light = np.zeros(1, dtype=[("normal", "<f4", 4), ("tangent", "<f4", 4), ("bitangent", "<f4", 4), ("origin", "<f4", 4)])
    light["normal"] = (0, -1, 0, 0)
    light["tangent"] = (1, 0, 0, 1.5)
    light["bitangent"] = (0, 0, 1, 1.5)
    light["origin"] = (0.0, 8.6, -1.0, 30.0)
    total = sum(m["indices"].size // 3 for m in meshes)
    return {"name": "synthetic_hall(seed=%d)" % seed, "meshes": meshes, "instances": instances, "materials": materials, "images": [], "lights": [light],
            "probe": sky_probe(256, 128), "triangles": total, "camera": {"origin": (6.5, 2.2, -6.5), "direction": (-0.75, -0.12, 0.7)}}


def origin_dust(seed=4, n=3000):
    """SPEC §7's margin as a scene: `n` millimetre-sized triangles around the WORLD ORIGIN of a scene 2 000 units across (two far triangles set the extent).  The Woop test's
    rounding grows with the ray's coordinates, a triangle's padding with the triangle's own: rays from far away that graze these triangles are where a tree would lose a hit."""
    rng = np.random.default_rng(seed)
    tiny = np.zeros((0, 3, 3), np.float32)
    while tiny.shape[0] < n:   # well-shaped triangles only: the Woop map of a sliver (three nearly collinear vertices) is ill-conditioned, and from a million triangle sizes
        # away its rounding accepts rays that pass several triangle LENGTHS beyond the sliver's tip — no box padding is a margin for that (profiles/r05_experiments_ab.txt U)
        c = rng.uniform(-0.05, 0.05, (n, 1, 3)).astype(np.float32)
        t = (c + rng.normal(0, 1e-3, (n, 3, 3))).astype(np.float32)
        e = np.stack([t[:, 1] - t[:, 0], t[:, 2] - t[:, 1], t[:, 0] - t[:, 2]], axis=1).astype(np.float64)
        longest = np.linalg.norm(e, axis=2).max(axis=1)
        area2 = np.linalg.norm(np.cross(e[:, 0], -e[:, 2]), axis=1)
        tiny = np.concatenate([tiny, t[area2 / longest >= 0.35 * longest]])[:n]      # altitude over the longest edge >= 0.35 of it
    far = np.array([[[-1000, -1000, -1000], [-999, -1000, -1000], [-1000, -999, -1000]], [[1000, 1000, 1000], [999, 1000, 1000], [1000, 999, 1000]]], np.float32)
    tris = np.concatenate([tiny, far])
    mesh = _mesh(tris.reshape(-1, 3), np.arange(tris.shape[0] * 3, dtype=np.uint32))
    light = np.zeros(1, dtype=[("normal", "<f4", 4), ("tangent", "<f4", 4), ("bitangent", "<f4", 4), ("origin", "<f4", 4)])
    light["normal"] = (0, -1, 0, 0)
    light["tangent"] = (1, 0, 0, 1.0)
    light["bitangent"] = (0, 0, 1, 1.0)
    light["origin"] = (0.0, 500.0, 0.0, 10.0)
    return {"name": "origin_dust(seed=%d)" % seed, "meshes": [mesh], "instances": [(1, _translate(0, 0, 0), 1)], "materials": [((0.8, 0.8, 0.8, 1), 0.5, 0.0, INVALID, INVALID)],
            "images": [], "lights": [light], "probe": sky_probe(64, 32), "triangles": int(tris.shape[0]), "camera": {"origin": (0.0, 0.0, 3.0), "direction": (0.0, 0.0, -1.0)},
            "dust": tiny}


def grazing_rays(dust, m, seed=5, extent=1000.0):
    """`m` rays towards vertices (half of them) and random points (many near an edge) of random triangles of `dust` (k, 3, 3), a third from nearby, the rest from up to
    `extent` away; returns origins, unit directions and the distance to the point aimed at"""
    rng = np.random.default_rng(seed)
    o = rng.uniform(-extent, extent, (m, 3)).astype(np.float32)
    o[: m // 3] = rng.uniform(-2, 2, (m // 3, 3)).astype(np.float32)
    t = rng.integers(0, dust.shape[0], m)
    w = rng.dirichlet((0.6, 0.6, 0.6), m).astype(np.float32)
    w[: m // 2] = np.eye(3, dtype=np.float32)[rng.integers(0, 3, m // 2)]
    target = (dust[t] * w[:, :, None]).sum(axis=1).astype(np.float32)
    d = (target - o).astype(np.float32)
    dist = np.linalg.norm(d, axis=1).astype(np.float32)
    d /= dist[:, None]
    return o, d.astype(np.float32), dist


def to_product(desc):
    """feed a description through the C ABI; returns loupiote_amd.Scene"""
    from . import api
    s = api.Scene()
    for m in desc["meshes"]:
        s.add_mesh(m["positions"], m["normals"], m["uvs"], m["indices"])
    for color, rough, metal, at, mt in desc["materials"]:
        s.add_material(color, rough, metal, at if at != INVALID else INVALID, mt if mt != INVALID else INVALID)
    for img in desc["images"]:
        s.add_image(img)
    for blas, mat16, material in desc["instances"]:
        s.add_instance(blas, mat16, material)
    for i, l in enumerate(desc["lights"]):
        if i == 0:
            s.set_light(0, l)
        else:
            s.add_light(l)
    return s
