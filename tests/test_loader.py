"""CPU: glTF loader of the product (lpt_load_gltf, C++) against the oracle's numpy restatement of
reference crates/lib/src/loaders/gltf.rs:46-156 — byte-exact flat arrays — plus the error
behaviour (Error::FileNotFound on every parse failure, gltf.rs:49-53) and append semantics."""
import base64
import io
import json
import struct

import numpy as np
import pytest

import loupiote_amd as lp
from oracle import gltf_oracle as G

ARRAYS = ["materials", "entries", "vertices", "indices", "instances", "lights"]


def both(data, pre=None):
    s, o = lp.Scene(), G.Scene()
    if pre is not None:
        lp.loaders.load_gltf(pre, s)
        G.load_gltf(pre, o)
    lp.loaders.load_gltf(data, s)
    G.load_gltf(data, o)
    return s, o


def assert_same(s, o):
    for name in ARRAYS:
        a, b = getattr(s, name), getattr(o, name)
        assert a.shape == b.shape, name
        assert a.tobytes() == b.tobytes(), name
    assert s.counts().images == len(o.images)
    for i, img in enumerate(o.images):
        assert np.array_equal(s.image(i), img)


def make_gltf(meshes, nodes, materials=(), images=(), textures=(), glb=False):
    """tiny glTF writer: meshes = [[prim dict(pos, nrm?, uv?, idx?, mode?, material?, idx_type?)]]"""
    blob = bytearray()
    views, accessors = [], []

    def add(arr, ctype, atype, stride=None):
        while len(blob) % 4:
            blob.append(0)
        raw = np.ascontiguousarray(arr).tobytes()
        view = {"buffer": 0, "byteOffset": len(blob), "byteLength": len(raw)}
        if stride:
            view["byteStride"] = stride
        views.append(view)
        blob.extend(raw)
        accessors.append({"bufferView": len(views) - 1, "componentType": ctype, "count": int(np.asarray(arr).shape[0]), "type": atype})
        return len(accessors) - 1

    jm = []
    for prims in meshes:
        jp = []
        for p in prims:
            attrs = {}
            if "pos" in p:
                attrs["POSITION"] = add(np.asarray(p["pos"], "<f4"), 5126, "VEC3")
            if "nrm" in p:
                attrs["NORMAL"] = add(np.asarray(p["nrm"], "<f4"), 5126, "VEC3")
            if "uv" in p:
                attrs["TEXCOORD_0"] = add(np.asarray(p["uv"], "<f4"), 5126, "VEC2")
            d = {"attributes": attrs}
            if "idx" in p:
                dt, ct = {"u8": ("u1", 5121), "u16": ("<u2", 5123), "u32": ("<u4", 5125)}[p.get("idx_type", "u16")]
                d["indices"] = add(np.asarray(p["idx"], dt), ct, "SCALAR")
            if "mode" in p:
                d["mode"] = p["mode"]
            if "material" in p:
                d["material"] = p["material"]
            jp.append(d)
        jm.append({"primitives": jp})
    jimages = []
    for img in images:
        while len(blob) % 4:
            blob.append(0)
        views.append({"buffer": 0, "byteOffset": len(blob), "byteLength": len(img)})
        blob.extend(img)
        jimages.append({"bufferView": len(views) - 1, "mimeType": "image/png"})
    js = {"asset": {"version": "2.0"}, "meshes": jm, "nodes": list(nodes), "accessors": accessors, "bufferViews": views,
          "materials": list(materials), "images": jimages, "textures": [{"source": t} for t in textures]}
    if glb:
        js["buffers"] = [{"byteLength": len(blob)}]
        j = json.dumps(js).encode()
        j += b" " * (-len(j) % 4)
        b = bytes(blob) + b"\0" * (-len(blob) % 4)
        return struct.pack("<III", 0x46546C67, 2, 12 + 8 + len(j) + 8 + len(b)) + struct.pack("<II", len(j), 0x4E4F534A) + j + struct.pack("<II", len(b), 0x004E4942) + b
    js["buffers"] = [{"byteLength": len(blob), "uri": "data:application/octet-stream;base64," + base64.b64encode(bytes(blob)).decode()}]
    return json.dumps(js).encode()


def png_bytes(arr):
    from PIL import Image
    buf = io.BytesIO()
    Image.fromarray(arr).save(buf, format="PNG")
    return buf.getvalue()


QUAD = np.array([[0, 0, 0], [1, 0, 0], [1, 1, 0], [0, 1, 0]], np.float32)
QN = np.tile(np.array([[0, 0, 1]], np.float32), (4, 1))
QUV = np.array([[0, 0], [1, 0], [1, 1], [0, 1]], np.float32)


def test_cornell_box_matches_oracle_loader(cornell_glb):
    s, o = both(cornell_glb)
    assert_same(s, o)
    c = s.counts()
    # 5 primitives -> 5 BLAS entries + 5 instances + 3 materials, after the dummies
    assert (c.entries, c.instances, c.materials, c.images) == (6, 6, 4, 0)
    assert c.indices == 34 * 3 and c.vertices == 103
    m = s.materials
    assert np.allclose(m["color"][2], (0, 1, 0, 1)) and m["albedo_texture"][1] == lp.INVALID_INDEX
    assert np.isclose(m["roughness"][1], 0.4) and m["reflectivity"][1] == 0


def test_append_semantics_offsets(cornell_glb):
    """loading twice appends: bvh_offset / mat_offset are taken before the load (gltf.rs:60,109-110)"""
    s, o = both(cornell_glb, pre=cornell_glb)
    assert_same(s, o)
    inst = s.instances
    assert list(inst["blas_index"]) == [0] + list(range(1, 11))
    assert list(inst["material_index"][6:]) == [4, 4, 4, 5, 6]


@pytest.mark.parametrize("glb", [False, True])
def test_synthetic_variants_match_oracle(glb):
    rng = np.random.default_rng(3)
    img = rng.integers(0, 256, (5, 7, 3), dtype=np.uint8)      # RGB: alpha stays 0 (gltf.rs:26-38)
    img2 = rng.integers(0, 256, (4, 4, 4), dtype=np.uint8)
    strip = np.array([[0, 0, 1], [1, 0, 1], [0, 1, 1], [1, 1, 1], [0, 2, 1]], np.float32)
    meshes = [
        [{"pos": QUAD, "nrm": QN, "uv": QUV, "idx": [0, 1, 2, 0, 2, 3], "material": 0},
         {"pos": QUAD + 2, "idx": [0, 1, 2, 0, 2, 3], "idx_type": "u32", "material": 1},           # no normals -> generated
         {"nrm": QN},                                                                                # no POSITION -> skipped
         {"pos": QUAD, "mode": 1},                                                                   # LINES -> skipped
         {"pos": QUAD - 3, "nrm": QN}],                                                              # non-indexed (count 4 -> 1 tri), no material
        [{"pos": strip, "mode": 5, "idx": [0, 1, 2, 3, 4], "idx_type": "u8"},
         {"pos": strip, "mode": 6}],
    ]
    nodes = [{"mesh": 0, "translation": [1, 2, 3], "rotation": [0.1825742, 0.3651484, 0.5477226, 0.7302967], "scale": [1, 2, 0.5]},
             {"mesh": 1, "matrix": [1, 0, 0, 0, 0, 0, 1, 0, 0, -1, 0, 0, 4, 5, 6, 1]},
             {"name": "empty"},
             {"mesh": 0}]
    materials = [{"pbrMetallicRoughness": {"baseColorFactor": [0.2, 0.4, 0.6, 1], "roughnessFactor": 0.3, "metallicFactor": 0.9,
                                           "baseColorTexture": {"index": 1}, "metallicRoughnessTexture": {"index": 0}}},
                 {}]
    data = make_gltf(meshes, nodes, materials, images=[png_bytes(img), png_bytes(img2)], textures=[1, 0], glb=glb)
    s, o = both(data)
    assert_same(s, o)
    c = s.counts()
    assert c.entries == 1 + 5 and c.instances == 1 + 3 + 2 + 3
    m = s.materials
    assert m["albedo_texture"][1] == 0 and m["mra_texture"][1] == 1          # through textures[i].source
    assert m["roughness"][2] == 1.0 and m["reflectivity"][2] == 1.0           # glTF defaults
    inst = s.instances
    assert list(inst["blas_index"][1:4]) == [1, 2, 3] and list(inst["material_index"][1:4]) == [1, 2, 0]
    assert np.array_equal(s.image(0)[..., 3], np.zeros((5, 7), np.uint8))
    # a failed load leaves the scene untouched
    before = {n: getattr(s, n).tobytes() for n in ARRAYS}
    with pytest.raises(lp.Error):
        lp.loaders.load_gltf(data[: len(data) // 2], s)
    assert before == {n: getattr(s, n).tobytes() for n in ARRAYS}


@pytest.mark.parametrize("bad", [b"", b"garbage", b"glTF\x02\x00\x00\x00\x10\x00\x00\x00", b'{"asset":{}, "meshes": [{"primitives": [{"attributes": {"POSITION": 3}}]}]}'])
def test_parse_failures_are_file_not_found(bad):
    s = lp.Scene()
    with pytest.raises(lp.Error) as e:
        lp.loaders.load_gltf(bad, s)
    assert e.value.kind == "FileNotFound" and str(e.value).startswith("file not found")
    with pytest.raises(FileNotFoundError):
        G.load_gltf(bad, G.Scene())


def _hostile(mutate):
    js = json.loads(make_gltf([[dict(pos=QUAD, nrm=QN, uv=QUV, idx=[0, 1, 2, 0, 2, 3], material=0)]], [{"mesh": 0}],
                              [{"pbrMetallicRoughness": {"baseColorTexture": {"index": 0}}}], images=[png_bytes(np.zeros((4, 4, 4), np.uint8))], textures=[0]))
    mutate(js)
    return json.dumps(js).encode()


@pytest.mark.parametrize("mutate", [
    lambda j: j["bufferViews"][0].__setitem__("byteOffset", -16),                       # negative offset wraps through size_t
    lambda j: j["accessors"][0].__setitem__("byteOffset", -4),
    lambda j: j["accessors"][0].__setitem__("byteOffset", 1.8446744073709552e19),       # 2^64: off + span wraps to a small number
    lambda j: j["accessors"][0].__setitem__("count", 2 ** 61),                          # (count - 1) * stride overflows
    lambda j: j["accessors"][0].__setitem__("count", -1),
    lambda j: j["accessors"][0].__setitem__("count", 2.5),
    lambda j: j["bufferViews"][0].__setitem__("byteStride", 2 ** 62),
    lambda j: j["accessors"][3].__setitem__("count", 10 ** 7),                          # indices past the buffer
    lambda j: j["accessors"][0].__setitem__("bufferView", 99),
    lambda j: j["accessors"][0].__setitem__("bufferView", -1),
    lambda j: j["images"][0].__setitem__("bufferView", 99),                             # was never range-checked
    lambda j: j["images"][0].__setitem__("bufferView", -3),
    lambda j: j["bufferViews"][4].__setitem__("byteOffset", -8),
    lambda j: j["bufferViews"][4].__setitem__("byteLength", 1.8446744073709552e19),
    lambda j: j["bufferViews"][4].__setitem__("byteLength", -1),
    lambda j: j["textures"][0].__setitem__("source", 5),
    lambda j: j["bufferViews"][0].__setitem__("buffer", 3),
])
def test_hostile_offsets_and_counts_are_rejected(mutate):
    """ADVICE r1 (gltf.cpp:119,175): untrusted byteOffset / count / byteLength / indices must not wrap the bounds checks;
    every such file is a FileNotFound and the scene stays as it was (run under ASan on the CPU build during development)"""
    s = lp.Scene()
    before = s.counts()
    with pytest.raises(lp.Error) as e:
        lp.loaders.load_gltf(_hostile(mutate), s)
    assert e.value.kind == "FileNotFound"
    after = s.counts()
    assert (after.vertices, after.entries, after.images) == (before.vertices, before.entries, before.images)


def test_hostile_baseline_still_loads():
    s = lp.Scene()
    lp.loaders.load_gltf(_hostile(lambda j: None), s)
    assert s.counts().entries == 2 and s.counts().images == 1


def test_load_gltf_path(tmp_path, cornell_glb):
    p = tmp_path / "c.glb"
    p.write_bytes(cornell_glb)
    s = lp.Scene()
    lp.loaders.load_gltf_path(p, s)
    assert s.counts().entries == 6
    with pytest.raises(lp.Error) as e:
        lp.loaders.load_gltf_path(tmp_path / "missing.glb", s)
    assert e.value.kind == "FileNotFound"


def test_png_decoder_matches_pil():
    rng = np.random.default_rng(11)
    for shape in [(33, 17, 4), (16, 16, 3), (9, 31)]:
        arr = rng.integers(0, 256, shape, dtype=np.uint8)
        data = make_gltf([[{"pos": QUAD}]], [{"mesh": 0}], images=[png_bytes(arr)], textures=[0])
        s = lp.Scene()
        lp.loaders.load_gltf(data, s)
        got = s.image(0)
        want = np.zeros(arr.shape[:2] + (4,), np.uint8)
        want[..., : (arr.shape[2] if arr.ndim == 3 else 1)] = arr if arr.ndim == 3 else arr[..., None]
        assert np.array_equal(got, want)


def _smooth_rgb(h=97, w=131):
    y, x = np.mgrid[0:h, 0:w]
    return np.stack([127 + 100 * np.sin(x / 9.0) * np.cos(y / 13.0), 127 + 90 * np.cos(x / 17.0 + y / 11.0),
                     60 + x * 1.2 + y * 0.3], axis=-1).clip(0, 255).astype(np.uint8)


def _jpeg(arr, **kw):
    from PIL import Image
    buf = io.BytesIO()
    Image.fromarray(arr).save(buf, format="JPEG", **kw)
    return buf.getvalue()


@pytest.mark.parametrize("kw,mean_tol,max_tol", [
    (dict(quality=92, subsampling=0), 0.1, 4),                          # 4:4:4: only IDCT rounding differs
    (dict(quality=92, subsampling=1), 2.5, 16),                         # 4:2:2: + replicate vs fancy upsampling
    (dict(quality=92, subsampling=2), 3.0, 24),                         # 4:2:0
    (dict(quality=60, subsampling=0, restart_marker_blocks=5), 0.2, 4), # DRI / RSTn
    (dict(quality=85, subsampling=2, restart_marker_rows=1), 3.0, 24),
    (dict(quality=92, subsampling=0, progressive=True), 0.1, 4),        # SOF2: spectral selection + successive approximation
    (dict(quality=75, subsampling=2, progressive=True), 3.0, 24),
    (dict(quality=50, subsampling=1, progressive=True, optimize=True), 2.5, 16),
    (dict(quality=85, subsampling=2, progressive=True, restart_marker_rows=1), 3.0, 24),
])
def test_jpeg_decoder_close_to_pil(kw, mean_tol, max_tol):
    """Baseline JPEG textures (DamagedHelmet / Sponza): the product decoder against libjpeg (PIL) — float IDCT
    and replicated chroma, so a stated tolerance instead of bit-exactness (SPEC §14.5)."""
    from PIL import Image
    raw = _jpeg(_smooth_rgb(), **kw)
    ref = np.asarray(Image.open(io.BytesIO(raw)).convert("RGB")).astype(int)
    s = lp.Scene()
    lp.loaders.load_gltf(make_gltf([[{"pos": QUAD}]], [{"mesh": 0}], images=[raw], textures=[0]), s)
    got = s.image(0)
    assert got.shape == (97, 131, 4) and np.all(got[..., 3] == 0)       # RGB -> RGBA with alpha 0 (gltf.rs:26-38)
    d = np.abs(got[..., :3].astype(int) - ref)
    assert d.mean() < mean_tol and d.max() <= max_tol


def test_jpeg_grey_and_unsupported_modes():
    from PIL import Image
    grey = _smooth_rgb()[..., 0]
    raw = _jpeg(grey, quality=90)
    s = lp.Scene()
    lp.loaders.load_gltf(make_gltf([[{"pos": QUAD}]], [{"mesh": 0}], images=[raw], textures=[0]), s)
    got = s.image(0)
    ref = np.asarray(Image.open(io.BytesIO(raw))).astype(int)
    assert np.abs(got[..., 0].astype(int) - ref).max() <= 2 and np.all(got[..., 1:] == 0)   # R8 -> (r,0,0,0)
    prog = _jpeg(grey, quality=90, progressive=True)      # progressive grey: decodes like the baseline file of the same image
    s2 = lp.Scene()
    lp.loaders.load_gltf(make_gltf([[{"pos": QUAD}]], [{"mesh": 0}], images=[prog], textures=[0]), s2)
    ref2 = np.asarray(Image.open(io.BytesIO(prog))).astype(int)
    assert np.abs(s2.image(0)[..., 0].astype(int) - ref2).max() <= 2
    for broken in (raw[: len(raw) // 2], raw[:20], prog[:40]):
        with pytest.raises(lp.Error) as e:
            lp.loaders.load_gltf(make_gltf([[{"pos": QUAD}]], [{"mesh": 0}], images=[broken], textures=[0]), lp.Scene())
        assert e.value.kind == "FileNotFound"


def _hdr_bytes(px, rle=True, header=b"#?RADIANCE\nSOFTWARE=test\nFORMAT=32-bit_rle_rgbe\nEXPOSURE=1.0\n\n"):
    """write (h, w, 4) RGBE pixels as a Radiance file: new-style RLE scanlines (runs where a value repeats) or flat"""
    h, w, _ = px.shape
    out = bytearray(header + b"-Y %d +X %d\n" % (h, w))
    for y in range(h):
        if rle and 8 <= w < 32768:
            out += bytes([2, 2, w >> 8, w & 255])
            for ch in range(4):
                row = px[y, :, ch]
                x = 0
                while x < w:
                    run = 1
                    while x + run < w and run < 127 and row[x + run] == row[x]:
                        run += 1
                    if run >= 3:
                        out += bytes([128 + run, int(row[x])]); x += run
                    else:
                        lit = 1
                        while x + lit < w and lit < 128 and not (x + lit + 2 < w and row[x + lit] == row[x + lit + 1] == row[x + lit + 2]):
                            lit += 1
                        out += bytes([lit]) + row[x:x + lit].tobytes(); x += lit
        else:
            out += px[y].tobytes()
    return bytes(out)


def test_hdr_decoder_matches_oracle_and_source():
    """load_env (app.rs:138-155): Radiance .hdr -> RGBE8 pixels; product == numpy restatement == what was encoded"""
    rng = np.random.default_rng(11)
    for (h, w) in [(5, 7), (16, 64), (9, 300)]:
        px = rng.integers(0, 256, (h, w, 4), dtype=np.uint8)
        px[:, w // 3: w // 3 + min(w // 2, 150)] = px[:, w // 3: w // 3 + 1]      # long runs
        px[h // 2] = 128                                                      # a constant scanline
        for rle in (True, False):
            raw = _hdr_bytes(px, rle)
            got = lp.load_env(raw)
            assert got.tobytes() == px.tobytes() and G.decode_hdr(raw).tobytes() == px.tobytes()
    # old-style runs inside flat scanlines: pixel, then (1,1,1,n) repeats it n times
    px = np.zeros((2, 10, 4), np.uint8)
    px[0, :] = (10, 20, 30, 129); px[1, :4] = (1, 2, 3, 128); px[1, 4:] = (9, 8, 7, 130)
    raw = b"#?RGBE\n\n-Y 2 +X 10\n" + bytes([10, 20, 30, 129, 1, 1, 1, 9]) + bytes([1, 2, 3, 128, 1, 1, 1, 3, 9, 8, 7, 130, 1, 1, 1, 5])
    assert lp.load_env(raw).tobytes() == px.tobytes() and G.decode_hdr(raw).tobytes() == px.tobytes()
    # the synthetic sky probe survives a file round trip and uploads as a probe
    from loupiote_amd import scenes
    sky = scenes.sky_probe(64, 32)
    assert lp.load_env(_hdr_bytes(sky)).tobytes() == sky.tobytes()
    for bad in (b"", b"#?RADIANCE\n\n-Y 2 +X 2\n\x00", b"#?RADIANCE\nFORMAT=32-bit_rle_xyze\n\n-Y 1 +X 1\n\x00\x00\x00\x00",
                b"#?RADIANCE\n\n+Y 1 +X 1\n\x00\x00\x00\x00", b"P6\n1 1\n255\n\x00\x00\x00", _hdr_bytes(sky)[:-9]):
        with pytest.raises(lp.Error) as e:
            lp.load_env(bad)
        assert e.value.kind == "FileNotFound"


def test_hdr_writer_round_trip(tmp_path):
    """lpt_write_hdr -> file -> lpt_decode_hdr: the RGBE pixels are Ward's float2rgbe of the input and decode to within
    one mantissa step (1/128 of the largest channel)"""
    rng = np.random.default_rng(5)
    img = np.zeros((9, 13, 4), np.float32)
    img[..., :3] = np.exp(rng.uniform(-12, 8, (9, 13, 3))).astype(np.float32)
    img[0, 0, :3] = 0; img[0, 1, :3] = (-1.0, np.nan, 2.0); img[0, 2, :3] = 1e-38; img[..., 3] = 1.0
    path = tmp_path / "r.hdr"
    lp.save_radiance(img, path)
    px = lp.load_env_path(path)
    assert px.shape == (9, 13, 4) and G.decode_hdr(open(path, "rb").read()).tobytes() == px.tobytes()
    clean = np.where(img[..., :3] > 0, img[..., :3], 0).astype(np.float32)
    m = clean.max(axis=2)
    mant, e = np.frexp(m)
    scale = np.where(m >= 1e-32, (mant.astype(np.float32) * np.float32(256.0) / np.where(m >= 1e-32, m, 1)).astype(np.float32), 0)
    want = np.zeros_like(px)
    want[..., :3] = (clean * scale[..., None]).astype(np.uint8)
    want[..., 3] = np.where(m >= 1e-32, e + 128, 0)
    want[m < 1e-32] = 0
    assert px.tobytes() == want.tobytes()
    dec = px[..., :3].astype(np.float64) * np.exp2(px[..., 3:4].astype(np.float64) - 136.0)
    assert np.all(np.abs(dec - clean) <= m[..., None] / 128.0 + 1e-30)


def test_jpeg_hostile_headers_and_truncated_scans_fail_fast():
    """untrusted input: a frame header may not make the decoder allocate more than the renderer's 8192 x 8192 bound, and a scan
    whose entropy-coded data stops early (a marker or the end of the file before its last block) fails instead of walking every
    remaining block — of every remaining scan — on zero bits"""
    import time
    rgb = _smooth_rgb(640, 480)
    for kw in (dict(quality=85, subsampling=2, progressive=True), dict(quality=85, subsampling=2), dict(quality=85, subsampling=0, restart_marker_rows=1)):
        raw = _jpeg(rgb, **kw)
        assert lp.decode_image(raw).shape == (640, 480, 4)
        sos = raw.index(b"\xff\xda")
        for cut in (sos + 40, sos + (len(raw) - sos) // 3, len(raw) - 200):
            with pytest.raises(lp.Error):
                lp.decode_image(raw[:cut] + b"\xff\xd9")
    # a 1 MB file whose SOF2 announces 65535 x 65535 (4:2:0): rejected at the header, nothing allocated, no time spent
    raw = bytearray(_jpeg(rgb, quality=85, subsampling=2, progressive=True))
    sof = raw.index(b"\xff\xc2")
    raw[sof + 5:sof + 9] = b"\xff\xff\xff\xff"
    raw += bytes(1 << 20)
    t0 = time.time()
    with pytest.raises(lp.Error):
        lp.decode_image(bytes(raw))
    assert time.time() - t0 < 1.0
    # 8192 x 8192 itself is inside the bound; 8192 x 8200 is not (header check only: the scan data is that of the small image)
    raw[sof + 5:sof + 9] = (8200).to_bytes(2, "big") + (8192).to_bytes(2, "big")
    with pytest.raises(lp.Error):
        lp.decode_image(bytes(raw))


def test_decode_image_export(tmp_path):
    """lpt_decode_image: the decoder behind load_blue_noise (app.rs:116-132) — PNG exact, JPEG within the loader's tolerance"""
    from PIL import Image
    rng = np.random.default_rng(8)
    rgba = rng.integers(0, 256, (33, 47, 4), dtype=np.uint8)
    assert np.array_equal(lp.decode_image(png_bytes(rgba)), rgba)
    rgb = _smooth_rgb()
    got = lp.decode_image(_jpeg(rgb, quality=92, subsampling=0, progressive=True))
    ref = np.asarray(Image.open(io.BytesIO(_jpeg(rgb, quality=92, subsampling=0, progressive=True))).convert("RGB")).astype(int)
    assert got.shape == (97, 131, 4) and np.abs(got[..., :3].astype(int) - ref).max() <= 4
    with pytest.raises(lp.Error) as e:
        lp.decode_image(b"not an image")
    assert e.value.kind == "FileNotFound"
