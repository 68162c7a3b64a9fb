"""CPU ORACLE (test infrastructure) — glTF -> flat scene arrays, and instance baking.

Numpy restatement of the reference loader ``loaders::load_gltf``
(reference crates/lib/src/loaders/gltf.rs:46-156) and of the scene layout
``Scene::default`` seeds (reference crates/lib/src/scene.rs:37-54: ONE dummy
element at index 0 of every array).  Only tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg may import this module.

PARITY UNPINNED: the reference has no loader tests; BVH construction and the
`Vertex`/`Instance` packing live in the absent `albedo_rtx` crate.  What is pinned
by the reference text and restated here:
  * positions -> [x,y,z,0] f32, normals [f32;3], uv0 [f32;2], indices -> u32  (gltf.rs:77-98)
  * one BLAS entry per accepted primitive; primitives without POSITION or with a
    non-triangle mode are skipped                                              (gltf.rs:61-73)
  * Material {color, roughness, reflectivity=metallic, albedo_texture, mra_texture},
    INVALID_INDEX when a texture is absent                                     (gltf.rs:109-127)
  * one instance per node x primitive with node.transform() only (no parent chain) (gltf.rs:129-148)
  * append semantics: bvh_offset / mat_offset / texture_offset taken before loading (gltf.rs:60,109-110)
Documented deviations (SPEC.md §14) shared with the product:
  * instances index BLAS entries per (mesh, primitive), not per mesh (fixes gltf.rs:134,141-145)
  * a primitive without material maps to material 0 (reference: mat_offset + u32::MAX overflows, :137-145)
  * texture ids resolve through textures[i].source (reference uses the texture index, :119,123)
"""
import base64
import json
import struct

import numpy as np

INVALID = 0xFFFFFFFF

MATERIAL_DT = np.dtype([("color", "<f4", 4), ("roughness", "<f4"), ("reflectivity", "<f4"),
                        ("albedo_texture", "<u4"), ("mra_texture", "<u4")])
VERTEX_DT = np.dtype([("position", "<f4", 4), ("normal", "<f4", 4)])
LIGHT_DT = np.dtype([("normal", "<f4", 4), ("tangent", "<f4", 4), ("bitangent", "<f4", 4), ("origin", "<f4", 4)])
INSTANCE_DT = np.dtype([("model_to_world", "<f4", 16), ("blas_index", "<u4"), ("material_index", "<u4"),
                        ("pad", "<u4", 2)])
ENTRY_DT = np.dtype([("vertex_offset", "<u4"), ("vertex_count", "<u4"), ("index_offset", "<u4"),
                     ("index_count", "<u4")])

_COMP = {5120: ("i1", 1), 5121: ("u1", 1), 5122: ("<i2", 2), 5123: ("<u2", 2), 5125: ("<u4", 4), 5126: ("<f4", 4)}
_NCOMP = {"SCALAR": 1, "VEC2": 2, "VEC3": 3, "VEC4": 4, "MAT4": 16}


def default_material():
    m = np.zeros(1, MATERIAL_DT)
    m["color"] = (1.0, 1.0, 1.0, 1.0)
    m["roughness"] = 1.0
    m["reflectivity"] = 0.0
    m["albedo_texture"] = INVALID
    m["mra_texture"] = INVALID
    return m


def default_light():
    """Light::new() of the build (SPEC.md §2.4): unit square at the origin facing +Z, radiance 1."""
    l = np.zeros(1, LIGHT_DT)
    l["normal"] = (0, 0, 1, 0)
    l["tangent"] = (1, 0, 0, 0.5)
    l["bitangent"] = (0, 1, 0, 0.5)
    l["origin"] = (0, 0, 0, 1.0)
    return l


def identity16():
    return np.eye(4, dtype=np.float32).T.reshape(16).copy()


class Scene:
    """Scene::default(): one dummy element in every array (scene.rs:37-54)."""

    def __init__(self):
        self.materials = default_material()
        self.entries = np.zeros(1, ENTRY_DT)
        self.vertices = np.zeros(1, VERTEX_DT)
        self.indices = np.zeros(0, np.uint32)
        inst = np.zeros(1, INSTANCE_DT)
        inst["model_to_world"] = identity16()
        self.instances = inst
        self.lights = default_light()
        self.images = []  # list of (h, w, 4) uint8

    # BLASArray::add_bvh / add_bvh_indexed
    def add_mesh(self, positions, normals=None, uvs=None, indices=None):
        positions = np.asarray(positions, np.float32)[:, :3]
        n = positions.shape[0]
        if indices is None:
            indices = np.arange(n, dtype=np.uint32)
        indices = np.asarray(indices, np.uint32)
        if indices.size % 3 != 0 or (indices.size and indices.max() >= n):
            raise ValueError("accel build: bad indices")
        v = np.zeros(n, VERTEX_DT)
        v["position"][:, :3] = positions
        if uvs is not None:
            uvs = np.asarray(uvs, np.float32)
            v["position"][:, 3] = uvs[:, 0]
            v["normal"][:, 3] = uvs[:, 1]
        if normals is not None:
            v["normal"][:, :3] = np.asarray(normals, np.float32)[:, :3]
        else:
            v["normal"][:, :3] = flat_normals(positions, indices)
        e = np.zeros(1, ENTRY_DT)
        e["vertex_offset"] = self.vertices.shape[0]
        e["vertex_count"] = n
        e["index_offset"] = self.indices.shape[0]
        e["index_count"] = indices.size
        self.vertices = np.concatenate([self.vertices, v])
        self.indices = np.concatenate([self.indices, indices])
        self.entries = np.concatenate([self.entries, e])
        return self.entries.shape[0] - 1

    def add_instance(self, blas_index, model_to_world, material_index):
        i = np.zeros(1, INSTANCE_DT)
        i["model_to_world"] = np.asarray(model_to_world, np.float32).reshape(16)
        i["blas_index"] = blas_index
        i["material_index"] = material_index
        self.instances = np.concatenate([self.instances, i])
        return self.instances.shape[0] - 1

    def add_material(self, color, roughness, reflectivity, albedo_texture=INVALID, mra_texture=INVALID):
        m = np.zeros(1, MATERIAL_DT)
        m["color"] = color
        m["roughness"] = roughness
        m["reflectivity"] = reflectivity
        m["albedo_texture"] = albedo_texture
        m["mra_texture"] = mra_texture
        self.materials = np.concatenate([self.materials, m])
        return self.materials.shape[0] - 1


def _f32(x):
    return np.asarray(x, np.float32)


def _normalize_rows(v):
    """SPEC §3: v * (1/sqrt((x*x + y*y) + z*z)); zero rows stay zero.  fp32, one rounding per op."""
    v = _f32(v)
    l2 = (v[:, 0] * v[:, 0] + v[:, 1] * v[:, 1]) + v[:, 2] * v[:, 2]
    ok = l2 > 0
    inv = np.zeros_like(l2)
    inv[ok] = np.float32(1.0) / np.sqrt(l2[ok])
    return v * inv[:, None]


def _cross(a, b):
    return np.stack([a[:, 1] * b[:, 2] - a[:, 2] * b[:, 1],
                     a[:, 2] * b[:, 0] - a[:, 0] * b[:, 2],
                     a[:, 0] * b[:, 1] - a[:, 1] * b[:, 0]], axis=1).astype(np.float32)


def flat_normals(positions, indices):
    """SPEC §2.2: per-vertex normal = normalize(sum over incident triangles, in index order, of
    cross(p1-p0, p2-p0)).  (reference binary.rs:31-49 uses flat face normals for soups.)"""
    p = _f32(positions)
    acc = np.zeros((p.shape[0], 3), np.float32)
    tri = indices.reshape(-1, 3)
    fn = _cross(p[tri[:, 1]] - p[tri[:, 0]], p[tri[:, 2]] - p[tri[:, 0]])
    for t in range(tri.shape[0]):  # order matters for fp32 sums
        for k in range(3):
            acc[tri[t, k]] = acc[tri[t, k]] + fn[t]
    return _normalize_rows(acc)


# ---------------------------------------------------------------------------------- glTF
def _split_glb(data):
    if data[:4] == b"glTF":
        magic, version, length = struct.unpack_from("<III", data, 0)
        off = 12
        js, bin_chunk = None, None
        while off + 8 <= min(length, len(data)):
            clen, ctype = struct.unpack_from("<II", data, off)
            off += 8
            chunk = data[off:off + clen]
            off += clen
            if ctype == 0x4E4F534A:
                js = json.loads(chunk.decode("utf-8"))
            elif ctype == 0x004E4942 and bin_chunk is None:
                bin_chunk = bytes(chunk)
        if js is None:
            raise ValueError("glb without JSON chunk")
        return js, bin_chunk
    return json.loads(bytes(data).decode("utf-8")), None


def _buffers(js, bin_chunk):
    out = []
    for i, b in enumerate(js.get("buffers", [])):
        uri = b.get("uri")
        if uri is None:
            if bin_chunk is None:
                raise ValueError("missing BIN chunk")
            out.append(bin_chunk)
        elif uri.startswith("data:"):
            out.append(base64.b64decode(uri.split(",", 1)[1]))
        else:
            raise ValueError("external buffers are not supported by load_gltf(&[u8])")
    return out


def _accessor(js, buffers, idx):
    a = js["accessors"][idx]
    dt, size = _COMP[a["componentType"]]
    nc = _NCOMP[a["type"]]
    count = a["count"]
    if "bufferView" not in a:
        return np.zeros((count, nc), dt)
    bv = js["bufferViews"][a["bufferView"]]
    base = bv.get("byteOffset", 0) + a.get("byteOffset", 0)
    stride = bv.get("byteStride", 0) or size * nc
    buf = buffers[bv["buffer"]]
    arr = np.ndarray((count, nc), dt, buf, base, (stride, size))
    out = np.array(arr)
    if a.get("normalized", False):
        info = {"u1": 255.0, "<u2": 65535.0, "i1": 127.0, "<i2": 32767.0}[dt]
        out = np.maximum(out.astype(np.float32) / np.float32(info), np.float32(-1.0))
    return out


def _node_matrix(node):
    """gltf::scene::Transform::matrix(): explicit matrix, or T*R*S composed in fp32 (SPEC §14.4)."""
    if "matrix" in node:
        return _f32(node["matrix"]).reshape(16)
    t = _f32(node.get("translation", [0, 0, 0]))
    q = _f32(node.get("rotation", [0, 0, 0, 1]))
    s = _f32(node.get("scale", [1, 1, 1]))
    x, y, z, w = q
    one, two = np.float32(1), np.float32(2)
    r = np.array([[one - two * (y * y + z * z), two * (x * y - w * z), two * (x * z + w * y)],
                  [two * (x * y + w * z), one - two * (x * x + z * z), two * (y * z - w * x)],
                  [two * (x * z - w * y), two * (y * z + w * x), one - two * (x * x + y * y)]], np.float32)
    m = np.zeros((4, 4), np.float32)
    m[:3, 0] = r[:, 0] * s[0]
    m[:3, 1] = r[:, 1] * s[1]
    m[:3, 2] = r[:, 2] * s[2]
    m[:3, 3] = t
    m[3, 3] = 1
    return m.T.reshape(16).copy()  # column-major


def load_gltf(data, scene):
    """loaders::load_gltf(&[u8], &mut Scene) (gltf.rs:46-156).  Raises FileNotFoundError on parse failure."""
    try:
        js, bin_chunk = _split_glb(bytes(data))
        buffers = _buffers(js, bin_chunk)
        bvh_offset = scene.entries.shape[0]
        mesh_first_entry = []
        mesh_prim_entry = []
        n_entries = 0
        for mesh in js.get("meshes", []):
            mesh_first_entry.append(n_entries)
            per_prim = []
            for prim in mesh.get("primitives", []):
                attrs = prim.get("attributes", {})
                mode = prim.get("mode", 4)
                if "POSITION" not in attrs or mode not in (4, 5, 6):
                    per_prim.append(None)
                    continue
                pos = _accessor(js, buffers, attrs["POSITION"]).astype(np.float32)
                nrm = _accessor(js, buffers, attrs["NORMAL"]).astype(np.float32) if "NORMAL" in attrs else None
                uv = None
                if "TEXCOORD_0" in attrs:
                    uv = _accessor(js, buffers, attrs["TEXCOORD_0"]).astype(np.float32)
                idx = None
                if "indices" in prim:
                    idx = _accessor(js, buffers, prim["indices"]).astype(np.uint32).reshape(-1)
                else:
                    idx = np.arange(pos.shape[0], dtype=np.uint32)
                if mode == 5:  # strip
                    tri = []
                    for i in range(len(idx) - 2):
                        tri += [idx[i], idx[i + 1 + (i & 1)], idx[i + 2 - (i & 1)]]
                    idx = np.asarray(tri, np.uint32)
                elif mode == 6:  # fan
                    tri = []
                    for i in range(1, len(idx) - 1):
                        tri += [idx[0], idx[i], idx[i + 1]]
                    idx = np.asarray(tri, np.uint32)
                else:
                    idx = idx[: (idx.size // 3) * 3]
                scene.add_mesh(pos, nrm, uv, idx)
                per_prim.append(n_entries)
                n_entries += 1
            mesh_prim_entry.append(per_prim)
        mat_offset = scene.materials.shape[0]
        texture_offset = len(scene.images)
        textures = js.get("textures", [])

        def tex_id(info):
            if info is None:
                return INVALID
            src = textures[info["index"]].get("source")
            return INVALID if src is None else texture_offset + src

        for mat in js.get("materials", []):
            pbr = mat.get("pbrMetallicRoughness", {})
            scene.add_material(_f32(pbr.get("baseColorFactor", [1, 1, 1, 1])),
                               np.float32(pbr.get("roughnessFactor", 1.0)),
                               np.float32(pbr.get("metallicFactor", 1.0)),
                               tex_id(pbr.get("baseColorTexture")), tex_id(pbr.get("metallicRoughnessTexture")))
        for node in js.get("nodes", []):
            if "mesh" not in node:
                continue
            m = _node_matrix(node)
            mesh = js["meshes"][node["mesh"]]
            for k, prim in enumerate(mesh.get("primitives", [])):
                e = mesh_prim_entry[node["mesh"]][k]
                if e is None:
                    continue
                mi = prim.get("material")
                scene.add_instance(bvh_offset + e, m, 0 if mi is None else mat_offset + mi)
        for img in js.get("images", []):
            scene.images.append(_decode_image(js, buffers, img))
    except (ValueError, KeyError, IndexError, struct.error, json.JSONDecodeError, UnicodeDecodeError) as e:
        raise FileNotFoundError("file not found: " + str(e))


def _decode_image(js, buffers, img):
    """Images are decoded with PIL when present; the product carries its own PNG decoder."""
    import io
    if "bufferView" in img:
        bv = js["bufferViews"][img["bufferView"]]
        raw = buffers[bv["buffer"]][bv.get("byteOffset", 0): bv.get("byteOffset", 0) + bv["byteLength"]]
    elif img.get("uri", "").startswith("data:"):
        raw = base64.b64decode(img["uri"].split(",", 1)[1])
    else:
        raise ValueError("external image")
    from PIL import Image
    im = Image.open(io.BytesIO(raw))
    a = np.asarray(im)
    if a.ndim == 2:
        a = a[:, :, None]
    out = np.zeros((a.shape[0], a.shape[1], 4), np.uint8)  # gltf.rs:26-38: missing channels stay 0
    out[:, :, : a.shape[2]] = a[:, :, :4]
    return out


# ---------------------------------------------------------------------------------- baking
def bake(scene):
    """SPEC §2.5: instances -> world-space triangle soup (3 vertices per triangle, lpt_vertex layout)
    plus one material id per triangle.  Instance 0 (the dummy) contributes nothing."""
    verts, mats = [], []
    for ii in range(scene.instances.shape[0]):
        inst = scene.instances[ii]
        bi = int(inst["blas_index"])
        if bi >= scene.entries.shape[0]:
            continue
        e = scene.entries[bi]
        ntri = int(e["index_count"]) // 3
        if ntri == 0:
            continue
        m = _f32(inst["model_to_world"])
        v = scene.vertices[int(e["vertex_offset"]): int(e["vertex_offset"]) + int(e["vertex_count"])]
        idx = scene.indices[int(e["index_offset"]): int(e["index_offset"]) + int(e["index_count"])]
        p = v["position"][:, :3]
        x, y, z = p[:, 0], p[:, 1], p[:, 2]
        wp = np.stack([((m[0] * x + m[4] * y) + m[8] * z) + m[12],
                       ((m[1] * x + m[5] * y) + m[9] * z) + m[13],
                       ((m[2] * x + m[6] * y) + m[10] * z) + m[14]], axis=1).astype(np.float32)
        # cofactor matrix of the upper 3x3 (column-major a[col][row])
        a00, a10, a20 = m[0], m[1], m[2]
        a01, a11, a21 = m[4], m[5], m[6]
        a02, a12, a22 = m[8], m[9], m[10]
        c00 = a11 * a22 - a12 * a21
        c01 = a12 * a20 - a10 * a22
        c02 = a10 * a21 - a11 * a20
        c10 = a02 * a21 - a01 * a22
        c11 = a00 * a22 - a02 * a20
        c12 = a01 * a20 - a00 * a21
        c20 = a01 * a12 - a02 * a11
        c21 = a02 * a10 - a00 * a12
        c22 = a00 * a11 - a01 * a10
        n = v["normal"][:, :3]
        nx, ny, nz = n[:, 0], n[:, 1], n[:, 2]
        # n' = cof(A) n  (= det(A) * A^-T n), c{row}{col}
        wn = np.stack([(c00 * nx + c01 * ny) + c02 * nz,
                       (c10 * nx + c11 * ny) + c12 * nz,
                       (c20 * nx + c21 * ny) + c22 * nz], axis=1).astype(np.float32)
        wn = _normalize_rows(wn)
        out = np.zeros(v.shape[0], VERTEX_DT)
        out["position"][:, :3] = wp
        out["position"][:, 3] = v["position"][:, 3]
        out["normal"][:, :3] = wn
        out["normal"][:, 3] = v["normal"][:, 3]
        verts.append(out[idx])
        mi = int(inst["material_index"])
        if mi >= scene.materials.shape[0]:
            mi = 0
        mats.append(np.full(ntri, mi, np.uint32))
    if not verts:
        return np.zeros(0, VERTEX_DT), np.zeros(0, np.uint32)
    return np.concatenate(verts), np.concatenate(mats)


def decode_hdr(data):
    """Radiance RGBE (.hdr) -> (h, w, 4) uint8 RGBE pixels, the bytes `image::codecs::hdr::HdrDecoder::read_image_native`
    hands to ProbeGPU::new (reference crates/standalone/src/app.rs:138-155).  Plain restatement of the format:
    text header up to an empty line, "-Y H +X W", then per scanline either the new RLE (2 2 hi lo, four channel planes
    of runs / literals), or flat pixels with old-style (1,1,1,n) repeat markers."""
    data = bytes(data)
    pos = data.index(b"\n") + 1
    if not data.startswith(b"#?"):
        raise ValueError("not a Radiance file")
    while True:
        end = data.index(b"\n", pos)
        line = data[pos:end].rstrip(b"\r")
        pos = end + 1
        if not line:
            break
        if line.startswith(b"FORMAT=") and line != b"FORMAT=32-bit_rle_rgbe":
            raise ValueError("unsupported FORMAT")
    end = data.index(b"\n", pos)
    parts = data[pos:end].split()
    pos = end + 1
    if len(parts) != 4 or parts[0] != b"-Y" or parts[2] != b"+X":
        raise ValueError("unsupported orientation")
    H, W = int(parts[1]), int(parts[3])
    out = np.zeros((H, W, 4), np.uint8)
    for y in range(H):
        if 8 <= W < 32768 and data[pos] == 2 and data[pos + 1] == 2 and (data[pos + 2] << 8 | data[pos + 3]) == W:
            pos += 4
            for ch in range(4):
                x = 0
                while x < W:
                    c = data[pos]; pos += 1
                    if c > 128:
                        c -= 128
                        out[y, x:x + c, ch] = data[pos]; pos += 1
                    else:
                        out[y, x:x + c, ch] = np.frombuffer(data, np.uint8, c, pos); pos += c
                    x += c
        else:
            x, shift = 0, 0
            while x < W:
                p = data[pos:pos + 4]; pos += 4
                if p[0] == 1 and p[1] == 1 and p[2] == 1:
                    n = p[3] << shift
                    out[y, x:x + n] = out[y, x - 1]
                    x += n; shift += 8
                else:
                    out[y, x] = np.frombuffer(p, np.uint8); x += 1; shift = 0
    return out
