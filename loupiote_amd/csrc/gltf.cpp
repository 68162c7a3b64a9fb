// gltf.cpp — glTF 2.0 / GLB -> flat Scene arrays.
// Mirrors reference crates/lib/src/loaders/gltf.rs:46-156 (load_gltf) and :158-161
// (load_gltf_path); deviations are listed in SPEC.md §14.
#include <cmath>
#include <fstream>

#include "common.h"
#include "json.h"

namespace lpt {
namespace {

struct GltfError { std::string what; };
[[noreturn]] void bad(const std::string &w) { throw GltfError{w}; }

std::vector<uint8_t> base64_decode(const char *s, size_t n) {
    static int8_t table[256];
    static bool init = false;
    if (!init) {
        for (int i = 0; i < 256; ++i) table[i] = -1;
        const char *abc = "ABCDEFGHIJKLMNOPQRSTUVWXYZabcdefghijklmnopqrstuvwxyz0123456789+/";
        for (int i = 0; i < 64; ++i) table[(uint8_t)abc[i]] = (int8_t)i;
        init = true;
    }
    std::vector<uint8_t> out;
    out.reserve(n * 3 / 4);
    uint32_t acc = 0;
    int bits = 0;
    for (size_t i = 0; i < n; ++i) {
        int8_t v = table[(uint8_t)s[i]];
        if (v < 0) continue;  // '=', whitespace
        acc = (acc << 6) | (uint32_t)v;
        bits += 6;
        if (bits >= 8) { bits -= 8; out.push_back((uint8_t)((acc >> bits) & 0xFF)); }
    }
    return out;
}

struct Doc {
    Json js;
    std::vector<std::vector<uint8_t>> buffers;
};

uint32_t rd32(const uint8_t *p) { return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24); }

void split(const uint8_t *data, size_t size, Doc &doc) {
    std::vector<uint8_t> bin;
    bool have_bin = false;
    if (size >= 12 && memcmp(data, "glTF", 4) == 0) {
        size_t length = rd32(data + 8);
        if (length > size) length = size;
        size_t off = 12;
        bool have_json = false;
        while (off + 8 <= length) {
            uint32_t clen = rd32(data + off), ctype = rd32(data + off + 4);
            off += 8;
            if (off + clen > size) bad("glb chunk exceeds file");
            if (ctype == 0x4E4F534Au) {
                JsonParser jp((const char *)data + off, clen);
                if (!jp.parse(doc.js)) bad("glb JSON chunk does not parse");
                have_json = true;
            } else if (ctype == 0x004E4942u && !have_bin) {
                bin.assign(data + off, data + off + clen);
                have_bin = true;
            }
            off += clen;
        }
        if (!have_json) bad("glb without JSON chunk");
    } else {
        JsonParser jp((const char *)data, size);
        if (!jp.parse(doc.js)) bad("gltf JSON does not parse");
    }
    if (!doc.js.is_obj()) bad("gltf root is not an object");
    const Json &bufs = doc.js.at("buffers");
    for (size_t i = 0; i < bufs.size(); ++i) {
        const Json *uri = bufs[i].find("uri");
        if (!uri) {
            if (!have_bin) bad("missing BIN chunk");
            doc.buffers.push_back(bin);
        } else if (uri->is_str() && uri->str.rfind("data:", 0) == 0) {
            size_t comma = uri->str.find(',');
            if (comma == std::string::npos) bad("malformed data URI");
            doc.buffers.push_back(base64_decode(uri->str.c_str() + comma + 1, uri->str.size() - comma - 1));
        } else bad("external buffers are not supported by load_gltf(&[u8])");
    }
}

int comp_size(long long ct) {
    switch (ct) { case 5120: case 5121: return 1; case 5122: case 5123: return 2; case 5125: case 5126: return 4; }
    bad("unknown componentType");
}
int type_count(const std::string &t) {
    if (t == "SCALAR") return 1; if (t == "VEC2") return 2; if (t == "VEC3") return 3; if (t == "VEC4") return 4; if (t == "MAT4") return 16;
    bad("unsupported accessor type " + t);
}

struct View { const uint8_t *base = nullptr; size_t stride = 0; size_t count = 0; long long ct = 0; int nc = 0; bool normalized = false; };

// A JSON number that must be a non-negative integer (offsets, counts, lengths, indices of untrusted files):
// negative, fractional, non-finite or > 2^53 values are rejected instead of wrapping through a cast.
size_t json_size(const Json &j, size_t dflt, const char *what) {
    if (j.kind != Json::Num) return dflt;
    const double v = j.num;
    if (!(v >= 0.0) || v > 9007199254740992.0 || v != std::floor(v)) bad(std::string(what) + " is not a non-negative integer");
    return (size_t)v;
}
// [off, off + need) inside a buffer of `size` bytes, without forming off + need
bool in_range(size_t off, size_t need, size_t size) { return off <= size && need <= size - off; }

View accessor(const Doc &d, long long idx) {
    const Json &accs = d.js.at("accessors");
    if (idx < 0 || (size_t)idx >= accs.size()) bad("accessor index out of range");
    const Json &a = accs[(size_t)idx];
    View v;
    v.ct = a.at("componentType").integer(0);
    v.nc = type_count(a.at("type").str);
    v.count = json_size(a.at("count"), 0, "accessor.count");
    v.normalized = a.at("normalized").kind == Json::Bool && a.at("normalized").b;
    const int cs = comp_size(v.ct);
    const Json *bvi = a.find("bufferView");
    if (!bvi) return v;  // zeros
    const Json &bvs = d.js.at("bufferViews");
    const size_t bvn = json_size(*bvi, (size_t)-1, "accessor.bufferView");
    if (bvn >= bvs.size()) bad("bufferView index out of range");
    const Json &bv = bvs[bvn];
    const size_t bi = json_size(bv.at("buffer"), 0, "bufferView.buffer");
    if (bi >= d.buffers.size()) bad("buffer index out of range");
    const size_t size = d.buffers[bi].size();
    const size_t off_bv = json_size(bv.at("byteOffset"), 0, "bufferView.byteOffset"), off_a = json_size(a.at("byteOffset"), 0, "accessor.byteOffset");
    if (!in_range(off_bv, off_a, size)) bad("accessor exceeds buffer");
    const size_t off = off_bv + off_a;
    size_t stride = json_size(bv.at("byteStride"), 0, "bufferView.byteStride");
    const size_t elem = (size_t)cs * (size_t)v.nc;
    if (!stride) stride = elem;
    if (v.count) {
        size_t span;  // (count - 1) * stride + elem, overflow-checked
        if (__builtin_mul_overflow(v.count - 1, stride, &span) || __builtin_add_overflow(span, elem, &span) || !in_range(off, span, size))
            bad("accessor exceeds buffer");
    }
    v.base = d.buffers[bi].data() + off;
    v.stride = stride;
    return v;
}

float read_float(const View &v, size_t i, int c) {
    if (!v.base) return 0.f;
    const uint8_t *p = v.base + i * v.stride;
    switch (v.ct) {
        case 5126: { float f; memcpy(&f, p + 4 * c, 4); return f; }
        case 5121: { uint8_t x = p[c]; return v.normalized ? (float)x / 255.0f : (float)x; }
        case 5123: { uint16_t x; memcpy(&x, p + 2 * c, 2); return v.normalized ? (float)x / 65535.0f : (float)x; }
        case 5120: { int8_t x = (int8_t)p[c]; float f = v.normalized ? (float)x / 127.0f : (float)x; return (v.normalized && f < -1.f) ? -1.f : f; }
        case 5122: { int16_t x; memcpy(&x, p + 2 * c, 2); float f = v.normalized ? (float)x / 32767.0f : (float)x; return (v.normalized && f < -1.f) ? -1.f : f; }
        case 5125: { uint32_t x; memcpy(&x, p + 4 * c, 4); return (float)x; }
    }
    return 0.f;
}
uint32_t read_index(const View &v, size_t i) {
    if (!v.base) return 0;
    const uint8_t *p = v.base + i * v.stride;
    switch (v.ct) {
        case 5121: return p[0];
        case 5123: { uint16_t x; memcpy(&x, p, 2); return x; }
        case 5125: { uint32_t x; memcpy(&x, p, 4); return x; }
    }
    bad("index accessor must be unsigned");
}

// gltf::scene::Transform::matrix(): explicit matrix or T*R*S composed in fp32 (SPEC §14.4)
void node_matrix(const Json &node, float m[16]) {
    const Json *mat = node.find("matrix");
    if (mat && mat->size() == 16) { for (int i = 0; i < 16; ++i) m[i] = (float)(*mat)[i].number(0.0); return; }
    float t[3] = {0, 0, 0}, q[4] = {0, 0, 0, 1}, s[3] = {1, 1, 1};
    const Json *jt = node.find("translation"), *jr = node.find("rotation"), *js = node.find("scale");
    if (jt && jt->size() == 3) for (int i = 0; i < 3; ++i) t[i] = (float)(*jt)[i].number(0.0);
    if (jr && jr->size() == 4) for (int i = 0; i < 4; ++i) q[i] = (float)(*jr)[i].number(0.0);
    if (js && js->size() == 3) for (int i = 0; i < 3; ++i) s[i] = (float)(*js)[i].number(1.0);
    const float x = q[0], y = q[1], z = q[2], w = q[3];
    const float r00 = 1.0f - 2.0f * (y * y + z * z), r01 = 2.0f * (x * y - w * z), r02 = 2.0f * (x * z + w * y);
    const float r10 = 2.0f * (x * y + w * z), r11 = 1.0f - 2.0f * (x * x + z * z), r12 = 2.0f * (y * z - w * x);
    const float r20 = 2.0f * (x * z - w * y), r21 = 2.0f * (y * z + w * x), r22 = 1.0f - 2.0f * (x * x + y * y);
    m[0] = r00 * s[0]; m[1] = r10 * s[0]; m[2] = r20 * s[0]; m[3] = 0.f;
    m[4] = r01 * s[1]; m[5] = r11 * s[1]; m[6] = r21 * s[1]; m[7] = 0.f;
    m[8] = r02 * s[2]; m[9] = r12 * s[2]; m[10] = r22 * s[2]; m[11] = 0.f;
    m[12] = t[0]; m[13] = t[1]; m[14] = t[2]; m[15] = 1.f;
}

void decode_image(const Doc &d, const Json &img, Image &out) {
    std::vector<uint8_t> raw;
    const uint8_t *p = nullptr;
    size_t n = 0;
    if (const Json *bvi = img.find("bufferView")) {
        const Json &bvs = d.js.at("bufferViews");
        const size_t bvn = json_size(*bvi, (size_t)-1, "image.bufferView");
        if (bvn >= bvs.size()) bad("image bufferView index out of range");
        const Json &bv = bvs[bvn];
        const size_t bi = json_size(bv.at("buffer"), 0, "bufferView.buffer"), off = json_size(bv.at("byteOffset"), 0, "bufferView.byteOffset");
        const size_t len = json_size(bv.at("byteLength"), 0, "bufferView.byteLength");
        if (bi >= d.buffers.size() || !in_range(off, len, d.buffers[bi].size())) bad("image bufferView out of range");
        p = d.buffers[bi].data() + off; n = len;
    } else if (img.at("uri").is_str() && img.at("uri").str.rfind("data:", 0) == 0) {
        const std::string &u = img.at("uri").str;
        size_t comma = u.find(',');
        if (comma == std::string::npos) bad("malformed image data URI");
        raw = base64_decode(u.c_str() + comma + 1, u.size() - comma - 1);
        p = raw.data(); n = raw.size();
    } else bad("external images are not supported by load_gltf(&[u8])");
    if (!decode_png(p, n, out) && !decode_jpeg(p, n, out)) bad("image is neither a decodable PNG nor a baseline JPEG");
}

void load(lpt_scene *scene, const uint8_t *data, size_t size) {
    Doc d;
    split(data, size, d);
    // work on a copy so a failure leaves the scene untouched
    lpt_scene tmp = *scene;
    const uint32_t bvh_offset = (uint32_t)tmp.entries.size();
    const Json &meshes = d.js.at("meshes");
    std::vector<std::vector<int>> prim_entry(meshes.size());
    int n_entries = 0;
    for (size_t mi = 0; mi < meshes.size(); ++mi) {
        const Json &prims = meshes[mi].at("primitives");
        for (size_t pi = 0; pi < prims.size(); ++pi) {
            const Json &prim = prims[pi];
            const Json &attrs = prim.at("attributes");
            const long long mode = prim.at("mode").integer(4);
            const Json *jpos = attrs.find("POSITION");
            if (!jpos || !(mode == 4 || mode == 5 || mode == 6)) { prim_entry[mi].push_back(-1); continue; }
            View pos = accessor(d, jpos->integer(-1));
            if (pos.nc < 3) bad("POSITION must be VEC3");
            const size_t nv = pos.count;
            std::vector<float> P(nv * 3), N, UV;
            for (size_t i = 0; i < nv; ++i) for (int c = 0; c < 3; ++c) P[3 * i + c] = read_float(pos, i, c);
            if (const Json *jn = attrs.find("NORMAL")) {
                View nrm = accessor(d, jn->integer(-1));
                if (nrm.count != nv || nrm.nc < 3) bad("NORMAL count mismatch");
                N.resize(nv * 3);
                for (size_t i = 0; i < nv; ++i) for (int c = 0; c < 3; ++c) N[3 * i + c] = read_float(nrm, i, c);
            }
            if (const Json *ju = attrs.find("TEXCOORD_0")) {
                View uv = accessor(d, ju->integer(-1));
                if (uv.count != nv || uv.nc < 2) bad("TEXCOORD_0 count mismatch");
                UV.resize(nv * 2);
                for (size_t i = 0; i < nv; ++i) for (int c = 0; c < 2; ++c) UV[2 * i + c] = read_float(uv, i, c);
            }
            std::vector<uint32_t> idx;
            if (const Json *ji = prim.find("indices")) {
                View iv = accessor(d, ji->integer(-1));
                idx.resize(iv.count);
                for (size_t i = 0; i < iv.count; ++i) idx[i] = read_index(iv, i);
            } else {
                idx.resize(nv);
                for (size_t i = 0; i < nv; ++i) idx[i] = (uint32_t)i;
            }
            std::vector<uint32_t> tri;
            if (mode == 5) {
                for (size_t i = 0; i + 2 < idx.size(); ++i) { tri.push_back(idx[i]); tri.push_back(idx[i + 1 + (i & 1)]); tri.push_back(idx[i + 2 - (i & 1)]); }
            } else if (mode == 6) {
                for (size_t i = 1; i + 1 < idx.size(); ++i) { tri.push_back(idx[0]); tri.push_back(idx[i]); tri.push_back(idx[i + 1]); }
            } else {
                tri.assign(idx.begin(), idx.begin() + (idx.size() / 3) * 3);
            }
            uint32_t blas = 0;
            int st = lpt_scene_add_mesh(&tmp, P.data(), 12, N.empty() ? nullptr : N.data(), 12, UV.empty() ? nullptr : UV.data(), 8,
                                        (uint32_t)nv, tri.data(), (uint32_t)tri.size(), &blas);
            if (st != LPT_OK) bad(std::string("mesh rejected: ") + lpt_last_error());
            prim_entry[mi].push_back(n_entries++);
        }
    }
    const uint32_t mat_offset = (uint32_t)tmp.materials.size();
    const uint32_t texture_offset = (uint32_t)tmp.images.size();
    const Json &textures = d.js.at("textures");
    const size_t n_images = d.js.at("images").size();
    auto tex_id = [&](const Json &info) -> uint32_t {
        if (!info.is_obj()) return LPT_INVALID_INDEX;
        size_t ti = (size_t)info.at("index").integer(-1);
        if (ti >= textures.size()) bad("texture index out of range");
        const Json *src = textures[ti].find("source");
        if (!src) return LPT_INVALID_INDEX;
        const size_t si = json_size(*src, (size_t)-1, "texture.source");
        if (si >= n_images) bad("texture.source out of range");
        return texture_offset + (uint32_t)si;
    };
    const Json &mats = d.js.at("materials");
    for (size_t i = 0; i < mats.size(); ++i) {
        const Json &pbr = mats[i].at("pbrMetallicRoughness");
        lpt_material m = {{1.f, 1.f, 1.f, 1.f}, 1.f, 1.f, LPT_INVALID_INDEX, LPT_INVALID_INDEX};
        const Json &bc = pbr.at("baseColorFactor");
        if (bc.size() == 4) for (int c = 0; c < 4; ++c) m.color[c] = (float)bc[c].number(1.0);
        m.roughness = (float)pbr.at("roughnessFactor").number(1.0);
        m.reflectivity = (float)pbr.at("metallicFactor").number(1.0);
        m.albedo_texture = tex_id(pbr.at("baseColorTexture"));
        m.mra_texture = tex_id(pbr.at("metallicRoughnessTexture"));
        tmp.materials.push_back(m);
    }
    const Json &nodes = d.js.at("nodes");
    for (size_t i = 0; i < nodes.size(); ++i) {
        const Json *jm = nodes[i].find("mesh");
        if (!jm) continue;
        size_t mi = (size_t)jm->integer(-1);
        if (mi >= meshes.size()) bad("node.mesh out of range");
        float m[16];
        node_matrix(nodes[i], m);
        const Json &prims = meshes[mi].at("primitives");
        for (size_t pi = 0; pi < prims.size(); ++pi) {
            int e = prim_entry[mi][pi];
            if (e < 0) continue;
            const Json *jmat = prims[pi].find("material");
            uint32_t material = 0;
            if (jmat) {
                if ((size_t)jmat->integer(-1) >= mats.size()) bad("primitive.material out of range");
                material = mat_offset + (uint32_t)jmat->integer(0);
            }
            lpt_scene_add_instance(&tmp, bvh_offset + (uint32_t)e, m, material, nullptr);
        }
    }
    const Json &images = d.js.at("images");
    for (size_t i = 0; i < images.size(); ++i) {
        Image im;
        decode_image(d, images[i], im);
        tmp.images.push_back(std::move(im));
    }
    *scene = std::move(tmp);
}

}  // namespace
}  // namespace lpt

extern "C" {

int lpt_load_gltf(lpt_scene *scene, const uint8_t *data, size_t size) {
    if (!scene || !data) return lpt::fail(LPT_ERR_INVALID_ARG, "lpt_load_gltf: null");
    try {
        lpt::load(scene, data, size);
    } catch (const lpt::GltfError &e) {
        // reference maps every gltf::Error to Error::FileNotFound (gltf.rs:49-53)
        return lpt::fail(LPT_ERR_FILE_NOT_FOUND, "file not found: %s", e.what.c_str());
    } catch (const std::exception &e) {
        return lpt::fail(LPT_ERR_FILE_NOT_FOUND, "file not found: %s", e.what());
    }
    return LPT_OK;
}

int lpt_load_gltf_path(lpt_scene *scene, const char *path) {
    if (!scene || !path) return lpt::fail(LPT_ERR_INVALID_ARG, "lpt_load_gltf_path: null");
    std::ifstream f(path, std::ios::binary);
    if (!f) return lpt::fail(LPT_ERR_FILE_NOT_FOUND, "file not found: %s", path);
    std::vector<uint8_t> bytes((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
    return lpt_load_gltf(scene, bytes.data(), bytes.size());
}

}  // extern "C"
