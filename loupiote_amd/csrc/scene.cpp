// scene.cpp — status/error plumbing and the CPU-side Scene (flat arrays).
// Mirrors reference crates/lib/src/scene.rs:30-54 (Scene::default with one dummy
// element per array) and crates/lib/src/errors.rs (Error -> String).
#include <cmath>

#include "common.h"

namespace lpt {

static thread_local char g_error[512] = "";

void set_error(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_error, sizeof g_error, fmt, ap);
    va_end(ap);
}
int fail(int status, const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_error, sizeof g_error, fmt, ap);
    va_end(ap);
    return status;
}

static void identity(float m[16]) {
    for (int i = 0; i < 16; ++i) m[i] = (i % 5 == 0) ? 1.f : 0.f;
}

static inline void normalize3(float v[3]) {
    float l2 = (v[0] * v[0] + v[1] * v[1]) + v[2] * v[2];
    if (!(l2 > 0.f)) { v[0] = v[1] = v[2] = 0.f; return; }
    float inv = 1.0f / sqrtf(l2);
    v[0] *= inv; v[1] *= inv; v[2] *= inv;
}

}  // namespace lpt

using namespace lpt;

extern "C" {

const char *lpt_last_error(void) { return g_error; }

const char *lpt_status_string(int status) {
    switch (status) {
        case LPT_OK: return "ok";
        case LPT_ERR_FILE_NOT_FOUND: return "file not found";
        case LPT_ERR_READBACK: return "failed to read pixels from GPU to CPU";
        case LPT_ERR_ACCEL_BUILD: return "failed to build acceleration structure";
        case LPT_ERR_HIP: return "HIP runtime error";
        case LPT_ERR_RCCL: return "RCCL error";
        case LPT_ERR_INVALID_ARG: return "invalid argument";
    }
    return "unknown status";
}

uint32_t lpt_abi_version(void) { return LPT_ABI_VERSION; }

int lpt_light_default(lpt_light *out) {
    if (!out) return fail(LPT_ERR_INVALID_ARG, "lpt_light_default: null");
    const lpt_light l = {{0.f, 0.f, 1.f, 0.f}, {1.f, 0.f, 0.f, 0.5f}, {0.f, 1.f, 0.f, 0.5f}, {0.f, 0.f, 0.f, 1.f}};
    *out = l;
    return LPT_OK;
}

int lpt_scene_create(lpt_scene **out) {
    if (!out) return fail(LPT_ERR_INVALID_ARG, "lpt_scene_create: null out");
    lpt_scene *s = new lpt_scene();
    lpt_material m = {{1.f, 1.f, 1.f, 1.f}, 1.f, 0.f, LPT_INVALID_INDEX, LPT_INVALID_INDEX};
    s->materials.push_back(m);
    s->entries.push_back(lpt_blas_entry{0, 0, 0, 0});
    s->vertices.push_back(lpt_vertex{{0, 0, 0, 0}, {0, 0, 0, 0}});
    lpt_instance inst;
    memset(&inst, 0, sizeof inst);
    identity(inst.model_to_world);
    s->instances.push_back(inst);
    lpt_light l;
    lpt_light_default(&l);
    s->lights.push_back(l);
    *out = s;
    return LPT_OK;
}

int lpt_scene_destroy(lpt_scene *scene) {
    delete scene;
    return LPT_OK;
}

int lpt_scene_counts_get(const lpt_scene *s, lpt_scene_counts *out) {
    if (!s || !out) return fail(LPT_ERR_INVALID_ARG, "lpt_scene_counts_get: null");
    out->materials = (uint32_t)s->materials.size();
    out->entries = (uint32_t)s->entries.size();
    out->vertices = (uint32_t)s->vertices.size();
    out->indices = (uint32_t)s->indices.size();
    out->instances = (uint32_t)s->instances.size();
    out->lights = (uint32_t)s->lights.size();
    out->images = (uint32_t)s->images.size();
    return LPT_OK;
}

int lpt_scene_add_mesh(lpt_scene *s, const void *positions, size_t position_stride, const void *normals,
                       size_t normal_stride, const void *uvs, size_t uv_stride, uint32_t vertex_count,
                       const uint32_t *indices, uint32_t index_count, uint32_t *out_blas_index) {
    if (!s || (!positions && vertex_count)) return fail(LPT_ERR_INVALID_ARG, "lpt_scene_add_mesh: null");
    if (position_stride < 12 || (normals && normal_stride < 12) || (uvs && uv_stride < 8))
        return fail(LPT_ERR_INVALID_ARG, "lpt_scene_add_mesh: stride too small");
    const uint32_t n_idx = indices ? index_count : vertex_count;
    if (n_idx % 3u != 0u) return fail(LPT_ERR_ACCEL_BUILD, "index count %u is not a multiple of 3", n_idx);
    if (indices)
        for (uint32_t i = 0; i < index_count; ++i)
            if (indices[i] >= vertex_count)
                return fail(LPT_ERR_ACCEL_BUILD, "index %u out of range (%u vertices)", indices[i], vertex_count);
    lpt_blas_entry e;
    e.vertex_offset = (uint32_t)s->vertices.size();
    e.vertex_count = vertex_count;
    e.index_offset = (uint32_t)s->indices.size();
    e.index_count = n_idx;
    const size_t v0 = s->vertices.size();
    s->vertices.resize(v0 + vertex_count);
    const uint8_t *pp = (const uint8_t *)positions, *pn = (const uint8_t *)normals, *pu = (const uint8_t *)uvs;
    for (uint32_t i = 0; i < vertex_count; ++i) {
        lpt_vertex &v = s->vertices[v0 + i];
        float p[3];
        memcpy(p, pp + (size_t)i * position_stride, 12);
        v.position[0] = p[0]; v.position[1] = p[1]; v.position[2] = p[2]; v.position[3] = 0.f;
        v.normal[0] = v.normal[1] = v.normal[2] = v.normal[3] = 0.f;
        if (pn) { float n[3]; memcpy(n, pn + (size_t)i * normal_stride, 12); v.normal[0] = n[0]; v.normal[1] = n[1]; v.normal[2] = n[2]; }
        if (pu) { float t[2]; memcpy(t, pu + (size_t)i * uv_stride, 8); v.position[3] = t[0]; v.normal[3] = t[1]; }
    }
    const size_t i0 = s->indices.size();
    s->indices.resize(i0 + n_idx);
    for (uint32_t i = 0; i < n_idx; ++i) s->indices[i0 + i] = indices ? indices[i] : i;
    if (!pn) {
        // vertex normal = normalize(sum, in index order, of cross(p1-p0, p2-p0)) (SPEC §2.2)
        for (uint32_t t = 0; t + 2 < n_idx; t += 3) {
            lpt_vertex *v = &s->vertices[v0];
            const uint32_t a = s->indices[i0 + t], b = s->indices[i0 + t + 1], c = s->indices[i0 + t + 2];
            const float e1[3] = {v[b].position[0] - v[a].position[0], v[b].position[1] - v[a].position[1], v[b].position[2] - v[a].position[2]};
            const float e2[3] = {v[c].position[0] - v[a].position[0], v[c].position[1] - v[a].position[1], v[c].position[2] - v[a].position[2]};
            const float fn[3] = {e1[1] * e2[2] - e1[2] * e2[1], e1[2] * e2[0] - e1[0] * e2[2], e1[0] * e2[1] - e1[1] * e2[0]};
            const uint32_t ids[3] = {a, b, c};
            for (int k = 0; k < 3; ++k)
                for (int ax = 0; ax < 3; ++ax) v[ids[k]].normal[ax] = v[ids[k]].normal[ax] + fn[ax];
        }
        for (uint32_t i = 0; i < vertex_count; ++i) normalize3(s->vertices[v0 + i].normal);
    }
    s->entries.push_back(e);
    if (out_blas_index) *out_blas_index = (uint32_t)s->entries.size() - 1u;
    return LPT_OK;
}

int lpt_scene_add_instance(lpt_scene *s, uint32_t blas_index, const float m[16], uint32_t material_index,
                           uint32_t *out_instance_index) {
    if (!s || !m) return fail(LPT_ERR_INVALID_ARG, "lpt_scene_add_instance: null");
    lpt_instance inst;
    memset(&inst, 0, sizeof inst);
    memcpy(inst.model_to_world, m, sizeof(float) * 16);
    inst.blas_index = blas_index;
    inst.material_index = material_index;
    s->instances.push_back(inst);
    if (out_instance_index) *out_instance_index = (uint32_t)s->instances.size() - 1u;
    return LPT_OK;
}

int lpt_scene_set_instance_transform(lpt_scene *s, uint32_t i, const float m[16]) {
    if (!s || !m || i >= s->instances.size()) return fail(LPT_ERR_INVALID_ARG, "lpt_scene_set_instance_transform: bad index %u", i);
    memcpy(s->instances[i].model_to_world, m, sizeof(float) * 16);
    return LPT_OK;
}

int lpt_scene_add_material(lpt_scene *s, const lpt_material *m, uint32_t *out_index) {
    if (!s || !m) return fail(LPT_ERR_INVALID_ARG, "lpt_scene_add_material: null");
    s->materials.push_back(*m);
    if (out_index) *out_index = (uint32_t)s->materials.size() - 1u;
    return LPT_OK;
}

int lpt_scene_add_image(lpt_scene *s, const uint8_t *rgba8, uint32_t w, uint32_t h, uint32_t *out_index) {
    if (!s || !rgba8 || !w || !h) return fail(LPT_ERR_INVALID_ARG, "lpt_scene_add_image: null or empty");
    Image im;
    im.width = w; im.height = h;
    im.rgba8.assign(rgba8, rgba8 + (size_t)w * h * 4);
    s->images.push_back(std::move(im));
    if (out_index) *out_index = (uint32_t)s->images.size() - 1u;
    return LPT_OK;
}

int lpt_scene_add_light(lpt_scene *s, const lpt_light *l, uint32_t *out_index) {
    if (!s || !l) return fail(LPT_ERR_INVALID_ARG, "lpt_scene_add_light: null");
    s->lights.push_back(*l);
    if (out_index) *out_index = (uint32_t)s->lights.size() - 1u;
    return LPT_OK;
}

int lpt_scene_set_light(lpt_scene *s, uint32_t i, const lpt_light *l) {
    if (!s || !l || i >= s->lights.size()) return fail(LPT_ERR_INVALID_ARG, "lpt_scene_set_light: bad index %u", i);
    s->lights[i] = *l;
    return LPT_OK;
}

#define LPT_GETTER(NAME, TYPE, FIELD)                                                            \
    int NAME(const lpt_scene *s, uint32_t first, uint32_t count, TYPE *dst) {                    \
        if (!s || (!dst && count)) return fail(LPT_ERR_INVALID_ARG, #NAME ": null");             \
        if ((size_t)first + count > s->FIELD.size())                                             \
            return fail(LPT_ERR_INVALID_ARG, #NAME ": range [%u,%u) exceeds %zu", first, first + count, s->FIELD.size()); \
        if (count) memcpy(dst, s->FIELD.data() + first, sizeof(TYPE) * (size_t)count);           \
        return LPT_OK;                                                                           \
    }
LPT_GETTER(lpt_scene_get_materials, lpt_material, materials)
LPT_GETTER(lpt_scene_get_entries, lpt_blas_entry, entries)
LPT_GETTER(lpt_scene_get_vertices, lpt_vertex, vertices)
LPT_GETTER(lpt_scene_get_indices, uint32_t, indices)
LPT_GETTER(lpt_scene_get_instances, lpt_instance, instances)
LPT_GETTER(lpt_scene_get_lights, lpt_light, lights)

int lpt_scene_get_image(const lpt_scene *s, uint32_t index, uint32_t *w, uint32_t *h, uint8_t *dst) {
    if (!s || index >= s->images.size()) return fail(LPT_ERR_INVALID_ARG, "lpt_scene_get_image: bad index %u", index);
    const Image &im = s->images[index];
    if (w) *w = im.width;
    if (h) *h = im.height;
    if (dst) memcpy(dst, im.rgba8.data(), im.rgba8.size());
    return LPT_OK;
}

}  // extern "C"
