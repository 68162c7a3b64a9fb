/*
 * lpt_oracle.c — CPU ORACLE (test infrastructure; see lpt_oracle.h for the
 * "parity unpinned" statement and the reference citations).
 *
 * One pixel = one independent loop: ray generation -> [closest hit -> shade
 * (emission / environment, next-event estimation with an inline shadow ray,
 * BSDF sample)] x max_bounces -> accumulate.  This is the per-pixel view of the
 * stage sequence Renderer::raytrace records (crates/lib/src/renderer.rs:444-538).
 * Arithmetic follows SPEC.md section by section; comments name the section.
 */
#include "lpt_oracle.h"

#include <math.h>
#include <pthread.h>
#include <stdatomic.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------ SPEC §3 */
#define ORC_PI 3.14159265358979323846f
#define ORC_INV_PI 0.31830988618379067154f
#define ORC_INV_2PI 0.15915494309189533577f
#define ORC_HALF_PI 1.57079632679489661923f
#define ORC_T_INF 1.0e30f
#define ORC_TAG_RAYGEN 0x52415947u /* "RAYG" */
#define ORC_TAG_SHADE 0u
#define ORC_MIN_ROUGHNESS 0.045f
#define ORC_MIN_NOV 1.0e-4f

typedef struct { float x, y, z; } v3;

static inline v3 V3(float x, float y, float z) { v3 r = {x, y, z}; return r; }
static inline v3 add3(v3 a, v3 b) { return V3(a.x + b.x, a.y + b.y, a.z + b.z); }
static inline v3 sub3(v3 a, v3 b) { return V3(a.x - b.x, a.y - b.y, a.z - b.z); }
static inline v3 mul3(v3 a, float s) { return V3(a.x * s, a.y * s, a.z * s); }
static inline v3 neg3(v3 a) { return V3(-a.x, -a.y, -a.z); }
static inline float dot3(v3 a, v3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
static inline v3 cross3(v3 a, v3 b) {
    return V3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
static inline float fmax2(float a, float b) { return a > b ? a : b; }
static inline float fmin2(float a, float b) { return a < b ? a : b; }
static inline float clampf(float x, float lo, float hi) { return fmin2(fmax2(x, lo), hi); }
/* normalize: v * (1 / sqrt(dot)); zero vector stays zero */
static inline v3 normalize3(v3 a) {
    float l2 = dot3(a, a);
    if (!(l2 > 0.0f)) return V3(0.0f, 0.0f, 0.0f);
    float inv = 1.0f / sqrtf(l2);
    return mul3(a, inv);
}

/* ------------------------------------------------------------------ SPEC §4 RNG */
uint32_t orc_pcg_hash(uint32_t v) {
    uint32_t s = v * 747796405u + 2891336453u;
    uint32_t w = ((s >> ((s >> 28u) + 4u)) ^ s) * 277803737u;
    return (w >> 22u) ^ w;
}
typedef struct { uint32_t state; } rng_t;
static inline uint32_t stage_seed(uint32_t user_seed, uint32_t seed_counter) {
    return user_seed * 0x9E3779B9u + seed_counter;
}
static inline rng_t rng_init(uint32_t pixel, uint32_t sseed, uint32_t tag) {
    rng_t r;
    r.state = orc_pcg_hash(pixel ^ orc_pcg_hash(sseed ^ tag));
    return r;
}
static inline float rng_next(rng_t *r) {
    r->state = r->state * 747796405u + 2891336453u;
    uint32_t s = r->state;
    uint32_t w = ((s >> ((s >> 28u) + 4u)) ^ s) * 277803737u;
    w = (w >> 22u) ^ w;
    return (float)(w >> 8) * 5.9604644775390625e-8f; /* 2^-24 */
}
void orc_rng_stream(uint32_t pixel, uint32_t user_seed, uint32_t seed_counter, uint32_t tag, uint32_t n, float *out) {
    rng_t r = rng_init(pixel, stage_seed(user_seed, seed_counter), tag);
    for (uint32_t i = 0; i < n; ++i) out[i] = rng_next(&r);
}

/* ------------------------------------------------------------------ SPEC §5 approximations */
void orc_sincos2pi(float u, float *s, float *c) {
    float q = u * 4.0f;
    int k = (int)q;
    float f = q - (float)k;
    k &= 3;
    float x = f * ORC_HALF_PI;
    float x2 = x * x;
    float ps = fmaf(x2, 2.7557319223985893e-6f, -1.984126984126984e-4f);
    ps = fmaf(x2, ps, 8.333333333333333e-3f);
    ps = fmaf(x2, ps, -1.6666666666666666e-1f);
    ps = fmaf(x2, ps, 1.0f);
    float sn = x * ps;
    float pc = fmaf(x2, -2.755731922398589e-7f, 2.48015873015873e-5f);
    pc = fmaf(x2, pc, -1.3888888888888889e-3f);
    pc = fmaf(x2, pc, 4.1666666666666664e-2f);
    pc = fmaf(x2, pc, -0.5f);
    float cs = fmaf(x2, pc, 1.0f);
    if (k == 0) { *s = sn; *c = cs; }
    else if (k == 1) { *s = cs; *c = -sn; }
    else if (k == 2) { *s = -sn; *c = -cs; }
    else { *s = -cs; *c = sn; }
}
float orc_atan2(float y, float x) {
    float ax = fabsf(x), ay = fabsf(y);
    float mx = fmax2(ax, ay), mn = fmin2(ax, ay);
    if (!(mx > 0.0f)) return 0.0f;
    float a = mn / mx;
    float s = a * a;
    float r = fmaf(s, -0.0464964749f, 0.15931422f);
    r = fmaf(s, r, -0.327622764f);
    r = fmaf(r * s, a, a);
    if (ay > ax) r = ORC_HALF_PI - r;
    if (x < 0.0f) r = ORC_PI - r;
    if (y < 0.0f) r = -r;
    return r;
}
float orc_acos(float x) {
    float a = fabsf(x);
    if (a > 1.0f) a = 1.0f;
    float p = fmaf(a, -0.0187293f, 0.0742610f);
    p = fmaf(a, p, -0.2121144f);
    p = fmaf(a, p, 1.5707288f);
    float r = sqrtf(1.0f - a) * p;
    return x < 0.0f ? ORC_PI - r : r;
}
void orc_onb(const float n[3], float t[3], float b[3]) {
    float sign = copysignf(1.0f, n[2]);
    float a = -1.0f / (sign + n[2]);
    float bb = n[0] * n[1] * a;
    t[0] = 1.0f + sign * n[0] * n[0] * a;
    t[1] = sign * bb;
    t[2] = -sign * n[0];
    b[0] = bb;
    b[1] = sign + n[1] * n[1] * a;
    b[2] = -n[1];
}

/* ------------------------------------------------------------------ scene */
typedef struct { float lo[3], hi[3]; uint32_t left, right, first, count; uint32_t axis; } bnode; /* count>0: leaf; axis: split axis (near child first) */

struct orc_scene {
    uint32_t n_tris;
    orc_vertex *verts;     /* 3 per triangle */
    uint32_t *tri_material;
    float *woop;           /* 12 per triangle */
    uint32_t n_materials; orc_material *materials;
    uint32_t n_lights; orc_light *lights;
    uint32_t n_images; orc_image *images; uint8_t **image_data;
    uint32_t probe_w, probe_h; uint8_t *probe;
    uint32_t noise_w, noise_h; uint8_t *noise;
    float srgb_lut[256];
    /* private acceleration structure: binned-SAH BVH2 over triangle centroids (median split as the fall-back);
       hits do not depend on it (SPEC §7), only the oracle's speed does */
    bnode *nodes; uint32_t n_nodes; uint32_t *order;
    float max_abs;         /* the largest |coordinate| of any vertex: the scale the Woop test's rounding follows for rays that start anywhere in the scene */
};

/* SPEC §6: world->unit-triangle affine map, computed in double, rounded once */
void orc_woop(const float p0[3], const float p1[3], const float p2[3], float out[12]) {
    double ax = p0[0], ay = p0[1], az = p0[2];
    double e1x = (double)p1[0] - ax, e1y = (double)p1[1] - ay, e1z = (double)p1[2] - az;
    double e2x = (double)p2[0] - ax, e2y = (double)p2[1] - ay, e2z = (double)p2[2] - az;
    double nx = e1y * e2z - e1z * e2y, ny = e1z * e2x - e1x * e2z, nz = e1x * e2y - e1y * e2x;
    /* M = [e1 e2 n] (columns); det = n.n */
    double det = nx * nx + ny * ny + nz * nz;
    if (!(det > 0.0)) { for (int i = 0; i < 12; ++i) out[i] = 0.0f; return; }
    double inv = 1.0 / det;
    /* rows of M^-1: r0 = (e2 x n)/det, r1 = (n x e1)/det, r2 = n/det */
    double r0x = (e2y * nz - e2z * ny) * inv, r0y = (e2z * nx - e2x * nz) * inv, r0z = (e2x * ny - e2y * nx) * inv;
    double r1x = (ny * e1z - nz * e1y) * inv, r1y = (nz * e1x - nx * e1z) * inv, r1z = (nx * e1y - ny * e1x) * inv;
    double r2x = nx * inv, r2y = ny * inv, r2z = nz * inv;
    out[0] = (float)r0x; out[1] = (float)r0y; out[2] = (float)r0z; out[3] = (float)(-(r0x * ax + r0y * ay + r0z * az));
    out[4] = (float)r1x; out[5] = (float)r1y; out[6] = (float)r1z; out[7] = (float)(-(r1x * ax + r1y * ay + r1z * az));
    out[8] = (float)r2x; out[9] = (float)r2y; out[10] = (float)r2z; out[11] = (float)(-(r2x * ax + r2y * ay + r2z * az));
}

/* SPEC §7: Woop test.  Accepts tmin < t < tmax (callers apply the tie rule). */
int orc_ray_triangle(const float m[12], const float o[3], const float d[3], float tmin, float tmax,
                     float *t, float *u, float *v) {
    float oz = fmaf(m[10], o[2], fmaf(m[9], o[1], fmaf(m[8], o[0], m[11])));
    float dz = fmaf(m[10], d[2], fmaf(m[9], d[1], m[8] * d[0]));
    float tt = -oz / dz;
    if (!(tt > tmin && tt <= tmax)) return 0;
    float ox = fmaf(m[2], o[2], fmaf(m[1], o[1], fmaf(m[0], o[0], m[3])));
    float dx = fmaf(m[2], d[2], fmaf(m[1], d[1], m[0] * d[0]));
    float uu = fmaf(tt, dx, ox);
    if (!(uu >= 0.0f)) return 0;
    float oy = fmaf(m[6], o[2], fmaf(m[5], o[1], fmaf(m[4], o[0], m[7])));
    float dy = fmaf(m[6], d[2], fmaf(m[5], d[1], m[4] * d[0]));
    float vv = fmaf(tt, dy, oy);
    if (!(vv >= 0.0f) || !(uu + vv <= 1.0f)) return 0;
    *t = tt; *u = uu; *v = vv;
    return 1;
}

static void tri_bounds(const orc_scene *s, uint32_t tri, float lo[3], float hi[3]) {
    for (int a = 0; a < 3; ++a) { lo[a] = 1e30f; hi[a] = -1e30f; }
    for (int k = 0; k < 3; ++k)
        for (int a = 0; a < 3; ++a) {
            float x = s->verts[3 * tri + k].position[a];
            if (x < lo[a]) lo[a] = x;
            if (x > hi[a]) hi[a] = x;
        }
    /* generous padding: the oracle's boxes only have to be conservative.  The pad follows the largest coordinate of the
     * triangle on ANY axis: the Woop test's error scales with the position, also on an axis where the triangle sits at 0
     * (a floor at y = 0 got no padding from a per-axis rule; round 3: one wrong pixel-sample in 5.3e8 at 3840x2160x64) */
    float m = 0.0f;
    for (int a = 0; a < 3; ++a) m = fmax2(m, fmax2(fabsf(lo[a]), fabsf(hi[a])));
    /* round 5: ... and with the RAY's position, which may be anywhere in the scene: tiny triangles around the origin of a 2 000-unit scene lost 1 grazing hit in
     * 60 000 to the tree (brute force and the product's tree found it; profiles/r05_experiments_ab.txt U).  2e-6 of the scene's largest coordinate on top */
    for (int a = 0; a < 3; ++a) {
        float e = 1e-5f * m + 2e-6f * s->max_abs + 1e-6f * (hi[a] - lo[a]) + 1e-20f;
        lo[a] -= e; hi[a] += e;
    }
}

static float half_area3(const float lo[3], const float hi[3]) {
    float dx = hi[0] - lo[0], dy = hi[1] - lo[1], dz = hi[2] - lo[2];
    if (dx < 0.0f) return 0.0f;
    return dx * dy + dy * dz + dz * dx;
}

/* binned SAH (16 bins on the axis with the cheapest split); equal centroids fall back to an index median */
static uint32_t build_rec(orc_scene *s, const float *cent, const float *tlo, const float *thi,
                          uint32_t first, uint32_t count, uint32_t depth) {
    enum { NB = 16 };
    uint32_t id = s->n_nodes++;
    bnode *n = &s->nodes[id];
    float clo[3] = {1e30f, 1e30f, 1e30f}, chi[3] = {-1e30f, -1e30f, -1e30f};
    for (int a = 0; a < 3; ++a) { n->lo[a] = 1e30f; n->hi[a] = -1e30f; }
    for (uint32_t i = first; i < first + count; ++i) {
        uint32_t t = s->order[i];
        for (int a = 0; a < 3; ++a) {
            if (tlo[3 * t + a] < n->lo[a]) n->lo[a] = tlo[3 * t + a];
            if (thi[3 * t + a] > n->hi[a]) n->hi[a] = thi[3 * t + a];
            if (cent[3 * t + a] < clo[a]) clo[a] = cent[3 * t + a];
            if (cent[3 * t + a] > chi[a]) chi[a] = cent[3 * t + a];
        }
    }
    n->first = first; n->count = 0; n->left = n->right = 0; n->axis = 0;
    if (count <= 4) { n->count = count; return id; }
    int best_axis = -1; uint32_t best_bin = 0; float best_cost = 1e30f;
    for (int a = 0; a < 3 && depth < 48u; ++a) {   /* beyond depth 48 only balanced splits: the traversal stack is finite */
        const float ext = chi[a] - clo[a];
        if (!(ext > 0.0f)) continue;
        const float scale = (float)NB / ext;
        uint32_t cnt[NB]; float blo[NB][3], bhi[NB][3];
        for (int b = 0; b < NB; ++b) { cnt[b] = 0; for (int k = 0; k < 3; ++k) { blo[b][k] = 1e30f; bhi[b][k] = -1e30f; } }
        for (uint32_t i = first; i < first + count; ++i) {
            const uint32_t t = s->order[i];
            int b = (int)((cent[3 * t + a] - clo[a]) * scale);
            if (b < 0) b = 0;
            if (b > NB - 1) b = NB - 1;
            cnt[b]++;
            for (int k = 0; k < 3; ++k) {
                if (tlo[3 * t + k] < blo[b][k]) blo[b][k] = tlo[3 * t + k];
                if (thi[3 * t + k] > bhi[b][k]) bhi[b][k] = thi[3 * t + k];
            }
        }
        float rarea[NB]; uint32_t rcnt[NB];
        float alo[3] = {1e30f, 1e30f, 1e30f}, ahi[3] = {-1e30f, -1e30f, -1e30f};
        uint32_t acc = 0;
        for (int b = NB - 1; b > 0; --b) {
            for (int k = 0; k < 3; ++k) { if (blo[b][k] < alo[k]) alo[k] = blo[b][k]; if (bhi[b][k] > ahi[k]) ahi[k] = bhi[b][k]; }
            acc += cnt[b];
            rarea[b] = half_area3(alo, ahi); rcnt[b] = acc;
        }
        for (int k = 0; k < 3; ++k) { alo[k] = 1e30f; ahi[k] = -1e30f; }
        acc = 0;
        for (int b = 0; b < NB - 1; ++b) {
            for (int k = 0; k < 3; ++k) { if (blo[b][k] < alo[k]) alo[k] = blo[b][k]; if (bhi[b][k] > ahi[k]) ahi[k] = bhi[b][k]; }
            acc += cnt[b];
            if (acc == 0 || rcnt[b + 1] == 0) continue;
            const float cost = half_area3(alo, ahi) * (float)acc + rarea[b + 1] * (float)rcnt[b + 1];
            if (cost < best_cost) { best_cost = cost; best_axis = a; best_bin = (uint32_t)b; }
        }
    }
    uint32_t half = 0;
    if (best_axis >= 0) {
        /* partition in place: bins <= best_bin to the left */
        const float scale = (float)NB / (chi[best_axis] - clo[best_axis]);
        uint32_t i = first, j = first + count;
        while (i < j) {
            const uint32_t t = s->order[i];
            int b = (int)((cent[3 * t + best_axis] - clo[best_axis]) * scale);
            if (b < 0) b = 0;
            if (b > NB - 1) b = NB - 1;
            if ((uint32_t)b <= best_bin) ++i;
            else { --j; s->order[i] = s->order[j]; s->order[j] = t; }
        }
        half = i - first;
    }
    if (half == 0 || half == count) { half = count / 2; best_axis = 0; }   /* coincident centroids: split the run */
    uint32_t l = build_rec(s, cent, tlo, thi, first, half, depth + 1u);
    uint32_t r = build_rec(s, cent, tlo, thi, first + half, count - half, depth + 1u);
    s->nodes[id].left = l; s->nodes[id].right = r; s->nodes[id].axis = (uint32_t)best_axis;
    return id;
}

static double srgb_to_linear(double c) {
    return c <= 0.04045 ? c / 12.92 : pow((c + 0.055) / 1.055, 2.4);
}
float orc_srgb_lut(uint32_t i) { return (float)srgb_to_linear((double)(i & 255u) / 255.0); }

orc_scene *orc_scene_create(uint32_t n_tris, const orc_vertex *tri_verts, const uint32_t *tri_material,
                            uint32_t n_materials, const orc_material *materials,
                            uint32_t n_lights, const orc_light *lights,
                            uint32_t n_images, const orc_image *images,
                            uint32_t probe_w, uint32_t probe_h, const uint8_t *probe) {
    orc_scene *s = (orc_scene *)calloc(1, sizeof(orc_scene));
    s->n_tris = n_tris;
    s->verts = (orc_vertex *)malloc(sizeof(orc_vertex) * 3 * (size_t)(n_tris ? n_tris : 1));
    s->tri_material = (uint32_t *)malloc(sizeof(uint32_t) * (size_t)(n_tris ? n_tris : 1));
    s->woop = (float *)malloc(sizeof(float) * 12 * (size_t)(n_tris ? n_tris : 1));
    if (n_tris) {
        memcpy(s->verts, tri_verts, sizeof(orc_vertex) * 3 * (size_t)n_tris);
        memcpy(s->tri_material, tri_material, sizeof(uint32_t) * (size_t)n_tris);
    }
    for (uint32_t t = 0; t < n_tris; ++t)
        orc_woop(s->verts[3 * t].position, s->verts[3 * t + 1].position, s->verts[3 * t + 2].position, s->woop + 12 * (size_t)t);
    s->n_materials = n_materials;
    s->materials = (orc_material *)malloc(sizeof(orc_material) * (n_materials ? n_materials : 1));
    if (n_materials) memcpy(s->materials, materials, sizeof(orc_material) * n_materials);
    s->n_lights = n_lights;
    s->lights = (orc_light *)malloc(sizeof(orc_light) * (n_lights ? n_lights : 1));
    if (n_lights) memcpy(s->lights, lights, sizeof(orc_light) * n_lights);
    s->n_images = n_images;
    s->images = (orc_image *)calloc(n_images ? n_images : 1, sizeof(orc_image));
    s->image_data = (uint8_t **)calloc(n_images ? n_images : 1, sizeof(uint8_t *));
    for (uint32_t i = 0; i < n_images; ++i) {
        size_t bytes = (size_t)images[i].width * images[i].height * 4;
        s->image_data[i] = (uint8_t *)malloc(bytes ? bytes : 1);
        memcpy(s->image_data[i], images[i].rgba8, bytes);
        s->images[i].width = images[i].width; s->images[i].height = images[i].height;
        s->images[i].rgba8 = s->image_data[i];
    }
    if (probe && probe_w && probe_h) {
        s->probe_w = probe_w; s->probe_h = probe_h;
        s->probe = (uint8_t *)malloc((size_t)probe_w * probe_h * 4);
        memcpy(s->probe, probe, (size_t)probe_w * probe_h * 4);
    } else {
        s->probe_w = s->probe_h = 1;
        s->probe = (uint8_t *)calloc(4, 1);
    }
    for (uint32_t i = 0; i < 256; ++i) s->srgb_lut[i] = orc_srgb_lut(i);
    /* BVH */
    s->nodes = (bnode *)malloc(sizeof(bnode) * (2 * (size_t)n_tris + 2));
    s->order = (uint32_t *)malloc(sizeof(uint32_t) * (size_t)(n_tris ? n_tris : 1));
    s->n_nodes = 0;
    if (n_tris) {
        float *cent = (float *)malloc(sizeof(float) * 3 * (size_t)n_tris);
        float *tlo = (float *)malloc(sizeof(float) * 3 * (size_t)n_tris);
        float *thi = (float *)malloc(sizeof(float) * 3 * (size_t)n_tris);
        s->max_abs = 0.0f;
        for (size_t i = 0; i < 3 * (size_t)n_tris; ++i)
            for (int a = 0; a < 3; ++a) s->max_abs = fmax2(s->max_abs, fabsf(s->verts[i].position[a]));
        for (uint32_t t = 0; t < n_tris; ++t) {
            tri_bounds(s, t, tlo + 3 * (size_t)t, thi + 3 * (size_t)t);
            for (int a = 0; a < 3; ++a) cent[3 * (size_t)t + a] = 0.5f * (tlo[3 * (size_t)t + a] + thi[3 * (size_t)t + a]);
            s->order[t] = t;
        }
        build_rec(s, cent, tlo, thi, 0, n_tris, 0u);
        free(cent); free(tlo); free(thi);
    }
    return s;
}
void orc_scene_set_noise(orc_scene *s, const uint8_t *rgba8, uint32_t w, uint32_t h, uint32_t row_bytes) {
    free(s->noise);
    s->noise = NULL; s->noise_w = s->noise_h = 0;
    if (!rgba8 || !w || !h) return;
    s->noise = (uint8_t *)malloc((size_t)w * h * 4);
    for (uint32_t y = 0; y < h; ++y) memcpy(s->noise + (size_t)y * w * 4, rgba8 + (size_t)y * row_bytes, (size_t)w * 4);
    s->noise_w = w; s->noise_h = h;
}
void orc_scene_destroy(orc_scene *s) {
    if (!s) return;
    for (uint32_t i = 0; i < s->n_images; ++i) free(s->image_data[i]);
    free(s->image_data); free(s->images); free(s->verts); free(s->tri_material); free(s->woop);
    free(s->materials); free(s->lights); free(s->probe); free(s->noise); free(s->nodes); free(s->order);
    free(s);
}

/* ------------------------------------------------------------------ SPEC §7 closest / any hit */
/* tie rule: smaller t wins; equal t -> smaller prim id; triangles beat lights at equal t */
static inline void consider_tri(const orc_scene *s, uint32_t tri, const float o[3], const float d[3], orc_hit *best,
                                orc_counters *c) {
    float t, u, v;
    if (c) c->tris++;
    if (!orc_ray_triangle(s->woop + 12 * (size_t)tri, o, d, 0.0f, best->t, &t, &u, &v)) return;
    if (t < best->t || tri < best->prim) { best->t = t; best->u = u; best->v = v; best->prim = tri; }
}

/* Only the Woop test decides hits (SPEC §7), and its t carries the rounding of an affine map of the ORIGIN — at grazing
 * incidence far more than the slab arithmetic below.  A node is therefore culled against the best hit only with a margin
 * (1e-3 relative + absolute): a triangle whose Woop t undercuts the best hit by less than that is still tested, so the
 * result is the brute-force one whatever the tree (tests/test_oracle_kat.py: BVH == brute force). */
static inline int box_hit(const bnode *n, const float o[3], const float inv[3], float tbest) {
    float tn = 0.0f, tf = tbest + 1e-3f * (1.0f + fabsf(tbest));
    for (int a = 0; a < 3; ++a) {
        float t0 = (n->lo[a] - o[a]) * inv[a], t1 = (n->hi[a] - o[a]) * inv[a];
        if (t0 != t0 || t1 != t1) continue; /* 0 * inf: origin on a slab plane of a parallel ray — keep */
        if (t0 > t1) { float tmp = t0; t0 = t1; t1 = tmp; }
        /* widen by a few ulp (conservative) */
        t0 = t0 - fabsf(t0) * 4e-7f; t1 = t1 + fabsf(t1) * 4e-7f;
        if (t0 > tn) tn = t0;
        if (t1 < tf) tf = t1;
    }
    return tn <= tf;
}

static void closest_one(const orc_scene *s, const float o[3], const float d[3], orc_hit *best, int brute, orc_counters *c) {
    best->t = ORC_T_INF; best->u = 0.0f; best->v = 0.0f; best->prim = ORC_INVALID;
    if (brute || s->n_nodes == 0) {
        for (uint32_t t = 0; t < s->n_tris; ++t) consider_tri(s, t, o, d, best, c);
    } else {
        float inv[3] = {1.0f / d[0], 1.0f / d[1], 1.0f / d[2]};
        uint32_t stack[192]; int sp = 0;   /* depth <= 48 + log2(n) */
        stack[sp++] = 0;
        while (sp) {
            const bnode *n = &s->nodes[stack[--sp]];
            if (c) c->nodes++;
            if (!box_hit(n, o, inv, best->t)) continue;
            if (n->count) {
                for (uint32_t i = 0; i < n->count; ++i) consider_tri(s, s->order[n->first + i], o, d, best, c);
            } else if (d[n->axis] < 0.0f) { stack[sp++] = n->left; stack[sp++] = n->right; }   /* near child on top */
            else { stack[sp++] = n->right; stack[sp++] = n->left; }
        }
    }
    /* SPEC §8: rectangular emitters, front face only, strictly closer than any triangle */
    for (uint32_t l = 0; l < s->n_lights; ++l) {
        const orc_light *L = &s->lights[l];
        v3 nl = V3(L->normal[0], L->normal[1], L->normal[2]);
        v3 dd = V3(d[0], d[1], d[2]), oo = V3(o[0], o[1], o[2]);
        float dn = dot3(dd, nl);
        if (!(dn < 0.0f)) continue;
        v3 ctr = V3(L->origin[0], L->origin[1], L->origin[2]);
        float t = dot3(sub3(ctr, oo), nl) / dn;
        if (!(t > 0.0f && t < best->t)) continue;
        v3 p = V3(fmaf(dd.x, t, oo.x), fmaf(dd.y, t, oo.y), fmaf(dd.z, t, oo.z));
        v3 r = sub3(p, ctr);
        float a = dot3(r, V3(L->tangent[0], L->tangent[1], L->tangent[2]));
        float b = dot3(r, V3(L->bitangent[0], L->bitangent[1], L->bitangent[2]));
        if (fabsf(a) <= L->tangent[3] && fabsf(b) <= L->bitangent[3]) {
            best->t = t; best->u = a; best->v = b; best->prim = ORC_LIGHT_BIT | l;
        }
    }
}

static int occluded_one(const orc_scene *s, const float o[3], const float d[3], float tmax, int brute, orc_counters *c) {
    float t, u, v;
    if (brute || s->n_nodes == 0) {
        for (uint32_t i = 0; i < s->n_tris; ++i)
            if (orc_ray_triangle(s->woop + 12 * (size_t)i, o, d, 0.0f, tmax, &t, &u, &v)) return 1;
        return 0;
    }
    float inv[3] = {1.0f / d[0], 1.0f / d[1], 1.0f / d[2]};
    uint32_t stack[192]; int sp = 0;   /* depth <= 48 + log2(n) */
    stack[sp++] = 0;
    while (sp) {
        const bnode *n = &s->nodes[stack[--sp]];
        if (c) c->nodes++;
        if (!box_hit(n, o, inv, tmax)) continue;
        if (n->count) {
            for (uint32_t i = 0; i < n->count; ++i) {
                if (c) c->tris++;
                if (orc_ray_triangle(s->woop + 12 * (size_t)s->order[n->first + i], o, d, 0.0f, tmax, &t, &u, &v)) return 1;
            }
        } else if (d[n->axis] < 0.0f) { stack[sp++] = n->left; stack[sp++] = n->right; }
        else { stack[sp++] = n->right; stack[sp++] = n->left; }
    }
    return 0;
}

void orc_trace_closest(const orc_scene *s, const float *origins, const float *dirs, uint32_t n, orc_hit *out,
                       int brute, orc_counters *c) {
    for (uint32_t i = 0; i < n; ++i) closest_one(s, origins + 3 * (size_t)i, dirs + 3 * (size_t)i, &out[i], brute, c);
}
void orc_trace_occluded(const orc_scene *s, const float *origins, const float *dirs, const float *tmax, uint32_t n,
                        uint8_t *out, int brute) {
    for (uint32_t i = 0; i < n; ++i)
        out[i] = (uint8_t)occluded_one(s, origins + 3 * (size_t)i, dirs + 3 * (size_t)i, tmax[i], brute, NULL);
}

/* ------------------------------------------------------------------ SPEC §9 textures / environment */
static inline int wrap_i(int x, int n) { int m = x % n; return m < 0 ? m + n : m; }
void orc_texture_lookup(const orc_scene *s, uint32_t image, float u, float v, int srgb, float out[4]) {
    const orc_image *im = &s->images[image];
    int W = (int)im->width, H = (int)im->height;
    float fx = u * (float)W - 0.5f, fy = v * (float)H - 0.5f;
    float x0f = floorf(fx), y0f = floorf(fy);
    float tx = fx - x0f, ty = fy - y0f;
    int x0 = wrap_i((int)x0f, W), x1 = wrap_i((int)x0f + 1, W);
    int y0 = wrap_i((int)y0f, H), y1 = wrap_i((int)y0f + 1, H);
    const uint8_t *p00 = im->rgba8 + 4 * ((size_t)y0 * W + x0), *p10 = im->rgba8 + 4 * ((size_t)y0 * W + x1);
    const uint8_t *p01 = im->rgba8 + 4 * ((size_t)y1 * W + x0), *p11 = im->rgba8 + 4 * ((size_t)y1 * W + x1);
    for (int ch = 0; ch < 4; ++ch) {
        float c00, c10, c01, c11;
        if (srgb && ch < 3) { c00 = s->srgb_lut[p00[ch]]; c10 = s->srgb_lut[p10[ch]]; c01 = s->srgb_lut[p01[ch]]; c11 = s->srgb_lut[p11[ch]]; }
        else {
            c00 = (float)p00[ch] * 0.003921568859368563f; c10 = (float)p10[ch] * 0.003921568859368563f;
            c01 = (float)p01[ch] * 0.003921568859368563f; c11 = (float)p11[ch] * 0.003921568859368563f;
        }
        float top = c00 * (1.0f - tx) + c10 * tx, bot = c01 * (1.0f - tx) + c11 * tx;
        out[ch] = top * (1.0f - ty) + bot * ty;
    }
}
static inline void rgbe_decode(const uint8_t *p, float rgb[3]) {
    uint32_t e = p[3];
    float scale = 0.0f;
    if (e >= 10u) { uint32_t bits = (e - 9u) << 23; memcpy(&scale, &bits, 4); } /* 2^(e-136) */
    rgb[0] = (float)p[0] * scale; rgb[1] = (float)p[1] * scale; rgb[2] = (float)p[2] * scale;
}
void orc_env_lookup(const orc_scene *s, const float d[3], float rgb[3]) {
    int W = (int)s->probe_w, H = (int)s->probe_h;
    if (W == 1 && H == 1) { rgbe_decode(s->probe, rgb); return; }
    float phi = orc_atan2(d[2], d[0]);
    float u = phi * ORC_INV_2PI + 0.5f;
    float th = orc_acos(clampf(d[1], -1.0f, 1.0f));
    float v = th * ORC_INV_PI;
    float fx = u * (float)W - 0.5f, fy = v * (float)H - 0.5f;
    float x0f = floorf(fx), y0f = floorf(fy);
    float tx = fx - x0f, ty = fy - y0f;
    int x0 = wrap_i((int)x0f, W), x1 = wrap_i((int)x0f + 1, W);
    int y0 = (int)y0f, y1 = (int)y0f + 1;
    if (y0 < 0) y0 = 0; if (y0 > H - 1) y0 = H - 1;
    if (y1 < 0) y1 = 0; if (y1 > H - 1) y1 = H - 1;
    float c00[3], c10[3], c01[3], c11[3];
    rgbe_decode(s->probe + 4 * ((size_t)y0 * W + x0), c00); rgbe_decode(s->probe + 4 * ((size_t)y0 * W + x1), c10);
    rgbe_decode(s->probe + 4 * ((size_t)y1 * W + x0), c01); rgbe_decode(s->probe + 4 * ((size_t)y1 * W + x1), c11);
    for (int ch = 0; ch < 3; ++ch) {
        float top = c00[ch] * (1.0f - tx) + c10[ch] * tx, bot = c01[ch] * (1.0f - tx) + c11[ch] * tx;
        rgb[ch] = top * (1.0f - ty) + bot * ty;
    }
}

/* ------------------------------------------------------------------ SPEC §10 BSDF */
typedef struct { v3 diff, f0; float alpha, a2; } surf_t;
static inline surf_t make_surface(const float base[3], float roughness, float metallic) {
    surf_t s;
    float r = clampf(roughness, ORC_MIN_ROUGHNESS, 1.0f);
    float m = clampf(metallic, 0.0f, 1.0f);
    s.alpha = r * r; s.a2 = s.alpha * s.alpha;
    float om = 1.0f - m;
    s.diff = V3(base[0] * om, base[1] * om, base[2] * om);
    s.f0 = V3(0.04f * om + base[0] * m, 0.04f * om + base[1] * m, 0.04f * om + base[2] * m);
    return s;
}
static inline float pow5(float m) { float m2 = m * m; return (m2 * m2) * m; }
static inline float lum3(v3 c) { return (0.2126f * c.x + 0.7152f * c.y) + 0.0722f * c.z; }
/* probability of picking the specular lobe, from the view direction only */
static inline float spec_probability(const surf_t *s, float NoV) {
    float fc = pow5(1.0f - NoV);
    v3 Fv = V3(s->f0.x + (1.0f - s->f0.x) * fc, s->f0.y + (1.0f - s->f0.y) * fc, s->f0.z + (1.0f - s->f0.z) * fc);
    float ws = lum3(Fv);
    float wd = lum3(s->diff) * (1.0f - ws);
    if (!(wd > 0.0f)) return 1.0f;
    return clampf(ws / (ws + wd), 0.1f, 0.9f);
}
static inline void bsdf_eval(const surf_t *s, v3 N, v3 Ng, v3 V, float NoV, float pspec, v3 L, v3 *f, float *pdf) {
    *f = V3(0.0f, 0.0f, 0.0f); *pdf = 0.0f;
    float NoL = dot3(N, L);
    if (!(NoL > 0.0f) || !(dot3(Ng, L) > 0.0f)) return;
    v3 H = normalize3(add3(V, L));
    float NoH = fmax2(dot3(N, H), 0.0f);
    float VoH = fmax2(dot3(V, H), 0.0f);
    float dd = (NoH * NoH) * (s->a2 - 1.0f) + 1.0f;
    float D = s->a2 / (ORC_PI * (dd * dd));
    float k = s->alpha * 0.5f;
    float gl = NoL * (1.0f - k) + k, gv = NoV * (1.0f - k) + k;
    float vis = 1.0f / (4.0f * (gl * gv));
    float fc = pow5(1.0f - VoH);
    v3 F = V3(s->f0.x + (1.0f - s->f0.x) * fc, s->f0.y + (1.0f - s->f0.y) * fc, s->f0.z + (1.0f - s->f0.z) * fc);
    float dv = D * vis;
    f->x = (s->diff.x * ORC_INV_PI) * (1.0f - F.x) + dv * F.x;
    f->y = (s->diff.y * ORC_INV_PI) * (1.0f - F.y) + dv * F.y;
    f->z = (s->diff.z * ORC_INV_PI) * (1.0f - F.z) + dv * F.z;
    float pdf_d = NoL * ORC_INV_PI;
    float pdf_s = VoH > 0.0f ? (D * NoH) / (4.0f * VoH) : 0.0f;
    *pdf = pspec * pdf_s + (1.0f - pspec) * pdf_d;
}
static inline int bsdf_sample(const surf_t *s, v3 N, v3 Ng, v3 V, float NoV, float pspec, float r3, float r4, float r5,
                              v3 *Lout, v3 *weight, float *pdf) {
    float nn[3] = {N.x, N.y, N.z}, tt[3], bb[3];
    orc_onb(nn, tt, bb);
    v3 T = V3(tt[0], tt[1], tt[2]), B = V3(bb[0], bb[1], bb[2]);
    float sn, cs;
    orc_sincos2pi(r5, &sn, &cs);
    v3 L;
    if (r3 < pspec) {
        float cos2 = (1.0f - r4) / (1.0f + (s->a2 - 1.0f) * r4);
        float ct = sqrtf(cos2);
        float st = sqrtf(fmax2(0.0f, 1.0f - cos2));
        float hx = st * cs, hy = st * sn;
        v3 H = add3(add3(mul3(T, hx), mul3(B, hy)), mul3(N, ct));
        float vh2 = 2.0f * dot3(V, H);
        L = V3(vh2 * H.x - V.x, vh2 * H.y - V.y, vh2 * H.z - V.z);
    } else {
        float r = sqrtf(r4);
        float lx = r * cs, ly = r * sn, lz = sqrtf(fmax2(0.0f, 1.0f - r4));
        L = add3(add3(mul3(T, lx), mul3(B, ly)), mul3(N, lz));
    }
    L = normalize3(L);
    v3 f; float p;
    bsdf_eval(s, N, Ng, V, NoV, pspec, L, &f, &p);
    if (!(p > 0.0f)) return 0;
    float NoL = dot3(N, L);
    float w = NoL / p;
    *weight = V3(f.x * w, f.y * w, f.z * w);
    *Lout = L; *pdf = p;
    return 1;
}
void orc_bsdf_eval(const float base[3], float roughness, float metallic, const float N[3], const float Ng[3],
                   const float V[3], const float L[3], float f[3], float *pdf) {
    surf_t s = make_surface(base, roughness, metallic);
    v3 n = V3(N[0], N[1], N[2]), v = V3(V[0], V[1], V[2]);
    float NoV = fmax2(dot3(n, v), ORC_MIN_NOV);
    float ps = spec_probability(&s, NoV);
    v3 ff;
    bsdf_eval(&s, n, V3(Ng[0], Ng[1], Ng[2]), v, NoV, ps, V3(L[0], L[1], L[2]), &ff, pdf);
    f[0] = ff.x; f[1] = ff.y; f[2] = ff.z;
}
int orc_bsdf_sample(const float base[3], float roughness, float metallic, const float N[3], const float Ng[3],
                    const float V[3], float r3, float r4, float r5, float L[3], float weight[3], float *pdf) {
    surf_t s = make_surface(base, roughness, metallic);
    v3 n = V3(N[0], N[1], N[2]), v = V3(V[0], V[1], V[2]);
    float NoV = fmax2(dot3(n, v), ORC_MIN_NOV);
    float ps = spec_probability(&s, NoV);
    v3 l, w;
    int ok = bsdf_sample(&s, n, V3(Ng[0], Ng[1], Ng[2]), v, NoV, ps, r3, r4, r5, &l, &w, pdf);
    if (ok) { L[0] = l.x; L[1] = l.y; L[2] = l.z; weight[0] = w.x; weight[1] = w.y; weight[2] = w.z; }
    return ok;
}

/* ------------------------------------------------------------------ SPEC §11 camera */
typedef struct { v3 origin, right, up, fwd; float ax, ay; } cam_t;
static cam_t make_camera(const orc_render_params *p) {
    cam_t c;
    const float *m = p->view;
    c.right = V3(m[0], m[1], m[2]); c.up = V3(m[4], m[5], m[6]); c.fwd = V3(m[8], m[9], m[10]);
    c.origin = V3(m[12], m[13], m[14]);
    float th = tanf(0.5f * p->vfov);
    float aspect = (float)p->width / (float)p->height;
    c.ax = aspect * th; c.ay = th;
    return c;
}
static inline void noise_shift(const orc_scene *s, const orc_render_params *p, uint32_t x, uint32_t y, uint32_t seed_counter,
                               float *r0, float *r1) {
    /* SPEC §4.3: blue-noise Cranley-Patterson shift of the first two dimensions of a stage */
    if (!p->use_noise || !s || !s->noise) return;
    const uint8_t *t = s->noise + 4 * ((size_t)(y % s->noise_h) * s->noise_w + (x % s->noise_w));
    float g = (float)(seed_counter & 1023u) * 0.61803398875f;
    float a = *r0 + ((float)t[0] + 0.5f) * 0.00390625f + g;
    float b = *r1 + ((float)t[1] + 0.5f) * 0.00390625f + g;
    a = a - floorf(a); b = b - floorf(b);
    if (a >= 1.0f) a = 0.0f; if (b >= 1.0f) b = 0.0f;
    *r0 = a; *r1 = b;
}
static void raygen(const orc_scene *s, const orc_render_params *p, const cam_t *c, uint32_t x, uint32_t y, uint32_t seed_counter,
                   v3 *o, v3 *d) {
    uint32_t pixel = y * p->width + x;
    rng_t r = rng_init(pixel, stage_seed(p->user_seed, seed_counter), ORC_TAG_RAYGEN);
    float jx = rng_next(&r), jy = rng_next(&r);
    noise_shift(s, p, x, y, seed_counter, &jx, &jy);
    float sx = ((float)x + jx) / (float)p->width;
    float sy = ((float)y + jy) / (float)p->height;
    float cx = (2.0f * sx - 1.0f) * c->ax;
    float cy = (1.0f - 2.0f * sy) * c->ay;
    v3 dir = V3((c->right.x * cx + c->up.x * cy) + c->fwd.x, (c->right.y * cx + c->up.y * cy) + c->fwd.y,
                (c->right.z * cx + c->up.z * cy) + c->fwd.z);
    *o = c->origin; *d = normalize3(dir);
}
void orc_raygen(const orc_render_params *p, uint32_t x, uint32_t y, float origin[3], float dir[3]) {
    cam_t c = make_camera(p);
    v3 o, d;
    raygen(NULL, p, &c, x, y, p->seed_counter, &o, &d);
    origin[0] = o.x; origin[1] = o.y; origin[2] = o.z; dir[0] = d.x; dir[1] = d.y; dir[2] = d.z;
}

/* ------------------------------------------------------------------ SPEC §12 one sample of one pixel */
/* primary-hit record for the G-buffer (SPEC §15.1) */
typedef struct { uint32_t prim; float depth; v3 n; float albedo[3]; v3 P; int has_P; } primary_t;

static void trace_sample(const orc_scene *s, const orc_render_params *p, const cam_t *cam, uint32_t x, uint32_t y,
                         uint32_t seed_counter, float Lout[3], orc_counters *c, primary_t *pr) {
    uint32_t pixel = y * p->width + x;
    v3 o, d;
    raygen(s, p, cam, x, y, seed_counter, &o, &d);
    v3 T = V3(1.0f, 1.0f, 1.0f), Lsum = V3(0.0f, 0.0f, 0.0f);
    float pdf_prev = -1.0f; /* <0: camera ray, emission counted in full */
    float inv_nl = s->n_lights ? 1.0f / (float)s->n_lights : 0.0f;
    for (uint32_t b = 0; b < p->max_bounces; ++b) {
        seed_counter += 1u; /* renderer.rs:453,487 */
        float oo[3] = {o.x, o.y, o.z}, dd[3] = {d.x, d.y, d.z};
        orc_hit h;
        closest_one(s, oo, dd, &h, (int)p->brute_force, c);
        if (c) c->closest++;
        if (pr && b == 0u) { /* defaults: miss / emitter / degenerate */
            pr->prim = h.prim; pr->depth = h.t; pr->n = neg3(d);
            pr->albedo[0] = pr->albedo[1] = pr->albedo[2] = 1.0f;
            pr->has_P = h.prim != ORC_INVALID;
            pr->P = V3(fmaf(d.x, h.t, o.x), fmaf(d.y, h.t, o.y), fmaf(d.z, h.t, o.z));
            if ((h.prim & ORC_LIGHT_BIT) && h.prim != ORC_INVALID) {
                const orc_light *Lg = &s->lights[h.prim & ~ORC_LIGHT_BIT];
                pr->n = V3(Lg->normal[0], Lg->normal[1], Lg->normal[2]);
            }
        }
        if (h.prim == ORC_INVALID) { /* miss: environment */
            float e[3];
            orc_env_lookup(s, dd, e);
            Lsum = V3(Lsum.x + T.x * e[0], Lsum.y + T.y * e[1], Lsum.z + T.z * e[2]);
            break;
        }
        if (h.prim & ORC_LIGHT_BIT) { /* emitter hit by a BSDF / camera ray */
            const orc_light *L = &s->lights[h.prim & ~ORC_LIGHT_BIT];
            float Le = L->origin[3];
            float w = 1.0f;
            if (pdf_prev >= 0.0f) {
                float cl = -dot3(V3(L->normal[0], L->normal[1], L->normal[2]), d);
                float area = 4.0f * (L->tangent[3] * L->bitangent[3]);
                float pl = ((h.t * h.t) / (cl * area)) * inv_nl;
                float pb2 = pdf_prev * pdf_prev;
                w = pb2 / (pb2 + pl * pl);
            }
            float k = Le * w;
            Lsum = V3(Lsum.x + T.x * k, Lsum.y + T.y * k, Lsum.z + T.z * k);
            break;
        }
        if (c) c->shaded++;
        /* ---- surface */
        const orc_vertex *v0 = &s->verts[3 * (size_t)h.prim], *v1 = v0 + 1, *v2 = v0 + 2;
        float bw = (1.0f - h.u) - h.v;
        v3 p0 = V3(v0->position[0], v0->position[1], v0->position[2]);
        v3 p1 = V3(v1->position[0], v1->position[1], v1->position[2]);
        v3 p2 = V3(v2->position[0], v2->position[1], v2->position[2]);
        v3 P = V3((p0.x * bw + p1.x * h.u) + p2.x * h.v, (p0.y * bw + p1.y * h.u) + p2.y * h.v, (p0.z * bw + p1.z * h.u) + p2.z * h.v);
        v3 Ng = cross3(sub3(p1, p0), sub3(p2, p0));
        float l2 = dot3(Ng, Ng);
        if (!(l2 > 0.0f)) break;
        Ng = mul3(Ng, 1.0f / sqrtf(l2));
        v3 Ns = V3((v0->normal[0] * bw + v1->normal[0] * h.u) + v2->normal[0] * h.v,
                   (v0->normal[1] * bw + v1->normal[1] * h.u) + v2->normal[1] * h.v,
                   (v0->normal[2] * bw + v1->normal[2] * h.u) + v2->normal[2] * h.v);
        float n2 = dot3(Ns, Ns);
        Ns = n2 > 0.0f ? mul3(Ns, 1.0f / sqrtf(n2)) : Ng;
        if (dot3(Ng, d) > 0.0f) Ng = neg3(Ng);
        if (dot3(Ns, Ng) < 0.0f) Ns = neg3(Ns);
        float tu = (v0->position[3] * bw + v1->position[3] * h.u) + v2->position[3] * h.v;
        float tv = (v0->normal[3] * bw + v1->normal[3] * h.u) + v2->normal[3] * h.v;
        uint32_t mi = s->tri_material[h.prim];
        if (mi >= s->n_materials) mi = 0;
        const orc_material *M = &s->materials[mi];
        float base[3] = {M->color[0], M->color[1], M->color[2]};
        float rough = M->roughness, metal = M->reflectivity;
        if (M->albedo_texture < s->n_images) {
            float tex[4];
            orc_texture_lookup(s, M->albedo_texture, tu, tv, 1, tex);
            base[0] *= tex[0]; base[1] *= tex[1]; base[2] *= tex[2];
        }
        if (M->mra_texture < s->n_images) {
            float tex[4];
            orc_texture_lookup(s, M->mra_texture, tu, tv, 0, tex);
            rough *= tex[1]; metal *= tex[2];
        }
        if (pr && b == 0u) {
            pr->n = Ns; pr->P = P;
            pr->albedo[0] = clampf(base[0], 0.0f, 1.0f); pr->albedo[1] = clampf(base[1], 0.0f, 1.0f); pr->albedo[2] = clampf(base[2], 0.0f, 1.0f);
        }
        surf_t sf = make_surface(base, rough, metal);
        v3 Vv = neg3(d);
        float NoV = fmax2(dot3(Ns, Vv), ORC_MIN_NOV);
        float pspec = spec_probability(&sf, NoV);
        /* random numbers of this stage, fixed order (SPEC §4.2) */
        rng_t rg = rng_init(pixel, stage_seed(p->user_seed, seed_counter), ORC_TAG_SHADE);
        float r0 = rng_next(&rg), r1 = rng_next(&rg), r2 = rng_next(&rg);
        float r3 = rng_next(&rg), r4 = rng_next(&rg), r5 = rng_next(&rg);
        noise_shift(s, p, x, y, seed_counter, &r4, &r5);
        float am = fmax2(fmax2(fabsf(P.x), fabsf(P.y)), fabsf(P.z));
        float eps = 1.0e-4f * (1.0f + am);
        v3 Po = V3(P.x + Ng.x * eps, P.y + Ng.y * eps, P.z + Ng.z * eps);
        /* ---- next-event estimation (SPEC §12.3) */
        if (s->n_lights) {
            uint32_t li = (uint32_t)(r0 * (float)s->n_lights);
            if (li > s->n_lights - 1u) li = s->n_lights - 1u;
            const orc_light *L = &s->lights[li];
            float Le = L->origin[3];
            float hw = L->tangent[3], hh = L->bitangent[3];
            float a = (2.0f * r1 - 1.0f) * hw, bq = (2.0f * r2 - 1.0f) * hh;
            v3 q = V3((L->origin[0] + L->tangent[0] * a) + L->bitangent[0] * bq,
                      (L->origin[1] + L->tangent[1] * a) + L->bitangent[1] * bq,
                      (L->origin[2] + L->tangent[2] * a) + L->bitangent[2] * bq);
            v3 w = sub3(q, Po);
            float d2 = dot3(w, w);
            if (Le > 0.0f && d2 > 0.0f) {
                float dist = sqrtf(d2);
                v3 wi = mul3(w, 1.0f / dist);
                float cl = -dot3(V3(L->normal[0], L->normal[1], L->normal[2]), wi);
                if (cl > 0.0f) {
                    v3 f; float pb;
                    bsdf_eval(&sf, Ns, Ng, Vv, NoV, pspec, wi, &f, &pb);
                    if (pb > 0.0f) {
                        float area = 4.0f * (hw * hh);
                        float pl = (d2 / (cl * area)) * inv_nl;
                        float pl2 = pl * pl;
                        float wm = pl2 / (pl2 + pb * pb);
                        float NoL = dot3(Ns, wi);
                        float k = ((NoL * Le) * wm) / pl;
                        v3 contrib = V3((T.x * f.x) * k, (T.y * f.y) * k, (T.z * f.z) * k);
                        if (contrib.x > 0.0f || contrib.y > 0.0f || contrib.z > 0.0f) {
                            float so[3] = {Po.x, Po.y, Po.z}, sd[3] = {wi.x, wi.y, wi.z};
                            if (c) c->shadow++;
                            if (!occluded_one(s, so, sd, dist * 0.999f, (int)p->brute_force, c))
                                Lsum = add3(Lsum, contrib);
                        }
                    }
                }
            }
        }
        /* ---- BSDF sample (SPEC §12.4) */
        v3 Ln, wgt; float pdf;
        if (!bsdf_sample(&sf, Ns, Ng, Vv, NoV, pspec, r3, r4, r5, &Ln, &wgt, &pdf)) break;
        T = V3(T.x * wgt.x, T.y * wgt.y, T.z * wgt.z);
        if (!(T.x > 0.0f || T.y > 0.0f || T.z > 0.0f)) break;
        o = Po; d = Ln; pdf_prev = pdf;
    }
    Lout[0] = Lsum.x; Lout[1] = Lsum.y; Lout[2] = Lsum.z;
}

/* ------------------------------------------------------------------ driver */
typedef struct {
    const orc_scene *s; const orc_render_params *p; float *accum; cam_t cam;
    atomic_uint next_row; orc_counters *counters; pthread_mutex_t lock;
    uint32_t x0, y0, x1, y1;
} job_t;

static int owns_pixel(const orc_render_params *p, uint32_t x, uint32_t y) {
    if (p->world_size <= 1u) return 1;
    uint32_t tw = p->tile_w ? p->tile_w : 32u, th = p->tile_h ? p->tile_h : 8u;
    uint32_t tiles_x = (p->width + tw - 1u) / tw;
    uint32_t tile = (y / th) * tiles_x + (x / tw);
    return (tile % p->world_size) == p->rank;
}

static void *worker(void *arg) {
    job_t *j = (job_t *)arg;
    const orc_render_params *p = j->p;
    orc_counters local; memset(&local, 0, sizeof local);
    /* work items: 16x16 pixel tiles of the crop window, handed out by an atomic counter */
    const uint32_t TW = 16u, TH = 16u;
    const uint32_t ntx = (j->x1 - j->x0 + TW - 1u) / TW, nty = (j->y1 - j->y0 + TH - 1u) / TH;
    for (;;) {
        uint32_t tile = atomic_fetch_add(&j->next_row, 1u);
        if (tile >= ntx * nty) break;
        const uint32_t ty0 = j->y0 + (tile / ntx) * TH, tx0 = j->x0 + (tile % ntx) * TW;
        const uint32_t ty1 = ty0 + TH < j->y1 ? ty0 + TH : j->y1, tx1 = tx0 + TW < j->x1 ? tx0 + TW : j->x1;
        for (uint32_t y = ty0; y < ty1; ++y)
        for (uint32_t x = tx0; x < tx1; ++x) {
            if (!owns_pixel(p, x, y)) continue;
            float *acc = j->accum + 4 * ((size_t)y * p->width + x);
            uint32_t seed = p->seed_counter;
            uint32_t frame_count = 1; /* reset_accumulation(): renderer.rs:610 */
            for (uint32_t f = 0; f < p->frames; ++f) {
                float L[3];
                trace_sample(j->s, p, &j->cam, x, y, seed, L, j->counters ? &local : NULL, NULL);
                seed += p->max_bounces;
                /* AccumulationPass (renderer.rs:525-537), stored as (sum, count) */
                if (frame_count == 1u) { acc[0] = L[0]; acc[1] = L[1]; acc[2] = L[2]; acc[3] = 1.0f; }
                else { acc[0] += L[0]; acc[1] += L[1]; acc[2] += L[2]; acc[3] += 1.0f; }
                /* accumulate == true for every emulated frame: frame_count += 1 (renderer.rs:535-537) */
                frame_count += 1u;
            }
        }
    }
    if (j->counters) {
        pthread_mutex_lock(&j->lock);
        j->counters->closest += local.closest; j->counters->shadow += local.shadow; j->counters->shaded += local.shaded;
        j->counters->nodes += local.nodes; j->counters->tris += local.tris;
        pthread_mutex_unlock(&j->lock);
    }
    return NULL;
}

/* Persistent worker pool: threads are created once and parked on a condition variable between renders (a frame of
   the bench workload is a few hundred milliseconds on a large host; creating 255 threads per frame is not free). */
static struct {
    pthread_mutex_t m; pthread_cond_t work, done;
    pthread_t th[256]; uint64_t start_gen[256]; uint32_t n_threads;
    job_t *job; uint64_t gen; uint32_t want, pending;
    pthread_mutex_t render_lock;
} g_pool = {PTHREAD_MUTEX_INITIALIZER, PTHREAD_COND_INITIALIZER, PTHREAD_COND_INITIALIZER, {0}, {0}, 0, NULL, 0, 0, 0, PTHREAD_MUTEX_INITIALIZER};

static void *pool_main(void *arg) {
    const uint32_t me = (uint32_t)(uintptr_t)arg;   /* 1-based: thread 0 is the caller */
    pthread_mutex_lock(&g_pool.m);
    uint64_t seen = g_pool.start_gen[me];           /* the generation before the render that created this thread */
    for (;;) {
        while (g_pool.gen == seen) pthread_cond_wait(&g_pool.work, &g_pool.m);
        seen = g_pool.gen;
        if (me >= g_pool.want) continue;            /* this render uses fewer threads */
        job_t *j = g_pool.job;
        pthread_mutex_unlock(&g_pool.m);
        worker(j);
        pthread_mutex_lock(&g_pool.m);
        if (--g_pool.pending == 0) pthread_cond_signal(&g_pool.done);
    }
    return NULL;
}

uint32_t orc_render(const orc_scene *s, const orc_render_params *p, float *accum, orc_counters *c) {
    job_t j;
    j.s = s; j.p = p; j.accum = accum; j.cam = make_camera(p); j.counters = c;
    atomic_init(&j.next_row, 0u);
    pthread_mutex_init(&j.lock, NULL);
    j.x0 = p->x0; j.y0 = p->y0; j.x1 = p->x1; j.y1 = p->y1;
    if (j.x1 == 0u && j.y1 == 0u) { j.x0 = j.y0 = 0u; j.x1 = p->width; j.y1 = p->height; }
    uint32_t nt = p->threads ? p->threads : 1u;
    if (nt > 256u) nt = 256u;
    pthread_mutex_lock(&g_pool.render_lock);         /* one render at a time owns the pool */
    pthread_mutex_lock(&g_pool.m);
    while (g_pool.n_threads + 1u < nt) {             /* grow the pool to nt - 1 helpers */
        const uint32_t idx = g_pool.n_threads + 1u;
        g_pool.start_gen[idx] = g_pool.gen;
        if (pthread_create(&g_pool.th[idx], NULL, pool_main, (void *)(uintptr_t)idx) != 0) { nt = idx; break; }
        pthread_detach(g_pool.th[idx]);
        g_pool.n_threads = idx;
    }
    g_pool.job = &j; g_pool.want = nt; g_pool.pending = nt - 1u; g_pool.gen++;
    pthread_cond_broadcast(&g_pool.work);
    pthread_mutex_unlock(&g_pool.m);
    worker(&j);
    pthread_mutex_lock(&g_pool.m);
    while (g_pool.pending) pthread_cond_wait(&g_pool.done, &g_pool.m);
    pthread_mutex_unlock(&g_pool.m);
    pthread_mutex_unlock(&g_pool.render_lock);
    pthread_mutex_destroy(&j.lock);
    return p->seed_counter + p->frames * p->max_bounces;
}

void orc_resolve(const float *accum, uint32_t n, float *mean) {
    for (uint32_t i = 0; i < n; ++i) {
        float cnt = accum[4 * (size_t)i + 3];
        if (cnt > 0.0f) {
            mean[4 * (size_t)i + 0] = accum[4 * (size_t)i + 0] / cnt;
            mean[4 * (size_t)i + 1] = accum[4 * (size_t)i + 1] / cnt;
            mean[4 * (size_t)i + 2] = accum[4 * (size_t)i + 2] / cnt;
            mean[4 * (size_t)i + 3] = 1.0f;
        } else { mean[4 * (size_t)i + 0] = mean[4 * (size_t)i + 1] = mean[4 * (size_t)i + 2] = mean[4 * (size_t)i + 3] = 0.0f; }
    }
}
/* SPEC §13.2: the sRGB OETF rounded to 8 bits, evaluated EXACTLY: code i is produced from the linear value where the ideal
   curve crosses (i - 0.5) / 255 — the inverse OETF in binary64, rounded to binary32 once — so the result is a table search on
   float comparisons and does not depend on anybody's powf. */
static float g_srgb_thr[256];
static int g_srgb_thr_ready = 0;
void orc_srgb_thresholds(float out[256]) {
    out[0] = 0.0f;
    for (int i = 1; i < 256; ++i) {
        const double s = ((double)i - 0.5) / 255.0;
        out[i] = (float)(s <= 0.04045 ? s / 12.92 : pow((s + 0.055) / 1.055, 2.4));
    }
}
static uint8_t encode_srgb8(float c) {
    if (!g_srgb_thr_ready) { orc_srgb_thresholds(g_srgb_thr); g_srgb_thr_ready = 1; }
    c = clampf(c, 0.0f, 1.0f);
    uint32_t idx = 0;
    for (uint32_t step = 128u; step; step >>= 1)
        if (idx + step <= 255u && c >= g_srgb_thr[idx + step]) idx += step;
    return (uint8_t)idx;
}
void orc_tonemap(const float *accum, uint32_t n, uint8_t *rgba8) {
    for (uint32_t i = 0; i < n; ++i) {
        float cnt = accum[4 * (size_t)i + 3];
        for (int ch = 0; ch < 3; ++ch) rgba8[4 * (size_t)i + ch] = encode_srgb8(cnt > 0.0f ? accum[4 * (size_t)i + ch] / cnt : 0.0f);
        rgba8[4 * (size_t)i + 3] = 255;
    }
}

/* ================================================================== SPEC §15: denoiser path
 * Restates the pass sequence of reference crates/lib/src/render/asvgf.rs:250-291 (temporal ->
 * copy -> a-trous x even count -> composite; ping-pong resources :9-152) and of
 * Renderer::raytrace in BlitMode::DenoisedPathrace / Temporal (renderer.rs:466-481,512-522,542-546).
 * The per-pixel filters are specified in SPEC.md §15 (the reference's live in albedo_rtx). */
struct orc_denoiser {
    uint32_t w, h; int cur;
    uint32_t *gbuf[2]; float *rad[2]; float *mom[2]; uint32_t *hist[2];
    float *motion, *temp, *lsum;
    cam_t prev_cam;
};

orc_denoiser *orc_denoiser_create(uint32_t w, uint32_t h) {
    orc_denoiser *d = (orc_denoiser *)calloc(1, sizeof *d);
    size_t n = (size_t)w * h;
    d->w = w; d->h = h; d->cur = 1; /* current_frame_back starts true (asvgf.rs:233), start() flips it */
    for (int k = 0; k < 2; ++k) {
        d->gbuf[k] = (uint32_t *)calloc(n * 4, 4); d->rad[k] = (float *)calloc(n * 4, 4);
        d->mom[k] = (float *)calloc(n * 2, 4); d->hist[k] = (uint32_t *)calloc(n, 4);
    }
    d->motion = (float *)calloc(n * 2, 4); d->temp = (float *)calloc(n * 4, 4); d->lsum = (float *)calloc(n * 4, 4);
    /* prev_model_to_screen starts as the identity (renderer.rs:319) */
    d->prev_cam.origin = V3(0, 0, 0); d->prev_cam.right = V3(1, 0, 0); d->prev_cam.up = V3(0, 1, 0); d->prev_cam.fwd = V3(0, 0, 1);
    d->prev_cam.ax = d->prev_cam.ay = 1.0f;
    return d;
}
void orc_denoiser_destroy(orc_denoiser *d) {
    if (!d) return;
    for (int k = 0; k < 2; ++k) { free(d->gbuf[k]); free(d->rad[k]); free(d->mom[k]); free(d->hist[k]); }
    free(d->motion); free(d->temp); free(d->lsum); free(d);
}
void orc_denoiser_read(const orc_denoiser *d, uint32_t *gbuf_cur, float *motion, float *rad_cur, uint32_t *hist_cur) {
    size_t n = (size_t)d->w * d->h;
    if (gbuf_cur) memcpy(gbuf_cur, d->gbuf[d->cur], n * 16);
    if (motion) memcpy(motion, d->motion, n * 8);
    if (rad_cur) memcpy(rad_cur, d->rad[d->cur], n * 16);
    if (hist_cur) memcpy(hist_cur, d->hist[d->cur], n * 4);
}

static inline uint32_t oct_encode(v3 n) {
    float l1 = (fabsf(n.x) + fabsf(n.y)) + fabsf(n.z);
    float px = 0.0f, py = 0.0f;
    if (l1 > 0.0f) { px = n.x / l1; py = n.y / l1; }
    if (n.z < 0.0f) {
        float tx = (1.0f - fabsf(py)) * (px >= 0.0f ? 1.0f : -1.0f);
        float ty = (1.0f - fabsf(px)) * (py >= 0.0f ? 1.0f : -1.0f);
        px = tx; py = ty;
    }
    uint32_t ux = (uint32_t)(clampf(px * 0.5f + 0.5f, 0.0f, 1.0f) * 65535.0f + 0.5f);
    uint32_t uy = (uint32_t)(clampf(py * 0.5f + 0.5f, 0.0f, 1.0f) * 65535.0f + 0.5f);
    return ux | (uy << 16);
}
static inline v3 oct_decode(uint32_t p) {
    float fx = (float)(p & 0xFFFFu) * 3.0518043793392844e-05f - 1.0f;
    float fy = (float)(p >> 16) * 3.0518043793392844e-05f - 1.0f;
    float fz = (1.0f - fabsf(fx)) - fabsf(fy);
    if (fz < 0.0f) {
        float tx = (1.0f - fabsf(fy)) * (fx >= 0.0f ? 1.0f : -1.0f);
        float ty = (1.0f - fabsf(fx)) * (fy >= 0.0f ? 1.0f : -1.0f);
        fx = tx; fy = ty;
    }
    return normalize3(V3(fx, fy, fz));
}
static inline uint32_t pack_albedo(const float a[3]) {
    uint32_t r = (uint32_t)(clampf(a[0], 0.0f, 1.0f) * 255.0f + 0.5f), g = (uint32_t)(clampf(a[1], 0.0f, 1.0f) * 255.0f + 0.5f);
    uint32_t b = (uint32_t)(clampf(a[2], 0.0f, 1.0f) * 255.0f + 0.5f);
    return r | (g << 8) | (b << 16) | 0xFF000000u;
}
static inline v3 demod_albedo(uint32_t p) {
    return V3(fmax2((float)(p & 0xFFu) * 0.003921568859368563f, 0.05f), fmax2((float)((p >> 8) & 0xFFu) * 0.003921568859368563f, 0.05f),
              fmax2((float)((p >> 16) & 0xFFu) * 0.003921568859368563f, 0.05f));
}
static inline int project(const cam_t *c, v3 P, float *u, float *v) {
    v3 w = sub3(P, c->origin);
    float cz = dot3(w, c->fwd);
    if (!(cz > 1.0e-6f)) return 0;
    float cx = dot3(w, c->right), cy = dot3(w, c->up);
    *u = 0.5f + 0.5f * (cx / (cz * c->ax));
    *v = 0.5f - 0.5f * (cy / (cz * c->ay));
    return 1;
}
static inline float pow128(float x) { x = x * x; x = x * x; x = x * x; x = x * x; x = x * x; x = x * x; x = x * x; return x; }

static void atrous_pass(const orc_denoiser *d, const uint32_t *gb, const float *in, float *out, int step) {
    static const float kw[5] = {0.0625f, 0.25f, 0.375f, 0.25f, 0.0625f};
    const int W = (int)d->w, H = (int)d->h;
    for (int y = 0; y < H; ++y)
        for (int x = 0; x < W; ++x) {
            size_t i = (size_t)y * W + x;
            const uint32_t *g = gb + 4 * i;
            const float *c = in + 4 * i;
            if (g[0] == ORC_INVALID) { out[4 * i] = c[0]; out[4 * i + 1] = c[1]; out[4 * i + 2] = c[2]; out[4 * i + 3] = c[3]; continue; }
            v3 nc = oct_decode(g[2]);
            float zc; memcpy(&zc, &g[1], 4);
            float lc = lum3(V3(c[0], c[1], c[2]));
            float sigma_l = 4.0f * sqrtf(fmax2(c[3], 0.0f)) + 1.0e-4f;
            float sigma_z = 0.02f * zc + 1.0e-6f;
            float wc = kw[2] * kw[2];
            float sr = c[0] * wc, sg = c[1] * wc, sb = c[2] * wc, sv = c[3] * (wc * wc), sw = wc;
            for (int dy = -2; dy <= 2; ++dy)
                for (int dx = -2; dx <= 2; ++dx) {
                    if (dx == 0 && dy == 0) continue;
                    int qx = x + dx * step, qy = y + dy * step;
                    if (qx < 0 || qy < 0 || qx >= W || qy >= H) continue;
                    size_t j = (size_t)qy * W + qx;
                    const uint32_t *gq = gb + 4 * j;
                    if (gq[0] == ORC_INVALID) continue;
                    const float *q = in + 4 * j;
                    float zq; memcpy(&zq, &gq[1], 4);
                    float wn = pow128(fmax2(dot3(nc, oct_decode(gq[2])), 0.0f));
                    float rz = fabsf(zc - zq) / sigma_z;
                    float wz = 1.0f / (1.0f + rz * rz);
                    float rl = fabsf(lc - lum3(V3(q[0], q[1], q[2]))) / sigma_l;
                    float wl = 1.0f / (1.0f + rl * rl);
                    float w = ((kw[dx + 2] * kw[dy + 2]) * wn) * (wz * wl);
                    sr += q[0] * w; sg += q[1] * w; sb += q[2] * w; sv += q[3] * (w * w); sw += w;
                }
            float inv = 1.0f / sw;
            out[4 * i] = sr * inv; out[4 * i + 1] = sg * inv; out[4 * i + 2] = sb * inv; out[4 * i + 3] = sv * (inv * inv);
        }
}

/* one raytrace() call in BlitMode::DenoisedPathrace (mode 1) or Temporal (mode 2); out_main = w*h*4 floats */
void orc_denoise_frame(orc_denoiser *d, const orc_scene *s, const orc_render_params *p, int mode, float *out_main) {
    const int W = (int)d->w, H = (int)d->h;
    cam_t cam = make_camera(p);
    d->cur = 1 - d->cur; /* asvgf.start() (renderer.rs:467) */
    const int cur = d->cur, prv = 1 - d->cur;
    /* 1. path trace one sample per pixel; the primary pass also writes G-buffer and motion */
    for (int y = 0; y < H; ++y)
        for (int x = 0; x < W; ++x) {
            size_t i = (size_t)y * W + x;
            primary_t pr; memset(&pr, 0, sizeof pr);
            trace_sample(s, p, &cam, (uint32_t)x, (uint32_t)y, p->seed_counter, d->lsum + 4 * i, NULL, &pr);
            uint32_t *g = d->gbuf[cur] + 4 * i;
            g[0] = pr.prim; memcpy(&g[1], &pr.depth, 4); g[2] = oct_encode(pr.n); g[3] = pack_albedo(pr.albedo);
            float mu = 0.0f, mv = 0.0f, cu, cv, pu, pv;
            if (pr.has_P && project(&cam, pr.P, &cu, &cv) && project(&d->prev_cam, pr.P, &pu, &pv)) { mu = pu - cu; mv = pv - cv; }
            d->motion[2 * i] = mu; d->motion[2 * i + 1] = mv;
        }
    /* 2. TemporalAccumulationPass (asvgf.rs:245-247): nearest reprojection, consistency test, moments, history */
    for (int y = 0; y < H; ++y)
        for (int x = 0; x < W; ++x) {
            size_t i = (size_t)y * W + x;
            const uint32_t *g = d->gbuf[cur] + 4 * i;
            v3 a = demod_albedo(g[3]);
            const float *L = d->lsum + 4 * i;
            v3 il = V3(L[0] / a.x, L[1] / a.y, L[2] / a.z);
            float lm = lum3(il);
            int mx = (int)floorf(((float)x + 0.5f) + d->motion[2 * i] * (float)W);
            int my = (int)floorf(((float)y + 0.5f) + d->motion[2 * i + 1] * (float)H);
            uint32_t hn = 1u;
            v3 col = il; float m1 = lm, m2 = lm * lm;
            if (mx >= 0 && my >= 0 && mx < W && my < H) {
                size_t j = (size_t)my * W + mx;
                const uint32_t *gp = d->gbuf[prv] + 4 * j;
                uint32_t hp = d->hist[prv][j];
                float zc, zp; memcpy(&zc, &g[1], 4); memcpy(&zp, &gp[1], 4);
                int ok = hp > 0u && gp[0] == g[0] && dot3(oct_decode(g[2]), oct_decode(gp[2])) >= 0.9f &&
                         fabsf(zc - zp) <= 0.1f * fmax2(zc, zp);
                if (ok) {
                    hn = hp + 1u; if (hn > 64u) hn = 64u;
                    float al = 1.0f / (float)hn;
                    const float *pc = d->rad[prv] + 4 * j; const float *pm = d->mom[prv] + 2 * j;
                    col = V3(pc[0] + (il.x - pc[0]) * al, pc[1] + (il.y - pc[1]) * al, pc[2] + (il.z - pc[2]) * al);
                    m1 = pm[0] + (lm - pm[0]) * al; m2 = pm[1] + (lm * lm - pm[1]) * al;
                }
            }
            float var = fmax2(m2 - m1 * m1, 0.0f);
            if (hn < 4u) var = var + (m1 * m1) * ((float)(4u - hn) * 0.25f);
            float *rc = d->rad[cur] + 4 * i;
            rc[0] = col.x; rc[1] = col.y; rc[2] = col.z; rc[3] = var;
            d->mom[cur][2 * i] = m1; d->mom[cur][2 * i + 1] = m2; d->hist[cur][i] = hn;
        }
    /* 3. copy -> a-trous x4 (steps 1,2,4,8; main <-> temp, result in temp) -> composite (asvgf.rs:257-290) */
    size_t n = (size_t)W * H;
    const float *result = d->rad[cur];
    if (mode == 1) {
        memcpy(d->temp, d->rad[cur], n * 16);
        atrous_pass(d, d->gbuf[cur], d->temp, out_main, 1);
        atrous_pass(d, d->gbuf[cur], out_main, d->temp, 2);
        atrous_pass(d, d->gbuf[cur], d->temp, out_main, 4);
        atrous_pass(d, d->gbuf[cur], out_main, d->temp, 8);
        result = d->temp;
    }
    for (size_t i = 0; i < n; ++i) { /* CompositingPass: re-modulate with the primary albedo */
        v3 a = demod_albedo(d->gbuf[cur][4 * i + 3]);
        out_main[4 * i] = result[4 * i] * a.x; out_main[4 * i + 1] = result[4 * i + 1] * a.y; out_main[4 * i + 2] = result[4 * i + 2] * a.z;
        out_main[4 * i + 3] = 1.0f;
    }
    d->prev_cam = cam; /* prev_model_to_screen = P * V^-1 (renderer.rs:542-546) */
}
