/*
 * lpt_oracle.h — CPU ORACLE for the path-tracing hot path.  TEST INFRASTRUCTURE ONLY.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * build, load or call this.  The product (loupiote_amd/) never does.
 *
 * PARITY UNPINNED: the reference tree (DavidPeicho/loupiote @ 2025-03-01) holds
 * no integrator arithmetic, no tests and no golden vectors — BVH traversal,
 * ray/triangle, BSDF, RNG, NEE and accumulation live in the out-of-tree path
 * dependency `albedo_rtx 0.0.1-beta.0` (crates/lib/Cargo.toml:11,17,21;
 * Cargo.lock:50-82, no checksum / revision), which is absent here.  This file
 * therefore restates the *structure* the reference pins — stage order and the
 * seed / bounce / frame_count protocol of Renderer::raytrace
 * (crates/lib/src/renderer.rs:392-549), the flat scene arrays
 * (crates/lib/src/scene.rs:30-64), Material / Vertex layouts
 * (crates/lib/src/loaders/binary.rs:20-28,63-69), the camera-to-world convention
 * (crates/standalone/src/camera.rs:101-108) and RGBE8 probes
 * (crates/lib/src/scene.rs:79-114) — and the arithmetic written down in SPEC.md,
 * which is pinned instead by analytic known-answer tests (tests/test_oracle_*.py).
 *
 * Plain C11, scalar fp32, compiled with -ffp-contract=off so every rounding is
 * the one SPEC.md states; explicit fmaf() marks the fused operations.
 */
#ifndef LPT_ORACLE_H
#define LPT_ORACLE_H
#include <stddef.h>
#include <stdint.h>

#define ORC_INVALID 0xFFFFFFFFu
#define ORC_LIGHT_BIT 0x80000000u

/* same byte layouts as the boundary structs (SPEC.md §2), declared independently */
typedef struct { float color[4]; float roughness, reflectivity; uint32_t albedo_texture, mra_texture; } orc_material;
typedef struct { float position[4]; float normal[4]; } orc_vertex;           /* uv in the .w lanes */
typedef struct { float normal[4], tangent[4], bitangent[4], origin[4]; } orc_light;
typedef struct { float t, u, v; uint32_t prim; } orc_hit;
typedef struct { uint32_t width, height; const uint8_t *rgba8; } orc_image;

typedef struct orc_scene orc_scene;

typedef struct {
    uint32_t width, height;
    float view[16];            /* camera-to-world, column-major */
    float vfov;                /* radians */
    uint32_t max_bounces;      /* reference constant: 3 (renderer.rs:398-399) */
    uint32_t user_seed;
    uint32_t seed_counter;     /* global_uniforms.seed before the first frame (renderer.rs:288) */
    uint32_t frames;           /* number of raytrace() calls emulated (1 sample / pixel each) */
    uint32_t rank, world_size, tile_w, tile_h; /* pixel-tile shard; (0,1,*,*) = all */
    uint32_t threads;          /* worker threads (0 = 1) */
    uint32_t brute_force;      /* 1: no BVH, test every triangle */
    uint32_t use_noise;        /* RadianceParameters.use_noise_texture (renderer.rs:666-673) */
    /* crop: only pixels with x0<=x<x1, y0<=y<y1 are traced (all zero = full frame) */
    uint32_t x0, y0, x1, y1;
} orc_render_params;

typedef struct {
    uint64_t closest, shadow, shaded, nodes, tris;
} orc_counters;

/* ---- scene ---------------------------------------------------------------- */
/* tri_verts: 3*n_tris world-space vertices (baked instances); tri_material: n_tris ids */
orc_scene *orc_scene_create(uint32_t n_tris, const orc_vertex *tri_verts, const uint32_t *tri_material,
                            uint32_t n_materials, const orc_material *materials,
                            uint32_t n_lights, const orc_light *lights,
                            uint32_t n_images, const orc_image *images,
                            uint32_t probe_w, uint32_t probe_h, const uint8_t *probe_rgbe8 /* NULL = black 1x1 */);
void orc_scene_set_noise(orc_scene *s, const uint8_t *rgba8, uint32_t w, uint32_t h, uint32_t row_bytes);
void orc_scene_destroy(orc_scene *s);

/* ---- stages --------------------------------------------------------------- */
void orc_trace_closest(const orc_scene *s, const float *origins, const float *dirs, uint32_t n,
                       orc_hit *out, int brute_force, orc_counters *c);
void orc_trace_occluded(const orc_scene *s, const float *origins, const float *dirs, const float *tmax,
                        uint32_t n, uint8_t *out, int brute_force);
/* Emulates: reset_accumulation(); accumulate = true; frames × raytrace(view)
 * (frame_count 1 overwrites, later frames add — SPEC.md §13).
 * accum: width*height*4 floats, rgb = radiance sum, a = sample count (zero where not owned).
 * Returns the seed counter after the last frame. */
uint32_t orc_render(const orc_scene *s, const orc_render_params *p, float *accum, orc_counters *c);
/* per-pixel primary-ray dump for the ray-generation stage (frame 0 of the params) */
void orc_raygen(const orc_render_params *p, uint32_t x, uint32_t y, float origin[3], float dir[3]);
/* accum (sum,count) -> mean radiance rgba (a = count>0) and -> sRGB8 */
void orc_resolve(const float *accum, uint32_t n_pixels, float *mean_rgba);
void orc_tonemap(const float *accum, uint32_t n_pixels, uint8_t *rgba8);
void orc_srgb_thresholds(float out[256]);   /* SPEC §13.2: linear value at which sRGB code i starts (out[0] = 0) */

/* ---- denoiser path (BlitMode::DenoisedPathrace / Temporal; SPEC §15, reference render/asvgf.rs) */
typedef struct orc_denoiser orc_denoiser;
orc_denoiser *orc_denoiser_create(uint32_t w, uint32_t h);
void orc_denoiser_destroy(orc_denoiser *d);
/* one raytrace() call: mode 1 = DenoisedPathrace, 2 = Temporal; uses p->view, p->seed_counter (frames ignored) */
void orc_denoise_frame(orc_denoiser *d, const orc_scene *s, const orc_render_params *p, int mode, float *out_main);
void orc_denoiser_read(const orc_denoiser *d, uint32_t *gbuf_cur, float *motion, float *rad_cur, uint32_t *hist_cur);

/* ---- known-answer surfaces ------------------------------------------------- */
uint32_t orc_pcg_hash(uint32_t v);
/* n floats of the stream keyed by (pixel, stage_seed, tag) */
void orc_rng_stream(uint32_t pixel, uint32_t user_seed, uint32_t seed_counter, uint32_t tag, uint32_t n, float *out);
void orc_sincos2pi(float u, float *s, float *c);
float orc_atan2(float y, float x);
float orc_acos(float x);
void orc_woop(const float p0[3], const float p1[3], const float p2[3], float out12[12]);
int orc_ray_triangle(const float woop12[12], const float o[3], const float d[3], float tmin, float tmax,
                     float *t, float *u, float *v);
void orc_onb(const float n[3], float t[3], float b[3]);
/* BSDF at a surface point: returns f (rgb) and the mixture pdf for direction L */
void orc_bsdf_eval(const float base[3], float roughness, float metallic, const float N[3], const float Ng[3],
                   const float V[3], const float L[3], float f[3], float *pdf);
/* samples L from (r3,r4,r5); returns 0 when the path terminates */
int orc_bsdf_sample(const float base[3], float roughness, float metallic, const float N[3], const float Ng[3],
                    const float V[3], float r3, float r4, float r5, float L[3], float weight[3], float *pdf);
void orc_env_lookup(const orc_scene *s, const float d[3], float rgb[3]);
void orc_texture_lookup(const orc_scene *s, uint32_t image, float u, float v, int srgb, float rgba[4]);
float orc_srgb_lut(uint32_t i);

#endif
