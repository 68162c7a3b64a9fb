/*
 * lpt.h — C ABI of the MI355X-native path-tracing core that sits behind
 * Loupiote's `crates/lib` API (crate `loupiote-core`).
 *
 * The reference has no FFI layer: its public Rust API *is* the boundary
 * (reference crates/lib/src/lib.rs:1-11).  Every entry point below replaces one
 * public Rust item; the citation after "replaces:" names it (paths relative to
 * the reference root).  A Rust shim that keeps the `Renderer/Scene/SceneGPU/
 * ProbeGPU/Device/loaders` signatures and forwards to these symbols is shown in
 * INTEGRATION.md.
 *
 * Conventions
 *   - every function returns an `int` status (LPT_OK == 0); nothing unwinds
 *     across the boundary.  `lpt_last_error()` gives a thread-local message.
 *   - `create/upload` return a handle the caller destroys; input pointers are
 *     borrowed for the duration of the call only (the reference's
 *     `new_storage_with_data` / `write_texture` copy semantics,
 *     crates/lib/src/scene.rs:134-146, crates/lib/src/renderer.rs:646-660).
 *   - handles are not thread-safe; one caller thread per lpt_device (the
 *     reference drives everything from the winit thread,
 *     crates/standalone/src/app.rs:259-344).
 *   - matrices are 16 floats, column-major (glam::Mat4::to_cols_array()).
 *   - plain pointers and sizes only; no torch / HIP types in any signature.
 */
#ifndef LPT_H
#define LPT_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define LPT_ABI_VERSION 6u

/* replaces: albedo_rtx::uniforms::INVALID_INDEX (crates/lib/src/loaders/gltf.rs:120,124) */
#define LPT_INVALID_INDEX 0xFFFFFFFFu

/* ---- status codes ---------------------------------------------------------
 * replaces: enum Error { FileNotFound, TextureToBufferReadFail, AccelBuild }
 * (crates/lib/src/errors.rs:2-6) plus the conditions the reference panics on. */
enum {
    LPT_OK = 0,
    LPT_ERR_FILE_NOT_FOUND = 1, /* Error::FileNotFound(String)          */
    LPT_ERR_READBACK = 2,       /* Error::TextureToBufferReadFail       */
    LPT_ERR_ACCEL_BUILD = 3,    /* Error::AccelBuild(String)            */
    LPT_ERR_HIP = 4,            /* a HIP runtime call failed / no GPU   */
    LPT_ERR_RCCL = 5,           /* an RCCL call of the frame exchange failed */
    LPT_ERR_INVALID_ARG = 6
};

const char *lpt_last_error(void);
/* String form of a status, same text as `impl From<Error> for String`
 * (crates/lib/src/errors.rs:8-20) for the three reference variants. */
const char *lpt_status_string(int status);
uint32_t lpt_abi_version(void);

/* ---- plain-data structs ---------------------------------------------------- */

/* replaces: albedo_rtx::uniforms::Material — exactly the five fields the
 * reference writes (crates/lib/src/loaders/gltf.rs:113-126,
 * crates/lib/src/loaders/binary.rs:63-69).  32 bytes. */
typedef struct lpt_material {
    float color[4];          /* baseColorFactor                        */
    float roughness;         /* roughnessFactor                        */
    float reflectivity;      /* metallicFactor                         */
    uint32_t albedo_texture; /* image index or LPT_INVALID_INDEX       */
    uint32_t mra_texture;    /* image index or LPT_INVALID_INDEX       */
} lpt_material;

/* replaces: albedo_rtx::uniforms::Vertex { position:[f32;4], normal:[f32;4] }
 * (crates/lib/src/loaders/binary.rs:20-28).  uv rides in the two .w lanes. */
typedef struct lpt_vertex {
    float position[4]; /* x y z, u */
    float normal[4];   /* x y z, v */
} lpt_vertex;

/* replaces: albedo_rtx::uniforms::Light (crates/lib/src/scene.rs:33,50).
 * One-sided rectangular emitter.  64 bytes. */
typedef struct lpt_light {
    float normal[4];    /* xyz unit normal (emitting side), w unused        */
    float tangent[4];   /* xyz unit tangent,   w = half width               */
    float bitangent[4]; /* xyz unit bitangent, w = half height              */
    float origin[4];    /* xyz centre,         w = radiance (white)         */
} lpt_light;

/* replaces: albedo_rtx::uniforms::Instance as built by BLASArray::add_instance
 * (crates/lib/src/loaders/gltf.rs:141-145) and edited by Instance::set_transform
 * (crates/standalone/src/lib.rs:117-121). */
typedef struct lpt_instance {
    float model_to_world[16];
    uint32_t blas_index;
    uint32_t material_index;
    uint32_t pad[2];
} lpt_instance;

/* replaces: one BLASArray.entries element (crates/lib/src/scene.rs:43-49). */
typedef struct lpt_blas_entry {
    uint32_t vertex_offset; /* first vertex in Scene.vertices          */
    uint32_t vertex_count;
    uint32_t index_offset;  /* first index in the scene index pool     */
    uint32_t index_count;   /* 3 * triangle count                      */
} lpt_blas_entry;

typedef struct lpt_scene_counts {
    uint32_t materials, entries, vertices, indices, instances, lights, images;
} lpt_scene_counts;

typedef struct lpt_accel_stats {
    uint32_t triangles;  /* baked world-space triangles                 */
    uint32_t nodes;      /* wide-BVH nodes                              */
    uint32_t node_bytes; /* bytes per node                              */
    uint32_t tri_bytes;  /* bytes per pre-transformed triangle          */
    uint32_t max_depth;
    float build_ms;
    uint32_t host_baked_triangles; /* triangles transformed to world space on the HOST by the last upload (0 for
                                      LPT_ACCEL_BUILD_GPU_LBVH: instances are baked by k_bake_instance on the device) */
    float upload_ms;               /* wall time of the whole lpt_scene_upload(_ex) call: bake + build + copies */
    /* ABI 5: texel bytes this scene keeps in device memory — the tiled atlas (every image some material samples on its own, or none
     * references) + the paired (albedo, mra) texels of the materials whose two textures have one size; an image that only ever
     * appears as half of a pair is stored once, in the pair (the reference keeps each image once: scene.rs:172-184) */
    uint32_t texture_pairs;
    uint64_t texture_bytes_resident;
} lpt_accel_stats;

/* result of a closest-hit query (the build's `Intersection`, 16 bytes) */
typedef struct lpt_hit {
    float t, u, v;
    uint32_t prim; /* baked triangle id, 0x80000000|light for an emitter, LPT_INVALID_INDEX on miss */
} lpt_hit;

typedef struct lpt_ray_counts {
    uint64_t closest; /* closest-hit rays traced since the last reset */
    uint64_t shadow;  /* shadow (any-hit) rays traced                 */
    uint64_t shaded;  /* surface hits shaded                          */
    uint64_t nodes;   /* closest-hit kernel: BVH nodes visited (stats enabled only) */
    uint64_t tris;    /* closest-hit kernel: triangles tested  (stats enabled only) */
    uint64_t shadow_nodes; /* shadow kernel: nodes visited (stats enabled only)     */
    uint64_t shadow_tris;  /* shadow kernel: triangles tested (stats enabled only)  */
    /* closest-hit kernel, stats enabled only: wave utilisation of the persistent traversal loop */
    uint64_t wave_steps;   /* loop iterations summed over waves                        */
    uint64_t live_lanes;   /* lanes carrying a ray, summed over those iterations        */
    uint64_t node_lanes;   /* lanes that entered the node test                         */
    uint64_t tri_lanes;    /* lanes that ran a triangle test                           */
    /* ABI 4: bounce 0 is traced by packet traversal (one tree walk per 64 coherent rays): these rays are part of `closest`, their
     * node / triangle fetches are NOT part of `nodes` / `tris` */
    uint64_t primary;      /* closest-hit rays traced as packets                               */
    uint64_t packet_nodes; /* nodes entered, once per PACKET (stats enabled only)              */
    uint64_t packet_tris;  /* triangles fetched, once per PACKET (stats enabled only)          */
    /* ABI 5, stats enabled only, per-bounce launches only: the shadow rays that found an occluder, and the occluder-cache PROBE — what a
     * table of the last occluding triangle per cell of a grid over the shadow rays' origins (LPT_EXP_OCC_CELL_MILLI) would have answered
     * at each ray's start: entries found, and entries whose triangle occludes the ray (a hit would end that ray after one triangle test).
     * A measurement; no kernel uses such a cache. */
    uint64_t shadow_occluded;
    uint64_t occluder_cache_found;
    uint64_t occluder_cache_hits;
    /* ABI 6: rays of the per-bounce traversal launches that a WHOLE WAVE finished: handed over by the step budget (k_trace_coop re-traces them) or finished in
     * place at the tail of a launch (LPT_OPT_TAIL_LANES).  Which rays these are depends on scheduling; the frame does not.  Lets a test see that the path ran. */
    uint64_t wave_rays;
} lpt_ray_counts;

typedef struct lpt_timing {
    char label[32]; /* "ray generation", "intersection", "shading", "shadow", "accumulation", "asvgf", "exchange", "primary intersection", "path" */
    float ms;       /* summed over every raytrace() since enable_timings(1) */
    uint32_t launches;
} lpt_timing;

/* replaces: enum BlitMode (crates/lib/src/renderer.rs:160-167; spelling of
 * `Pahtrace` is the reference's). */
enum {
    LPT_BLIT_PATHTRACE = 0,
    LPT_BLIT_DENOISED = 1,
    LPT_BLIT_TEMPORAL = 2,
    LPT_BLIT_GBUFFER = 3,
    LPT_BLIT_MOTION = 4
};

typedef struct lpt_device lpt_device;
typedef struct lpt_scene lpt_scene;
typedef struct lpt_scene_gpu lpt_scene_gpu;
typedef struct lpt_probe lpt_probe;
typedef struct lpt_renderer lpt_renderer;
typedef struct lpt_comm lpt_comm;

/* ---- Device ---------------------------------------------------------------
 * replaces: Device::new(wgpu::Device) (crates/lib/src/device.rs:80).  One
 * process drives one GPU; `hip_ordinal` picks it.  Fails with LPT_ERR_HIP when
 * no gfx950 device is visible — there is no CPU fallback. */
int lpt_device_create(int hip_ordinal, lpt_device **out);
int lpt_device_destroy(lpt_device *dev);
int lpt_device_synchronize(lpt_device *dev);
/* name, CU count and the HIP stream handle (void*: a hipStream_t) the renderer
 * enqueues on — bench.py brackets that stream with HIP events. */
int lpt_device_info(lpt_device *dev, char *name, size_t name_cap, int *compute_units);
int lpt_device_stream(lpt_device *dev, void **hip_stream);

/* ---- Scene (CPU side) -----------------------------------------------------
 * replaces: Scene::default() and its pub fields (crates/lib/src/scene.rs:30-54).
 * As in the reference, a fresh scene holds ONE dummy element in every array
 * (material 0, BLAS entry 0, vertex 0, instance 0, light 0 = Light::new()), so
 * loaded content starts at index 1 (crates/standalone/src/lib.rs:117). */
int lpt_scene_create(lpt_scene **out);
int lpt_scene_destroy(lpt_scene *scene);
int lpt_scene_counts_get(const lpt_scene *scene, lpt_scene_counts *out);

/* replaces: BLASArray::add_bvh(MeshDescriptor) / add_bvh_indexed(IndexedMeshDescriptor)
 * (crates/lib/src/loaders/gltf.rs:91-105).  Strides are in BYTES (pas::Slice);
 * positions may be vec3 or vec4 (only xyz read); normals / uvs / indices may be
 * NULL.  Without indices, vertices are consumed three at a time.  Missing
 * normals become flat face normals (crates/lib/src/loaders/binary.rs:31-49).
 * LPT_ERR_ACCEL_BUILD on an out-of-range index or a count that is not a
 * multiple of three. */
int lpt_scene_add_mesh(lpt_scene *scene, const void *positions, size_t position_stride,
                       const void *normals, size_t normal_stride, const void *uvs,
                       size_t uv_stride, uint32_t vertex_count, const uint32_t *indices,
                       uint32_t index_count, uint32_t *out_blas_index);
/* replaces: BLASArray::add_instance(blas, model_to_world, material)
 * (crates/lib/src/loaders/gltf.rs:141-145). */
int lpt_scene_add_instance(lpt_scene *scene, uint32_t blas_index, const float model_to_world[16],
                           uint32_t material_index, uint32_t *out_instance_index);
/* replaces: Instance::set_transform (crates/standalone/src/lib.rs:118-121). */
int lpt_scene_set_instance_transform(lpt_scene *scene, uint32_t instance_index,
                                     const float model_to_world[16]);
/* replaces: scene.materials.push(..) (crates/lib/src/loaders/gltf.rs:113) */
int lpt_scene_add_material(lpt_scene *scene, const lpt_material *m, uint32_t *out_index);
/* replaces: scene.images.push(ImageData::new(rgba8,w,h)) (gltf.rs:150-153) */
int lpt_scene_add_image(lpt_scene *scene, const uint8_t *rgba8, uint32_t width, uint32_t height,
                        uint32_t *out_index);
/* replaces: pub lights: Vec<Light> (crates/lib/src/scene.rs:33) */
int lpt_scene_add_light(lpt_scene *scene, const lpt_light *l, uint32_t *out_index);
int lpt_scene_set_light(lpt_scene *scene, uint32_t index, const lpt_light *l);
/* replaces: Light::new() (crates/lib/src/scene.rs:50) */
int lpt_light_default(lpt_light *out);

/* Read-back of the flat arrays (the reference exposes them as pub Vec fields). */
int lpt_scene_get_materials(const lpt_scene *s, uint32_t first, uint32_t count, lpt_material *dst);
int lpt_scene_get_entries(const lpt_scene *s, uint32_t first, uint32_t count, lpt_blas_entry *dst);
int lpt_scene_get_vertices(const lpt_scene *s, uint32_t first, uint32_t count, lpt_vertex *dst);
int lpt_scene_get_indices(const lpt_scene *s, uint32_t first, uint32_t count, uint32_t *dst);
int lpt_scene_get_instances(const lpt_scene *s, uint32_t first, uint32_t count, lpt_instance *dst);
int lpt_scene_get_lights(const lpt_scene *s, uint32_t first, uint32_t count, lpt_light *dst);
int lpt_scene_get_image(const lpt_scene *s, uint32_t index, uint32_t *width, uint32_t *height,
                        uint8_t *dst_rgba8 /* may be NULL to query the size */);

/* ---- loaders --------------------------------------------------------------
 * replaces: loaders::load_gltf(&[u8], &mut Scene) -> Result<(), Error>
 * (crates/lib/src/loaders/gltf.rs:46-156).  Accepts .glb or .gltf JSON with
 * embedded (data: URI) buffers; appends to `scene` with the reference's offset
 * rules (:60,109-110).  LPT_ERR_FILE_NOT_FOUND on any parse failure (:49-53). */
int lpt_load_gltf(lpt_scene *scene, const uint8_t *data, size_t size);
/* replaces: loaders::load_gltf_path (gltf.rs:158-161) */
int lpt_load_gltf_path(lpt_scene *scene, const char *path);

/* replaces: the `image::ImageBuffer::save` half of ApplicationContext::save_screenshot
 * (crates/standalone/src/app.rs:172-187): writes RGBA8 rows (e.g. from lpt_renderer_read_pixels) as a PNG. */
int lpt_write_png(const char *path, const uint8_t *rgba8, uint32_t width, uint32_t height, size_t row_bytes);

/* replaces: the `image::codecs::hdr::HdrDecoder` step of ApplicationContext::load_env
 * (crates/standalone/src/app.rs:138-155): Radiance RGBE (.hdr) bytes -> the 4-byte RGBE pixels ProbeGPU::new takes,
 * rows top to bottom.  Call with rgbe8 == NULL to get the size, then with a buffer of width*height*4 bytes.
 * LPT_ERR_FILE_NOT_FOUND when the data is not a decodable "-Y H +X W" 32-bit_rle_rgbe image. */
int lpt_decode_hdr(const uint8_t *data, size_t size, uint8_t *rgbe8, size_t capacity, uint32_t *width, uint32_t *height);

/* replaces: the `image::io::Reader::open(path).decode()` step of ApplicationContext::load_blue_noise
 * (crates/standalone/src/app.rs:116-132): PNG or Huffman-coded JPEG bytes -> RGBA8 pixels, rows top to bottom (channels the
 * file lacks are 0, as in the glTF image path), e.g. for lpt_renderer_upload_noise.  Call with rgba8 == NULL to get the size. */
int lpt_decode_image(const uint8_t *data, size_t size, uint8_t *rgba8, size_t capacity, uint32_t *width, uint32_t *height);

/* new (the linear-radiance counterpart of lpt_write_png): RGBA float rows (e.g. lpt_renderer_read_radiance)
 * as a Radiance RGBE .hdr file (alpha dropped, negative / NaN -> 0).  row_floats = floats between row starts (>= 4*width). */
int lpt_write_hdr(const char *path, const float *rgba, uint32_t width, uint32_t height, size_t row_floats);

/* ---- SceneGPU / ProbeGPU --------------------------------------------------
 * replaces: SceneGPU::new_from_scene(&Scene,&Device,&Queue)
 * (crates/lib/src/scene.rs:151-188).  Bakes every instance into world space,
 * builds the wide BVH on the host and copies nodes / triangles / vertices /
 * materials / lights / image atlas into HBM.  The CPU scene stays with the
 * caller.  LPT_ERR_ACCEL_BUILD when the build fails. */
int lpt_scene_upload(lpt_device *dev, const lpt_scene *scene, lpt_scene_gpu **out);
/* new (the reference builds its BVH on the CPU at load time, loaders/gltf.rs:97-105): the same upload with
 * a choice of builder.  LPT_ACCEL_BUILD_HOST_SAH = binned-SAH + SAH-optimal 8-wide collapse on the host (what
 * lpt_scene_upload does); LPT_ACCEL_BUILD_GPU_LBVH = Morton-code radix tree collapsed to 8-wide nodes on the GPU
 * (a few milliseconds; lower tree quality) for scenes that are rebuilt every frame; with it the instances are also baked
 * on the device (object-space meshes + one transform per instance travel, nothing else is computed on the host).
 * Rendered results are identical. */
#define LPT_ACCEL_BUILD_HOST_SAH 0u
#define LPT_ACCEL_BUILD_GPU_LBVH 1u
/* or-ed into `flags`: do not interleave the (albedo, mra) textures of a material (experiments; the frame does not change by a bit) */
#define LPT_UPLOAD_NO_TEXTURE_PAIRS 0x100u
int lpt_scene_upload_ex(lpt_device *dev, const lpt_scene *scene, uint32_t flags, lpt_scene_gpu **out);
int lpt_scene_gpu_destroy(lpt_scene_gpu *sg);
/* replaces: Instance::set_transform + a new SceneGPU (crates/standalone/src/lib.rs:118-121, scene.rs:151):
 * after lpt_scene_set_instance_transform on the CPU scene, re-bakes only the changed instances and refits the
 * wide BVH on the GPU (topology kept).  The meshes and the instance list must be the uploaded ones
 * (LPT_ERR_INVALID_ARG otherwise: upload again).  Renderers bound to this lpt_scene_gpu keep working; call
 * lpt_renderer_reset_accumulation as after any scene change.  out_rebaked may be NULL. */
int lpt_scene_gpu_update_instances(lpt_scene_gpu *scene_gpu, const lpt_scene *scene, uint32_t *out_rebaked);
/* new: like lpt_scene_gpu_update_instances, but every instance is re-baked and the tree is REBUILT on the GPU
 * (Morton radix tree, as LPT_ACCEL_BUILD_GPU_LBVH) instead of refitted — for edits that move things far, where a refit
 * keeps a topology that no longer fits.  Renderers bound to this lpt_scene_gpu keep working. */
int lpt_scene_gpu_rebuild(lpt_scene_gpu *scene_gpu, const lpt_scene *scene);
int lpt_scene_gpu_stats(const lpt_scene_gpu *sg, lpt_accel_stats *out);

/* replaces: ProbeGPU::new(device, queue, data, width, height)
 * (crates/lib/src/scene.rs:72-121): 4 bytes/pixel RGBE8, equirectangular. */
int lpt_probe_upload(lpt_device *dev, const uint8_t *rgbe8, uint32_t width, uint32_t height,
                     lpt_probe **out);
int lpt_probe_destroy(lpt_probe *probe);

/* ---- ray queries (the IntersectorPass on its own) -------------------------
 * replaces: passes::IntersectorPass dispatch (crates/lib/src/renderer.rs:458-463)
 * applied to caller-supplied rays.  `origins`/`dirs` are n×3 floats on the
 * HOST; results come back to the host.  Used by the parity tests. */
int lpt_trace_closest(lpt_device *dev, const lpt_scene_gpu *sg, const float *origins,
                      const float *dirs, uint32_t n, lpt_hit *out_hits);
int lpt_trace_occluded(lpt_device *dev, const lpt_scene_gpu *sg, const float *origins,
                       const float *dirs, const float *tmax, uint32_t n, uint8_t *out_occluded);

/* ---- Renderer --------------------------------------------------------------
 * replaces: Renderer::new(&Device, (w,h), swapchain_format)
 * (crates/lib/src/renderer.rs:220-324).  Applies downsample_factor 0.5 like
 * the reference (:225); there is no swapchain. */
int lpt_renderer_create(lpt_device *dev, uint32_t width, uint32_t height, lpt_renderer **out);
int lpt_renderer_destroy(lpt_renderer *r);
/* replaces: pub downsample_factor (renderer.rs:203).  Takes effect at the next resize. */
int lpt_renderer_set_downsample(lpt_renderer *r, float factor);
/* replaces: Renderer::resize(&Device,&SceneGPU,Option<&ProbeGPU>,(w,h)) (renderer.rs:326-358) */
int lpt_renderer_resize(lpt_renderer *r, const lpt_scene_gpu *sg, const lpt_probe *probe_or_null,
                        uint32_t width, uint32_t height);
/* replaces: Renderer::get_size (renderer.rs:683) — the path-traced size */
int lpt_renderer_get_size(const lpt_renderer *r, uint32_t *width, uint32_t *height);
/* replaces: Renderer::max_ssbo_element_in_bytes (renderer.rs:209-218) */
uint32_t lpt_max_per_pixel_bytes(void);
/* replaces: Renderer::set_resources(&Device,&SceneGPU,Option<&ProbeGPU>)
 * (renderer.rs:687-725); resets frame_count to 1 (:724); a NULL probe is the
 * 1×1 default texture, i.e. a black environment (:693-696, device.rs:13-26). */
int lpt_renderer_set_resources(lpt_renderer *r, const lpt_scene_gpu *sg,
                               const lpt_probe *probe_or_null);
/* replaces: Renderer::raytrace(&mut encoder,&queue,&Mat4) (renderer.rs:392-549).
 * RECORDS one sample per pixel and returns: like the reference, which records the passes of a frame into a command
 * encoder that the app submits once (crates/standalone/src/app.rs:335-337), the launches happen at the next SUBMISSION
 * POINT — lpt_renderer_submit, every call that reads or waits (read_*, blit, exchange, synchronize, get_ray_counts,
 * get_timings, ...) and every setter whose value the passes read (resize, set_resources, set_blit_mode, set_shard, ...).
 * The protocol state of the reference (frame_count, seed, the frame_back toggle) moves at record time.  Consecutive
 * recorded calls with the same view that continue one accumulation are submitted as ONE wavefront of n samples per
 * pixel (bit for bit the result of n separate launches: lpt_renderer_raytrace_n's contract), which is what keeps the
 * persistent traversal waves fed when the caller issues its samples one call at a time.  `view_transform` is
 * camera-to-world (crates/standalone/src/camera.rs:101-108).  Returns LPT_OK and does nothing when resources are unset
 * (renderer.rs:403-407,419-422).  The denoising BlitModes (one temporal pass per call) submit at once.  An error of a
 * deferred launch (e.g. out of device memory) is reported by the call that submits it. */
int lpt_renderer_raytrace(lpt_renderer *r, const float view_transform[16]);
/* replaces: queue.submit(Some(encoder.finish())) (crates/standalone/src/app.rs:335-337): launches what has been recorded.
 * Asynchronous (nothing waits for the GPU); a no-op when nothing is pending. */
int lpt_renderer_submit(lpt_renderer *r);
/* new (no reference knob): how recorded calls become wavefronts.  0 = automatic: up to 64 fusable calls wait for the next
 * submission point, which launches them as wavefronts of about 4 M rays — a larger batch is cut SPATIALLY into runs of
 * whole tile rows (whole tiles on a sharded frame), every run with all the recorded samples; at 1920x1080 the 4 samples of a
 * frame leave as two wavefronts, the upper and the lower half of the image, which take the renderer's wavefront lanes in turn
 * and overlap.  n >= 1: n calls are ONE wavefront of n samples over the whole frame, launched when the n-th is recorded
 * (1 = every raytrace() launches immediately, the round-2 behaviour), up to 64.  Results never depend on the setting.
 * Ray-queue memory: 176 B per ray of a wavefront and wavefront lane. */
int lpt_renderer_set_max_fused(lpt_renderer *r, uint32_t n);
/* new: what record-then-submit has done so far — raytrace() calls recorded (with resources set), wavefronts launched for them,
 * and calls recorded but not yet submitted.  Pure host state: does not submit, does not wait. */
int lpt_renderer_get_submission_stats(const lpt_renderer *r, uint64_t *recorded_calls, uint64_t *wavefronts, uint32_t *pending_calls);
/* Build-only batching of the call above: exactly equivalent (bit for bit) to
 * n x { lpt_renderer_raytrace(r, view); accumulate = true (app.rs:318); } but traced as ONE wavefront
 * of n samples per pixel (sample-major queues), which keeps the persistent traversal waves fed.
 * Seeds advance by max_bounces per sample and samples are accumulated in call order.  1 <= n <= 64. */
int lpt_renderer_raytrace_n(lpt_renderer *r, const float view_transform[16], uint32_t n_samples);
/* replaces: Renderer::reset_accumulation (renderer.rs:609-618) */
int lpt_renderer_reset_accumulation(lpt_renderer *r);
/* replaces: pub accumulate (renderer.rs:204; set by the app at app.rs:318) */
int lpt_renderer_set_accumulate(lpt_renderer *r, int accumulate);
int lpt_renderer_get_accumulate(const lpt_renderer *r, int *accumulate);
/* replaces: global_uniforms.{frame_count,seed} (renderer.rs:286-290) — read-only view */
int lpt_renderer_get_frame_state(const lpt_renderer *r, uint32_t *frame_count, uint32_t *seed);
/* replaces: Renderer::upload_noise_texture (renderer.rs:620-664) */
int lpt_renderer_upload_noise(lpt_renderer *r, const uint8_t *rgba8, uint32_t width,
                              uint32_t height, uint32_t bytes_per_row);
/* replaces: Renderer::use_noise_texture(&queue,bool) (renderer.rs:666-673) */
int lpt_renderer_use_noise(lpt_renderer *r, int flag);
/* replaces: Renderer::set_blit_mode (renderer.rs:675-681) */
int lpt_renderer_set_blit_mode(lpt_renderer *r, int mode);
/* replaces: the ASVGF ping-pong targets the debug BlitModes expose (renderer.rs:557-589,
 * render/asvgf.rs:9-152): current G-buffer (w*h*4 u32: prim id, depth bits, oct normal, RGBA8 albedo),
 * motion (w*h*2 f32, uv units), temporally accumulated radiance+variance (w*h*4 f32), history (w*h u32).
 * Any pointer may be NULL.  Blocking.  Parity surface for the denoiser path. */
int lpt_renderer_read_denoiser(lpt_renderer *r, uint32_t *gbuffer, float *motion, float *radiance, uint32_t *history);
/* replaces: Renderer::blit(&Device,&mut encoder,&TextureView) (renderer.rs:551-607):
 * tonemapped sRGB RGBA8 of the current target into caller memory.  Like lpt_renderer_read_radiance, a frame that is still
 * recorded is submitted together with its read-back: every wavefront's pixel rows are tonemapped and copied as soon as
 * that wavefront has been accumulated (single-GPU Pathtrace frames; also lpt_renderer_read_pixels). */
int lpt_renderer_blit_rgba8(lpt_renderer *r, uint8_t *dst, size_t row_bytes);
/* replaces: async Renderer::read_pixels -> Result<Vec<u8>,Error> (renderer.rs:727-811).
 * Blocking (the reference's device.poll(Wait), :791); w*h*4 bytes, tight rows. */
int lpt_renderer_read_pixels(lpt_renderer *r, uint8_t *dst);
/* Parity surface (no reference twin): mean radiance, w*h*4 floats, a = 1 where
 * this process owns the pixel and has accumulated at least one sample.  Blocking.  `dst` may be pageable memory (the
 * HIP runtime stages the copy: 27 GB/s measured) or memory from lpt_host_alloc (one DMA at link speed).  When the frame is
 * still recorded (the usual case: raytrace() x spp, then this call) the read-back is part of the submission: the pixel rows
 * of every wavefront are resolved and copied as soon as that wavefront has been accumulated, under the wavefronts that are
 * still tracing the other rows (single-GPU Pathtrace frames; 0.39 instead of 0.62 ms exposed at 1920x1080). */
int lpt_renderer_read_radiance(lpt_renderer *r, float *dst);
/* new: page-locked host memory for read-back destinations (the wgpu staging buffer the reference maps in read_pixels,
 * renderer.rs:772-800, is such memory too).  Any lpt_device must exist first.  Free with lpt_host_free. */
int lpt_host_alloc(size_t bytes, void **out);
int lpt_host_free(void *ptr);
/* new: page-lock and map caller-owned host memory for the device (hipHostRegister) — e.g. a POSIX shared-memory segment that every rank
 * of a node maps, as the destination of lpt_renderer_read_radiance_owned.  Any lpt_device must exist first. */
int lpt_host_register(void *ptr, size_t bytes);
int lpt_host_unregister(void *ptr);
/* new (multi-GPU, HOST-SIDE GATHER; no reference counterpart): this rank's OWNED pixels of the mean radiance written straight into
 * `frame_dst`, a whole-frame w*h*4 float destination in page-locked host memory (lpt_host_alloc / lpt_host_register); the other ranks'
 * pixels are not touched.  When the consumer of the frame is the host (the SURVEY 8d span ends in read_radiance) the ranks of a node can
 * all write into ONE shared-memory frame — each GPU pushes its 1/N over its own PCIe link, nothing is gathered on rank 0's GPU first,
 * and a host-side barrier completes the frame: the alternative to lpt_renderer_exchange + lpt_renderer_read_radiance on rank 0
 * (DESIGN 6).  BlitMode::Pathtrace only.  Blocking. */
int lpt_renderer_read_radiance_owned(lpt_renderer *r, float *frame_dst);
/* new (multi-GPU, HOST-SIDE GATHER behind the boundary; replaces, on N ranks, the one blocking call the reference ends a frame with:
 * async Renderer::read_pixels, renderer.rs:727-811 — copy to a staging buffer, map it, device.poll(Wait)).  ONE whole frame (w*h*4 floats)
 * in POSIX shared memory that every rank of a node maps and page-locks; every rank writes its OWN pixels into it straight from its GPU
 * (lpt_renderer_read_radiance_owned(r, frame)) and lpt_host_frame_barrier completes the frame: a line of per-rank progress words in the same
 * segment, pause-spinning for the first microseconds, then futex sleeps.  Rank 0 creates (`name` as for shm_open: "/something"; fails with
 * LPT_ERR_FILE_NOT_FOUND if it exists), the other ranks attach once rank 0 has (the name travels out of band, like the communicator id);
 * destroy unmaps, the creator also unlinks.  A device must exist first (the segment is registered with HIP) unless LPT_HOST_FRAME_HOST_ONLY is
 * passed: a participant that only reads the finished frame (a compositor or encoder process) needs no GPU.  `world` = the barrier's participants. */
typedef struct lpt_host_frame lpt_host_frame;
enum { LPT_HOST_FRAME_HOST_ONLY = 1 };
int lpt_host_frame_create(const char *name, uint32_t width, uint32_t height, uint32_t world, uint32_t flags, lpt_host_frame **out);
int lpt_host_frame_attach(const char *name, uint32_t width, uint32_t height, uint32_t world, uint32_t flags, lpt_host_frame **out);
/* the frame: width * height * 4 floats, row-major, in page-locked shared memory (valid until lpt_host_frame_destroy) */
int lpt_host_frame_ptr(lpt_host_frame *f, float **frame);
/* returns when every one of the `world` participants has called it with this frame number (1, 2, 3, ...: the caller's frame counter, the same
 * on every rank; a rank calls it after its lpt_renderer_read_radiance_owned of that frame has returned).  LPT_ERR_READBACK after timeout_ms.
 * ARRIVE-ONLY over ONE frame: once it returns, frame `frame_no` is complete — and any rank may start writing frame_no + 1 into the same segment.  A consumer that
 * still reads frame_no (rank 0 encoding it, a LPT_HOST_FRAME_HOST_ONLY participant) must hold the writers back itself: a SECOND barrier call (another frame number,
 * e.g. 2 k for "written" and 2 k + 1 for "consumed": bench.py and examples/multi_gpu.c do this) before anybody's next lpt_renderer_read_radiance_owned, or a copy. */
int lpt_host_frame_barrier(lpt_host_frame *f, uint32_t rank, uint32_t frame_no, uint32_t timeout_ms);
int lpt_host_frame_destroy(lpt_host_frame *f);
/* replaces: renderer.queries.values()/labels()
 * (crates/standalone/src/gui/windows/performance_info.rs:19-20) */
int lpt_renderer_get_timings(lpt_renderer *r, lpt_timing *out, int *inout_count);
int lpt_renderer_enable_timings(lpt_renderer *r, int flag);

/* ---- build-only extensions (no reference knob) -------- */
/* reference constant STATIC/MOVING_NUM_BOUNCES = 3 (renderer.rs:398-399) */
int lpt_renderer_set_max_bounces(lpt_renderer *r, uint32_t bounces);
int lpt_renderer_set_seed(lpt_renderer *r, uint32_t user_seed);
/* vertical field of view in radians (reference: Camera::default inside albedo) */
int lpt_renderer_set_vfov(lpt_renderer *r, float radians);
/* Pixel-tile sharding for one-process-per-GPU rendering: this process traces
 * the tiles whose index (row-major over ceil(w/tile_w) × ceil(h/tile_h)) is
 * ≡ rank (mod world_size).  Default (0,1,32,8) = everything.  The tile area must be a multiple of 64 (one wave). */
int lpt_renderer_set_shard(lpt_renderer *r, uint32_t rank, uint32_t world_size, uint32_t tile_w,
                           uint32_t tile_h);
/* new: the same with WEIGHTED ownership.  Tile t belongs to virtual rank t % V, V = the sum of the weights, and the virtual
 * ranks are dealt to the ranks in proportion to `weights[rank]` (world_size entries, each 0..8, sum 1..64, at most 32 ranks;
 * NULL = every rank 1 = the rule above).  A rank that also assembles, reads back or filters the frame (rank 0 of an exchange)
 * is given a smaller weight so that its frame takes as long as the others': e.g. {5,8,8,8,8,8,8,8}.  A weight of 0 makes a
 * rank a pure compositor.  Every rank must pass the same weights.  Results do not depend on the weights (the RNG is keyed by
 * the global pixel): tests/test_gpu_exchange.py. */
int lpt_renderer_set_shard_weighted(lpt_renderer *r, uint32_t rank, uint32_t world_size, uint32_t tile_w, uint32_t tile_h,
                                    const uint32_t *weights_or_null);
/* new (multi-GPU denoising; reference: asvgf passes run on the one GPU, renderer.rs:513-522).  With set_shard(world > 1)
 * and a denoising BlitMode, raytrace() fills only this rank's tiles of the filter inputs.  `lpt_renderer_denoiser_inputs`
 * returns the device pointers of the full-frame input buffers of the CURRENT frame (noisy radiance float4, G-buffer
 * uint4, motion float2; n_pixels each; zero outside the rank's tiles) so that the host can sum them onto rank 0 (RCCL
 * reduce; integer sum for the G-buffer).  `lpt_renderer_denoise_filter` then runs temporal accumulation, the a-trous
 * iterations and the composite over the whole frame on that rank (no-op for world == 1: raytrace() already did). */
int lpt_renderer_denoiser_inputs(lpt_renderer *r, void **noisy, void **gbuffer, void **motion, size_t *n_pixels);
int lpt_renderer_denoise_filter(lpt_renderer *r);
/* Device address + byte size of the fp32 RGBA accumulation buffer (rgb = radiance SUM, a = sample count; zero where
 * not owned) for hosts that run their own collective.  Such a host must combine into a buffer of its own: this one is
 * rewritten only on owned pixels, so an in-place reduce would count rank 0's foreign pixels again on the next frame
 * (lpt_renderer_exchange avoids that with a separate presented frame). */
int lpt_renderer_radiance_device_ptr(lpt_renderer *r, void **device_ptr, size_t *bytes);
/* ---- multi-GPU frame exchange (new functionality: the reference is single-GPU; its Device::new is
 * crates/lib/src/device.rs:79 and the only caller of Renderer::raytrace / read_pixels is the one-threaded event loop,
 * crates/standalone/src/app.rs:297-318).  North star: frames shard by pixel tile across the GPUs of a node and the
 * accumulated radiance is combined on rank 0 with RCCL over xGMI.  Everything below is plain RCCL inside the library:
 * a Rust (or any other) host needs no torch.
 *
 *   one process per GPU                          | one process, several GPUs
 *   rank 0: lpt_comm_unique_id(id); ship the     | lpt_comm_unique_id(id); lpt_comm_group_begin();
 *   128 bytes to the other ranks out of band     | for i in 0..n: lpt_comm_create(dev[i], id, i, n, &comm[i]);
 *   every rank: lpt_comm_create(dev, id, rank,   | lpt_comm_group_end();   (and the per-frame exchanges of the n
 *   world, &comm)                                | renderers inside a begin/end pair as well)
 *
 * Then per renderer: lpt_renderer_set_comm(r, comm) (= set_shard(rank, world, 32, 8) + the binding), and per frame
 * raytrace(...) [x spp]; lpt_renderer_exchange(r, mode); on rank 0 read_radiance / read_pixels / blit return the whole
 * frame.  The exchange is enqueued on the renderer's stream behind the frame's kernels (no host synchronisation), so
 * with several renderers the exchange of frame k overlaps the kernels of frame k+1. */
#define LPT_COMM_ID_BYTES 128
/* replaces: nothing (ncclGetUniqueId).  Call once, on one rank. */
int lpt_comm_unique_id(void *out_id /* LPT_COMM_ID_BYTES */);
/* ncclCommInitRank on `dev`.  Collective over the `world_size` callers; LPT_ERR_RCCL on failure. */
int lpt_comm_create(lpt_device *dev, const void *id /* LPT_COMM_ID_BYTES */, int rank, int world_size, lpt_comm **out);
/* Unbind every renderer first (lpt_renderer_set_comm(r, NULL) or destroy it). */
int lpt_comm_destroy(lpt_comm *comm);
/* rank and size as RCCL reports them for the communicator (ncclCommUserRank / ncclCommCount) */
int lpt_comm_info(const lpt_comm *comm, int *rank, int *world_size);
/* ncclGroupStart / ncclGroupEnd: needed only when ONE thread drives several communicators (right column above).
 * lpt_renderer_exchange inside such a bracket enqueues its pack + send / recv / reduce only; RCCL issues them at the
 * outermost end, and lpt_comm_group_end then enqueues what consumes the received data (unpack, filter passes). */
int lpt_comm_group_begin(void);
int lpt_comm_group_end(void);

enum {
    /* every rank sends only the pixels it owns (W*H/N * 16 B; 4 MB per rank at 1080p, N = 8) straight to rank 0 —
     * grouped ncclSend / ncclRecv, seven xGMI links in parallel — which scatters them into the frame.  Default. */
    LPT_EXCHANGE_GATHER_TILES = 0,
    /* ncclReduce(sum, root 0) of the full-frame accumulation buffers (zero outside a rank's tiles): the literal form the
     * north star names; 33 MB around a ring at 1080p.  Bit-identical result (x + 0 = x). */
    LPT_EXCHANGE_REDUCE = 1
};
/* Binds a communicator (NULL unbinds: single-GPU again).  Implies lpt_renderer_set_shard(rank, world, 32, 8) with the
 * communicator's rank / size and therefore resets the accumulation. */
int lpt_renderer_set_comm(lpt_renderer *r, lpt_comm *comm_or_null);
/* lpt_renderer_set_comm with weighted tile ownership (lpt_renderer_set_shard_weighted; `weights` has one entry per rank of the
 * communicator, the same on every rank). */
int lpt_renderer_set_comm_weighted(lpt_renderer *r, lpt_comm *comm_or_null, const uint32_t *weights_or_null);
/* Combines the ranks' accumulation buffers into rank 0's PRESENTED frame (a separate full-frame buffer: every rank's own
 * accumulation buffer stays owned-pixels-only, so progressive frames can be exchanged again and again).  Collective over
 * the communicator; asynchronous; a no-op without a communicator.  In the denoising BlitModes it exchanges the filter
 * inputs instead (lpt_renderer_denoiser_inputs: noisy radiance, G-buffer, motion — owned pixels only, 40 B each, or three
 * ncclReduce in LPT_EXCHANGE_REDUCE mode) and runs the temporal / a-trous / composite passes on rank 0. */
int lpt_renderer_exchange(lpt_renderer *r, int mode);
/* The same exchange among renderers of ONE process without RCCL (peer copies): `root` presents the frame assembled from
 * its own tiles and those of `peers` (n_peers renderers sharded with the same world size and tile shape, on the same or
 * on other devices of the process).  In the denoising BlitModes the filter inputs travel and `root` filters the frame.
 * Used by single-process hosts and by the tests that emulate N ranks on one GPU. */
int lpt_renderer_exchange_local(lpt_renderer *root, lpt_renderer *const *peers, int n_peers);

/* new (no reference knob): the number of wavefront lanes of a renderer (1..4, default 2).  Consecutive SUBMISSIONS (the fused
 * batches of recorded raytrace() calls, or lpt_renderer_raytrace_n calls) take the lanes in turn, each on its own HIP stream
 * with its own ray queues, so the shading of one wavefront overlaps the traversal of the next; accumulation stays in call
 * order on the renderer's stream, so results do not change by a bit.  Costs one set of ray buffers per lane in use. */
int lpt_renderer_set_lanes(lpt_renderer *r, int lanes);
/* new (north star: "a sorted shade / next-event stage using wavefront ballot / prefix-sum"; the reference dispatches the
 * full pixel grid unsorted, renderer.rs:484-509).  flag != 0: the shading pass writes the next-bounce and the shadow-ray
 * queue ordered by direction octant inside every 256-ray block (ballot + popcount per key, LDS prefix sum, still one
 * atomic per block), so a traversal wave's 64 rays share one or two octants.  Results are keyed by pixel slot and do
 * not change by a bit; only traversal coherence does.  flag bits: 1 = next-bounce queue, 2 = shadow queue; any other non-zero
 * value = both queues.  Default: off (each measured slower or neutral on the bench scene). */
int lpt_renderer_set_sort_queues(lpt_renderer *r, int flag);
/* new (the reference has three settings, crates/standalone/src/settings.rs:3-17, none of them about launches): what a host may sanely
 * set about HOW a frame is launched.  EVERY value gives the same frame bit for bit — only which kernels run, and the size of their
 * grids, changes.  The library reads no environment variable for any of this (a host's environment must not change which kernels run). */
typedef enum lpt_option {
    LPT_OPT_PACKET_PRIMARY = 1,     /* bounce 0 by packet traversal, one tree walk per 8x8-pixel patch: 2 (default) = where the patch is narrow (up to 1.8 mrad
                                     * per pixel: 1080p, 4K, 1024^2 at 45 degrees; not a 270p preview), 1 = always, 0 = never (per ray) */
    LPT_OPT_WAVEFRONT_RAYS = 2,     /* rays per wavefront an automatic submission aims at (default 4 194 304) */
    LPT_OPT_PATH_RAYS = 3,          /* wavefronts of at most this many rays run every bounce behind the primary hits in ONE launch (the
                                     * path kernel: no chip-wide barrier per bounce — small frames and the tile shards of a wide multi-GPU
                                     * frame); larger ones take the per-bounce launches of renderer.rs:484-509.  Default 120 000, the
                                     * measured cross-over; 0: never */
    LPT_OPT_COOP_RAYS = 4,          /* wavefronts of at most this many rays — fewer than the chip has wave slots — take the per-bounce launches with EVERY ray
                                     * traced by a whole wave (eight lanes per node): the frame of a tiny wavefront is a chain of dependent traversal steps,
                                     * and a wave per ray shortens it several times over (64x36 pixels, 4 spp: 0.72 -> 0.39 ms per frame).  Default 32 000, the measured cross-over;
                                     * 0 = never.  Not used while lpt_renderer_enable_stats is on */
    LPT_OPT_TAIL_LANES = 5,         /* the tails of the per-bounce traversal launches finished IN PLACE: a wave whose queues are dry and that is down to this
                                     * many live rays (1..8) finishes them cooperatively, all 64 lanes per ray, from where each stands — nothing is
                                     * dropped, restarted at the root or launched behind.  Default 4; 0 = off (then LPT_EXP_STEP_BUDGET applies).  For submissions that
                                     * leave as ONE wavefront; not used while lpt_renderer_enable_stats is on */
    LPT_OPT_EXPERIMENT_BASE = 256   /* LPT_OPT_EXPERIMENT(name): the tuning knobs of the A/B tools and the variant tests — NOT a stable surface */
} lpt_option;
#define LPT_OPT_EXPERIMENT(name) (LPT_OPT_EXPERIMENT_BASE + (name))
/* The experiment knobs (tools/dev, bench.py --opt, tests that compare kernel variants).  Numbers and meanings may change with any build. */
typedef enum lpt_experiment {
    LPT_EXP_PIPE_RAYS = 0,          /* wavefronts of at most this many rays trace with the one-round-trip step (default: all); 0: never */
    LPT_EXP_REFILL = 1,             /* per-bounce traversal: lanes live below which a wave refills (default 44) */
    LPT_EXP_TRACE_WAVES_PER_CU = 2, /* per-bounce traversal: persistent waves per CU; 0 (default): sized from the ray count */
    LPT_EXP_SHADE_BLOCKS_PER_CU = 3,/* shading pass: blocks per CU; 0 (default) = 4 for a wavefront alone on the chip, 3 for the pieces of a cut batch */
    LPT_EXP_PATH_WAVES_PER_CU = 4,  /* path kernel: persistent waves per CU (default 16) */
    LPT_EXP_PATH_REFILL = 5,        /* path kernel: lanes tracing below which a batch of lanes is shaded / restarted (default 32) */
    LPT_EXP_OCC_CELL_MILLI = 6,     /* stats only: grid cell of the occluder-cache probe in 1/1000 scene units (default 250; 0: probe off) */
    LPT_EXP_STEP_BUDGET = 7,        /* per-bounce traversal launches with LPT_OPT_TAIL_LANES 0: a ray not finished after this many steps is dropped by the per-lane
                                     * kernel and traced again by a whole wave (k_trace_coop); default 48, 0 = off.  Not applied while lpt_renderer_enable_stats is on */
    LPT_EXP_BUDGET_RAYS = 8,        /* the tail in place / the step budget apply to submissions that leave as ONE wavefront of at most this many rays (default 3 000 000) */
    LPT_EXP_PACKET_QUADS = 9,       /* packet traversal of bounce 0: 1 (default) = a packet is the four samples of a 4x4-pixel quarter where the frame allows it
                                     * (dense tiles, a multiple of four samples); 0 = always one sample of an 8x8-pixel patch */
    LPT_EXP_SPLIT_RAYS = 10,        /* a batch above this many rays that would still fit one wavefront leaves as two, on the renderer's lanes (default
                                     * 3 000 000; 0: never split below LPT_OPT_WAVEFRONT_RAYS) */
    LPT_EXP_BUDGET_SPLIT = 11       /* 1: the tail in place / the step budget also apply to the pieces of a cut batch (default 0: only to submissions that leave
                                     * as one wavefront — LPT_EXP_BUDGET_RAYS alone does not turn them on for the pieces) */
} lpt_experiment;
int lpt_renderer_set_option(lpt_renderer *r, int option, uint64_t value);
int lpt_renderer_get_option(const lpt_renderer *r, int option, uint64_t *value);
/* The shard layout lpt_renderer_set_shard / lpt_renderer_exchange use, for hosts that run their own exchange (pure host
 * arithmetic, no GPU needed): the number of pixel slots rank `rank` owns (whole tiles, including the part of edge tiles
 * outside the image) and where its slots start in the concatenation of all ranks' slot arrays on rank 0. */
int lpt_shard_layout(uint32_t width, uint32_t height, uint32_t tile_w, uint32_t tile_h, uint32_t world_size, uint32_t rank,
                     uint32_t *out_slots, uint32_t *out_slot_offset);
int lpt_shard_layout_weighted(uint32_t width, uint32_t height, uint32_t tile_w, uint32_t tile_h, uint32_t world_size, uint32_t rank,
                              const uint32_t *weights_or_null, uint32_t *out_slots, uint32_t *out_slot_offset);
/* the rank that owns tile `tile` (row-major over the tile grid) under the ownership rule (weights NULL = tile % world_size) */
int lpt_shard_owner(uint32_t world_size, const uint32_t *weights_or_null, uint32_t tile, uint32_t *out_rank);
int lpt_renderer_get_ray_counts(lpt_renderer *r, lpt_ray_counts *out);
/* per-bounce queue sizes of the LAST traced frame: closest[b] = closest-hit rays of bounce b, shadow[b] = shadow rays
 * emitted by bounce b; up to `cap` entries each (either pointer may be NULL).  Blocking. */
int lpt_renderer_get_queue_counts(lpt_renderer *r, uint32_t *closest, uint32_t *shadow, uint32_t cap);
int lpt_renderer_reset_ray_counts(lpt_renderer *r);
/* new, stats enabled only: traversal steps per ray over the per-bounce traversal launches of the LAST wavefront — the maximum (the longest
 * ray of a launch sets its duration, docs/ROUNDS.md §5.5) and a histogram by power of two (hist12[k]: 2^k <= steps < 2^(k+1)).  Blocking.
 * The stats kernels run WITHOUT the step budget (LPT_EXP_STEP_BUDGET): every ray is traced to its end by the per-lane kernel, so the histogram, the
 * maximum and the nodes / triangles per ray are those of complete traversals, whatever the option says. */
int lpt_renderer_get_step_histogram(lpt_renderer *r, uint32_t *max_steps, uint32_t *hist12);
/* count BVH nodes visited / triangles tested per ray (slower kernel variant) */
int lpt_renderer_enable_stats(lpt_renderer *r, int flag);
int lpt_renderer_synchronize(lpt_renderer *r);
/* The HIP stream (void*: a hipStream_t) this renderer enqueues on.  Each renderer owns one, so two
 * renderers on one device pipeline consecutive frames; the collective layer orders its reduce on it. */
int lpt_renderer_stream(lpt_renderer *r, void **hip_stream);

#ifdef __cplusplus
}
#endif
#endif /* LPT_H */
