// loupiote.hpp — header-only C++ mirror of the reference's `loupiote-core` API over the C ABI
// (include/lpt.h).  Same type and method names as reference crates/lib/src/{device,scene,renderer,
// errors}.rs and loaders/gltf.rs; wgpu arguments are dropped.  Errors that the reference returns as
// Result<_, Error> are thrown as loupiote::Error (kind mirrors errors.rs:2-6).
#pragma once
#include <array>
#include <stdexcept>
#include <string>
#include <vector>

#include "lpt.h"

namespace loupiote {

struct Error : std::runtime_error {
    enum Kind { FileNotFound = 1, TextureToBufferReadFail = 2, AccelBuild = 3, Hip = 4, Rccl = 5, InvalidArg = 6 } kind;
    Error(int status, const char *msg) : std::runtime_error(msg), kind(static_cast<Kind>(status)) {}
};
inline void check(int status) {
    if (status != LPT_OK) throw Error(status, lpt_last_error());
}

enum class BlitMode { Pahtrace = 0, DenoisedPathrace = 1, Temporal = 2, GBuffer = 3, MotionVector = 4 };  // renderer.rs:160-167
using Mat4 = std::array<float, 16>;  // column-major (glam::Mat4::to_cols_array)

class Device {  // device.rs:72-141
   public:
    explicit Device(int hip_ordinal = 0) { check(lpt_device_create(hip_ordinal, &h_)); }
    ~Device() { lpt_device_destroy(h_); }
    Device(const Device &) = delete;
    Device &operator=(const Device &) = delete;
    lpt_device *inner() const { return h_; }
    void synchronize() { check(lpt_device_synchronize(h_)); }

   private:
    lpt_device *h_ = nullptr;
};

// one rank of the multi-GPU frame exchange (new functionality; plain RCCL inside the library, include/lpt.h)
class Comm {
   public:
    using Id = std::array<unsigned char, LPT_COMM_ID_BYTES>;
    static Id unique_id() { Id id{}; check(lpt_comm_unique_id(id.data())); return id; }   // on ONE rank; ship the bytes to the others
    Comm(const Device &dev, const Id &id, int rank, int world) { check(lpt_comm_create(dev.inner(), id.data(), rank, world, &h_)); }
    ~Comm() { lpt_comm_destroy(h_); }
    Comm(const Comm &) = delete;
    Comm &operator=(const Comm &) = delete;
    lpt_comm *handle() const { return h_; }

   private:
    lpt_comm *h_ = nullptr;
};

class Scene {  // scene.rs:30-54
   public:
    Scene() { check(lpt_scene_create(&h_)); }
    ~Scene() { lpt_scene_destroy(h_); }
    Scene(const Scene &) = delete;
    Scene &operator=(const Scene &) = delete;
    lpt_scene *handle() const { return h_; }
    // BLASArray::add_bvh / add_bvh_indexed (gltf.rs:97-105); strides in bytes
    uint32_t add_bvh(const void *positions, size_t pstride, const void *normals, size_t nstride, const void *uvs, size_t ustride,
                     uint32_t vertex_count, const uint32_t *indices = nullptr, uint32_t index_count = 0) {
        uint32_t id = 0;
        check(lpt_scene_add_mesh(h_, positions, pstride, normals, nstride, uvs, ustride, vertex_count, indices, index_count, &id));
        return id;
    }
    uint32_t add_instance(uint32_t blas, const Mat4 &model_to_world, uint32_t material) {  // gltf.rs:141-145
        uint32_t id = 0;
        check(lpt_scene_add_instance(h_, blas, model_to_world.data(), material, &id));
        return id;
    }
    void set_instance_transform(uint32_t i, const Mat4 &m) { check(lpt_scene_set_instance_transform(h_, i, m.data())); }
    uint32_t add_material(const lpt_material &m) { uint32_t id = 0; check(lpt_scene_add_material(h_, &m, &id)); return id; }
    uint32_t add_image(const uint8_t *rgba8, uint32_t w, uint32_t h) { uint32_t id = 0; check(lpt_scene_add_image(h_, rgba8, w, h, &id)); return id; }
    uint32_t add_light(const lpt_light &l) { uint32_t id = 0; check(lpt_scene_add_light(h_, &l, &id)); return id; }
    void set_light(uint32_t i, const lpt_light &l) { check(lpt_scene_set_light(h_, i, &l)); }
    lpt_scene_counts counts() const { lpt_scene_counts c; check(lpt_scene_counts_get(h_, &c)); return c; }

   private:
    lpt_scene *h_ = nullptr;
};

namespace loaders {  // loaders/gltf.rs:46-161
inline void load_gltf(const uint8_t *data, size_t size, Scene &scene) { check(lpt_load_gltf(scene.handle(), data, size)); }
inline void load_gltf_path(const std::string &path, Scene &scene) { check(lpt_load_gltf_path(scene.handle(), path.c_str())); }
/// the decoding half of ApplicationContext::load_env (app.rs:138-155): Radiance .hdr bytes -> RGBE8 pixels for ProbeGPU
inline std::vector<uint8_t> load_env(const uint8_t *data, size_t size, uint32_t &width, uint32_t &height) {
    check(lpt_decode_hdr(data, size, nullptr, 0, &width, &height));
    std::vector<uint8_t> px((size_t)width * height * 4);
    check(lpt_decode_hdr(data, size, px.data(), px.size(), &width, &height));
    return px;
}
}  // namespace loaders

class SceneGPU {  // scene.rs:56-64,151-188
   public:
    static SceneGPU new_from_scene(const Scene &scene, const Device &device, bool gpu_build = false) {
        SceneGPU s;
        check(lpt_scene_upload_ex(device.inner(), scene.handle(), gpu_build ? LPT_ACCEL_BUILD_GPU_LBVH : LPT_ACCEL_BUILD_HOST_SAH, &s.h_));
        return s;
    }
    SceneGPU(SceneGPU &&o) noexcept : h_(o.h_) { o.h_ = nullptr; }
    ~SceneGPU() { lpt_scene_gpu_destroy(h_); }
    lpt_scene_gpu *handle() const { return h_; }
    lpt_accel_stats stats() const { lpt_accel_stats s; check(lpt_scene_gpu_stats(h_, &s)); return s; }
    /// re-bake the instances whose transform changed and refit the BVH on the GPU (standalone/src/lib.rs:118-121)
    /// re-bake every instance and rebuild the BVH on the GPU (for edits too large for a refit)
    void rebuild(const Scene &scene) { check(lpt_scene_gpu_rebuild(h_, scene.handle())); }
    uint32_t update_instances(const Scene &scene) { uint32_t n = 0; check(lpt_scene_gpu_update_instances(h_, scene.handle(), &n)); return n; }

   private:
    SceneGPU() = default;
    lpt_scene_gpu *h_ = nullptr;
};

class ProbeGPU {  // scene.rs:66-121
   public:
    ProbeGPU(const Device &device, const uint8_t *rgbe8, uint32_t width, uint32_t height) { check(lpt_probe_upload(device.inner(), rgbe8, width, height, &h_)); }
    ~ProbeGPU() { lpt_probe_destroy(h_); }
    ProbeGPU(const ProbeGPU &) = delete;
    lpt_probe *handle() const { return h_; }

   private:
    lpt_probe *h_ = nullptr;
};

class Renderer {  // renderer.rs:169-811
   public:
    float downsample_factor = 0.5f;  // pub field, renderer.rs:203
    bool accumulate = false;         // pub field, renderer.rs:204

    Renderer(const Device &device, uint32_t width, uint32_t height) { check(lpt_renderer_create(device.inner(), width, height, &h_)); }
    ~Renderer() { lpt_renderer_destroy(h_); }
    Renderer(const Renderer &) = delete;
    static uint32_t max_ssbo_element_in_bytes() { return lpt_max_per_pixel_bytes(); }
    void resize(const SceneGPU &scene, const ProbeGPU *probe, uint32_t width, uint32_t height) {
        check(lpt_renderer_set_downsample(h_, downsample_factor));
        check(lpt_renderer_resize(h_, scene.handle(), probe ? probe->handle() : nullptr, width, height));
    }
    void set_resources(const SceneGPU &scene, const ProbeGPU *probe = nullptr) { check(lpt_renderer_set_resources(h_, scene.handle(), probe ? probe->handle() : nullptr)); }
    void raytrace(const Mat4 &view_transform) {
        check(lpt_renderer_set_accumulate(h_, accumulate ? 1 : 0));
        check(lpt_renderer_raytrace(h_, view_transform.data()));
    }
    /// queue.submit(encoder.finish()) (app.rs:335-337): launches what raytrace() has recorded; asynchronous
    void submit() { check(lpt_renderer_submit(h_)); }
    void reset_accumulation() { accumulate = false; check(lpt_renderer_reset_accumulation(h_)); }
    void upload_noise_texture(const uint8_t *rgba8, uint32_t w, uint32_t h, uint32_t bytes_per_row) { check(lpt_renderer_upload_noise(h_, rgba8, w, h, bytes_per_row)); }
    void use_noise_texture(bool flag) { check(lpt_renderer_use_noise(h_, flag ? 1 : 0)); }
    void set_blit_mode(BlitMode m) { check(lpt_renderer_set_blit_mode(h_, static_cast<int>(m))); }
    std::pair<uint32_t, uint32_t> get_size() const { uint32_t w = 0, h = 0; check(lpt_renderer_get_size(h_, &w, &h)); return {w, h}; }
    std::vector<uint8_t> read_pixels() {
        auto [w, h] = get_size();
        std::vector<uint8_t> out(static_cast<size_t>(w) * h * 4);
        check(lpt_renderer_read_pixels(h_, out.data()));
        return out;
    }
    void blit(uint8_t *dst, size_t row_bytes) { check(lpt_renderer_blit_rgba8(h_, dst, row_bytes)); }
    // build-only extensions
    std::vector<float> read_radiance() {
        auto [w, h] = get_size();
        std::vector<float> out(static_cast<size_t>(w) * h * 4);
        check(lpt_renderer_read_radiance(h_, out.data()));
        return out;
    }
    void set_max_bounces(uint32_t n) { check(lpt_renderer_set_max_bounces(h_, n)); }
    void set_seed(uint32_t s) { check(lpt_renderer_set_seed(h_, s)); }
    void set_vfov(float radians) { check(lpt_renderer_set_vfov(h_, radians)); }
    /// `weights` (one small integer per rank, the same on every rank; nullptr = equal shares): unequal tile shares, e.g. fewer tiles for the rank that also assembles the frame
    void set_shard(uint32_t rank, uint32_t world, uint32_t tile_w = 32, uint32_t tile_h = 8, const uint32_t *weights = nullptr) { check(lpt_renderer_set_shard_weighted(h_, rank, world, tile_w, tile_h, weights)); }
    void set_comm(const Comm *comm, const uint32_t *weights = nullptr) { check(lpt_renderer_set_comm_weighted(h_, comm ? comm->handle() : nullptr, weights)); }     // = set_shard(rank, world, 32, 8, weights) + the binding
    void exchange(int mode = LPT_EXCHANGE_GATHER_TILES) { check(lpt_renderer_exchange(h_, mode)); }             // rank 0 presents the whole frame
    void set_sort_queues(int flag) { check(lpt_renderer_set_sort_queues(h_, flag)); }
    void set_option(lpt_option option, uint64_t value) { check(lpt_renderer_set_option(h_, (int)option, value)); }
    uint64_t get_option(lpt_option option) const { uint64_t v = 0; check(lpt_renderer_get_option(h_, (int)option, &v)); return v; }
    lpt_ray_counts ray_counts() { lpt_ray_counts c; check(lpt_renderer_get_ray_counts(h_, &c)); return c; }
    /// multi-GPU denoising: this rank's filter inputs (device pointers) and, on rank 0 after the exchange, the filter passes
    void denoiser_inputs(void **noisy, void **gbuffer, void **motion, size_t *n_pixels) { check(lpt_renderer_denoiser_inputs(h_, noisy, gbuffer, motion, n_pixels)); }
    void denoise_filter() { check(lpt_renderer_denoise_filter(h_)); }
    lpt_renderer *handle() const { return h_; }

   private:
    lpt_renderer *h_ = nullptr;
};

}  // namespace loupiote
