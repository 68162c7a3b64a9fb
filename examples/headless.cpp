// headless.cpp — the per-frame protocol of the reference's standalone app, without the window.
//
// Replays what crates/standalone does around the hot path (reference crates/standalone/src/lib.rs:64-126,
// app.rs:62-68 start pose, app.rs:297-318 frame loop, app.rs:172-187 save_screenshot) through the C++
// mirror of `loupiote-core` (include/loupiote.hpp):
//     headless <scene.glb> <out.png> [width height frames bounces]
// Build:  g++ -std=c++17 -Iinclude examples/headless.cpp -Lloupiote_amd -lloupiote_hip -Wl,-rpath,$PWD/loupiote_amd -o headless
#include <cmath>
#include <cstdio>
#include <cstdlib>

#include "loupiote.hpp"

using namespace loupiote;

// CameraController::update (crates/standalone/src/camera.rs:66-110) for a resting camera
static Mat4 camera_to_world(const float o[3], const float dir_in[3]) {
    auto norm = [](float v[3]) { float l = std::sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]); v[0] /= l; v[1] /= l; v[2] /= l; };
    float d[3] = {dir_in[0], dir_in[1], dir_in[2]};
    norm(d);
    float r[3] = {d[1] * 0.f - d[2] * 1.f, d[2] * 0.f - d[0] * 0.f, d[0] * 1.f - d[1] * 0.f};  // direction x Y
    norm(r);
    float u[3] = {r[1] * d[2] - r[2] * d[1], r[2] * d[0] - r[0] * d[2], r[0] * d[1] - r[1] * d[0]};  // right x direction
    norm(u);
    return Mat4{r[0], r[1], r[2], 0, u[0], u[1], u[2], 0, d[0], d[1], d[2], 0, o[0], o[1], o[2], 1};
}

int main(int argc, char **argv) {
    if (argc < 3) { std::fprintf(stderr, "usage: %s scene.glb out.png [width height frames bounces]\n", argv[0]); return 2; }
    const uint32_t width = argc > 3 ? (uint32_t)std::atoi(argv[3]) : 1280, height = argc > 4 ? (uint32_t)std::atoi(argv[4]) : 720;
    const int frames = argc > 5 ? std::atoi(argv[5]) : 16;
    const uint32_t bounces = argc > 6 ? (uint32_t)std::atoi(argv[6]) : 3;  // reference constant (renderer.rs:398-399)
    setenv("GPU_MAX_HW_QUEUES", "8", 0);   // one hardware queue per HIP stream; the host's decision, before the first HIP call
    try {
        Device device(0);
        Scene scene;                                       // Scene::default()
        loaders::load_gltf_path(argv[1], scene);           // lib.rs:107-123
        SceneGPU scene_gpu = SceneGPU::new_from_scene(scene, device);
        Renderer renderer(device, width, height);          // lib.rs:66-70 (downsample_factor 0.5 like the reference)
        renderer.resize(scene_gpu, nullptr, width, height);  // app.resize -> Renderer::resize
        renderer.set_max_bounces(bounces);
        renderer.set_blit_mode(BlitMode::Pahtrace);
        const float origin[3] = {-10.f, 1.f, 0.f}, dir[3] = {1.f, 0.35f, 0.f};  // app.rs:64-67
        const Mat4 view = camera_to_world(origin, dir);
        bool accumulate_setting = true, camera_static = true;
        for (int f = 0; f < frames; ++f) {                 // app.rs:297-318
            if (!accumulate_setting || !camera_static || f == 0) renderer.reset_accumulation();
            renderer.use_noise_texture(false);
            renderer.raytrace(view);
            renderer.accumulate = true;
        }
        const auto size = renderer.get_size();
        const std::vector<uint8_t> px = renderer.read_pixels();   // save_screenshot (app.rs:172-187)
        check(lpt_write_png(argv[2], px.data(), size.first, size.second, (size_t)size.first * 4));
        const lpt_ray_counts c = renderer.ray_counts();
        std::printf("{\"width\": %u, \"height\": %u, \"frames\": %d, \"closest_rays\": %llu, \"shadow_rays\": %llu, \"png\": \"%s\"}\n",
                    size.first, size.second, frames, (unsigned long long)c.closest, (unsigned long long)c.shadow, argv[2]);
    } catch (const Error &e) {
        std::fprintf(stderr, "error (%d): %s\n", (int)e.kind, e.what());
        return 1;
    }
    return 0;
}
