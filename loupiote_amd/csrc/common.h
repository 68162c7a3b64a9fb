// common.h — internal declarations shared by the host-side translation units of
// libloupiote_hip.so.  Nothing here is part of the ABI (include/lpt.h is).
#pragma once
#include <cmath>
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/lpt.h"

namespace lpt {

// thread-local error text behind lpt_last_error()
void set_error(const char *fmt, ...);
int fail(int status, const char *fmt, ...);

struct Image {
    uint32_t width = 0, height = 0;
    std::vector<uint8_t> rgba8;
};

}  // namespace lpt

// The CPU-side scene: the flat arrays of the reference's Scene / BLASArray
// (reference crates/lib/src/scene.rs:30-54).
struct lpt_scene {
    std::vector<lpt_material> materials;
    std::vector<lpt_blas_entry> entries;
    std::vector<lpt_vertex> vertices;
    std::vector<uint32_t> indices;
    std::vector<lpt_instance> instances;
    std::vector<lpt_light> lights;
    std::vector<lpt::Image> images;
};

namespace lpt {

// ---- baked, device-ready acceleration data (host copies) -------------------
// 64 B compressed 8-wide node: FOUR 16-byte loads per visit (round 6: a load instruction of a visit is worth ~80 VALU instructions, and the fifth row of the
// 80-byte node cost 12 % of the traversal — profiles/r06_experiments_ab.txt).  What made room:
//  * the origin is a point of a 65536^3 SCENE GRID (Accel::grid_lo / grid_step, power-of-two steps): p = grid_lo + o * grid_step, at or below the node's box
//    minimum; the quantisation steps (ex, ey, ez) cover the node's box from THERE.  The grid spans 1.5 x the scene, so a node is at most 1 / 43 690 of the scene
//    looser than with an fp32 origin (conservative either way: the children are quantised against the origin the node really stores);
//  * no triangle base: the triangles of node i's leaf children have FIXED places, 16 * i + 2 * slot + k (k = 0, 1: a leaf child carries one or two triangles),
//    in woop / leaf_prim arrays of 16 entries per node (unused places are holes: zero Woop rows, LPT_INVALID_INDEX);
//  * leaf masks instead of a byte per slot: bit s of leaf1 = slot s is a leaf child, of leaf2 = ... with two triangles; in neither mask nor imask = empty (inverted box).
struct alignas(16) Node8 {
    uint16_t ox, oy, oz;         // origin on the scene grid
    uint8_t ex, ey, ez;          // biased power-of-two quantisation step per axis
    uint8_t imask, leaf1, leaf2; // bit s: slot s is an inner node / a leaf child / a leaf child with two triangles
    uint32_t child_base;         // index of the first inner child (inner children are contiguous, in slot order)
    uint8_t qlox[8], qloy[8], qloz[8], qhix[8], qhiy[8], qhiz[8];
};
static_assert(sizeof(Node8) == 64, "Node8 must be 64 bytes");
constexpr uint32_t kNodeTris = 16;   // triangle places per node

// The scene grid of the node origins: per axis a power-of-two step with 65535 steps >= 1.5 x the extent, the scene centred in it (a refit may move things by a quarter
// of the scene before the grid has to change; it is recomputed with every refit anyway).  Shared by the host builder and the device paths (device.hip).
inline void scene_grid(const float lo[3], const float hi[3], float grid_lo[3], float grid_step[3]) {
    for (int a = 0; a < 3; ++a) {
        const double m = std::fmax(std::fabs((double)lo[a]), std::fabs((double)hi[a]));
        // never narrower than 1e-4 of the largest |coordinate|: the slack around the scene (a quarter of this) must exceed the triangle padding (6e-6 of that coordinate,
        // bvh.cpp padded_box) even for a scene that is a speck far from the origin — a node's padded box must not reach below the grid
        const double ext = std::fmax((double)hi[a] - (double)lo[a], std::fmax(m * 1e-4, 1e-30));
        int k;
        std::frexp(1.5 * ext / 65535.0, &k);                    // 1.5 ext / 65535 = f * 2^k, f in [0.5, 1)  =>  2^k is above it
        k = k < -120 ? -120 : (k > 100 ? 100 : k);
        const double step = std::ldexp(1.0, k);
        const double slack = 0.5 * (65535.0 * step - ext);
        grid_step[a] = (float)step;
        grid_lo[a] = (float)(std::floor(((double)lo[a] - slack) / step) * step);
    }
}
// the grid point at or below `v` (what a node stores) and its coordinate (what every consumer of the node computes: ONE fp32 fma, kernels.h node_origin)
inline uint16_t grid_snap(float v, float grid_lo, float grid_step, float &p) {
    double u = std::floor(((double)v - (double)grid_lo) / (double)grid_step);
    u = u < 0.0 ? 0.0 : (u > 65535.0 ? 65535.0 : u);
    uint32_t ui = (uint32_t)u;
    p = std::fmaf((float)ui, grid_step, grid_lo);
    while (p > v && ui > 0u) { --ui; p = std::fmaf((float)ui, grid_step, grid_lo); }
    return (uint16_t)ui;
}

struct alignas(16) WoopTri {  // 48 B world -> unit-triangle affine map
    float r0[4], r1[4], r2[4];
};
static_assert(sizeof(WoopTri) == 48, "WoopTri must be 48 bytes");

struct Accel {
    std::vector<lpt_vertex> tri_verts;   // 3 per baked triangle (world space)
    std::vector<uint32_t> tri_material;  // per baked triangle
    std::vector<Node8> nodes;            // node 0 = root
    std::vector<WoopTri> woop;           // kNodeTris places per node (Node8); holes are zero
    std::vector<uint32_t> leaf_prim;     // triangle place -> baked triangle id (holes: LPT_INVALID_INDEX)
    float grid_lo[3] = {0.f, 0.f, 0.f}, grid_step[3] = {1.f, 1.f, 1.f};   // the scene grid of the node origins (scene_grid)
    uint32_t max_depth = 0;
    float build_ms = 0.f;
    float max_abs = 0.f;                 // the largest |coordinate| of any baked vertex: what the scene-wide part of the triangle padding follows (bvh.cpp padded_box)
    // bookkeeping for refits (lpt_scene_gpu_update_instances)
    std::vector<uint32_t> inst_first, inst_count;  // per scene instance: its run of baked triangles
    std::vector<uint32_t> level_start;             // nodes are stored breadth first: level l = [level_start[l], level_start[l+1])
};

constexpr float kScenePad = 2e-6f;   // triangle padding per unit of the scene's largest |coordinate| (host builder, LBVH and refit alike)

// bvh.cpp
int bake_and_build(const lpt_scene &scene, Accel &out);
void woop_from_triangle(const float p0[3], const float p1[3], const float p2[3], WoopTri &w);

// png.cpp
bool decode_png(const uint8_t *data, size_t size, Image &out);

// jpeg.cpp
bool decode_jpeg(const uint8_t *data, size_t size, Image &out);

// hdr.cpp
bool decode_hdr(const uint8_t *data, size_t size, uint32_t &width, uint32_t &height, std::vector<uint8_t> &rgbe);

}  // namespace lpt
