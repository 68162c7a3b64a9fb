// common.h — internal declarations shared by the host-side translation units of
// libloupiote_hip.so.  Nothing here is part of the ABI (include/lpt.h is).
#pragma once
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
struct alignas(16) Node8 {  // 80 B compressed 8-wide node (five 16-byte loads)
    float px, py, pz;            // origin of the node-local quantisation grid (= node box min)
    uint8_t ex, ey, ez, imask;   // biased power-of-two grid step per axis; bit s of imask: slot s is an inner node
    uint32_t child_base;         // index of the first inner child (inner children are contiguous, in slot order)
    uint32_t tri_base;           // first triangle of this node's leaf children (<= 24, contiguous)
    uint8_t meta[8];             // per slot: 0 empty | 0x20|(24+slot) inner | unary tri count<<5 | tri offset
    uint8_t qlox[8], qloy[8], qloz[8], qhix[8], qhiy[8], qhiz[8];
};
static_assert(sizeof(Node8) == 80, "Node8 must be 80 bytes");

struct alignas(16) WoopTri {  // 48 B world -> unit-triangle affine map
    float r0[4], r1[4], r2[4];
};
static_assert(sizeof(WoopTri) == 48, "WoopTri must be 48 bytes");

struct Accel {
    std::vector<lpt_vertex> tri_verts;   // 3 per baked triangle (world space)
    std::vector<uint32_t> tri_material;  // per baked triangle
    std::vector<Node8> nodes;            // node 0 = root
    std::vector<WoopTri> woop;           // in leaf order
    std::vector<uint32_t> leaf_prim;     // leaf slot -> baked triangle id
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
