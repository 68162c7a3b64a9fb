// bvh_check.cpp — TEST TOOL (never linked into the product): validates the host builder's output.
//
// Builds the 8-wide compressed BVH (loupiote_amd/csrc/bvh.cpp) for a triangle soup read from a
// file, then walks it on the CPU with a plain restatement of the node decoding that
// kernels.h:ray_step performs, and
//   * for small soups compares the closest hit (t, prim) of random rays with a brute-force loop
//     over every Woop triangle (the tree must never lose a hit: conservative boxes, full coverage),
//   * checks that every triangle is referenced exactly once,
//   * reports nodes / triangles visited per ray (a build-quality figure used for A/B of builders).
//
// usage: bvh_check <soup.bin> <n_rays> <brute:0|1> [ox oy oz]   (soup.bin: u32 n_tris, then 9 f32 per triangle)
// With an origin the rays are a mix of camera-like rays from that point and random segment rays;
// without, random segment rays inside the scene bounds.
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <random>

#include "../../loupiote_amd/csrc/common.h"

using namespace lpt;

namespace {

struct Hit { float t = 1e30f; uint32_t prim = 0xFFFFFFFFu; };

bool ray_tri(const WoopTri &w, const float o[3], const float d[3], float tmax, float &t) {
    const float oz = fmaf(w.r2[2], o[2], fmaf(w.r2[1], o[1], fmaf(w.r2[0], o[0], w.r2[3])));
    const float dz = fmaf(w.r2[2], d[2], fmaf(w.r2[1], d[1], w.r2[0] * d[0]));
    const float tt = -oz / dz;
    if (!(tt > 0.0f && tt <= tmax)) return false;
    const float ox = fmaf(w.r0[2], o[2], fmaf(w.r0[1], o[1], fmaf(w.r0[0], o[0], w.r0[3])));
    const float dx = fmaf(w.r0[2], d[2], fmaf(w.r0[1], d[1], w.r0[0] * d[0]));
    const float uu = fmaf(tt, dx, ox);
    if (!(uu >= 0.0f)) return false;
    const float oy = fmaf(w.r1[2], o[2], fmaf(w.r1[1], o[1], fmaf(w.r1[0], o[0], w.r1[3])));
    const float dy = fmaf(w.r1[2], d[2], fmaf(w.r1[1], d[1], w.r1[0] * d[0]));
    const float vv = fmaf(tt, dy, oy);
    if (!(vv >= 0.0f) || !(uu + vv <= 1.0f)) return false;
    t = tt;
    return true;
}

std::vector<uint32_t> g_tests;   // BVH_CHECK_TOP: tests per baked triangle (which triangles a tree makes the rays test again and again)
void consider(const Accel &a, uint32_t ti, const float o[3], const float d[3], Hit &best) {
    float t;
    if (!g_tests.empty() && a.leaf_prim[ti] < g_tests.size()) g_tests[a.leaf_prim[ti]]++;
    if (ray_tri(a.woop[ti], o, d, best.t, t)) {
        const uint32_t prim = a.leaf_prim[ti];
        if (t < best.t || prim < best.prim) { best.t = t; best.prim = prim; }
    }
}

float safe_inv(float d) { return fabsf(d) > 1.0e-30f ? 1.0f / d : copysignf(1.0e30f, d); }

struct Stats { uint64_t nodes = 0, tris = 0; uint32_t max_stack = 0; };
bool near_first = false;   // BVH_CHECK_ORDER=near1
bool by_distance = false;  // BVH_CHECK_ORDER=dist: visit inner children nearest first (bound on what ordering can save)

Hit walk(const Accel &a, const float o[3], const float d[3], Stats &st) {
    Hit best;
    const float inv[3] = {safe_inv(d[0]), safe_inv(d[1]), safe_inv(d[2])};
    const uint32_t oinv = 7u - ((inv[0] < 0 ? 1u : 0u) | (inv[1] < 0 ? 2u : 0u) | (inv[2] < 0 ? 4u : 0u));
    struct Entry { uint32_t node; uint32_t depth; };
    std::vector<Entry> stack;
    stack.push_back({0, 1});
    while (!stack.empty()) {
        const Entry e = stack.back();
        stack.pop_back();
        const Node8 &n = a.nodes[e.node];
        st.nodes++;
        const uint8_t ebytes[3] = {n.ex, n.ey, n.ez};
        const float p[3] = {fmaf((float)n.ox, a.grid_step[0], a.grid_lo[0]), fmaf((float)n.oy, a.grid_step[1], a.grid_lo[1]), fmaf((float)n.oz, a.grid_step[2], a.grid_lo[2])};
        const uint8_t *qlo[3] = {n.qlox, n.qloy, n.qloz}, *qhi[3] = {n.qhix, n.qhiy, n.qhiz};
        float an[3], bn[3], af[3], bf[3];
        for (int k = 0; k < 3; ++k) {
            uint32_t bits = (uint32_t)ebytes[k] << 23;
            float scale;
            memcpy(&scale, &bits, 4);
            const float A = scale * inv[k], B = (p[k] - o[k]) * inv[k];
            const float E = fmaf(fabsf(A), 255.0f, fabsf(B)) * 4.76837158203125e-7f;
            an[k] = A; bn[k] = B - E; af[k] = A; bf[k] = B + E;
        }
        // visit order: kernels.h takes hit bits from the top, bit = 24 + (slot ^ oinv)
        struct Child { uint32_t key, node; };
        Child inner[8];
        float inner_tn[8];
        int n_inner = 0;
        uint32_t rel = 0;
        for (int sl = 0; sl < 8; ++sl) {
            const bool is_inner = (n.imask >> sl) & 1u, is_leaf = (n.leaf1 >> sl) & 1u;
            const uint32_t my_rel = rel;
            if (is_inner) rel++;
            if (!is_inner && !is_leaf) continue;
            float tn = 0.0f, tf = best.t;
            for (int k = 0; k < 3; ++k) {
                const bool neg = inv[k] < 0.0f;
                const float qn = (float)(neg ? qhi[k][sl] : qlo[k][sl]), qf = (float)(neg ? qlo[k][sl] : qhi[k][sl]);
                tn = fmaxf(tn, fmaf(qn, an[k], bn[k]));
                tf = fminf(tf, fmaf(qf, af[k], bf[k]));
            }
            if (!(tn <= tf)) continue;
            if (is_inner) { inner_tn[n_inner] = tn; inner[n_inner++] = {by_distance ? ~__builtin_bit_cast(uint32_t, tn) : ((uint32_t)sl ^ oinv), n.child_base + my_rel}; }
            else {
                const uint32_t cnt = 1u + ((n.leaf2 >> sl) & 1u);
                for (uint32_t k = 0; k < cnt; ++k) { st.tris++; consider(a, e.node * kNodeTris + 2u * (uint32_t)sl + k, o, d, best); }
            }
        }
        if (near_first && n_inner > 1) {   // BVH_CHECK_ORDER=near1: the child entered first goes first, the others keep the octant order
            int bi = 0;
            for (int i = 1; i < n_inner; ++i) if (inner_tn[i] < inner_tn[bi]) bi = i;
            inner[bi].key = 0xFFFFFFFFu;
        }
        // push so that the largest key pops first
        for (int i = 0; i < n_inner; ++i)
            for (int j = i + 1; j < n_inner; ++j)
                if (inner[j].key < inner[i].key) std::swap(inner[i], inner[j]);
        for (int i = 0; i < n_inner; ++i) stack.push_back({inner[i].node, e.depth + 1});
        st.max_stack = std::max(st.max_stack, e.depth);
    }
    return best;
}

}  // namespace

int main(int argc, char **argv) {
    if (argc < 4) { fprintf(stderr, "usage: bvh_check soup.bin n_rays brute [ox oy oz]\n"); return 2; }
    FILE *f = fopen(argv[1], "rb");
    if (!f) { perror("soup"); return 2; }
    uint32_t n = 0;
    if (fread(&n, 4, 1, f) != 1) return 2;
    std::vector<float> pos((size_t)n * 9);
    if (n && fread(pos.data(), 4, pos.size(), f) != pos.size()) return 2;
    fclose(f);
    by_distance = getenv("BVH_CHECK_ORDER") && !strcmp(getenv("BVH_CHECK_ORDER"), "dist");
    near_first = getenv("BVH_CHECK_ORDER") && !strcmp(getenv("BVH_CHECK_ORDER"), "near1");
    const int n_rays = atoi(argv[2]);
    const bool brute = atoi(argv[3]) != 0;

    lpt_scene *scene = nullptr;
    lpt_scene_create(&scene);
    uint32_t blas = 0, inst = 0;
    const float ident[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
    if (n) {
        if (lpt_scene_add_mesh(scene, pos.data(), 12, nullptr, 0, nullptr, 0, n * 3, nullptr, 0, &blas) != LPT_OK) { fprintf(stderr, "add_mesh: %s\n", lpt_last_error()); return 1; }
        if (lpt_scene_add_instance(scene, blas, ident, 0, &inst) != LPT_OK) { fprintf(stderr, "add_instance: %s\n", lpt_last_error()); return 1; }
    }
    Accel acc;
    if (bake_and_build(*scene, acc) != LPT_OK) { fprintf(stderr, "build: %s\n", lpt_last_error()); return 1; }

    // structure checks: every baked triangle referenced exactly once; inner children contiguous
    const size_t n_baked = acc.tri_material.size();
    std::vector<uint8_t> seen(n_baked, 0);
    size_t bad_refs = 0;
    if (n_baked)
        for (uint32_t p : acc.leaf_prim) { if (p == LPT_INVALID_INDEX) continue; if (p >= n_baked) bad_refs++; else if (seen[p] < 255) seen[p]++; }   // holes: unused places; a split triangle (bvh.cpp presplit) has several
    for (size_t i = 0; i < n_baked; ++i) if (!seen[i]) bad_refs++;

    if (getenv("BVH_CHECK_TOP")) g_tests.assign(n_baked, 0u);
    float lo[3] = {1e30f, 1e30f, 1e30f}, hi[3] = {-1e30f, -1e30f, -1e30f};
    for (size_t i = 0; i < pos.size(); ++i) { lo[i % 3] = std::min(lo[i % 3], pos[i]); hi[i % 3] = std::max(hi[i % 3], pos[i]); }
    if (!n) { lo[0] = lo[1] = lo[2] = -1; hi[0] = hi[1] = hi[2] = 1; }
    std::mt19937 rng(12345);
    std::uniform_real_distribution<float> U(0.f, 1.f);
    Stats st;
    size_t mismatches = 0, hits = 0;
    for (int r = 0; r < n_rays; ++r) {
        float o[3], d[3];
        const bool from_eye = argc >= 7 && (r & 1);
        for (int k = 0; k < 3; ++k) {
            o[k] = from_eye ? (float)atof(argv[4 + k]) : lo[k] + (hi[k] - lo[k]) * U(rng);
            const float e = lo[k] + (hi[k] - lo[k]) * U(rng);
            d[k] = e - o[k];
        }
        if (n && (r & 3) == 2) {
            // aim at a vertex / an edge point of a random triangle: the ray grazes that triangle's box,
            // which is where a box test that is not conservative would lose the hit
            const size_t t = (size_t)(U(rng) * n) % n;
            const float w = (r & 4) ? U(rng) : 0.0f;
            for (int k = 0; k < 3; ++k) d[k] = (pos[9 * t + k] * (1.0f - w) + pos[9 * t + 3 + k] * w) - o[k];
        }
        const float len = sqrtf(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
        if (!(len > 0.f)) continue;
        for (int k = 0; k < 3; ++k) d[k] /= len;
        if ((r & 7) == 0) d[r % 3] = 0.0f;  // axis-parallel components exercise safe_inv
        const Hit h = walk(acc, o, d, st);
        if (h.prim != 0xFFFFFFFFu) hits++;
        if (brute) {
            Hit b;
            for (uint32_t ti = 0; ti < (uint32_t)acc.woop.size() && n_baked; ++ti) if (acc.leaf_prim[ti] != LPT_INVALID_INDEX) consider(acc, ti, o, d, b);
            if (b.prim != h.prim || b.t != h.t) mismatches++;
        }
    }
    if (!g_tests.empty()) {
        std::vector<uint32_t> order(g_tests.size());
        for (uint32_t i = 0; i < order.size(); ++i) order[i] = i;
        std::sort(order.begin(), order.end(), [&](uint32_t x, uint32_t y) { return g_tests[x] > g_tests[y]; });
        unsigned long long total = 0, top = 0;
        for (uint32_t c : g_tests) total += c;
        for (int k = 0; k < 20 && k < (int)order.size(); ++k) {
            const uint32_t pr = order[k];
            const lpt_vertex *v = &acc.tri_verts[3 * (size_t)pr];
            float lo3[3] = {1e30f, 1e30f, 1e30f}, hi3[3] = {-1e30f, -1e30f, -1e30f};
            for (int q = 0; q < 3; ++q) for (int a2 = 0; a2 < 3; ++a2) { lo3[a2] = std::min(lo3[a2], v[q].position[a2]); hi3[a2] = std::max(hi3[a2], v[q].position[a2]); }
            top += g_tests[pr];
            fprintf(stderr, "top %2d: prim %u tested %u times, box extent %.3f x %.3f x %.3f at (%.2f, %.2f, %.2f)\n", k, pr, g_tests[pr], hi3[0] - lo3[0], hi3[1] - lo3[1], hi3[2] - lo3[2], lo3[0], lo3[1], lo3[2]);
        }
        fprintf(stderr, "top 20 triangles: %llu of %llu tests\n", top, total);
        for (int dch = 0; dch < 20; ++dch) {
            unsigned long long part = 0;
            const size_t a0 = g_tests.size() * dch / 20, a1 = g_tests.size() * (dch + 1) / 20;
            for (size_t i = a0; i < a1; ++i) part += g_tests[i];
            fprintf(stderr, "prims [%zu, %zu): %llu tests\n", a0, a1, part);
        }
    }
    size_t n_refs = 0;
    for (uint32_t p : acc.leaf_prim) n_refs += p != LPT_INVALID_INDEX;
    printf("{\"triangles\": %zu, \"references\": %zu, \"nodes\": %zu, \"depth\": %u, \"bad_refs\": %zu, \"rays\": %d, \"hits\": %zu, \"mismatches\": %zu, "
           "\"nodes_per_ray\": %.4f, \"tris_per_ray\": %.4f, \"max_stack_depth\": %u, \"build_ms\": %.1f}\n",
           n_baked, n_refs, acc.nodes.size(), acc.max_depth, bad_refs, n_rays, hits, mismatches, (double)st.nodes / std::max(n_rays, 1),
           (double)st.tris / std::max(n_rays, 1), st.max_stack, acc.build_ms);
    lpt_scene_destroy(scene);
    return (bad_refs || mismatches) ? 1 : 0;
}
