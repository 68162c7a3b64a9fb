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
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <array>
#include <vector>
#include <cstring>

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

void consider(const Accel &a, uint32_t ti, const float o[3], const float d[3], Hit &best) {
    float t;
    if (ray_tri(a.woop[ti], o, d, best.t, t)) {
        const uint32_t prim = a.leaf_prim[ti];
        if (t < best.t || prim < best.prim) { best.t = t; best.prim = prim; }
    }
}

float safe_inv(float d) { return fabsf(d) > 1.0e-30f ? 1.0f / d : copysignf(1.0e30f, d); }

struct Stats { uint64_t nodes = 0, tris = 0; uint32_t max_stack = 0; };
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
        const float p[3] = {n.px, n.py, n.pz};
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
        int n_inner = 0;
        uint32_t rel = 0;
        for (int sl = 0; sl < 8; ++sl) {
            const uint8_t meta = n.meta[sl];
            const bool is_inner = (n.imask >> sl) & 1u;
            const uint32_t my_rel = rel;
            if (is_inner) rel++;
            if (!meta) continue;
            float tn = 0.0f, tf = best.t;
            for (int k = 0; k < 3; ++k) {
                const bool neg = inv[k] < 0.0f;
                const float qn = (float)(neg ? qhi[k][sl] : qlo[k][sl]), qf = (float)(neg ? qlo[k][sl] : qhi[k][sl]);
                tn = fmaxf(tn, fmaf(qn, an[k], bn[k]));
                tf = fminf(tf, fmaf(qf, af[k], bf[k]));
            }
            if (!(tn <= tf)) continue;
            if (is_inner) inner[n_inner++] = {by_distance ? ~__builtin_bit_cast(uint32_t, tn) : ((uint32_t)sl ^ oinv), n.child_base + my_rel};
            else {
                const uint32_t cnt_bits = meta >> 5, off = meta & 31u;
                for (uint32_t k = 0; k < 3; ++k)
                    if ((cnt_bits >> k) & 1u) { st.tris++; consider(a, n.tri_base + off + k, o, d, best); }
            }
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


// packet traversal estimate: one stack for all rays of the packet; a child is visited when ANY ray's clipped interval hits it;
// every triangle of a hit leaf is tested by every ray.  Returns (node visits, triangle tests) of the PACKET.
void walk_packet(const Accel &a, const std::vector<std::array<float,3>> &O, const std::vector<std::array<float,3>> &D, uint64_t &pn, uint64_t &pt, std::vector<Hit> &best) {
    const size_t R = O.size();
    best.assign(R, Hit());
    std::vector<std::array<float,3>> inv(R);
    for (size_t r = 0; r < R; ++r) for (int k = 0; k < 3; ++k) inv[r][k] = safe_inv(D[r][k]);
    const uint32_t oinv = 7u - ((inv[0][0] < 0 ? 1u : 0u) | (inv[0][1] < 0 ? 2u : 0u) | (inv[0][2] < 0 ? 4u : 0u));
    std::vector<uint32_t> stack{0};
    while (!stack.empty()) {
        const uint32_t ni = stack.back(); stack.pop_back();
        const Node8 &n = a.nodes[ni];
        pn++;
        const uint8_t ebytes[3] = {n.ex, n.ey, n.ez};
        const float p[3] = {n.px, n.py, n.pz};
        const uint8_t *qlo[3] = {n.qlox, n.qloy, n.qloz}, *qhi[3] = {n.qhix, n.qhiy, n.qhiz};
        struct Child { uint32_t key, node; };
        Child inner[8]; int n_inner = 0; uint32_t rel = 0;
        for (int sl = 0; sl < 8; ++sl) {
            const uint8_t meta = n.meta[sl];
            const bool is_inner = (n.imask >> sl) & 1u;
            const uint32_t my_rel = rel;
            if (is_inner) rel++;
            if (!meta) continue;
            bool any = false;
            for (size_t r = 0; r < R && !any; ++r) {
                float tn = 0.0f, tf = best[r].t;
                for (int k = 0; k < 3; ++k) {
                    uint32_t bits = (uint32_t)ebytes[k] << 23; float scale; memcpy(&scale, &bits, 4);
                    const float A = scale * inv[r][k], B = (p[k] - O[r][k]) * inv[r][k];
                    const float E = fmaf(fabsf(A), 255.0f, fabsf(B)) * 4.76837158203125e-7f;
                    const bool neg = inv[r][k] < 0.0f;
                    const float qn = (float)(neg ? qhi[k][sl] : qlo[k][sl]), qf = (float)(neg ? qlo[k][sl] : qhi[k][sl]);
                    tn = fmaxf(tn, fmaf(qn, A, B - E));
                    tf = fminf(tf, fmaf(qf, A, B + E));
                }
                any = tn <= tf;
            }
            if (!any) continue;
            if (is_inner) inner[n_inner++] = {(uint32_t)sl ^ oinv, n.child_base + my_rel};
            else {
                const uint32_t cnt_bits = meta >> 5, off = meta & 31u;
                for (uint32_t k = 0; k < 3; ++k)
                    if ((cnt_bits >> k) & 1u) { pt++; for (size_t r = 0; r < R; ++r) consider(a, n.tri_base + off + k, O[r].data(), D[r].data(), best[r]); }
            }
        }
        for (int i = 0; i < n_inner; ++i) for (int j = i + 1; j < n_inner; ++j) if (inner[j].key < inner[i].key) std::swap(inner[i], inner[j]);
        for (int i = 0; i < n_inner; ++i) stack.push_back(inner[i].node);
    }
}

// any-hit of ONE ray within [0, tmax]: nodes entered until the first hit (k_trace's shadow rays)
bool walk_any(const Accel &a, const float o[3], const float d[3], float tmax, uint64_t &nodes, uint64_t &tris) {
    Hit best; best.t = tmax;
    const float inv[3] = {safe_inv(d[0]), safe_inv(d[1]), safe_inv(d[2])};
    const uint32_t oinv = 7u - ((inv[0] < 0 ? 1u : 0u) | (inv[1] < 0 ? 2u : 0u) | (inv[2] < 0 ? 4u : 0u));
    std::vector<uint32_t> stack{0};
    while (!stack.empty()) {
        const uint32_t ni = stack.back(); stack.pop_back();
        const Node8 &n = a.nodes[ni];
        nodes++;
        const uint8_t ebytes[3] = {n.ex, n.ey, n.ez};
        const float p[3] = {n.px, n.py, n.pz};
        const uint8_t *qlo[3] = {n.qlox, n.qloy, n.qloz}, *qhi[3] = {n.qhix, n.qhiy, n.qhiz};
        struct Child { uint32_t key, node; };
        Child inner[8]; int n_inner = 0; uint32_t rel = 0;
        for (int sl = 0; sl < 8; ++sl) {
            const uint8_t meta = n.meta[sl];
            const bool is_inner = (n.imask >> sl) & 1u;
            const uint32_t my_rel = rel;
            if (is_inner) rel++;
            if (!meta) continue;
            float tn = 0.0f, tf = best.t;
            for (int k = 0; k < 3; ++k) {
                uint32_t bits = (uint32_t)ebytes[k] << 23; float scale; memcpy(&scale, &bits, 4);
                const float A = scale * inv[k], B = (p[k] - o[k]) * inv[k];
                const float E = fmaf(fabsf(A), 255.0f, fabsf(B)) * 4.76837158203125e-7f;
                const bool neg = inv[k] < 0.0f;
                tn = fmaxf(tn, fmaf((float)(neg ? qhi[k][sl] : qlo[k][sl]), A, B - E));
                tf = fminf(tf, fmaf((float)(neg ? qlo[k][sl] : qhi[k][sl]), A, B + E));
            }
            if (!(tn <= tf)) continue;
            if (is_inner) inner[n_inner++] = {(uint32_t)sl ^ oinv, n.child_base + my_rel};
            else {
                const uint32_t cnt_bits = meta >> 5, off = meta & 31u;
                for (uint32_t k = 0; k < 3; ++k)
                    if ((cnt_bits >> k) & 1u) { tris++; float t; if (ray_tri(a.woop[n.tri_base + off + k], o, d, tmax, t)) return true; }
            }
        }
        for (int i = 0; i < n_inner; ++i) for (int j = i + 1; j < n_inner; ++j) if (inner[j].key < inner[i].key) std::swap(inner[i], inner[j]);
        for (int i = 0; i < n_inner; ++i) stack.push_back(inner[i].node);
    }
    return false;
}
// any-hit of a packet: a ray that has found its occluder stops taking part
void walk_packet_any(const Accel &a, const std::vector<std::array<float,3>> &O, const std::vector<std::array<float,3>> &D, const std::vector<float> &tmax, uint64_t &pn, uint64_t &pt) {
    const size_t R = O.size();
    std::vector<char> done(R, 0);
    std::vector<std::array<float,3>> inv(R);
    for (size_t r = 0; r < R; ++r) for (int k = 0; k < 3; ++k) inv[r][k] = safe_inv(D[r][k]);
    const uint32_t oinv = 7u - ((inv[0][0] < 0 ? 1u : 0u) | (inv[0][1] < 0 ? 2u : 0u) | (inv[0][2] < 0 ? 4u : 0u));
    std::vector<uint32_t> stack{0};
    size_t left = R;
    while (!stack.empty() && left) {
        const uint32_t ni = stack.back(); stack.pop_back();
        const Node8 &n = a.nodes[ni];
        pn++;
        const uint8_t ebytes[3] = {n.ex, n.ey, n.ez};
        const float p[3] = {n.px, n.py, n.pz};
        const uint8_t *qlo[3] = {n.qlox, n.qloy, n.qloz}, *qhi[3] = {n.qhix, n.qhiy, n.qhiz};
        struct Child { uint32_t key, node; };
        Child inner[8]; int n_inner = 0; uint32_t rel = 0;
        for (int sl = 0; sl < 8; ++sl) {
            const uint8_t meta = n.meta[sl];
            const bool is_inner = (n.imask >> sl) & 1u;
            const uint32_t my_rel = rel;
            if (is_inner) rel++;
            if (!meta) continue;
            bool any = false;
            for (size_t r = 0; r < R && !any; ++r) {
                if (done[r]) continue;
                float tn = 0.0f, tf = tmax[r];
                for (int k = 0; k < 3; ++k) {
                    uint32_t bits = (uint32_t)ebytes[k] << 23; float scale; memcpy(&scale, &bits, 4);
                    const float A = scale * inv[r][k], B = (p[k] - O[r][k]) * inv[r][k];
                    const float E = fmaf(fabsf(A), 255.0f, fabsf(B)) * 4.76837158203125e-7f;
                    const bool neg = inv[r][k] < 0.0f;
                    tn = fmaxf(tn, fmaf((float)(neg ? qhi[k][sl] : qlo[k][sl]), A, B - E));
                    tf = fminf(tf, fmaf((float)(neg ? qlo[k][sl] : qhi[k][sl]), A, B + E));
                }
                any = tn <= tf;
            }
            if (!any) continue;
            if (is_inner) inner[n_inner++] = {(uint32_t)sl ^ oinv, n.child_base + my_rel};
            else {
                const uint32_t cnt_bits = meta >> 5, off = meta & 31u;
                for (uint32_t k = 0; k < 3; ++k)
                    if ((cnt_bits >> k) & 1u) { pt++; for (size_t r = 0; r < R; ++r) { float t; if (!done[r] && ray_tri(a.woop[n.tri_base + off + k], O[r].data(), D[r].data(), tmax[r], t)) { done[r] = 1; left--; } } }
            }
        }
        for (int i = 0; i < n_inner; ++i) for (int j = i + 1; j < n_inner; ++j) if (inner[j].key < inner[i].key) std::swap(inner[i], inner[j]);
        for (int i = 0; i < n_inner; ++i) stack.push_back(inner[i].node);
    }
}

}  // namespace

int main(int argc, char **argv) {
    if (argc < 2) { fprintf(stderr, "usage: packet_probe soup.bin\n"); return 2; }
    FILE *f = fopen(argv[1], "rb");
    uint32_t n = 0; if (!f || fread(&n, 4, 1, f) != 1) return 1;
    std::vector<float> pos((size_t)n * 9); if (n && fread(pos.data(), 4, pos.size(), f) != pos.size()) return 1; fclose(f);
    lpt_scene *scene = nullptr; lpt_scene_create(&scene);
    const float ident[16] = {1,0,0,0, 0,1,0,0, 0,0,1,0, 0,0,0,1};
    uint32_t blas = 0, inst = 0;
    lpt_scene_add_mesh(scene, pos.data(), 12, nullptr, 0, nullptr, 0, n * 3, nullptr, 0, &blas);
    lpt_scene_add_instance(scene, blas, ident, 0, &inst);
    Accel acc; if (bake_and_build(*scene, acc) != LPT_OK) return 1;
    const int W = 1920, H = 1080; const float vfov = 0.78539816339744830962f, th = tanf(0.5f * vfov), ax = th * W / H, ay = th;
    float fwd[3] = {1.0f, 0.35f, 0.0f}; { float l = sqrtf(fwd[0]*fwd[0]+fwd[1]*fwd[1]+fwd[2]*fwd[2]); for (float &v : fwd) v /= l; }
    float right[3] = {fwd[2]*0 - 0*fwd[1], 0, 0}; // right = fwd x up (up = +Y)
    right[0] = -fwd[2]; right[1] = 0; right[2] = fwd[0]; { float l = sqrtf(right[0]*right[0]+right[2]*right[2]); right[0]/=l; right[2]/=l; }
    float up[3] = {right[1]*fwd[2]-right[2]*fwd[1], right[2]*fwd[0]-right[0]*fwd[2], right[0]*fwd[1]-right[1]*fwd[0]};
    const float eye[3] = {-10.0f, 1.0f, 0.0f};
    std::mt19937 rng(3);
    for (int shape = 0; shape < 5; ++shape) {
        // shapes 3, 4: jittered rays (as the integrator's are) — 8x8 pixels x 1 sample, 4x4 pixels x 4 samples
        const int spp = shape == 4 ? 4 : 1;
        const bool jitter = shape >= 3;
        const int bw = shape == 0 || shape == 3 ? 8 : shape == 1 ? 32 : shape == 2 ? 16 : 4, bh = 64 / spp / bw;
        std::uniform_real_distribution<float> J(0.f, 1.f);
        uint64_t in = 0, it = 0, pn = 0, pt = 0, packets = 0, mism = 0;
        uint64_t sh_pn = 0, sh_pt = 0, sh_packets = 0, sh_in = 0, sh_it = 0, sh_rays = 0;
        for (int k = 0; k < 3000; ++k) {
            const int bx = (int)(rng() % (W / bw)) * bw, by = (int)(rng() % (H / bh)) * bh;
            std::vector<std::array<float,3>> O, D;
            for (int sm = 0; sm < spp; ++sm) for (int y = 0; y < bh; ++y) for (int x = 0; x < bw; ++x) {
                const float jx = jitter ? J(rng) : 0.5f, jy = jitter ? J(rng) : 0.5f;
                const float sx = (bx + x + jx) / W, sy = (by + y + jy) / H, cx = (2*sx-1)*ax, cy = (1-2*sy)*ay;
                float d[3]; for (int c = 0; c < 3; ++c) d[c] = right[c]*cx + up[c]*cy + fwd[c];
                const float l = sqrtf(d[0]*d[0]+d[1]*d[1]+d[2]*d[2]);
                O.push_back({eye[0], eye[1], eye[2]}); D.push_back({d[0]/l, d[1]/l, d[2]/l});
            }
            std::vector<Hit> hp; walk_packet(acc, O, D, pn, pt, hp);
            if (shape == 0) {   // the shadow rays of this packet's hits towards the atrium's 10 x 3 m ceiling light
                std::vector<std::array<float,3>> SO, SD; std::vector<float> ST;
                std::uniform_real_distribution<float> U(-1.f, 1.f);
                for (size_t r = 0; r < O.size(); ++r) {
                    if (hp[r].prim == 0xFFFFFFFFu) continue;
                    float P[3], L[3] = {0.f + 5.f * U(rng), 10.9f, 0.f + 1.5f * U(rng)}, dd[3];
                    for (int c = 0; c < 3; ++c) P[c] = O[r][c] + hp[r].t * D[r][c];
                    float len = 0; for (int c = 0; c < 3; ++c) { dd[c] = L[c] - P[c]; len += dd[c] * dd[c]; }
                    len = sqrtf(len); if (!(len > 1e-3f)) continue;
                    for (int c = 0; c < 3; ++c) dd[c] /= len;
                    SO.push_back({P[0] + 1e-3f * dd[0], P[1] + 1e-3f * dd[1], P[2] + 1e-3f * dd[2]}); SD.push_back({dd[0], dd[1], dd[2]}); ST.push_back(len - 2e-3f);
                }
                if (!SO.empty()) {
                    walk_packet_any(acc, SO, SD, ST, sh_pn, sh_pt); sh_packets++;
                    for (size_t r = 0; r < SO.size(); ++r) { walk_any(acc, SO[r].data(), SD[r].data(), ST[r], sh_in, sh_it); sh_rays++; }
                }
            }
            for (size_t r = 0; r < O.size(); ++r) { Stats st; Hit h = walk(acc, O[r].data(), D[r].data(), st); in += st.nodes; it += st.tris; if (h.prim != hp[r].prim || h.t != hp[r].t) mism++; }
            packets++;
        }
        const double Ni = (double)in / (packets * 64), Ti = (double)it / (packets * 64), Np = (double)pn / packets, Tp = (double)pt / packets;
        printf("packet %2dx%-2d%s: individual nodes/ray %.2f tris/ray %.2f | packet nodes %.1f tris %.1f | mismatches %llu | VALU model: packet %.0f vs individual %.0f wave-instr per 64 rays\n",
               bw, bh, shape == 4 ? " x 4 spp (jittered)" : shape == 3 ? " (jittered)" : "", Ni, Ti, Np, Tp, (unsigned long long)mism, Np * 245 + Tp * 60, 64.0 * Ni / 44.0 * 330.0);
        if (sh_packets) printf("  shadow rays of those hits: individual nodes/ray %.2f tris/ray %.2f (%.1f rays per packet) | packet nodes %.1f tris %.1f\n",
                               (double)sh_in / sh_rays, (double)sh_it / sh_rays, (double)sh_rays / sh_packets, (double)sh_pn / sh_packets, (double)sh_pt / sh_packets);
    }
    return 0;
}
