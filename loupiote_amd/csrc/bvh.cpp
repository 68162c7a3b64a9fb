// bvh.cpp — instance baking, host BVH construction and triangle pre-transform.
//
// Replaces what the reference delegates to albedo_rtx::BLASArray::add_bvh(_indexed)
// (reference crates/lib/src/loaders/gltf.rs:97-105 -> tinybvh-rs / obvhs, both absent
// from the reference tree): the acceleration structure the IntersectorPass walks.
// Instead of per-BLAS trees plus a linear instance loop, every instance is baked into
// world space (SPEC §2.5) and ONE tree is built over all triangles.
//
//   builder : binned SAH (16 bins / axis) binary tree, leaves <= 3 triangles, depth-capped
//             (falls back to median splits), then collapsed breadth-first into 8-wide nodes
//             by repeatedly opening the child with the largest surface area.
//   layout  : Node8 — 80-byte compressed wide node in the style of Ylitie, Karras & Laine,
//             "Efficient Incoherent Ray Traversal on GPUs Through Compressed Wide BVHs"
//             (HPG 2017): child boxes quantised to 8 bits on a per-node power-of-two grid,
//             children placed in octant-ordered slots so (slot XOR ray octant) is a
//             front-to-back order, leaf children referencing runs of <= 3 Woop triangles.
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdlib>
#include <memory>

#include "common.h"

// Builder A/B knobs exist only in builds made with -DLPT_EXPERIMENTS (csrc/Makefile `variant`): the shipped library reads no
// environment variable that changes what it builds.
static inline const char *lpt_experiment_env(const char *name) {
#ifdef LPT_EXPERIMENTS
    return getenv(name);
#else
    (void)name;
    return nullptr;
#endif
}

namespace lpt {

// SPEC §6 — computed in double, rounded to fp32 once.
void woop_from_triangle(const float p0[3], const float p1[3], const float p2[3], WoopTri &w) {
    const double ax = p0[0], ay = p0[1], az = p0[2];
    const double e1x = (double)p1[0] - ax, e1y = (double)p1[1] - ay, e1z = (double)p1[2] - az;
    const double e2x = (double)p2[0] - ax, e2y = (double)p2[1] - ay, e2z = (double)p2[2] - az;
    const double nx = e1y * e2z - e1z * e2y, ny = e1z * e2x - e1x * e2z, nz = e1x * e2y - e1y * e2x;
    const double det = nx * nx + ny * ny + nz * nz;
    if (!(det > 0.0)) { memset(&w, 0, sizeof w); return; }
    const double inv = 1.0 / det;
    const double r0x = (e2y * nz - e2z * ny) * inv, r0y = (e2z * nx - e2x * nz) * inv, r0z = (e2x * ny - e2y * nx) * inv;
    const double r1x = (ny * e1z - nz * e1y) * inv, r1y = (nz * e1x - nx * e1z) * inv, r1z = (nx * e1y - ny * e1x) * inv;
    const double r2x = nx * inv, r2y = ny * inv, r2z = nz * inv;
    w.r0[0] = (float)r0x; w.r0[1] = (float)r0y; w.r0[2] = (float)r0z; w.r0[3] = (float)(-(r0x * ax + r0y * ay + r0z * az));
    w.r1[0] = (float)r1x; w.r1[1] = (float)r1y; w.r1[2] = (float)r1z; w.r1[3] = (float)(-(r1x * ax + r1y * ay + r1z * az));
    w.r2[0] = (float)r2x; w.r2[1] = (float)r2y; w.r2[2] = (float)r2z; w.r2[3] = (float)(-(r2x * ax + r2y * ay + r2z * az));
}

namespace {

constexpr int kBins = 16;
constexpr uint32_t kLeafMax = 2;  // a leaf child of an 8-wide node carries at most 2 triangles (common.h Node8: fixed triangle places, leaf masks).  The SAH collapse rarely
                                  // wanted more: of the bench scene's 156 059 leaves 2 914 had three; capped, a ray visits 13.60 nodes instead of 13.58 and tests 4.15 triangles for 4.24
constexpr uint32_t kMaxDepth = 30;  // traversal stack is sized from this

struct Box {
    float lo[3] = {1e30f, 1e30f, 1e30f}, hi[3] = {-1e30f, -1e30f, -1e30f};
    void grow(const Box &b) { for (int a = 0; a < 3; ++a) { lo[a] = std::min(lo[a], b.lo[a]); hi[a] = std::max(hi[a], b.hi[a]); } }
    void grow(const float p[3]) { for (int a = 0; a < 3; ++a) { lo[a] = std::min(lo[a], p[a]); hi[a] = std::max(hi[a], p[a]); } }
    float half_area() const {
        float dx = hi[0] - lo[0], dy = hi[1] - lo[1], dz = hi[2] - lo[2];
        if (dx < 0.f) return 0.f;
        return dx * dy + dy * dz + dz * dx;
    }
};

struct Ref { Box box; float c[3]; uint32_t prim; };

// count != 0 marks a leaf of the binary tree; [pfirst, pfirst + pcount) is the ref range of any node's subtree
struct BuildNode { Box box; int32_t left = -1, right = -1; uint32_t first = 0, count = 0, pfirst = 0, pcount = 0; };

struct Builder {
    std::vector<Ref> refs;
    std::vector<BuildNode> nodes;
    uint32_t max_depth = 0;
    uint32_t leaf_max = kLeafMax;  // leaf size of the BINARY tree (1 when the collapse forms the leaves itself)

    static uint32_t ceil_log2(uint32_t x) { uint32_t l = 0; while ((1u << l) < x) ++l; return l; }

    int32_t build(uint32_t first, uint32_t count, uint32_t depth) {
        const int32_t id = (int32_t)nodes.size();
        nodes.emplace_back();
        Box box, cbox;
        for (uint32_t i = first; i < first + count; ++i) { box.grow(refs[i].box); cbox.grow(refs[i].c); }
        nodes[id].box = box;
        nodes[id].pfirst = first;
        nodes[id].pcount = count;
        max_depth = std::max(max_depth, depth);
        if (count <= leaf_max) { nodes[id].first = first; nodes[id].count = count; return id; }

        // depth budget: below this node we may still need ceil(log2(count/leaf)) median levels
        const uint32_t need = ceil_log2((count + leaf_max - 1) / leaf_max);
        const bool force_median = depth + need + 1 >= kMaxDepth;

        int best_axis = -1, best_split = -1;
        float best_cost = 1e30f;
        if (!force_median) {
            for (int a = 0; a < 3; ++a) {
                const float ext = cbox.hi[a] - cbox.lo[a];
                if (!(ext > 0.f)) continue;
                Box bb[kBins];
                uint32_t bc[kBins] = {0};
                const float scale = (float)kBins / ext;
                for (uint32_t i = first; i < first + count; ++i) {
                    int b = (int)((refs[i].c[a] - cbox.lo[a]) * scale);
                    b = std::min(std::max(b, 0), kBins - 1);
                    bb[b].grow(refs[i].box);
                    bc[b]++;
                }
                float right_area[kBins];
                uint32_t right_cnt[kBins];
                Box acc;
                uint32_t cnt = 0;
                for (int b = kBins - 1; b > 0; --b) { acc.grow(bb[b]); cnt += bc[b]; right_area[b] = acc.half_area(); right_cnt[b] = cnt; }
                Box lacc;
                uint32_t lcnt = 0;
                for (int b = 0; b < kBins - 1; ++b) {
                    lacc.grow(bb[b]);
                    lcnt += bc[b];
                    if (!lcnt || !right_cnt[b + 1]) continue;
                    const float cost = lacc.half_area() * (float)lcnt + right_area[b + 1] * (float)right_cnt[b + 1];
                    if (cost < best_cost) { best_cost = cost; best_axis = a; best_split = b; }
                }
            }
        }
        uint32_t mid = first;
        if (best_axis >= 0) {
            const float ext = cbox.hi[best_axis] - cbox.lo[best_axis];
            const float scale = (float)kBins / ext;
            const float lo = cbox.lo[best_axis];
            const int a = best_axis, s = best_split;
            auto it = std::partition(refs.begin() + first, refs.begin() + first + count, [&](const Ref &r) {
                int b = (int)((r.c[a] - lo) * scale);
                b = std::min(std::max(b, 0), kBins - 1);
                return b <= s;
            });
            mid = (uint32_t)(it - refs.begin());
        }
        if (mid == first || mid == first + count) {
            // median split along the widest centroid axis (also the depth-cap path)
            int a = 0;
            float ext = cbox.hi[0] - cbox.lo[0];
            for (int k = 1; k < 3; ++k) if (cbox.hi[k] - cbox.lo[k] > ext) { a = k; ext = cbox.hi[k] - cbox.lo[k]; }
            mid = first + count / 2;
            std::nth_element(refs.begin() + first, refs.begin() + mid, refs.begin() + first + count,
                             [a](const Ref &x, const Ref &y) { return x.c[a] < y.c[a] || (x.c[a] == y.c[a] && x.prim < y.prim); });
        }
        const int32_t l = build(first, mid - first, depth + 1);
        const int32_t r = build(mid, first + count - mid, depth + 1);
        nodes[id].left = l;
        nodes[id].right = r;
        return id;
    }
};

// Insertion-based optimisation of the binary tree (after Bittner, Hapala & Havran, "Fast Insertion-Based Optimization of
// Bounding Volume Hierarchies", CGF 2013): take a subtree out, close the gap, and put it back where it adds the least
// surface area to the tree (branch-and-bound search from the root; the place it came from is one of the candidates, so a
// move never makes the tree worse).  A pass works through the nodes with the largest boxes.  Topology only: every
// triangle stays in exactly one leaf, so the traversal results cannot change.
struct Reinserter {
    struct Item { float induced; int32_t id; };
    static constexpr int kMaxPops = 512;
    static bool later(const Item &a, const Item &b) { return a.induced > b.induced; }   // min-heap on the induced cost
    std::vector<BuildNode> &n;
    std::vector<int32_t> parent;
    std::vector<Item> heap;
    explicit Reinserter(std::vector<BuildNode> &nodes) : n(nodes), parent(nodes.size(), -1) {
        for (size_t i = 0; i < n.size(); ++i)
            if (!n[i].count) { parent[n[i].left] = (int32_t)i; parent[n[i].right] = (int32_t)i; }
    }
    static Box join(const Box &a, const Box &b) { Box r = a; r.grow(b); return r; }

    void reinsert(int32_t node) {
        const int32_t P = parent[node];
        if (P <= 0) return;                              // the root and its two children stay where they are
        const int32_t G = parent[P];
        const int32_t S = n[P].left == node ? n[P].right : n[P].left;
        const Box nb = n[node].box;
        const float na = nb.half_area();
        // take the subtree out: its sibling S takes the place of their parent P, the ancestors shrink
        (n[G].left == P ? n[G].left : n[G].right) = S;
        parent[S] = G;
        for (int32_t a = G; a >= 0; a = parent[a]) n[a].box = join(n[n[a].left].box, n[n[a].right].box);
        // best place X: minimise area(union(X, subtree)) — the box of the new parent — plus the growth of X's ancestors
        heap.clear();
        const float root_growth = join(n[0].box, nb).half_area() - n[0].box.half_area();
        heap.push_back({root_growth, n[0].left});
        heap.push_back({root_growth, n[0].right});
        std::make_heap(heap.begin(), heap.end(), later);
        // the place it came from is the first candidate, so the search — capped at kMaxPops nodes: coincident or heavily overlapping
        // boxes (hostile input) give the bound nothing to prune with — can only improve on it
        float best_cost = join(n[S].box, nb).half_area();
        for (int32_t a = G; a >= 0; a = parent[a]) best_cost += join(n[a].box, nb).half_area() - n[a].box.half_area();
        int32_t X = S;
        for (int pops = 0; !heap.empty() && pops < kMaxPops; ++pops) {
            std::pop_heap(heap.begin(), heap.end(), later);
            const Item it = heap.back();
            heap.pop_back();
            if (it.induced + na >= best_cost) break;     // even a union that adds nothing costs na
            const BuildNode &x = n[it.id];
            const float total = it.induced + join(x.box, nb).half_area();
            if (total < best_cost) { best_cost = total; X = it.id; }
            const float below = total - x.box.half_area();   // growth of x and of its ancestors if the subtree goes below x
            if (!x.count && below + na < best_cost) {
                heap.push_back({below, x.left});
                std::push_heap(heap.begin(), heap.end(), later);
                heap.push_back({below, x.right});
                std::push_heap(heap.begin(), heap.end(), later);
            }
        }
        // P, the free inner node, takes X's place and holds X and the subtree
        const int32_t XP = parent[X];
        (n[XP].left == X ? n[XP].left : n[XP].right) = P;
        parent[P] = XP;
        n[P].left = X; n[P].right = node;
        parent[X] = P; parent[node] = P;
        for (int32_t a = P; a >= 0; a = parent[a]) n[a].box = join(n[n[a].left].box, n[n[a].right].box);
    }

    void run(int passes, float fraction) {
        std::vector<std::pair<float, int32_t>> order;
        for (int pass = 0; pass < passes; ++pass) {
            order.clear();
            for (size_t i = 1; i < n.size(); ++i)
                if (parent[i] > 0) order.push_back({n[i].box.half_area(), (int32_t)i});
            const size_t take = std::min(order.size(), (size_t)((double)n.size() * fraction) + 1u);
            std::partial_sort(order.begin(), order.begin() + take, order.end(),
                              [](const std::pair<float, int32_t> &a, const std::pair<float, int32_t> &b) { return a.first > b.first || (a.first == b.first && a.second < b.second); });
            for (size_t k = 0; k < take; ++k) reinsert(order[k].second);
        }
    }

    // the builder's invariants again: node ids parents-first (the collapse walks them backwards), the refs of every
    // subtree contiguous (pfirst / pcount; the collapse forms leaves of up to kLeafMax triangles from such runs)
    void relinearise(std::vector<Ref> &refs, uint32_t &max_depth) {
        std::vector<BuildNode> out;
        std::vector<Ref> new_refs;
        out.reserve(n.size());
        new_refs.reserve(refs.size());
        struct Frame { int32_t old_id, new_id; uint32_t depth; int state; };
        std::vector<Frame> st;
        out.push_back(n[0]);
        st.push_back({0, 0, 0u, 0});
        max_depth = 0;
        while (!st.empty()) {
            Frame &f = st.back();
            const BuildNode &src = n[f.old_id];
            if (src.count) {
                BuildNode &leaf = out[f.new_id];
                leaf.first = leaf.pfirst = (uint32_t)new_refs.size();
                leaf.pcount = src.count;
                for (uint32_t t = 0; t < src.count; ++t) new_refs.push_back(refs[src.first + t]);
                max_depth = std::max(max_depth, f.depth);
                st.pop_back();
                continue;
            }
            if (f.state == 0 || f.state == 1) {
                const int32_t child = f.state == 0 ? src.left : src.right;
                const int32_t id = (int32_t)out.size();
                out.push_back(n[child]);
                (f.state == 0 ? out[f.new_id].left : out[f.new_id].right) = id;
                const uint32_t depth = f.depth + 1u;
                f.state++;
                st.push_back({child, id, depth, 0});      // invalidates f
                continue;
            }
            BuildNode &inner = out[f.new_id];
            inner.pfirst = out[inner.left].pfirst;
            inner.pcount = out[inner.left].pcount + out[inner.right].pcount;
            st.pop_back();
        }
        n.swap(out);
        refs.swap(new_refs);
    }
};

inline void normalize3(float v[3]) {
    float l2 = (v[0] * v[0] + v[1] * v[1]) + v[2] * v[2];
    if (!(l2 > 0.f)) { v[0] = v[1] = v[2] = 0.f; return; }
    float inv = 1.0f / sqrtf(l2);
    v[0] *= inv; v[1] *= inv; v[2] *= inv;
}

// SPEC §2.5: one instance -> world-space triangles (appended to `verts`, 3 per triangle); returns the material
uint32_t bake_one(const lpt_scene &s, size_t ii, std::vector<lpt_vertex> &verts) {
    const lpt_instance &inst = s.instances[ii];
    uint32_t mi = inst.material_index;
    if (mi >= s.materials.size()) mi = 0;
    if (inst.blas_index >= s.entries.size()) return mi;
    const lpt_blas_entry &e = s.entries[inst.blas_index];
    const uint32_t ntri = e.index_count / 3u;
    const float *m = inst.model_to_world;
    const float a00 = m[0], a10 = m[1], a20 = m[2], a01 = m[4], a11 = m[5], a21 = m[6], a02 = m[8], a12 = m[9], a22 = m[10];
    const float c00 = a11 * a22 - a12 * a21, c01 = a12 * a20 - a10 * a22, c02 = a10 * a21 - a11 * a20;
    const float c10 = a02 * a21 - a01 * a22, c11 = a00 * a22 - a02 * a20, c12 = a01 * a20 - a00 * a21;
    const float c20 = a01 * a12 - a02 * a11, c21 = a02 * a10 - a00 * a12, c22 = a00 * a11 - a01 * a10;
    for (uint32_t t = 0; t < ntri; ++t) {
        for (int k = 0; k < 3; ++k) {
            const lpt_vertex &v = s.vertices[e.vertex_offset + s.indices[e.index_offset + 3 * t + k]];
            const float x = v.position[0], y = v.position[1], z = v.position[2];
            lpt_vertex o;
            o.position[0] = ((m[0] * x + m[4] * y) + m[8] * z) + m[12];
            o.position[1] = ((m[1] * x + m[5] * y) + m[9] * z) + m[13];
            o.position[2] = ((m[2] * x + m[6] * y) + m[10] * z) + m[14];
            o.position[3] = v.position[3];
            const float nx = v.normal[0], ny = v.normal[1], nz = v.normal[2];
            float n[3] = {(c00 * nx + c01 * ny) + c02 * nz, (c10 * nx + c11 * ny) + c12 * nz, (c20 * nx + c21 * ny) + c22 * nz};
            normalize3(n);
            o.normal[0] = n[0]; o.normal[1] = n[1]; o.normal[2] = n[2]; o.normal[3] = v.normal[3];
            verts.push_back(o);
        }
    }
    return mi;
}

// SPEC §2.5: instances -> world-space soup
void bake(const lpt_scene &s, Accel &out) {
    out.tri_verts.clear();
    out.tri_material.clear();
    out.inst_first.assign(s.instances.size(), 0u);
    out.inst_count.assign(s.instances.size(), 0u);
    for (size_t ii = 0; ii < s.instances.size(); ++ii) {
        const size_t before = out.tri_verts.size() / 3;
        const uint32_t mi = bake_one(s, ii, out.tri_verts);
        const size_t ntri = out.tri_verts.size() / 3 - before;
        out.inst_first[ii] = (uint32_t)before;
        out.inst_count[ii] = (uint32_t)ntri;
        out.tri_material.insert(out.tri_material.end(), ntri, mi);
    }
}

// SAH-optimal collapse of the binary tree into 8-wide nodes (Ylitie, Karras & Laine 2017, §4.1).
// C(n,i) = cheapest way to represent the subtree of binary node n by at most i roots, a root being
// either an 8-wide node or a leaf of <= kLeafMax triangles:
//   C(n,1) = min( A_n * P_n * c_prim  [P_n <= kLeafMax],  D(n,8) + A_n * c_node )
//   C(n,i) = min( D(n,i), C(n,i-1) ),   D(n,j) = min_{0<k<j} C(left,k) + C(right,j-k)
// The binary tree is built down to single triangles so that the leaves are formed here.
struct Kid { int32_t node; bool leaf; };
struct Collapse {
    // relative cost of visiting an 8-wide node and of testing one triangle (LPT_BVH_PRIM_COST: experiments)
    static constexpr float kNodeCost = 1.0f;
    float kPrimCost = 0.3f;
    const std::vector<BuildNode> &n;
    std::vector<float> C;        // [7 * node + i - 1]
    std::vector<uint8_t> how;    // i == 1: 1 = 8-wide node, 0 = leaf;  i >= 2: 0 = same as C(n,i-1), else k of D(n,i)
    std::vector<uint8_t> k8;     // k of D(n,8), used when n becomes an 8-wide node
    explicit Collapse(const std::vector<BuildNode> &nodes) : n(nodes), C(7 * nodes.size()), how(7 * nodes.size(), 0), k8(nodes.size(), 0) {
        if (const char *ev = lpt_experiment_env("LPT_BVH_PRIM_COST")) { const float v = (float)atof(ev); if (v > 0.f) kPrimCost = v; }
        for (size_t id = nodes.size(); id-- > 0;) {
            const BuildNode &b = n[id];
            const float A = b.box.half_area();
            float *c = &C[7 * id];
            const float leaf = b.pcount <= kLeafMax ? A * (float)b.pcount * kPrimCost : 1e30f;
            if (b.count) { for (int i = 0; i < 7; ++i) c[i] = leaf; continue; }
            const float *l = &C[7 * (size_t)b.left], *r = &C[7 * (size_t)b.right];
            float D[9];
            uint8_t Dk[9];
            for (int j = 2; j <= 8; ++j) {
                D[j] = 1e30f; Dk[j] = 1;
                for (int k = 1; k < j; ++k) {
                    const float v = l[k - 1] + r[j - k - 1];
                    if (v < D[j]) { D[j] = v; Dk[j] = (uint8_t)k; }
                }
            }
            k8[id] = Dk[8];
            const float inner = D[8] + A * kNodeCost;
            c[0] = std::min(leaf, inner);
            how[7 * id] = inner < leaf ? 1 : 0;
            for (int i = 2; i <= 7; ++i) {
                if (D[i] < c[i - 2]) { c[i - 1] = D[i]; how[7 * id + i - 1] = Dk[i]; }
                else { c[i - 1] = c[i - 2]; how[7 * id + i - 1] = 0; }
            }
        }
    }
    void roots(int32_t m, int i, Kid *out, int &nk) const {   // the <= i roots C(m,i) stands for
        if (n[m].count) { out[nk++] = {m, true}; return; }
        while (i > 1 && how[7 * (size_t)m + i - 1] == 0) --i;
        if (i == 1) { out[nk++] = {m, how[7 * (size_t)m] == 0}; return; }
        const int k = how[7 * (size_t)m + i - 1];
        roots(n[m].left, k, out, nk);
        roots(n[m].right, i - k, out, nk);
    }
    int children(int32_t m, Kid *out) const {                  // m is an 8-wide node: its <= 8 children
        int nk = 0;
        if (n[m].count) { out[nk++] = {m, true}; return nk; }
        const int k = k8[m];
        roots(n[m].left, k, out, nk);
        roots(n[m].right, 8 - k, out, nk);
        return nk;
    }
};

// Triangle boxes are padded a little so that the box test stays conservative with
// respect to the Woop test's own rounding (the slab test adds its own ulp margins).  That rounding follows the triangle's coordinates AND the ray's, which may
// start anywhere in the scene: `pad_abs` = kScenePad x the scene's largest |coordinate| (round 5: millimetre triangles around the origin of a 2 000-unit scene lost
// 1-4 of 200 000 grazing hits from 1 000 units away without it; SPEC 7).
Box padded_box(const lpt_vertex *v, float pad_abs) {
    Box b;
    for (int k = 0; k < 3; ++k) b.grow(v[k].position);
    for (int a = 0; a < 3; ++a) {
        const float m = std::max(fabsf(b.lo[a]), fabsf(b.hi[a]));
        const float e = 4e-6f * m + pad_abs + 1e-6f * (b.hi[a] - b.lo[a]) + 1e-30f;
        b.lo[a] -= e;
        b.hi[a] += e;
    }
    return b;
}

// ---- TRIANGLE PRE-SPLITTING (round 6).  A triangle whose bounding box is mostly EMPTY — a long sliver across the hall — drags that box through the
// tree: in the mixed-scale hall (scenes.synthetic_hall) a ray tested 28.1 triangles for its 14.7 nodes and the frame took 21.0 ms (profiles/r06b_configs_timing.jsonl,
// profiles/r06_experiments_ab.txt E).  Before the build, the LARGEST references are split at the
// middle of their box's longest axis — the triangle clipped against both halves (Sutherland-Hodgman, binary64), each half's box = the clipped polygon's, padded like a
// triangle's — until no reference's EMPTY box area (below) exceeds kSplitRatio times the mean box area, or the budget of kSplitBudget extra references per triangle is
// spent.  What is split is decided by emptiness, not size: a ray enters a box in proportion to its surface and hits the triangle in proportion to its area, so
// half_area(box) - 2 * area(triangle part) is the share of entries that cannot end in a hit.  A wall of two axis-aligned triangles fills its (flat) box: nothing to gain,
// and splitting it by size alone cost 9 % on the quad-shelled atrium (more references along every wall); a sliver across the hall leaves its box empty: splitting it
// took the mixed-scale hall from 21.0 to 15.3 ms per frame at the ratio shipped.  The tree then holds a
// split triangle in several leaves: a ray may test it more than once (the closest-hit rule does not care), every point of it lies in at least one reference's box (the
// halves share their cut, the padding covers the clipping's rounding), so the boxes stay conservative and only the Woop test decides (SPEC §7): frames are unchanged.
// A scene of evenly sized triangles (the bench stand-in: largest box 10 x the mean) is not touched.  A refit (lpt_scene_gpu_update_instances) recomputes leaf boxes
// from whole triangles: correct, and as loose as before the split, until the next upload.
constexpr float kSplitRatio = 4.0f;     // a reference is split while the EMPTY part of its box's half area exceeds this many times the mean box half area
constexpr float kSplitBudget = 0.3f;    // at most this many extra references per triangle

// the triangle's part inside the box [lo, hi]: its bounds (false: nothing of it is inside)
static bool clipped_bounds(const lpt_vertex *v, const double lo[3], const double hi[3], double blo[3], double bhi[3], double &area) {
    double poly[16][3], tmp[16][3];
    int np = 3;
    for (int k = 0; k < 3; ++k) for (int a = 0; a < 3; ++a) poly[k][a] = (double)v[k].position[a];
    for (int a = 0; a < 3 && np; ++a)
        for (int side = 0; side < 2 && np; ++side) {
            const double plane = side ? hi[a] : lo[a], sgn = side ? -1.0 : 1.0;     // inside: sgn * (x - plane) >= 0
            int nt = 0;
            for (int i = 0; i < np; ++i) {
                const double *p = poly[i], *q = poly[(i + 1) % np];
                const double dp = sgn * (p[a] - plane), dq = sgn * (q[a] - plane);
                if (dp >= 0.0) { for (int c = 0; c < 3; ++c) tmp[nt][c] = p[c]; nt++; }
                if ((dp >= 0.0) != (dq >= 0.0)) {
                    const double t = dp / (dp - dq);
                    for (int c = 0; c < 3; ++c) tmp[nt][c] = p[c] + t * (q[c] - p[c]);
                    tmp[nt][a] = plane;
                    nt++;
                }
            }
            np = std::min(nt, 15);
            memcpy(poly, tmp, sizeof(double) * 3 * (size_t)np);
        }
    if (np < 1) return false;
    for (int a = 0; a < 3; ++a) { blo[a] = 1e300; bhi[a] = -1e300; }
    for (int i = 0; i < np; ++i) for (int a = 0; a < 3; ++a) { blo[a] = std::min(blo[a], poly[i][a]); bhi[a] = std::max(bhi[a], poly[i][a]); }
    double nx = 0.0, ny = 0.0, nz = 0.0;     // the (planar, convex) polygon's area: half the norm of the summed fan cross products
    for (int i = 1; i + 1 < np; ++i) {
        const double ux = poly[i][0] - poly[0][0], uy = poly[i][1] - poly[0][1], uz = poly[i][2] - poly[0][2];
        const double wx = poly[i + 1][0] - poly[0][0], wy = poly[i + 1][1] - poly[0][1], wz = poly[i + 1][2] - poly[0][2];
        nx += uy * wz - uz * wy; ny += uz * wx - ux * wz; nz += ux * wy - uy * wx;
    }
    area = 0.5 * std::sqrt(nx * nx + ny * ny + nz * nz);
    return true;
}

static Box padded_from(const double blo[3], const double bhi[3], float pad_abs) {
    Box b;
    for (int a = 0; a < 3; ++a) {
        // outward to fp32, then the triangle padding (padded_box): the clipping's own rounding is far inside it
        float l = (float)blo[a], h = (float)bhi[a];
        if ((double)l > blo[a]) l = std::nextafter(l, -3.0e38f);
        if ((double)h < bhi[a]) h = std::nextafter(h, 3.0e38f);
        const float m = std::max(fabsf(l), fabsf(h));
        const float e = 4e-6f * m + pad_abs + 1e-6f * (h - l) + 1e-30f;
        b.lo[a] = l - e;
        b.hi[a] = h + e;
    }
    return b;
}

static void presplit(const std::vector<lpt_vertex> &tri_verts, float pad_abs, std::vector<Ref> &refs) {
    const size_t n = refs.size();
    if (n < 2) return;
    double sum = 0.0;
    for (const Ref &r : refs) sum += (double)r.box.half_area();
    if (!(sum > 0.0)) return;
    auto tri_area = [&](uint32_t prim) {
        const lpt_vertex *v = &tri_verts[3 * (size_t)prim];
        const double ux = (double)v[1].position[0] - v[0].position[0], uy = (double)v[1].position[1] - v[0].position[1], uz = (double)v[1].position[2] - v[0].position[2];
        const double wx = (double)v[2].position[0] - v[0].position[0], wy = (double)v[2].position[1] - v[0].position[1], wz = (double)v[2].position[2] - v[0].position[2];
        const double cx = uy * wz - uz * wy, cy = uz * wx - ux * wz, cz = ux * wy - uy * wx;
        return 0.5 * std::sqrt(cx * cx + cy * cy + cz * cz);
    };
    // a max-heap of (empty half area of the box, ref index); a reference whose split makes no progress leaves it
    std::vector<std::pair<float, uint32_t>> heap;
    heap.reserve(n);
    for (uint32_t i = 0; i < (uint32_t)n; ++i) heap.push_back({(float)std::max(0.0, (double)refs[i].box.half_area() - 2.0 * tri_area(refs[i].prim)), i});
    std::make_heap(heap.begin(), heap.end());
    float ratio = kSplitRatio, budget_f = kSplitBudget;
    if (const char *ev = lpt_experiment_env("LPT_BVH_SPLIT")) sscanf(ev, "%f,%f", &ratio, &budget_f);   // A/B builds only: "ratio,budget"; "1e30,0" = no splitting
    const size_t budget = (size_t)((double)n * budget_f);
    size_t extra = 0;
    while (!heap.empty() && extra < budget) {
        const float empty = heap.front().first;
        const uint32_t idx = heap.front().second;
        if (!((double)empty > (double)ratio * sum / (double)refs.size())) break;
        std::pop_heap(heap.begin(), heap.end());
        heap.pop_back();
        const Ref r = refs[idx];
        const float area = r.box.half_area();
        int ax = 0;
        for (int a = 1; a < 3; ++a) if (r.box.hi[a] - r.box.lo[a] > r.box.hi[ax] - r.box.lo[ax]) ax = a;
        const double mid = 0.5 * ((double)r.box.lo[ax] + (double)r.box.hi[ax]);
        const lpt_vertex *v = &tri_verts[3 * (size_t)r.prim];
        Box half[2];
        bool have[2];
        double part[2] = {0.0, 0.0};
        for (int side = 0; side < 2; ++side) {
            double lo[3], hi[3], blo[3], bhi[3];
            for (int a = 0; a < 3; ++a) { lo[a] = r.box.lo[a]; hi[a] = r.box.hi[a]; }
            (side ? lo : hi)[ax] = mid;
            have[side] = clipped_bounds(v, lo, hi, blo, bhi, part[side]);
            if (have[side]) {
                half[side] = padded_from(blo, bhi, pad_abs);
                for (int a = 0; a < 3; ++a) { half[side].lo[a] = std::max(half[side].lo[a], r.box.lo[a]); half[side].hi[a] = std::min(half[side].hi[a], r.box.hi[a]); }   // never beyond the parent's
            }
        }
        if (!have[0] && !have[1]) continue;   // (cannot happen: the triangle is inside its own box)
        auto put = [&](uint32_t at, const Box &bx, double tri_part) {
            refs[at].box = bx;
            for (int a = 0; a < 3; ++a) refs[at].c[a] = 0.5f * (bx.lo[a] + bx.hi[a]);
            refs[at].prim = r.prim;
            const float ha = bx.half_area();
            if (ha < 0.9f * area) { heap.push_back({(float)std::max(0.0, (double)ha - 2.0 * tri_part), at}); std::push_heap(heap.begin(), heap.end()); }   // no progress: leave it be
            return ha;
        };
        sum -= (double)area;
        if (have[0] && have[1]) {
            refs.push_back(r);
            sum += (double)put(idx, half[0], part[0]);
            sum += (double)put((uint32_t)refs.size() - 1u, half[1], part[1]);
            extra++;
        } else {
            const int sd = have[0] ? 0 : 1;
            sum += (double)put(idx, half[sd], part[sd]);   // the triangle only reaches into one half: a tighter box, no new reference
        }
    }
    if (lpt_experiment_env("LPT_BVH_STATS"))
        fprintf(stderr, "[lpt bvh] presplit: %zu triangles -> %zu references (budget %zu); emptiest box now %.4g, mean box %.4g\n", n, refs.size(), budget,
                heap.empty() ? 0.0 : (double)heap.front().first, sum / (double)refs.size());
}

}  // namespace

int bake_and_build(const lpt_scene &scene, Accel &out) {
    const auto t0 = std::chrono::steady_clock::now();
    bake(scene, out);
    const uint32_t n = (uint32_t)out.tri_material.size();
    out.nodes.clear();
    out.woop.clear();
    out.leaf_prim.clear();
    out.max_depth = 0;
    if (n >= (1u << 29)) return fail(LPT_ERR_ACCEL_BUILD, "too many triangles (%u)", n);
    for (uint32_t t = 0; t < n; ++t)
        for (int k = 0; k < 3; ++k)
            for (int a = 0; a < 3; ++a)
                if (!std::isfinite(out.tri_verts[3 * (size_t)t + k].position[a]))
                    return fail(LPT_ERR_ACCEL_BUILD, "non-finite vertex in baked triangle %u", t);
    if (n == 0) {
        // a single node with eight empty slots: nothing can be hit
        Node8 root;
        memset(&root, 0, sizeof root);
        root.ex = root.ey = root.ez = 127;
        memset(root.qlox, 255, 24);
        out.nodes.push_back(root);
        WoopTri z;
        memset(&z, 0, sizeof z);
        out.woop.assign(kNodeTris, z);
        out.leaf_prim.assign(kNodeTris, LPT_INVALID_INDEX);
        out.max_depth = 1;
        out.level_start = {0u, 1u};
        return LPT_OK;
    }
    Builder b;
    b.refs.resize(n);
    out.max_abs = 0.0f;
    for (const lpt_vertex &v : out.tri_verts)
        for (int a = 0; a < 3; ++a) out.max_abs = std::max(out.max_abs, fabsf(v.position[a]));
    const float pad_abs = kScenePad * out.max_abs;
    for (uint32_t t = 0; t < n; ++t) {
        Ref &r = b.refs[t];
        r.box = padded_box(&out.tri_verts[3 * (size_t)t], pad_abs);
        for (int a = 0; a < 3; ++a) r.c[a] = 0.5f * (r.box.lo[a] + r.box.hi[a]);
        r.prim = t;
    }
    presplit(out.tri_verts, pad_abs, b.refs);
    const uint32_t n_refs = (uint32_t)b.refs.size();   // >= n: a split triangle has several references
    b.nodes.reserve(2 * (size_t)n_refs);
    const char *mode = lpt_experiment_env("LPT_BVH_COLLAPSE");  // "greedy" keeps the round-1 builder for A/B runs
    const bool use_dp = !(mode && strcmp(mode, "greedy") == 0);
    b.leaf_max = use_dp ? 1u : kLeafMax;
    b.build(0, n_refs, 0);
    if (use_dp && n_refs >= 64u) {
        // LPT_BVH_REINSERT="passes,fraction" (experiments); "0" keeps the tree as the top-down build left it
        // defaults: 4 passes over the 30 % largest boxes — on the 262 k-triangle atrium 2.6 % fewer nodes per ray and 1.7 % less
        // traversal time for 3x the (host) build time; more passes add little (profiles/r03_experiments_ab.txt)
        int passes = 4;
        float fraction = 0.3f;
        if (const char *ev = lpt_experiment_env("LPT_BVH_REINSERT")) { float f = fraction; const int got = sscanf(ev, "%d,%f", &passes, &f); if (got == 2 && f > 0.f && f <= 1.f) fraction = f; }
        if (passes > 0) {
            Reinserter opt(b.nodes);
            opt.run(std::min(passes, 64), fraction);
            opt.relinearise(b.refs, b.max_depth);
            if (b.max_depth > 2u * kMaxDepth) {   // moves may deepen a chain without bound on adversarial input: fall back to the depth-capped tree
                b.nodes.clear();
                b.max_depth = 0;
                b.build(0, n_refs, 0);
            }
        }
    }
    std::unique_ptr<Collapse> collapse;
    if (use_dp) collapse.reset(new Collapse(b.nodes));
    uint32_t stat_kids[9] = {0}, stat_leaf_tris[kLeafMax + 1] = {0};
    double sah = 0.0;

    // ---- collapse the binary tree into 8-wide nodes (breadth first, so siblings are adjacent)
    struct Pending { int32_t bvh2; uint32_t depth; };
    std::vector<Pending> queue;
    queue.push_back({0, 1});
    out.nodes.resize(1);
    {   // the scene grid of the node origins, from the padded triangle boxes (the root's box)
        const Box &rb = b.nodes[0].box;
        scene_grid(rb.lo, rb.hi, out.grid_lo, out.grid_step);
    }
    uint32_t tris_placed = 0;
    uint32_t max_depth = 1;
    out.level_start.clear();
    for (size_t w = 0; w < queue.size(); ++w) {
        const int32_t root = queue[w].bvh2;
        const uint32_t depth = queue[w].depth;
        max_depth = std::max(max_depth, depth);
        if (out.level_start.size() < depth) out.level_start.push_back((uint32_t)w);  // breadth first: depths are non-decreasing
        Kid kids[8];
        int nk = 0;
        if (collapse) {
            // the root may itself be a leaf-sized subtree: one node with one leaf child
            if (w == 0 && !b.nodes[root].count && collapse->how[7 * (size_t)root] == 0) kids[nk++] = {root, true};
            else nk = collapse->children(root, kids);
        } else {
            // greedy: open the inner member with the largest surface area until 8 (or none is left)
            if (b.nodes[root].count) kids[nk++] = {root, true};  // the whole tree is one leaf
            else { kids[nk++] = {b.nodes[root].left, false}; kids[nk++] = {b.nodes[root].right, false}; }
            for (int i = 0; i < nk; ++i) kids[i].leaf = b.nodes[kids[i].node].count != 0;
            while (nk < 8) {
                int best = -1;
                float best_area = -1.f;
                for (int i = 0; i < nk; ++i)
                    if (!kids[i].leaf) {
                        const float a = b.nodes[kids[i].node].box.half_area();
                        if (a > best_area) { best_area = a; best = i; }
                    }
                if (best < 0) break;
                const int32_t open = kids[best].node;
                kids[best] = {b.nodes[open].left, b.nodes[b.nodes[open].left].count != 0};
                kids[nk++] = {b.nodes[open].right, b.nodes[b.nodes[open].right].count != 0};
            }
        }
        stat_kids[nk]++;
        // node box = union of the children (they are padded already)
        Box nb;
        for (int i = 0; i < nk; ++i) nb.grow(b.nodes[kids[i].node].box);
        // slot assignment: slot s stands for the diagonal direction ds = (+-1,+-1,+-1) (bit set = +);
        // greedily give every child the free slot its offset from the node centre points to most.
        // Traversal visits slots in the order (slot XOR ray octant), i.e. roughly front to back.
        int slot_of[8];
        bool slot_used[8] = {false}, kid_done[8] = {false};
        float cx[3];
        for (int a = 0; a < 3; ++a) cx[a] = 0.5f * (nb.lo[a] + nb.hi[a]);
        for (int round = 0; round < nk; ++round) {
            float best = -1e30f;
            int bi = 0, bs = 0;
            for (int i = 0; i < nk; ++i) {
                if (kid_done[i]) continue;
                const Box &cb = b.nodes[kids[i].node].box;
                const float off[3] = {0.5f * (cb.lo[0] + cb.hi[0]) - cx[0], 0.5f * (cb.lo[1] + cb.hi[1]) - cx[1], 0.5f * (cb.lo[2] + cb.hi[2]) - cx[2]};
                for (int sl = 0; sl < 8; ++sl) {
                    if (slot_used[sl]) continue;
                    const float c = ((sl & 1) ? off[0] : -off[0]) + ((sl & 2) ? off[1] : -off[1]) + ((sl & 4) ? off[2] : -off[2]);
                    if (c > best) { best = c; bi = i; bs = sl; }
                }
            }
            kid_done[bi] = true;
            slot_used[bs] = true;
            slot_of[bi] = bs;
        }
        int kid_in_slot[8];
        for (int sl = 0; sl < 8; ++sl) kid_in_slot[sl] = -1;
        for (int i = 0; i < nk; ++i) kid_in_slot[slot_of[i]] = i;

        Node8 node;
        memset(&node, 0, sizeof node);
        float org[3];   // the node's origin: the grid point at or below its box minimum
        node.ox = grid_snap(nb.lo[0], out.grid_lo[0], out.grid_step[0], org[0]);
        node.oy = grid_snap(nb.lo[1], out.grid_lo[1], out.grid_step[1], org[1]);
        node.oz = grid_snap(nb.lo[2], out.grid_lo[2], out.grid_step[2], org[2]);
        double scale[3];
        uint8_t *eb[3] = {&node.ex, &node.ey, &node.ez};
        for (int a = 0; a < 3; ++a) {
            const double ext = (double)nb.hi[a] - (double)org[a];
            int e = -126;
            if (ext > 0.0) {
                int k;
                std::frexp(ext / 255.0, &k);  // ext/255 = m * 2^k, m in [0.5,1)  =>  255 * 2^k >= ext
                e = std::min(std::max(k, -126), 127);
            }
            *eb[a] = (uint8_t)(e + 127);
            scale[a] = std::ldexp(1.0, e);
        }
        node.child_base = (uint32_t)queue.size();
        if (out.woop.size() < (w + 1) * kNodeTris) {
            WoopTri hole;
            memset(&hole, 0, sizeof hole);
            out.woop.resize((w + 1) * kNodeTris, hole);
            out.leaf_prim.resize((w + 1) * kNodeTris, LPT_INVALID_INDEX);
        }
        for (int sl = 0; sl < 8; ++sl) {
            const int i = kid_in_slot[sl];
            uint8_t *q[6] = {&node.qlox[sl], &node.qloy[sl], &node.qloz[sl], &node.qhix[sl], &node.qhiy[sl], &node.qhiz[sl]};
            if (i < 0) {  // empty slot: inverted box, in no mask
                *q[0] = *q[1] = *q[2] = 255;
                *q[3] = *q[4] = *q[5] = 0;
                continue;
            }
            const BuildNode &c = b.nodes[kids[i].node];
            for (int a = 0; a < 3; ++a) {
                const double lo = std::floor(((double)c.box.lo[a] - (double)org[a]) / scale[a]);
                const double hi = std::ceil(((double)c.box.hi[a] - (double)org[a]) / scale[a]);
                *q[a] = (uint8_t)std::min(std::max(lo, 0.0), 255.0);
                *q[3 + a] = (uint8_t)std::min(std::max(hi, 0.0), 255.0);
            }
            if (kids[i].leaf) {
                // leaf child: its one or two triangles at their fixed places
                node.leaf1 |= (uint8_t)(1u << sl);
                if (c.pcount == 2u) node.leaf2 |= (uint8_t)(1u << sl);
                stat_leaf_tris[c.pcount]++;
                sah += (double)c.box.half_area() * c.pcount * 0.3;
                for (uint32_t t = 0; t < c.pcount; ++t) {
                    const uint32_t prim = b.refs[c.pfirst + t].prim;
                    const lpt_vertex *v = &out.tri_verts[3 * (size_t)prim];
                    WoopTri wt;
                    woop_from_triangle(v[0].position, v[1].position, v[2].position, wt);
                    const size_t place = w * kNodeTris + 2u * (size_t)sl + t;
                    out.woop[place] = wt;
                    out.leaf_prim[place] = prim;
                    tris_placed++;
                }
            } else {
                node.imask |= (uint8_t)(1u << sl);
                queue.push_back({kids[i].node, depth + 1});
            }
        }
        sah += (double)nb.half_area() * Collapse::kNodeCost;
        if (out.nodes.size() < queue.size()) out.nodes.resize(queue.size());
        out.nodes[w] = node;
    }
    if (lpt_experiment_env("LPT_BVH_STATS")) {
        fprintf(stderr, "[lpt bvh] %s: %zu nodes, depth %u, SAH %.3f, children/node:", use_dp ? "dp" : "greedy", out.nodes.size(), max_depth,
                sah / std::max((double)b.nodes[0].box.half_area(), 1e-30));
        for (int i = 1; i <= 8; ++i) fprintf(stderr, " %u", stat_kids[i]);
        fprintf(stderr, "; tris/leaf:");
        for (uint32_t i = 1; i <= kLeafMax; ++i) fprintf(stderr, " %u", stat_leaf_tris[i]);
        fprintf(stderr, "\n");
    }
    out.max_depth = max_depth;
    out.level_start.push_back((uint32_t)out.nodes.size());
    if (tris_placed != n_refs) return fail(LPT_ERR_ACCEL_BUILD, "internal: %u of %u triangle references placed", tris_placed, n_refs);
    {   // whole nodes' worth of places for every node (the last levels' nodes have no children of their own to make the arrays grow)
        WoopTri hole;
        memset(&hole, 0, sizeof hole);
        out.woop.resize(out.nodes.size() * kNodeTris, hole);
        out.leaf_prim.resize(out.nodes.size() * kNodeTris, LPT_INVALID_INDEX);
    }
    out.build_ms = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t0).count();
    return LPT_OK;
}

}  // namespace lpt
