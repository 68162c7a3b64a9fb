// build_kernels.h — GPU construction of the 8-wide BVH (lpt_scene_upload_ex with LPT_ACCEL_BUILD_GPU_LBVH).
//
// The reference builds its BVH on the CPU at load time (loaders/gltf.rs:97-105 -> tinybvh / obvhs); this is the
// MI355X-side alternative for scenes that change every frame:
//   k_lbvh_morton   30-bit Morton code of every triangle's (padded) box centre
//   (hipcub radix sort of code / triangle pairs, device.hip)
//   k_lbvh_tree     binary radix tree over the sorted codes — one thread per internal node, no synchronisation
//                   (Karras, "Maximizing Parallelism in the Construction of BVHs, Octrees, and k-d Trees", HPG 2012)
//   k_lbvh_fit      boxes bottom-up; the second thread to reach a node continues (atomic flags)
//   k_lbvh_collapse one launch per LEVEL of the wide tree: every 8-wide node opens its binary subtree greedily
//                   (largest box first) into <= 8 children, subtrees of <= 2 triangles become leaf children;
//                   children get octant-ordered slots; an exclusive scan over the level gives child / triangle bases
//   k_lbvh_emit     node topology (imask, leaf masks, bases), the triangles' places (leaf_prim), the next level's work list
//   k_refit_level   (kernels.h) then fills every node's grid origin / exponents / quantised planes bottom-up.
// The tree only decides WHICH boxes are visited; hits are decided by the Woop test and the (t, prim id) tie rule,
// so a scene built here renders bit-identically to one built by the host SAH builder (tests/test_gpu_lbvh.py).
#pragma once
#include "kernels.h"

namespace lptd {

struct LbvhTree {
    // binary radix tree over n sorted triangles: internal nodes 0..n-2 (0 = root); child refs >= 0 are internal
    // nodes, < 0 are ~(sorted triangle position)
    int *left, *right, *parent;       // parent of internal nodes
    int *leaf_parent;                 // parent of leaves
    uint32_t *first, *last;           // sorted range covered by an internal node
    float4 *lo, *hi;                  // internal node boxes
    float4 *tri_lo, *tri_hi;          // per SORTED position
    uint32_t *flag;
    const uint32_t *sorted_tri;       // sorted position -> baked triangle id
    const uint32_t *keys;             // sorted Morton codes
    uint32_t n;
};

__device__ __forceinline__ uint32_t expand10(uint32_t v) {
    v = (v * 0x00010001u) & 0xFF0000FFu;
    v = (v * 0x00000101u) & 0x0F00F00Fu;
    v = (v * 0x00000011u) & 0xC30C30C3u;
    v = (v * 0x00000005u) & 0x49249249u;
    return v;
}

__global__ __launch_bounds__(256) void k_lbvh_morton(DScene sc, uint32_t n, float3 blo, float3 binv, uint32_t *keys, uint32_t *vals) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n) return;
    float lo[3] = {3.0e38f, 3.0e38f, 3.0e38f}, hi[3] = {-3.0e38f, -3.0e38f, -3.0e38f};
    refit_grow_tri(sc, t, lo, hi);
    const float b[3] = {blo.x, blo.y, blo.z}, s[3] = {binv.x, binv.y, binv.z};
    uint32_t q[3];
    for (int a = 0; a < 3; ++a) {
        const float c = (0.5f * (lo[a] + hi[a]) - b[a]) * s[a];
        q[a] = (uint32_t)fminf(fmaxf(c * 1024.0f, 0.0f), 1023.0f);
    }
    keys[t] = (expand10(q[0]) << 2) | (expand10(q[1]) << 1) | expand10(q[2]);
    vals[t] = t;
}

// boxes of the sorted triangles (leaf boxes of the binary tree)
__global__ __launch_bounds__(256) void k_lbvh_leaf_boxes(DScene sc, LbvhTree T) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= T.n) return;
    float lo[3] = {3.0e38f, 3.0e38f, 3.0e38f}, hi[3] = {-3.0e38f, -3.0e38f, -3.0e38f};
    refit_grow_tri(sc, T.sorted_tri[i], lo, hi);
    T.tri_lo[i] = make_float4(lo[0], lo[1], lo[2], 0.f);
    T.tri_hi[i] = make_float4(hi[0], hi[1], hi[2], 0.f);
}

__device__ __forceinline__ int lbvh_delta(const uint32_t *keys, int n, int i, int j) {
    if (j < 0 || j >= n) return -1;
    const uint32_t a = keys[i], b = keys[j];
    if (a == b) return 32 + __clz((int)((uint32_t)i ^ (uint32_t)j));  // duplicate codes: the position breaks the tie
    return __clz((int)(a ^ b));
}

__global__ __launch_bounds__(256) void k_lbvh_tree(LbvhTree T) {
    const int i = (int)(blockIdx.x * blockDim.x + threadIdx.x);
    const int n = (int)T.n;
    if (i >= n - 1) return;
    const uint32_t *k = T.keys;
    const int d = (lbvh_delta(k, n, i, i + 1) - lbvh_delta(k, n, i, i - 1)) >= 0 ? 1 : -1;
    const int dmin = lbvh_delta(k, n, i, i - d);
    int lmax = 2;
    while (lbvh_delta(k, n, i, i + lmax * d) > dmin) lmax *= 2;
    int l = 0;
    for (int t = lmax / 2; t >= 1; t /= 2)
        if (lbvh_delta(k, n, i, i + (l + t) * d) > dmin) l += t;
    const int j = i + l * d;
    const int dnode = lbvh_delta(k, n, i, j);
    int s = 0, t = l;
    do {
        t = (t + 1) >> 1;
        if (lbvh_delta(k, n, i, i + (s + t) * d) > dnode) s += t;
    } while (t > 1);
    const int gamma = i + s * d + min(d, 0);
    const int lo = min(i, j), hi = max(i, j);
    const int left = lo == gamma ? ~gamma : gamma;
    const int right = hi == gamma + 1 ? ~(gamma + 1) : gamma + 1;
    T.left[i] = left; T.right[i] = right;
    T.first[i] = (uint32_t)lo; T.last[i] = (uint32_t)hi;
    if (left >= 0) T.parent[left] = i; else T.leaf_parent[~left] = i;
    if (right >= 0) T.parent[right] = i; else T.leaf_parent[~right] = i;
    if (i == 0) T.parent[0] = -1;
}

__device__ __forceinline__ void lbvh_child_box(const LbvhTree &T, int ref, float4 &lo, float4 &hi) {
    if (ref >= 0) { lo = T.lo[ref]; hi = T.hi[ref]; } else { lo = T.tri_lo[~ref]; hi = T.tri_hi[~ref]; }
}

__global__ __launch_bounds__(256) void k_lbvh_fit(LbvhTree T) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= T.n) return;
    int p = T.leaf_parent[i];
    while (p >= 0) {
        __threadfence();
        if (atomicAdd(&T.flag[p], 1u) == 0u) return;  // the sibling subtree is not finished: its thread will continue
        __threadfence();
        float4 al, ah, bl, bh;
        lbvh_child_box(T, T.left[p], al, ah);
        lbvh_child_box(T, T.right[p], bl, bh);
        T.lo[p] = make_float4(fminf(al.x, bl.x), fminf(al.y, bl.y), fminf(al.z, bl.z), 0.f);
        T.hi[p] = make_float4(fmaxf(ah.x, bh.x), fmaxf(ah.y, bh.y), fmaxf(ah.z, bh.z), 0.f);
        p = T.parent[p];
    }
}

// ---- collapse: one work item = one 8-wide node (the binary internal node it stands for)
struct LbvhLevel {
    const int *items;        // binary node per wide node of this level
    uint32_t n_items;
    int *kid_ref;            // 8 per item, slot order; kEmptyRef = empty
    uint32_t *inner_count;   // per item
    uint32_t *tri_count;     // per item
};
constexpr int kEmptyRef = 0x7FFFFFFF;

__device__ __forceinline__ uint32_t lbvh_count(const LbvhTree &T, int ref) { return ref >= 0 ? T.last[ref] - T.first[ref] + 1u : 1u; }
__device__ __forceinline__ bool lbvh_openable(const LbvhTree &T, int ref) { return ref >= 0 && lbvh_count(T, ref) > 2u; }   // a leaf child carries one or two triangles (common.h Node8)

__global__ __launch_bounds__(64) void k_lbvh_collapse(LbvhTree T, LbvhLevel L) {
    const uint32_t it = blockIdx.x * blockDim.x + threadIdx.x;
    if (it >= L.n_items) return;
    const int root = L.items[it];
    int kids[8];
    int nk = 2;
    kids[0] = T.left[root]; kids[1] = T.right[root];
    while (nk < 8) {
        int best = -1;
        float best_area = -1.0f;
        for (int i = 0; i < nk; ++i)
            if (lbvh_openable(T, kids[i])) {
                const float4 l = T.lo[kids[i]], h = T.hi[kids[i]];
                const float ex = h.x - l.x, ey = h.y - l.y, ez = h.z - l.z;
                const float a = ex * ey + ey * ez + ez * ex;
                if (a > best_area) { best_area = a; best = i; }
            }
        if (best < 0) break;
        const int open = kids[best];
        kids[best] = T.left[open];
        kids[nk++] = T.right[open];
    }
    // octant-ordered slots: greedily give every child the free slot its offset from the node centre points to most
    float4 klo[8], khi[8];
    float nlo[3] = {3.0e38f, 3.0e38f, 3.0e38f}, nhi[3] = {-3.0e38f, -3.0e38f, -3.0e38f};
    for (int i = 0; i < nk; ++i) {
        lbvh_child_box(T, kids[i], klo[i], khi[i]);
        nlo[0] = fminf(nlo[0], klo[i].x); nlo[1] = fminf(nlo[1], klo[i].y); nlo[2] = fminf(nlo[2], klo[i].z);
        nhi[0] = fmaxf(nhi[0], khi[i].x); nhi[1] = fmaxf(nhi[1], khi[i].y); nhi[2] = fmaxf(nhi[2], khi[i].z);
    }
    const float cx = 0.5f * (nlo[0] + nhi[0]), cy = 0.5f * (nlo[1] + nhi[1]), cz = 0.5f * (nlo[2] + nhi[2]);
    int slot_kid[8];
    for (int s = 0; s < 8; ++s) slot_kid[s] = kEmptyRef;
    uint32_t used = 0u, done = 0u;
    for (int round = 0; round < nk; ++round) {
        float best = -3.0e38f;
        int bi = 0, bs = 0;
        for (int i = 0; i < nk; ++i) {
            if ((done >> i) & 1u) continue;
            const float ox = 0.5f * (klo[i].x + khi[i].x) - cx, oy = 0.5f * (klo[i].y + khi[i].y) - cy, oz = 0.5f * (klo[i].z + khi[i].z) - cz;
            for (int s = 0; s < 8; ++s) {
                if ((used >> s) & 1u) continue;
                const float c = ((s & 1) ? ox : -ox) + ((s & 2) ? oy : -oy) + ((s & 4) ? oz : -oz);
                if (c > best) { best = c; bi = i; bs = s; }
            }
        }
        done |= 1u << bi; used |= 1u << bs;
        slot_kid[bs] = kids[bi];
    }
    uint32_t n_inner = 0, n_tris = 0;
    for (int s = 0; s < 8; ++s) {
        L.kid_ref[8 * (size_t)it + s] = slot_kid[s];
        if (slot_kid[s] == kEmptyRef) continue;
        if (lbvh_openable(T, slot_kid[s])) n_inner++;
        else n_tris += lbvh_count(T, slot_kid[s]);
    }
    L.inner_count[it] = n_inner;
    L.tri_count[it] = n_tris;
}

__global__ __launch_bounds__(64) void k_lbvh_emit(LbvhTree T, LbvhLevel L, const uint32_t *inner_off, const uint32_t *tri_off, uint32_t level_first,
                                                   uint32_t next_first, uint32_t tri_first, uint4 *nodes, uint32_t *leaf_prim, int *next_items) {
    const uint32_t it = blockIdx.x * blockDim.x + threadIdx.x;
    if (it >= L.n_items) return;
    const uint32_t child_base = next_first + inner_off[it], tri_base = (level_first + it) * kNodeTris;   // fixed triangle places: 16 * node + 2 * slot + k (common.h Node8)
    uint32_t imask = 0u, rel = 0u, leaf1 = 0u, leaf2 = 0u;
    (void)tri_off; (void)tri_first;
    for (int s = 0; s < 8; ++s) {
        const int ref = L.kid_ref[8 * (size_t)it + s];
        if (ref == kEmptyRef) continue;
        if (lbvh_openable(T, ref)) {
            imask |= 1u << s;
            next_items[inner_off[it] + rel++] = ref;
        } else {
            const uint32_t cnt = lbvh_count(T, ref);
            const uint32_t first = ref >= 0 ? T.first[ref] : (uint32_t)~ref;
            leaf1 |= 1u << s;
            if (cnt == 2u) leaf2 |= 1u << s;
            for (uint32_t k = 0; k < cnt; ++k) {
                const uint32_t prim = T.sorted_tri[first + k];
                leaf_prim[tri_base + 2u * (uint32_t)s + k] = prim;
            }
        }
    }
    uint4 *nw = nodes + 4u * (size_t)(level_first + it);
    nw[0] = make_uint4(0u, 0u, (imask << 8) | (leaf1 << 16) | (leaf2 << 24), child_base);  // origin / exponents / planes: k_refit_level
}

// scene bounds of the baked triangles (positions of the shading records): per-block reduction, then ordered-int atomics
__device__ __forceinline__ int lbvh_f2o(float f) { const int i = __float_as_int(f); return i >= 0 ? i : i ^ 0x7FFFFFFF; }  // order-preserving
__global__ __launch_bounds__(256) void k_lbvh_bounds(DScene sc, uint32_t n, int *bounds /* lo xyz, hi xyz as ordered ints */) {
    float lo[3] = {3.0e38f, 3.0e38f, 3.0e38f}, hi[3] = {-3.0e38f, -3.0e38f, -3.0e38f};
    for (uint32_t t = blockIdx.x * blockDim.x + threadIdx.x; t < n; t += gridDim.x * blockDim.x) {
        const float4 *tv = sc.tri_verts + kTriRec * (size_t)t;
        for (int k = 0; k < 3; ++k) {
            const float4 P = tv[2 * k];
            lo[0] = fminf(lo[0], P.x); lo[1] = fminf(lo[1], P.y); lo[2] = fminf(lo[2], P.z);
            hi[0] = fmaxf(hi[0], P.x); hi[1] = fmaxf(hi[1], P.y); hi[2] = fmaxf(hi[2], P.z);
        }
    }
    for (int a = 0; a < 3; ++a)
        for (int off = 32; off > 0; off >>= 1) { lo[a] = fminf(lo[a], __shfl_down(lo[a], off)); hi[a] = fmaxf(hi[a], __shfl_down(hi[a], off)); }
    if ((threadIdx.x & 63u) == 0)
        for (int a = 0; a < 3; ++a) { atomicMin(&bounds[a], lbvh_f2o(lo[a])); atomicMax(&bounds[3 + a], lbvh_f2o(hi[a])); }
}

// prim-ordered Woop maps -> the triangles' places in the tree.  PLACE-driven (every place looks up its triangle): a triangle the host builder split
// (bvh.cpp presplit) has several places, a hole has none.  Only the places of triangles [first, first + count) are written.
__global__ __launch_bounds__(256) void k_lbvh_scatter_woop(const float4 *src, float4 *woop, const uint32_t *leaf_prim, uint32_t n_places, uint32_t first, uint32_t count) {
    const uint32_t place = blockIdx.x * blockDim.x + threadIdx.x;
    if (place >= n_places) return;
    const uint32_t prim = leaf_prim[place];
    if (prim - first >= count) return;     // (holes are 0xFFFFFFFF)
    woop[3u * (size_t)place] = src[3u * (size_t)prim];
    woop[3u * (size_t)place + 1] = src[3u * (size_t)prim + 1];
    woop[3u * (size_t)place + 2] = src[3u * (size_t)prim + 2];
}

}  // namespace lptd
