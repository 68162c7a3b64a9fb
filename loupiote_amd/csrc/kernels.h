// kernels.h — the wavefront path-tracing kernels for gfx950 (included by device.hip).
//
// Stage kernels, one per pass the reference records in Renderer::raytrace
// (reference crates/lib/src/renderer.rs:444-538):
//   k_raygen      <- passes::RayPass           (:444-448)
//   k_trace (+ k_trace_packet for bounce 0) <- passes::IntersectorPass (:458-463, :493-498) and the shadow rays the reference's
//                    shading pass traces inline
//   k_shade       <- passes::PrimaryRayPass / passes::ShadingPass (:472-479, :502-508)
//   k_accumulate  <- passes::AccumulationPass  (:525-533)
// Unlike the reference (full pixel grid per dispatch, one ray slot per pixel) live
// paths are stream-compacted into a queue after every bounce with a wave64
// ballot + popcount prefix and ONE atomic per wave, and the traversal kernels are
// persistent: a fixed grid of waves pulls 64-ray packets from a device-side head.
#pragma once
#include "device_math.h"

namespace lptd {

constexpr int kMaxBounces = 64;
constexpr int kBlock = 256;
constexpr int kTraceBlock = 64;  // traversal kernels: one wave per block, so LDS / slots free up per wave

// 64 B compressed 8-wide node, mirrors lpt::Node8 (common.h); read as FOUR 16-byte loads
struct DNode8 {
    uint4 n0;  // ox | oy << 16, oz | ex << 16 | ey << 24, ez | imask << 8 | leaf1 << 16 | leaf2 << 24, child_base
    uint4 n2;  // qlo_x[0..3], qlo_x[4..7], qlo_y[0..3], qlo_y[4..7]
    uint4 n3;  // qlo_z[0..3], qlo_z[4..7], qhi_x[0..3], qhi_x[4..7]
    uint4 n4;  // qhi_y[0..3], qhi_y[4..7], qhi_z[0..3], qhi_z[4..7]
};
using lpt::kNodeTris;   // triangle places per node (common.h): node i's leaf slot s keeps its one or two triangles at 16 i + 2 s + k

// The node's header words.  The origin is a point of the scene grid (DScene::grid_*): ONE fp32 fma per axis, the same expression in every builder and consumer.
__device__ __forceinline__ uint32_t node_imask(const uint4 &n0) { return (n0.z >> 8) & 0xFFu; }
__device__ __forceinline__ uint32_t node_leaves(const uint4 &n0) { return n0.z >> 16; }   // V = leaf1 | leaf2 << 8: bit s = slot s is a leaf child, bit 8 + s = ... with two triangles
__device__ __forceinline__ uint32_t leaf_count(uint32_t V, uint32_t slot) { return 1u + ((V >> (8u + slot)) & 1u); }
struct NodeGrid { float px, py, pz, sx, sy, sz; };   // origin and quantisation steps (powers of two) of a node
template <typename Scene>
__device__ __forceinline__ NodeGrid node_grid(const Scene &sc, const uint4 &n0) {
    NodeGrid g;
    g.px = fmaf((float)(n0.x & 0xFFFFu), sc.grid_step[0], sc.grid_lo[0]);
    g.py = fmaf((float)(n0.x >> 16), sc.grid_step[1], sc.grid_lo[1]);
    g.pz = fmaf((float)(n0.y & 0xFFFFu), sc.grid_step[2], sc.grid_lo[2]);
    g.sx = __uint_as_float((n0.y << 7) & 0x7F800000u);
    g.sy = __uint_as_float((n0.y >> 1) & 0x7F800000u);
    g.sz = __uint_as_float((n0.z << 23) & 0x7F800000u);
    return g;
}
// A TRIANGLE GROUP (16 * node, T): T = the triangles of one node that a ray still has to test — bit s = the first, bit 8 + s = the second triangle of leaf slot s;
// their places are fixed: 16 * node + 2 * s + k.  (Places packed slot after slot inside a node's block — the index by popcount over the leaf masks carried in the
// group's upper half — measured 2.6 % slower: the pop is on every triangle's path, profiles/r06_experiments_ab.txt D.)
__device__ __forceinline__ uint32_t tg_index(uint32_t base, uint32_t bit) { return base + 2u * (bit & 7u) + (bit >> 3); }
__device__ __forceinline__ uint32_t tg_pop(uint2 &tg) {   // takes the next triangle out of a non-empty group
    const uint32_t bit = (uint32_t)__ffs((int)tg.y) - 1u;
    tg.y &= tg.y - 1u;
    return tg_index(tg.x, bit);
}
// bit s of an 8-bit mask -> bit s ^ o (o = 0..7): three conditional delta swaps
__device__ __forceinline__ uint32_t xor_permute8(uint32_t x, uint32_t o) {
    uint32_t t = ((x >> 1) ^ x) & ((o & 1u) ? 0x55u : 0u); x ^= t | (t << 1);
    t = ((x >> 2) ^ x) & ((o & 2u) ? 0x33u : 0u); x ^= t | (t << 2);
    t = ((x >> 4) ^ x) & ((o & 4u) ? 0x0Fu : 0u); x ^= t | (t << 4);
    return x;
}

struct DImage { uint32_t offset, width, height, pad; };  // pad = 8x4-texel tiles per row (texels are tiled, device.hip)
constexpr uint32_t kTriRec = 8;

struct DScene {
    const DNode8 *nodes;
    uint32_t stack_entries;      // wide-tree depth + 1: per-lane traversal stack capacity (uint2 entries)
    const float4 *woop;          // 3 per leaf slot
    const uint32_t *leaf_prim;   // leaf slot -> baked triangle id
    // shading record: kTriRec float4 = 128 B per baked triangle, one cache line: (pos,u)(nrm,v) x3 and the triangle's
    // material verbatim (colour | roughness, reflectivity, albedo / mra texture).  One fetch instead of a 96-B run that
    // straddles two lines half of the time + the material index + the material (k_shade is bound by L2-miss traffic)
    const float4 *tri_verts;
    const lpt_material *materials;
    const lpt_light *lights;
    const uint8_t *texels;       // RGBA8, all images back to back
    const DImage *images;
    const float *srgb_lut;       // 256 entries
    // paired textures: (albedo RGBA8, mra RGBA8) per texel in APRON tiles — 4x4 stored texels of 128 B that cover a 3x3 block of the image plus
    // its right / lower neighbours (wrapped), so the four taps of a lookup never leave one cache line — for materials whose two textures
    // have one size; such a material's record carries kPairedBit | the pair's texel offset as its albedo texture and width | height << 16 as its
    // mra texture (device.hip: device_material)
    const uint2 *pair_texels;
    const DImage *pair_images;   // offset in 8-byte texels, pad = tiles per row (ceil(width / 3))
    uint32_t n_tris, n_materials, n_lights, n_images, n_pairs;
    float grid_lo[3], grid_step[3];   // the scene grid of the node origins (common.h scene_grid): origin = grid_lo + o * grid_step
    float pad_abs;               // the scene-wide part of the triangle padding (refit / LBVH; bvh.cpp padded_box): kScenePad x the scene's largest |coordinate|
};
constexpr uint32_t kPairedBit = 0x40000000u;

struct DProbe { const uint8_t *rgbe; uint32_t w, h; };
struct DNoise { const uint8_t *rgba; uint32_t w, h, enabled; };

// o.w = pdf of the sampling bounce (<0: camera), d.w = pixel slot bits, T.w = x | y << 13 | sample << 26.
// k_trace reads o and d; k_shade reads d and T (and o only for the emitter MIS weight / the G-buffer): the pdf rides
// with the origin so that shading streams 48 B per ray (d, T, hit) instead of 64
struct Queue { float4 *o; float4 *d; float4 *T; };
struct ShadowQueue { float4 *o; float4 *d; float4 *c; };    // o.w = tmax, d.w = pixel slot bits, c = contribution

constexpr uint32_t kTailCounters = 64u;
struct FrameCounters {
    // Queue sizes.  The shadow-ray count of bounce b and the closest-hit count of bounce b + 1 — the two queues ONE shading pass fills — sit side by
    // side (qs[2b], qs[2b + 1]), so that a block reserves its slots in both with ONE 64-bit atomic (k_shade's default path) instead of two round trips;
    // the primary rays' count has a word of its own.  Accessors: QC (closest-hit rays of bounce b), SC (shadow rays emitted by bounce b).
    uint32_t q0, q0_pad;
    uint32_t qs[2 * kMaxBounces];
    // chunk heads: 8 per bounce (one per XCD), each on its own 128-byte line so that the atomics of
    // different heads do not serialise on one L2 line
    uint32_t ihead[kMaxBounces * 8 * 32];
    uint32_t shead[kMaxBounces * 8 * 32];
    // the stragglers of a traversal launch (k_trace with a step budget): rays that were not finished within the budget, re-traced by k_trace_coop;
    // one count per launch of a wavefront (index: the bounce of its closest-hit rays, max_bounces for the last, shadow-only launch)
    uint32_t strag_count[kMaxBounces + 1];
    uint32_t phead[8 * 32];       // k_path's own chunk heads over the bounce-0 queue (ihead[0] may have been drained by a per-ray bounce-0 launch)
    uint32_t shaded[kMaxBounces];
    // k_shade's surface-hit counts: ONE atomic per block and launch, spread over 8 words per bounce on separate 128-byte lines (one atomic per WAVE on
    // shaded[bounce] — 4096 of them on one word, all in the last microseconds of a launch — was the floor of a shard-sized shading launch: DESIGN §5.2)
    uint32_t shaded_part[kMaxBounces * 8 * 32];
    unsigned long long nodes, tris, shadow_nodes, shadow_tris;
    unsigned long long wave_steps, live_lanes, node_lanes, tri_lanes;  // closest-hit kernel, stats only
    unsigned long long packet_nodes, packet_tris;   // k_trace_packet, stats only: nodes entered / triangles tested per PACKET, summed
    // stats only: the shadow rays that found an occluder, and the occluder-cache PROBE (k_trace<STATS>: what a cache of the last occluding
    // triangle per origin cell would have answered — entries found, entries whose triangle occludes the new ray; the traversal is not changed)
    unsigned long long shadow_occluded, occ_found, occ_hits;
    // stats only: traversal steps of a ray in k_trace — the longest one of the launch sets its duration —: maximum and a histogram by
    // power of two (bucket k: 2^k <= steps < 2^(k+1), k = 0..11)
    uint32_t max_steps;
    uint32_t step_hist[12];
    // rays the waves of k_trace<.., TAIL> finished cooperatively, in place: one atomic per wave that got there, spread over kTailCounters words on lines of their own — every
    // wave of a launch gets there within a few microseconds, and 6 000 atomics on ONE word (~88 per us) held each launch's end back by ~30 us (profiles/r06_experiments_ab.txt I)
    uint32_t tail_rays[kTailCounters * 32];
};
__device__ __host__ __forceinline__ uint32_t &QC(FrameCounters *c, int b) { return b == 0 ? c->q0 : c->qs[2 * (b - 1) + 1]; }
__device__ __host__ __forceinline__ uint32_t &SC(FrameCounters *c, int b) { return c->qs[2 * b]; }
struct Totals { unsigned long long closest, shadow, shaded, nodes, tris, shadow_nodes, shadow_tris, wave_steps, live_lanes, node_lanes, tri_lanes,
                                   primary, packet_nodes, packet_tris, shadow_occluded, occ_found, occ_hits, wave_rays; };   // mirrors lpt_ray_counts

// Tile ownership.  Tile t (row-major over the tile grid) belongs to VIRTUAL rank t % V; the V virtual ranks are dealt to
// the ranks in proportion to their weights (a rank that also assembles, reads back or filters the frame gets fewer tiles).  With
// every weight 1 this is V = world, virtual rank = rank: the plain "tile id mod N" rule.
constexpr uint32_t kMaxVirtual = 64, kMaxWorld = 32, kMaxWeight = 8;
struct ShardMap {
    uint32_t V;                   // virtual ranks = the sum of the weights
    uint32_t w;                   // this rank's weight = the number of virtual ranks it holds
    uint32_t vlist[kMaxWeight];   // ... which ones, ascending
};
// rank 0's view of the same rule, for the unpack kernels: virtual rank -> (owner, position in the owner's list), and where
// every rank's slots start in the staging area
// Unit weights (any world size) use the closed form — owner = tile % world, shard_slot_offset — and leave the arrays unused:
// `unit_world` != 0 says so.  The arrays only ever hold a WEIGHTED rule, which lpt_renderer_set_shard_weighted limits to kMaxWorld ranks.
struct ShardTable {
    uint32_t V;
    uint32_t unit_world;
    uint8_t owner[kMaxVirtual], j[kMaxVirtual];
    uint32_t w[kMaxWorld];
    uint32_t offset[kMaxWorld + 1];   // in slots; offset[world] = all slots
};

struct FrameParams {
    f3 origin, right, up, fwd;
    float ax, ay;
    uint32_t width, height;
    uint32_t user_seed, seed_counter;
    ShardMap map;
    uint32_t rank, world, tile_w, tile_h, tiles_x, n_tiles, n_slots;
    // a wavefront may cover only a run of the rank's slots (device.hip flush_pending: a batch too large for one wavefront is cut
    // spatially): n_slots = the slots of this wavefront, slot0 = the first of them among the rank's slots.  Everything inside a
    // wavefront (queues, Lsum) is indexed by the local slot; only the slot -> pixel map needs slot0
    uint32_t slot0;
    // pixels inside a tile: in 8x8 blocks (block after block along the tile's rows, row-major inside a block) when both tile sides
    // are multiples of 8 — 64 consecutive slots are then a SQUARE patch of the image, the most coherent packet of primary rays
    // (k_trace_packet) — else row-major over the whole tile
    uint32_t block8;
    uint32_t frame_count, max_bounces;
    // batched samples (lpt_renderer_raytrace_n): sample k of the batch behaves like the k-th of n
    // consecutive raytrace() calls: seeds advance by max_bounces per sample, frame_count by fc_inc0
    // after the first sample and by 1 after every later one.  Virtual slot = sample * n_slots + slot.
    uint32_t n_samples, fc_inc0;
};

// streaming accesses (ray queues, hit records): the `nt` hint keeps the ~400 MB that stream through a frame from
// evicting the data that is reused (BVH, shading records, textures) from L2 / Infinity Cache
typedef float f4v __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float4 ld_nt(const float4 *p) { const f4v v = __builtin_nontemporal_load(reinterpret_cast<const f4v *>(p)); return make_float4(v.x, v.y, v.z, v.w); }
__device__ __forceinline__ void st_nt(float4 *p, const float4 v) { const f4v w = {v.x, v.y, v.z, v.w}; __builtin_nontemporal_store(w, reinterpret_cast<f4v *>(p)); }

// pixel slot -> pixel.  Slots enumerate this rank's tiles in ascending tile id (period after period of V tiles, the rank's
// virtual ranks inside a period; with unit weights: tile ids rank, rank+world, ...) and the pixels inside each tile row-major,
// (in 8x8 blocks when the tile sides allow, within_to_xy below), so a wave64 covers an 8x8 pixel block.
__device__ __forceinline__ void within_to_xy(const FrameParams &p, uint32_t within, uint32_t &wx, uint32_t &wy) {
    if (p.block8) {
        const uint32_t block = within >> 6, in = within & 63u, per_row = p.tile_w >> 3;
        const uint32_t by = block / per_row, bx = block - by * per_row;
        wx = bx * 8u + (in & 7u);
        wy = by * 8u + (in >> 3);
    } else {
        wy = within / p.tile_w;
        wx = within - wy * p.tile_w;
    }
}
__device__ __forceinline__ uint32_t xy_to_within(const FrameParams &p, uint32_t wx, uint32_t wy) {
    if (p.block8) return (((wy >> 3) * (p.tile_w >> 3) + (wx >> 3)) << 6) + ((wy & 7u) << 3) + (wx & 7u);
    return wy * p.tile_w + wx;
}
__device__ __forceinline__ bool slot_to_pixel(const FrameParams &p, uint32_t slot, uint32_t &x, uint32_t &y) {
    const uint32_t per_tile = p.tile_w * p.tile_h;
    const uint32_t k = slot / per_tile, within = slot - k * per_tile;
    uint32_t period = k, v = p.map.vlist[0];
    if (p.map.w != 1u) {   // wave-uniform
        period = k / p.map.w;
        const uint32_t j = k - period * p.map.w;
#pragma unroll
        for (uint32_t q = 1; q < kMaxWeight; ++q) v = j == q ? p.map.vlist[q] : v;   // a select chain: the list lives in SGPRs
    }
    const uint32_t tile = period * p.map.V + v;
    if (tile >= p.n_tiles) return false;
    const uint32_t ty = tile / p.tiles_x, tx = tile - ty * p.tiles_x;
    uint32_t wx, wy;
    within_to_xy(p, within, wx, wy);
    x = tx * p.tile_w + wx;
    y = ty * p.tile_h + wy;
    return x < p.width && y < p.height;
}

// wave64 stream compaction: returns this lane's output index (valid lanes only)
__device__ __forceinline__ uint32_t wave_compact(bool valid, uint32_t *counter) {
    const unsigned long long mask = __ballot(valid);
    const uint32_t lane = threadIdx.x & 63u;
    uint32_t base = 0;
    if (lane == 0 && mask) base = atomicAdd(counter, (uint32_t)__popcll(mask));
    base = __shfl(base, 0);
    return base + (uint32_t)__popcll(mask & ((1ull << lane) - 1ull));
}

// block-aggregated stream compaction: ONE global atomic per 256-thread block and counter.
// A single device-scope word sustains only ~88 returning atomics/us on MI355X, so a
// per-wave atomic (32K of them for a 1080p queue) would cost more than the shading itself.
// Must be called by every thread of the block (two barriers).  `lds` needs 8 uint32.
__device__ __forceinline__ uint32_t block_compact(bool valid, uint32_t *counter, uint32_t *lds) {
    const unsigned long long mask = __ballot(valid);
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const uint32_t n = (uint32_t)__popcll(mask);
    if (lane == 0) lds[wave] = n;
    __syncthreads();
    if (threadIdx.x == 0) {
        const uint32_t n0 = lds[0], n1 = lds[1], n2 = lds[2], n3 = lds[3];
        const uint32_t total = n0 + n1 + n2 + n3;
        const uint32_t base = total ? atomicAdd(counter, total) : 0u;
        lds[4] = base; lds[5] = base + n0; lds[6] = base + n0 + n1; lds[7] = base + n0 + n1 + n2;
    }
    __syncthreads();
    const uint32_t idx = lds[4 + wave] + (uint32_t)__popcll(mask & ((1ull << lane) - 1ull));
    __syncthreads();  // lds is reused by the next call
    return idx;
}

// Sorted variant of block_compact (the "sorted shade / next-event stage" of the north star): the block's valid
// elements are written in KEY order (8 keys: the direction octant of the ray), so that the 64 consecutive rays a
// traversal wave pulls share one or two octants instead of eight.  Per wave and key one ballot + popcount, the
// (key, wave) counts are prefix-summed in LDS (key-major), still ONE global atomic per block and counter.
// Results are keyed by pixel slot, so the order of a queue never changes them.  `lds` needs 72 uint32.
__device__ __forceinline__ uint32_t block_compact_binned(bool valid, uint32_t key, uint32_t *counter, uint32_t *lds) {
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    unsigned long long mine = 0ull;
    uint32_t cnt_k = 0;  // lane k < 8 keeps this wave's count of key k
#pragma unroll
    for (uint32_t k = 0; k < 8u; ++k) {
        const unsigned long long m = __ballot(valid && key == k);
        if (key == k) mine = m;
        if (lane == k) cnt_k = (uint32_t)__popcll(m);
    }
    if (lane < 8u) lds[lane * 4u + wave] = cnt_k;   // [key][wave]
    __syncthreads();
    if (threadIdx.x < 32u) {
        // exclusive prefix over the 32 (key, wave) cells in key-major order: one wave32-style shuffle scan
        const uint32_t c = lds[threadIdx.x];
        uint32_t incl = c;
#pragma unroll
        for (int off = 1; off < 32; off <<= 1) {
            const uint32_t v = __shfl_up(incl, off);
            if ((int)lane >= off) incl += v;
        }
        lds[32u + threadIdx.x] = incl - c;
        if (threadIdx.x == 31u) lds[64u] = incl ? atomicAdd(counter, incl) : 0u;
    }
    __syncthreads();
    const uint32_t idx = lds[64u] + lds[32u + key * 4u + wave] + (uint32_t)__popcll(mine & ((1ull << lane) - 1ull));
    __syncthreads();  // lds is reused by the next call
    return idx;
}
__device__ __forceinline__ uint32_t dir_octant(float x, float y, float z) { return (x < 0.0f ? 1u : 0u) | (y < 0.0f ? 2u : 0u) | (z < 0.0f ? 4u : 0u); }

__device__ __forceinline__ void noise_shift(const DNoise &nz, uint32_t x, uint32_t y, uint32_t seed_counter, float &r0, float &r1) {
    if (!nz.enabled) return;
    const uint8_t *t = nz.rgba + 4u * ((size_t)(y % nz.h) * nz.w + (x % nz.w));
    float g = (float)(seed_counter & 1023u) * 0.61803398875f;
    float a = r0 + ((float)t[0] + 0.5f) * 0.00390625f + g;
    float b = r1 + ((float)t[1] + 0.5f) * 0.00390625f + g;
    a = a - floorf(a);
    b = b - floorf(b);
    if (a >= 1.0f) a = 0.0f;
    if (b >= 1.0f) b = 0.0f;
    r0 = a;
    r1 = b;
}

// ------------------------------------------------------------------ ray generation
// DENSE: every slot maps to a pixel (image is a whole number of tiles) -> queue index = slot and
// qcount[0] is simply the slot count; otherwise invalid slots are compacted away.
template <bool DENSE>
__global__ __launch_bounds__(kBlock) void k_raygen(FrameParams p, DNoise nz, Queue q, float4 *Lsum, FrameCounters *ctr) {
    __shared__ uint32_t lds[8];
    const uint32_t stride = gridDim.x * blockDim.x;
    // the tile area is only a multiple of 64: whole blocks stay in the loop for the barriers of block_compact
    const uint32_t total = p.n_slots * p.n_samples;
    const uint32_t rounded = (total + (kBlock - 1u)) & ~(uint32_t)(kBlock - 1u);
    if (DENSE && blockIdx.x == 0 && threadIdx.x == 0) QC(ctr, 0) = total;   // the traversal launch behind this one reads it
    for (uint32_t vslot = blockIdx.x * blockDim.x + threadIdx.x; vslot < rounded; vslot += stride) {
        const uint32_t sample = vslot / p.n_slots, slot = vslot - sample * p.n_slots;
        const uint32_t seed_counter = p.seed_counter + sample * p.max_bounces;
        uint32_t x = 0, y = 0;
        const bool valid = vslot < total && slot_to_pixel(p, p.slot0 + slot, x, y);
        if (vslot < total) Lsum[vslot] = make_float4(0.f, 0.f, 0.f, 0.f);
        f3 d = mk3(0.f, 0.f, 0.f);
        if (valid) {
            const uint32_t pixel = y * p.width + x;
            Rng r = rng_init(pixel, stage_seed(p.user_seed, seed_counter), LPT_TAG_RAYGEN);
            float jx = rng_next(r), jy = rng_next(r);
            noise_shift(nz, x, y, seed_counter, jx, jy);
            float sx = ((float)x + jx) / (float)p.width;
            float sy = ((float)y + jy) / (float)p.height;
            float cx = (2.0f * sx - 1.0f) * p.ax;
            float cy = (1.0f - 2.0f * sy) * p.ay;
            f3 dir = mk3((p.right.x * cx + p.up.x * cy) + p.fwd.x, (p.right.y * cx + p.up.y * cy) + p.fwd.y,
                         (p.right.z * cx + p.up.z * cy) + p.fwd.z);
            d = normalize(dir);
        }
        const uint32_t idx = DENSE ? vslot : block_compact(valid, &QC(ctr, 0), lds);
        if (valid) {
            st_nt(q.o + idx, make_float4(p.origin.x, p.origin.y, p.origin.z, -1.0f));
            st_nt(q.d + idx, make_float4(d.x, d.y, d.z, __uint_as_float(vslot)));
            q.T[idx] = make_float4(1.f, 1.f, 1.f, __uint_as_float(x | (y << 13) | (sample << 26)));  // pixel + sample ride along: no divisions in k_shade
        }
    }
}

// ------------------------------------------------------------------ traversal
struct Hit { float t, u, v; uint32_t prim; };

// SPEC §7 Woop test; accepts tmin(0) < t <= tmax, caller applies the tie rule
__device__ __forceinline__ bool ray_triangle(const float4 r0, const float4 r1, const float4 r2, f3 o, f3 d, float tmax, float &t, float &u, float &v) {
    float oz = fmaf(r2.z, o.z, fmaf(r2.y, o.y, fmaf(r2.x, o.x, r2.w)));
    float dz = fmaf(r2.z, d.z, fmaf(r2.y, d.y, r2.x * d.x));
    float tt = -oz / dz;
    if (!(tt > 0.0f && tt <= tmax)) return false;
    float ox = fmaf(r0.z, o.z, fmaf(r0.y, o.y, fmaf(r0.x, o.x, r0.w)));
    float dx = fmaf(r0.z, d.z, fmaf(r0.y, d.y, r0.x * d.x));
    float uu = fmaf(tt, dx, ox);
    if (!(uu >= 0.0f)) return false;
    float oy = fmaf(r1.z, o.z, fmaf(r1.y, o.y, fmaf(r1.x, o.x, r1.w)));
    float dy = fmaf(r1.z, d.z, fmaf(r1.y, d.y, r1.x * d.x));
    float vv = fmaf(tt, dy, oy);
    if (!(vv >= 0.0f) || !(uu + vv <= 1.0f)) return false;
    t = tt; u = uu; v = vv;
    return true;
}

__device__ __forceinline__ float safe_inv(float d) {
    // a zero component must not produce 0*inf = NaN in the slab test
    return fabsf(d) > 1.0e-30f ? 1.0f / d : copysignf(1.0e30f, d);
}

// One ray against the compressed 8-wide BVH (layout: common.h Node8; traversal after Ylitie,
// Karras & Laine, HPG 2017).  ANY: stop at the first hit in (0, tmax].
//
//  * a "node group" (base index, hit bits << 24 | imask) names the still-unvisited inner
//    children of one node; a "triangle group" (16 * node, 16 hit bits: tg_pop above) its hit triangles.
//    One 8-byte stack entry per tree level, kept in LDS (`stack` = this lane's column,
//    stride kTraceBlock entries).
//  * child boxes are decoded on the fly: t = q * (2^e / d) + (p - o) / d, near / far byte
//    planes picked per axis from the ray's direction signs.  The [tn, tf] interval is widened
//    by a bound on its own rounding error and triangle boxes are padded at build time, so the
//    box tests are conservative; only the Woop test below decides hits (SPEC §7).
//  * children sit in octant-ordered slots: visiting hit bits from the top after XOR-ing the
//    slot with the inverted ray octant is an approximate front-to-back order.
// Per-lane traversal state: one step() = at most one node visit plus one triangle test, so a
// wave can interleave lanes that are in different phases and refill lanes whose ray is done.
struct RayState {
    f3 o, d;
    float ix, iy, iz;
    uint32_t oinv;
    Hit best;
    uint2 ng, tg;
    uint2 tg2;   // ray_step_pipe only: the triangle group found while rs.tg was still being worked off
    int sp;
};

__device__ __forceinline__ void ray_begin(RayState &rs, f3 o, f3 d, float tmax) {
    rs.o = o; rs.d = d;
    rs.ix = safe_inv(d.x); rs.iy = safe_inv(d.y); rs.iz = safe_inv(d.z);
    rs.oinv = 7u - ((rs.ix < 0.0f ? 1u : 0u) | (rs.iy < 0.0f ? 2u : 0u) | (rs.iz < 0.0f ? 4u : 0u));
    rs.best.t = tmax; rs.best.u = 0.f; rs.best.v = 0.f; rs.best.prim = 0xFFFFFFFFu;
    rs.ng = make_uint2(0u, 0x80000000u);  // the root, as the single hit child of a virtual parent
    rs.tg = make_uint2(0u, 0u);
    rs.tg2 = make_uint2(0u, 0u);
    rs.sp = 0;
}

// The node visit of a step, in two halves so that a step can put other work between the fetch and the test (ray_step_pipe).
// node_fetch: takes the next member out of the current node group rs.ng (which must have one), pushes what is left of the group and fetches the node's four rows.
// node_test: tests the eight children against the ray's CURRENT best hit; rs.ng = the node's hit inner children; returns its hit triangles.
struct NodeRows { uint4 n0, n2, n3, n4; uint32_t index; };
template <bool STATS>
__device__ __forceinline__ void node_fetch(const DScene &sc, RayState &rs, uint2 *stack, uint32_t &n_nodes, NodeRows &nr) {
    const uint32_t hits = rs.ng.y;
    const uint32_t bit = 31u - (uint32_t)__clz((int)hits);
    rs.ng.y &= ~(1u << bit);
    if (rs.ng.y & 0xFF000000u) { stack[rs.sp * kTraceBlock] = rs.ng; rs.sp++; }
    const uint32_t slot = (bit - 24u) ^ rs.oinv;
    const uint32_t rel = (uint32_t)__popc(hits & ~(0xFFFFFFFFu << slot));  // inner children before `slot`
    nr.index = rs.ng.x + rel;
    const DNode8 *n = sc.nodes + nr.index;
    nr.n0 = n->n0; nr.n2 = n->n2; nr.n3 = n->n3; nr.n4 = n->n4;
    if (STATS) n_nodes++;
}
// `lut` (LDS, xor_permute8_lut below) or nullptr (the delta swaps)
__device__ __forceinline__ uint2 node_test(const DScene &sc, RayState &rs, const NodeRows &nr, const uint8_t *lut = nullptr) {
    {
        const uint4 n0 = nr.n0, n2 = nr.n2, n3 = nr.n3, n4 = nr.n4;
        const uint32_t node_index = nr.index;
        const NodeGrid g = node_grid(sc, n0);
        const bool negx = rs.ix < 0.0f, negy = rs.iy < 0.0f, negz = rs.iz < 0.0f;
        // t(q) = q * a + b per axis; a is exact (power-of-two step times 1/d), b carries three roundings.
        // |error of the computed t| <= 2^-24 * (4|b| + 510|a|), so widening b by eps = 2^-21 * (|b| + 255|a|)
        // towards the outside on both ends keeps the test conservative wherever the ray starts.
        const float kEps = 4.76837158203125e-7f;  // 2^-21
        const float ax = g.sx * rs.ix, ay = g.sy * rs.iy, az = g.sz * rs.iz;
        const float bx = (g.px - rs.o.x) * rs.ix;
        const float by = (g.py - rs.o.y) * rs.iy;
        const float bz = (g.pz - rs.o.z) * rs.iz;
        const float ex = fmaf(fabsf(ax), 255.0f, fabsf(bx)) * kEps;
        const float ey = fmaf(fabsf(ay), 255.0f, fabsf(by)) * kEps;
        const float ez = fmaf(fabsf(az), 255.0f, fabsf(bz)) * kEps;
        const float bnx = bx - ex, bny = by - ey, bnz = bz - ez;
        const float bfx = bx + ex, bfy = by + ey, bfz = bz + ez;
        const float tbest = rs.best.t;
        uint32_t h = 0u;   // bit s: the ray's interval meets slot s's box
#pragma unroll
        for (int half = 1; half >= 0; --half) {   // slot 7 first: every test shifts the bits found so far up and adds its own (one add with carry per child)
            const uint32_t lox = half ? n2.y : n2.x, loy = half ? n2.w : n2.z, loz = half ? n3.y : n3.x;
            const uint32_t hix = half ? n3.w : n3.z, hiy = half ? n4.y : n4.x, hiz = half ? n4.w : n4.z;
            const uint32_t qnx = negx ? hix : lox, qfx = negx ? lox : hix;
            const uint32_t qny = negy ? hiy : loy, qfy = negy ? loy : hiy;
            const uint32_t qnz = negz ? hiz : loz, qfz = negz ? loz : hiz;
#pragma unroll
            for (int j = 3; j >= 0; --j) {
                const int sh = 8 * j;
                const float tnx = fmaf((float)((qnx >> sh) & 0xFFu), ax, bnx);
                const float tny = fmaf((float)((qny >> sh) & 0xFFu), ay, bny);
                const float tnz = fmaf((float)((qnz >> sh) & 0xFFu), az, bnz);
                const float tfx = fmaf((float)((qfx >> sh) & 0xFFu), ax, bfx);
                const float tfy = fmaf((float)((qfy >> sh) & 0xFFu), ay, bfy);
                const float tfz = fmaf((float)((qfz >> sh) & 0xFFu), az, bfz);
                const float tn = fmaxf(fmaxf(tnx, tny), fmaxf(tnz, 0.0f));
                const float tf = fminf(fminf(tfx, tfy), fminf(tfz, tbest));
                // h = 2 h + (tn <= tf): the compare's lane mask goes in as the carry of ONE add (the compiler's own form is a v_cndmask per child and a v_or3 per two)
                asm("v_cmp_le_f32_e32 vcc, %1, %2\n\tv_addc_co_u32_e32 %0, vcc, %0, %0, vcc" : "+v"(h) : "v"(tn), "v"(tf) : "vcc");
            }
        }
        // empty slots have inverted boxes; should the widening ever let one pass, it is in neither mask below
        const uint32_t imask = node_imask(n0), V = node_leaves(n0);
        // the inner hits into visit order (slot XOR ray octant): two lookups in a 128-byte LDS table where the kernel set one up (k_trace's throughput variant: -2.5 % of
        // the launch against the delta swaps), else three delta swaps (18 VALU instructions, no memory round trip: what a drain of dependent steps wants)
        const uint32_t hin = h & imask;
        const uint32_t hp = lut ? ((uint32_t)lut[(rs.oinv << 4) | (hin & 15u)] | (uint32_t)lut[((rs.oinv ^ 4u) << 4) | (hin >> 4)]) : xor_permute8(hin, rs.oinv);
        rs.ng = make_uint2(n0.w, (hp << 24) | imask);
        return make_uint2(node_index * kNodeTris, (h | (h << 8)) & V);
    }
}

// returns true when the ray is finished (ANY: also as soon as something is hit; best.prim != ~0 then)
template <bool STATS>
__device__ __forceinline__ bool ray_step_any(const DScene &sc, RayState &rs, uint2 *stack, const bool ANY, uint32_t &n_nodes, uint32_t &n_tris, const uint8_t *lut = nullptr) {
    if (rs.tg.y == 0u) {
        if (!(rs.ng.y & 0xFF000000u)) {
            if (rs.sp == 0) return true;
            rs.sp--;
            rs.ng = stack[rs.sp * kTraceBlock];
        }
        NodeRows nr;
        node_fetch<STATS>(sc, rs, stack, n_nodes, nr);
        rs.tg = node_test(sc, rs, nr, lut);
    }
    if (rs.tg.y != 0u) {
        const uint32_t ti = tg_pop(rs.tg);
        const float4 *w = sc.woop + 3u * (size_t)ti;
        const float4 r0 = w[0], r1 = w[1], r2 = w[2];
        if (STATS) n_tris++;
        float t, u, v;
        if (ray_triangle(r0, r1, r2, rs.o, rs.d, rs.best.t, t, u, v)) {
            const uint32_t prim = sc.leaf_prim[ti];
            if (t < rs.best.t || prim < rs.best.prim) { rs.best.t = t; rs.best.u = u; rs.best.v = v; rs.best.prim = prim; }
            // the occluder's leaf slot, for the occluder-cache probe of the stats kernels (an any-hit ray's v is not read).  Round 5: only ray_step_pipe did this,
            // so `--opt pipe_rays=0` with stats on fed the probe a barycentric's bit pattern as a leaf slot: an out-of-bounds read (a GPU memory fault at bench size)
            if (STATS && ANY) rs.best.v = __uint_as_float(ti);
            if (ANY) return true;
        }
    }
    return false;
}

// The same step with ONE memory round trip instead of two: the triangle tested in a step is one that an EARLIER step found, so its
// fetch is issued together with this step's node fetch instead of behind the node test; a lane visits a node in every step in
// which it has one and room for what the visit may find (rs.tg2).  The triangle tests lag the node visits by a step, so the best
// hit shrinks a little later: ~3 % more nodes and ~12 % more triangles are fetched per ray — and the launch is still 1-3 % shorter at
// every size measured.  Round 6: within the step the fetched triangle is tested BEFORE the node's children (its rows are dead by
// then: 72 VGPRs where the other order needed 78-79, and the children are tested against the hit it may have found).
// The host picks the variant per wavefront (device.hip: pipe_rays); the default is this one.
template <bool STATS>
__device__ __forceinline__ bool ray_step_pipe(const DScene &sc, RayState &rs, uint2 *stack, const bool ANY, uint32_t &n_nodes, uint32_t &n_tris, const uint8_t *lut = nullptr) {
    const bool node_work = (rs.ng.y & 0xFF000000u) != 0u || rs.sp != 0;
    const bool tri_work = (rs.tg.y != 0u);      // rs.tg2 is only ever occupied while rs.tg is
    if (!node_work && !tri_work) return true;
    float4 r0 = make_float4(0.f, 0.f, 0.f, 0.f), r1 = r0, r2 = r0;
    uint32_t ti = 0;
    if (tri_work) {
        ti = tg_pop(rs.tg);
        const float4 *w = sc.woop + 3u * (size_t)ti;
        r0 = w[0]; r1 = w[1]; r2 = w[2];
        if (STATS) n_tris++;
    }
    const bool visit = node_work && (rs.tg2.y == 0u);
    NodeRows nr;
    if (visit) {
        if (!(rs.ng.y & 0xFF000000u)) {
            rs.sp--;
            rs.ng = stack[rs.sp * kTraceBlock];
        }
        node_fetch<STATS>(sc, rs, stack, n_nodes, nr);
    }
    // the triangle FIRST (round 6): its rows were requested first and arrive first, a hit shrinks the interval the children below are tested against (fewer of them pass),
    // and its twelve row registers are dead before the child tests — the step's register peak — begin
    if (tri_work) {
        float t, u, v;
        if (ray_triangle(r0, r1, r2, rs.o, rs.d, rs.best.t, t, u, v)) {
            const uint32_t prim = sc.leaf_prim[ti];
            if (t < rs.best.t || prim < rs.best.prim) { rs.best.t = t; rs.best.u = u; rs.best.v = v; rs.best.prim = prim; }
            if (STATS && ANY) rs.best.v = __uint_as_float(ti);   // the occluder's leaf slot, for the occluder-cache probe (an any-hit ray's v is not read)
            if (ANY) return true;
        }
    }
    uint2 found = make_uint2(0u, 0u);
    if (visit) found = node_test(sc, rs, nr, lut);
    if (rs.tg.y == 0u) { rs.tg = rs.tg2; rs.tg2.y = 0u; }
    if (found.y != 0u) { if (rs.tg.y == 0u) rs.tg = found; else rs.tg2 = found; }
    return false;
}

template <bool ANY, bool STATS>
__device__ __forceinline__ bool ray_step(const DScene &sc, RayState &rs, uint2 *stack, uint32_t &n_nodes, uint32_t &n_tris) {
    return ray_step_any<STATS>(sc, rs, stack, ANY, n_nodes, n_tris);
}

// one ray start to finish (stand-alone queries)
template <bool ANY, bool STATS>
__device__ __forceinline__ bool traverse(const DScene &sc, f3 o, f3 d, float tmax, uint2 *stack, Hit &best, uint32_t &n_nodes, uint32_t &n_tris) {
    RayState rs;
    ray_begin(rs, o, d, tmax);
    while (!ray_step<ANY, STATS>(sc, rs, stack, n_nodes, n_tris)) {}
    best = rs.best;
    return best.prim != 0xFFFFFFFFu;
}

// Work distribution of the persistent traversal kernels.  The queue is cut into chunks of
// `chunk` rays (64..256); chunk c belongs to head c % 8 and a wave only ever pulls from head
// blockIdx % 8 (block b runs on XCD b % 8 in practice, so each XCD drains its own head; the
// mapping only matters for speed).  A pull is ONE returning atomic by lane 0 per chunk — about
// 1K atomics per head for a 2M-ray queue, far below the ~88 atomics/us a single word sustains.
struct ChunkPuller {
    uint32_t *head;
    uint32_t count, chunk, n_chunks, home;
    uint32_t next, end;
    bool dry;
};
__device__ __forceinline__ void puller_init(ChunkPuller &p, uint32_t *heads8, uint32_t count) {
    p.home = blockIdx.x & 7u;
    p.head = heads8 + p.home * 32u;
    p.count = count;
    const uint32_t per_wave = count / (2u * 64u * max(gridDim.x, 1u));  // aim at >= 2 chunks per wave
    p.chunk = 64u * min(max(per_wave, 1u), 4u);
    p.n_chunks = (count + p.chunk - 1u) / p.chunk;
    p.next = p.end = 0u;
    p.dry = false;
}
// wave-uniform: make [next,end) non-empty if any chunk is left for this wave's head
// OWN_FIRST (launches whose whole grid is resident at once: a solo wavefront, k_trace<.., TAIL>): a wave's FIRST chunk is its own (block h + 8 k takes chunk
// h + 8 k) without an atomic — at the start of such a launch every wave pulls at once, 768 atomics on each head's word, ~9 us until the last wave is served at the
// ~88 returning atomics a word sustains per microsecond; the head then counts the pulls behind that round.  Not where blocks become resident over time (two
// wavefronts sharing the chip: a block that starts late would sit on its chunk until then — whole frame 11.61 -> 11.87 ms, round 5).
template <bool OWN_FIRST = false>
__device__ __forceinline__ void puller_pull(ChunkPuller &p) {
    if (p.next < p.end || p.dry) return;
    uint32_t k;
    if (OWN_FIRST && p.end == 0u) k = blockIdx.x >> 3;
    else {
        uint32_t t = 0;
        if ((threadIdx.x & 63u) == 0) t = atomicAdd(p.head, 1u);
        k = (uint32_t)__builtin_amdgcn_readfirstlane((int)t) + (OWN_FIRST ? (gridDim.x - p.home + 7u) / 8u : 0u);
    }
    const uint32_t c = p.home + 8u * k;
    if (c >= p.n_chunks) { p.dry = true; return; }
    p.next = c * p.chunk;
    p.end = min(p.count, p.next + p.chunk);
}

// SPEC §8: rectangular emitters (front face only, strictly closer than any triangle)
__device__ __forceinline__ void intersect_lights(const DScene &sc, f3 o, f3 d, Hit &best) {
    for (uint32_t l = 0; l < sc.n_lights; ++l) {
        const float4 *L = reinterpret_cast<const float4 *>(sc.lights + l);
        const float4 n4 = L[0], t4 = L[1], b4 = L[2], o4 = L[3];
        const f3 nl = mk3(n4.x, n4.y, n4.z);
        float dn = dot(d, nl);
        if (!(dn < 0.0f)) continue;
        const f3 ctr = mk3(o4.x, o4.y, o4.z);
        float t = dot(ctr - o, nl) / dn;
        if (!(t > 0.0f && t < best.t)) continue;
        f3 p = mk3(fmaf(d.x, t, o.x), fmaf(d.y, t, o.y), fmaf(d.z, t, o.z));
        f3 r = p - ctr;
        float a = dot(r, mk3(t4.x, t4.y, t4.z));
        float b = dot(r, mk3(b4.x, b4.y, b4.z));
        if (fabsf(a) <= t4.w && fabsf(b) <= b4.w) { best.t = t; best.u = a; best.v = b; best.prim = LPT_LIGHT_BIT | l; }
    }
}

// dynamic LDS: sc.stack_entries * kTraceBlock uint2 (16-byte aligned, Guideline 17)
extern __shared__ __attribute__((aligned(16))) unsigned char lds_dyn[];

// Occluder-cache PROBE (STATS variants only): a table of the last occluding triangle per cell of a grid over the shadow
// rays' origins.  The probe asks, for every finished shadow ray, what the cache would have answered at its start — is there an entry,
// does that triangle occlude this ray (Woop test) — and then records the ray's own occluder.  Nothing about the traversal changes.
struct OccProbe { uint32_t *table; uint32_t mask; float inv_cell; };
__device__ __forceinline__ uint32_t occ_key(const OccProbe &oc, f3 o) {
    const int x = (int)floorf(o.x * oc.inv_cell), y = (int)floorf(o.y * oc.inv_cell), z = (int)floorf(o.z * oc.inv_cell);
    return pcg_hash((uint32_t)x * 73856093u ^ (uint32_t)y * 19349663u ^ (uint32_t)z * 83492791u) & oc.mask;
}

// Closest-hit rays of bounce `cb` and shadow rays of bounce `sb` in ONE persistent launch (either may be
// absent: -1).  Both queues were filled by the same shading pass; tracing them together halves the number of
// traversal launches per frame and lets the short shadow rays fill the lanes that the tail of the closest-hit
// queue leaves idle.  A wave drains closest-hit chunks first, then shadow chunks; a lane remembers which kind
// of ray it carries.  The only state the two kinds share is Lsum, which only the shadow part touches.
// PIPE: ray_step_pipe (one memory round trip per step) instead of ray_step_any — same results, for launches of few rays.
// ---- one ray, a whole wave (k_trace_coop, and the tail of k_trace) ----------------------------------------------------------------------------------
// Eight lanes per node (lane j of a group tests child j), up to eight pending nodes of the ray per round.  The node stack (node indices) is one LDS
// column per wave.  A lane tests the triangles of the leaf child it found; a round's candidates are merged by the rule of the per-lane traversal — the
// smallest t, ties to the lower primitive id — through a 64-bit key, so the hit (t, u, v, prim) is the one the per-lane steps find: the slab tests are
// conservative on both sides and only the Woop test decides (SPEC §7).
// A round pops m <= 8 nodes and pushes at most 8 m.  Up to `cap` entries the walk is as broad as it can be; beyond, m = 1: depth first, which adds at
// most 7 entries per level below the node it pops — the host sizes the LDS column for cap + 8 + 7 * (tree depth + 1) entries (coop_stack_entries), so the
// column cannot overflow whatever the ray's frontier looks like.
constexpr uint32_t kCoopStack = 384u;   // k_trace_coop: a straggler's walk starts at the root and may be broad
constexpr uint32_t kTailStack = 64u;    // k_trace's tail: the walk starts from a lane's depth-first frontier (at most 7 * depth + 1 nodes, below the column's size either way)
__host__ __device__ __forceinline__ uint32_t coop_stack_entries(uint32_t cap, uint32_t tree_depth) { return cap + 8u + 7u * (tree_depth + 1u); }

// a round's candidates (at most one per lane) into the ray's best hit; every lane ends with the same `best`
__device__ __forceinline__ void coop_merge(Hit &best, const bool cand, const float ct, const float cu, const float cv, const uint32_t cprim) {
    // the wave's smallest (t, prim): t > 0, so its bit pattern orders like its value
    unsigned long long key = cand ? (((unsigned long long)__float_as_uint(ct) << 32) | cprim) : ~0ull;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const unsigned long long other = __shfl_xor(key, off);
        key = other < key ? other : key;
    }
    const unsigned long long old = ((unsigned long long)__float_as_uint(best.t) << 32) | best.prim;
    if (key < old) {
        const int src = __ffsll((long long)__ballot(cand && ((((unsigned long long)__float_as_uint(ct) << 32) | cprim) == key))) - 1;
        best.t = __shfl(ct, src); best.u = __shfl(cu, src); best.v = __shfl(cv, src); best.prim = __shfl(cprim, src);
    }
}

// the rounds: `count` node indices wait in stk[0..count); all arguments but the lane's own role are wave-uniform.  Shadow rays: best.prim = 0 as soon as anything is hit
template <bool STATS>
__device__ __forceinline__ void coop_walk(const DScene &sc, const f3 o, const f3 d, const float ix, const float iy, const float iz, const bool shadow, Hit &best,
                                          uint32_t *stk, uint32_t count, const uint32_t cap, const uint32_t lane, uint32_t &n_nodes, uint32_t &n_tris) {
    const bool negx = ix < 0.0f, negy = iy < 0.0f, negz = iz < 0.0f;
    // NEAREST ON TOP: lane j of a group tests the child the ray enters j-th LAST (slot = j ^ oinv, the visit order of the lane-per-ray steps: node_fetch pops the highest bit),
    // and the group with the round's top node is the last one — so the pushes (lane order) leave the nearest child of the nearest node on top of the column
    const uint32_t grp = lane >> 3, c = (lane & 7u) ^ (7u - ((negx ? 1u : 0u) | (negy ? 2u : 0u) | (negz ? 4u : 0u)));
    bool done = false;
    while (count && !done) {
        const uint32_t room = count < cap ? cap - count : 0u;
        const uint32_t m = min(min(count, 8u), max(room / 7u, 1u));
        const bool work = grp < m;
        uint32_t node = 0u;
        if (work) node = stk[count - m + grp];
        __syncthreads();            // every pop is read before the pushes below overwrite the slots
        count -= m;
        bool hit_inner = false, hit_leaf = false;
        uint32_t child = 0u, first = 0u, bits = 0u;
        if (work) {
            const DNode8 *n = sc.nodes + node;
            const uint4 n0 = n->n0, n2 = n->n2, n3 = n->n3, n4 = n->n4;
            const float kEps = 4.76837158203125e-7f;  // 2^-21, as in node_visit
            const NodeGrid g = node_grid(sc, n0);
            const float ax = g.sx * ix, ay = g.sy * iy, az = g.sz * iz;
            const float bx = (g.px - o.x) * ix;
            const float by = (g.py - o.y) * iy;
            const float bz = (g.pz - o.z) * iz;
            const float ex = fmaf(fabsf(ax), 255.0f, fabsf(bx)) * kEps;
            const float ey = fmaf(fabsf(ay), 255.0f, fabsf(by)) * kEps;
            const float ez = fmaf(fabsf(az), 255.0f, fabsf(bz)) * kEps;
            const uint32_t sh = 8u * (c & 3u);
            const bool hi4 = c >= 4u;
            const uint32_t lox = ((hi4 ? n2.y : n2.x) >> sh) & 0xFFu, loy = ((hi4 ? n2.w : n2.z) >> sh) & 0xFFu, loz = ((hi4 ? n3.y : n3.x) >> sh) & 0xFFu;
            const uint32_t hix = ((hi4 ? n3.w : n3.z) >> sh) & 0xFFu, hiy = ((hi4 ? n4.y : n4.x) >> sh) & 0xFFu, hiz = ((hi4 ? n4.w : n4.z) >> sh) & 0xFFu;
            const float tnx = fmaf((float)(negx ? hix : lox), ax, bx - ex), tfx = fmaf((float)(negx ? lox : hix), ax, bx + ex);
            const float tny = fmaf((float)(negy ? hiy : loy), ay, by - ey), tfy = fmaf((float)(negy ? loy : hiy), ay, by + ey);
            const float tnz = fmaf((float)(negz ? hiz : loz), az, bz - ez), tfz = fmaf((float)(negz ? loz : hiz), az, bz + ez);
            const float tn = fmaxf(fmaxf(tnx, tny), fmaxf(tnz, 0.0f));
            const float tf = fminf(fminf(tfx, tfy), fminf(tfz, best.t));
            const uint32_t imask = node_imask(n0), V = node_leaves(n0);
            const bool hit = tn <= tf;     // empty slots (inverted boxes) are in neither mask
            hit_inner = hit && ((imask >> c) & 1u) != 0u;
            hit_leaf = hit && ((V >> c) & 1u) != 0u;
            child = n0.w + (uint32_t)__popc(imask & ~(0xFFFFFFFFu << c));
            first = node * kNodeTris + 2u * c;
            bits = leaf_count(V, c);
        }
        if (STATS) n_nodes += m;
        // inner children: appended to the stack (ballot + prefix count)
        const unsigned long long im = __ballot(hit_inner);
        if (hit_inner) stk[count + (uint32_t)__popcll(im & ((1ull << lane) - 1ull))] = child;
        count += (uint32_t)__popcll(im);
        // leaf children: the lane tests its (one or two) triangles against the round's best
        float ct = 0.f, cu = 0.f, cv = 0.f;
        uint32_t cprim = 0xFFFFFFFFu;
        bool cand = false;
        if (hit_leaf) {
            for (uint32_t k = 0; k < bits; ++k) {
                const uint32_t ti = first + k;
                const float4 *w = sc.woop + 3u * (size_t)ti;
                const float4 r0 = w[0], r1 = w[1], r2 = w[2];
                float t, u, v;
                if (ray_triangle(r0, r1, r2, o, d, cand ? ct : best.t, t, u, v)) {
                    const uint32_t prim = sc.leaf_prim[ti];
                    if (!cand || t < ct || prim < cprim) { ct = t; cu = u; cv = v; cprim = prim; cand = true; }
                }
            }
        }
        if (STATS) n_tris += (uint32_t)__popcll(__ballot(hit_leaf));
        if (__ballot(cand)) {
            if (shadow) { best.prim = 0u; done = true; }   // any hit in (0, tmax] occludes
            else coop_merge(best, cand, ct, cu, cv, cprim);
        }
        __syncthreads();            // the pushes are visible to the next round's pops
    }
}

// THE TAIL OF A TRAVERSAL LAUNCH, IN PLACE.  A launch lasts as long as its last ray, and a lane takes one dependent step (2-3 us with every wave resident) per node.
// A wave whose queues are dry and that is down to `tail` live rays or fewer stops stepping them lane by lane and finishes them one after the other with all 64
// lanes, each from where its lane stands: the ray's frontier — the node groups on the lane's LDS stack and the one in hand, expanded to node indices — seeds the
// cooperative walk above, the triangle groups the lane still holds are tested first, the best hit so far carries over.  Nothing is restarted at the root and
// no second launch is needed (the step budget + k_trace_coop pair this replaces: 48 steps thrown away per straggler, ~19 us of launch per bounce).
// `columns` = the wave's per-lane stacks (entry e of lane L at columns[e * kTraceBlock + L]); `tail_lds` = tail_lds_words(depth) words behind them: the live
// rays' states first (parked there so that the walk behind the loop starts from LDS, not from the loop's registers: 74 VGPRs, 6 waves per SIMD), then the node column.
constexpr uint32_t kTailMax = 8u;          // rays a wave finishes this way at most (one after the other: beyond a handful the per-lane steps are faster)
constexpr uint32_t kTailStateWords = 20u;  // five 16-byte rows per ray
__host__ __device__ __forceinline__ uint32_t tail_lds_words(uint32_t tree_depth) { return kTailMax * kTailStateWords + coop_stack_entries(kTailStack, tree_depth); }
// the lane's part, inside the traversal loop: park the live rays' states; returns their number
template <bool PIPE>
__device__ __forceinline__ uint32_t tail_park(const RayState &rs, const unsigned long long live, const bool lane_active, const bool lane_shadow, const uint32_t lane_ray, uint32_t *tail_lds) {
    uint32_t lane = threadIdx.x;
    asm volatile("" : "+v"(lane));   // opaque: what is derived from it is computed HERE, not hoisted in front of the traversal loop (where it would cost the loop registers)
    if (lane_active) {
        uint4 *st = reinterpret_cast<uint4 *>(tail_lds) + 5u * (uint32_t)__popcll(live & ((1ull << lane) - 1ull));
        st[0] = make_uint4(__float_as_uint(rs.o.x), __float_as_uint(rs.o.y), __float_as_uint(rs.o.z), __float_as_uint(rs.d.x));
        st[1] = make_uint4(__float_as_uint(rs.d.y), __float_as_uint(rs.d.z), __float_as_uint(rs.best.t), __float_as_uint(rs.best.u));
        st[2] = make_uint4(__float_as_uint(rs.best.v), rs.best.prim, rs.ng.x, rs.ng.y);
        st[3] = make_uint4(rs.tg.x, rs.tg.y, PIPE ? rs.tg2.x : 0u, PIPE ? rs.tg2.y : 0u);
        st[4] = make_uint4((uint32_t)rs.sp | (lane_shadow ? 0x80000000u : 0u), lane_ray, lane, 0u);
    }
    return (uint32_t)__popcll(live);
}
// the wave's part, behind the traversal loop (its registers are not the loop's)
__device__ __forceinline__ void tail_walk(const DScene &sc, const uint32_t n_live, float4 *hits, const float4 *sq_c, float4 *Lsum) {
    static_assert(kTraceBlock == 64, "one wave per block: the cooperative stack is the wave's");
    uint32_t lane = threadIdx.x;
    asm volatile("" : "+v"(lane));   // as in tail_park
    const uint2 *columns = reinterpret_cast<const uint2 *>(lds_dyn);
    uint32_t *tail_lds = reinterpret_cast<uint32_t *>(lds_dyn + sc.stack_entries * kTraceBlock * sizeof(uint2));
    const uint4 *state = reinterpret_cast<const uint4 *>(tail_lds);
    uint32_t *stk = tail_lds + kTailMax * kTailStateWords;
    __syncthreads();
    for (uint32_t r = 0; r < n_live; ++r) {
        const uint4 s0 = state[5u * r], s1 = state[5u * r + 1u], s2 = state[5u * r + 2u], s3 = state[5u * r + 3u], s4 = state[5u * r + 4u];
        const f3 o = mk3(__uint_as_float(s0.x), __uint_as_float(s0.y), __uint_as_float(s0.z)), d = mk3(__uint_as_float(s0.w), __uint_as_float(s1.x), __uint_as_float(s1.y));
        const float ix = safe_inv(d.x), iy = safe_inv(d.y), iz = safe_inv(d.z);          // as ray_begin
        const uint32_t oinv = 7u - ((ix < 0.0f ? 1u : 0u) | (iy < 0.0f ? 2u : 0u) | (iz < 0.0f ? 4u : 0u));
        Hit best;
        best.t = __uint_as_float(s1.z); best.u = __uint_as_float(s1.w); best.v = __uint_as_float(s2.x); best.prim = s2.y;
        const uint32_t slot_bits = s1.w;   // a shadow ray's pixel slot rides in `u` (k_trace)
        const uint2 ng = make_uint2(s2.z, s2.w);
        const bool shadow = (s4.x >> 31) != 0u;
        const int sp = (int)(s4.x & 0x7FFFFFFFu);
        const uint32_t ray = s4.y, L = s4.z;
        // the frontier: groups 0..sp-1 of the lane's column, then the group in hand (the deepest on top, and within a group the bit the lane would take next)
        uint32_t count = 0u;
        for (int base = 0; base <= sp; base += 8) {
            const int e = base + (int)(lane >> 3);
            uint2 g = make_uint2(0u, 0u);
            if (e < sp) g = columns[e * kTraceBlock + L];
            else if (e == sp) g = ng;
            const uint32_t bit = lane & 7u;
            const bool has = ((g.y >> (24u + bit)) & 1u) != 0u;
            const uint32_t slot = bit ^ oinv;
            const uint32_t node = g.x + (uint32_t)__popc(g.y & ~(0xFFFFFFFFu << slot));   // node_visit's `rel`: the inner children before the slot
            const unsigned long long hm = __ballot(has);
            if (has) stk[count + (uint32_t)__popcll(hm & ((1ull << lane) - 1ull))] = node;
            count += (uint32_t)__popcll(hm);
        }
        // the triangles the lane had found and not yet tested: lanes 0..15 the group in work, 16..31 the one waiting behind it (one-round-trip step)
        bool done = false;
        {
            const uint32_t k = lane & 15u;
            const uint2 g = lane < 16u ? make_uint2(s3.x, s3.y) : make_uint2(s3.z, s3.w);
            const bool mine = lane < 32u && ((g.y >> k) & 1u) != 0u;
            float ct = 0.f, cu = 0.f, cv = 0.f;
            uint32_t cprim = 0xFFFFFFFFu;
            bool cand = false;
            if (mine) {
                const uint32_t ti = tg_index(g.x, k);
                const float4 *w = sc.woop + 3u * (size_t)ti;
                const float4 r0 = w[0], r1 = w[1], r2 = w[2];
                if (ray_triangle(r0, r1, r2, o, d, best.t, ct, cu, cv)) { cprim = sc.leaf_prim[ti]; cand = true; }
            }
            if (__ballot(cand)) {
                if (shadow) { best.prim = 0u; done = true; }
                else coop_merge(best, cand, ct, cu, cv, cprim);
            }
        }
        __syncthreads();            // the frontier is in the column before the first round pops it
        uint32_t nn = 0, nt = 0;
        if (!done) coop_walk<false>(sc, o, d, ix, iy, iz, shadow, best, stk, count, kTailStack, lane, nn, nt);
        if (shadow) {
            if (lane == 0 && best.prim == 0xFFFFFFFFu) {   // unoccluded: deposit the light sample
                const float4 cc = sq_c[ray];
                float4 Lp = Lsum[slot_bits];
                Lp.x = Lp.x + cc.x; Lp.y = Lp.y + cc.y; Lp.z = Lp.z + cc.z;
                Lsum[slot_bits] = Lp;
            }
        } else {
            intersect_lights(sc, o, d, best);
            if (lane == 0) st_nt(hits + ray, make_float4(best.t, best.u, best.v, __uint_as_float(best.prim)));
        }
        __syncthreads();            // the column is reused by the next ray
    }
}

// TAIL: the variant whose waves finish their last rays cooperatively (tail_walk); kept to the 6 waves per SIMD of the one-round-trip step (without the attribute the
// compiler takes the extra code as a licence for 90-100 VGPRs in the loop).  The other variants are compiled as before.
template <bool STATS, bool PIPE = false, bool TAIL = false>
__global__ __launch_bounds__(kTraceBlock) __attribute__((amdgpu_waves_per_eu(TAIL ? 6 : 1))) void k_trace(DScene sc, Queue q, float4 *hits, ShadowQueue sq, float4 *Lsum, FrameCounters *ctr,
                                                       int cb, int sb, int refill, OccProbe occ, uint32_t budget, uint32_t *strag, int launch, uint32_t tail) {
    uint2 *stack = reinterpret_cast<uint2 *>(lds_dyn) + threadIdx.x;
    ChunkPuller pc, ps;
    puller_init(pc, &ctr->ihead[(cb < 0 ? 0 : cb) * 8 * 32], cb < 0 ? 0u : QC(ctr, cb));
    puller_init(ps, &ctr->shead[(sb < 0 ? 0 : sb) * 8 * 32], sb < 0 ? 0u : SC(ctr, sb));
    if (cb < 0) pc.dry = true;
    if (sb < 0) ps.dry = true;
    const uint32_t lane = threadIdx.x;
    uint32_t n_nodes = 0, n_tris = 0, s_nodes = 0, s_tris = 0;
    uint32_t w_steps = 0, w_live = 0, w_node = 0, w_tri = 0;  // wave-uniform utilisation counters (STATS)
    uint32_t n_occluded = 0, n_found = 0, n_would = 0;        // occluder-cache probe (STATS)
    uint32_t my_steps = 0;                                    // traversal steps of the ray in hand (STATS; the step budget)
    uint32_t *s_hist = reinterpret_cast<uint32_t *>(lds_dyn + sc.stack_entries * kTraceBlock * sizeof(uint2));   // STATS: 12 buckets + the maximum, behind the stacks (host: + 64 B)
    if (STATS) { if (threadIdx.x < 16u) s_hist[threadIdx.x] = 0u; __syncthreads(); }
    const uint8_t *lut = nullptr;
    // the 8-bit XOR permutation of node_test as two table lookups: perm[o][nibble] = the low four slots' bits moved to slot ^ o (the high four: perm[o ^ 4]).  Not in
    // the TAIL variant: a shard-sized launch is a drain of dependent steps, and two LDS round trips per step cost it more than the 18 VALU instructions they replace
    if (!TAIL) {
        __shared__ uint8_t s_perm[128];
        for (uint32_t e = threadIdx.x; e < 128u; e += kTraceBlock) {
            const uint32_t o = e >> 4, nib = e & 15u;
            uint32_t v = 0u;
            for (uint32_t sl = 0; sl < 4u; ++sl) if ((nib >> sl) & 1u) v |= 1u << (sl ^ o);
            s_perm[e] = (uint8_t)v;
        }
        __syncthreads();
        lut = s_perm;
    }
    RayState rs;
    bool active = false, finished = false, shadow = false;
    uint32_t ray = 0;
    uint32_t tail_live = 0u;   // TAIL: rays parked for the cooperative walk behind the loop
    for (;;) {
        const unsigned long long amask = __ballot(active);
        const int n_active = __popcll(amask);
        if (n_active <= refill) {
            // results are written here, together with the refill, so that the emitter test / the deposit and
            // the stores run for a batch of lanes instead of once per finishing lane
            if (finished) {
                if (STATS) {   // the wave's own histogram in LDS (a global atomic per ray on a dozen hot words would take longer than the launch itself)
                    atomicMax(&s_hist[12], my_steps);
                    atomicAdd(&s_hist[min(11, 31 - __clz((int)max(my_steps, 1u)))], 1u);
                    my_steps = 0;
                }
                if (shadow) {
                    if (STATS && occ.table) {
                        const uint32_t key = occ_key(occ, rs.o);
                        const uint32_t cached = occ.table[key];
                        const bool occluded = rs.best.prim != 0xFFFFFFFFu;
                        bool would_hit = false;
                        if (cached) {
                            const float4 *w = sc.woop + 3u * (size_t)(cached - 1u);
                            float t, u, v;
                            would_hit = ray_triangle(w[0], w[1], w[2], rs.o, rs.d, sq.o[ray].w, t, u, v);
                        }
                        if (occluded) occ.table[key] = __float_as_uint(rs.best.v) + 1u;
                        n_occluded += occluded ? 1u : 0u; n_found += cached ? 1u : 0u; n_would += would_hit ? 1u : 0u;
                    }
                    if (rs.best.prim == 0xFFFFFFFFu) {  // unoccluded: deposit the light sample
                        const uint32_t slot = __float_as_uint(rs.best.u);   // the pixel slot rides in the unused `u` of an any-hit ray
                        const float4 c = sq.c[ray];
                        float4 L = Lsum[slot];
                        L.x = L.x + c.x; L.y = L.y + c.y; L.z = L.z + c.z;
                        Lsum[slot] = L;
                    }
                } else {
                    intersect_lights(sc, rs.o, rs.d, rs.best);
                    st_nt(hits + ray, make_float4(rs.best.t, rs.best.u, rs.best.v, __uint_as_float(rs.best.prim)));
                }
                finished = false;
            }
            puller_pull<TAIL>(pc);
            const bool use_s = !(pc.next < pc.end);  // wave-uniform
            if (use_s) puller_pull<TAIL>(ps);
            const uint32_t nx = use_s ? ps.next : pc.next, en = use_s ? ps.end : pc.end;
            if (nx < en) {
                const uint32_t idx = nx + (uint32_t)__popcll(~amask & ((1ull << lane) - 1ull));
                if (!active && idx < en) {
                    float4 o4, d4;
                    if (use_s) { o4 = sq.o[idx]; d4 = sq.d[idx]; } else { o4 = q.o[idx]; d4 = q.d[idx]; }
                    ray_begin(rs, mk3(o4.x, o4.y, o4.z), mk3(d4.x, d4.y, d4.z), use_s ? o4.w : LPT_T_INF);
                    my_steps = 0;
                    if (use_s) rs.best.u = d4.w;   // kept for the deposit; a hit overwrites it, and then nothing is deposited
                    ray = idx;
                    shadow = use_s;
                    active = true;
                }
                const uint32_t adv = min(en, nx + (uint32_t)(64 - n_active));
                if (use_s) ps.next = adv; else pc.next = adv;
            } else if (n_active == 0) break;
            else if (TAIL && n_active <= (int)tail) {
                // both queues are dry and few rays are left: the wave finishes them cooperatively, in place (tail_park / tail_walk; the host sets `tail` only without the stats)
                tail_live = tail_park<PIPE>(rs, amask, active, shadow, ray, reinterpret_cast<uint32_t *>(lds_dyn + sc.stack_entries * kTraceBlock * sizeof(uint2)));
                break;
            }
        }
        uint32_t dn = 0, dt = 0;
        if (STATS) {
            w_steps++;
            w_live += (uint32_t)__popcll(__ballot(active));
            w_node += (uint32_t)__popcll(__ballot(PIPE ? active && (rs.tg2.y == 0u) && ((rs.ng.y & 0xFF000000u) != 0u || rs.sp != 0) : active && (rs.tg.y == 0u)));
        }
        if ((STATS || budget) && active) my_steps++;
        if (active && (PIPE ? ray_step_pipe<STATS>(sc, rs, stack, shadow, dn, dt, lut) : ray_step_any<STATS>(sc, rs, stack, shadow, dn, dt, lut))) {
            active = false;
            finished = true;
        }
        // STEP BUDGET: one ray in 10^4..10^5 needs 64..360 steps, and a launch lasts as long as its longest ray.  A ray that has used
        // its budget is dropped here — unfinished, its lane free for the next ray — and listed for k_trace_coop, which traces it again with a whole
        // wave (eight lanes per node, every pending node of the ray in one round).  Hits are decided by the Woop test alone, so the result is the same.
        if (budget && active && my_steps >= budget) {
            strag[atomicAdd(&ctr->strag_count[launch], 1u)] = ray | (shadow ? 0x80000000u : 0u);
            active = false;
            if (!STATS) my_steps = 0;
        }
        if (STATS) {
            w_tri += (uint32_t)__popcll(__ballot(dt != 0u));
            if (shadow) { s_nodes += dn; s_tris += dt; } else { n_nodes += dn; n_tris += dt; }
        }
    }
    if (TAIL && tail_live) {
        if (lane == 0) atomicAdd(&ctr->tail_rays[(blockIdx.x % kTailCounters) * 32u], tail_live);
        tail_walk(sc, tail_live, hits, sq.c, Lsum);
    }
    if (STATS) {
        atomicAdd(&ctr->nodes, (unsigned long long)n_nodes);
        atomicAdd(&ctr->tris, (unsigned long long)n_tris);
        atomicAdd(&ctr->shadow_nodes, (unsigned long long)s_nodes);
        atomicAdd(&ctr->shadow_tris, (unsigned long long)s_tris);
        __syncthreads();
        if (lane < 12u && s_hist[lane]) atomicAdd(&ctr->step_hist[lane], s_hist[lane]);
        if (lane == 12u) atomicMax(&ctr->max_steps, s_hist[12]);
        if (occ.table) {
            atomicAdd(&ctr->shadow_occluded, (unsigned long long)n_occluded);
            atomicAdd(&ctr->occ_found, (unsigned long long)n_found);
            atomicAdd(&ctr->occ_hits, (unsigned long long)n_would);
        }
        if (lane == 0) {
            atomicAdd(&ctr->wave_steps, (unsigned long long)w_steps);
            atomicAdd(&ctr->live_lanes, (unsigned long long)w_live);
            atomicAdd(&ctr->node_lanes, (unsigned long long)w_node);
            atomicAdd(&ctr->tri_lanes, (unsigned long long)w_tri);
        }
    }
}

// The stragglers of a traversal launch (k_trace with a step budget) — or every ray of a tiny wavefront —, each traced from the root by a WHOLE WAVE (coop_walk
// above): a ray that would take a lane 100-360 dependent steps takes the wave 15-40 rounds.  Closest-hit rays of queue `cb`, shadow rays of queue `sb`.
template <bool STATS>
__global__ __launch_bounds__(kTraceBlock) void k_trace_coop(DScene sc, Queue q, float4 *hits, ShadowQueue sq, float4 *Lsum, FrameCounters *ctr,
                                                            int cb, int sb, const uint32_t *strag, int launch) {
    static_assert(kTraceBlock == 64, "one wave per block: the LDS stack is the wave's");
    uint32_t *stk = reinterpret_cast<uint32_t *>(lds_dyn);   // coop_stack_entries(kCoopStack, depth) node indices (host)
    const uint32_t lane = threadIdx.x;
    // strag == nullptr: EVERY ray of the two queues, closest-hit rays first — a wavefront with fewer rays than the chip has wave slots (LPT_OPT_COOP_RAYS): a lane
    // per ray would leave the chip empty and make the frame one long chain of dependent steps; a wave per ray shortens the chain several times over
    const uint32_t n_closest = strag ? 0u : (cb < 0 ? 0u : QC(ctr, cb));
    const uint32_t n_strag = strag ? ctr->strag_count[launch] : n_closest + (sb < 0 ? 0u : SC(ctr, sb));
    uint32_t n_nodes = 0, n_tris = 0, s_nodes = 0, s_tris = 0;
    for (uint32_t si = blockIdx.x; si < n_strag; si += gridDim.x) {
        const uint32_t e = strag ? strag[si] : (si < n_closest ? si : ((si - n_closest) | 0x80000000u));
        const bool shadow = (e >> 31) != 0u;
        const uint32_t ray = e & 0x7FFFFFFFu;
        const float4 o4 = shadow ? sq.o[ray] : q.o[ray], d4 = shadow ? sq.d[ray] : q.d[ray];
        const f3 o = mk3(o4.x, o4.y, o4.z), d = mk3(d4.x, d4.y, d4.z);
        Hit best;
        best.t = shadow ? o4.w : LPT_T_INF; best.u = 0.f; best.v = 0.f; best.prim = 0xFFFFFFFFu;
        if (lane == 0) stk[0] = 0u;     // the root
        __syncthreads();
        if (shadow) coop_walk<STATS>(sc, o, d, safe_inv(d.x), safe_inv(d.y), safe_inv(d.z), true, best, stk, 1u, kCoopStack, lane, s_nodes, s_tris);
        else coop_walk<STATS>(sc, o, d, safe_inv(d.x), safe_inv(d.y), safe_inv(d.z), false, best, stk, 1u, kCoopStack, lane, n_nodes, n_tris);
        if (shadow) {
            if (lane == 0 && best.prim == 0xFFFFFFFFu) {   // unoccluded: deposit the light sample
                const uint32_t slot = __float_as_uint(d4.w);
                const float4 cc = sq.c[ray];
                float4 L = Lsum[slot];
                L.x = L.x + cc.x; L.y = L.y + cc.y; L.z = L.z + cc.z;
                Lsum[slot] = L;
            }
        } else {
            intersect_lights(sc, o, d, best);
            if (lane == 0) st_nt(hits + ray, make_float4(best.t, best.u, best.v, __uint_as_float(best.prim)));
        }
        __syncthreads();                // the stack is reused by the wave's next ray
    }
    if (STATS && lane == 0) {   // most waves of the grid found no straggler: no atomic for them
        if (n_nodes) atomicAdd(&ctr->nodes, (unsigned long long)n_nodes);
        if (n_tris) atomicAdd(&ctr->tris, (unsigned long long)n_tris);
        if (s_nodes) atomicAdd(&ctr->shadow_nodes, (unsigned long long)s_nodes);
        if (s_tris) atomicAdd(&ctr->shadow_tris, (unsigned long long)s_tris);
    }
}

// five / three consecutive 16-byte scalar loads from a wave-uniform address (the compiler keeps uniform loads of memory it cannot
// prove unwritten on the vector path: 64 lanes fetching one address)
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
// the node's header (grid origin, exponents, masks, child base): the 48 plane bytes behind it reach the lanes through LDS
__device__ __forceinline__ void sload_node_header(const void *p, uint4 &a) {
    u32x4 va;
    asm volatile("s_load_dwordx4 %0, %1, 0x0\n\ts_waitcnt lgkmcnt(0)" : "=&s"(va) : "s"(p) : "memory");
    a = make_uint4(va.x, va.y, va.z, va.w);
}
__device__ __forceinline__ void sload_tri(const void *p, float4 &a, float4 &b, float4 &c) {
    u32x4 va, vb, vc;
    asm volatile("s_load_dwordx4 %0, %3, 0x0\n\ts_load_dwordx4 %1, %3, 0x10\n\ts_load_dwordx4 %2, %3, 0x20\n\ts_waitcnt lgkmcnt(0)"
                 : "=&s"(va), "=&s"(vb), "=&s"(vc) : "s"(p) : "memory");
    a = make_float4(__uint_as_float(va.x), __uint_as_float(va.y), __uint_as_float(va.z), __uint_as_float(va.w));
    b = make_float4(__uint_as_float(vb.x), __uint_as_float(vb.y), __uint_as_float(vb.z), __uint_as_float(vb.w));
    c = make_float4(__uint_as_float(vc.x), __uint_as_float(vc.y), __uint_as_float(vc.z), __uint_as_float(vc.w));
}

// Closest hits of COHERENT rays — the primary rays of bounce 0, 64 consecutive queue entries = one 8x8-pixel patch of one sample —
// by packet traversal: the wave walks the tree ONCE for its 64 rays.  The node (and every triangle) a step looks at is the same for
// all lanes, so it is fetched with scalar loads, once per wave instead of once per lane; every lane tests its own ray against it.
// A child is entered when ANY lane's clipped interval [0, its best hit] meets the child's box; every lane tests every triangle of an
// entered leaf.  A lane therefore tests a superset of the triangles its own traversal would, and only the Woop test decides: the
// hits are those of k_trace bit for bit (closest hit, ties to the lower primitive id: order-independent).  On the bench scene a packet
// of four samples over 4x4 pixels enters 16.4 nodes and tests 8.0 triangles where ONE of its rays alone enters 14.9 and tests 3.6 (tools/dev/packet_probe.cpp,
// the stats kernel); per node 2.45 of the 8 children pass the packet's own test and 1.24 are entered.
// The wave's stack (node indices) is one LDS column per wave.  Works for any rays; it only pays for coherent ones.
template <bool STATS>
__global__ __launch_bounds__(kTraceBlock) __attribute__((amdgpu_waves_per_eu(8))) void k_trace_packet(DScene sc, Queue q, float4 *hits, FrameCounters *ctr, int bounce, uint32_t quad_slots) {
    // LDS of the wave: the 48 child planes of the node in hand as floats (192 B), then the stack of node indices — 7 siblings per
    // level + the path: (7 * depth + 8) entries (host).  ONE wave per block: the hand-overs through `planes` and `stk` below are
    // ordered by the wave's own program order (LDS operations of a wave complete in order) plus the barriers that keep the compiler
    // from moving them; with more than one wave per block they would race.
    static_assert(kTraceBlock == 64, "k_trace_packet: planes[] and stk[] are per-wave LDS, the block must be one wave");
    float *planes = reinterpret_cast<float *>(lds_dyn);
    uint32_t *stk = reinterpret_cast<uint32_t *>(lds_dyn) + 48;
    // read-only for the whole launch and never aliased by what the kernel writes: lets the uniform fetches below become scalar loads
    const DNode8 *__restrict__ nodes = sc.nodes;
    const float4 *__restrict__ woop = sc.woop;
    const uint32_t *__restrict__ leaf_prim = sc.leaf_prim;
    const uint32_t count = QC(ctr, bounce);
    const uint32_t lane = threadIdx.x;
    uint32_t visits = 0, tests = 0;
    for (uint32_t base = blockIdx.x * 64u; base < count; base += gridDim.x * 64u) {
        // Which 64 rays form the packet.  Queue order (a dense frame): sample k's rays at k * quad_slots + slot, 64 consecutive slots = an 8x8-pixel patch.
        // quad_slots != 0 (4 samples per pixel, or a multiple): the packet is the FOUR samples of a 4x4-pixel quarter of that patch instead — the same 64 rays per
        // four packets, but a footprint half as wide: 16.3 nodes and 7.4 triangles per packet instead of 17.7 and 12.1 (tools/dev/packet_probe.cpp).
        uint32_t ray = base + lane;
        if (quad_slots) {
            const uint32_t pk = base >> 6;                      // packet number: (sample group, 8x8 block, quarter)
            const uint32_t blocks = quad_slots >> 6;            // 8x8 blocks per sample
            const uint32_t grp = pk / (blocks * 4u), rem = pk - grp * (blocks * 4u);
            const uint32_t blk = rem >> 2, qtr = rem & 3u;
            const uint32_t smp = grp * 4u + (lane >> 4), sub = lane & 15u;
            const uint32_t px = (qtr & 1u) * 4u + (sub & 3u), py = (qtr >> 1) * 4u + (sub >> 2);
            ray = smp * quad_slots + blk * 64u + py * 8u + px;
        }
        const bool live = ray < count;
        f3 o = mk3(0.f, 0.f, 0.f), d = mk3(0.f, 0.f, 1.f);
        if (live) { const float4 o4 = q.o[ray], d4 = q.d[ray]; o = mk3(o4.x, o4.y, o4.z); d = mk3(d4.x, d4.y, d4.z); }
        const float ix = safe_inv(d.x), iy = safe_inv(d.y), iz = safe_inv(d.z);
        const bool negx = ix < 0.0f, negy = iy < 0.0f, negz = iz < 0.0f;
        Hit best;
        best.t = live ? LPT_T_INF : -1.0f; best.u = 0.f; best.v = 0.f; best.prim = 0xFFFFFFFFu;   // a dead lane's interval [0, -1] is empty: it enters nothing
        // where this lane finds its near / far planes among the node's 48 (qlo_x[8] qlo_y[8] qlo_z[8] qhi_x[8] qhi_y[8] qhi_z[8]): fixed per ray
        const uint32_t onx = negx ? 24u : 0u, ofx = negx ? 0u : 24u, ony = negy ? 32u : 8u, ofy = negy ? 8u : 32u, onz = negz ? 40u : 16u, ofz = negz ? 16u : 40u;
        // visit order: the first lane's octant stands for the packet
        const uint32_t oinv = (uint32_t)__builtin_amdgcn_readfirstlane((int)(7u - ((negx ? 1u : 0u) | (negy ? 2u : 0u) | (negz ? 4u : 0u))));
        // THE PACKET'S OWN BOUNDS, for a test of all eight children at once (lane j: child j) before any lane tests a child for itself.  It needs what primary rays
        // have: ONE origin and, per axis, one sign of the direction — then the entry / exit distance of a plane over the packet lies between its products with the
        // smallest and the largest 1 / d of the packet.  A packet without that (coherent = false) tests every child per lane, as before.
        const unsigned long long lv = __ballot(live);
        const int l0 = lv ? __ffsll((long long)lv) - 1 : 0;
        const float o0x = __uint_as_float((uint32_t)__builtin_amdgcn_readlane((int)__float_as_uint(o.x), l0)), o0y = __uint_as_float((uint32_t)__builtin_amdgcn_readlane((int)__float_as_uint(o.y), l0)),
                    o0z = __uint_as_float((uint32_t)__builtin_amdgcn_readlane((int)__float_as_uint(o.z), l0));
        const bool coherent = lv != 0ull && __ballot(live && (o.x != o0x || o.y != o0y || o.z != o0z)) == 0ull &&
                              ((__ballot(live && negx) == 0ull) || (__ballot(live && negx) == lv)) && ((__ballot(live && negy) == 0ull) || (__ballot(live && negy) == lv)) &&
                              ((__ballot(live && negz) == 0ull) || (__ballot(live && negz) == lv));
        float ixlo = live ? ix : 3.0e38f, ixhi = live ? ix : -3.0e38f, iylo = live ? iy : 3.0e38f, iyhi = live ? iy : -3.0e38f, izlo = live ? iz : 3.0e38f, izhi = live ? iz : -3.0e38f;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            ixlo = fminf(ixlo, __shfl_xor(ixlo, off)); ixhi = fmaxf(ixhi, __shfl_xor(ixhi, off));
            iylo = fminf(iylo, __shfl_xor(iylo, off)); iyhi = fmaxf(iyhi, __shfl_xor(iyhi, off));
            izlo = fminf(izlo, __shfl_xor(izlo, off)); izhi = fmaxf(izhi, __shfl_xor(izhi, off));
        }
        // wave-uniform from here on: scalar registers
        const auto uni = [](float v) { return __uint_as_float((uint32_t)__builtin_amdgcn_readfirstlane((int)__float_as_uint(v))); };
        ixlo = uni(ixlo); ixhi = uni(ixhi); iylo = uni(iylo); iyhi = uni(iyhi); izlo = uni(izlo); izhi = uni(izhi);
        const float iax = fmaxf(fabsf(ixlo), fabsf(ixhi)), iay = fmaxf(fabsf(iylo), fabsf(iyhi)), iaz = fmaxf(fabsf(izlo), fabsf(izhi));
        // where lane j (child j & 7) finds the packet's near / far plane of its child
        const bool pnegx = ((__ballot(negx) >> l0) & 1ull) != 0ull, pnegy = ((__ballot(negy) >> l0) & 1ull) != 0ull, pnegz = ((__ballot(negz) >> l0) & 1ull) != 0ull;
        const uint32_t c8 = lane & 7u;
        const uint32_t cnx = (pnegx ? 24u : 0u) + c8, cfx = (pnegx ? 0u : 24u) + c8, cny = (pnegy ? 32u : 8u) + c8, cfy = (pnegy ? 8u : 32u) + c8, cnz = (pnegz ? 40u : 16u) + c8, cfz = (pnegz ? 16u : 40u) + c8;
        float tb_max = LPT_T_INF;   // the largest best hit of the packet (wave-uniform): no child beyond it can matter
        int sp = 0;
        uint32_t node_index = 0;
        bool have = true;
        while (have) {
            // the 48 quantised planes are the same for every lane: lane j fetches and converts plane j — ONE conversion instruction for the wave
            // instead of 48 — and LDS hands each lane the ones it needs (its near and far plane per axis, four children per read).  The fetch is issued
            // BEFORE the header's scalar loads (which wait for themselves only): the two round trips overlap
            const uint32_t plane_byte = reinterpret_cast<const uint8_t *>(nodes + node_index)[16u + (lane < 48u ? lane : lane - 48u)];   // every lane: no branch around the load
            uint4 n0;
            sload_node_header(nodes + node_index, n0);   // wave-uniform address
            __syncthreads();   // the previous node's plane reads are done before its planes are overwritten (one wave: no wait, a compiler fence)
            if (lane < 48u) planes[lane] = (float)plane_byte;
            __syncthreads();
            if (STATS) visits++;
            const float kEps = 4.76837158203125e-7f;  // 2^-21, as in node_visit
            const NodeGrid g = node_grid(sc, n0);   // wave-uniform
            const float ax = g.sx * ix, ay = g.sy * iy, az = g.sz * iz;
            const float bx = (g.px - o.x) * ix;
            const float by = (g.py - o.y) * iy;
            const float bz = (g.pz - o.z) * iz;
            const float ex = fmaf(fabsf(ax), 255.0f, fabsf(bx)) * kEps;
            const float ey = fmaf(fabsf(ay), 255.0f, fabsf(by)) * kEps;
            const float ez = fmaf(fabsf(az), 255.0f, fabsf(bz)) * kEps;
            const float bnx = bx - ex, bny = by - ey, bnz = bz - ez;
            const float bfx = bx + ex, bfy = by + ey, bfz = bz + ez;
            const float tbest = best.t;
            // all eight children against the packet's bounds, lane j: child j & 7 (conservative: the slack covers the per-lane widening `e` and every rounding below).
            // On the bench frame 2.45 of a node's 8 children pass (1.24 are entered): the per-lane tests below run for those only
            uint32_t cand = 0xFFu;
            if (coherent) {
                const float sx = g.sx, sy = g.sy, sz = g.sz;
                const float gx = g.px - o0x, gy = g.py - o0y, gz = g.pz - o0z;
                const float dnx = fmaf(planes[cnx], sx, gx), dfx = fmaf(planes[cfx], sx, gx);
                const float dny = fmaf(planes[cny], sy, gy), dfy = fmaf(planes[cfy], sy, gy);
                const float dnz = fmaf(planes[cnz], sz, gz), dfz = fmaf(planes[cfz], sz, gz);
                const float pn = fmaxf(fmaxf(fminf(dnx * ixlo, dnx * ixhi), fminf(dny * iylo, dny * iyhi)), fmaxf(fminf(dnz * izlo, dnz * izhi), 0.0f));
                const float pf = fminf(fminf(fmaxf(dfx * ixlo, dfx * ixhi), fmaxf(dfy * iylo, dfy * iyhi)), fminf(fmaxf(dfz * izlo, dfz * izhi), tb_max));
                // |a| * 255 + |b| of any lane is at most (255 * scale + |node origin - o|) * max |1 / d|: four times the per-lane widening, per axis, summed
                const float slack = (fmaf(255.0f, sx, fabsf(gx)) * iax + fmaf(255.0f, sy, fabsf(gy)) * iay + fmaf(255.0f, sz, fabsf(gz)) * iaz) * 1.9073486328125e-6f;   // 2^-19
                cand = (uint32_t)(__ballot(pn - slack <= pf + slack) & 0xFFull);
            }
            uint32_t entered = 0u;   // wave-uniform: slots some lane's ray enters
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                if (!(cand & (1u << j))) continue;   // wave-uniform
                const float tnx = fmaf(planes[onx + j], ax, bnx), tfx = fmaf(planes[ofx + j], ax, bfx);
                const float tny = fmaf(planes[ony + j], ay, bny), tfy = fmaf(planes[ofy + j], ay, bfy);
                const float tnz = fmaf(planes[onz + j], az, bnz), tfz = fmaf(planes[ofz + j], az, bfz);
                const float tn = fmaxf(fmaxf(tnx, tny), fmaxf(tnz, 0.0f));
                const float tf = fminf(fminf(tfx, tfy), fminf(tfz, tbest));
                if (__ballot(tn <= tf) != 0ull) entered |= 1u << j;
            }
            // empty slots have inverted boxes (lo 255, hi 0) and are never entered
            const uint32_t imask = node_imask(n0);
            // leaves first: every lane tests every triangle of an entered leaf slot
            const uint32_t V = node_leaves(n0);
            uint32_t leaves = entered & V & 0xFFu;
            while (leaves) {
                const uint32_t sl = (uint32_t)__ffs((int)leaves) - 1u;
                leaves &= leaves - 1u;
                const uint32_t first = node_index * kNodeTris + 2u * sl, cnt = leaf_count(V, sl);
                for (uint32_t k = 0; k < cnt; ++k) {
                    const uint32_t ti = first + k;
                    float4 r0, r1, r2;
                    sload_tri(woop + 3u * (size_t)ti, r0, r1, r2);
                    if (STATS) tests++;
                    float t, u, v;
                    if (ray_triangle(r0, r1, r2, o, d, best.t, t, u, v)) {
                        const uint32_t prim = leaf_prim[ti];
                        if (t < best.t || prim < best.prim) { best.t = t; best.u = u; best.v = v; best.prim = prim; }
                    }
                }
                if (coherent) {   // the packet's largest best hit, after the slot's triangles (dead lanes carry -1)
                    float m = best.t;
#pragma unroll
                    for (int off = 32; off > 0; off >>= 1) m = fmaxf(m, __shfl_xor(m, off));
                    tb_max = __uint_as_float((uint32_t)__builtin_amdgcn_readfirstlane((int)__float_as_uint(m)));
                }
            }
            // inner children: pushed so that the one nearest along the packet's octant order pops first
            // (k_trace takes bit 24 + (slot ^ oinv) from the top: the largest slot ^ oinv first)
            uint32_t inner = entered & imask;
            uint32_t keyed = 0u;   // bit (slot ^ oinv) set for every entered inner slot
            for (uint32_t m = inner; m; m &= m - 1u) keyed |= 1u << (((uint32_t)__ffs((int)m) - 1u) ^ oinv);
            for (uint32_t m = keyed; m; m &= m - 1u) {   // ascending key: the largest key is pushed last and pops first
                const uint32_t sl = ((uint32_t)__ffs((int)m) - 1u) ^ oinv;
                const uint32_t rel = (uint32_t)__popc(imask & ~(0xFFFFFFFFu << sl));
                if (lane == 0) stk[sp] = n0.w + rel;
                sp++;
            }
            have = sp > 0;
            __syncthreads();   // lane 0's pushes are visible to the pop below
            if (have) { sp--; node_index = (uint32_t)__builtin_amdgcn_readfirstlane((int)stk[sp]); }
        }
        if (live) {
            intersect_lights(sc, o, d, best);
            st_nt(hits + ray, make_float4(best.t, best.u, best.v, __uint_as_float(best.prim)));
        }
    }
    if (STATS && lane == 0) {
        atomicAdd(&ctr->packet_nodes, (unsigned long long)visits);
        atomicAdd(&ctr->packet_tris, (unsigned long long)tests);
    }
}

// stand-alone closest-hit query (lpt_trace_closest): one ray per lane, no refill
__global__ __launch_bounds__(kTraceBlock) void k_query_closest(DScene sc, const float4 *o, const float4 *d, float4 *hits, uint32_t n) {
    uint2 *stack = reinterpret_cast<uint2 *>(lds_dyn) + threadIdx.x;
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float4 o4 = o[i], d4 = d[i];
    const f3 oo = mk3(o4.x, o4.y, o4.z), dd = mk3(d4.x, d4.y, d4.z);
    Hit h;
    uint32_t a = 0, b = 0;
    traverse<false, false>(sc, oo, dd, LPT_T_INF, stack, h, a, b);
    intersect_lights(sc, oo, dd, h);
    hits[i] = make_float4(h.t, h.u, h.v, __uint_as_float(h.prim));
}

// ------------------------------------------------------------------ SPEC §9 textures / environment
__device__ __forceinline__ int wrap_i(int x, int n) {
    if ((n & (n - 1)) == 0) return x & (n - 1);  // power-of-two sizes (the usual case): same value, no division
    int m = x % n;
    return m < 0 ? m + n : m;
}
// wrap(x + 1) from w = wrap(x): no second modulo
__device__ __forceinline__ int wrap_next(int w, int n) { return w + 1 == n ? 0 : w + 1; }

// `lut`: the 256-entry sRGB decode table, staged in LDS by the caller (12 table reads per albedo lookup)
__device__ __forceinline__ float4 texture_lookup(const DScene &sc, const float *lut, uint32_t image, float u, float v, bool srgb) {
    const DImage im = sc.images[image];
    const int W = (int)im.width, H = (int)im.height;
    float fx = u * (float)W - 0.5f, fy = v * (float)H - 0.5f;
    float x0f = floorf(fx), y0f = floorf(fy);
    float tx = fx - x0f, ty = fy - y0f;
    int x0 = wrap_i((int)x0f, W), x1 = wrap_next(x0, W);
    int y0 = wrap_i((int)y0f, H), y1 = wrap_next(y0, H);
    const uchar4 *base = reinterpret_cast<const uchar4 *>(sc.texels) + im.offset;
    const uint32_t tiles_x = im.pad;
    const uint32_t r0 = ((uint32_t)y0 >> 2) * tiles_x * 32u + ((uint32_t)y0 & 3u) * 8u, r1 = ((uint32_t)y1 >> 2) * tiles_x * 32u + ((uint32_t)y1 & 3u) * 8u;
    const uint32_t c0 = ((uint32_t)x0 >> 3) * 32u + ((uint32_t)x0 & 7u), c1 = ((uint32_t)x1 >> 3) * 32u + ((uint32_t)x1 & 7u);
    const uchar4 p00 = base[r0 + c0], p10 = base[r0 + c1];
    const uchar4 p01 = base[r1 + c0], p11 = base[r1 + c1];
    const uint8_t a00[4] = {p00.x, p00.y, p00.z, p00.w}, a10[4] = {p10.x, p10.y, p10.z, p10.w};
    const uint8_t a01[4] = {p01.x, p01.y, p01.z, p01.w}, a11[4] = {p11.x, p11.y, p11.z, p11.w};
    float out[4];
#pragma unroll
    for (int ch = 0; ch < 4; ++ch) {
        float c00, c10, c01, c11;
        if (srgb && ch < 3) { c00 = lut[a00[ch]]; c10 = lut[a10[ch]]; c01 = lut[a01[ch]]; c11 = lut[a11[ch]]; }
        else {
            c00 = (float)a00[ch] * 0.003921568859368563f; c10 = (float)a10[ch] * 0.003921568859368563f;
            c01 = (float)a01[ch] * 0.003921568859368563f; c11 = (float)a11[ch] * 0.003921568859368563f;
        }
        float top = c00 * (1.0f - tx) + c10 * tx, bot = c01 * (1.0f - tx) + c11 * tx;
        out[ch] = top * (1.0f - ty) + bot * ty;
    }
    return make_float4(out[0], out[1], out[2], out[3]);
}

// Both textures of a material from ONE set of four taps (paired texels: x = albedo RGBA8, y = mra RGBA8).  The coordinates, the
// weights and every product are those of two texture_lookup calls on images of this size, so the result is theirs bit for bit;
// only albedo.rgb (sRGB-decoded) and mra.g / mra.b are produced, which is all the shading reads.
__device__ __forceinline__ void texture_lookup_pair(const DScene &sc, const float *lut, uint32_t offset, uint32_t wh, float u, float v, f3 &albedo, float &mra_g, float &mra_b) {
    // the pair's descriptor rides in the shading record itself (offset in 8-byte texels, width | height << 16): no table fetch between
    // the record and the texels
    const int W = (int)(wh & 0xFFFFu), H = (int)(wh >> 16);
    float fx = u * (float)W - 0.5f, fy = v * (float)H - 0.5f;
    float x0f = floorf(fx), y0f = floorf(fy);
    float tx = fx - x0f, ty = fy - y0f;
    int x0 = wrap_i((int)x0f, W), x1 = wrap_next(x0, W);
    int y0 = wrap_i((int)y0f, H), y1 = wrap_next(y0, H);
    const uint2 *base = sc.pair_texels + offset;
    const uint32_t tiles_x = (((uint32_t)W + 2u) * 43691u) >> 17;   // ceil(W / 3)
    // apron tiles: the stored 4x4-texel tile (tx, ty) holds texels 3tx .. 3tx+3 x 3ty .. 3ty+3 (wrapped), so the 2x2 footprint of ANY lookup
    // lies in the ONE 128-byte tile of its upper-left texel; x1 / y1 are the tile's next column / row by construction of the apron
    (void)x1; (void)y1;
    const uint32_t txi = ((uint32_t)x0 * 43691u) >> 17, tyi = ((uint32_t)y0 * 43691u) >> 17;   // / 3, exact below 98304
    const uint32_t ix = (uint32_t)x0 - 3u * txi, iy = (uint32_t)y0 - 3u * tyi;
    const uint2 *tile = base + ((size_t)tyi * tiles_x + txi) * 16u + iy * 4u + ix;
    const uint2 p00 = tile[0], p10 = tile[1], p01 = tile[4], p11 = tile[5];
    float a[3];
#pragma unroll
    for (int ch = 0; ch < 3; ++ch) {
        const float c00 = lut[(p00.x >> (8 * ch)) & 0xFFu], c10 = lut[(p10.x >> (8 * ch)) & 0xFFu];
        const float c01 = lut[(p01.x >> (8 * ch)) & 0xFFu], c11 = lut[(p11.x >> (8 * ch)) & 0xFFu];
        float top = c00 * (1.0f - tx) + c10 * tx, bot = c01 * (1.0f - tx) + c11 * tx;
        a[ch] = top * (1.0f - ty) + bot * ty;
    }
    albedo = mk3(a[0], a[1], a[2]);
    float m[2];
#pragma unroll
    for (int ch = 1; ch < 3; ++ch) {
        const float c00 = (float)((p00.y >> (8 * ch)) & 0xFFu) * 0.003921568859368563f, c10 = (float)((p10.y >> (8 * ch)) & 0xFFu) * 0.003921568859368563f;
        const float c01 = (float)((p01.y >> (8 * ch)) & 0xFFu) * 0.003921568859368563f, c11 = (float)((p11.y >> (8 * ch)) & 0xFFu) * 0.003921568859368563f;
        float top = c00 * (1.0f - tx) + c10 * tx, bot = c01 * (1.0f - tx) + c11 * tx;
        m[ch - 1] = top * (1.0f - ty) + bot * ty;
    }
    mra_g = m[0]; mra_b = m[1];
}

__device__ __forceinline__ f3 rgbe_decode(uchar4 p) {
    const uint32_t e = p.w;
    float scale = 0.0f;
    if (e >= 10u) scale = __uint_as_float((e - 9u) << 23);  // 2^(e-136)
    return mk3((float)p.x * scale, (float)p.y * scale, (float)p.z * scale);
}

__device__ __forceinline__ f3 env_lookup(const DProbe &pr, f3 d) {
    const int W = (int)pr.w, H = (int)pr.h;
    const uchar4 *tex = reinterpret_cast<const uchar4 *>(pr.rgbe);
    if (W == 1 && H == 1) return rgbe_decode(tex[0]);
    float phi = atan2_approx(d.z, d.x);
    float u = phi * LPT_INV_2PI + 0.5f;
    float th = acos_approx(clampf(d.y, -1.0f, 1.0f));
    float v = th * LPT_INV_PI;
    float fx = u * (float)W - 0.5f, fy = v * (float)H - 0.5f;
    float x0f = floorf(fx), y0f = floorf(fy);
    float tx = fx - x0f, ty = fy - y0f;
    int x0 = wrap_i((int)x0f, W), x1 = wrap_next(x0, W);
    int y0 = (int)y0f, y1 = (int)y0f + 1;
    y0 = y0 < 0 ? 0 : (y0 > H - 1 ? H - 1 : y0);
    y1 = y1 < 0 ? 0 : (y1 > H - 1 ? H - 1 : y1);
    const f3 c00 = rgbe_decode(tex[(size_t)y0 * W + x0]), c10 = rgbe_decode(tex[(size_t)y0 * W + x1]);
    const f3 c01 = rgbe_decode(tex[(size_t)y1 * W + x0]), c11 = rgbe_decode(tex[(size_t)y1 * W + x1]);
    f3 r;
    { float top = c00.x * (1.0f - tx) + c10.x * tx, bot = c01.x * (1.0f - tx) + c11.x * tx; r.x = top * (1.0f - ty) + bot * ty; }
    { float top = c00.y * (1.0f - tx) + c10.y * tx, bot = c01.y * (1.0f - tx) + c11.y * tx; r.y = top * (1.0f - ty) + bot * ty; }
    { float top = c00.z * (1.0f - tx) + c10.z * tx, bot = c01.z * (1.0f - tx) + c11.z * tx; r.z = top * (1.0f - ty) + bot * ty; }
    return r;
}

// ------------------------------------------------------------------ SPEC §15 helpers (denoiser path)
__device__ __forceinline__ uint32_t oct_encode(f3 n) {
    float l1 = (fabsf(n.x) + fabsf(n.y)) + fabsf(n.z);
    float px = 0.0f, py = 0.0f;
    if (l1 > 0.0f) { px = n.x / l1; py = n.y / l1; }
    if (n.z < 0.0f) {
        float tx = (1.0f - fabsf(py)) * (px >= 0.0f ? 1.0f : -1.0f);
        float ty = (1.0f - fabsf(px)) * (py >= 0.0f ? 1.0f : -1.0f);
        px = tx; py = ty;
    }
    uint32_t ux = (uint32_t)(clampf(px * 0.5f + 0.5f, 0.0f, 1.0f) * 65535.0f + 0.5f);
    uint32_t uy = (uint32_t)(clampf(py * 0.5f + 0.5f, 0.0f, 1.0f) * 65535.0f + 0.5f);
    return ux | (uy << 16);
}
__device__ __forceinline__ f3 oct_decode(uint32_t p) {
    float fx = (float)(p & 0xFFFFu) * 3.0518043793392844e-05f - 1.0f;
    float fy = (float)(p >> 16) * 3.0518043793392844e-05f - 1.0f;
    float fz = (1.0f - fabsf(fx)) - fabsf(fy);
    if (fz < 0.0f) {
        float tx = (1.0f - fabsf(fy)) * (fx >= 0.0f ? 1.0f : -1.0f);
        float ty = (1.0f - fabsf(fx)) * (fy >= 0.0f ? 1.0f : -1.0f);
        fx = tx; fy = ty;
    }
    return normalize(mk3(fx, fy, fz));
}
__device__ __forceinline__ uint32_t pack_albedo(f3 a) {
    uint32_t r = (uint32_t)(clampf(a.x, 0.0f, 1.0f) * 255.0f + 0.5f), g = (uint32_t)(clampf(a.y, 0.0f, 1.0f) * 255.0f + 0.5f);
    uint32_t b = (uint32_t)(clampf(a.z, 0.0f, 1.0f) * 255.0f + 0.5f);
    return r | (g << 8) | (b << 16) | 0xFF000000u;
}
__device__ __forceinline__ f3 demod_albedo(uint32_t p) {
    return mk3(max2((float)(p & 0xFFu) * 0.003921568859368563f, 0.05f), max2((float)((p >> 8) & 0xFFu) * 0.003921568859368563f, 0.05f),
               max2((float)((p >> 16) & 0xFFu) * 0.003921568859368563f, 0.05f));
}
struct CamBasis { f3 origin, right, up, fwd; float ax, ay; };  // world -> screen, the build's `model_to_screen`
__device__ __forceinline__ bool project(const CamBasis &c, f3 P, float &u, float &v) {
    f3 w = P - c.origin;
    float cz = dot(w, c.fwd);
    if (!(cz > 1.0e-6f)) return false;
    float cx = dot(w, c.right), cy = dot(w, c.up);
    u = 0.5f + 0.5f * (cx / (cz * c.ax));
    v = 0.5f - 0.5f * (cy / (cz * c.ay));
    return true;
}
__device__ __forceinline__ float pow128(float x) { x = x * x; x = x * x; x = x * x; x = x * x; x = x * x; x = x * x; x = x * x; return x; }
// what the primary pass writes besides the path state (reference PrimaryRayPass: gbuffer Rgba32Uint + motion Rg32F,
// asvgf.rs:39-70,104-117; prev_model_to_screen push constant, renderer.rs:472-479)
struct GBufArgs { uint4 *gbuf; float2 *motion; CamBasis cur, prev; };

// ------------------------------------------------------------------ shading (SPEC §12)
// What shading ONE hit produces besides the radiance it adds: the next ray of the path and its next-event shadow ray,
// in the queue layout (Queue / ShadowQueue).
struct ShadeOut {
    bool want_next, want_shadow, is_surface;
    float4 no4, nd4, nT4;   // (origin, pdf) (direction, slot bits) (throughput, x | y << 13 | sample << 26)
    float4 so4, sd4, sc4;   // (origin, tmax) (direction, slot bits) (contribution)
};

// One hit of bounce `bounce` shaded (PrimaryRayPass / ShadingPass): miss -> RGBE probe; emitter -> MIS-weighted emission;
// surface -> shading record, textures, NEE shadow ray and BSDF sample.  Shared by k_shade (one thread per queued ray,
// radiance deposited into Lsum) and k_path (a lane carries its path through every bounce, radiance kept in registers):
// `load_o()` returns the ray's (origin, pdf) record — only emitter hits and the G-buffer need it —, `add_l(r, g, b)`
// adds to the path's radiance.
template <bool GBUF, typename LoadO, typename AddL>
__device__ __forceinline__ void shade_hit(const DScene &sc, const DProbe &probe, const DNoise &nz, const FrameParams &p, const float *s_lut,
                                          const uint32_t bounce, const bool last_bounce, const uint32_t seed_base, const float inv_nl, const GBufArgs &gb,
                                          const float4 d4, const float4 T4, const float4 h4, LoadO load_o, AddL add_l, ShadeOut &out) {
    out.want_next = false; out.want_shadow = false; out.is_surface = false;
    const uint32_t pxy = __float_as_uint(T4.w);  // x | y << 13 | sample << 26 (k_raygen)
    const f3 d = mk3(d4.x, d4.y, d4.z);
    const f3 T = mk3(T4.x, T4.y, T4.z);
    const uint32_t prim = __float_as_uint(h4.w);
    // primary-hit record for the G-buffer (SPEC §15.1); defaults cover miss / emitter / degenerate
    f3 g_n = neg(d), g_alb = mk3(1.0f, 1.0f, 1.0f);
    f3 g_P = mk3(0.f, 0.f, 0.f);
    if (GBUF && bounce == 0) {
        const float4 o4 = load_o();
        g_P = mk3(fmaf(d.x, h4.x, o4.x), fmaf(d.y, h4.x, o4.y), fmaf(d.z, h4.x, o4.z));
    }
    if (prim == 0xFFFFFFFFu) {
        const f3 e = env_lookup(probe, d);
        add_l(T.x * e.x, T.y * e.y, T.z * e.z);
    } else if (prim & LPT_LIGHT_BIT) {
        const float4 *Lt = reinterpret_cast<const float4 *>(sc.lights + (prim & ~LPT_LIGHT_BIT));
        const float4 n4 = Lt[0], t4 = Lt[1], b4 = Lt[2], lo4 = Lt[3];
        const float Le = lo4.w;
        g_n = mk3(n4.x, n4.y, n4.z);
        float w = 1.0f;
        const float pdf_prev = load_o().w;   // emitter hits are rare: the origin record is read only here
        if (pdf_prev >= 0.0f) {
            float cl = -dot(mk3(n4.x, n4.y, n4.z), d);
            float area = 4.0f * (t4.w * b4.w);
            float pl = ((h4.x * h4.x) / (cl * area)) * inv_nl;
            float pb2 = pdf_prev * pdf_prev;
            w = pb2 / (pb2 + pl * pl);
        }
        float k = Le * w;
        add_l(T.x * k, T.y * k, T.z * k);
    } else {
        const float hu = h4.y, hv = h4.z;
        const float4 *tv = sc.tri_verts + kTriRec * (size_t)prim;
        const float4 P0 = tv[0], N0 = tv[1], P1 = tv[2], N1 = tv[3], P2 = tv[4], N2 = tv[5];
        const float4 mc = tv[6], mp = tv[7];
        float bw = (1.0f - hu) - hv;
        const f3 p0 = mk3(P0.x, P0.y, P0.z), p1 = mk3(P1.x, P1.y, P1.z), p2 = mk3(P2.x, P2.y, P2.z);
        f3 P = mk3((p0.x * bw + p1.x * hu) + p2.x * hv, (p0.y * bw + p1.y * hu) + p2.y * hv, (p0.z * bw + p1.z * hu) + p2.z * hv);
        f3 Ng = cross(p1 - p0, p2 - p0);
        float l2 = dot(Ng, Ng);
        if (l2 > 0.0f) {
            out.is_surface = true;
            Ng = Ng * (1.0f / sqrtf(l2));
            f3 Ns = mk3((N0.x * bw + N1.x * hu) + N2.x * hv, (N0.y * bw + N1.y * hu) + N2.y * hv, (N0.z * bw + N1.z * hu) + N2.z * hv);
            float n2 = dot(Ns, Ns);
            Ns = n2 > 0.0f ? Ns * (1.0f / sqrtf(n2)) : Ng;
            if (dot(Ng, d) > 0.0f) Ng = neg(Ng);
            if (dot(Ns, Ng) < 0.0f) Ns = neg(Ns);
            float tu = (P0.w * bw + P1.w * hu) + P2.w * hv;
            float tvv = (N0.w * bw + N1.w * hu) + N2.w * hv;
            f3 base = mk3(mc.x, mc.y, mc.z);
            float rough = mp.x, metal = mp.y;
            const uint32_t atex = __float_as_uint(mp.z), mtex = __float_as_uint(mp.w);
            if ((atex >> 30) == 1u) {   // kPairedBit set, not LPT_INVALID_INDEX: both textures of the material from one set of taps
                f3 alb;
                float mg, mb;
                texture_lookup_pair(sc, s_lut, atex & ~kPairedBit, mtex, tu, tvv, alb, mg, mb);
                base.x *= alb.x; base.y *= alb.y; base.z *= alb.z;
                rough *= mg; metal *= mb;
            } else {
                if (atex < sc.n_images) {
                    const float4 tex = texture_lookup(sc, s_lut, atex, tu, tvv, true);
                    base.x *= tex.x; base.y *= tex.y; base.z *= tex.z;
                }
                if (mtex < sc.n_images) {
                    const float4 tex = texture_lookup(sc, s_lut, mtex, tu, tvv, false);
                    rough *= tex.y; metal *= tex.z;
                }
            }
            g_n = Ns; g_P = P;
            g_alb = mk3(clampf(base.x, 0.0f, 1.0f), clampf(base.y, 0.0f, 1.0f), clampf(base.z, 0.0f, 1.0f));
            const Surface sf = make_surface(base, rough, metal);
            const f3 V = neg(d);
            const float NoV = max2(dot(Ns, V), LPT_MIN_NOV);
            const float pspec = spec_probability(sf, NoV);
            const uint32_t x = pxy & 0x1FFFu, y = (pxy >> 13) & 0x1FFFu, sample = pxy >> 26;
            const uint32_t pixel = y * p.width + x;
            const uint32_t seed_counter = seed_base + sample * p.max_bounces;
            Rng rg = rng_init(pixel, stage_seed(p.user_seed, seed_counter), LPT_TAG_SHADE);
            float r0 = rng_next(rg), r1 = rng_next(rg), r2 = rng_next(rg);
            float r3 = rng_next(rg), r4 = rng_next(rg), r5 = rng_next(rg);
            noise_shift(nz, x, y, seed_counter, r4, r5);
            float am = max2(max2(fabsf(P.x), fabsf(P.y)), fabsf(P.z));
            float eps = 1.0e-4f * (1.0f + am);
            const f3 Po = mk3(P.x + Ng.x * eps, P.y + Ng.y * eps, P.z + Ng.z * eps);
            // next-event estimation
            if (sc.n_lights) {
                uint32_t li = (uint32_t)(r0 * (float)sc.n_lights);
                if (li > sc.n_lights - 1u) li = sc.n_lights - 1u;
                const float4 *Lt = reinterpret_cast<const float4 *>(sc.lights + li);
                const float4 n4 = Lt[0], t4 = Lt[1], b4 = Lt[2], lo4 = Lt[3];
                const float Le = lo4.w, hw = t4.w, hh = b4.w;
                float a = (2.0f * r1 - 1.0f) * hw, bq = (2.0f * r2 - 1.0f) * hh;
                f3 qp = mk3((lo4.x + t4.x * a) + b4.x * bq, (lo4.y + t4.y * a) + b4.y * bq, (lo4.z + t4.z * a) + b4.z * bq);
                f3 w = qp - Po;
                float d2 = dot(w, w);
                if (Le > 0.0f && d2 > 0.0f) {
                    float dist = sqrtf(d2);
                    f3 wi = w * (1.0f / dist);
                    float cl = -dot(mk3(n4.x, n4.y, n4.z), wi);
                    if (cl > 0.0f) {
                        f3 f;
                        float pb;
                        bsdf_eval(sf, Ns, Ng, V, NoV, pspec, wi, f, pb);
                        if (pb > 0.0f) {
                            float area = 4.0f * (hw * hh);
                            float pl = (d2 / (cl * area)) * inv_nl;
                            float pl2 = pl * pl;
                            float wm = pl2 / (pl2 + pb * pb);
                            float NoL = dot(Ns, wi);
                            float k = ((NoL * Le) * wm) / pl;
                            f3 contrib = mk3((T.x * f.x) * k, (T.y * f.y) * k, (T.z * f.z) * k);
                            if (contrib.x > 0.0f || contrib.y > 0.0f || contrib.z > 0.0f) {
                                out.want_shadow = true;
                                out.so4 = make_float4(Po.x, Po.y, Po.z, dist * 0.999f);
                                out.sd4 = make_float4(wi.x, wi.y, wi.z, d4.w);
                                out.sc4 = make_float4(contrib.x, contrib.y, contrib.z, 0.f);
                            }
                        }
                    }
                }
            }
            // BSDF sample -> next ray
            if (!last_bounce) {
                f3 Ln, wgt;
                float pdf;
                if (bsdf_sample(sf, Ns, Ng, V, NoV, pspec, r3, r4, r5, Ln, wgt, pdf)) {
                    f3 Tn = mk3(T.x * wgt.x, T.y * wgt.y, T.z * wgt.z);
                    if (Tn.x > 0.0f || Tn.y > 0.0f || Tn.z > 0.0f) {
                        out.want_next = true;
                        out.no4 = make_float4(Po.x, Po.y, Po.z, pdf);
                        out.nd4 = make_float4(Ln.x, Ln.y, Ln.z, d4.w);
                        out.nT4 = make_float4(Tn.x, Tn.y, Tn.z, T4.w);
                    }
                }
            }
        }
    }
    if (GBUF && bounce == 0) {
        const uint32_t gx = pxy & 0x1FFFu, gy = (pxy >> 13) & 0x1FFFu;
        const size_t px = (size_t)gy * p.width + gx;
        gb.gbuf[px] = make_uint4(prim, __float_as_uint(h4.x), oct_encode(g_n), pack_albedo(g_alb));
        float mu = 0.0f, mv = 0.0f, cu, cv, pu, pv;
        if (prim != 0xFFFFFFFFu && project(gb.cur, g_P, cu, cv) && project(gb.prev, g_P, pu, pv)) { mu = pu - cu; mv = pv - cv; }
        gb.motion[px] = make_float2(mu, mv);
    }
}

template <bool GBUF>
__global__ __launch_bounds__(kBlock) __attribute__((amdgpu_waves_per_eu(4))) void k_shade(DScene sc, DProbe probe, DNoise nz, FrameParams p, Queue qin, const float4 *hits,
                                                  Queue qout, ShadowQueue sq, float4 *Lsum, FrameCounters *ctr, int bounce,
                                                  uint32_t seed_base, GBufArgs gb, int sorted) {
    __shared__ uint32_t lds[72];
    __shared__ float s_lut[256];
    // ONE RESERVATION FOR TWO ITERATIONS (round 6, log N).  A block reserves its slots in both outgoing queues with one returning atomic — on ONE word for the whole grid:
    // 32 000 of them in the first launch of a frame, at the ~88 a word sustains per microsecond that alone is 370 us of a 431 us launch (328 us with the atomic taken out,
    // 308 without any compaction).  So every other iteration only PARKS what it would have stored — a thread's six 16-byte rows and its index within its wave's batch, in
    // its own LDS slot; no barrier, no atomic — and the iteration behind it reserves for both: half the atomics, three barriers per two iterations.  Which slots a ray
    // gets is whatever the atomics' order gives, as before; results are keyed by pixel slot.
    __shared__ float4 s_park[6 * kBlock];
    __shared__ uint32_t s_at[2 * kBlock];
    s_lut[threadIdx.x] = sc.srgb_lut[threadIdx.x];  // kBlock == 256
    __syncthreads();
    bool parked = false;                   // block-uniform: the previous iteration's rows are waiting in s_park (its per-wave counts in lds[16..19])
    const uint32_t count = QC(ctr, bounce);
    const uint32_t stride = gridDim.x * blockDim.x;
    const uint32_t rounded = (count + 255u) & ~255u;  // keep whole blocks in the loop for the barriers
    uint32_t n_surface = 0;
    const bool last_bounce = (uint32_t)bounce + 1u >= p.max_bounces;
    const float inv_nl = sc.n_lights ? 1.0f / (float)sc.n_lights : 0.0f;
    for (uint32_t i0 = blockIdx.x * blockDim.x + threadIdx.x; i0 < rounded; i0 += stride) {
        ShadeOut so;
        so.want_next = false; so.want_shadow = false; so.is_surface = false;
        const uint32_t i = i0;
        if (i < count) {
            const float4 d4 = ld_nt(qin.d + i), T4 = ld_nt(qin.T + i), h4 = ld_nt(hits + i);
            const uint32_t slot = __float_as_uint(d4.w);
            shade_hit<GBUF>(sc, probe, nz, p, s_lut, (uint32_t)bounce, last_bounce, seed_base, inv_nl, gb, d4, T4, h4,
                            [&]() { return ld_nt(qin.o + i); },
                            [&](float r, float g, float b) {
                                float4 L = Lsum[slot];
                                L.x = L.x + r; L.y = L.y + g; L.z = L.z + b;
                                Lsum[slot] = L;
                            }, so);
        }
        // `sorted` (wave-uniform; bit 0: next-bounce queue, bit 1: shadow queue): the queue leaves the block ordered by direction octant
        if (!(sorted & 3)) {   // the default: both queues' slots with one atomic (the last bounce emits no next ray: its word gets + 0) per TWO iterations
            const bool va = so.want_shadow, vb = so.want_next && !last_bounce;
            const unsigned long long ma = __ballot(va), mb = __ballot(vb);
            const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
            const unsigned long long lt = (1ull << lane) - 1ull;
            const uint32_t wa = (uint32_t)__popcll(ma & lt), wb = (uint32_t)__popcll(mb & lt);     // the lane's place in its wave's batch
            if (!parked && i0 + stride < rounded) {   // park; the next iteration reserves for both
                if (lane == 0) lds[16 + wave] = (uint32_t)__popcll(ma) | ((uint32_t)__popcll(mb) << 16);
                s_at[threadIdx.x] = va ? wa : 0xFFFFFFFFu;
                s_at[kBlock + threadIdx.x] = vb ? wb : 0xFFFFFFFFu;
                if (va) { s_park[threadIdx.x] = so.so4; s_park[kBlock + threadIdx.x] = so.sd4; s_park[2 * kBlock + threadIdx.x] = so.sc4; }
                if (vb) { s_park[3 * kBlock + threadIdx.x] = so.no4; s_park[4 * kBlock + threadIdx.x] = so.nd4; s_park[5 * kBlock + threadIdx.x] = so.nT4; }
                parked = true;
            } else {
                if (lane == 0) lds[wave] = (uint32_t)__popcll(ma) | ((uint32_t)__popcll(mb) << 16);
                __syncthreads();
                if (threadIdx.x == 0) {
                    // the block's reservation: [parked wave 0 .. 3][this iteration's wave 0 .. 3], in either queue
                    uint32_t ra = 0u, rb = 0u;
                    for (uint32_t w = 0; w < 8u; ++w) {
                        const uint32_t c = w < 4u ? (parked ? lds[16 + w] : 0u) : lds[w - 4u];
                        lds[24 + w] = ra; lds[32 + w] = rb;
                        ra += c & 0xFFFFu; rb += c >> 16;
                    }
                    unsigned long long base = 0ull;
                    if (ra | rb) base = atomicAdd(reinterpret_cast<unsigned long long *>(&SC(ctr, bounce)), (unsigned long long)ra | ((unsigned long long)rb << 32));
                    lds[40] = (uint32_t)base; lds[41] = (uint32_t)(base >> 32);
                }
                __syncthreads();
                if (parked) {
                    const uint32_t pa = s_at[threadIdx.x], pb = s_at[kBlock + threadIdx.x];
                    if (pa != 0xFFFFFFFFu) {
                        const uint32_t si = lds[40] + lds[24 + wave] + pa;
                        st_nt(sq.o + si, s_park[threadIdx.x]); st_nt(sq.d + si, s_park[kBlock + threadIdx.x]); st_nt(sq.c + si, s_park[2 * kBlock + threadIdx.x]);
                    }
                    if (pb != 0xFFFFFFFFu) {
                        const uint32_t ni = lds[41] + lds[32 + wave] + pb;
                        st_nt(qout.o + ni, s_park[3 * kBlock + threadIdx.x]); st_nt(qout.d + ni, s_park[4 * kBlock + threadIdx.x]); st_nt(qout.T + ni, s_park[5 * kBlock + threadIdx.x]);
                    }
                }
                if (va) { const uint32_t si = lds[40] + lds[28 + wave] + wa; st_nt(sq.o + si, so.so4); st_nt(sq.d + si, so.sd4); st_nt(sq.c + si, so.sc4); }
                if (vb) { const uint32_t ni = lds[41] + lds[36 + wave] + wb; st_nt(qout.o + ni, so.no4); st_nt(qout.d + ni, so.nd4); st_nt(qout.T + ni, so.nT4); }
                __syncthreads();  // lds is reused by the next reservation
                parked = false;
            }
        } else {
            const uint32_t si = (sorted & 2) ? block_compact_binned(so.want_shadow, so.want_shadow ? dir_octant(so.sd4.x, so.sd4.y, so.sd4.z) : 0u, &SC(ctr, bounce), lds)
                                       : block_compact(so.want_shadow, &SC(ctr, bounce), lds);
            if (so.want_shadow) { st_nt(sq.o + si, so.so4); st_nt(sq.d + si, so.sd4); st_nt(sq.c + si, so.sc4); }
            if (!last_bounce) {
                const uint32_t ni = (sorted & 1) ? block_compact_binned(so.want_next, so.want_next ? dir_octant(so.nd4.x, so.nd4.y, so.nd4.z) : 0u, &QC(ctr, bounce + 1), lds)
                                           : block_compact(so.want_next, &QC(ctr, bounce + 1), lds);
                if (so.want_next) { st_nt(qout.o + ni, so.no4); st_nt(qout.d + ni, so.nd4); st_nt(qout.T + ni, so.nT4); }
            }
        }
        n_surface += so.is_surface ? 1u : 0u;
    }
    // surface-hit count: wave reduce, block reduce through LDS, ONE atomic per block and launch on one of 8 words (FrameCounters::shaded_part)
    for (int off = 32; off > 0; off >>= 1) n_surface += __shfl_down(n_surface, off);
    if ((threadIdx.x & 63u) == 0) lds[threadIdx.x >> 6] = n_surface;
    __syncthreads();
    if (threadIdx.x == 0) {
        const uint32_t total = (lds[0] + lds[1]) + (lds[2] + lds[3]);
        if (total) atomicAdd(&ctr->shaded_part[((uint32_t)bounce * 8u + (blockIdx.x & 7u)) * 32u], total);
    }
}

// ------------------------------------------------------------------ the path kernel: every bounce of a wavefront in ONE launch
// The per-bounce launches above (reference renderer.rs:484-509: one dispatch per pass and bounce) make the whole chip wait for the
// longest ray of every bounce: seventeen dependent launches per wavefront, each with its own start-up and drain.  For a full frame
// (8 M rays per launch) the drains are a few per cent; for a tile shard of a multi-GPU frame (1 M rays) they are half of the time
// (DESIGN §5.5).  k_path removes the barrier: a LANE carries a path from its primary hit (k_raygen + k_trace_packet have run) to its
// end — shade, trace the shadow ray, trace the next ray, shade ... — and the wave refills the lanes whose path ended with the next
// primary hits of its chunk.  No queue between the bounces, no compaction, no hit records: the path state stays in registers.
//   * traversal is k_trace's step machine (ray_step_pipe, LDS stacks); a lane remembers whether its ray is the shadow ray (any-hit) of
//     its path or the next closest-hit ray, and traces them in that order, so the additions to the path's radiance happen in the order
//     of the per-bounce launches (shadow deposit of bounce b before anything of bounce b + 1): the frame is theirs bit for bit;
//   * shading runs in BATCHES: the lanes whose closest-hit ray has finished wait until at least `64 - refill` lanes want shading (or
//     as many as are still tracing), so the shading code — three dependent fetches and ~400 instructions — runs for a third of
//     the wave at a time instead of once per finishing lane;
//   * nothing waits for another wave (no spinning on queues that another wave fills): a wave ends when its head has no chunk left
//     and its lanes are done, so the launch cannot deadlock whatever else shares the chip.
// LDS per wave: the traversal stacks, the sRGB table (1 KB), per-bounce counters (lpt_renderer_get_ray_counts / get_queue_counts).
// 4 waves per SIMD (128 VGPRs; the allocator wants 150 and spills 15 dwords per lane around the shading batch): measured on a 1 M-ray
// tile shard, 8 / 12 / 16 waves per CU -> 4.28 / 3.22 / 2.70 ms — the kernel is latency-bound and occupancy is worth more than the spills
// cost (profiles/r04_experiments_ab.txt)
#ifndef LPT_PATH_ATTR
#define LPT_PATH_ATTR __attribute__((amdgpu_waves_per_eu(4, 4)))
#endif
constexpr uint32_t kPathLdsExtra = 1024u + 3u * kMaxBounces * 4u;   // sRGB table + counters: what a wave needs besides its stacks
template <bool GBUF, bool STATS>
__global__ __launch_bounds__(kTraceBlock) LPT_PATH_ATTR void k_path(DScene sc, DProbe probe, DNoise nz, FrameParams p, Queue q0, const float4 *hits0, float4 *Lsum,
                                                      FrameCounters *ctr, uint32_t seed0, GBufArgs gb, int refill) {
    static_assert(kTraceBlock == 64, "one wave per block: the LDS hand-overs below are ordered by the wave's own program order");
    uint2 *stack = reinterpret_cast<uint2 *>(lds_dyn) + threadIdx.x;
    const uint32_t stack_bytes = sc.stack_entries * kTraceBlock * (uint32_t)sizeof(uint2);
    float *s_lut = reinterpret_cast<float *>(lds_dyn + stack_bytes);
    uint32_t *s_cnt = reinterpret_cast<uint32_t *>(lds_dyn + stack_bytes + 1024u);   // [3][kMaxBounces]: next rays, shadow rays, surface hits per bounce
    const uint32_t lane = threadIdx.x;
#pragma unroll
    for (int k = 0; k < 4; ++k) s_lut[lane + 64u * k] = sc.srgb_lut[lane + 64u * k];
#pragma unroll
    for (int k = 0; k < 3; ++k) s_cnt[lane + 64u * k] = 0u;   // kMaxBounces == 64
    __syncthreads();
    ChunkPuller pl;
    puller_init(pl, &ctr->phead[0], QC(ctr, 0));
    const uint32_t nb = p.max_bounces;
    const float inv_nl = sc.n_lights ? 1.0f / (float)sc.n_lights : 0.0f;
    const int min_batch = 64 - refill;
    uint32_t n_nodes = 0, n_tris = 0, s_nodes = 0, s_tris = 0;
    uint32_t w_steps = 0, w_live = 0, w_node = 0, w_tri = 0;
    RayState rs;
    rs.o = mk3(0.f, 0.f, 0.f); rs.d = mk3(0.f, 0.f, 1.f);
    ray_begin(rs, rs.o, rs.d, 0.0f);
    // 0: no path, 1: tracing (shadow: the path's shadow ray, else its next closest-hit ray), 2: closest hit found, waits for shading
    uint32_t phase = 0u;
    bool shadow = false, has_next = false;
    uint32_t bounce = 0u, vslot = 0u, pxy = 0u;
    float pdf = -1.0f;
    f3 T = mk3(0.f, 0.f, 0.f), L = mk3(0.f, 0.f, 0.f), nd = mk3(0.f, 0.f, 0.f), cs = mk3(0.f, 0.f, 0.f);
    for (;;) {
        const unsigned long long amask = __ballot(phase == 1u);
        const unsigned long long wmask = __ballot(phase == 2u);
        const int n_active = __popcll(amask), n_wait = __popcll(wmask);
        puller_pull(pl);                                   // wave-uniform; a no-op while the chunk in hand lasts or the head is dry
        const bool more = pl.next < pl.end;
        const int n_eff = more ? 64 - n_active : n_wait;   // lanes a batch would shade: idle lanes get a new path first
        if (n_eff > 0 && (n_eff >= min_batch || n_eff >= n_active)) {
            for (int rep = 0; rep < 2; ++rep) {            // a chunk that ends inside the batch: the rest of the idle lanes start on the next one
                puller_pull(pl);
                const unsigned long long imask = __ballot(phase == 0u);
                if (!(pl.next < pl.end) || imask == 0ull) break;
                const uint32_t idx = pl.next + (uint32_t)__popcll(imask & ((1ull << lane) - 1ull));
                if (phase == 0u && idx < pl.end) {
                    const float4 d4 = ld_nt(q0.d + idx);
                    pxy = __float_as_uint(q0.T[idx].w);
                    vslot = __float_as_uint(d4.w);
                    T = mk3(1.f, 1.f, 1.f); L = mk3(0.f, 0.f, 0.f);
                    pdf = -1.0f; bounce = 0u;
                    if (hits0) {   // wave-uniform: the primary hits are there (k_trace_packet has run)
                        const float4 h4 = ld_nt(hits0 + idx);
                        rs.o = p.origin; rs.d = mk3(d4.x, d4.y, d4.z);
                        rs.best.t = h4.x; rs.best.u = h4.y; rs.best.v = h4.z; rs.best.prim = __float_as_uint(h4.w);
                        phase = 2u;
                    } else {       // no packet launch for this frame (wide pixels): the lane traces its primary ray itself
                        ray_begin(rs, p.origin, mk3(d4.x, d4.y, d4.z), LPT_T_INF);
                        shadow = false;
                        phase = 1u;
                    }
                }
                pl.next = min(pl.end, pl.next + (uint32_t)__popcll(imask));
            }
            if (phase == 2u) {
                ShadeOut so;
                const float4 d4 = make_float4(rs.d.x, rs.d.y, rs.d.z, __uint_as_float(vslot));
                const float4 T4 = make_float4(T.x, T.y, T.z, __uint_as_float(pxy));
                const float4 h4 = make_float4(rs.best.t, rs.best.u, rs.best.v, __uint_as_float(rs.best.prim));
                shade_hit<GBUF>(sc, probe, nz, p, s_lut, bounce, bounce + 1u >= nb, seed0 + bounce + 1u, inv_nl, gb, d4, T4, h4,
                                [&]() { return make_float4(rs.o.x, rs.o.y, rs.o.z, pdf); },
                                [&](float r, float g, float b) { L.x = L.x + r; L.y = L.y + g; L.z = L.z + b; }, so);
                if (so.is_surface) atomicAdd(&s_cnt[128u + bounce], 1u);
                if (so.want_shadow) atomicAdd(&s_cnt[64u + bounce], 1u);
                if (so.want_next) {
                    atomicAdd(&s_cnt[bounce + 1u], 1u);
                    T = mk3(so.nT4.x, so.nT4.y, so.nT4.z);
                    pdf = so.no4.w;
                    nd = mk3(so.nd4.x, so.nd4.y, so.nd4.z);
                }
                has_next = so.want_next;
                if (so.want_shadow) {
                    ray_begin(rs, mk3(so.so4.x, so.so4.y, so.so4.z), mk3(so.sd4.x, so.sd4.y, so.sd4.z), so.so4.w);
                    cs = mk3(so.sc4.x, so.sc4.y, so.sc4.z);
                    shadow = true;
                    phase = 1u;
                } else if (so.want_next) {
                    ray_begin(rs, mk3(so.no4.x, so.no4.y, so.no4.z), nd, LPT_T_INF);
                    shadow = false;
                    bounce++;
                    phase = 1u;
                } else {
                    Lsum[vslot] = make_float4(L.x, L.y, L.z, 0.0f);
                    phase = 0u;
                }
            }
        }
        if (__ballot(phase != 0u) == 0ull) {
            if (pl.dry) break;
            continue;
        }
        uint32_t dn = 0, dt = 0;
        if (STATS) {
            w_steps++;
            w_live += (uint32_t)__popcll(__ballot(phase == 1u));
            w_node += (uint32_t)__popcll(__ballot(phase == 1u && (rs.tg2.y == 0u) && ((rs.ng.y & 0xFF000000u) != 0u || rs.sp != 0)));
        }
        if (phase == 1u && ray_step_pipe<STATS>(sc, rs, stack, shadow, dn, dt)) {
            if (shadow) {
                if (rs.best.prim == 0xFFFFFFFFu) { L.x = L.x + cs.x; L.y = L.y + cs.y; L.z = L.z + cs.z; }   // unoccluded: deposit the light sample
                if (has_next) {
                    ray_begin(rs, rs.o, nd, LPT_T_INF);   // the next ray leaves the point the shadow ray left
                    shadow = false;
                    bounce++;
                } else {
                    Lsum[vslot] = make_float4(L.x, L.y, L.z, 0.0f);
                    phase = 0u;
                }
            } else {
                intersect_lights(sc, rs.o, rs.d, rs.best);
                phase = 2u;
            }
        }
        if (STATS) {
            w_tri += (uint32_t)__popcll(__ballot(dt != 0u));
            if (shadow) { s_nodes += dn; s_tris += dt; } else { n_nodes += dn; n_tris += dt; }
        }
    }
    __syncthreads();
    if (lane >= 1u && lane <= nb && s_cnt[lane]) atomicAdd(&QC(ctr, lane), s_cnt[lane]);
    if (lane < nb) {
        if (s_cnt[64u + lane]) atomicAdd(&SC(ctr, lane), s_cnt[64u + lane]);
        if (s_cnt[128u + lane]) atomicAdd(&ctr->shaded[lane], s_cnt[128u + lane]);
    }
    if (STATS) {
        atomicAdd(&ctr->nodes, (unsigned long long)n_nodes);
        atomicAdd(&ctr->tris, (unsigned long long)n_tris);
        atomicAdd(&ctr->shadow_nodes, (unsigned long long)s_nodes);
        atomicAdd(&ctr->shadow_tris, (unsigned long long)s_tris);
        if (lane == 0) {
            atomicAdd(&ctr->wave_steps, (unsigned long long)w_steps);
            atomicAdd(&ctr->live_lanes, (unsigned long long)w_live);
            atomicAdd(&ctr->node_lanes, (unsigned long long)w_node);
            atomicAdd(&ctr->tri_lanes, (unsigned long long)w_tri);
        }
    }
}

// ------------------------------------------------------------------ accumulation (SPEC §13)
__global__ __launch_bounds__(kBlock) void k_accumulate(FrameParams p, const float4 *Lsum, float4 *accum) {
    const uint32_t stride = gridDim.x * blockDim.x;
    for (uint32_t slot = blockIdx.x * blockDim.x + threadIdx.x; slot < p.n_slots; slot += stride) {
        uint32_t x, y;
        if (!slot_to_pixel(p, p.slot0 + slot, x, y)) continue;
        const size_t px = (size_t)y * p.width + x;
        uint32_t fc = p.frame_count;
        float4 a = fc == 1u ? make_float4(0.f, 0.f, 0.f, 0.f) : accum[px];
        for (uint32_t k = 0; k < p.n_samples; ++k) {  // in call order: the fp32 sums are order-sensitive
            const float4 L = Lsum[(size_t)k * p.n_slots + slot];
            if (fc == 1u) a = make_float4(L.x, L.y, L.z, 1.0f);
            else { a.x = a.x + L.x; a.y = a.y + L.y; a.z = a.z + L.z; a.w = a.w + 1.0f; }
            fc += k == 0u ? p.fc_inc0 : 1u;
        }
        accum[px] = a;
    }
}

// ------------------------------------------------------------------ frame exchange (owned-tile gather)
// A rank's owned pixels in SLOT order (tile after tile, inside a tile as slot_to_pixel orders them): what travels to rank 0.  Slots of
// edge tiles that fall outside the image carry zeros and are never read back.
__global__ __launch_bounds__(kBlock) void k_pack_owned(FrameParams p, const float4 *accum, float4 *out) {
    const uint32_t stride = gridDim.x * blockDim.x;
    for (uint32_t slot = blockIdx.x * blockDim.x + threadIdx.x; slot < p.n_slots; slot += stride) {
        uint32_t x, y;
        out[slot] = slot_to_pixel(p, slot, x, y) ? accum[(size_t)y * p.width + x] : make_float4(0.f, 0.f, 0.f, 0.f);
    }
}
// HOST-SIDE GATHER: the rank's owned pixels of the MEAN radiance, written straight into a whole-frame destination — page-locked host memory
// mapped into the device's address space (every rank of a node maps the same shared-memory frame).  Each GPU pushes its 1/N of the frame over its own
// PCIe link, and nothing is gathered on rank 0's GPU first.  Pixels inside a tile are walked row by row here (32 x 16 B = 512-byte runs on the link).
__global__ __launch_bounds__(kBlock) void k_resolve_owned(FrameParams p, const float4 *accum, float4 *dst) {
    p.block8 = 0u;   // any enumeration of the owned pixels will do: row-major inside a tile
    const uint32_t stride = gridDim.x * blockDim.x;
    for (uint32_t slot = blockIdx.x * blockDim.x + threadIdx.x; slot < p.n_slots; slot += stride) {
        uint32_t x, y;
        if (!slot_to_pixel(p, slot, x, y)) continue;
        const size_t px = (size_t)y * p.width + x;
        const float4 a = accum[px];
        dst[px] = a.w > 0.0f ? make_float4(a.x / a.w, a.y / a.w, a.z / a.w, 1.0f) : make_float4(0.f, 0.f, 0.f, 0.f);   // k_resolve's arithmetic
    }
}
// first slot of rank `q` in the concatenation of all ranks' slot arrays: ranks own floor(n_tiles / world) tiles, the
// first n_tiles % world of them one more (tile id mod world = owner)
__device__ __host__ __forceinline__ uint32_t shard_slot_offset(uint32_t n_tiles, uint32_t world, uint32_t tile_area, uint32_t q) {
    const uint32_t base = n_tiles / world, rem = n_tiles - base * world;
    return (q * base + (q < rem ? q : rem)) * tile_area;
}
// tile -> (owner, slot of the tile's first pixel in the owner's slot array): the inverse of slot_to_pixel for every owner
__device__ __forceinline__ void tile_owner(const ShardTable &t, uint32_t tile, uint32_t area, uint32_t &owner, uint32_t &slot0) {
    const uint32_t period = tile / t.V, v = tile - period * t.V;
    if (t.unit_world) { owner = v; slot0 = period * area; return; }   // tile id mod world
    owner = t.owner[v];
    slot0 = (period * t.w[owner] + t.j[v]) * area;
}
// where rank q's slots start in the staging area
__device__ __forceinline__ uint32_t rank_offset(const ShardTable &t, uint32_t n_tiles, uint32_t area, uint32_t q) {
    return t.unit_world ? shard_slot_offset(n_tiles, t.unit_world, area, q) : t.offset[q];
}
// rank 0: the whole frame from the concatenated slot arrays
__global__ __launch_bounds__(kBlock) void k_unpack_frame(FrameParams p, const ShardTable *tp, const float4 *staged, float4 *frame) {
    const ShardTable &st = *tp;   // 400 bytes in device memory, read per lane (L1-resident)
    const uint32_t stride = gridDim.x * blockDim.x, npx = p.width * p.height;
    const uint32_t area = p.tile_w * p.tile_h;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < npx; i += stride) {
        const uint32_t y = i / p.width, x = i - y * p.width;
        const uint32_t ty = y / p.tile_h, tx = x / p.tile_w;
        uint32_t owner, slot0;
        tile_owner(st, ty * p.tiles_x + tx, area, owner, slot0);
        const uint32_t slot = slot0 + xy_to_within(p, x - tx * p.tile_w, y - ty * p.tile_h);
        frame[i] = staged[(size_t)rank_offset(st, p.n_tiles, area, owner) + slot];
    }
}

// The same exchange for the denoising BlitModes: a rank's owned pixels of the three filter inputs (noisy radiance float4,
// G-buffer uint4, motion float2 = 40 B per slot), staged as three consecutive runs of n_slots elements each.
__global__ __launch_bounds__(kBlock) void k_pack_den(FrameParams p, const float4 *noisy, const uint4 *gbuf, const float2 *motion, unsigned char *out) {
    const uint32_t stride = gridDim.x * blockDim.x;
    float4 *oa = reinterpret_cast<float4 *>(out);
    uint4 *ob = reinterpret_cast<uint4 *>(out + 16u * (size_t)p.n_slots);
    float2 *oc = reinterpret_cast<float2 *>(out + 32u * (size_t)p.n_slots);
    for (uint32_t slot = blockIdx.x * blockDim.x + threadIdx.x; slot < p.n_slots; slot += stride) {
        uint32_t x, y;
        const bool in = slot_to_pixel(p, slot, x, y);
        const size_t px = in ? (size_t)y * p.width + x : 0;
        oa[slot] = in ? noisy[px] : make_float4(0.f, 0.f, 0.f, 0.f);
        ob[slot] = in ? gbuf[px] : make_uint4(0u, 0u, 0u, 0u);
        oc[slot] = in ? motion[px] : make_float2(0.f, 0.f);
    }
}
__global__ __launch_bounds__(kBlock) void k_unpack_den(FrameParams p, const ShardTable *tp, const unsigned char *staged, float4 *noisy, uint4 *gbuf, float2 *motion) {
    const ShardTable &st = *tp;
    const uint32_t stride = gridDim.x * blockDim.x, npx = p.width * p.height;
    const uint32_t area = p.tile_w * p.tile_h;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < npx; i += stride) {
        const uint32_t y = i / p.width, x = i - y * p.width;
        const uint32_t ty = y / p.tile_h, tx = x / p.tile_w;
        uint32_t owner, slot0;
        tile_owner(st, ty * p.tiles_x + tx, area, owner, slot0);
        const uint32_t slot = slot0 + xy_to_within(p, x - tx * p.tile_w, y - ty * p.tile_h);
        const uint32_t first = rank_offset(st, p.n_tiles, area, owner);
        const size_t n_owner = rank_offset(st, p.n_tiles, area, owner + 1u) - first;
        const unsigned char *base = staged + 40u * (size_t)first;
        noisy[i] = reinterpret_cast<const float4 *>(base)[slot];
        gbuf[i] = reinterpret_cast<const uint4 *>(base + 16u * n_owner)[slot];
        motion[i] = reinterpret_cast<const float2 *>(base + 32u * n_owner)[slot];
    }
}

// ------------------------------------------------------------------ denoiser passes (SPEC §15.2-15.4)
// TemporalAccumulationPass (asvgf.rs:245-247): nearest reprojection + consistency test, moments, history
// the frame's noisy radiance, per PIXEL (the path state is per slot of this rank's tiles): what the filter passes read,
// and — with gbuffer and motion — what the ranks of a sharded frame exchange before rank 0 filters
__global__ __launch_bounds__(kBlock) void k_den_scatter(FrameParams p, const float4 *Lsum, float4 *noisy) {
    const uint32_t stride = gridDim.x * blockDim.x;
    for (uint32_t slot = blockIdx.x * blockDim.x + threadIdx.x; slot < p.n_slots; slot += stride) {
        uint32_t x, y;
        if (!slot_to_pixel(p, p.slot0 + slot, x, y)) continue;
        noisy[(size_t)y * p.width + x] = Lsum[slot];
    }
}

__global__ __launch_bounds__(kBlock) void k_temporal(int W, int H, const float4 *noisy, const uint4 *g_cur, const uint4 *g_prev, const float2 *motion,
                                                     const float4 *rad_prev, const float2 *mom_prev, const uint32_t *hist_prev,
                                                     float4 *rad_cur, float2 *mom_cur, uint32_t *hist_cur) {
    const uint32_t stride = gridDim.x * blockDim.x, npx = (uint32_t)(W * H);
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < npx; i += stride) {
        const uint32_t y = i / (uint32_t)W, x = i - y * (uint32_t)W;
        const uint4 g = g_cur[i];
        const f3 a = demod_albedo(g.w);
        const float4 L = noisy[i];
        const f3 il = mk3(L.x / a.x, L.y / a.y, L.z / a.z);
        const float lm = lum(il);
        const float2 mo = motion[i];
        const int mx = (int)floorf(((float)x + 0.5f) + mo.x * (float)W);
        const int my = (int)floorf(((float)y + 0.5f) + mo.y * (float)H);
        uint32_t hn = 1u;
        f3 col = il;
        float m1 = lm, m2 = lm * lm;
        if (mx >= 0 && my >= 0 && mx < W && my < H) {
            const size_t j = (size_t)my * W + mx;
            const uint4 gp = g_prev[j];
            const uint32_t hp = hist_prev[j];
            const float zc = __uint_as_float(g.y), zp = __uint_as_float(gp.y);
            const bool ok = hp > 0u && gp.x == g.x && dot(oct_decode(g.z), oct_decode(gp.z)) >= 0.9f && fabsf(zc - zp) <= 0.1f * max2(zc, zp);
            if (ok) {
                hn = hp + 1u;
                if (hn > 64u) hn = 64u;
                const float al = 1.0f / (float)hn;
                const float4 pc = rad_prev[j];
                const float2 pm = mom_prev[j];
                col = mk3(pc.x + (il.x - pc.x) * al, pc.y + (il.y - pc.y) * al, pc.z + (il.z - pc.z) * al);
                m1 = pm.x + (lm - pm.x) * al;
                m2 = pm.y + (lm * lm - pm.y) * al;
            }
        }
        float var = max2(m2 - m1 * m1, 0.0f);
        if (hn < 4u) var = var + (m1 * m1) * ((float)(4u - hn) * 0.25f);
        rad_cur[i] = make_float4(col.x, col.y, col.z, var);
        mom_cur[i] = make_float2(m1, m2);
        hist_cur[i] = hn;
    }
}

// The G-buffer's normal and depth decoded ONCE per frame for the four a-trous passes (24 taps each would decode the
// octahedral normal again: 96 decodes per pixel): (n.x, n.y, n.z, depth), n.x = 2 marks a pixel without a primary hit.
// The decoded values are those oct_decode returns, so the filter's result does not change by a bit.
__global__ __launch_bounds__(kBlock) void k_decode_gbuf(const uint4 *gb, float4 *nd, uint32_t n) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint4 g = gb[i];
    if (g.x == 0xFFFFFFFFu) { nd[i] = make_float4(2.0f, 0.0f, 0.0f, __uint_as_float(g.y)); return; }
    const f3 nn = oct_decode(g.z);
    nd[i] = make_float4(nn.x, nn.y, nn.z, __uint_as_float(g.y));
}

// ATrousPass (asvgf.rs:278-287): 5x5 B3-spline taps `step` pixels apart, edge-stopping on normal / depth / luminance
__global__ __launch_bounds__(kBlock) void k_atrous(const float4 *nd, const float4 *in, float4 *out, int W, int H, int step) {
    const uint32_t idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (uint32_t)(W * H)) return;
    const int y = (int)(idx / (uint32_t)W), x = (int)(idx - (uint32_t)y * (uint32_t)W);
    const float kw[5] = {0.0625f, 0.25f, 0.375f, 0.25f, 0.0625f};
    const float4 g = nd[idx];
    const float4 c = in[idx];
    if (g.x > 1.5f) { out[idx] = c; return; }
    const f3 nc = mk3(g.x, g.y, g.z);
    const float zc = g.w;
    const float lc = lum(mk3(c.x, c.y, c.z));
    const float sigma_l = 4.0f * sqrtf(max2(c.w, 0.0f)) + 1.0e-4f;
    const float sigma_z = 0.02f * zc + 1.0e-6f;
    const float wc = kw[2] * kw[2];
    float sr = c.x * wc, sg = c.y * wc, sb = c.z * wc, sv = c.w * (wc * wc), sw = wc;
    for (int dy = -2; dy <= 2; ++dy)
        for (int dx = -2; dx <= 2; ++dx) {
            if (dx == 0 && dy == 0) continue;
            const int qx = x + dx * step, qy = y + dy * step;
            if (qx < 0 || qy < 0 || qx >= W || qy >= H) continue;
            const size_t j = (size_t)qy * W + qx;
            const float4 gq = nd[j];
            if (gq.x > 1.5f) continue;
            const float4 q = in[j];
            const float zq = gq.w;
            const float wn = pow128(max2(dot(nc, mk3(gq.x, gq.y, gq.z)), 0.0f));
            const float rz = fabsf(zc - zq) / sigma_z;
            const float wz = 1.0f / (1.0f + rz * rz);
            const float rl = fabsf(lc - lum(mk3(q.x, q.y, q.z))) / sigma_l;
            const float wl = 1.0f / (1.0f + rl * rl);
            const float w = ((kw[dx + 2] * kw[dy + 2]) * wn) * (wz * wl);
            sr += q.x * w; sg += q.y * w; sb += q.z * w; sv += q.w * (w * w); sw += w;
        }
    const float inv = 1.0f / sw;
    out[idx] = make_float4(sr * inv, sg * inv, sb * inv, sv * (inv * inv));
}

// CompositingPass (asvgf.rs:288-290): re-modulate with the primary albedo into the main target
__global__ __launch_bounds__(kBlock) void k_composite(const uint4 *gb, const float4 *in, float4 *out, uint32_t n) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const f3 a = demod_albedo(gb[i].w);
    const float4 c = in[i];
    out[i] = make_float4(c.x * a.x, c.y * a.y, c.z * a.z, 1.0f);
}

// debug views of BlitMode::GBuffer / MotionVector (renderer.rs:574-586)
__global__ __launch_bounds__(kBlock) void k_debug_view(const uint4 *gb, const float2 *motion, uchar4 *out, int W, int H, int mode) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (uint32_t)(W * H)) return;
    if (mode == 3) {
        const f3 n = oct_decode(gb[i].z);
        out[i] = make_uchar4((uint8_t)((n.x * 0.5f + 0.5f) * 255.0f + 0.5f), (uint8_t)((n.y * 0.5f + 0.5f) * 255.0f + 0.5f),
                             (uint8_t)((n.z * 0.5f + 0.5f) * 255.0f + 0.5f), 255);
    } else {
        const float2 m = motion[i];
        out[i] = make_uchar4((uint8_t)(clampf(fabsf(m.x) * (float)W * 0.125f, 0.0f, 1.0f) * 255.0f + 0.5f),
                             (uint8_t)(clampf(fabsf(m.y) * (float)H * 0.125f, 0.0f, 1.0f) * 255.0f + 0.5f), 0, 255);
    }
}

// one wave: lane b gathers bounce b's counts (kMaxBounces = 64 lanes), a shuffle reduction adds them up — one thread walking 90 dependent loads took 15 us
// at the end of every wavefront
__global__ __launch_bounds__(64) void k_finish_frame(FrameCounters *ctr, Totals *tot, uint32_t bounces, uint32_t packet_primary) {
    static_assert(kMaxBounces <= 64, "a lane per bounce");
    if (blockIdx.x != 0) return;
    const uint32_t b = threadIdx.x;
    unsigned long long c = 0, s = 0, sh = 0;
    if (b < bounces) {
        c = QC(ctr, (int)b); s = SC(ctr, (int)b); sh = ctr->shaded[b];
        for (uint32_t k = 0; k < 8u; ++k) sh += ctr->shaded_part[(b * 8u + k) * 32u];
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) { c += __shfl_xor(c, off); s += __shfl_xor(s, off); sh += __shfl_xor(sh, off); }
    if (b != 0) return;
    if (packet_primary) { tot->primary += QC(ctr, 0); tot->packet_nodes += ctr->packet_nodes; tot->packet_tris += ctr->packet_tris; }
    tot->closest += c; tot->shadow += s; tot->shaded += sh; tot->nodes += ctr->nodes; tot->tris += ctr->tris;
    tot->shadow_nodes += ctr->shadow_nodes; tot->shadow_tris += ctr->shadow_tris;
    tot->wave_steps += ctr->wave_steps; tot->live_lanes += ctr->live_lanes; tot->node_lanes += ctr->node_lanes; tot->tri_lanes += ctr->tri_lanes;
    tot->shadow_occluded += ctr->shadow_occluded; tot->occ_found += ctr->occ_found; tot->occ_hits += ctr->occ_hits;
    unsigned long long wr = 0ull;
    for (uint32_t k = 0; k < kTailCounters; ++k) wr += ctr->tail_rays[k * 32u];
    for (uint32_t l = 0; l <= bounces && l <= (uint32_t)kMaxBounces; ++l) wr += ctr->strag_count[l];
    tot->wave_rays += wr;
}

// mean radiance (a = 1 where sampled) and sRGB8 (SPEC §13.2)
__global__ __launch_bounds__(kBlock) void k_resolve(const float4 *accum, float4 *mean, uint32_t n) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float4 a = accum[i];
    mean[i] = a.w > 0.0f ? make_float4(a.x / a.w, a.y / a.w, a.z / a.w, 1.0f) : make_float4(0.f, 0.f, 0.f, 0.f);
}
// SPEC §13.2: code i starts at thr[i] (the inverse OETF of (i - 0.5) / 255, computed in binary64 on the host): an 8-step search
// on float comparisons — exact, and independent of the device's powf
__device__ __forceinline__ uint8_t encode_srgb8(float c, const float *thr) {
    c = clampf(c, 0.0f, 1.0f);
    uint32_t idx = 0;
#pragma unroll
    for (uint32_t step = 128u; step; step >>= 1)
        if (idx + step <= 255u && c >= thr[idx + step]) idx += step;
    return (uint8_t)idx;
}
__global__ __launch_bounds__(kBlock) void k_tonemap(const float4 *accum, uchar4 *out, uint32_t n, const float *thr_global) {
    __shared__ float thr[256];
    thr[threadIdx.x] = thr_global[threadIdx.x];  // kBlock == 256
    __syncthreads();
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float4 a = accum[i];
    const bool ok = a.w > 0.0f;
    out[i] = make_uchar4(encode_srgb8(ok ? a.x / a.w : 0.f, thr), encode_srgb8(ok ? a.y / a.w : 0.f, thr), encode_srgb8(ok ? a.z / a.w : 0.f, thr), 255);
}

// stand-alone ray queries (lpt_trace_closest / lpt_trace_occluded)
__global__ __launch_bounds__(kTraceBlock) void k_query_occluded(DScene sc, const float4 *o, const float4 *d, uint8_t *out, uint32_t n) {
    uint2 *stack = reinterpret_cast<uint2 *>(lds_dyn) + threadIdx.x;
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float4 o4 = o[i], d4 = d[i];
    Hit h;
    uint32_t a = 0, b = 0;
    out[i] = traverse<true, false>(sc, mk3(o4.x, o4.y, o4.z), mk3(d4.x, d4.y, d4.z), o4.w, stack, h, a, b) ? 1 : 0;
}


// ------------------------------------------------------------------ refit (lpt_scene_gpu_update_instances)
// The tree keeps its topology; every node's child boxes, grid origin and exponents are recomputed from the
// (moved) triangles, one breadth-first level per launch, deepest level first.  Mirrors the quantisation of
// the host builder (bvh.cpp): padded triangle boxes, power-of-two grid step >= extent/255, floor / ceil.
__device__ __forceinline__ void refit_grow_tri(const DScene &sc, uint32_t prim, float lo[3], float hi[3]) {
    const float4 *tv = sc.tri_verts + kTriRec * (size_t)prim;
    const float4 P0 = tv[0], P1 = tv[2], P2 = tv[4];
    const float px[3] = {P0.x, P1.x, P2.x}, py[3] = {P0.y, P1.y, P2.y}, pz[3] = {P0.z, P1.z, P2.z};
    const float *pp[3] = {px, py, pz};
    for (int a = 0; a < 3; ++a) {
        float l = fminf(fminf(pp[a][0], pp[a][1]), pp[a][2]), h = fmaxf(fmaxf(pp[a][0], pp[a][1]), pp[a][2]);
        const float m = fmaxf(fabsf(l), fabsf(h));
        const float e = 4e-6f * m + sc.pad_abs + 1e-6f * (h - l) + 1e-30f;   // bvh.cpp padded_box
        l -= e; h += e;
        lo[a] = fminf(lo[a], l); hi[a] = fmaxf(hi[a], h);
    }
}

__global__ __launch_bounds__(64) void k_refit_level(DScene sc, uint4 *nodes_rw, float4 *node_lo, float4 *node_hi, uint32_t first, uint32_t last) {
    const uint32_t i = first + blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= last) return;
    uint4 *nw = nodes_rw + 4u * (size_t)i;
    const uint4 n0 = nw[0];
    const uint32_t imask = node_imask(n0);
    float clo[8][3], chi[8][3];
    float nlo[3] = {3.0e38f, 3.0e38f, 3.0e38f}, nhi[3] = {-3.0e38f, -3.0e38f, -3.0e38f};
    uint32_t rel = 0;
    bool any = false;
    const uint32_t V = node_leaves(n0);
    for (int sl = 0; sl < 8; ++sl) {
        for (int a = 0; a < 3; ++a) { clo[sl][a] = 3.0e38f; chi[sl][a] = -3.0e38f; }
        if ((imask >> sl) & 1u) {
            const uint32_t c = n0.w + rel++;
            const float4 l = node_lo[c], h = node_hi[c];
            clo[sl][0] = l.x; clo[sl][1] = l.y; clo[sl][2] = l.z; chi[sl][0] = h.x; chi[sl][1] = h.y; chi[sl][2] = h.z;
        } else if ((V >> sl) & 1u) {
            const uint32_t cnt = leaf_count(V, (uint32_t)sl);
            for (uint32_t k = 0; k < cnt; ++k) refit_grow_tri(sc, sc.leaf_prim[i * kNodeTris + 2u * (uint32_t)sl + k], clo[sl], chi[sl]);
        } else continue;
        any = true;
        for (int a = 0; a < 3; ++a) { nlo[a] = fminf(nlo[a], clo[sl][a]); nhi[a] = fmaxf(nhi[a], chi[sl][a]); }
    }
    if (!any) return;  // the empty scene's single node
    node_lo[i] = make_float4(nlo[0], nlo[1], nlo[2], 0.f);
    node_hi[i] = make_float4(nhi[0], nhi[1], nhi[2], 0.f);
    // the node's origin: the scene-grid point at or below its box minimum (common.h grid_snap); the steps cover the box from there
    uint32_t og[3];
    float org[3];
    for (int a = 0; a < 3; ++a) {
        double u = floor(((double)nlo[a] - (double)sc.grid_lo[a]) / (double)sc.grid_step[a]);
        u = u < 0.0 ? 0.0 : (u > 65535.0 ? 65535.0 : u);
        uint32_t ui = (uint32_t)u;
        float p = fmaf((float)ui, sc.grid_step[a], sc.grid_lo[a]);
        while (p > nlo[a] && ui > 0u) { --ui; p = fmaf((float)ui, sc.grid_step[a], sc.grid_lo[a]); }
        og[a] = ui; org[a] = p;
    }
    uint32_t eb[3];
    double scale[3];
    for (int a = 0; a < 3; ++a) {
        const double ext = (double)nhi[a] - (double)org[a];
        int e = -126;
        if (ext > 0.0) {
            int k;
            frexp(ext / 255.0, &k);
            e = min(max(k, -126), 127);
        }
        eb[a] = (uint32_t)(e + 127);
        scale[a] = ldexp(1.0, e);
    }
    uint8_t q[6][8];
    for (int sl = 0; sl < 8; ++sl) {
        if (!(((imask | V) >> sl) & 1u)) { for (int a = 0; a < 3; ++a) { q[a][sl] = 255; q[3 + a][sl] = 0; } continue; }
        for (int a = 0; a < 3; ++a) {
            const double l = floor(((double)clo[sl][a] - (double)org[a]) / scale[a]);
            const double h = ceil(((double)chi[sl][a] - (double)org[a]) / scale[a]);
            q[a][sl] = (uint8_t)fmin(fmax(l, 0.0), 255.0);
            q[3 + a][sl] = (uint8_t)fmin(fmax(h, 0.0), 255.0);
        }
    }
    auto pack4 = [&](int plane, int h) { return (uint32_t)q[plane][4 * h] | ((uint32_t)q[plane][4 * h + 1] << 8) | ((uint32_t)q[plane][4 * h + 2] << 16) | ((uint32_t)q[plane][4 * h + 3] << 24); };
    nw[0] = make_uint4(og[0] | (og[1] << 16), og[2] | (eb[0] << 16) | (eb[1] << 24), eb[2] | (n0.z & 0xFFFFFF00u), n0.w);   // masks and child base: the builder's
    nw[1] = make_uint4(pack4(0, 0), pack4(0, 1), pack4(1, 0), pack4(1, 1));
    nw[2] = make_uint4(pack4(2, 0), pack4(2, 1), pack4(3, 0), pack4(3, 1));
    nw[3] = make_uint4(pack4(4, 0), pack4(4, 1), pack4(5, 0), pack4(5, 1));
}

// ------------------------------------------------------------------ device-side baking of ONE instance (SPEC §2.5, §6)
// The object-space mesh stays on the device; a moved instance is re-baked here with exactly the host's operations
// (bvh.cpp bake_one / woop_from_triangle: fp32 transform with the parentheses shown there, cofactor normals,
// Woop maps in binary64 rounded once) — -ffp-contract=off on both sides, so the results are bit-identical.
struct BakeArgs {
    float m[16];        // model_to_world, column-major
    float c[9];         // cofactors of its 3x3 part, as bake_one computes them (row-major c00..c22)
    uint32_t vertex_offset, index_offset, first_tri, n_tris;
    float4 mat[2];      // the instance's material (lpt_material verbatim), copied into every shading record
};

__device__ __forceinline__ void woop_device(const float p0[3], const float p1[3], const float p2[3], float4 out[3]) {
    const double ax = p0[0], ay = p0[1], az = p0[2];
    const double e1x = (double)p1[0] - ax, e1y = (double)p1[1] - ay, e1z = (double)p1[2] - az;
    const double e2x = (double)p2[0] - ax, e2y = (double)p2[1] - ay, e2z = (double)p2[2] - az;
    const double nx = e1y * e2z - e1z * e2y, ny = e1z * e2x - e1x * e2z, nz = e1x * e2y - e1y * e2x;
    const double det = nx * nx + ny * ny + nz * nz;
    if (!(det > 0.0)) { out[0] = out[1] = out[2] = make_float4(0.f, 0.f, 0.f, 0.f); return; }
    const double inv = 1.0 / det;
    const double r0x = (e2y * nz - e2z * ny) * inv, r0y = (e2z * nx - e2x * nz) * inv, r0z = (e2x * ny - e2y * nx) * inv;
    const double r1x = (ny * e1z - nz * e1y) * inv, r1y = (nz * e1x - nx * e1z) * inv, r1z = (nx * e1y - ny * e1x) * inv;
    const double r2x = nx * inv, r2y = ny * inv, r2z = nz * inv;
    out[0] = make_float4((float)r0x, (float)r0y, (float)r0z, (float)(-(r0x * ax + r0y * ay + r0z * az)));
    out[1] = make_float4((float)r1x, (float)r1y, (float)r1z, (float)(-(r1x * ax + r1y * ay + r1z * az)));
    out[2] = make_float4((float)r2x, (float)r2y, (float)r2z, (float)(-(r2x * ax + r2y * ay + r2z * az)));
}

__global__ __launch_bounds__(256) void k_bake_instance(BakeArgs a, const float4 *obj_verts /* 2 float4 per vertex */, const uint32_t *indices,
                                                       float4 *tri_verts, float4 *woop_prim, uint32_t *bad) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= a.n_tris) return;
    const float *m = a.m;
    float P[3][3];
    for (int k = 0; k < 3; ++k) {
        const uint32_t vi = a.vertex_offset + indices[a.index_offset + 3u * t + k];
        const float4 vp = obj_verts[2u * (size_t)vi], vn = obj_verts[2u * (size_t)vi + 1];
        const float x = vp.x, y = vp.y, z = vp.z;
        float4 op;
        op.x = ((m[0] * x + m[4] * y) + m[8] * z) + m[12];
        op.y = ((m[1] * x + m[5] * y) + m[9] * z) + m[13];
        op.z = ((m[2] * x + m[6] * y) + m[10] * z) + m[14];
        op.w = vp.w;
        const float nx = vn.x, ny = vn.y, nz = vn.z;
        float n0 = (a.c[0] * nx + a.c[1] * ny) + a.c[2] * nz, n1 = (a.c[3] * nx + a.c[4] * ny) + a.c[5] * nz, n2 = (a.c[6] * nx + a.c[7] * ny) + a.c[8] * nz;
        const float l2 = (n0 * n0 + n1 * n1) + n2 * n2;
        if (!(l2 > 0.f)) { n0 = n1 = n2 = 0.f; }
        else { const float inv = 1.0f / sqrtf(l2); n0 *= inv; n1 *= inv; n2 *= inv; }
        P[k][0] = op.x; P[k][1] = op.y; P[k][2] = op.z;
        if (!(fabsf(op.x) <= 3.0e38f) || !(fabsf(op.y) <= 3.0e38f) || !(fabsf(op.z) <= 3.0e38f)) *bad = 1u;  // inf / NaN
        float4 *dst = tri_verts + kTriRec * (size_t)(a.first_tri + t) + 2u * k;
        dst[0] = op;
        dst[1] = make_float4(n0, n1, n2, vn.w);
    }
    tri_verts[kTriRec * (size_t)(a.first_tri + t) + 6u] = a.mat[0];
    tri_verts[kTriRec * (size_t)(a.first_tri + t) + 7u] = a.mat[1];
    float4 w[3];
    woop_device(P[0], P[1], P[2], w);
    // prim order: k_lbvh_scatter_woop takes the maps to the triangle's place(s) in the tree (a split triangle has several)
    const uint32_t slot = a.first_tri + t;
    woop_prim[3u * (size_t)slot] = w[0]; woop_prim[3u * (size_t)slot + 1] = w[1]; woop_prim[3u * (size_t)slot + 2] = w[2];
}

}  // namespace lptd
