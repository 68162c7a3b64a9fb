// pool_kernels.h — k_pool: every bounce of a shard-sized wavefront in ONE launch, as a CU-LOCAL POOL of trace and shade work
// (included by device.hip behind kernels.h).
//
// The reference dispatches one pass per bounce (renderer.rs:484-509); for a tile shard of a multi-GPU frame (1 M rays) the seventeen
// dependent launches of that loop spend half their time draining, and the lane-carried path kernel (k_path) loses its
// lanes to waiting: a lane whose closest hit is ready idles until its OWN wave shades a batch (lane efficiency 0.37).  Here a path is a
// 128-byte RECORD in global memory (one cache line, written and read by shading waves only), and the waves of a block hand work to
// each other through four rings in LDS:
//     FREE  (record indices)  -> a shading wave takes records for new paths (primary hits of the block's chunk of the bounce-0 queue)
//     TRACE (48-byte payload: origin, shadow direction + tmax, next direction, record index)  <- shading;  -> ANY wave's idle lanes refill
//           from it: a tracing wave reads LDS only — no global load stands between two traversal steps of its other lanes
//     SURF / OTHER (record indices)  <- tracing: the closest hit of the path's ray is known (surface / miss, emitter, path end);  -> a wave
//           without rays in flight shades 64 of ONE kind with all its lanes (north star: the sorted shade stage — the producer picks the queue)
// A lane traces the shadow ray of a bounce BEFORE the path's next ray; whether the shadow ray was unoccluded travels with the hit (the sign
// bit of t) and the shading wave adds the light sample before anything of the next bounce, so a path's radiance is summed in the order of
// the per-bounce launches and the frame is theirs bit for bit (the argument of k_path).
// No wave waits for another wave: a ring operation holds an LDS spin lock for a handful of instructions (the holder never blocks), room in
// TRACE is RESERVED before a batch is shaded (so a shaded batch can always be handed on), a wave that finds nothing to do sleeps and looks
// again, and the block ends when every record is back in FREE and the queue of primary hits is used up.  Every spin is bounded: a cap sets
// the block's abort word and the renderer's error word (-> LPT_ERR_HIP).  All traffic between waves stays inside one CU, so nothing here
// depends on visibility across XCDs.  Protocol model under ThreadSanitizer: tests/tools/pool_model.cpp.
#pragma once

namespace lptd {

constexpr uint32_t kPoolRec = 8u;             // float4 per record: 128 B, one cache line
// record: [0] origin.xyz of the pending closest-hit ray, pdf of the sampling bounce   [1] its direction, pixel slot bits
//         [2] throughput, x | y << 13 | sample << 26    [3] radiance so far, bounce of the pending hit
//         [5] light sample of the pending shadow ray    [6] the closest hit (t, u, v, prim) — by the tracing lane; sign of t: the shadow ray was unoccluded
constexpr uint32_t kPoolNextBit = 0x10000u;   // TRACE payload, word B.w: record index | kPoolNextBit (the path has a next ray)
constexpr uint32_t kPoolFinalPrim = 0xFFFFFFFEu;   // "hit" of a path that ended with its shadow ray: nothing to shade, only the light sample to add
enum { RING_FREE = 0, RING_SURF = 1, RING_OTHER = 2, RING_COUNT = 3 };
constexpr uint32_t kPoolSpinCap = 1u << 18;   // x s_sleep(1): a few milliseconds, far beyond any legitimate wait
constexpr uint32_t kPoolIdleCap = 1u << 20;   // x s_sleep(8): a wave that finds nothing to do for ~0.2 s gives up (the block's last paths take microseconds)

struct PoolRing { uint32_t lock, head, tail, reserved; };
struct PoolCtl {
    PoolRing ring[RING_COUNT];
    PoolRing trace;                                  // the payload ring; `reserved` = slots promised to shading batches in progress
    uint32_t adm_lock, adm_next, adm_end, adm_dry;   // the block's chunk of the bounce-0 queue
    uint32_t abort, pad0, pad1, pad2;
};
struct PoolArgs {
    float4 *slab;         // gridDim.x * entries * kPoolRec
    uint32_t entries;     // records per block, a power of two <= 32768
    uint32_t trace_cap;   // TRACE payload slots per block, a power of two >= 128
    uint32_t shaders;     // waves of a block that prefer shading to tracing
    int refill;           // lanes tracing at or below which a wave retires its finished rays and refills
    uint32_t chunk;       // primary hits per pull of the block
    uint32_t *error;      // the renderer's error word (page-locked host memory mapped into the device: device.hip check_device_error)
};
__host__ __device__ __forceinline__ uint32_t pool_lds_bytes(uint32_t stack_entries, uint32_t waves, uint32_t entries, uint32_t trace_cap) {
    return waves * stack_entries * kTraceBlock * 8u + 1024u + 3u * kMaxBounces * 4u + (uint32_t)sizeof(PoolCtl) + RING_COUNT * entries * 2u + trace_cap * 48u;
}

// ---- ring operations: called by EVERY lane of a wave (wave-uniform control flow); lane 0 takes the lock
__device__ __forceinline__ void pool_lock(uint32_t *lock, PoolCtl *ctl) {
    uint32_t spins = 0;
    while (atomicCAS(lock, 0u, 1u) != 0u) {
        __builtin_amdgcn_s_sleep(1);
        if (++spins > kPoolSpinCap) { atomicExch(&ctl->abort, 1u); break; }   // give up: the frame is void, every index below stays in bounds
    }
}
__device__ __forceinline__ void pool_unlock(uint32_t *lock) { __hip_atomic_store(lock, 0u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP); }
__device__ __forceinline__ uint32_t pool_bcast(uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); }

// the valid lanes append their index; the release covers the records the wave wrote before (global) and the ring slots (LDS)
__device__ __forceinline__ void pool_push(PoolCtl *ctl, uint16_t *rbuf, uint32_t P, int ring, bool valid, uint32_t idx) {
    const unsigned long long m = __ballot(valid);
    if (m == 0ull) return;
    const uint32_t lane = threadIdx.x & 63u;
    PoolRing *rg = &ctl->ring[ring];
    uint32_t t = 0;
    if (lane == 0) { pool_lock(&rg->lock, ctl); t = *(volatile uint32_t *)&rg->tail; }
    t = pool_bcast(t);
    if (valid) rbuf[(uint32_t)ring * P + ((t + (uint32_t)__popcll(m & ((1ull << lane) - 1ull))) & (P - 1u))] = (uint16_t)idx;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    if (lane == 0) { *(volatile uint32_t *)&rg->tail = t + (uint32_t)__popcll(m); pool_unlock(&rg->lock); }
}
// up to `want` (<= 64) indices; lane k < n gets the k-th.  all: `want` or nothing.  The slots are read UNDER the lock: once the head has moved and the
// lock is free another wave may take the following entries, use them and push them back into this ring — over the slots just taken, if the ring was full
// (tests/tools/pool_model.cpp under ThreadSanitizer found exactly that in the first version, which read them after the unlock)
__device__ __forceinline__ uint32_t pool_pop(PoolCtl *ctl, const uint16_t *rbuf, uint32_t P, int ring, uint32_t want, bool all, uint32_t &idx) {
    const uint32_t lane = threadIdx.x & 63u;
    PoolRing *rg = &ctl->ring[ring];
    uint32_t h = 0, n = 0;
    if (lane == 0) {
        pool_lock(&rg->lock, ctl);
        h = *(volatile uint32_t *)&rg->head;
        const uint32_t t = *(volatile uint32_t *)&rg->tail;
        n = min(want, t - h);
        if (all && n < want) n = 0u;
    }
    h = pool_bcast(h); n = pool_bcast(n);
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    if (lane < n) idx = rbuf[(uint32_t)ring * P + ((h + lane) & (P - 1u))];
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");   // every lane's slot has arrived before lane 0 lets go of the ring
    if (lane == 0) { *(volatile uint32_t *)&rg->head = h + n; pool_unlock(&rg->lock); }
    return n;
}
// entries a ring holds — a policy hint, read without the lock (tail first: never more than it held at some moment in between)
__device__ __forceinline__ uint32_t pool_count(const PoolRing *rg) {
    const uint32_t t = *(volatile const uint32_t *)&rg->tail;
    const uint32_t h = *(volatile const uint32_t *)&rg->head;
    return (t - h) > 0x7FFFFFFFu ? 0u : t - h;
}
// ---- the TRACE ring: 48-byte payloads (three float4) in LDS
// room for a whole batch (64 entries) is promised BEFORE the batch is shaded, so that a shaded batch can always be handed on
__device__ __forceinline__ bool pool_trace_reserve(PoolCtl *ctl, uint32_t cap) {
    uint32_t ok = 0;
    if ((threadIdx.x & 63u) == 0) {
        PoolRing *rg = &ctl->trace;
        pool_lock(&rg->lock, ctl);
        const uint32_t used = (*(volatile uint32_t *)&rg->tail - *(volatile uint32_t *)&rg->head) + *(volatile uint32_t *)&rg->reserved;
        if (used + 64u <= cap) { *(volatile uint32_t *)&rg->reserved = *(volatile uint32_t *)&rg->reserved + 64u; ok = 1u; }
        pool_unlock(&rg->lock);
    }
    return pool_bcast(ok) != 0u;
}
// hands on the valid lanes' payloads and gives back the batch's reservation (also with no valid lane)
__device__ __forceinline__ void pool_trace_push(PoolCtl *ctl, float4 *tbuf, uint32_t cap, bool valid, const float4 a, const float4 b, const float4 c) {
    const unsigned long long m = __ballot(valid);
    const uint32_t lane = threadIdx.x & 63u;
    PoolRing *rg = &ctl->trace;
    uint32_t t = 0;
    if (lane == 0) { pool_lock(&rg->lock, ctl); t = *(volatile uint32_t *)&rg->tail; }
    t = pool_bcast(t);
    if (valid) {
        float4 *slot = tbuf + 3u * ((t + (uint32_t)__popcll(m & ((1ull << lane) - 1ull))) & (cap - 1u));
        slot[0] = a; slot[1] = b; slot[2] = c;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    if (lane == 0) {
        *(volatile uint32_t *)&rg->tail = t + (uint32_t)__popcll(m);
        *(volatile uint32_t *)&rg->reserved = *(volatile uint32_t *)&rg->reserved - 64u;
        pool_unlock(&rg->lock);
    }
}
__device__ __forceinline__ void pool_trace_unreserve(PoolCtl *ctl) {
    if ((threadIdx.x & 63u) == 0) {
        PoolRing *rg = &ctl->trace;
        pool_lock(&rg->lock, ctl);
        *(volatile uint32_t *)&rg->reserved = *(volatile uint32_t *)&rg->reserved - 64u;
        pool_unlock(&rg->lock);
    }
}
// the idle lanes (imask) take the next payloads, in lane order; read UNDER the lock, as pool_pop.  Returns how many were taken.
__device__ __forceinline__ uint32_t pool_trace_pop(PoolCtl *ctl, const float4 *tbuf, uint32_t cap, unsigned long long imask, float4 &a, float4 &b, float4 &c) {
    const uint32_t lane = threadIdx.x & 63u;
    PoolRing *rg = &ctl->trace;
    uint32_t h = 0, n = 0;
    if (lane == 0) {
        pool_lock(&rg->lock, ctl);
        h = *(volatile uint32_t *)&rg->head;
        n = min((uint32_t)__popcll(imask), *(volatile uint32_t *)&rg->tail - h);
    }
    h = pool_bcast(h); n = pool_bcast(n);
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    const uint32_t rank = (uint32_t)__popcll(imask & ((1ull << lane) - 1ull));
    if (((imask >> lane) & 1ull) && rank < n) {
        const float4 *slot = tbuf + 3u * ((h + rank) & (cap - 1u));
        a = slot[0]; b = slot[1]; c = slot[2];
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    if (lane == 0) { *(volatile uint32_t *)&rg->head = h + n; pool_unlock(&rg->lock); }
    return n;
}
// up to 64 primary hits of the block's chunk; the block pulls chunks of `chunk` rays from the head of its XCD (ChunkPuller's rule)
__device__ __forceinline__ uint32_t pool_admit(PoolCtl *ctl, uint32_t *heads8, uint32_t count, uint32_t chunk, uint32_t &first) {
    const uint32_t lane = threadIdx.x & 63u;
    uint32_t f = 0, n = 0;
    if (lane == 0) {
        pool_lock(&ctl->adm_lock, ctl);
        uint32_t nx = *(volatile uint32_t *)&ctl->adm_next, en = *(volatile uint32_t *)&ctl->adm_end;
        if (nx == en && !*(volatile uint32_t *)&ctl->adm_dry) {
            const uint32_t home = blockIdx.x & 7u;
            const uint32_t k = atomicAdd(heads8 + home * 32u, 1u);
            const uint32_t c = home + 8u * k, n_chunks = (count + chunk - 1u) / chunk;
            if (c >= n_chunks) *(volatile uint32_t *)&ctl->adm_dry = 1u;
            else { nx = c * chunk; en = min(count, nx + chunk); *(volatile uint32_t *)&ctl->adm_end = en; }
        }
        n = min(64u, en - nx);
        f = nx;
        *(volatile uint32_t *)&ctl->adm_next = nx + n;
        pool_unlock(&ctl->adm_lock);
    }
    first = pool_bcast(f);
    return pool_bcast(n);
}
__device__ __forceinline__ bool pool_primaries_left(PoolCtl *ctl) {
    return !*(volatile uint32_t *)&ctl->adm_dry || *(volatile uint32_t *)&ctl->adm_next != *(volatile uint32_t *)&ctl->adm_end;
}

template <bool GBUF, bool STATS>
__global__ __launch_bounds__(1024) __attribute__((amdgpu_waves_per_eu(4, 4))) void k_pool(DScene sc, DProbe probe, DNoise nz, FrameParams p, Queue q0, const float4 *hits0,
                                                                                          float4 *Lsum, FrameCounters *ctr, uint32_t seed0, GBufArgs gb, PoolArgs pa) {
    const uint32_t lane = threadIdx.x & 63u, wv = threadIdx.x >> 6, n_waves = blockDim.x >> 6;
    const uint32_t stack_bytes = sc.stack_entries * kTraceBlock * (uint32_t)sizeof(uint2);
    uint2 *stack = reinterpret_cast<uint2 *>(lds_dyn + wv * stack_bytes) + lane;   // this wave's columns, stride kTraceBlock (node_visit)
    unsigned char *lp = lds_dyn + n_waves * stack_bytes;
    float *s_lut = reinterpret_cast<float *>(lp); lp += 1024u;
    uint32_t *s_cnt = reinterpret_cast<uint32_t *>(lp); lp += 3u * kMaxBounces * 4u;   // [3][kMaxBounces]: next rays, shadow rays, surface hits per bounce
    PoolCtl *ctl = reinterpret_cast<PoolCtl *>(lp); lp += sizeof(PoolCtl);
    const uint32_t P = pa.entries, TC = pa.trace_cap;
    float4 *tbuf = reinterpret_cast<float4 *>(lp); lp += TC * 48u;
    uint16_t *rbuf = reinterpret_cast<uint16_t *>(lp);
    float4 *slab = pa.slab + (size_t)blockIdx.x * P * kPoolRec;
    for (uint32_t i = threadIdx.x; i < 256u; i += blockDim.x) s_lut[i] = sc.srgb_lut[i];
    for (uint32_t i = threadIdx.x; i < 3u * kMaxBounces; i += blockDim.x) s_cnt[i] = 0u;
    for (uint32_t i = threadIdx.x; i < P; i += blockDim.x) rbuf[RING_FREE * P + i] = (uint16_t)i;
    if (threadIdx.x < sizeof(PoolCtl) / 4u) reinterpret_cast<uint32_t *>(ctl)[threadIdx.x] = 0u;
    __syncthreads();
    if (threadIdx.x == 0) ctl->ring[RING_FREE].tail = P;
    __syncthreads();

    const uint32_t nb = p.max_bounces, count0 = QC(ctr, 0);
    const float inv_nl = sc.n_lights ? 1.0f / (float)sc.n_lights : 0.0f;
    const bool shader_first = wv < pa.shaders;
    uint32_t n_nodes = 0, n_tris = 0, s_nodes = 0, s_tris = 0;
    uint32_t w_steps = 0, w_live = 0, w_node = 0, w_tri = 0;
    RayState rs;
    ray_begin(rs, mk3(0.f, 0.f, 0.f), mk3(0.f, 0.f, 1.f), 0.0f);
    // lane: 0 idle, 1 traces the path's shadow ray, 2 traces its closest-hit ray, 4 done with the path's rays of this bounce (waits for the wave's next retire point)
    // e: record index | kPoolNextBit (the path has a next ray) | bit 17 (its shadow ray of this bounce was unoccluded)
    uint32_t st = 0u, e = 0u, idle_spins = 0u;
    f3 nd = mk3(0.f, 0.f, 1.f);
    for (;;) {
        if (*(volatile uint32_t *)&ctl->abort) break;
        const int n_active = __popcll(__ballot(st == 1u || st == 2u));
        if (n_active <= pa.refill) {
            // ---- retire: the lane's result goes into the path's record (ONE 16-byte store, no load) and the record to the ring of its kind
            bool to_surf = false, to_other = false;
            if (st == 4u) {
                const bool unocc = (e & 0x20000u) != 0u;
                float4 h4;
                if (e & kPoolNextBit) {
                    intersect_lights(sc, rs.o, rs.d, rs.best);
                    h4 = make_float4(unocc ? -rs.best.t : rs.best.t, rs.best.u, rs.best.v, __uint_as_float(rs.best.prim));   // t > 0 always: its sign carries the shadow result
                    to_surf = rs.best.prim != 0xFFFFFFFFu && !(rs.best.prim & LPT_LIGHT_BIT);
                } else h4 = make_float4(unocc ? -1.0f : 1.0f, 0.f, 0.f, __uint_as_float(kPoolFinalPrim));   // the path ended with its shadow ray
                to_other = !to_surf;
                slab[(size_t)(e & 0xFFFFu) * kPoolRec + 6u] = h4;
                st = 0u;
            }
            pool_push(ctl, rbuf, P, RING_SURF, to_surf, e & 0xFFFFu);
            pool_push(ctl, rbuf, P, RING_OTHER, to_other, e & 0xFFFFu);
            // ---- refill the idle lanes from the TRACE ring (LDS only).  A wave that prefers shading leaves it alone while there is something to shade and
            // room to hand the results on
            const unsigned long long imask = __ballot(st == 0u);
            bool take = imask != 0ull;
            if (take && shader_first) {
                const bool work = pool_count(&ctl->ring[RING_SURF]) + pool_count(&ctl->ring[RING_OTHER]) != 0u || (pool_primaries_left(ctl) && pool_count(&ctl->ring[RING_FREE]) >= 64u);
                const bool room = pool_count(&ctl->trace) + *(volatile uint32_t *)&ctl->trace.reserved + 64u <= TC;
                take = !(work && room);
            }
            if (take) {
                float4 a, b, c;
                const uint32_t n = pool_trace_pop(ctl, tbuf, TC, imask, a, b, c);
                const uint32_t rank = (uint32_t)__popcll(imask & ((1ull << lane) - 1ull));
                if (st == 0u && rank < n) {
                    e = __float_as_uint(b.w);
                    nd = mk3(c.x, c.y, c.z);
                    if (a.w > 0.0f) { ray_begin(rs, mk3(a.x, a.y, a.z), mk3(b.x, b.y, b.z), a.w); st = 1u; }
                    else { ray_begin(rs, mk3(a.x, a.y, a.z), nd, LPT_T_INF); st = 2u; }
                }
            }
        }
        if (__ballot(st == 1u || st == 2u) == 0ull) {
            // ---- no ray in flight in this wave (every finished one was retired above): shade a batch with all 64 lanes, if its results can be handed on
            // source: a full batch of surface hits, a full batch of misses / emitter hits / path ends, new paths (primary hits) while records are free,
            // then whatever is there
            uint32_t got = 0u, n = 0u, first = 0u;
            bool fresh = false;
            const bool room = pool_trace_reserve(ctl, TC);
            if (room) {
                const uint32_t c_surf = pool_count(&ctl->ring[RING_SURF]), c_other = pool_count(&ctl->ring[RING_OTHER]);
                if (c_surf >= 64u) n = pool_pop(ctl, rbuf, P, RING_SURF, 64u, false, got);
                else if (c_other >= 64u) n = pool_pop(ctl, rbuf, P, RING_OTHER, 64u, false, got);
                if (n == 0u && pool_primaries_left(ctl)) {
                    if (pool_pop(ctl, rbuf, P, RING_FREE, 64u, true, got) == 64u) {   // a record for every lane first: a path that has been shaded cannot be dropped
                        n = pool_admit(ctl, &ctr->phead[0], count0, pa.chunk, first);
                        fresh = true;
                        if (n == 0u) { pool_push(ctl, rbuf, P, RING_FREE, true, got); fresh = false; }
                    }
                }
                if (n == 0u && c_surf) n = pool_pop(ctl, rbuf, P, RING_SURF, 64u, false, got);
                if (n == 0u && c_other) n = pool_pop(ctl, rbuf, P, RING_OTHER, 64u, false, got);
                if (n == 0u) pool_trace_unreserve(ctl);
            }
            if (n == 0u) {
                // nothing to shade (or no room to hand results on: then TRACE is not empty and the refill above finds work next time round)
                if (!pool_primaries_left(ctl) && pool_count(&ctl->ring[RING_FREE]) == P) break;   // every record is free and no primary hit is left: done
                __builtin_amdgcn_s_sleep(8);
                if (++idle_spins > kPoolIdleCap) { atomicExch(&ctl->abort, 2u); break; }
                continue;
            }
            idle_spins = 0u;
            e = got;                      // fresh: the record reserved for the lane
            const bool mine = lane < n;
            float4 *E = slab + (size_t)e * kPoolRec;
            float4 d4 = make_float4(0.f, 0.f, 1.f, 0.f), T4 = d4, h4 = d4, L4 = make_float4(0.f, 0.f, 0.f, 0.f), c5 = L4;
            if (mine) {
                if (fresh) {
                    const uint32_t idx = first + lane;
                    d4 = ld_nt(q0.d + idx);
                    T4 = make_float4(1.f, 1.f, 1.f, q0.T[idx].w);
                    if (hits0) h4 = ld_nt(hits0 + idx);
                } else { d4 = E[1]; T4 = E[2]; L4 = E[3]; c5 = E[5]; h4 = E[6]; }
            }
            const bool trace_primary = fresh && !hits0;   // wave-uniform: no packet launch ran, the path starts with its primary ray
            const uint32_t bounce = fresh ? 0u : (__float_as_uint(L4.w) & 0xFFu);
            bool cont = false;
            float4 pa4 = make_float4(0.f, 0.f, 0.f, -1.0f), pb4 = pa4, pc4 = pa4;   // the TRACE payload
            if (trace_primary) {
                if (mine) {
                    E[0] = make_float4(p.origin.x, p.origin.y, p.origin.z, -1.0f);
                    E[1] = d4; E[2] = T4;
                    E[3] = make_float4(0.f, 0.f, 0.f, __uint_as_float(0u));
                    pa4 = make_float4(p.origin.x, p.origin.y, p.origin.z, -1.0f);
                    pb4 = make_float4(0.f, 0.f, 1.f, __uint_as_float(e | kPoolNextBit));
                    pc4 = make_float4(d4.x, d4.y, d4.z, 0.f);
                    cont = true;
                }
            } else if (mine) {
                f3 L = mk3(L4.x, L4.y, L4.z);
                const bool unocc = !fresh && (__float_as_uint(h4.x) >> 31) != 0u;
                h4.x = fabsf(h4.x);
                if (unocc) { L.x = L.x + c5.x; L.y = L.y + c5.y; L.z = L.z + c5.z; }   // the light sample of the bounce before this hit: before anything of this one
                const uint32_t prim = __float_as_uint(h4.w);
                if (prim == kPoolFinalPrim) {
                    Lsum[__float_as_uint(d4.w)] = make_float4(L.x, L.y, L.z, 0.0f);
                } else {
                    ShadeOut so;
                    shade_hit<GBUF>(sc, probe, nz, p, s_lut, bounce, bounce + 1u >= nb, seed0 + bounce + 1u, inv_nl, gb, d4, T4, h4,
                                    [&]() { return fresh ? make_float4(p.origin.x, p.origin.y, p.origin.z, -1.0f) : E[0]; },
                                    [&](float r, float g, float b) { L.x = L.x + r; L.y = L.y + g; L.z = L.z + b; }, so);
                    if (so.is_surface) atomicAdd(&s_cnt[128u + bounce], 1u);
                    if (so.want_shadow) atomicAdd(&s_cnt[64u + bounce], 1u);
                    if (so.want_next) atomicAdd(&s_cnt[bounce + 1u], 1u);
                    cont = so.want_shadow || so.want_next;
                    if (cont) {
                        // both rays of a bounce leave the same point: so4.xyz == no4.xyz (shade_hit's Po)
                        if (so.want_next) { E[0] = so.no4; E[2] = so.nT4; }
                        E[1] = so.want_next ? so.nd4 : d4;   // .w = the pixel slot either way (a new path's record holds another path's leftovers)
                        E[3] = make_float4(L.x, L.y, L.z, __uint_as_float(so.want_next ? bounce + 1u : bounce));
                        if (so.want_shadow) E[5] = so.sc4;
                        const float4 po = so.want_next ? so.no4 : so.so4;
                        pa4 = make_float4(po.x, po.y, po.z, so.want_shadow ? so.so4.w : -1.0f);
                        pb4 = make_float4(so.sd4.x, so.sd4.y, so.sd4.z, __uint_as_float(e | (so.want_next ? kPoolNextBit : 0u)));
                        pc4 = make_float4(so.nd4.x, so.nd4.y, so.nd4.z, 0.f);
                    } else {
                        Lsum[__float_as_uint(d4.w)] = make_float4(L.x, L.y, L.z, 0.0f);
                    }
                }
            }
            pool_trace_push(ctl, tbuf, TC, cont, pa4, pb4, pc4);                 // also gives back the batch's reservation
            pool_push(ctl, rbuf, P, RING_FREE, !cont && (mine || fresh), e);   // ended paths; the records a short batch of new paths did not need
            ray_begin(rs, mk3(0.f, 0.f, 0.f), mk3(0.f, 0.f, 1.f), 0.0f);       // nothing of the traversal state lives across a shading batch
            st = 0u;
            continue;
        }
        idle_spins = 0u;
        uint32_t dn = 0, dt = 0;
        if (STATS) {
            w_steps++;
            w_live += (uint32_t)__popcll(__ballot(st == 1u || st == 2u));
            w_node += (uint32_t)__popcll(__ballot((st == 1u || st == 2u) && rs.tg2.y == 0u && ((rs.ng.y & 0xFF000000u) != 0u || rs.sp != 0)));
        }
        const bool was_shadow = st == 1u;
        if ((st == 1u || st == 2u) && ray_step_pipe<STATS>(sc, rs, stack, st == 1u, dn, dt)) {
            if (st == 1u) {
                if (rs.best.prim == 0xFFFFFFFFu) e |= 0x20000u;   // unoccluded: the shading wave adds the light sample
                if (e & kPoolNextBit) { ray_begin(rs, rs.o, nd, LPT_T_INF); st = 2u; }   // the next ray leaves the point the shadow ray left
                else st = 4u;
            } else st = 4u;
        }
        if (STATS) {
            w_tri += (uint32_t)__popcll(__ballot(dt != 0u));
            if (was_shadow) { s_nodes += dn; s_tris += dt; } else { n_nodes += dn; n_tris += dt; }
        }
    }
    __syncthreads();
    if (threadIdx.x == 0 && *(volatile uint32_t *)&ctl->abort) *(volatile uint32_t *)pa.error = 0x100u + *(volatile uint32_t *)&ctl->abort;   // page-locked host memory: a plain store, no atomic over PCIe
    if (threadIdx.x >= 1u && threadIdx.x <= nb && s_cnt[threadIdx.x]) atomicAdd(&QC(ctr, threadIdx.x), s_cnt[threadIdx.x]);
    if (threadIdx.x < nb) {
        if (s_cnt[64u + threadIdx.x]) atomicAdd(&SC(ctr, threadIdx.x), s_cnt[64u + threadIdx.x]);
        if (s_cnt[128u + threadIdx.x]) atomicAdd(&ctr->shaded[threadIdx.x], s_cnt[128u + threadIdx.x]);
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

}  // namespace lptd
