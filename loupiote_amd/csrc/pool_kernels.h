// pool_kernels.h — k_pool: every bounce of a shard-sized wavefront in ONE launch, as a CU-LOCAL POOL of trace and shade work
// (included by device.hip behind kernels.h).
//
// The reference dispatches one pass per bounce (renderer.rs:484-509); for a tile shard of a multi-GPU frame (1 M rays) the seventeen
// dependent launches of that loop spend half their time draining, and the lane-carried path kernel (k_path) loses its
// lanes to waiting: a lane whose closest hit is ready idles until its OWN wave shades a batch (lane efficiency 0.37).  Here the unit
// that carries a path is a 112-byte RECORD in global memory, and the waves of a block — one block of up to 16 waves per CU — hand
// records to each other through four rings of record indices in LDS:
//     FREE  -> a shading wave takes records for new paths (primary hits of the block's chunk of the bounce-0 queue)
//     TRACE <- shading: the path has a shadow ray and / or a next ray;  -> ANY wave's idle lanes refill from it
//     SURF / OTHER <- tracing: the closest hit of the path's ray is known (surface hit / miss or emitter);  -> a wave without rays in
//              flight shades 64 of them with all its lanes — the shading input arrives grouped by kind (north star: the sorted shade stage)
// A lane traces the shadow ray of a bounce BEFORE the path's next ray and adds the light sample in between, so a path's radiance is
// summed in the order of the per-bounce launches and the frame is theirs bit for bit (the argument of k_path).
// No wave waits for another wave: a ring operation holds an LDS spin lock for a handful of instructions (the holder never blocks),
// a wave that finds nothing to do sleeps and looks again, and the block ends when every record is back in FREE and the queue of
// primary hits is used up.  Every spin is bounded: a cap sets the block's abort word and the frame's error word (-> LPT_ERR_HIP).
// All traffic between waves stays inside one CU (the records live in a per-block slab that only this block touches: L1 / L2),
// so nothing here depends on visibility across XCDs.  Protocol model under ThreadSanitizer: tests/tools/pool_model.cpp.
#pragma once

namespace lptd {

constexpr uint32_t kPoolRec = 7u;             // float4 per record
// record: [0] origin.xyz, pdf of the sampling bounce   [1] direction of the pending closest-hit ray, pixel slot bits
//         [2] throughput, x | y << 13 | sample << 26    [3] radiance so far, state (bounce | kPoolShadow | kPoolNext)
//         [4] shadow direction, tmax                    [5] light sample the shadow ray carries   [6] the closest hit (t, u, v, prim)
constexpr uint32_t kPoolShadow = 0x100u, kPoolNext = 0x200u;
enum { RING_FREE = 0, RING_TRACE = 1, RING_SURF = 2, RING_OTHER = 3, RING_COUNT = 4 };
constexpr uint32_t kPoolSpinCap = 1u << 18;   // x s_sleep(1): a few milliseconds, far beyond any legitimate wait
constexpr uint32_t kPoolIdleCap = 1u << 20;   // x s_sleep(8): a wave that finds nothing to do for ~0.2 s gives up (the block's last paths take microseconds)

struct PoolRing { uint32_t lock, head, tail, pad; };
struct PoolCtl {
    PoolRing ring[RING_COUNT];
    uint32_t adm_lock, adm_next, adm_end, adm_dry;   // the block's chunk of the bounce-0 queue
    uint32_t abort, pad0, pad1, pad2;
};
struct PoolArgs {
    float4 *slab;         // gridDim.x * entries * kPoolRec
    uint32_t entries;     // records per block, a power of two <= 65536
    uint32_t shaders;     // waves of a block that prefer shading to tracing
    int refill;           // lanes tracing at or below which a wave retires its finished rays and refills
    uint32_t chunk;       // primary hits per pull of the block
    uint32_t *error;      // the renderer's error word (page-locked host memory mapped into the device: device.hip check_device_error)
};
__host__ __device__ __forceinline__ uint32_t pool_lds_bytes(uint32_t stack_entries, uint32_t waves, uint32_t entries) {
    return waves * stack_entries * kTraceBlock * 8u + 1024u + 3u * kMaxBounces * 4u + (uint32_t)sizeof(PoolCtl) + RING_COUNT * entries * 2u;
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
__device__ __forceinline__ uint32_t pool_count(PoolCtl *ctl, int ring) {
    const uint32_t t = *(volatile uint32_t *)&ctl->ring[ring].tail;
    const uint32_t h = *(volatile uint32_t *)&ctl->ring[ring].head;
    return (t - h) > 0x7FFFFFFFu ? 0u : t - h;
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
    uint16_t *rbuf = reinterpret_cast<uint16_t *>(lp);
    const uint32_t P = pa.entries;
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
    // lane: 0 idle, 1 traces the path's shadow ray, 2 traces its closest-hit ray, 3 / 4 that ray is finished and waits for the wave's next retire point
    uint32_t st = 0u, e = 0u, idle_spins = 0u;
    for (;;) {
        if (*(volatile uint32_t *)&ctl->abort) break;
        const int n_active = __popcll(__ballot(st == 1u || st == 2u));
        if (n_active <= pa.refill) {
            // ---- retire: the light sample of an unoccluded shadow ray, then the path's next ray; a closest hit goes to the shading rings
            bool to_free = false, to_surf = false, to_other = false;
            if (st == 3u) {
                float4 *E = slab + (size_t)e * kPoolRec;
                float4 e3 = E[3];
                const float4 e1 = E[1];
                const uint32_t state = __float_as_uint(e3.w);
                if (rs.best.prim == 0xFFFFFFFFu) {
                    const float4 c = E[5];
                    e3.x = e3.x + c.x; e3.y = e3.y + c.y; e3.z = e3.z + c.z;
                    if (state & kPoolNext) E[3] = e3;
                }
                if (state & kPoolNext) {
                    ray_begin(rs, rs.o, mk3(e1.x, e1.y, e1.z), LPT_T_INF);   // the next ray leaves the point the shadow ray left
                    st = 2u;
                } else {
                    Lsum[__float_as_uint(e1.w)] = make_float4(e3.x, e3.y, e3.z, 0.0f);
                    to_free = true;
                    st = 0u;
                }
            } else if (st == 4u) {
                intersect_lights(sc, rs.o, rs.d, rs.best);
                slab[(size_t)e * kPoolRec + 6u] = make_float4(rs.best.t, rs.best.u, rs.best.v, __uint_as_float(rs.best.prim));
                const bool surf = rs.best.prim != 0xFFFFFFFFu && !(rs.best.prim & LPT_LIGHT_BIT);
                to_surf = surf; to_other = !surf;
                st = 0u;
            }
            pool_push(ctl, rbuf, P, RING_SURF, to_surf, e);
            pool_push(ctl, rbuf, P, RING_OTHER, to_other, e);
            pool_push(ctl, rbuf, P, RING_FREE, to_free, e);
            // ---- refill the idle lanes from the TRACE ring.  A wave that prefers shading leaves it alone while there is something to shade
            const unsigned long long imask = __ballot(st == 0u);
            const uint32_t n_idle = (uint32_t)__popcll(imask);
            bool take = n_idle != 0u;
            if (take && shader_first)
                take = pool_count(ctl, RING_SURF) + pool_count(ctl, RING_OTHER) == 0u && !(pool_primaries_left(ctl) && pool_count(ctl, RING_FREE) >= 64u);
            if (take) {
                uint32_t got = 0u;
                const uint32_t n = pool_pop(ctl, rbuf, P, RING_TRACE, n_idle, false, got);
                const uint32_t rank = (uint32_t)__popcll(imask & ((1ull << lane) - 1ull));
                const uint32_t mine = (uint32_t)__shfl((int)got, (int)(rank & 63u));
                if (st == 0u && rank < n) {
                    e = mine;
                    const float4 *E = slab + (size_t)e * kPoolRec;
                    const float4 o4 = E[0];
                    const uint32_t state = __float_as_uint(E[3].w);
                    if (state & kPoolShadow) {
                        const float4 s4 = E[4];
                        ray_begin(rs, mk3(o4.x, o4.y, o4.z), mk3(s4.x, s4.y, s4.z), s4.w);
                        st = 1u;
                    } else {
                        const float4 d4 = E[1];
                        ray_begin(rs, mk3(o4.x, o4.y, o4.z), mk3(d4.x, d4.y, d4.z), LPT_T_INF);
                        st = 2u;
                    }
                }
            }
        }
        if (__ballot(st == 1u || st == 2u) == 0ull) {
            // ---- no ray in flight in this wave (every finished one was retired above): shade a batch with all 64 lanes
            // source: a full batch of surface hits, a full batch of misses / emitter hits, new paths (primary hits) while records are free,
            // then whatever is there
            uint32_t got = 0u, n = 0u, first = 0u;
            bool fresh = false;
            const uint32_t c_surf = pool_count(ctl, RING_SURF), c_other = pool_count(ctl, RING_OTHER);
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
            if (n == 0u) {
                // nothing to shade, nothing to trace: done when every record is free and no primary hit is left
                if (!pool_primaries_left(ctl) && pool_count(ctl, RING_FREE) == P) break;
                __builtin_amdgcn_s_sleep(8);
                if (++idle_spins > kPoolIdleCap) { atomicExch(&ctl->abort, 2u); break; }
                continue;
            }
            idle_spins = 0u;
            e = got;                      // fresh: the record reserved for the lane
            const bool mine = lane < n;
            float4 *E = slab + (size_t)e * kPoolRec;
            float4 d4 = make_float4(0.f, 0.f, 1.f, 0.f), T4 = d4, h4 = d4, L4 = make_float4(0.f, 0.f, 0.f, 0.f);
            bool trace_primary = false;   // wave-uniform: no packet launch ran, the path starts with its primary ray
            if (mine) {
                if (fresh) {
                    const uint32_t idx = first + lane;
                    d4 = ld_nt(q0.d + idx);
                    T4 = make_float4(1.f, 1.f, 1.f, q0.T[idx].w);
                    if (hits0) h4 = ld_nt(hits0 + idx);
                } else { d4 = E[1]; T4 = E[2]; L4 = E[3]; h4 = E[6]; }
            }
            trace_primary = fresh && !hits0;
            const uint32_t bounce = fresh ? 0u : (__float_as_uint(L4.w) & 0xFFu);
            bool cont = false;
            if (trace_primary) {
                if (mine) {
                    E[0] = make_float4(p.origin.x, p.origin.y, p.origin.z, -1.0f);
                    E[1] = d4; E[2] = T4;
                    E[3] = make_float4(0.f, 0.f, 0.f, __uint_as_float(kPoolNext));
                    cont = true;
                }
            } else if (mine) {
                f3 L = mk3(L4.x, L4.y, L4.z);
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
                    E[0] = so.want_next ? so.no4 : make_float4(so.so4.x, so.so4.y, so.so4.z, 0.0f);
                    E[1] = so.want_next ? so.nd4 : d4;   // .w = the pixel slot either way (a new path's record holds another path's leftovers)
                    if (so.want_next) E[2] = so.nT4;
                    if (so.want_shadow) { E[4] = so.so4; E[5] = so.sc4; }
                    E[3] = make_float4(L.x, L.y, L.z, __uint_as_float((so.want_next ? bounce + 1u : bounce) | (so.want_shadow ? kPoolShadow : 0u) | (so.want_next ? kPoolNext : 0u)));
                } else {
                    Lsum[__float_as_uint(d4.w)] = make_float4(L.x, L.y, L.z, 0.0f);
                }
            }
            pool_push(ctl, rbuf, P, RING_TRACE, cont, e);
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
        if ((st == 1u || st == 2u) && ray_step_pipe<STATS>(sc, rs, stack, st == 1u, dn, dt)) st += 2u;
        if (STATS) {
            w_tri += (uint32_t)__popcll(__ballot(dt != 0u));
            if (st == 1u || st == 3u) { s_nodes += dn; s_tris += dt; } else { n_nodes += dn; n_tris += dt; }
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
