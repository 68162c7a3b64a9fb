// pool_model.cpp — CPU model of k_pool's hand-over protocol (loupiote_amd/csrc/pool_kernels.h), run under ThreadSanitizer by tests/test_pool_model.py.
// Test infrastructure: a restatement of the PROTOCOL (rings of record indices behind spin locks, the block's admission word, the end condition), with
// std::atomic where the kernel uses LDS atomics / volatile words under a lock and PLAIN memory where the kernel uses plain LDS / global accesses — so that
// TSan checks exactly what the kernel relies on: every plain access to a ring slot or a path record is ordered by the locks' acquire / release chain.
// A "wave" is a thread with 64 lane states; a "block" is a group of waves around one set of rings; "tracing" a ray takes a pseudo-random number of steps;
// "shading" decides pseudo-randomly (from the path id and bounce) whether the path has a shadow ray and / or a next ray.  Checked at the end: every path
// finished exactly once with the radiance its own serial evaluation gives (the shadow deposit of bounce b before anything of bounce b + 1), every record
// is back in FREE, no index was ever in two places.
//   g++ -O1 -g -std=c++17 -fsanitize=thread -pthread tests/tools/pool_model.cpp -o pool_model && ./pool_model [blocks waves records paths shaders refill trace_slots]
#include <atomic>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <thread>
#include <vector>

enum { RING_FREE = 0, RING_SURF = 1, RING_OTHER = 2, RING_COUNT = 3, RING_TRACE = 3 };   // TRACE: the payload ring (its own buffer), index 3 for the checker only
static const uint32_t kNextBit = 0x10000u, kUnoccBit = 0x20000u, kSpinCap = 1u << 26;

struct Ring { std::atomic<uint32_t> lock{0}, head{0}, tail{0}, reserved{0}; };
struct Record { uint32_t path, bounce, hit_kind, hit_unocc, hit_final; uint64_t L, contrib; uint32_t owner; };   // plain memory (global in the kernel)
struct Payload { uint32_t rec_flags, shadow_steps, next_steps, occluded; };   // plain memory (LDS in the kernel): what a tracing lane needs, nothing else
struct Block {
    Ring ring[RING_COUNT], trace;
    std::atomic<uint32_t> adm_lock{0}, adm_next{0}, adm_end{0}, adm_dry{0}, abort_{0};
    std::vector<uint16_t> rbuf;     // RING_COUNT * P, plain
    std::vector<Payload> tbuf;      // TC, plain
    uint32_t TC = 0;
    std::vector<Record> slab;       // P, plain
    std::vector<std::atomic<int>> where;   // checker only: the ring a record's index is in, -1 = in a wave's hands
    uint32_t P = 0;
};
struct Shared {
    std::atomic<uint32_t> head{0};                 // the global chunk head (one "XCD")
    uint32_t count = 0, chunk = 0, nb = 8;
    std::vector<uint64_t> result;                  // per path, plain: written once by whoever ends the path
    std::vector<std::atomic<uint32_t>> done;       // per path: times finished
    explicit Shared(uint32_t n) : result(n), done(n) {}
};

static inline uint32_t hash(uint32_t v) { v ^= v >> 16; v *= 0x7feb352du; v ^= v >> 15; v *= 0x846ca68bu; v ^= v >> 16; return v; }
// what "shading" bounce b of path p yields: (has_shadow, has_next, steps of each ray, kind of the next hit, the two radiance terms)
struct Shade { bool shadow, next; uint32_t ssteps, nsteps, kind; uint64_t add, contrib; };
static Shade shade(uint32_t p, uint32_t b, uint32_t nb) {
    const uint32_t h = hash(p * 97u + b * 7919u + 1u);
    Shade s;
    s.shadow = (h & 3u) != 0u;
    s.next = b + 1u < nb && ((h >> 2) & 7u) != 0u;
    s.ssteps = 1u + ((h >> 5) & 15u);
    s.nsteps = 1u + ((h >> 9) & 31u) + (((h >> 14) & 255u) == 0u ? 300u : 0u);   // a rare long ray
    s.kind = (h >> 22) & 3u ? RING_SURF : RING_OTHER;
    s.add = (h >> 3) | 1u;
    s.contrib = ((uint64_t)hash(h) << 8) | 1u;
    return s;
}
// order-sensitive accumulation (stands for the fp32 sums): L' = L * 31 + term
static inline uint64_t acc(uint64_t L, uint64_t term) { return L * 31u + term; }
static uint64_t serial(uint32_t p, uint32_t nb) {
    uint64_t L = 0;
    for (uint32_t b = 0;; ++b) {
        const Shade s = shade(p, b, nb);
        L = acc(L, s.add);
        if (s.shadow && (s.contrib & 0x100u)) L = acc(L, s.contrib);   // "unoccluded"
        if (!s.next) return L;
    }
}

static void lock(std::atomic<uint32_t> &l, Block &B) {
    uint32_t spins = 0, z = 0;
    while (!l.compare_exchange_weak(z, 1u, std::memory_order_acquire, std::memory_order_relaxed)) {
        z = 0;
        if (++spins > kSpinCap) { B.abort_.store(1u); break; }
        std::this_thread::yield();
    }
}
static void unlock(std::atomic<uint32_t> &l) { l.store(0u, std::memory_order_release); }

static void push(Block &B, int ring, const bool *valid, const uint32_t *idx) {
    uint32_t n = 0;
    for (int l = 0; l < 64; ++l) n += valid[l];
    if (!n) return;
    Ring &rg = B.ring[ring];
    lock(rg.lock, B);
    const uint32_t t = rg.tail.load(std::memory_order_relaxed);
    uint32_t k = 0;
    for (int l = 0; l < 64; ++l) if (valid[l]) {
        const int was = B.where[idx[l]].exchange(ring);
        if (was != -1) { printf("push of record %u into ring %d: it was in ring %d\n", idx[l], ring, was); B.abort_.store(5u); }
        B.rbuf[ring * B.P + ((t + k++) & (B.P - 1u))] = (uint16_t)idx[l];
    }
    rg.tail.store(t + n, std::memory_order_relaxed);
    unlock(rg.lock);
}
static uint32_t pop(Block &B, int ring, uint32_t want, bool all, uint32_t *idx) {
    Ring &rg = B.ring[ring];
    lock(rg.lock, B);
    const uint32_t h = rg.head.load(std::memory_order_relaxed), t = rg.tail.load(std::memory_order_relaxed);
    uint32_t n = want < t - h ? want : t - h;
    if (all && n < want) n = 0;
    // the slots are read UNDER the lock: once the head has moved and the lock is free, another wave may take the following entries, use them and push them
    // back into this ring — over the slots just taken, if the ring was full (the first version of the kernel read them after the unlock: TSan found it)
    for (uint32_t k = 0; k < n; ++k) {
        idx[k] = B.rbuf[ring * B.P + ((h + k) & (B.P - 1u))];
        const int was = B.where[idx[k]].exchange(-1);
        if (was != ring) { printf("pop of record %u from ring %d: it was in %d\n", idx[k], ring, was); B.abort_.store(5u); }
    }
    rg.head.store(h + n, std::memory_order_relaxed);
    unlock(rg.lock);
    return n;
}
static uint32_t count(Block &B, int ring) {
    const uint32_t t = B.ring[ring].tail.load(std::memory_order_relaxed), h = B.ring[ring].head.load(std::memory_order_relaxed);
    return (t - h) > 0x7FFFFFFFu ? 0u : t - h;
}
static uint32_t admit(Block &B, Shared &S, uint32_t &first) {
    lock(B.adm_lock, B);
    uint32_t nx = B.adm_next.load(std::memory_order_relaxed), en = B.adm_end.load(std::memory_order_relaxed);
    if (nx == en && !B.adm_dry.load(std::memory_order_relaxed)) {
        const uint32_t c = S.head.fetch_add(1u), n_chunks = (S.count + S.chunk - 1u) / S.chunk;
        if (c >= n_chunks) B.adm_dry.store(1u, std::memory_order_relaxed);
        else { nx = c * S.chunk; en = nx + S.chunk < S.count ? nx + S.chunk : S.count; B.adm_end.store(en, std::memory_order_relaxed); }
    }
    const uint32_t n = en - nx < 64u ? en - nx : 64u;
    first = nx;
    B.adm_next.store(nx + n, std::memory_order_relaxed);
    unlock(B.adm_lock);
    return n;
}
static bool primaries_left(Block &B) { return !B.adm_dry.load(std::memory_order_relaxed) || B.adm_next.load(std::memory_order_relaxed) != B.adm_end.load(std::memory_order_relaxed); }

// ---- the TRACE ring: payloads; room for a whole batch is reserved before the batch is shaded
static bool trace_reserve(Block &B) {
    Ring &rg = B.trace;
    lock(rg.lock, B);
    const uint32_t used = (rg.tail.load(std::memory_order_relaxed) - rg.head.load(std::memory_order_relaxed)) + rg.reserved.load(std::memory_order_relaxed);
    const bool ok = used + 64u <= B.TC;
    if (ok) rg.reserved.store(rg.reserved.load(std::memory_order_relaxed) + 64u, std::memory_order_relaxed);
    unlock(rg.lock);
    return ok;
}
static void trace_unreserve(Block &B) {
    Ring &rg = B.trace;
    lock(rg.lock, B);
    rg.reserved.store(rg.reserved.load(std::memory_order_relaxed) - 64u, std::memory_order_relaxed);
    unlock(rg.lock);
}
static void trace_push(Block &B, const bool *valid, const Payload *pl) {
    Ring &rg = B.trace;
    lock(rg.lock, B);
    const uint32_t t = rg.tail.load(std::memory_order_relaxed);
    uint32_t k = 0;
    for (int l = 0; l < 64; ++l) if (valid[l]) {
        const int was = B.where[pl[l].rec_flags & 0xFFFFu].exchange(RING_TRACE);
        if (was != -1) { printf("push of record %u into TRACE: it was in ring %d\n", pl[l].rec_flags & 0xFFFFu, was); B.abort_.store(5u); }
        B.tbuf[(t + k++) & (B.TC - 1u)] = pl[l];
    }
    if (t + k - rg.head.load(std::memory_order_relaxed) > B.TC) { printf("TRACE overflow\n"); B.abort_.store(6u); }
    rg.tail.store(t + k, std::memory_order_relaxed);
    rg.reserved.store(rg.reserved.load(std::memory_order_relaxed) - 64u, std::memory_order_relaxed);
    unlock(rg.lock);
}
static uint32_t trace_pop(Block &B, uint32_t want, Payload *pl) {
    Ring &rg = B.trace;
    lock(rg.lock, B);
    const uint32_t h = rg.head.load(std::memory_order_relaxed), t = rg.tail.load(std::memory_order_relaxed);
    const uint32_t n = want < t - h ? want : t - h;
    for (uint32_t k = 0; k < n; ++k) {   // under the lock
        pl[k] = B.tbuf[(h + k) & (B.TC - 1u)];
        const int was = B.where[pl[k].rec_flags & 0xFFFFu].exchange(-1);
        if (was != RING_TRACE) { printf("pop of record %u from TRACE: it was in %d\n", pl[k].rec_flags & 0xFFFFu, was); B.abort_.store(5u); }
    }
    rg.head.store(h + n, std::memory_order_relaxed);
    unlock(rg.lock);
    return n;
}
static uint32_t ring_count(Ring &rg) {
    const uint32_t t = rg.tail.load(std::memory_order_relaxed), h = rg.head.load(std::memory_order_relaxed);
    return (t - h) > 0x7FFFFFFFu ? 0u : t - h;
}

static void wave(Block &B, Shared &S, uint32_t wv, uint32_t shaders, int refill) {
    const bool shader_first = wv < shaders;
    uint32_t st[64] = {0}, e[64] = {0}, steps[64] = {0}, nsteps[64] = {0};
    bool occluded[64] = {false};
    uint32_t idle_spins = 0;
    for (;;) {
        if (B.abort_.load(std::memory_order_relaxed)) break;
        int n_active = 0;
        for (int l = 0; l < 64; ++l) n_active += st[l] == 1u || st[l] == 2u;
        if (n_active <= refill) {
            bool to_surf[64] = {false}, to_other[64] = {false};
            uint32_t rec[64] = {0};
            for (int l = 0; l < 64; ++l) {
                rec[l] = e[l] & 0xFFFFu;
                if (st[l] != 4u) continue;
                Record &R = B.slab[rec[l]];   // ONE plain store into the record, no load (the kernel: the hit, the shadow result in the sign of t)
                R.hit_unocc = (e[l] & kUnoccBit) ? 1u : 0u;
                R.hit_final = (e[l] & kNextBit) ? 0u : 1u;
                ((e[l] & kNextBit) && R.hit_kind == RING_SURF ? to_surf : to_other)[l] = true;   // (the kind is the traversal's result in the kernel; here the shader chose it)
                st[l] = 0u;
            }
            push(B, RING_SURF, to_surf, rec); push(B, RING_OTHER, to_other, rec);
            uint32_t n_idle = 0;
            for (int l = 0; l < 64; ++l) n_idle += st[l] == 0u;
            bool take = n_idle != 0u;
            if (take && shader_first) {
                const bool work = count(B, RING_SURF) + count(B, RING_OTHER) != 0u || (primaries_left(B) && count(B, RING_FREE) >= 64u);
                const bool room = ring_count(B.trace) + B.trace.reserved.load(std::memory_order_relaxed) + 64u <= B.TC;
                take = !(work && room);
            }
            if (take) {
                Payload got[64];
                const uint32_t n = trace_pop(B, n_idle, got);
                uint32_t rank = 0;
                for (int l = 0; l < 64; ++l) {
                    if (st[l] != 0u) continue;
                    if (rank < n) {
                        e[l] = got[rank].rec_flags;
                        nsteps[l] = got[rank].next_steps;
                        occluded[l] = got[rank].occluded != 0u;
                        if (got[rank].shadow_steps) { steps[l] = got[rank].shadow_steps; st[l] = 1u; }
                        else { steps[l] = nsteps[l]; st[l] = 2u; }
                    }
                    rank++;
                }
            }
        }
        int tracing = 0;
        for (int l = 0; l < 64; ++l) tracing += st[l] == 1u || st[l] == 2u;
        if (!tracing) {
            uint32_t got[64] = {0}, n = 0, first = 0;
            bool fresh = false;
            const bool room = trace_reserve(B);
            if (room) {
                const uint32_t c_surf = count(B, RING_SURF), c_other = count(B, RING_OTHER);
                if (c_surf >= 64u) n = pop(B, RING_SURF, 64u, false, got);
                else if (c_other >= 64u) n = pop(B, RING_OTHER, 64u, false, got);
                if (n == 0u && primaries_left(B)) {
                    if (pop(B, RING_FREE, 64u, true, got) == 64u) {
                        n = admit(B, S, first);
                        fresh = true;
                        if (n == 0u) { bool all[64]; for (int l = 0; l < 64; ++l) all[l] = true; push(B, RING_FREE, all, got); fresh = false; }
                    }
                }
                if (n == 0u && c_surf) n = pop(B, RING_SURF, 64u, false, got);
                if (n == 0u && c_other) n = pop(B, RING_OTHER, 64u, false, got);
                if (n == 0u) trace_unreserve(B);
            }
            if (n == 0u) {
                if (!primaries_left(B) && count(B, RING_FREE) == B.P) break;
                if (++idle_spins > kSpinCap) { B.abort_.store(2u); break; }
                std::this_thread::yield();
                continue;
            }
            idle_spins = 0;
            bool cont[64] = {false}, back[64] = {false};
            Payload pl[64];
            for (uint32_t l = 0; l < 64u; ++l) {
                e[l] = got[l];
                pl[l] = Payload{got[l], 0u, 0u, 0u};
                const bool mine = l < n;
                if (!mine) { back[l] = fresh; continue; }
                Record &R = B.slab[e[l]];
                uint32_t path, bounce;
                uint64_t L;
                if (fresh) { path = first + l; bounce = 0u; L = 0u; }
                else {
                    path = R.path; bounce = R.bounce; L = R.L;
                    if (R.hit_unocc) L = acc(L, R.contrib);   // the light sample of the bounce before this hit
                    if (R.hit_final) { S.result[path] = L; S.done[path].fetch_add(1u); back[l] = true; continue; }
                }
                const Shade s = shade(path, bounce, S.nb);
                L = acc(L, s.add);
                if (s.shadow || s.next) {
                    R.path = path; R.L = L; R.contrib = s.contrib; R.bounce = s.next ? bounce + 1u : bounce;
                    R.hit_kind = shade(path, bounce + 1u, S.nb).kind;
                    pl[l] = Payload{e[l] | (s.next ? kNextBit : 0u), s.shadow ? s.ssteps : 0u, s.nsteps, (s.contrib & 0x100u) ? 0u : 1u};
                    cont[l] = true;
                } else {
                    S.result[path] = L; S.done[path].fetch_add(1u);
                    back[l] = true;
                }
            }
            trace_push(B, cont, pl);
            push(B, RING_FREE, back, e);
            for (int l = 0; l < 64; ++l) st[l] = 0u;
            continue;
        }
        idle_spins = 0;
        for (int l = 0; l < 64; ++l) {
            if (!(st[l] == 1u || st[l] == 2u) || --steps[l] != 0u) continue;
            if (st[l] == 1u) {
                if (!occluded[l]) e[l] |= kUnoccBit;
                if (e[l] & kNextBit) { steps[l] = nsteps[l]; st[l] = 2u; } else st[l] = 4u;
            } else st[l] = 4u;
        }
    }
}

int main(int argc, char **argv) {
    const uint32_t blocks = argc > 1 ? atoi(argv[1]) : 2, waves = argc > 2 ? atoi(argv[2]) : 6, P = argc > 3 ? atoi(argv[3]) : 256;
    const uint32_t paths = argc > 4 ? atoi(argv[4]) : 20000, shaders = argc > 5 ? atoi(argv[5]) : 2;
    const int refill = argc > 6 ? atoi(argv[6]) : 44;
    const uint32_t TC = argc > 7 ? atoi(argv[7]) : 128;
    Shared S(paths);
    S.count = paths; S.chunk = 256; S.nb = 8;
    std::vector<Block> B(blocks);
    for (Block &b : B) {
        b.P = P; b.TC = TC; b.rbuf.assign(RING_COUNT * P, 0); b.slab.assign(P, Record{}); b.tbuf.assign(TC, Payload{});
        b.where = std::vector<std::atomic<int>>(P);
        for (uint32_t i = 0; i < P; ++i) b.where[i].store(RING_FREE);
        for (uint32_t i = 0; i < P; ++i) b.rbuf[RING_FREE * P + i] = (uint16_t)i;
        b.ring[RING_FREE].tail.store(P);
    }
    std::vector<std::thread> th;
    for (uint32_t b = 0; b < blocks; ++b)
        for (uint32_t w = 0; w < waves; ++w) th.emplace_back(wave, std::ref(B[b]), std::ref(S), w, shaders, refill);
    for (auto &t : th) t.join();
    int bad = 0;
    for (Block &b : B) {
        if (b.abort_.load()) { printf("abort word %u\n", b.abort_.load()); bad++; }
        if (count(b, RING_FREE) != P || ring_count(b.trace) || b.trace.reserved.load() || count(b, RING_SURF) || count(b, RING_OTHER)) { printf("rings not at rest\n"); bad++; }
        std::vector<int> seen(P, 0);
        const uint32_t h = b.ring[RING_FREE].head.load();
        for (uint32_t i = 0; i < P; ++i) seen[b.rbuf[RING_FREE * P + ((h + i) & (P - 1u))]]++;
        for (uint32_t i = 0; i < P; ++i) if (seen[i] != 1) { printf("record %u is %d times in FREE\n", i, seen[i]); bad++; break; }
    }
    for (uint32_t p = 0; p < paths; ++p) {
        if (S.done[p].load() != 1u) { printf("path %u finished %u times\n", p, S.done[p].load()); bad++; break; }
        if (S.result[p] != serial(p, S.nb)) { printf("path %u: radiance differs from its serial evaluation\n", p); bad++; break; }
    }
    printf("pool_model: %u blocks x %u waves, %u records, %u TRACE slots, %u paths: %s\n", blocks, waves, P, TC, paths, bad ? "FAILED" : "OK");
    return bad ? 1 : 0;
}
