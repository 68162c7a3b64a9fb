// hostframe.cpp — the HOST-SIDE GATHER of a tile-sharded frame behind the C ABI (lpt_host_frame_*, include/lpt.h).
//
// One whole frame (w x h x 4 floats) in POSIX shared memory that every rank of a node maps and page-locks; every rank writes its OWN
// pixels into it straight from its GPU (lpt_renderer_read_radiance_owned) and a barrier over a line of progress words completes the
// frame.  On N ranks this replaces the one blocking call the reference ends a frame with, Renderer::read_pixels
// (reference crates/lib/src/renderer.rs:727-811: copy to a staging buffer, map, device.poll(Wait)): nothing is gathered on one GPU
// first, each GPU pushes its 1/N over its own PCIe link.
//
// Segment layout: [Header 64 B][progress line of rank 0][... of rank world-1][the "all arrived" line][pad to 4096][frame].
// Each word sits on its own 64-byte line.  Barrier of frame `no` (1, 2, 3, ... — the caller's frame counter, the same on every rank):
// a rank stores `no` into its word (release); rank 0 waits until every word is >= no, stores `no` into the "all arrived" word and wakes
// the sleepers; the others wait for that word.  Waiting = a bounded run of pause-spins (the barrier closes within microseconds of the
// last rank's arrival in the normal case), then futex sleeps with the caller's timeout: the others on the "all arrived" word, rank 0 on the word of the
// rank it is waiting for (it leaves a note first, so that the arriving rank knows to wake it).
#include <hip/hip_runtime_api.h>

#include <atomic>
#include <cerrno>
#include <climits>
#include <ctime>
#include <fcntl.h>
#include <linux/futex.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <sys/syscall.h>
#include <unistd.h>

#include "common.h"

using namespace lpt;

namespace {
constexpr uint32_t kMagic = 0x4C504846u;   // "LPHF"
struct Header { uint32_t magic, width, height, world, ready, pad[11]; };
static_assert(sizeof(Header) == 64, "one line");
struct alignas(64) Line { std::atomic<uint32_t> word; std::atomic<uint32_t> note; uint32_t pad[14]; };   // note: line 0 only — rank 0's "I sleep on rank q's word" (q + 1; 0 = awake)
static_assert(sizeof(Line) == 64, "one line");
inline size_t frame_offset(uint32_t world) { return ((sizeof(Header) + sizeof(Line) * ((size_t)world + 1u)) + 4095u) & ~(size_t)4095u; }
inline int futex(std::atomic<uint32_t> *addr, int op, uint32_t val, const timespec *ts) {
    return (int)syscall(SYS_futex, reinterpret_cast<uint32_t *>(addr), op, val, ts, nullptr, 0);
}
inline double now_ms() { timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return (double)t.tv_sec * 1e3 + (double)t.tv_nsec * 1e-6; }
}  // namespace

struct lpt_host_frame {
    std::string name;
    void *base = nullptr;
    size_t bytes = 0;
    uint32_t width = 0, height = 0, world = 0;
    bool creator = false, registered = false;
    Header *hdr() const { return reinterpret_cast<Header *>(base); }
    Line *lines() const { return reinterpret_cast<Line *>(reinterpret_cast<unsigned char *>(base) + sizeof(Header)); }
    float *frame() const { return reinterpret_cast<float *>(reinterpret_cast<unsigned char *>(base) + frame_offset(world)); }
};

static int open_frame(const char *name, uint32_t width, uint32_t height, uint32_t world, uint32_t flags, bool create, lpt_host_frame **out) {
    if (!name || name[0] != '/' || strchr(name + 1, '/') || !width || !height || !world || world > 4096u || !out || (flags & ~(uint32_t)LPT_HOST_FRAME_HOST_ONLY))
        return fail(LPT_ERR_INVALID_ARG, "lpt_host_frame_%s: the name must be \"/something\" (shm_open), the size and the world non-zero", create ? "create" : "attach");
    const size_t bytes = frame_offset(world) + sizeof(float) * 4u * (size_t)width * height;
    const int fd = create ? shm_open(name, O_CREAT | O_EXCL | O_RDWR, 0600) : shm_open(name, O_RDWR, 0600);
    if (fd < 0) return fail(LPT_ERR_FILE_NOT_FOUND, "lpt_host_frame_%s: shm_open(%s) failed: %s", create ? "create" : "attach", name, strerror(errno));
    if (create && ftruncate(fd, (off_t)bytes) != 0) {
        const int e = errno;
        close(fd); shm_unlink(name);
        return fail(LPT_ERR_READBACK, "lpt_host_frame_create: ftruncate(%zu) failed: %s", bytes, strerror(e));
    }
    if (!create) {
        struct stat sb;
        if (fstat(fd, &sb) != 0 || (size_t)sb.st_size != bytes) {
            close(fd);
            return fail(LPT_ERR_INVALID_ARG, "lpt_host_frame_attach: %s is not a %ux%u frame for %u ranks (size %lld, expected %zu)", name, width, height, world, (long long)sb.st_size, bytes);
        }
    }
    void *base = mmap(nullptr, bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    const int map_errno = errno;
    close(fd);
    if (base == MAP_FAILED) {
        if (create) shm_unlink(name);
        return fail(LPT_ERR_READBACK, "lpt_host_frame_%s: mmap failed: %s", create ? "create" : "attach", strerror(map_errno));
    }
    lpt_host_frame *f = new lpt_host_frame();
    f->name = name; f->base = base; f->bytes = bytes; f->width = width; f->height = height; f->world = world; f->creator = create;
    if (create) {   // a fresh segment is zero-filled: every progress word starts at 0
        Header *h = f->hdr();
        h->width = width; h->height = height; h->world = world;
        std::atomic_thread_fence(std::memory_order_release);
        h->magic = kMagic;
    } else {
        const Header *h = f->hdr();
        if (h->magic != kMagic || h->width != width || h->height != height || h->world != world) {
            munmap(base, bytes);
            delete f;
            return fail(LPT_ERR_INVALID_ARG, "lpt_host_frame_attach: %s does not hold a %ux%u frame for %u ranks (or its creator has not finished)", name, width, height, world);
        }
    }
    // page-lock + map for this process's GPU: lpt_renderer_read_radiance_owned writes through the mapping.  A participant that only reads the
    // finished frame (a compositor / encoder process without a GPU) attaches with LPT_HOST_FRAME_HOST_ONLY
    const hipError_t e = (flags & LPT_HOST_FRAME_HOST_ONLY) ? hipSuccess : hipHostRegister(base, bytes, hipHostRegisterMapped | hipHostRegisterPortable);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        munmap(base, bytes);
        if (create) shm_unlink(name);
        delete f;
        return fail(LPT_ERR_HIP, "lpt_host_frame_%s: hipHostRegister failed: %s (a device must exist first)", create ? "create" : "attach", hipGetErrorString(e));
    }
    f->registered = !(flags & LPT_HOST_FRAME_HOST_ONLY);
    *out = f;
    return LPT_OK;
}

// the spin loops' pause: the x86 hint, a yield on aarch64, a compiler barrier elsewhere
static inline void cpu_relax() {
#if defined(__x86_64__) || defined(__i386__)
    __builtin_ia32_pause();
#elif defined(__aarch64__)
    asm volatile("yield" ::: "memory");
#else
    asm volatile("" ::: "memory");
#endif
}

extern "C" {

int lpt_host_frame_create(const char *name, uint32_t width, uint32_t height, uint32_t world, uint32_t flags, lpt_host_frame **out) { return open_frame(name, width, height, world, flags, true, out); }
int lpt_host_frame_attach(const char *name, uint32_t width, uint32_t height, uint32_t world, uint32_t flags, lpt_host_frame **out) { return open_frame(name, width, height, world, flags, false, out); }

int lpt_host_frame_ptr(lpt_host_frame *f, float **frame) {
    if (!f || !frame) return fail(LPT_ERR_INVALID_ARG, "lpt_host_frame_ptr: null");
    *frame = f->frame();
    return LPT_OK;
}

int lpt_host_frame_barrier(lpt_host_frame *f, uint32_t rank, uint32_t frame_no, uint32_t timeout_ms) {
    if (!f || rank >= f->world || !frame_no) return fail(LPT_ERR_INVALID_ARG, "lpt_host_frame_barrier: bad rank, or frame number 0 (frames count from 1)");
    Line *ln = f->lines();
    std::atomic<uint32_t> &all = ln[f->world].word;
    // rank 0's "I sleep on word q" note, in its own line's padding: an arriving rank wakes it only then (no syscall per rank and frame in the normal case)
    std::atomic<uint32_t> &sleeping_on = ln[0].note;   // 0 = awake, q + 1 = asleep on rank q's word
    ln[rank].word.store(frame_no, std::memory_order_seq_cst);
    if (rank != 0u && sleeping_on.load(std::memory_order_seq_cst) == rank + 1u) futex(&ln[rank].word, FUTEX_WAKE, 1, nullptr);
    const double t_end = now_ms() + (double)timeout_ms;
    // serial numbers compare modulo 2^32: "word >= frame_no"
    auto reached = [frame_no](uint32_t v) { return (int32_t)(v - frame_no) >= 0; };
    if (rank == 0u) {
        for (uint32_t q = 1; q < f->world; ++q) {
            uint32_t spins = 0;
            for (;;) {
                uint32_t v = ln[q].word.load(std::memory_order_acquire);
                if (reached(v)) break;
                if (++spins < 20000u) { cpu_relax(); continue; }
                const double left = t_end - now_ms();
                if (left <= 0.0) return fail(LPT_ERR_READBACK, "failed to read pixels from GPU to CPU: rank %u did not reach frame %u within %u ms", q, frame_no, timeout_ms);
                // sleep ON rank q's word: say so first, look again (the store / load pairs on both sides are sequentially consistent: either rank q sees the note
                // and wakes us, or we see its word), then wait — woken the moment rank q arrives, not at the end of a polling slice
                sleeping_on.store(q + 1u, std::memory_order_seq_cst);
                v = ln[q].word.load(std::memory_order_seq_cst);
                if (!reached(v)) {
                    timespec ts;
                    const double slice = left < 50.0 ? left : 50.0;
                    ts.tv_sec = 0; ts.tv_nsec = (long)(slice * 1e6);
                    futex(&ln[q].word, FUTEX_WAIT, v, &ts);
                }
                sleeping_on.store(0u, std::memory_order_seq_cst);
            }
        }
        all.store(frame_no, std::memory_order_release);
        futex(&all, FUTEX_WAKE, INT_MAX, nullptr);
        return LPT_OK;
    }
    uint32_t spins = 0;
    for (;;) {
        const uint32_t v = all.load(std::memory_order_acquire);
        if (reached(v)) return LPT_OK;
        if (++spins < 20000u) { cpu_relax(); continue; }
        const double left = t_end - now_ms();
        if (left <= 0.0) return fail(LPT_ERR_READBACK, "failed to read pixels from GPU to CPU: frame %u was not completed within %u ms", frame_no, timeout_ms);
        timespec ts;
        const double slice = left < 50.0 ? left : 50.0;
        ts.tv_sec = 0; ts.tv_nsec = (long)(slice * 1e6);
        futex(&all, FUTEX_WAIT, v, &ts);   // returns at once when the word has moved on
    }
}

int lpt_host_frame_destroy(lpt_host_frame *f) {
    if (!f) return LPT_OK;
    if (f->registered) { if (hipHostUnregister(f->base) != hipSuccess) (void)hipGetLastError(); }
    munmap(f->base, f->bytes);
    if (f->creator) shm_unlink(f->name.c_str());
    delete f;
    return LPT_OK;
}

}  // extern "C"
