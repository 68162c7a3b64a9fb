/* fake_rccl.c — TEST INFRASTRUCTURE, not a collective library.
 *
 * A stand-in for the eleven librccl entry points libloupiote_hip.so resolves with dlsym (csrc/device.hip: struct Rccl), so that
 * the MULTI-PROCESS side of the frame exchange — N ranks in N processes, bench.py's launcher and control flow, the send / recv
 * sizes and staging offsets of every rank — can run on a box with ONE GPU, where real RCCL refuses two ranks on one device.
 * Selected with LPT_RCCL_LIBRARY=<this .so> (tests/test_gpu_multiproc.py); never loaded otherwise.
 *
 * Semantics it keeps of the real thing (round 4; round 3's stand-in moved the bytes synchronously inside the call and so could not
 * show an ordering bug):
 *   * ASYNCHRONOUS and STREAM-ORDERED: ncclSend / ncclRecv / ncclReduce only enqueue on the stream they are given — a device-to-host
 *     copy into page-locked staging, a host function (hipLaunchHostFunc) that moves the bytes through the mailbox, a host-to-device
 *     copy — and return.  Work enqueued on that stream BEFORE the call runs before the transfer, work enqueued AFTER it runs behind it,
 *     and nothing else is ordered: a consumer kernel launched on another stream, or before the operation was enqueued, reads stale data.
 *   * GROUPS DEFER: between ncclGroupStart and the OUTERMOST ncclGroupEnd of a thread nothing is enqueued; the recorded operations are
 *     enqueued, in call order, by that ncclGroupEnd.  Whatever the caller enqueues inside the bracket behind an operation therefore
 *     runs BEFORE it (the bug class of ADVICE r02: an unpack kernel enqueued inside an open bracket).
 * Transport: a POSIX shared-memory segment named after the unique id, one mailbox per ordered (src, dst) pair, 1 MiB chunks.
 * Sum reductions (float32 / int32) are done on the host by the root.  It says nothing about RCCL's performance or its
 * topology handling — that is what the 8-GPU run is for.
 *
 * build: gcc -O2 -shared -fPIC -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include fake_rccl.c -o libfake_rccl.so -L/opt/rocm/lib -lamdhip64 -lrt
 */
#define _GNU_SOURCE
#include <fcntl.h>
#include <hip/hip_runtime_api.h>
#include <stdatomic.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>
#include <time.h>
#include <unistd.h>

typedef enum { ncclSuccess = 0, ncclUnhandledCudaError = 1, ncclSystemError = 2, ncclInternalError = 3, ncclInvalidArgument = 4 } ncclResult_t;
typedef enum { ncclInt8 = 0, ncclUint8 = 1, ncclInt32 = 2, ncclUint32 = 3, ncclInt64 = 4, ncclUint64 = 5, ncclFloat16 = 6, ncclFloat32 = 7, ncclFloat64 = 8 } ncclDataType_t;
typedef enum { ncclSum = 0 } ncclRedOp_t;
typedef struct { char internal[128]; } ncclUniqueId;

#define MAX_RANKS 8
typedef struct {
    _Atomic uint64_t sent, taken;   /* messages published / consumed; message k of the pair lives in its own shared-memory object */
} mailbox;
typedef struct {
    _Atomic uint32_t ready, arrived;
    uint32_t world;
    mailbox box[MAX_RANKS * MAX_RANKS];   /* [src * MAX_RANKS + dst] */
} segment;
typedef struct fake_comm { segment *seg; int rank, world; char name[80]; } *ncclComm_t;

static size_t type_size(ncclDataType_t t) {
    switch (t) { case ncclInt8: case ncclUint8: return 1; case ncclFloat16: return 2; case ncclInt32: case ncclUint32: case ncclFloat32: return 4; default: return 8; }
}
static void nap(void) { struct timespec ts = {0, 20000}; nanosleep(&ts, NULL); }

const char *ncclGetErrorString(ncclResult_t r) { return r == ncclSuccess ? "no error" : "fake_rccl error (test stand-in)"; }

ncclResult_t ncclGetUniqueId(ncclUniqueId *id) {
    memset(id, 0, sizeof *id);
    struct timespec ts;
    clock_gettime(CLOCK_REALTIME, &ts);
    static _Atomic uint32_t counter;
    snprintf(id->internal, sizeof id->internal, "lptfake_%d_%ld_%ld_%u", (int)getpid(), (long)ts.tv_sec, (long)ts.tv_nsec, atomic_fetch_add(&counter, 1u));
    return ncclSuccess;
}

ncclResult_t ncclCommInitRank(ncclComm_t *out, int world, ncclUniqueId id, int rank) {
    if (getenv("FAKE_RCCL_HANG_INIT")) for (;;) nap();   /* tests: an RCCL bring-up that never returns (bench.py --exchange auto must go on without it) */
    if (world < 1 || world > MAX_RANKS || rank < 0 || rank >= world) return ncclInvalidArgument;
    struct fake_comm *c = (struct fake_comm *)calloc(1, sizeof *c);
    snprintf(c->name, sizeof c->name, "/%.*s", 70, id.internal);
    int fd = -1;
    if (rank == 0) {
        fd = shm_open(c->name, O_CREAT | O_EXCL | O_RDWR, 0600);
        if (fd < 0 || ftruncate(fd, (off_t)sizeof(segment)) != 0) { free(c); return ncclSystemError; }
    } else {
        for (int tries = 0; tries < 500000 && fd < 0; ++tries) { fd = shm_open(c->name, O_RDWR, 0600); if (fd < 0) nap(); }
        if (fd < 0) { free(c); return ncclSystemError; }
        for (int tries = 0; tries < 500000; ++tries) {   /* wait for rank 0's ftruncate */
            off_t sz = lseek(fd, 0, SEEK_END);
            if (sz >= (off_t)sizeof(segment)) break;
            nap();
        }
    }
    c->seg = (segment *)mmap(NULL, sizeof(segment), PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    close(fd);
    if (c->seg == MAP_FAILED) { free(c); return ncclSystemError; }
    c->rank = rank; c->world = world;
    if (rank == 0) { c->seg->world = (uint32_t)world; atomic_store(&c->seg->ready, 1u); }
    while (!atomic_load(&c->seg->ready)) nap();
    atomic_fetch_add(&c->seg->arrived, 1u);
    while (atomic_load(&c->seg->arrived) < (uint32_t)world) nap();   /* ncclCommInitRank is a collective */
    *out = c;
    return ncclSuccess;
}
ncclResult_t ncclCommDestroy(ncclComm_t c) {
    if (!c) return ncclSuccess;
    if (c->rank == 0) shm_unlink(c->name);
    munmap(c->seg, sizeof(segment));
    free(c);
    return ncclSuccess;
}
ncclResult_t ncclCommCount(const ncclComm_t c, int *n) { *n = c->world; return ncclSuccess; }
ncclResult_t ncclCommUserRank(const ncclComm_t c, int *r) { *r = c->rank; return ncclSuccess; }
/* ---- host halves of a message (run inside stream host functions: no HIP calls here) */
/* Sends are EAGER: message k from src to dst is a shared-memory object of its own ("<id>_<src>_<dst>_<k>", 8-byte length + payload),
 * so a sender never waits for its receiver — only receives block, and only for data.  With one host-function thread per process that
 * rules out the cross-communicator deadlock a rendezvous mailbox would have when frames are in flight on several communicators. */
static void message_name(char *out, size_t cap, ncclComm_t c, int src, int dst, uint64_t k) { snprintf(out, cap, "%.70s_%d_%d_%llu", c->name, src, dst, (unsigned long long)k); }
static int put_host(ncclComm_t c, int dst, const unsigned char *src, size_t bytes) {
    mailbox *m = &c->seg->box[c->rank * MAX_RANKS + dst];
    char name[128];
    message_name(name, sizeof name, c, c->rank, dst, atomic_load(&m->sent));
    const int fd = shm_open(name, O_CREAT | O_EXCL | O_RDWR, 0600);
    if (fd < 0 || ftruncate(fd, (off_t)(bytes + 8)) != 0) { if (fd >= 0) close(fd); return 1; }
    unsigned char *p = (unsigned char *)mmap(NULL, bytes + 8, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    close(fd);
    if (p == MAP_FAILED) return 1;
    const uint64_t n = bytes;
    memcpy(p, &n, 8);
    if (bytes) memcpy(p + 8, src, bytes);
    munmap(p, bytes + 8);
    atomic_fetch_add(&m->sent, 1u);
    return 0;
}
static int get_host(ncclComm_t c, int src, unsigned char *dst, size_t bytes) {
    mailbox *m = &c->seg->box[src * MAX_RANKS + c->rank];
    while (atomic_load(&m->sent) == atomic_load(&m->taken)) nap();
    char name[128];
    message_name(name, sizeof name, c, src, c->rank, atomic_load(&m->taken));
    const int fd = shm_open(name, O_RDWR, 0600);
    if (fd < 0) return 1;
    uint64_t n = 0;
    int bad = 0;
    if (pread(fd, &n, 8, 0) != 8 || n != bytes) bad = 1;   /* the two sides disagree about the size: exactly what this stand-in is for */
    if (!bad && bytes) {
        unsigned char *p = (unsigned char *)mmap(NULL, bytes + 8, PROT_READ, MAP_SHARED, fd, 0);
        if (p == MAP_FAILED) bad = 1;
        else { memcpy(dst, p + 8, bytes); munmap(p, bytes + 8); }
    }
    close(fd);
    shm_unlink(name);
    atomic_fetch_add(&m->taken, 1u);
    return bad;
}

/* ---- one operation: recorded by the call, enqueued on its stream at once or by the outermost ncclGroupEnd */
enum { OP_SEND, OP_RECV, OP_REDUCE };
typedef struct op {
    int kind, peer;
    ncclComm_t comm;
    const void *src;
    void *dst;
    size_t bytes, count;
    ncclDataType_t type;
    hipStream_t stream;
    unsigned char *stage, *tmp;   /* page-locked staging; tmp: the root's receive buffer of a reduce */
} op;
static _Atomic int g_transfer_errors;   /* a failed transfer inside a host function: reported by the next call */

static void host_half(void *arg) {
    op *o = (op *)arg;
    int bad = 0;
    if (o->kind == OP_SEND) bad = put_host(o->comm, o->peer, o->stage, o->bytes);
    else if (o->kind == OP_RECV) bad = get_host(o->comm, o->peer, o->stage, o->bytes);
    else {   /* the root of a reduce: own contribution is in stage already */
        for (int q = 0; q < o->comm->world && !bad; ++q) {   /* rank order: x + 0 = x makes the order irrelevant for disjoint tiles */
            if (q == o->peer) continue;
            bad = get_host(o->comm, q, o->tmp, o->bytes);
            if (bad) break;
            if (o->type == ncclFloat32) for (size_t i = 0; i < o->count; ++i) ((float *)o->stage)[i] += ((const float *)o->tmp)[i];
            else for (size_t i = 0; i < o->count; ++i) ((int32_t *)o->stage)[i] += ((const int32_t *)o->tmp)[i];
        }
    }
    if (bad) atomic_fetch_add(&g_transfer_errors, 1);
}
static void release_op(void *arg) {   /* last node of an operation on its stream */
    op *o = (op *)arg;
    /* the staging buffers are page-locked allocations: freeing them needs a HIP call, which a host function must not make.
     * They are handed to a list that the next call on this thread frees. */
    extern void fake_retire(op *o);
    fake_retire(o);
}
static op *g_retired[4096];
static _Atomic int g_n_retired;
void fake_retire(op *o) { const int k = atomic_fetch_add(&g_n_retired, 1); if (k < 4096) g_retired[k] = o; }
static void collect_retired(void) {
    static _Atomic int busy;
    int expected = 0;
    if (!atomic_compare_exchange_strong(&busy, &expected, 1)) return;
    const int n = atomic_load(&g_n_retired);
    if (n >= 64) {   /* only when a few have gathered: every operation in the list has finished its last stream node */
        for (int k = 0; k < n && k < 4096; ++k) {
            op *o = g_retired[k];
            if (!o) continue;
            if (o->stage) hipHostFree(o->stage);
            if (o->tmp) hipHostFree(o->tmp);
            free(o);
            g_retired[k] = NULL;
        }
        atomic_store(&g_n_retired, 0);
    }
    atomic_store(&busy, 0);
}

static ncclResult_t enqueue(op *o) {
    hipStream_t s = o->stream;
    const size_t n = o->bytes ? o->bytes : 4;
    if (hipHostMalloc((void **)&o->stage, n, hipHostMallocDefault) != hipSuccess) return ncclUnhandledCudaError;
    if (o->kind == OP_REDUCE && hipHostMalloc((void **)&o->tmp, n, hipHostMallocDefault) != hipSuccess) return ncclUnhandledCudaError;
    if (o->kind != OP_RECV && o->bytes && hipMemcpyAsync(o->stage, o->src, o->bytes, hipMemcpyDeviceToHost, s) != hipSuccess) return ncclUnhandledCudaError;
    if (hipLaunchHostFunc(s, host_half, o) != hipSuccess) return ncclUnhandledCudaError;
    if (o->kind != OP_SEND && o->bytes && hipMemcpyAsync(o->dst, o->stage, o->bytes, hipMemcpyHostToDevice, s) != hipSuccess) return ncclUnhandledCudaError;
    if (hipLaunchHostFunc(s, release_op, o) != hipSuccess) return ncclUnhandledCudaError;
    return ncclSuccess;
}

static __thread int t_depth;
static __thread op *t_pending[1024];
static __thread int t_n_pending;

static ncclResult_t submit(op *o) {
    if (atomic_load(&g_transfer_errors)) { free(o); return ncclInternalError; }
    collect_retired();
    if (t_depth > 0) {
        if (t_n_pending >= 1024) { free(o); return ncclInternalError; }
        t_pending[t_n_pending++] = o;   /* issued by the outermost ncclGroupEnd */
        return ncclSuccess;
    }
    return enqueue(o);
}
ncclResult_t ncclGroupStart(void) { ++t_depth; return ncclSuccess; }
ncclResult_t ncclGroupEnd(void) {
    if (t_depth <= 0) return ncclInvalidArgument;
    if (--t_depth > 0) return ncclSuccess;
    ncclResult_t st = ncclSuccess;
    for (int k = 0; k < t_n_pending; ++k) {
        if (st == ncclSuccess) st = enqueue(t_pending[k]);
        else free(t_pending[k]);
    }
    t_n_pending = 0;
    return st;
}

static op *new_op(int kind, int peer, ncclComm_t c, const void *src, void *dst, size_t count, ncclDataType_t t, hipStream_t s) {
    op *o = (op *)calloc(1, sizeof *o);
    o->kind = kind; o->peer = peer; o->comm = c; o->src = src; o->dst = dst; o->count = count; o->type = t; o->bytes = count * type_size(t); o->stream = s;
    return o;
}
ncclResult_t ncclSend(const void *buf, size_t count, ncclDataType_t t, int peer, ncclComm_t c, hipStream_t s) {
    if (peer < 0 || peer >= c->world) return ncclInvalidArgument;
    return submit(new_op(OP_SEND, peer, c, buf, NULL, count, t, s));
}
ncclResult_t ncclRecv(void *buf, size_t count, ncclDataType_t t, int peer, ncclComm_t c, hipStream_t s) {
    if (peer < 0 || peer >= c->world) return ncclInvalidArgument;
    return submit(new_op(OP_RECV, peer, c, NULL, buf, count, t, s));
}
ncclResult_t ncclReduce(const void *sendbuf, void *recvbuf, size_t count, ncclDataType_t t, ncclRedOp_t op_, int root, ncclComm_t c, hipStream_t s) {
    if (op_ != ncclSum || (t != ncclFloat32 && t != ncclInt32)) return ncclInvalidArgument;
    if (c->rank != root) return submit(new_op(OP_SEND, root, c, sendbuf, NULL, count, t, s));
    return submit(new_op(OP_REDUCE, root, c, sendbuf, recvbuf, count, t, s));
}
