/* fake_rccl.c — TEST INFRASTRUCTURE, not a collective library.
 *
 * A stand-in for the eleven librccl entry points libloupiote_hip.so resolves with dlsym (csrc/device.hip: struct Rccl), so that
 * the MULTI-PROCESS side of the frame exchange — N ranks in N processes, bench.py's launcher and control flow, the send / recv
 * sizes and staging offsets of every rank — can run on a box with ONE GPU, where real RCCL refuses two ranks on one device.
 * Selected with LPT_RCCL_LIBRARY=<this .so> (tests/test_gpu_multiproc.py); never loaded otherwise.
 *
 * Transport: a POSIX shared-memory segment named after the unique id, one mailbox per ordered (src, dst) pair, 1 MiB chunks,
 * everything synchronous: an operation first waits for the stream it was enqueued on, then moves the bytes with hipMemcpy
 * from the calling thread.  Group brackets are no-ops.  Sum reductions (float32 / int32) are done on the host by the root.
 * It says nothing about RCCL itself — that is what the 8-GPU run is for.
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

#define CHUNK (1u << 20)
#define MAX_RANKS 8
typedef struct {
    _Atomic uint64_t sent, taken;   /* chunks published / consumed */
    uint64_t bytes;                 /* payload of the chunk in flight */
    unsigned char data[CHUNK];
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
ncclResult_t ncclGroupStart(void) { return ncclSuccess; }
ncclResult_t ncclGroupEnd(void) { return ncclSuccess; }

/* host-side halves of a message: `dev` = device pointer (copied chunk by chunk) or, with dev == NULL, `host` */
static ncclResult_t put(ncclComm_t c, int dst, const void *dev, size_t bytes) {
    mailbox *m = &c->seg->box[c->rank * MAX_RANKS + dst];
    size_t off = 0;
    do {
        const size_t n = bytes - off < CHUNK ? bytes - off : CHUNK;
        while (atomic_load(&m->sent) != atomic_load(&m->taken)) nap();
        if (n && hipMemcpy(m->data, (const char *)dev + off, n, hipMemcpyDeviceToHost) != hipSuccess) return ncclUnhandledCudaError;
        m->bytes = n;
        atomic_fetch_add(&m->sent, 1u);
        off += n;
    } while (off < bytes);
    return ncclSuccess;
}
static ncclResult_t get(ncclComm_t c, int src, void *dev, void *host, size_t bytes) {
    mailbox *m = &c->seg->box[src * MAX_RANKS + c->rank];
    size_t off = 0;
    do {
        while (atomic_load(&m->sent) == atomic_load(&m->taken)) nap();
        const size_t n = m->bytes;
        if (off + n > bytes) return ncclInternalError;   /* the two sides disagree about the size: exactly what this stand-in is for */
        if (n) {
            if (dev) { if (hipMemcpy((char *)dev + off, m->data, n, hipMemcpyHostToDevice) != hipSuccess) return ncclUnhandledCudaError; }
            else memcpy((char *)host + off, m->data, n);
        }
        atomic_fetch_add(&m->taken, 1u);
        off += n;
        if (n < CHUNK && off < bytes) return ncclInternalError;   /* short message */
    } while (off < bytes);
    return ncclSuccess;
}

ncclResult_t ncclSend(const void *buf, size_t count, ncclDataType_t t, int peer, ncclComm_t c, hipStream_t s) {
    if (hipStreamSynchronize(s) != hipSuccess) return ncclUnhandledCudaError;
    return put(c, peer, buf, count * type_size(t));
}
ncclResult_t ncclRecv(void *buf, size_t count, ncclDataType_t t, int peer, ncclComm_t c, hipStream_t s) {
    if (hipStreamSynchronize(s) != hipSuccess) return ncclUnhandledCudaError;
    return get(c, peer, buf, NULL, count * type_size(t));
}
ncclResult_t ncclReduce(const void *sendbuf, void *recvbuf, size_t count, ncclDataType_t t, ncclRedOp_t op, int root, ncclComm_t c, hipStream_t s) {
    if (op != ncclSum || (t != ncclFloat32 && t != ncclInt32)) return ncclInvalidArgument;
    if (hipStreamSynchronize(s) != hipSuccess) return ncclUnhandledCudaError;
    const size_t bytes = count * 4;
    if (c->rank != root) return put(c, root, sendbuf, bytes);
    void *acc = malloc(bytes ? bytes : 4), *tmp = malloc(bytes ? bytes : 4);
    ncclResult_t st = ncclSuccess;
    if (hipMemcpy(acc, sendbuf, bytes, hipMemcpyDeviceToHost) != hipSuccess) st = ncclUnhandledCudaError;
    for (int q = 0; q < c->world && st == ncclSuccess; ++q) {   /* rank order: x + 0 = x makes the order irrelevant for disjoint tiles */
        if (q == root) continue;
        st = get(c, q, NULL, tmp, bytes);
        if (st != ncclSuccess) break;
        if (t == ncclFloat32) for (size_t i = 0; i < count; ++i) ((float *)acc)[i] += ((const float *)tmp)[i];
        else for (size_t i = 0; i < count; ++i) ((int32_t *)acc)[i] += ((const int32_t *)tmp)[i];
    }
    if (st == ncclSuccess && hipMemcpy(recvbuf, acc, bytes, hipMemcpyHostToDevice) != hipSuccess) st = ncclUnhandledCudaError;
    free(acc); free(tmp);
    return st;
}
