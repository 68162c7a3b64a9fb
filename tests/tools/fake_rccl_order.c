/* fake_rccl_order.c — TEST of the test stand-in (tests/tools/fake_rccl.c): does it keep the two properties of RCCL that an exchange
 * can get wrong?  Two threads are the two ranks of one communicator on GPU 0.
 *   1. stream order: the receive lands behind the work enqueued before it and ahead of the work enqueued after it, asynchronously;
 *   2. groups defer: an operation recorded inside ncclGroupStart / ncclGroupEnd is enqueued by the OUTERMOST ncclGroupEnd, so a
 *      consumer enqueued inside the bracket runs BEFORE the data arrives (the bug class of ADVICE r02) and reads the old contents.
 * Prints "ok" and exits 0 when both hold.
 * build: gcc -O2 -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include fake_rccl_order.c -o fake_rccl_order -L/opt/rocm/lib -lamdhip64 -ldl -lpthread */
#include <dlfcn.h>
#include <hip/hip_runtime_api.h>
#include <pthread.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

typedef struct { char internal[128]; } ncclUniqueId;
typedef void *ncclComm_t;
static int (*GetUniqueId)(ncclUniqueId *);
static int (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int);
static int (*CommDestroy)(ncclComm_t);
static int (*GroupStart)(void);
static int (*GroupEnd)(void);
static int (*Send)(const void *, size_t, int, int, ncclComm_t, hipStream_t);
static int (*Recv)(void *, size_t, int, int, ncclComm_t, hipStream_t);
enum { ncclUint8 = 1 };
#define N (3u << 20)
#define CHECK(x) do { if ((x) != 0) { fprintf(stderr, "failed: %s (line %d)\n", #x, __LINE__); exit(2); } } while (0)

static ncclUniqueId g_id;
static int g_fail;

static void *sender(void *arg) {
    (void)arg;
    CHECK(hipSetDevice(0));
    ncclComm_t c;
    CHECK(CommInitRank(&c, 2, g_id, 1));
    hipStream_t s;
    CHECK(hipStreamCreate(&s));
    unsigned char *d;
    CHECK(hipMalloc((void **)&d, N));
    for (int round = 0; round < 2; ++round) {
        CHECK(hipMemsetAsync(d, 0x40 + round, N, s));   /* enqueued BEFORE the send: the send must carry it */
        CHECK(Send(d, N, ncclUint8, 0, c, s));
        CHECK(hipMemsetAsync(d, 0xEE, N, s));           /* enqueued AFTER: must not leak into the message */
    }
    CHECK(hipStreamSynchronize(s));
    CHECK(CommDestroy(c));
    return NULL;
}

static int all_equal(const unsigned char *p, unsigned char v) { for (size_t i = 0; i < N; ++i) if (p[i] != v) return 0; return 1; }

static void *receiver(void *arg) {
    (void)arg;
    CHECK(hipSetDevice(0));
    ncclComm_t c;
    CHECK(CommInitRank(&c, 2, g_id, 0));
    hipStream_t s;
    CHECK(hipStreamCreate(&s));
    unsigned char *d, *before, *after;
    CHECK(hipMalloc((void **)&d, N));
    CHECK(hipHostMalloc((void **)&before, N, 0));
    CHECK(hipHostMalloc((void **)&after, N, 0));
    /* round 0, no group: stream order */
    CHECK(hipMemsetAsync(d, 0x11, N, s));
    CHECK(hipMemcpyAsync(before, d, N, hipMemcpyDeviceToHost, s));   /* ahead of the receive: the old contents */
    CHECK(Recv(d, N, ncclUint8, 1, c, s));
    CHECK(hipMemcpyAsync(after, d, N, hipMemcpyDeviceToHost, s));    /* behind it: the message */
    CHECK(hipStreamSynchronize(s));
    if (!all_equal(before, 0x11) || !all_equal(after, 0x40)) { fprintf(stderr, "stream order violated (before %02x, after %02x)\n", before[0], after[0]); g_fail = 1; }
    /* round 1, grouped: the consumer enqueued INSIDE the bracket runs before the data is there */
    CHECK(hipMemsetAsync(d, 0x22, N, s));
    CHECK(GroupStart());
    CHECK(GroupStart());
    CHECK(Recv(d, N, ncclUint8, 1, c, s));
    CHECK(GroupEnd());                                               /* inner end: still nothing enqueued */
    CHECK(hipMemcpyAsync(before, d, N, hipMemcpyDeviceToHost, s));   /* "phase 2 enqueued too early" */
    CHECK(GroupEnd());
    CHECK(hipMemcpyAsync(after, d, N, hipMemcpyDeviceToHost, s));
    CHECK(hipStreamSynchronize(s));
    if (!all_equal(before, 0x22)) { fprintf(stderr, "a grouped receive was enqueued before the outermost ncclGroupEnd (consumer saw %02x)\n", before[0]); g_fail = 1; }
    if (!all_equal(after, 0x41)) { fprintf(stderr, "the grouped receive did not land (saw %02x)\n", after[0]); g_fail = 1; }
    CHECK(CommDestroy(c));
    return NULL;
}

int main(int argc, char **argv) {
    if (argc < 2) { fprintf(stderr, "usage: fake_rccl_order <libfake_rccl.so>\n"); return 2; }
    void *h = dlopen(argv[1], RTLD_NOW | RTLD_LOCAL);
    if (!h) { fprintf(stderr, "%s\n", dlerror()); return 2; }
#define SYM(v, name) do { *(void **)&v = dlsym(h, name); if (!v) { fprintf(stderr, "missing %s\n", name); return 2; } } while (0)
    SYM(GetUniqueId, "ncclGetUniqueId"); SYM(CommInitRank, "ncclCommInitRank"); SYM(CommDestroy, "ncclCommDestroy");
    SYM(GroupStart, "ncclGroupStart"); SYM(GroupEnd, "ncclGroupEnd"); SYM(Send, "ncclSend"); SYM(Recv, "ncclRecv");
    CHECK(GetUniqueId(&g_id));
    pthread_t a, b;
    pthread_create(&a, NULL, receiver, NULL);
    pthread_create(&b, NULL, sender, NULL);
    pthread_join(a, NULL);
    pthread_join(b, NULL);
    if (g_fail) return 1;
    printf("ok\n");
    return 0;
}
