/* multi_gpu.c — the tile-sharded frame through the plain C ABI (include/lpt.h): what a host in any language binds.
 *
 *   multi_gpu <scene.glb> <ranks> [width height frames]
 *
 * Two ways to run N ranks, chosen by the environment:
 *   LPT_RANK / LPT_WORLD / LPT_ID_FILE set  -> ONE rank of an N-process job (one process per GPU): rank 0 writes the 128-byte
 *       RCCL id to LPT_ID_FILE, the others read it; lpt_comm_create (ncclCommInitRank); lpt_renderer_exchange per frame.
 *   otherwise                               -> all N ranks inside this process on device 0 (N sharded renderers,
 *       lpt_renderer_exchange_local): the single-process form, and how the example runs on a one-GPU box.
 * Rank 0 prints a checksum of the presented frame; it does not depend on N (the N-GPU image is the 1-GPU image bit for bit) — and the checksum of
 * the same frame assembled by the HOST-SIDE GATHER: every rank writes its own pixels into one whole frame in host memory
 * (lpt_renderer_read_radiance_owned).  In the N-process form that frame is POSIX shared memory behind lpt_host_frame_* (LPT_FRAME_NAME = "/name":
 * rank 0 creates it BEFORE it publishes the id file, the others attach after reading the id; lpt_host_frame_barrier completes a frame).
 * Build: gcc -std=c11 -Iinclude examples/multi_gpu.c -Lloupiote_amd -lloupiote_hip -Wl,-rpath,$PWD/loupiote_amd -lm -o multi_gpu */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>

#include "lpt.h"

#define CHECK(x) do { int st_ = (x); if (st_ != LPT_OK) { fprintf(stderr, "%s failed (%d): %s\n", #x, st_, lpt_last_error()); return 1; } } while (0)

int main(int argc, char **argv) {
    if (argc < 3) { fprintf(stderr, "usage: %s scene.glb ranks [width height frames]\n", argv[0]); return 2; }
    const int ranks = atoi(argv[2]);
    const uint32_t W = argc > 3 ? (uint32_t)atoi(argv[3]) : 640, H = argc > 4 ? (uint32_t)atoi(argv[4]) : 360;
    const int frames = argc > 5 ? atoi(argv[5]) : 4;
    const char *env_rank = getenv("LPT_RANK"), *env_world = getenv("LPT_WORLD"), *id_file = getenv("LPT_ID_FILE");
    const int multi_process = env_rank && env_world && id_file;
    const int rank = multi_process ? atoi(env_rank) : 0, world = multi_process ? atoi(env_world) : ranks;
    if (ranks < 1 || ranks > 64 || world < 1 || rank < 0 || rank >= world) { fprintf(stderr, "bad rank / world\n"); return 2; }

    lpt_device *dev = NULL;
    CHECK(lpt_device_create(multi_process ? rank : 0, &dev));      /* one process per GPU: device = local rank */
    lpt_scene *scene = NULL;
    CHECK(lpt_scene_create(&scene));
    CHECK(lpt_load_gltf_path(scene, argv[1]));
    lpt_scene_gpu *sg = NULL;
    CHECK(lpt_scene_upload(dev, scene, &sg));                       /* the scene is replicated on every GPU */
    /* camera-to-world, columns = right, up, direction, origin (crates/standalone/src/camera.rs:101-108) */
    const float view[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, -1, 0, 0.0f, 0.6f, 13.5f, 1};

    lpt_comm *comm = NULL;
    lpt_renderer *r[64] = {0};
    const int n_local = multi_process ? 1 : world;
    const char *frame_name = multi_process ? getenv("LPT_FRAME_NAME") : NULL;
    lpt_host_frame *hframe = NULL;
    if (multi_process) {
        unsigned char id[LPT_COMM_ID_BYTES];
        if (rank == 0) {
            if (frame_name) CHECK(lpt_host_frame_create(frame_name, W, H, (uint32_t)world, 0, &hframe));   /* shm + hipHostRegister; exists before the id does */
            CHECK(lpt_comm_unique_id(id));
            char tmp[1024];
            snprintf(tmp, sizeof tmp, "%s.tmp", id_file);
            FILE *f = fopen(tmp, "wb");
            if (!f || fwrite(id, 1, sizeof id, f) != sizeof id) { fprintf(stderr, "cannot write %s\n", tmp); return 1; }
            fclose(f);
            rename(tmp, id_file);                                  /* the out-of-band channel: a file, here */
        } else {
            FILE *f = NULL;
            for (int tries = 0; tries < 600 && !(f = fopen(id_file, "rb")); ++tries) usleep(100000);
            if (!f || fread(id, 1, sizeof id, f) != sizeof id) { fprintf(stderr, "cannot read %s\n", id_file); return 1; }
            fclose(f);
            if (frame_name) CHECK(lpt_host_frame_attach(frame_name, W, H, (uint32_t)world, 0, &hframe));
        }
        CHECK(lpt_comm_create(dev, id, rank, world, &comm));       /* ncclCommInitRank */
    }
    for (int k = 0; k < n_local; ++k) {
        CHECK(lpt_renderer_create(dev, W, H, &r[k]));
        CHECK(lpt_renderer_set_downsample(r[k], 1.0f));
        CHECK(lpt_renderer_resize(r[k], sg, NULL, W, H));
        CHECK(lpt_renderer_set_max_bounces(r[k], 4));
        if (multi_process) CHECK(lpt_renderer_set_comm(r[k], comm));                       /* = set_shard(rank, world, 32, 8) */
        else CHECK(lpt_renderer_set_shard(r[k], (uint32_t)k, (uint32_t)world, 32, 8));
        CHECK(lpt_renderer_set_resources(r[k], sg, NULL));
        CHECK(lpt_renderer_reset_accumulation(r[k]));
        CHECK(lpt_renderer_set_accumulate(r[k], 1));
    }
    for (int f = 0; f < frames; ++f) {
        for (int k = 0; k < n_local; ++k) CHECK(lpt_renderer_raytrace(r[k], view));
        if (multi_process) CHECK(lpt_renderer_exchange(r[0], LPT_EXCHANGE_GATHER_TILES));   /* every rank, same order */
        else CHECK(lpt_renderer_exchange_local(r[0], n_local > 1 ? &r[1] : NULL, n_local - 1));
    }
    /* The HOST-SIDE GATHER, for hosts that consume the frame on the CPU: no exchange on the GPUs — every rank writes its OWNED pixels straight into the one
     * shared frame, each GPU its 1/N over its own PCIe link, and the frame barrier completes it (every rank; frame numbers count from 1). */
    double sum_shared = -1.0;
    if (hframe) {
        float *frame = NULL;
        CHECK(lpt_host_frame_ptr(hframe, &frame));
        CHECK(lpt_renderer_read_radiance_owned(r[0], frame));
        CHECK(lpt_host_frame_barrier(hframe, (uint32_t)rank, 1u, 60000u));
        if (rank == 0) {
            sum_shared = 0.0;
            for (size_t i = 0; i < (size_t)W * H; ++i) sum_shared += frame[4 * i] + frame[4 * i + 1] + frame[4 * i + 2];
        }
        CHECK(lpt_host_frame_barrier(hframe, (uint32_t)rank, 2u, 60000u));   /* nobody unmaps (rank 0: unlinks) before rank 0 has read the frame */
    }
    if (rank == 0) {
        float *img = (float *)malloc(sizeof(float) * 4 * (size_t)W * H);
        CHECK(lpt_renderer_read_radiance(r[0], img));               /* the presented frame: every rank's tiles */
        double sum = 0.0;
        size_t covered = 0;
        for (size_t i = 0; i < (size_t)W * H; ++i) { sum += img[4 * i] + img[4 * i + 1] + img[4 * i + 2]; covered += img[4 * i + 3] == 1.0f; }
        /* the same gather inside ONE process: a page-locked buffer of the host's own (lpt_host_alloc), no barrier needed */
        double sum_owned = sum_shared;
        if (!multi_process) {
            float *frame = NULL;
            CHECK(lpt_host_alloc(sizeof(float) * 4 * (size_t)W * H, (void **)&frame));
            for (size_t i = 0; i < 4 * (size_t)W * H; ++i) frame[i] = -1.0f;
            for (int k = 0; k < n_local; ++k) CHECK(lpt_renderer_read_radiance_owned(r[k], frame));
            sum_owned = 0.0;
            for (size_t i = 0; i < (size_t)W * H; ++i) sum_owned += frame[4 * i] + frame[4 * i + 1] + frame[4 * i + 2];
            CHECK(lpt_host_free(frame));
        }
        printf("{\"ranks\": %d, \"multi_process\": %d, \"width\": %u, \"height\": %u, \"frames\": %d, \"covered\": %zu, \"checksum\": %.9g, \"host_gather_checksum\": %.9g}\n",
               world, multi_process, W, H, frames, covered, sum, sum_owned);
        free(img);
    }
    for (int k = 0; k < n_local; ++k) lpt_renderer_destroy(r[k]);
    if (hframe) lpt_host_frame_destroy(hframe);
    if (comm) lpt_comm_destroy(comm);
    lpt_scene_gpu_destroy(sg);
    lpt_scene_destroy(scene);
    lpt_device_destroy(dev);
    return 0;
}
